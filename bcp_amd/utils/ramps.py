"""The consistency ramp of the reference (utils/ramps.py:19-26): exp(-5 (1 - t)^2) with t = clip(step / length, 0, 1).  The train
scripts compute and log it; it never multiplies a loss term of the BCP step."""
import math


def sigmoid_rampup(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    t = min(max(float(current), 0.0), float(rampup_length)) / float(rampup_length)
    return math.exp(-5.0 * (1.0 - t) ** 2)
