#!/bin/bash
# round 6, GPU session 21: the transposed conv + norm with its output recomputed instead of stored (bcp_up_fwd_norm / bcp_up_norm_bwd): the
# device suite, then the step A/B over the three modes of up_recompute
out=$PWD/gpurun_out/r06_s21; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
WL="la pancreas" tools/ab_opts.sh "" "--opt up_recompute=0" "--opt up_recompute=1" 2>&1 | tee $out/ab.txt
