#!/bin/bash
out=$PWD/gpurun_out/r06_s20; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
