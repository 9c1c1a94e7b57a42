#!/bin/bash
# round 5, GPU session 17: the final build -- smoke(), the -m gpu suite three times in a row, the driver's default bench command
out=$PWD/gpurun_out/r05_s17; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $out/smoke.txt
for i in 1 2 3; do ( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED" | tee -a $out/pytest.txt; done
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | grep real | tee $out/bench_time.txt; cut -c1-420 $out/bench_default.json
