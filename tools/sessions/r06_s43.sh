#!/bin/bash
# round 6, GPU session 43 (run three times, three boxes): smoke() and the driver's default bench command on the final tree
out=$PWD/gpurun_out/r06_s43; mkdir -p $out
tag=$(date +%H%M%S)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke_$tag.txt
( time python bench.py ) 2> $out/bench_$tag.err | tail -1 | tee $out/bench_$tag.json | cut -c1-400
grep -E "^real" $out/bench_$tag.err
bash tools/probe/boxinfo.sh 2>/dev/null | grep -E "Unique ID" | tee $out/box_$tag.txt
