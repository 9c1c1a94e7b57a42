#!/bin/bash
# round 6, GPU session 1: the driver's bench command (is the last stdout line a parseable < 4 KB record?), the -m gpu suite, LA kernel statistics
out=$PWD/gpurun_out/r06_s1; mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.out 2> $out/bench.err; echo "bench rc $?"
tail -1 $out/bench.out | python3 -c "import sys,json; l=sys.stdin.read().strip(); d=json.loads(l); print('LINE', len(l), 'bytes; keys', sorted(d)); print(l)"
cp gpurun_out/bench_detail.json $out/bench_detail.json 2>/dev/null
( time python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; head -30 $out/kernel_stats.csv | cut -c1-200
