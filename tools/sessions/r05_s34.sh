#!/bin/bash
# round 5, GPU session 34: the FINAL build on a fresh box -- smoke(), the -m gpu suite twice, the driver's default bench command, and the
# rocprofv3 kernel statistics of the LA and ACDC bench commands
out=$PWD/gpurun_out/r05_s34; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $out/smoke.txt
for i in 1 2; do ( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED" | tee -a $out/pytest.txt; done
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | grep real | tee $out/bench_time.txt; cut -c1-420 $out/bench_default.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
for w in la acdc; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks$w -o ev --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ks$w.log 2>&1
  f=$(find /tmp/ks$w -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$w.csv; head -4 $out/kernel_stats_$w.csv | cut -c1-160
done
cd $R; bash tools/probe/boxinfo.sh > $out/box.txt 2>&1
