#!/bin/bash
# round 3, GPU session 2: parity suite (all failures listed), ACDC gradient diagnostic, in-step A/B of the fixed switches + k_c3g.
out=$PWD/gpurun_out/s2; mkdir -p $out
R=$PWD
( time python -m pytest tests -m gpu -q ) > $out/pytest_gpu.txt 2>&1
tail -8 $out/pytest_gpu.txt
( DIAG_FP64=1 python tools/diag/acdc_grad_diag.py 24 12 ) > $out/diag_b24.txt 2>&1; cat $out/diag_b24.txt | tail -80
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
{
for rep in 1 2; do
  echo "rep $rep all_off        $(ab --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_xcd=0 --opt conv3_b6_flatd=0)"
  echo "rep $rep xcd_only       $(ab --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_b6_flatd=0)"
  echo "rep $rep xcd+bwdstats   $(ab --opt norm_small=0 --opt conv3_b6_flatd=0)"
  echo "rep $rep xcd+bwd+small  $(ab --opt conv3_b6_flatd=0)"
  echo "rep $rep all_on         $(ab)"
  echo "rep $rep all_on_sk2     $(ab --opt conv3_b6_flat_sk=2)"
  echo "rep $rep all_on_sk4     $(ab --opt conv3_b6_flat_sk=4)"
  echo "rep $rep all_on_flatd2  $(ab --opt conv3_b6_flatd=2)"
  echo "rep $rep all_on_flatd4  $(ab --opt conv3_b6_flatd=4)"
  echo "rep $rep flatd_nosmall  $(ab --opt norm_small=0)"
done
echo "acdc all_off  $(ab --workload acdc --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_xcd=0 --opt conv3_b6_flatd=0)"
echo "acdc all_on   $(ab --workload acdc)"
echo "panc all_off  $(ab --workload pancreas --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_xcd=0 --opt conv3_b6_flatd=0)"
echo "panc all_on   $(ab --workload pancreas)"
} > $out/ab.txt 2>&1
cat $out/ab.txt
python tools/bench_conv.py --levels 128,256 --ops fwd_chain,bwd_chain,dgrad --json $out/bench_conv_deep.json --variants "r2chain:norm_small=0,conv3_b6_flatd=0;small_c3f:conv3_b6_flatd=0;small_c3g:;c3g_sk2:conv3_b6_flat_sk=2;c3g_sk4:conv3_b6_flat_sk=4;c3g_sk8:conv3_b6_flat_sk=8;c3g_mt2:conv3_b6_flatd=2;c3g_mt4:conv3_b6_flatd=4" > $out/bench_conv_deep.txt 2>&1; cat $out/bench_conv_deep.txt
python tools/bench_conv.py --levels 16,32,64 --ops fwd_chain,bwd_chain,fwd_stats,dgrad,wgrad --json $out/bench_conv_mid.json --variants "r2:norm_small=0,conv3_b6_flatd=0,conv3_xcd=0,fuse_bwd_stats=0;xcd:fuse_bwd_stats=0;xcd_bwdstats:;xcd2:conv3_xcd=2" > $out/bench_conv_mid.txt 2>&1; cat $out/bench_conv_mid.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; head -14 $out/kernel_stats.csv | cut -c1-170
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/timeline.json > $out/timeline.txt; head -50 $out/timeline.txt
