#!/bin/bash
out=$PWD/gpurun_out/s4; mkdir -p $out
python tools/step_segments2.py > $out/segments.txt 2>&1; cat $out/segments.txt | tail -14
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
{
for rep in 1 2; do
  echo "rep $rep default          $(ab)"
  echo "rep $rep teacher_prio-1   $(ab --opt teacher_prio=-1)"
  echo "rep $rep wgrad_prio-1     $(ab --opt wgrad_prio=-1)"
  echo "rep $rep both-1           $(ab --opt teacher_prio=-1 --opt wgrad_prio=-1)"
  echo "rep $rep teacher_prio1    $(ab --opt teacher_prio=1)"
  echo "rep $rep graphs2          $(ab --opt graphs=2)"
done
} > $out/ab.txt 2>&1
cat $out/ab.txt
