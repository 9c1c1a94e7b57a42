#!/bin/bash
# measurement only: interleaved A/B of two builds of libbcp_hip.so on the same box: tools/ab_libs.sh old.so new.so [bench args]
A=$1; B=$2; shift 2
cp bcp_amd/csrc/libbcp_hip.so /tmp/libbcp_keep.so
for r in 1 2 3; do
  for v in "$A" "$B"; do
    cp "$v" bcp_amd/csrc/libbcp_hip.so
    echo -n "$(basename $v) "; python bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
cp /tmp/libbcp_keep.so bcp_amd/csrc/libbcp_hip.so
