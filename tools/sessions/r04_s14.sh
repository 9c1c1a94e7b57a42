#!/bin/bash
# round 4, GPU session 14: where the backward-statistics epilogue's +35 us go (ablation builds of conv3b.hip: BCP_BW_ABLATE)
out=$PWD/gpurun_out/r04_s14; mkdir -p $out
for lib in "" tools/_abl/libbcp_bw1.so tools/_abl/libbcp_bw2.so tools/_abl/libbcp_bw3.so tools/_abl/libbcp_bw4.so; do
  echo "== lib ${lib:-product}"
  python tools/bench_conv.py --levels 32,64 --ops dgrad,bwd_chain --variants "d:" --json $out/c.json ${lib:+--lib $lib} 2>&1 | grep -v "amdgpu\|fp32" 
done | tee $out/abl.txt
