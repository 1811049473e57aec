#!/bin/bash
# round 5, call 18: renderKernelFlatK at 512 / 768 / 1024 lanes per workgroup (MCRT_FLAT_BLOCK) on C2 and C2-GGX
mkdir -p gpurun_out/r05
for wl in c2 c2_ggx; do
timeout 300 python tools/ab_probe.py $wl --steps 3 "karg512:" "karg768:MCRT_FLAT_BLOCK=768" "karg1024:MCRT_FLAT_BLOCK=1024" "karg512:" "karg768:MCRT_FLAT_BLOCK=768" "karg1024:MCRT_FLAT_BLOCK=1024" 2>&1 | grep '^{' | cut -c1-160 | sed "s/^/$wl /" | tee -a gpurun_out/r05/ab_c2_flat_karg_block.log
done
