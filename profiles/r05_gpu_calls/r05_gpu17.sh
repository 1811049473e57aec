#!/bin/bash
# round 5, call 17: renderKernelFlatK at 768 lanes (3 waves per SIMD, 168 VGPRs, 114 spilled) against 512 (256 VGPRs, 13 spilled) on C2
mkdir -p gpurun_out/r05
LIB=monte-carlo-ray-tracer_amd/csrc/libmcrt_hip.so
cp $LIB /tmp/lib_orig.so; cp tools/_build/libflat768.so $LIB
timeout 300 python tools/ab_probe.py c2 --steps 3 "karg512:" "karg768:MCRT_FLAT_BLOCK=768" "karg512:" "karg768:MCRT_FLAT_BLOCK=768" 2>&1 | grep '^{' | cut -c1-160 | tee gpurun_out/r05/ab_c2_flat_karg_768.log
cp /tmp/lib_orig.so $LIB
