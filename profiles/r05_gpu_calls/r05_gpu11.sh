#!/bin/bash
# round 5, call 11: (a) tools/shared_gpu_stress.py - 3 processes x 2 contexts on the one GPU, 100 frames each, WITHOUT the per-device
# ordering of frames (MCRT_DEVICE_ORDER=0), every frame compared with the reference's golden radiance; then once with the ordering;
# (b) renderKernelPM at 768 lanes (3 waves per SIMD, 170 VGPRs: tools/_build/libpm768.so) against 1024 on the C5 probe and pm
mkdir -p gpurun_out/r05
date
timeout 900 python tools/shared_gpu_stress.py --procs 3 --contexts 2 --frames 100 --order 0 2>&1 | grep '^{' | tee gpurun_out/r05/shared_gpu_stress.log
timeout 600 python tools/shared_gpu_stress.py --procs 3 --contexts 2 --frames 30 --order 1 2>&1 | grep '^{' | tee -a gpurun_out/r05/shared_gpu_stress.log
cp monte-carlo-ray-tracer_amd/csrc/libmcrt_hip.so /tmp/lib_keep.so
cp tools/_build/libpm768.so monte-carlo-ray-tracer_amd/csrc/libmcrt_hip.so
timeout 300 python tools/ab_probe.py c5 --sqrtspp 8 --emissions 1e7 --steps 2 "pm1024:" "pm768:MCRT_PM_BLOCK=768" "pm1024:" "pm768:MCRT_PM_BLOCK=768" 2>&1 | grep '^{' | cut -c1-200 | tee gpurun_out/r05/ab_pm768.log
timeout 300 python tools/ab_probe.py pm --steps 3 "pm1024:" "pm768:MCRT_PM_BLOCK=768" 2>&1 | grep '^{' | cut -c1-200 | tee -a gpurun_out/r05/ab_pm768.log
cp /tmp/lib_keep.so monte-carlo-ray-tracer_amd/csrc/libmcrt_hip.so
date
