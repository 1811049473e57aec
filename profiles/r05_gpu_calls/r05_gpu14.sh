#!/bin/bash
# round 5, call 14: the flat megakernel with its cull records in the kernel's ARGUMENT BLOCK (renderKernelFlatK: scalar loads into SGPRs
# instead of seven wave-wide LDS reads per triangle pair), MCRT_FLAT_KARG=1 (default) against 0, on C2 and C2-GGX; then the parity tests
mkdir -p gpurun_out/r05
date
for wl in c2 c2_ggx; do
  timeout 300 python tools/ab_probe.py $wl --steps 3 "lds:MCRT_FLAT_KARG=0" "karg:" "lds:MCRT_FLAT_KARG=0" "karg:" 2>&1 | grep '^{' | cut -c1-200 | sed "s/^/$wl /" | tee -a gpurun_out/r05/ab_c2_flat_karg.log
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiprocess.py tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r05/pytest_call14.log
date
