#!/bin/bash
# round 5, call 15: renderKernelFlatK with the next pair's record asked for ahead of the arithmetic (libkarg_ahead) against the plain
# form (libkarg_plain), library builds swapped on one box; each also against its own LDS form (MCRT_FLAT_KARG=0)
mkdir -p gpurun_out/r05
date
LIB=monte-carlo-ray-tracer_amd/csrc/libmcrt_hip.so
cp $LIB /tmp/lib_orig.so
for x in karg_plain karg_ahead karg_plain karg_ahead; do
  cp tools/_build/lib$x.so $LIB
  for wl in c2 c2_ggx; do
    echo "build $x $wl: $(timeout 300 python tools/ab_probe.py $wl --steps 3 "karg:" "lds:MCRT_FLAT_KARG=0" 2>&1 | grep '^{' | cut -c1-100 | tr '\n' ' ')" | tee -a gpurun_out/r05/ab_c2_flat_karg_ahead.log
  done
done
cp /tmp/lib_orig.so $LIB
date
