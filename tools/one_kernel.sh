#!/bin/bash
# Compile ONE kernel instance of csrc/mcrt_kernels.hpp for gfx950 and print its register / spill / instruction table
# (seconds instead of the library's minutes):  tools/one_kernel.sh 'renderKernelFlatK<768>' [extra hipcc flags]
cd "$(dirname "$0")" || exit 1
K=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "-DONE_KERNEL=$K" "$@" -save-temps=obj -c one_kernel.hip -o /tmp/one_kernel.o 2>&1 | grep -E "error|Error"
python kernel_resources.py /tmp/one_kernel-hip-amdgcn-amd-amdhsa-gfx950.s "${K%%<*}"
