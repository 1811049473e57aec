#!/bin/bash
# The wave-cooperative device code (csrc/mcrt_waveknn.hpp, csrc/mcrt_sharedleaf.hpp) under AddressSanitizer + UBSan, on the host
# emulation of a wavefront (tests/emu/wave_emu.hpp): ~1 min, no GPU.   tools/asan_wave_emu.sh
cd "$(dirname "$0")/.."
B=/tmp/mcrt_wave_asan; mkdir -p $B
for t in wave_knn_emu wave_walk_emu wave_kernel_emu; do
  g++ -std=c++17 -O1 -fno-inline -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -fPIC -shared -o $B/lib$t.so tests/emu/$t.cpp || exit 1
done
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python tools/asan_wave_emu.py
