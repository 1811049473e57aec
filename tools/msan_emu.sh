#!/bin/bash
# The wavefront integrator's code (host build) under clang's MemorySanitizer: is any value used before it is written?
#   tools/msan_emu.sh [scene image] [slots] [width] [wf | wfpm | pm | sm | mega | top | flat]   (needs /opt/rocm/lib/llvm/bin/clang++; ~1 min)
# The emulation's slot pool starts as garbage except for the planes the device clears too, so this covers the pool words as well.
cd "$(dirname "$0")/.."
IMG=${1:-tests/golden/coffee_maker_qsah.mcrt}
B=/tmp/mcrt_msan; mkdir -p $B
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=memory -fsanitize-memory-track-origins -fsanitize-recover=memory -fno-omit-frame-pointer \
  -ffp-contract=off -Iinclude -o $B/driver tools/msan_driver.cpp tests/emu/mcrt_emu.cpp monte-carlo-ray-tracer_amd/csrc/mcrt_image.cpp || exit 1
MSAN_OPTIONS=halt_on_error=0:exitcode=0 $B/driver "$IMG" ${2:-333} ${3:-48} ${4:-wf} > $B/out.txt 2> $B/err.txt
cat $B/out.txt
echo "reports after the image was loaded (none expected):"
awk '/MSAN-DRIVER: image loaded/{f=1} f' $B/err.txt | grep SUMMARY | sort | uniq -c | sort -rn
