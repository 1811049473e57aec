#!/bin/bash
# round 5, call 27: the GPU tier on the library with glibc's pow restated in the output stage (developKernel; every render kernel is
# instruction-identical to call 26's library: tools/device_code_hashes.py) - Image::save's bytes must now be the reference's
mkdir -p gpurun_out/r05_evidence6
( time timeout 1200 python -m pytest tests -m gpu -q -rA ) > gpurun_out/r05_evidence6/pytest_gpu.log 2>&1
tail -3 gpurun_out/r05_evidence6/pytest_gpu.log
grep -n "FAILED\|ERROR" gpurun_out/r05_evidence6/pytest_gpu.log | head
