#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_scene.py -m gpu -q -x -k "wavefront or pipeline or optional or spaceship" 2>&1 | tail -2
timeout 200 python tools/ab_probe.py c3 --sqrtspp 4 --steps 2 "wf:" 2>&1 | grep '^{' | cut -c1-110
timeout 200 python tools/ab_probe.py spaceship --steps 2 "auto:" 2>&1 | grep '^{' | cut -c1-110
