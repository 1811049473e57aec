#!/bin/bash
R=$PWD
O=$R/gpurun_out/r02_ab9
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python -m pytest $R/tests/test_bvh_build.py -m gpu -q -x -k "large_scene_sah and c3" > $O/kt.log 2>&1
cd $R
python tools/summarize_rocprof.py $O > $O/summary.md 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
head -30 $O/summary.md
