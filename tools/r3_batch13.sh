#!/bin/bash
# round 3, GPU batch 13: tail splitting v4 (helpers stay while tree waves run, hand-over only while helpers run, CAS claims, stats)
mkdir -p gpurun_out
timeout -k 10 180 python -m pytest tests/test_quad_shape_gpu.py -x -q -m gpu -k tail_splitting > gpurun_out/r3_b13_tests.log 2>&1; tail -15 gpurun_out/r3_b13_tests.log
grep -q " passed" gpurun_out/r3_b13_tests.log || exit 1
grep -q failed gpurun_out/r3_b13_tests.log && exit 1
run() { tag=$1; shift; env SUITE_PARTS=frames "$@" timeout -k 10 200 python tools/latency_suite.py $tag 2> gpurun_out/r3_split_$tag.err | tee -a gpurun_out/r3_exp_tail_split.txt; }
: > gpurun_out/r3_exp_tail_split.txt
run off CGRT_TAIL_SPLIT=0 &&
run never CGRT_TAIL_SPLIT=1 CGRT_SPILL_FIRST=100000 &&
run first16 CGRT_TAIL_SPLIT=1 CGRT_SPILL_FIRST=16 &&
run first24 CGRT_TAIL_SPLIT=1 CGRT_SPILL_FIRST=24 &&
run first24_keep16 CGRT_TAIL_SPLIT=1 CGRT_SPILL_FIRST=24 CGRT_SPILL_KEEP=16
