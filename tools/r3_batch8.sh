#!/bin/bash
# round 3, GPU batch 8: tail splitting -- parity first (bounded: every wait in the kernel has a hard count), then the frames with / without
mkdir -p gpurun_out
timeout -k 10 180 python -m pytest tests/test_quad_shape_gpu.py -x -q -m gpu -k tail_splitting > gpurun_out/r3_b8_tests.log 2>&1; tail -15 gpurun_out/r3_b8_tests.log
grep -q " passed" gpurun_out/r3_b8_tests.log || exit 1
grep -q failed gpurun_out/r3_b8_tests.log && exit 1
SUITE_PARTS=frames CGRT_TAIL_SPLIT=0 timeout -k 10 200 python tools/latency_suite.py split_off > gpurun_out/r3_suite_split_off.json 2> gpurun_out/r3_suite_split_off.err && cat gpurun_out/r3_suite_split_off.json &&
SUITE_PARTS=frames CGRT_TAIL_SPLIT=1 timeout -k 10 200 python tools/latency_suite.py split_on > gpurun_out/r3_suite_split_on.json 2> gpurun_out/r3_suite_split_on.err && cat gpurun_out/r3_suite_split_on.json
