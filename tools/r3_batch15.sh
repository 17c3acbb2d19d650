#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_boundary_gpu.py -x -q -m gpu > gpurun_out/r3_b15_tests.log 2>&1; tail -3 gpurun_out/r3_b15_tests.log
grep -q failed gpurun_out/r3_b15_tests.log && exit 1
echo "== leader watches a word in pinned memory"; PER_RAY_QUICK=1 timeout -k 10 300 python tools/measure_per_ray.py 2>&1 | grep "combining on"
echo "== leader in hipStreamSynchronize"; CGRT_COMBINE_WAIT=1 PER_RAY_QUICK=1 timeout -k 10 300 python tools/measure_per_ray.py 2>&1 | grep "combining on"
