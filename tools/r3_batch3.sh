#!/bin/bash
# round 3, GPU batch 3: unified quad-tail loads, lock-free call combining, cached render streams + overlapped read-back
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_quad_shape_gpu.py tests/test_boundary_gpu.py tests/test_host_mirror.py tests/test_certified.py tests/test_soft_shadows.py -x -q -m gpu > gpurun_out/r3_b3_tests.log 2>&1
tail -6 gpurun_out/r3_b3_tests.log
grep -q " passed" gpurun_out/r3_b3_tests.log || exit 1
grep -q failed gpurun_out/r3_b3_tests.log && exit 1
timeout -k 10 300 python tools/latency_suite.py auto_b3 > gpurun_out/r3_suite_auto_b3.json 2> gpurun_out/r3_suite_auto_b3.err && cat gpurun_out/r3_suite_auto_b3.json &&
SUITE_PARTS=lists CGRT_SHAPE=2 timeout -k 10 200 python tools/latency_suite.py lane16_b3 > gpurun_out/r3_suite_lane16_b3.json 2> gpurun_out/r3_suite_lane16_b3.err && cat gpurun_out/r3_suite_lane16_b3.json &&
timeout -k 10 400 python tools/measure_per_ray.py > gpurun_out/r3_per_ray_b3.txt 2> gpurun_out/r3_per_ray_b3.err; cat gpurun_out/r3_per_ray_b3.txt; tail -3 gpurun_out/r3_per_ray_b3.err
timeout -k 10 200 python tools/measure_host_mirror.py > gpurun_out/r3_host_mirror_b3.txt 2>&1; tail -2 gpurun_out/r3_host_mirror_b3.txt
timeout -k 10 200 python tools/measure_config3.py > gpurun_out/r3_config3_b3.txt 2>&1; cat gpurun_out/r3_config3_b3.txt
