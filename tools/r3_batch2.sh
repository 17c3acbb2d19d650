#!/bin/bash
# round 3, GPU batch 2: new parity tests (shapes, call combining, per-ray driver, mapped frame), the latency suite under the
# default shape policy, the per-ray boundary, the host mirror's frame path, config 3's kernel timeline
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_quad_shape_gpu.py tests/test_boundary_gpu.py tests/test_host_mirror.py -x -q -m gpu > gpurun_out/r3_b2_tests.log 2>&1
tail -6 gpurun_out/r3_b2_tests.log
grep -q passed gpurun_out/r3_b2_tests.log || exit 1
timeout -k 10 300 python tools/latency_suite.py auto > gpurun_out/r3_suite_auto.json 2> gpurun_out/r3_suite_auto.err && cat gpurun_out/r3_suite_auto.json &&
timeout -k 10 400 python tools/measure_per_ray.py > gpurun_out/r3_per_ray.txt 2> gpurun_out/r3_per_ray.err; cat gpurun_out/r3_per_ray.txt; tail -3 gpurun_out/r3_per_ray.err
timeout -k 10 200 python tools/measure_host_mirror.py > gpurun_out/r3_host_mirror.txt 2>&1; cat gpurun_out/r3_host_mirror.txt
timeout -k 10 200 bash tools/profile_render.sh cornell > gpurun_out/r3_config3_timeline.txt 2>&1; cat gpurun_out/r3_config3_timeline.txt
