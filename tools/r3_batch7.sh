#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_host_mirror.py tests/test_quad_shape_gpu.py tests/test_soft_shadows.py tests/test_boundary_gpu.py -x -q -m gpu > gpurun_out/r3_b7_tests.log 2>&1; tail -3 gpurun_out/r3_b7_tests.log
grep -q failed gpurun_out/r3_b7_tests.log && exit 1
echo "== shaded frames, level-0 shadow list issued before the read-back"; timeout -k 10 200 python tools/measure_config3.py > gpurun_out/r3_config3_b7.txt 2>&1; cat gpurun_out/r3_config3_b7.txt
echo "== per-ray, 2 rings"; PER_RAY_QUICK=1 timeout -k 10 300 python tools/measure_per_ray.py 2>&1 | grep "combining on"
cp cg-raytracer_amd/lib/libcgrt.so /tmp/libcgrt_keep.so && cp cg-raytracer_amd/lib/libcgrt_rings4.so cg-raytracer_amd/lib/libcgrt.so
echo "== per-ray, 4 rings"; PER_RAY_QUICK=1 timeout -k 10 300 python tools/measure_per_ray.py 2>&1 | grep "combining on"
cp /tmp/libcgrt_keep.so cg-raytracer_amd/lib/libcgrt.so
timeout -k 10 200 bash tools/profile_render.sh cornell > gpurun_out/r3_config3_timeline_b7.txt 2>&1; tail -14 gpurun_out/r3_config3_timeline_b7.txt
