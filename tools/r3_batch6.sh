#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_boundary_gpu.py -x -q -m gpu > gpurun_out/r3_b6_tests.log 2>&1; tail -3 gpurun_out/r3_b6_tests.log
grep -q failed gpurun_out/r3_b6_tests.log && exit 1
echo "== per-ray, leader polls the stream"; timeout -k 10 400 python tools/measure_per_ray.py > gpurun_out/r3_per_ray_b6.txt 2> gpurun_out/r3_per_ray_b6.err; grep -E "combining on|per-ray,|oracle" gpurun_out/r3_per_ray_b6.txt
echo "== per-ray, leader sleeps in hipStreamSynchronize"; CGRT_COMBINE_WAIT=1 timeout -k 10 400 python tools/measure_per_ray.py > gpurun_out/r3_per_ray_b6_blocking.txt 2>&1; grep -E "combining on|per-ray," gpurun_out/r3_per_ray_b6_blocking.txt
echo "== shaded frames: critical-path stream priority on / off"; timeout -k 10 200 python tools/measure_config3.py > gpurun_out/r3_config3_b6.txt 2>&1; cat gpurun_out/r3_config3_b6.txt
CGRT_AUX_PRIORITY=0 timeout -k 10 200 python tools/measure_config3.py > gpurun_out/r3_config3_b6_noprio.txt 2>&1; cat gpurun_out/r3_config3_b6_noprio.txt
timeout -k 10 200 bash tools/profile_render.sh cornell > gpurun_out/r3_config3_timeline_b6.txt 2>&1; tail -16 gpurun_out/r3_config3_timeline_b6.txt
bash tools/r3_batch4.sh
