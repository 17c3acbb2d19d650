#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/measure_per_ray.py > gpurun_out/r3_per_ray_b5.txt 2> gpurun_out/r3_per_ray_b5.err; cat gpurun_out/r3_per_ray_b5.txt; tail -3 gpurun_out/r3_per_ray_b5.err
bash tools/r3_batch4.sh
