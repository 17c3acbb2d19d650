#!/bin/bash
mkdir -p gpurun_out
PER_RAY_QUICK=1 timeout -k 10 300 python tools/measure_per_ray.py 2>&1 | grep "combining on"
