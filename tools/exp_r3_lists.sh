#!/bin/bash
# round 3, experiment 1: what a short ray list costs by how its rays are laid out on the lanes (results identical, tested)
#   lane64   lane per ray, 64 rays per wave (round 2's shape)
#   lane16   lane per ray, 16 rays per single-wave workgroup: the quad tail (4 lanes per ray, 4 stack entries in flight) from the first step
#   lane4    4 rays per wave
#   quad16   quad per ray (a node's four boxes in parallel), 16 rays per wave, a row of 16 lanes per ray once <= 4 are live
#   quad4    quad per ray, 4 rays per wave: 16 lanes per ray (4 entries x 4 boxes) from the first step
export SUITE_PARTS=lists
run() { tag=$1; shift; env "$@" timeout -k 10 200 python tools/latency_suite.py $tag 2> gpurun_out/r3_lists_$tag.err | tee -a gpurun_out/r3_exp_lists.txt; }
: > gpurun_out/r3_exp_lists.txt
run lane64 CGRT_QUAD_MODE=0 &&
run lane16 CGRT_QUAD_MODE=0 CGRT_LANE_RAYS_PER_WAVE=16 &&
run lane4 CGRT_QUAD_MODE=0 CGRT_LANE_RAYS_PER_WAVE=4 &&
run quad16 CGRT_QUAD_MODE=1 &&
run quad4 CGRT_QUAD_MODE=1 CGRT_QUAD_RAYS_PER_WAVE=4
