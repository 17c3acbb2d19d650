#!/bin/bash
# round 3, GPU batch 4: trip shapes of the lane-per-ray search loop (walk_fast.h CGRT_LOOP_NU): 1 = [node][U], 2 = [U][U], 3 = [node][node][U]
mkdir -p gpurun_out
for v in nu1 nu2 nu3; do
  SUITE_LONE=0 CGRT_LIB_NAME=libcgrt_$v.so timeout -k 10 200 python tools/latency_suite.py $v > gpurun_out/r3_suite_$v.json 2> gpurun_out/r3_suite_$v.err && cat gpurun_out/r3_suite_$v.json || exit 1
done
# rehearsal of the N > 1 bench path on one card (gloo, two ranks sharing the GPU): the line must carry cpu_baseline now
CGRT_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r3_bench_gloo_n2.json 2> gpurun_out/r3_bench_gloo_n2.err; tail -c 1500 gpurun_out/r3_bench_gloo_n2.json
