#!/bin/bash
# round 3: the round's measurement batch -- full GPU suite, bench line, strong-scaling prediction, latency suite, wave anatomy,
# then tools/profile_round.sh (kernel trace + PMC passes)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests_full.log 2>&1; tail -4 gpurun_out/r3_gpu_tests_full.log
grep -q failed gpurun_out/r3_gpu_tests_full.log && exit 1
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err && cut -c1-400 gpurun_out/r3_bench_n1.json &&
timeout -k 10 300 python bench.py --workload shaded --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_shaded.json 2> gpurun_out/r3_bench_shaded.err && cut -c1-300 gpurun_out/r3_bench_shaded.json &&
timeout -k 10 300 python tools/predict_strong_scaling.py 20 > gpurun_out/r3_predicted_strong_scaling.json 2> gpurun_out/r3_pred.err && python -c "
import json; j=json.load(open('gpurun_out/r3_predicted_strong_scaling.json'))
for r in j['curve']: print(r['n_gpus'], r['predicted_ms_per_step'], r['predicted_speedup'])" &&
timeout -k 10 300 python tools/latency_suite.py final > gpurun_out/r3_suite_final.json 2> gpurun_out/r3_suite_final.err && cat gpurun_out/r3_suite_final.json &&
timeout -k 10 200 python tools/measure_config3.py > gpurun_out/r3_config3_final.txt 2>&1; cat gpurun_out/r3_config3_final.txt
timeout -k 10 200 python tools/diag_wave_times.py > gpurun_out/r3_wave_anatomy.txt 2>&1; tail -12 gpurun_out/r3_wave_anatomy.txt
timeout -k 10 800 bash tools/profile_round.sh r3 > gpurun_out/r3_profile_round.log 2>&1; tail -25 gpurun_out/r3_profile_round.log | cut -c1-300
