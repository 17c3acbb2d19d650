#!/bin/bash
# age-based issue priority: full 1080p frame and the slowest 1/8 share of the 4K frame, per library build
for lib in "$@"; do
  echo "== $lib"
  CGRT_LIB_NAME=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('1080p', j['value'], j['roofline']['kernel_ms'])"
  CGRT_LIB_NAME=$lib python tools/predict_strong_scaling.py 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print([(r['n_gpus'], r['predicted_ms_per_step']) for r in j['curve']])"
done
