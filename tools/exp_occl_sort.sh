#!/bin/bash
# child ordering of occlusion / any-hit queries: shaded dragon frame and soft shadows per library build
for lib in "$@"; do
  echo "== $lib"
  for i in 1 2; do CGRT_LIB_NAME=$lib python bench.py --workload shaded --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shaded', j['value'], j['ms_per_step'])"; done
  CGRT_NO_CPU=1 CGRT_LIB_NAME=$lib python tests/diag/measure_soft_shadows.py 2>/dev/null | tail -4
done
