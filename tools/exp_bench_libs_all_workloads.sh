#!/bin/bash
# bench + config3/dragon-shaded for experiment builds
for lib in "$@"; do
  echo "== $lib"
  CGRT_LIB_NAME=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('certified', j['value'], j['roofline']['kernel_ms'])"
  CGRT_LIB_NAME=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --walk exact 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('exact', j['value'], j['roofline']['kernel_ms'])"
  CGRT_LIB_NAME=$lib python tools/measure_config3.py
done
