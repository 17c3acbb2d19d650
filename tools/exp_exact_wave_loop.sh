#!/bin/bash
for lib in libcgrt.so libcgrt_oldloop.so; do
  echo "== $lib exact walk, bench frame"
  CGRT_LIB_NAME=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --walk exact 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'])"
  echo "== $lib config 3 + dragon shaded"
  CGRT_LIB_NAME=$lib python tools/measure_config3.py
done
