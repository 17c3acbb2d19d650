#!/bin/bash
# workgroups that one CU receives in turn (j, j + 32, ... of an XCD lane) made neighbouring tiles: patch sizes 4, 8, 16 (+256 = Z order)
for v in 0 4 8 16 260 264 272 0 272; do
  echo "== CGRT_CU_PATCH=$v"
  CGRT_CU_PATCH=$v python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'])"
done
