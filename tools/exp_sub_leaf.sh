#!/bin/bash
for sl in 1 2 3 4 8; do
  echo "sub_leaf=$sl"
  CGRT_SUB_LEAF=$sl python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['per_ray'], j['roofline']['frac'])"
done
