#!/bin/bash
# experiment: does capping blocks/CU (via extra dynamic LDS) change the frame time?  (block->CU packing hypothesis)
for pad in 0 20000 50000 120000; do
  echo "pad=$pad"
  CGRT_EXP_LDS_PAD=$pad python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'])"
done
