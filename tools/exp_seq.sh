#!/bin/bash
# order in time of an XCD lane's workgroups: screen order (default) vs strided permutations
for v in 0 1 0 3 16 64 0; do
  echo "== CGRT_SEQ_SCRAMBLE=$v"
  CGRT_SEQ_SCRAMBLE=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'])"
done
