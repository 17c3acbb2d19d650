#!/bin/bash
# order of the 64 tiles inside a super-tile: row by row (0) or Z order (1)
for v in 0 1 0 1 0 1; do
  echo "== CGRT_TILE_MORTON=$v"
  CGRT_TILE_MORTON=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'])"
done
CGRT_TILE_MORTON=0 python bench.py --workload shaded --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shaded rows', j['value'], j['ms_per_step'])"
CGRT_TILE_MORTON=1 python bench.py --workload shaded --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shaded Z   ', j['value'], j['ms_per_step'])"
CGRT_TILE_MORTON=0 python bench.py --workload shaded --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shaded rows', j['value'], j['ms_per_step'])"
CGRT_TILE_MORTON=1 python bench.py --workload shaded --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shaded Z   ', j['value'], j['ms_per_step'])"
