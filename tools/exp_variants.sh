#!/bin/bash
# usage: exp_variants.sh name1 name2 ...   ("" = default lib); CGRT_SUB_LEAF honoured
for v in "$@"; do
  lib="libcgrt.so"; [ "$v" != "default" ] && lib="libcgrt_$v.so"
  echo "== $v (sub_leaf=${CGRT_SUB_LEAF:-default})"
  CGRT_LIB_NAME=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
done
