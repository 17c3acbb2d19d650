#!/bin/bash
# bench the default library and experiment builds (cg-raytracer_amd/lib/libcgrt_<name>.so)
for lib in "$@"; do
  echo "== $lib"
  CGRT_LIB_NAME=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['per_ray'])"
done
