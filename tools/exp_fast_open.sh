#!/bin/bash
# fast tree: levels of the leaf accelerators opened for the top tree's SAH -> bench + counters
for o in 0 1 2 3 4; do
  echo "== CGRT_FAST_OPEN=$o"
  CGRT_FAST_OPEN=$o python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['per_ray'], j['config']['bvh_build_and_upload_s'])"
done
