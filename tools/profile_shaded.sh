#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE: each in its own run, never combined with tracing) over the shaded dragon frame
# (tools/measure_config3.py dragon800k): HBM bytes per launch of the secondary-ray kernels.  Usage on the GPU box: bash tools/profile_shaded.sh
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_shaded
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python3 $R/tools/measure_config3.py dragon800k > /dev/null 2> $out/$c.err || exit 1
done
python3 - <<PY
import csv, collections
out="$out"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/{c}/p_counter_collection.csv")):
        agg[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        if "trace" in k: print(f"{c:11s} {k:60s} launches {len(v):3d}  mean {sum(v)/len(v)*1024/1e6:9.2f} MB per launch (counter unit 1 KB; FETCH_SIZE uncorrected: scattered 64-B records, profiles/r1_fetch_size_calibration.txt)")
PY
