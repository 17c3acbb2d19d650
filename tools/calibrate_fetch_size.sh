#!/bin/bash
# FETCH_SIZE calibration on a known byte count in the traversal kernels' access shape (64-B record gathers, 16 B per lane
# per instruction), as MI355X_MICROARCH.md asks before trusting an absolute.  Table 1 GiB (4x the Infinity Cache).
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/calib
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $R/tools/calibrate_fetch_size.py 16777216 > $out/run.log 2>&1 || exit 1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$out/fetch/p_counter_collection.csv")) if "k_gather_calib" in r["Kernel_Name"]]
known=16777216*64
for r in rows: print("FETCH_SIZE", r["Counter_Value"], "KB-units ->", float(r["Counter_Value"])*1024, "bytes; known", known, "ratio counter/known = %.4f" % (float(r["Counter_Value"])*1024/known))
PY
