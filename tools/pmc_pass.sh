#!/bin/bash
# usage: pmc_pass.sh <outdir-name> <counter> [<counter>...]   -- one rocprofv3 --pmc pass over a short bench run
name=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$name -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/$name.err
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/$name/p_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if "k_trace_primary<false" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("$name", k, "n=%d mean=%.5g" % (len(v), sum(v)/len(v)))
PY
