#!/bin/bash
# rocprofv3 kernel trace of the device render driver on config 3 and on the dragon stand-in (tools/measure_config3.py).
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_render
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $R/tools/measure_config3.py $1 > $out/run.log 2> $out/trace.err || exit 1
cat $out/run.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/trace/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame of the last scene: the last 9 kernels (generate, items, trace, [spawn, trace, shade] x levels..., combine)
tail = rows[-14:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    print("%-40s start %7.1f  end %7.1f  (%6.1f us)  grid %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                           (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
PY
