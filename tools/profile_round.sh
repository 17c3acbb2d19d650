#!/bin/bash
# Profile of the bench command for profiles/: kernel trace + stats, then PMC passes (each in its own run,
# --pmc never combined with tracing domains).  Usage on the GPU box:  bash tools/profile_round.sh <tag>
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_under_trace.json 2> $out/trace.err || exit 1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$n -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $out/$n.err || exit 1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
# what bounds the kernel (bench.py roofline.measured): the launch's cycles, the vector L1's busy cycles and its latency per
# wave-level load -- at most two raw counters of a block per pass (profiles/r2_pmc_memory_path.txt)
pass grbm GRBM_GUI_ACTIVE
pass tcp TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcp5 TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
python3 - <<PY
import csv, collections, json, re
out="$out"
def mean(path, kernel="k_trace_primary<false"):  # the timed instantiation (COUNT = false)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v)/len(v) for k,v in agg.items()}
res={}
for n in ("fetch","write","cache","sq","grbm","tcp","tcp5"): res.update(mean(f"{out}/{n}/p_counter_collection.csv"))
import subprocess, sys
src_hash = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as e; print(e.load_package().source_hash())"],
                          capture_output=True, text=True).stdout.strip()
# GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launch = /8; 256 CUs, 4 SIMDs each
cyc = res.get("GRBM_GUI_ACTIVE", 0) / 8.0
measured = None
if cyc > 0 and res.get("SQ_WAVE_CYCLES") and res.get("TCP_TA_TCP_STATE_READ_sum"):
    measured = {"valu_issue_frac": round(res["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 256 * 4), 3),
                "l1_busy_frac": round(res["TCP_GATE_EN1_sum"] / (cyc * 256), 3),
                "waitcnt_frac": round(res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"], 3),
                "l1_cycles_per_load": round(res["TCP_TCP_LATENCY_sum"] / res["TCP_TA_TCP_STATE_READ_sum"], 1),
                "gpu_cycles_per_launch": round(cyc),
                "definitions": "valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles; l1_busy_frac = TCP_GATE_EN1 / CU-cycles; waitcnt_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
                               "l1_cycles_per_load = TCP_TCP_LATENCY / TCP_TA_TCP_STATE_READ; launch cycles = GRBM_GUI_ACTIVE / 8 XCDs"}
bench=json.loads(open(f"{out}/bench_under_trace.json").read())
# FETCH_SIZE/WRITE_SIZE are in units of 1 KB (rocprofv3 derives them from TCC_EA0_RDREQ*64B / WRREQ).  On gfx950 FETCH_SIZE
# reads 1/2 of a WIDE COALESCED stream (MI355X_MICROARCH.md HBM section) and is uncalibrated for other shapes; for this
# kernel's shape -- scattered 64-byte records, 16 B per lane per instruction -- tools/calibrate_fetch_size.sh measured
# counter/known-bytes = 1.0000 on a 1 GiB table (profiles/r1_fetch_size_calibration.txt), so the raw value is used.
fetch_b = res.get("FETCH_SIZE",0)*1024; write_b = res.get("WRITE_SIZE",0)*1024
summary = {"counters": res, "fetch_bytes_raw": fetch_b, "fetch_bytes_x2_gfx950": 2*fetch_b, "write_bytes": write_b,
           "hbm_bytes_per_launch": fetch_b + write_b,
           "workload": "dragon%s_%s_%s" % (re.search(r"stand-in (\d+) tris", bench["config"]["workload"]).group(1), re.search(r"(\d+x\d+) primary", bench["config"]["workload"]).group(1),
                                            "certified" if "certified" in bench["config"]["workload"] else "exact"),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_round.sh %s" % "$tag",
           "source_hash": src_hash, "measured": measured,
           "bench_under_trace": {k: bench[k] for k in ("value","ms_per_step")}, "roofline": bench["roofline"]}
json.dump(summary, open(f"{out}/summary.json","w"), indent=1)
json.dump({k: summary[k] for k in ("workload", "hbm_bytes_per_launch", "fetch_bytes_raw", "write_bytes", "source", "source_hash", "measured")} | {"note": "FETCH_SIZE calibrated for this access shape (scattered 64-B records, 16 B per lane per instruction): counter/known = 1.0000 (profiles/r1_fetch_size_calibration.txt), so no gfx950 x2 correction applies"},
          open(f"{out}/hbm_traffic_latest.json","w"), indent=1)
print(json.dumps(summary)[:1500])
PY
cat $out/trace/t_kernel_stats.csv | cut -c1-220
