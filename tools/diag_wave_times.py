"""Diagnostic: distribution of per-wave durations and the launch timeline (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
if os.environ.get("CGRT_SUB_LEAF"):
    pkg.set_leaf_accel(True, int(os.environ["CGRT_SUB_LEAF"]))
sd = pkg.scenes.make_dragon(int(os.environ.get("TRIS", "800000")))
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
t = sc.debug_wave_times(cam, W, H).astype(np.int64)
dur = t[:, 1] - t[:, 0]           # shader cycles
rt = (t[:, 2:4] - t[:, 2].min()) / 100.0  # us (s_memrealtime ticks at 100 MHz)
print("waves", len(t), "kernel span us", rt[:, 1].max())
print("dur cycles: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
wall = rt[:, 1] - rt[:, 0]
print("wall us   : mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (wall.mean(), np.median(wall), np.percentile(wall, 90), np.percentile(wall, 99), wall.max()))
heavy = wall > 5
print("heavy waves (>5us):", heavy.sum(), "sum wall us", wall[heavy].sum(), "=> /4096 slots:", wall[heavy].sum() / 4096)
# timeline: number of waves running at each 10us
edges = np.arange(0, rt[:, 1].max() + 10, 10)
running = [(int(((rt[:, 0] <= x) & (rt[:, 1] > x)).sum())) for x in edges]
print("running waves every 10us:", running)
print("start time of waves (us) percentiles:", [round(float(np.percentile(rt[:, 0], p)), 1) for p in (1, 25, 50, 75, 99, 100)])
order = np.argsort(-wall)[:10]
print("slowest waves: idx, start, wall:", [(int(i), round(float(rt[i, 0]), 1), round(float(wall[i]), 1)) for i in order])

# work anatomy of the slow waves: columns 4..14 = sum inner, leaf, tri, sub | wave iters inner, sub, tri | lane max inner, leaf, tri, sub; 15 = active lanes
names = ["sum_inner", "sum_leaf", "sum_tri", "sum_sub", "w_inner", "w_sub", "w_tri", "max_inner", "max_leaf", "max_tri", "max_sub", "lanes"]
for i in order[:6]:
    print(int(i), "cycles", int(dur[i]), dict(zip(names, [int(x) for x in t[i, 4:16]])))
hv = np.nonzero(heavy)[0]
tot = t[hv, 4:16].sum(0)
print("all heavy waves: ", dict(zip(names, [int(x) for x in tot])))
print("lane efficiency inner %.3f sub %.3f tri %.3f" % (tot[0] / (64.0 * tot[4]), tot[3] / (64.0 * tot[5]), tot[2] / (64.0 * max(1, tot[6]))))
it = t[:, 8].astype(np.float64) + t[:, 9] + t[:, 10]
m = heavy & (it > 0)
A = np.stack([t[m, 8], t[m, 9], t[m, 10], np.ones(m.sum())], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur[m].astype(np.float64), rcond=None)
print("least-squares cycles per wave-iteration: inner %.0f sub %.0f tri %.0f const %.0f" % tuple(coef))

if os.environ.get("STAMP_SUB"):
    # CGRT_STAMP_SUB build with the one-loop walk: columns 12..14 = wave cycles inside the topology pieces (T T E), the
    # node steps (N N) and the run test + pop (R P); wave-iteration counts of the bodies in columns 8..10
    for i in order[:8]:
        print(int(i), "cycles", int(dur[i]), "topology pieces", int(t[i, 12]), "node steps", int(t[i, 13]), "run+pop", int(t[i, 14]),
              "| per executed body: topology %.0f node %.0f run %.0f" % (t[i, 12] / max(1, t[i, 8]), t[i, 13] / max(1, t[i, 9]), t[i, 14] / max(1, t[i, 10])))
    tot3 = t[hv, 12:15].sum(0)
    print("all heavy waves: topology %.3g node %.3g run+pop %.3g cycles; per executed body %.0f %.0f %.0f" %
          (tot3[0], tot3[1], tot3[2], tot3[0] / tot[4], tot3[1] / tot[5], tot3[2] / max(1, tot[6])))
