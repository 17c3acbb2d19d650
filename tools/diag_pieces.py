"""Diagnostic for the CGRT_STAMP_SUB build (make variant NAME=stamp DEFS=-DCGRT_STAMP_SUB=1; CGRT_LIB_NAME=libcgrt_stamp.so):
wave cycles spent in each piece of the one-loop walk (T T E N N R P), slowest waves and all heavy waves.  GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
sd = pkg.scenes.make_dragon(int(os.environ.get("TRIS", "800000")))
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
t = pkg.Scene(sd).debug_wave_times(cam, W, H).astype(np.int64)
dur = t[:, 1] - t[:, 0]
pieces = np.stack([t[:, 4], t[:, 5], t[:, 6], t[:, 7], t[:, 11], t[:, 2], t[:, 3]], 1)  # T1 T2 E N1 N2 R P
trips = t[:, 15] >> 8
w_inner, w_sub, w_tri = t[:, 8], t[:, 9], t[:, 10]
names = ["T1", "T2", "E", "N1", "N2", "R", "P"]
order = np.argsort(-dur)[:8]
for i in order:
    print(int(i), "cycles", int(dur[i]), "trips", int(trips[i]), "bodies T/N/R", int(w_inner[i]), int(w_sub[i]), int(w_tri[i]),
          " ".join("%s %d" % (n, v) for n, v in zip(names, pieces[i])), "| unaccounted", int(dur[i] - pieces[i].sum()))
heavy = dur > 12000
tot = pieces[heavy].sum(0)
print("heavy waves:", int(heavy.sum()), "cycles", int(dur[heavy].sum()), "trips", int(trips[heavy].sum()))
print("  share per piece:", " ".join("%s %.3f" % (n, v / dur[heavy].sum()) for n, v in zip(names, tot)))
print("  cycles per trip:", " ".join("%s %.0f" % (n, v / trips[heavy].sum()) for n, v in zip(names, tot)))
print("  per executed body: topology %.0f  node %.0f  run %.0f" % ((tot[0] + tot[1]) / w_inner[heavy].sum(), (tot[3] + tot[4]) / w_sub[heavy].sum(), tot[5] / max(1, w_tri[heavy].sum())))
