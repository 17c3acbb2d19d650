"""Scene setup for the 800 K-triangle stand-in: cgrt_scene_create (build on the host's threads + upload), best of 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
sd = pkg.scenes.make_dragon(800_000)
pkg.Scene(sd).close()  # (first HIP use of the process)
for threads in (0, 1):
    pkg.set_build_threads(threads)
    best = None
    for _ in range(4 if threads == 0 else 1):
        t0 = time.perf_counter()
        sc = pkg.Scene(sd)
        t = time.perf_counter() - t0
        b = sc.build_seconds() if hasattr(sc, "build_seconds") else float("nan")
        mb = sc.device_bytes() / 1e6
        sc.close()
        if best is None or t < best[0]:
            best = (t, b)
    print(f"threads {threads or 'all'}: create + upload {best[0]:.3f} s, of which the host build {best[1]:.3f} s; {mb:.0f} MB on the device", flush=True)
pkg.set_build_threads(0)
