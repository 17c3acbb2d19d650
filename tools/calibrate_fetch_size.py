import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 24)   # 16 Mi records = 1 GiB
rc = pkg.lib().cgrt_debug_gather_calibration(0, n, 3)
assert rc == 0, pkg.lib().cgrt_last_error()
print("gathered", n, "records x 64 B =", n * 64, "bytes per launch, 3 launches")
