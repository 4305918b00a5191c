"""How often do two HIP runs of one short stream differ in an integer statistic (the local BA sums in no fixed order)?  python scripts/flake_probe.py [n=10]"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("tdk", os.path.join(ROOT, "tests", "test_device_keyframes.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from rgbd_visualodometry_amd import capi, system
syn = capi.Synth(); stream = syn.render(syn.params(seed=11, speed=3.0), 0, 64, threads=8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
keys = ("keyframes", "map_points", "ba_runs", "ba_failed", "triangulated", "lost", "ba_points", "ba_edges", "ba_poses", "ba_fixed", "ba_outliers")
ref = m.run_system(system.HOST_LIB, stream, 48, 6, 1, lookahead=8, batch=4)
bad = 0
for i in range(n):
    cap = 1024 if i % 2 else 1 << 17
    r = m.run_system(system.HOST_LIB, stream, 48, 6, 1, lookahead=8, batch=4, map_capacity=cap)
    diff = [k for k in keys if r["stats"][k] != ref["stats"][k]]
    import numpy as np
    dt = float(np.abs(r["traj"] - ref["traj"]).max())
    if diff: bad += 1
    print("run %d (capacity %d): differing statistics %s, max trajectory difference %.3e" % (i, cap, diff, dt))
print("%d of %d runs differ from the first in an integer statistic" % (bad, n))
