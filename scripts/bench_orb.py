"""Per-kernel timing of one batched ORB launch chain (32 frame slots) with HIP events."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_visualodometry_amd import capi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
syn = capi.Synth(); bgr, depth, _, _ = syn.render(syn.params(seed=0), 0, F, threads=16)
L = capi.load(sys.argv[2] if len(sys.argv) > 2 else capi.HIP_LIB)
ctx = L.context(L.default_params(n_features=2000, max_frames=F))
for s in range(F): ctx.upload(s, bgr[s], depth[s])
ctx.orb(0, F); ctx.sync()
ctx.profile_enable(True)
for rep in range(5): ctx.orb(0, F)
ctx.sync()
for k, (ms, n) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][0]):
    print("%-12s %3d launches  avg %8.1f us" % (k, n, ms / n * 1e3))
