"""Row a-1 alone: batched ORB detect + describe over a 32-frame look-ahead batch, frames resident in HBM (bench.py's orb_only leg) + per-kernel HIP-event timing."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_visualodometry_amd import capi
W, H, N, F = 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 32
syn = capi.Synth(); bgr, depth, Twc, ts = syn.render(syn.params(seed=0, speed=3.0), 0, F, threads=16)
d_b = torch.from_numpy(bgr).cuda(); d_d = torch.from_numpy(depth.view(np.int16)).cuda()
L = capi.load(capi.HIP_LIB)
oc = L.context(L.default_params(width=W, height=H, n_features=N, max_frames=F))
for j in range(F): oc.bind_device(j, d_b.data_ptr() + j * W * H * 3, 3 * W, d_d.data_ptr() + j * W * H * 2, 2 * W)
for _ in range(3): oc.orb(0, F)
torch.cuda.synchronize()
reps = 20; t0 = time.perf_counter()
for _ in range(reps): oc.orb(0, F)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("orb_only frames/s %.0f  (%.1f us per %d-frame batch, %d features)" % (reps * F / dt, dt / reps * 1e6, F, N))
oc.profile_enable(True)
for _ in range(10): oc.orb(0, F)
torch.cuda.synchronize()
for name, (ms, calls) in sorted(oc.profile_read(64).items(), key=lambda kv: -kv[1][0]): print("  %-14s %8.1f us avg  (%d launches)" % (name, 1e3 * ms / max(calls, 1), calls))
