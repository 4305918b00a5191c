"""How uneven is the local BA's work per point?  Tracks the bench workload for N frames with the map on the device, fetches the observation table and prints the
distribution of observations per point inside the last local-BA window (the points the newest 25 keyframes see), and -- what bounds k_ba_upchi2 / k_ba_lin2, whose
lanes walk a point's edges four at a time -- the largest degree per 128-point workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import ctypes as C
import numpy as np
import torch
from rgbd_visualodometry_amd import capi, system

n, W, H, N = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 640, 480, 2000
syn = capi.Synth()
bgr, depth, Twc, stamps = syn.render(syn.params(seed=0, speed=3.0), 0, n, threads=16)
d_b = torch.from_numpy(bgr).cuda(); d_d = torch.from_numpy(depth.view(np.int16)).cuda()
bp = [d_b.data_ptr() + i * W * H * 3 for i in range(n)]; dp = [d_d.data_ptr() + i * W * H * 2 for i in range(n)]
s = system.VoSystem(system.HOST_LIB, width=W, height=H, number_of_features=N, max_frames_in_flight=32, enable_local_optimization=1, backend_lag_frames=8, track_batch=8,
                    map_capacity=1 << 20, ransac_iterations=100, ba_device_graph=1, map_descriptors_on_device=1, device_keyframes=1)
i = 0
while i < n:
    k = min(32, n - i)
    s.prefetch(stamps[i:i + k], bp[i:i + k], dp[i:i + k], 3 * W, 2 * W, True)
    for _ in range(k):
        s.add_prefetched()
    i += k
s.flush(); torch.cuda.synchronize()
st = s.stats()
L = capi.load(capi.HIP_LIB)
h = C.c_void_p(s.context_handle())
no = C.c_int64(); na = C.c_int32()
L.check(L.lib.vo_tables_fetch(h, 0, 0, None, None, None, None, C.byref(no), 0, 0, None, None, None, None, None, 0, C.byref(na)), "fetch")
m = no.value
kf = np.zeros(m, np.int32); mp = np.zeros(m, np.int32); al = np.zeros(m, np.uint8)
L.check(L.lib.vo_tables_fetch(h, 0, m, kf.ctypes.data, mp.ctypes.data, None, al.ctypes.data, C.byref(no), 0, 0, None, None, None, None, None, 0, C.byref(na)), "fetch")
live = al != 0
nk = int(kf.max()) + 1
free = np.arange(max(0, nk - 23), nk)
in_free = np.isin(kf, free) & live
pts = np.unique(mp[in_free])
deg_all = np.bincount(mp[live], minlength=int(mp.max()) + 1)[pts]                       # every live observation of the graph's points (free + fixed observers)
deg_free = np.bincount(mp[in_free], minlength=int(mp.max()) + 1)[pts]
print("frames %d keyframes %d observations %d | window of the newest 23 keyframes: %d points, %d edges (%d to free poses)" % (n, st["keyframes"], m, len(pts), int(deg_all.sum()), int(deg_free.sum())))
for name, d in (("edges per point", deg_all), ("edges to free poses per point", deg_free)):
    print("%s: mean %.2f  p50 %d  p90 %d  p99 %d  max %d;  points with > 8: %.1f %%, > 16: %.1f %%" % (name, d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(), 100 * (d > 8).mean(), 100 * (d > 16).mean()))
g = [deg_all[i:i + 128].max() for i in range(0, len(deg_all), 128)]
it = np.ceil(np.array(g) / 4.0)
print("per 128-point workgroup (ascending map slot = graph order): largest degree mean %.1f  p50 %d  max %d  -> serial edge rounds per lane: mean %.1f, max %d (a workgroup of points with <= 4 edges: 1)" % (np.mean(g), np.percentile(g, 50), max(g), it.mean(), it.max()))
pairs = (deg_free * (deg_free + 1) // 2).sum()
print("pairs of the Schur plan: %d (%.2f per edge to a free pose)" % (pairs, pairs / max(1, deg_free.sum())))
s.close()
