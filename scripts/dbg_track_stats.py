"""Developer probe: per-frame tracking counts (candidates / matches / RANSAC inliers / LM inliers) over the bench stream."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_visualodometry_amd import capi, system
syn = capi.Synth(); sp = syn.params(seed=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 330
bgr, depth, Twc, ts = syn.render(sp, 0, n)
db = torch.from_numpy(bgr).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
s = system.VoSystem(system.HOST_LIB, width=640, height=480, number_of_features=2000, max_frames_in_flight=32, backend_lag_frames=8, track_batch=8, map_capacity=1 << 20)
fb, fd = 640 * 480 * 3, 640 * 480 * 2
rows = []
i = 0
while i < n:
    k = min(32, n - i)
    s.prefetch(ts[i:i + k], [db.data_ptr() + j * fb for j in range(i, i + k)], [dd.data_ptr() + j * fd for j in range(i, i + k)], 1920, 1280, True)
    for _ in range(k):
        s.add_prefetched(); st = s.stats(); rows.append((st["last_candidates"], st["last_matches"], st["last_ransac_inliers"], st["last_lm_inliers"]))
    i += k
r = np.array(rows[1:])
for name, col in zip(("candidates", "matches", "ransac inliers", "lm inliers"), r.T):
    print("%-15s mean %7.0f  p50 %6.0f  p90 %6.0f  max %6.0f" % (name, col.mean(), np.percentile(col, 50), np.percentile(col, 90), col.max()))
