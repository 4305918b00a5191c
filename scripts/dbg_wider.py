import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from rgbd_visualodometry_amd import capi
import test_stream_group as T
L = capi.load(capi.HIP_LIB)
rng = np.random.default_rng(17)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 150
t, Ts, X, slots, flags, obs, dead, free = T._resident_scene(L, rng, n_kf=320, n_pts=700, n_free=nf)
c = L.context(L.default_params(n_features=64, map_capacity=64))
g = c.resident_graph(t, free)
print("graph: poses", len(g["pose_kf"]), "points", len(g["point_slots"]), "edges", len(g["edge_obs"]))
try:
    po, sl, pt, cu, r = c.local_ba_resident(t, free, cap_culled=1 << 18)
    print("resident: culled", len(cu), "chi2", r.chi2_initial, "->", r.chi2_final, "iters", r.lm_iters)
except Exception as e:
    print("resident failed:", e)
poses = np.array([Ts[k] for k in g["pose_kf"]])
pos_now = {int(s): x for s, x in zip(slots, np.array(T.t_positions(L, t, slots)))}
P0 = np.array([pos_now[int(s)] for s in g["point_slots"]])
pw, xw, fw, rw = c.local_ba(poses, len(free), P0, g["edge_pose"], g["edge_point"], g["edge_uv"])
print("explicit: culled", int(np.count_nonzero(fw & 3)), "chi2", rw.chi2_initial, "->", rw.chi2_final, "iters", rw.lm_iters)
