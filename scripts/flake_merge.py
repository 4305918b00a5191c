"""Run-to-run differences of two HIP solves of one resident BA (tests/test_stream_group.py::test_resident_merge_on_the_device_equals_the_host_write_back's scene), N times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rgbd_visualodometry_amd import capi
import test_stream_group as T
L = capi.load(capi.HIP_LIB)
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(23)
    t, Ts, X, slots, flags, obs, dead, free = T._resident_scene(L, rng, n_kf=12, n_pts=500, n_free=5)
    c = L.context(L.default_params(n_features=64, map_capacity=64))
    po, sl, pt, cu, r = c.local_ba_resident(t, free)
    cur = (po, pt, np.sort(cu), r.lm_iters, r.chi2_final)
    if ref is None:
        ref = cur
    print("rep %d: lm_iters %d culled %d chi %.12e | vs first: dpose %.2e dpts %.2e culled_equal %s iters_equal %s" % (rep, r.lm_iters, len(cu), r.chi2_final, np.abs(po - ref[0]).max(), np.abs(pt - ref[1]).max(), np.array_equal(cur[2], ref[2]), cur[3] == ref[3]))
    c.close(); t.close()
