"""debug: which points of the small BA problem differ between the HIP path and the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from bench_ba import make_problem
from rgbd_visualodometry_amd import capi
from oracle import ORACLE_LIB
H = capi.load(capi.HIP_LIB); O = capi.load(ORACLE_LIB)
p = H.default_params(map_capacity=1024)
nP, nfree, nX, window = 6, 4, 400, 6
prob = make_problem(p, nP, nfree, nX, window, 11)
res = []
for L in (H, O):
    ctx = L.context(p)
    res.append(ctx.local_ba(prob[0], nfree, prob[1], prob[2], prob[3], prob[4]))
    ctx.close()
(ph, xh, fh, rh), (po, xo, fo, ro) = res
d = np.abs(xh - xo).max(axis=1)
ep, el = prob[2], prob[3]
order = np.argsort(-d)[:12]
for k in order:
    e = np.where(el == k)[0]
    print("point", k, "diff %.3e" % d[k], "poses", ep[e].tolist(), "flags", fo[e].tolist(), "moved(oracle) %.3e" % np.abs(xo[k] - prob[1][k]).max())
print("n points with diff > 1e-9:", int((d > 1e-9).sum()), "of", nX, " pose diff %.3e" % np.abs(ph - po).max())
