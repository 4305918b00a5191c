import torch
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from rgbd_visualodometry_amd import capi
I12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], float)
L = capi.load(sys.argv[1])
nX, nK = 3000, 40
t = L.context(L.default_params(map_capacity=4096))
torch.cuda.init()
bg = torch.cuda.Stream()
A = torch.randn(2048, 2048, device='cuda')
def load():
    with torch.cuda.stream(bg):
        for _ in range(64): torch.mm(A, A)

rng = np.random.default_rng(5)
X = rng.uniform(-1.0, 1.0, (nX, 3)) + [0, 0, 4.0]
t.map_upsert(np.arange(nX, dtype=np.int32), X, np.tile([0, 0, 1.0], (nX, 1)), np.zeros((nX, 32), np.uint8), np.zeros(nX, np.uint8))
t.kf_set_pose(np.arange(nK), np.tile(I12, (nK, 1)))
uv = np.tile([320.0, 240.0], (nX, 1))
for k in range(nK):
    t.obs_append([k] * nX, np.arange(nX), uv)
wide, narrow = list(range(nK)), list(range(10, nK))
bad = []
for rep in range(10):
  for parity in (0, 1):
    L.lib.vo_scan_call_number(nX - 40 - parity)
    for i in range(80):
        if i % 4 == 0: load()
        a = t.map_set_active_covisible(wide, nX, 100)
        n = t.map_set_active_covisible(narrow, nX, 100)
        if a != nX or n != nX: bad.append((rep, parity, i, a, n, L.lib.vo_scan_call_number(-1)))
print(sys.argv[1].split('/')[-1], "wrong results:", len(bad), bad[:6])
