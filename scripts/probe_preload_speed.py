"""How fast does vo_frames_preload move a look-ahead batch (pinned host memory -> the context's slab)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_visualodometry_amd import system
W, H, N = 640, 480, 32
opts = dict(width=W, height=H, number_of_features=2000, max_frames_in_flight=32)
s = system.VoSystem(system.HOST_LIB, **opts)
hb = torch.zeros((N, H, W, 3), dtype=torch.uint8).pin_memory(); hd = torch.zeros((N, H, W), dtype=torch.int16).pin_memory()
fb, fd = W * H * 3, W * H * 2
bp = [hb.data_ptr() + i * fb for i in range(N)]; dp = [hd.data_ptr() + i * fd for i in range(N)]
torch.cuda.synchronize()
for n in (32, 20, 32, 32, 8, 32):
    t = time.perf_counter(); s.preload(bp[:n], dp[:n], 3 * W, 2 * W); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    mb = n * (fb + fd) / 1e6
    print("preload %2d frames (%.1f MB): call %.3f ms, copies done after %.3f ms -> %.1f GB/s" % (n, mb, 1e3 * (t1 - t), 1e3 * (t2 - t), mb / 1e3 / (t2 - t)))
s.close()
