"""Debug: run the HIP run_vo driver several times on the same PNG dataset (sequential / look-ahead) and print trajectory differences."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_visualodometry_amd import capi, dataset, evaluate as ev
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "rgbd_visualodometry_amd", "host", "app", "run_vo")
root = tempfile.mkdtemp()
syn = capi.Synth()
bgr, depth, Twc, ts = syn.render(syn.params(seed=21), 0, 16, threads=8)
dataset.write_tum_dataset(root, bgr, depth, ts, Twc)
def run(tag, **over):
    d = os.path.join(root, tag); os.makedirs(d, exist_ok=True)
    cfg, out = os.path.join(d, "cfg.yaml"), os.path.join(d, "traj.txt")
    dataset.write_config(cfg, root, out, **over)
    r = subprocess.run([HIP_BIN, cfg], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    t = ev.read_stamped_file(out)
    return np.array([[float(v) for v in t[k]] for k in sorted(t)])
a1 = run("a1", number_of_features=800); a2 = run("a2", number_of_features=800)
c1 = run("c1", number_of_features=800, lookahead_frames=8, decode_threads=4, track_batch=4)
c2 = run("c2", number_of_features=800, lookahead_frames=8, decode_threads=4, track_batch=4)
n1 = run("n1", number_of_features=800, enable_local_optimization=0)
n2 = run("n2", number_of_features=800, enable_local_optimization=0, lookahead_frames=8, decode_threads=4, track_batch=4)
l1 = run("l1", number_of_features=800, lookahead_frames=8, decode_threads=4, track_batch=1)
print("seq vs seq      ", np.abs(a1 - a2).max())
print("look vs look    ", np.abs(c1 - c2).max())
print("seq vs look     ", np.abs(a1 - c1).max(), np.abs(a1 - c1).max(axis=1))
print("noBA seq vs look", np.abs(n1 - n2).max())
print("seq vs look(batch 1)", np.abs(a1 - l1).max())
