"""N>1 path on CPU: world_size-2 gloo run of the stream sharding used by bench.py (one independent
stream per rank, MAX-over-ranks timing, no data-path collective)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %r)
from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import shard, system
g = shard.Group("gloo")
res = shard.track_stream(ORACLE_LIB, shard.stream_seed(40, g.rank), 8, features=300, local_ba=False)
g.barrier()
mx = g.max_scalar(res["elapsed_s"])
allr = g.gather_objects({"rank": g.rank, "seed": res["seed"], "frames": res["frames"], "elapsed": res["elapsed_s"],
                         "ate": res["ate_rmse_m"], "pose": res["first_pose"]})
if g.rank == 0:
    print("RESULT " + json.dumps({"world": g.world, "max": mx, "fps": shard.aggregate_fps(8, g.world, mx), "ranks": allr}))
g.close()
''' % ROOT


def test_two_rank_gloo_stream_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0]
    out = json.loads(line[len("RESULT "):])
    assert out["world"] == 2 and len(out["ranks"]) == 2
    a, b = sorted(out["ranks"], key=lambda x: x["rank"])
    assert (a["seed"], b["seed"]) == (40, 41) and a["frames"] == b["frames"] == 8
    assert a["pose"] != b["pose"]                                   # different streams -> different trajectories
    assert abs(out["max"] - max(a["elapsed"], b["elapsed"])) < 1e-9  # MAX over ranks
    assert abs(out["fps"] - 16 / out["max"]) < 1e-9                  # whole-job aggregate, weak scaling
    assert a["ate"] < 0.05 and b["ate"] < 0.05


WORKER_HYP = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi, shard
g = shard.Group("gloo")
L = capi.load(ORACLE_LIB)
syn = capi.Synth()
bgr, depth, Twc, _ = syn.render(syn.params(seed=77), 0, 4, threads=2)      # every rank sees the SAME stream
p = L.default_params(n_features=600, max_frames=2, map_capacity=4096, max_hypotheses=512)
ctx = L.context(p)
ctx.upload(0, bgr[0], depth[0]); ctx.upload(1, bgr[3], depth[3]); ctx.orb(0, 2)
k0, d0 = ctx.orb_fetch(0)
ok = k0["depth_raw"] > 0
z = k0["depth_raw"][ok] / 5000.0
pw = np.stack([(k0["x"][ok] - p.cx) * z / p.fx, (k0["y"][ok] - p.cy) * z / p.fy, z], 1)
idx = np.arange(len(pw), dtype=np.int32)
ctx.map_upsert(idx, pw, pw / np.linalg.norm(pw, axis=1, keepdims=True), d0[ok], np.zeros(len(pw), np.uint8))
ctx.map_set_active(idx)
tp = L.default_track_params(n_hyp=256)
ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], float)
r0, m0 = ctx.track(1, ident, tp)                                            # un-sharded
n_ex = [0]
def ex(a):
    n_ex[0] += 1
    g.all_reduce_sum_i32(a)
ctx.set_hypothesis_shard(g.rank, g.world, ex)
r1, m1 = ctx.track(1, ident, tp)                                            # hypotheses h %% world == rank scored here
same = (np.array_equal(np.array(r0.T_cw), np.array(r1.T_cw)) and r0.n_ransac_inliers == r1.n_ransac_inliers and r0.best_hypothesis == r1.best_hypothesis
        and r0.ransac_iters == r1.ransac_iters and np.array_equal(m0, m1))
allr = g.gather_objects({"rank": g.rank, "same": bool(same), "exchanges": n_ex[0], "inliers": int(r1.n_ransac_inliers), "pose": [float(v) for v in r1.T_cw]})
if g.rank == 0:
    print("RESULT " + json.dumps(allr))
g.close()
''' % ROOT


def test_two_rank_gloo_hypothesis_shard(tmp_path):
    """SURVEY.md 8e-2 over a real collective: two ranks track the same frame, each scores half of the RANSAC hypotheses, one
    all-reduce (sum) of the count vector per pass; both ranks end with the un-sharded result."""
    script = tmp_path / "worker_hyp.py"
    script.write_text(WORKER_HYP)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29578", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    assert len(out) == 2 and all(o["same"] for o in out)
    assert out[0]["exchanges"] == out[1]["exchanges"] == 2          # coarse + fine pass
    assert out[0]["pose"] == out[1]["pose"] and out[0]["inliers"] > 100


WORKER_BA = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi, shard
sys.path.insert(0, os.path.join(%r, "tests"))
from test_ba_shard import ba_problem
g = shard.Group("gloo")
L = capi.load(ORACLE_LIB)
pr = ba_problem(np.random.default_rng(11), 9, 400, 6, L.default_params())        # every rank holds the SAME problem
ctx = L.context(L.default_params(map_capacity=1024))
p0, x0, f0, r0 = ctx.local_ba(pr[0], 6, *pr[1:])                                  # un-sharded
n_ex = [0]
def ex(a):
    n_ex[0] += 1
    g.all_reduce_sum_f64(a)
ctx.set_ba_shard(g.rank, g.world, ex)
p1, x1, f1, r1 = ctx.local_ba(pr[0], 6, *pr[1:])                                  # this rank linearises the points k %% world == rank
allr = g.gather_objects({"rank": g.rank, "exchanges": n_ex[0], "flags_same": bool(np.array_equal(f0, f1)), "dpose": float(np.abs(p0 - p1).max()), "dpts": float(np.abs(x0 - x1).max()),
                         "iters": [int(r0.lm_iters), int(r1.lm_iters)], "chi": [float(r0.chi2_final), float(r1.chi2_final)], "pose": [float(v) for v in p1.ravel()], "culled": int((f1 != 0).sum())})
if g.rank == 0:
    print("RESULT " + json.dumps(allr))
g.close()
''' % (ROOT, ROOT)


def test_two_rank_gloo_ba_edge_shard(tmp_path):
    """SURVEY.md 8e item 2 (e-3) over a real collective: two ranks hold the same local BA, each linearises the edges of its points, three all-reduces
    (sum) per LM step -- the second one is the reduced system S, b_s -- and one for the result; both ranks end with the un-sharded result."""
    script = tmp_path / "worker_ba.py"
    script.write_text(WORKER_BA)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29579", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    assert len(out) == 2
    for o in out:
        assert o["flags_same"] and o["dpose"] < 1e-6 and o["dpts"] < 1e-5 and o["culled"] > 0
        assert abs(o["iters"][0] - o["iters"][1]) <= 2 and abs(o["chi"][0] - o["chi"][1]) <= 1e-6 * max(1.0, o["chi"][0])
    assert out[0]["exchanges"] == out[1]["exchanges"] and out[0]["exchanges"] >= 3 * 12 + 1 and (out[0]["exchanges"] - 1) % 3 == 0
    assert out[0]["pose"] == out[1]["pose"]                        # the ranks stayed in lockstep: bit-identical poses
