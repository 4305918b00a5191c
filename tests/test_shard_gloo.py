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
