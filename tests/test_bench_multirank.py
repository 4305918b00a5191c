"""`bench.py --gpus N` must mean N ranks whichever way it is started: called directly it starts its own
ranks as a child process before anything touches the GPU, under a launcher it refuses a world size that differs from --gpus.
CPU rehearsal: --dry-run makes no GPU call, --dist-backend gloo carries the barrier / MAX / SUM / gather."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GPU_MAX_HW_QUEUES")}
    env.update(OMP_NUM_THREADS="1", **kw)
    return env


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "7", "--warmup", "3", "--dist-backend", "gloo", "--dry-run"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line on stdout, everything else went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["hyp_shard"] == {"dry": True, "exchanges": 1, "sum_ok": True} and d["ba_shard"] == {"dry": True, "exchanges": 1, "sum_ok": True}      # the shard legs' collectives (int32 and f64 sums) over the same group
    assert d["dry_run"] is True and d["value"] is None    # a rehearsal can never be mistaken for a measurement
    dist = d["distributed"]
    assert dist["world_size"] == 2 and dist["backend"] == "gloo" and dist["allreduce_sum_of_ones"] == 2
    ranks = sorted(dist["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == [0, 1] and [x["local_rank"] for x in ranks] == [0, 1]
    assert ranks[0]["stream_seed"] + 1 == ranks[1]["stream_seed"]          # one independent stream per rank
    assert d["max_elapsed_s"] >= max(x["own_elapsed_s"] for x in ranks) - 1e-3   # MAX over ranks (rank 1 'works' twice as long)
    assert ranks[1]["own_elapsed_s"] > ranks[0]["own_elapsed_s"]
    # the two-hardware-queue default is the single-rank run's (its several-streams leg); ranks of an N > 1 job keep the runtime's own
    assert all(x["hip_hw_queues_env"] is None for x in ranks)


def test_bench_under_a_launcher_refuses_a_world_size_that_is_not_gpus():
    # torchrun with 2 ranks but --gpus 1: every rank must exit non-zero and no line may be printed
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29591", BENCH, "--gpus", "1", "--dist-backend", "gloo", "--dry-run"]
    r = subprocess.run(cmd, env=_env(MASTER_ADDR="127.0.0.1", MASTER_PORT="29591"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "WORLD_SIZE=2" in r.stderr


def test_bench_under_a_launcher_with_matching_gpus_prints_the_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29592", BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--dist-backend", "gloo", "--dry-run"]
    r = subprocess.run(cmd, env=_env(MASTER_ADDR="127.0.0.1", MASTER_PORT="29592"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["distributed"]["allreduce_sum_of_ones"] == 2 and len(d["distributed"]["ranks"]) == 2


def test_single_rank_dry_run_needs_no_launcher_and_no_gpu():
    r = subprocess.run([sys.executable, BENCH, "--dist-backend", "gloo", "--dry-run"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["distributed"]["world_size"] == 1 and d["distributed"]["allreduce_sum_of_ones"] == 1
    assert d["distributed"]["ranks"][0]["hip_hw_queues_env"] == "2"


@pytest.mark.gpu
def test_bench_gpus_2_tracks_two_streams_on_a_one_gpu_box():
    """The real N > 1 path with GPU work, as far as a one-GPU box allows: two self-started ranks share device 0 (--same-device), gloo carries the
    barrier / MAX / SUM / gather; each rank tracks its own stream and the line reports the whole job."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--same-device", "--steps", "12", "--warmup", "4", "--prologue", "40"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["scaling"] == "weak" and d["value"] > 0 and d.get("dry_run") is None
    dist = d["distributed"]
    assert dist["world_size"] == 2 and dist["allreduce_sum_of_ones"] == 2 and len(dist["ranks"]) == 2
    assert all(x["device"] == 0 and x["frames_per_s"] > 0 for x in dist["ranks"])
    assert abs(d["value"] - 2 * 12 / (d["ms_per_step"] * 12e-3)) / d["value"] < 0.01      # whole job: both ranks' frames over the slowest rank's time
    # the legs that shard WITHIN a stream / a BA, here over gloo host callbacks: both ranks end with the un-sharded results
    hs, bs = d["hyp_shard"], d["ba_shard"]
    assert hs["identical_to_unsharded"] and hs["identical_on_every_rank"] and hs["exchanges_per_frame"] == 2 and hs["frames_per_s"] > 0
    assert bs["flags_identical_to_unsharded"] and bs["identical_on_every_rank"] and bs["max_pose_diff"] < 1e-6 and bs["exchanges_per_ba"] >= 3 * 10 + 1
