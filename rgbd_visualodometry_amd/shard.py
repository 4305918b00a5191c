"""Multi-GPU sharding of the VO hot path: independent RGB-D streams, one per rank.

Frames of ONE stream are sequentially dependent (pose prior, map state: reference
src/frontend.cpp:96), so a stream never shards by frame; BASELINE.json configs[3] shards whole
streams, one per GPU.  There is no data-path collective: each rank tracks its stream locally, and
only the benchmark's wall time (MAX over ranks) and the per-rank summaries are exchanged.
`torch.distributed` is the plumbing: backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_seed(base_seed: int, rank: int) -> int:
    """Stream identity of a rank: rank r tracks synthetic stream base_seed + r."""
    return base_seed + rank


class Group:
    """Thin wrapper over torch.distributed for the two exchanges the benchmark needs."""

    def __init__(self, backend: Optional[str] = None, device=None):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.dist = None
        self.device = device if backend == "nccl" else None      # gloo reduces host tensors
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "gloo", **kw)

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_scalar(self, v: float) -> float:
        if not self.dist:
            return float(v)
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_reduce_sum_i32(self, arr) -> None:
        """In-place element-wise sum of a host int32 array over the ranks: the exchange step of hypothesis-sharded RANSAC
        (RCCL all-reduce over xGMI with backend nccl, gloo in the CPU tests)."""
        if not self.dist:
            return
        import torch
        t = torch.from_numpy(arr)
        if self.device is not None:
            d = t.to(self.device)
            self.dist.all_reduce(d, op=self.dist.ReduceOp.SUM)
            t.copy_(d.cpu())
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def stream_allreduce_i32(self, device_ptr: int, n: int, hip_stream: int) -> int:
        """vo_stream_allreduce_fn through torch.distributed: an in-place int32 SUM of `n` device-resident values, ENQUEUED behind the work of
        `hip_stream` (the launch chain's stream) -- with backend nccl this is an RCCL all-reduce over xGMI and the host never waits for it.
        Returns 0.  (gloo cannot reduce device memory: the CPU tests use the host callback form, all_reduce_sum_i32.)"""
        if not self.dist:
            return 0
        import torch

        class _Dev:                                          # a view of the caller's device buffer (no copy, no ownership)
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(device_ptr), False), "version": 2}
        t = torch.as_tensor(_Dev(), device=self.device)
        ext = torch.cuda.ExternalStream(int(hip_stream), device=self.device) if hip_stream else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(ext):
            work = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True)
            work.wait()                                      # stream-level dependency for what follows on `ext`; the host does not block
        return 0

    def all_reduce_sum_f64(self, arr) -> None:
        """In-place element-wise sum of a host float64 array over the ranks: the exchanges of the sharded local BA (vo_set_ba_shard)."""
        if not self.dist:
            return
        import torch
        t = torch.from_numpy(arr)
        if self.device is not None:
            d = t.to(self.device)
            self.dist.all_reduce(d, op=self.dist.ReduceOp.SUM)
            t.copy_(d.cpu())
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def stream_allreduce_f64(self, device_ptr: int, n: int, hip_stream: int) -> int:
        """vo_stream_allreduce_f64_fn through torch.distributed: an in-place float64 SUM of `n` device-resident values enqueued behind the work of
        `hip_stream` (backend nccl: an RCCL all-reduce over xGMI of the reduced system S, b_s -- 115 KB at D = 120 -- that the host never waits for)."""
        if not self.dist:
            return 0
        import torch

        class _Dev:
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(device_ptr), False), "version": 2}
        t = torch.as_tensor(_Dev(), device=self.device)
        ext = torch.cuda.ExternalStream(int(hip_stream), device=self.device) if hip_stream else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(ext):
            work = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True)
            work.wait()
        return 0

    def gather_objects(self, obj) -> List:
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()


def aggregate_fps(frames_per_rank: int, world: int, max_elapsed_s: float) -> float:
    """Whole-job throughput: frames all ranks processed / slowest rank's time (weak scaling)."""
    return world * frames_per_rank / max_elapsed_s


def track_stream(lib_path: str, seed: int, n_frames: int, features: int = 500, lookahead: int = 1, local_ba: bool = True,
                 device: int = 0) -> Dict:
    """Render synthetic stream `seed` on the host and track it through the given host library from host
    memory (used by the CPU tests and the CPU baseline; bench.py has its own HBM-resident driver)."""
    import time
    from . import capi, system, evaluate as ev
    syn = capi.Synth()
    sp = syn.params(seed=seed)
    bgr, depth, Twc, ts = syn.render(sp, 0, n_frames, threads=4)
    s = system.VoSystem(lib_path, number_of_features=features, enable_local_optimization=1 if local_ba else 0, device=device)
    gt, est = {}, {}
    t0 = time.perf_counter()
    for i in range(n_frames):
        ok, T = s.add_frame(ts[i], bgr[i], depth[i])
        gt[ts[i]] = capi.pose12_to_tum(Twc[i]); est[ts[i]] = capi.pose12_to_tum(T)
    el = time.perf_counter() - t0
    return {"seed": seed, "frames": n_frames, "elapsed_s": el, "ate_rmse_m": ev.ate(gt, est)["rmse"], "stats": s.stats(),
            "first_pose": est[ts[1]]}
