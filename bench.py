#!/usr/bin/env python3
"""bench.py -- VO frames/sec on synthetic 640x480 RGB-D streams (BASELINE.json metric).

One "step" = one frame through FrontEnd::AddFrame (ORB detect+describe, map match, P3P-RANSAC,
pose LM, keyframe work incl. local BA) on the HIP path.  Workload at N=1 is BASELINE.json
configs[1]: a single synthetic 640x480 stream, 2000 ORB features, default.yaml parameters
otherwise.  For N>1 every rank tracks its own independent stream (configs[3]): weak scaling,
no data-path collective (frames of one stream are sequentially dependent).

Inputs are rendered on the host before the timed region and are resident in HBM (torch tensors)
when it starts.  ORB of up to --lookahead future frames of the stream runs as one batched launch
chain (detection does not depend on earlier poses); matching/PnP/LM/BA run frame by frame.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, live HIP-event
timing on the kernel's own stream) and `cpu_baseline` (the CPU oracle port, timed on rank 0 at
N=1 on a bounded sample of the same frames).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F64_MFMA_PEAK_TFLOPS = 78.6 # MI355X FP64 matrix peak (AMD datasheet; the guide lists the f32 MFMA figure only)


def level_sizes(W, H, L=8, sf=1.2):
    out = []
    for l in range(L):
        s = np.float32(np.float64(np.float32(sf)) ** l)
        out.append((int(np.rint(np.float32(W) / s)), int(np.rint(np.float32(H) / s))))
    return out


def algorithmic_bytes(W, H, N, M, K, n_hyp, passes=2):
    """Per-frame algorithmic bytes by kernel (SURVEY.md 8d decomposition)."""
    lv = level_sizes(W, H)
    P = sum(w * h for w, h in lv)
    b = {
        "k_gray": 3 * W * H + W * H,                              # BGR in, level 0 out
        "k_resize": sum(lv[l - 1][0] * lv[l - 1][1] + lv[l][0] * lv[l][1] for l in range(1, len(lv))),
        "k_fast_nms": P,                                          # every pyramid pixel read once
        "k_select": 81 * 2 * N + 8 * 2 * N,                       # 9x9 Harris windows of the 2N survivors + list traffic
        "k_blur": 2 * P,                                          # level pyramid in, blurred pyramid out
        "k_describe": (709 + 512) * N + 32 * N + 64 * N,          # IC disc (709 px) + 512 BRIEF samples, kp in/out + descriptor
        "k_match": passes * (32 * M + 32 * N + 8 * M),
        "k_match_gate": passes * (8 * M + 36 * K),
        "k_ransac_hyp": passes * (20 * 4 * n_hyp + 96 * n_hyp),
        "k_ransac_score": passes * (20 * K + 4 * n_hyp),
        "k_ransac_select": passes * (20 * K + 4 * K),
        "k_pose_lm": passes * (20 * K),
        "depth": 2 * W * H * 0 + 2 * 5 * N,                       # depth samples at keypoints
    }
    total_survey = 5 * W * H + 4 * P + 2003 * N + 48 * N + (32 * M + 32 * N + 8 * M) + (20 * K + 4 * n_hyp)
    return b, total_survey, P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--lookahead", type=int, default=32, help="frames per batched ORB launch chain")
    ap.add_argument("--track-batch", type=int, default=8, help="frames tracked speculatively per launch chain (share prior + map)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-ba", action="store_true", help="disable local BA (enable_local_optimization: 0)")
    ap.add_argument("--ba-lag", type=int, default=8, help="0: BA synchronous in AddFrame; L>0: overlapped, merged L frames later (deterministic)")
    ap.add_argument("--hyps", type=int, default=100, help="PnP-RANSAC hypotheses per pass (default.yaml: 100; BASELINE config 3: 2048)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the driver's runs) | gloo (rehearsal of the multi-rank path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses device 0 (needs --dist-backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=330, help="bounded CPU-baseline sample (frames; ~14 s of CPU work at the default)")
    args = ap.parse_args()

    import torch
    from rgbd_visualodometry_amd import capi, system, shard, evaluate as ev
    rank, local_rank, world = shard.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    grp = shard.Group(args.dist_backend, device=torch.device("cuda", local_rank))     # RCCL; only barrier + MAX(time) cross ranks

    W, H, N = 640, 480, args.features
    K, Wm = args.steps, args.warmup
    total = K + Wm
    syn = capi.Synth()
    sp = syn.params(seed=shard.stream_seed(args.seed, rank))
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    t0 = time.time()
    bgr, depth, Twc, stamps = syn.render(sp, 0, total, threads=min(32, threads))
    t_render = time.time() - t0
    d_bgr = torch.from_numpy(bgr).cuda(local_rank)          # inputs resident in HBM
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda(local_rank)
    torch.cuda.synchronize()
    fb, fd = W * H * 3, W * H * 2
    bptr = [d_bgr.data_ptr() + i * fb for i in range(total)]
    dptr = [d_depth.data_ptr() + i * fd for i in range(total)]

    opts = dict(width=W, height=H, number_of_features=N, max_frames_in_flight=args.lookahead, device=local_rank,
                enable_local_optimization=0 if args.no_ba else 1, backend_lag_frames=args.ba_lag, track_batch=args.track_batch, map_capacity=1 << 20, ransac_iterations=args.hyps)
    sysm = system.VoSystem(system.HOST_LIB, **opts)
    assert sysm.backend == "hip-gfx950", sysm.backend

    est = {}

    def drive(i0, i1):
        i = i0
        while i < i1:
            n = min(args.lookahead, i1 - i)
            sysm.prefetch(stamps[i:i + n], bptr[i:i + n], dptr[i:i + n], 3 * W, 2 * W, True)
            for j in range(n):
                ok, T = sysm.add_prefetched()
                est[stamps[i + j]] = T
            i += n

    drive(0, Wm)                                            # warmup (also initialises the map)
    torch.cuda.synchronize()
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drive(Wm, total)
    torch.cuda.synchronize()
    grp.barrier()
    torch.cuda.synchronize()
    elapsed = grp.max_scalar(time.perf_counter() - t0)
    st = sysm.stats()

    # accuracy of the timed run (every rank checks its own stream; rank 0 reports)
    gt = {stamps[i]: capi.pose12_to_tum(Twc[i]) for i in range(total)}
    est_t = {k: capi.pose12_to_tum(v) for k, v in est.items()}
    ate_gpu = ev.ate(gt, est_t)["rmse"]

    out = None
    if rank == 0:
        fps = shard.aggregate_fps(K, world, elapsed)
        # ---- roofline of the dominant kernel -------------------------------------------------------
        # Second pass over the SAME frames (warmup + steps) on a fresh system with per-kernel HIP-event timing
        # enabled on every context stream (tracker + overlapped back-end); not part of `value`.
        prof_sys = system.VoSystem(system.HOST_LIB, **opts)
        L = capi.load(capi.HIP_LIB)
        import ctypes as C
        h = C.c_void_p(prof_sys.context_handle())
        L.check(L.lib.vo_profile_enable(h, 1))
        i = 0
        while i < total:
            n = min(args.lookahead, total - i)
            prof_sys.prefetch(stamps[i:i + n], bptr[i:i + n], dptr[i:i + n], 3 * W, 2 * W, True)
            for _ in range(n):
                prof_sys.add_prefetched()
            i += n
        pst = prof_sys.stats()
        prof_sys.close()                                   # joins the back-end worker; its context is merged on destroy
        sysm_ctx = C.c_void_p(sysm.context_handle())
        names = (C.c_char * 48 * 64)()
        ms = np.zeros(64)
        calls = np.zeros(64, dtype=np.int64)
        nn = C.c_int()
        L.check(L.lib.vo_profile_read(sysm_ctx, C.cast(names, C.c_void_p), ms.ctypes.data, calls.ctypes.data, 64, C.byref(nn)))
        L.check(L.lib.vo_profile_enable(sysm_ctx, 0))
        kern = {names[j].value.decode(): (float(ms[j]), int(calls[j])) for j in range(nn.value)}
        frames_prof = total
        M = max(1, pst["last_candidates"]); Kc = max(1, pst["last_matches"])
        per_frame, b_survey, P = algorithmic_bytes(W, H, N, M, Kc, args.hyps)
        table = {}
        for name, (tms, c) in kern.items():
            if c == 0:
                continue
            row = {"total_ms": round(tms, 3), "launches": c, "avg_us": round(tms / c * 1e3, 2)}
            if name in per_frame:
                bytes_total = per_frame[name] * frames_prof
                row["alg_bytes_per_launch"] = int(bytes_total / c)
                row["GBps"] = round(bytes_total / (tms * 1e-3) / 1e9, 2)
            table[name] = row
        # BA Cholesky: algorithmic flops = sum over BA runs of (D^3/3 + 2 D^2) multiply-adds x trials per run
        runs = max(1, pst["ba_runs"])
        if "k_ba_chol" in table:
            trials = table["k_ba_chol"]["launches"] / runs
            flops = sum(2.0 * ((6.0 * (k + 2)) ** 3 / 3.0 + 2.0 * (6.0 * (k + 2)) ** 2) for k in range(runs)) * trials
            table["k_ba_chol"]["alg_flops_per_launch"] = int(flops / table["k_ba_chol"]["launches"])
            table["k_ba_chol"]["TFLOPps"] = round(flops / (table["k_ba_chol"]["total_ms"] * 1e-3) / 1e12, 5)
        roof = None
        if table:
            dom = max(table, key=lambda k: table[k]["total_ms"])
            t = table[dom]
            if "GBps" in t:
                roof = {"bound": "hbm", "kernel": dom, "achieved": t["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(t["GBps"] / HBM_PEAK_GBS, 6), "traffic": None}
            elif "TFLOPps" in t:
                roof = {"bound": "mfma", "kernel": dom, "achieved": t["TFLOPps"], "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(t["TFLOPps"] / F64_MFMA_PEAK_TFLOPS, 6), "traffic": None}
            else:
                roof = {"bound": "hbm", "kernel": dom, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
            # HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
            # this process); null when no measurement for this kernel is on file
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                roof["traffic"] = pmc["kernels"][dom]["hbm_bytes_per_launch_corrected"]
                roof["traffic_source"] = "profiles/r01_pmc_hbm_traffic.json (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, (2*FETCH+WRITE)*1024)"
            except Exception:
                pass
            roof.update({"avg_launch_us": t["avg_us"], "launches": t["launches"],
                         "note": "single 640x480 stream: the chain is latency bound (small dependent kernels); see `kernels` for the streaming ORB kernels",
                         "kernels": table})
        # ---- CPU baseline: the oracle port on host cores, bounded sample ---------------------------
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            nf = min(args.cpu_frames, total)
            o = system.VoSystem(ORACLE_LIB, **{**opts, "max_frames_in_flight": 1, "track_batch": 1})
            est_c = {}
            tc = time.perf_counter()
            for i in range(nf):
                ok, T = o.add_frame(stamps[i], bgr[i], depth[i])
                est_c[stamps[i]] = capi.pose12_to_tum(T)
            tc = time.perf_counter() - tc
            gt_c = {stamps[i]: gt[stamps[i]] for i in range(nf)}
            est_g = {stamps[i]: est_t[stamps[i]] for i in range(nf)}
            cpu = {"value": round(nf / tc, 3), "unit": "frames/s", "cores": 1 if (args.ba_lag == 0 or args.no_ba) else 2, "kind": "port",
                   "sample": "first %d frames of the same synthetic stream, single thread, oracle/_build/liboracle_vo.so" % nf,
                   "ate_rmse_m": round(ev.ate(gt_c, est_c)["rmse"], 5), "gpu_ate_rmse_m_same_frames": round(ev.ate(gt_c, est_g)["rmse"], 5),
                   "host_cpus": os.cpu_count()}
        out = {
            "metric": "VO frames/sec (640x480 RGB-D)", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": "synthetic 640x480 RGB-D stream per GPU, %d ORB features, default.yaml tracking parameters" % N,
                       "streams_per_gpu": 1, "lookahead_frames": args.lookahead, "track_batch": args.track_batch, "local_ba": (False if args.no_ba else ("synchronous" if args.ba_lag == 0 else "overlapped, merged %d frames later" % args.ba_lag)), "ransac_hypotheses": args.hyps},
            "ate_rmse_m": round(ate_gpu, 5), "keyframes": st["keyframes"], "lost": st["lost"], "map_points": st["map_points"],
            "alg_bytes_per_frame_survey": b_survey, "hbm_frac_whole_frame": round(b_survey * (fps / world) / (HBM_PEAK_GBS * 1e9), 6),
            "render_s": round(t_render, 2),
            "host_stage_ms": {k: round(st[k], 2) for k in ("ms_extract", "ms_track", "ms_keyframe", "ms_backend")},
            "ba": {k: st[k] for k in ("ba_runs", "ba_poses", "ba_fixed", "ba_points", "ba_edges", "ba_outliers")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
