"""BASELINE config 5 on one MI355X: 1280x960 synthetic RGB-D, 8000 features, 2048 RANSAC hypotheses, local BA window ~20 keyframes.
Prints one JSON object: frames/s, ATE/RPE, BA sizes, per-kernel HIP-event table (the numbers behind DESIGN.md's 8e-2 decision)."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=72)
    ap.add_argument("--speed", type=float, default=3.0)
    ap.add_argument("--features", type=int, default=8000)
    ap.add_argument("--hyps", type=int, default=2048)
    ap.add_argument("--device", action="store_true", help="ba_device_graph + map_descriptors_on_device + device_keyframes")
    args = ap.parse_args()
    import torch
    import bench
    from rgbd_visualodometry_amd import capi, system, evaluate as ev
    W, H, n = 1280, 960, args.frames
    syn = capi.Synth()
    sp = syn.params(seed=0, speed=args.speed, width=W, height=H, fx=2 * 517.3, fy=2 * 516.5, cx=2 * 318.6, cy=2 * 255.3)
    bgr, depth, Twc, ts = syn.render(sp, 0, n, threads=min(32, os.cpu_count() or 8))
    db = torch.from_numpy(bgr).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
    torch.cuda.synchronize()
    fb, fd = W * H * 3, W * H * 2
    bptr = [db.data_ptr() + i * fb for i in range(n)]; dptr = [dd.data_ptr() + i * fd for i in range(n)]
    opts = dict(width=W, height=H, fx=2 * 517.3, fy=2 * 516.5, cx=2 * 318.6, cy=2 * 255.3, number_of_features=args.features, max_frames_in_flight=8,
                backend_lag_frames=8, track_batch=4, map_capacity=1 << 20, ransac_iterations=args.hyps)
    if args.device:                                          # graph cut and keyframe bookkeeping on the device tables, as the single 640x480 stream runs
        opts.update(ba_device_graph=1, map_descriptors_on_device=1, device_keyframes=1)

    def drive(s, i0, i1, est=None):
        i = i0
        while i < i1:
            m = min(8, i1 - i)
            s.prefetch(ts[i:i + m], bptr[i:i + m], dptr[i:i + m], 3 * W, 2 * W, True)
            for j in range(m):
                ok, T = s.add_prefetched()
                if est is not None:
                    est[ts[i + j]] = T
            i += m
    pre = system.VoSystem(system.HOST_LIB, **opts); drive(pre, 0, 8); pre.flush(); pre.close()
    s = system.VoSystem(system.HOST_LIB, **opts)
    est = {}
    drive(s, 0, 8, est); s.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    drive(s, 8, n, est); s.flush(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = s.stats()
    s.close()
    L = capi.load(capi.HIP_LIB)
    p = system.VoSystem(system.HOST_LIB, **opts)
    h = C.c_void_p(p.context_handle())
    L.check(L.lib.vo_profile_enable(h, 1))
    drive(p, 0, n); p.flush()
    names = (C.c_char * 48 * 96)(); ms = np.zeros(96); calls = np.zeros(96, dtype=np.int64); nn = C.c_int()
    L.check(L.lib.vo_profile_read(h, C.cast(names, C.c_void_p), ms.ctypes.data, calls.ctypes.data, 96, C.byref(nn)))
    L.check(L.lib.vo_profile_enable(h, 0))
    pst = p.stats(); p.close()
    table = {names[j].value.decode(): {"total_ms": round(float(ms[j]), 3), "launches": int(calls[j]), "avg_us": round(1e3 * float(ms[j]) / max(1, int(calls[j])), 2)} for j in range(nn.value)}
    tf = max(1, pst["tracked_frames"])
    chains = max(1, pst["track_launches"])
    ransac_us = sum(table[k]["total_ms"] for k in ("k_ransac_hyp", "k_ransac_score", "k_ransac_select") if k in table) * 1e3
    ba_us = sum(v["total_ms"] for k, v in table.items() if k.startswith("k_ba_")) * 1e3
    out = {"config": "1280x960 synthetic RGB-D, %d features, %d RANSAC hypotheses, speed %.2g, 1 MI355X" % (args.features, args.hyps, args.speed),
           "frames_timed": n - 8, "frames_per_s": round((n - 8) / el, 1), "ms_per_frame": round(1e3 * el / (n - 8), 3), **bench.accuracy(ev, capi, ts, Twc, est, 0, n),
           "keyframes": st["keyframes"], "ba": {k: st[k] for k in ("ba_runs", "ba_poses", "ba_fixed", "ba_points", "ba_edges", "ba_outliers", "ba_failed", "ba_capped")},
           "avg_per_tracked_frame": {k: round(pst["sum_" + k] / tf, 1) for k in ("active", "candidates", "matches", "ransac_inliers", "lm_iters")},
           "ransac_us_per_launch_chain": round(ransac_us / chains, 1), "ransac_us_per_frame": round(ransac_us / tf, 1),
           "ba_gpu_us_per_run": round(ba_us / max(1, pst["ba_runs"]), 1), "ba_gpu_us_per_frame": round(ba_us / n, 1),
           "kernels": dict(sorted(table.items(), key=lambda kv: -kv[1]["total_ms"]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
