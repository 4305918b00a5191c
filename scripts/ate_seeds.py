"""ATE / RPE over seeds 0-3 for the three schedules of the same local BA: GPU with the BA merged 8 frames late (the bench's default),
GPU with the BA synchronous inside AddFrame (lag 0), and the CPU restatement with the synchronous BA -- at default.yaml's 500 features and at the bench's 2000.
Prints one JSON record (profiles/r06_ate_seeds.json); frames/s of the two GPU schedules ride along (what lag 0 costs).

    python scripts/ate_seeds.py [--frames 300] [--seeds 0,1,2,3] [--features 500,2000] [--no-cpu]"""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--seeds", default="0,1,2,3")
    ap.add_argument("--features", default="500,2000")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lags", default="8,0", help="GPU schedules: frames behind the keyframe at which its BA is merged (0: synchronous)")
    args = ap.parse_args()
    import torch
    from rgbd_visualodometry_amd import capi, system, evaluate as ev
    from oracle import ORACLE_LIB                           # the checker: the CPU column of the table
    seeds = [int(s) for s in args.seeds.split(",")]
    lags = [int(s) for s in args.lags.split(",")]
    feats = [int(s) for s in args.features.split(",")]
    W, H, n = 640, 480, args.frames
    syn = capi.Synth()
    data = {}
    for sd in seeds:
        data[sd] = syn.render(syn.params(seed=sd, speed=3.0), 0, n, threads=16)

    def acc(stamps, Twc, est):
        gt = {stamps[i]: capi.pose12_to_tum(Twc[i]) for i in range(n)}
        e = {stamps[i]: capi.pose12_to_tum(est[i]) for i in range(n)}
        out = {"ate": ev.ate(gt, e)["rmse"]}
        tg = {k: ev.pose_matrix([k] + list(v)) for k, v in gt.items()}; te = {k: ev.pose_matrix([k] + list(v)) for k, v in e.items()}
        r = ev.rpe_summary(ev.rpe(tg, te, fixed_delta=True, delta=1.0, delta_unit="s"))
        out["rpe_t"] = r["trans_rmse"]; out["rpe_r"] = r["rot_deg_rmse"]
        return out

    def gpu_run(sd, N, lag):
        bgr, depth, Twc, stamps = data[sd]
        d_b = torch.from_numpy(bgr).cuda(); d_d = torch.from_numpy(depth.view(np.int16)).cuda()
        bp = [d_b.data_ptr() + i * W * H * 3 for i in range(n)]; dp = [d_d.data_ptr() + i * W * H * 2 for i in range(n)]
        s = system.VoSystem(system.HOST_LIB, width=W, height=H, number_of_features=N, max_frames_in_flight=32, enable_local_optimization=1, backend_lag_frames=lag, track_batch=8,
                            map_capacity=1 << 20, ransac_iterations=100, ba_device_graph=1, map_descriptors_on_device=1, device_keyframes=1)
        est = []
        t0 = time.perf_counter()
        i = 0
        while i < n:
            k = min(32, n - i)
            s.prefetch(stamps[i:i + k], bp[i:i + k], dp[i:i + k], 3 * W, 2 * W, True)
            for _ in range(k):
                est.append(s.add_prefetched()[1])
            i += k
        s.flush(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = s.stats(); s.close()
        return dict(acc(stamps, Twc, est), fps=n / dt, keyframes=st["keyframes"], lost=st["lost"])

    def cpu_run(sd, N, out, key):
        bgr, depth, Twc, stamps = data[sd]
        s = system.VoSystem(ORACLE_LIB, width=W, height=H, number_of_features=N, max_frames_in_flight=1, enable_local_optimization=1, backend_lag_frames=0, track_batch=1, ransac_iterations=100)
        est = [s.add_frame(stamps[i], bgr[i], depth[i])[1] for i in range(n)]
        st = s.stats(); s.close()
        out[key] = dict(acc(stamps, Twc, est), keyframes=st["keyframes"], lost=st["lost"])

    rec = {"frames": n, "seeds": seeds, "rows": []}
    cpu_out, ths = {}, []
    if not args.no_cpu:                                     # the CPU runs go first, one thread each, beside the GPU runs
        for N in feats:
            for sd in seeds:
                th = threading.Thread(target=cpu_run, args=(sd, N, cpu_out, (N, sd))); th.start(); ths.append(th)
    gpu_out = {}
    for N in feats:
        for lag in lags:
            gpu_run(seeds[0], N, lag)                       # scratch of this size class
            for sd in seeds:
                gpu_out[(N, lag, sd)] = gpu_run(sd, N, lag)
                print("gpu N=%d lag=%d seed=%d: %s" % (N, lag, sd, {k: round(v, 4) for k, v in gpu_out[(N, lag, sd)].items()}), file=sys.stderr, flush=True)
    for th in ths:
        th.join()
    for N in feats:
        row = {"features": N}
        cols = [("gpu_lag%d" % lg, (lambda sd, lg=lg: gpu_out[(N, lg, sd)])) for lg in lags] + [("cpu_sync", lambda sd: cpu_out.get((N, sd)))]
        for name, get in cols:
            vals = [get(sd) for sd in seeds]
            if any(v is None for v in vals):
                continue
            row[name] = {"ate_per_seed": [round(v["ate"], 5) for v in vals], "ate_mean": round(float(np.mean([v["ate"] for v in vals])), 5), "ate_std": round(float(np.std([v["ate"] for v in vals])), 5),
                         "rpe_t_mean": round(float(np.mean([v["rpe_t"] for v in vals])), 5), "rpe_r_mean": round(float(np.mean([v["rpe_r"] for v in vals])), 4), "lost": int(sum(v["lost"] for v in vals))}
            if "fps" in vals[0]:
                row[name]["fps_mean"] = round(float(np.mean([v["fps"] for v in vals])), 1)
        ref = row.get("cpu_sync") or row.get("gpu_lag0")
        for lg in lags:
            if ref:
                row["ate_ratio_lag%d_vs_%s" % (lg, "cpu_sync" if "cpu_sync" in row else "lag0")] = round(row["gpu_lag%d" % lg]["ate_mean"] / ref["ate_mean"], 3)
        rec["rows"].append(row)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
