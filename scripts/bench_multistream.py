"""Several independent VO streams on ONE MI355X (one host thread + one vo_ctx + one BA worker per stream).

Not the official bench line (bench.py measures BASELINE.json config 2: a single stream): this shows what the GPU delivers when
the per-frame dependency chain of one stream no longer bounds it -- the per-GPU batched figure SURVEY 8(d) asks for.
Prints one JSON line: {"streams": S, "value": total frames/s, "per_stream": [...], "hbm_frac_whole_frame": ...}."""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--ba-lag", type=int, default=8)
    ap.add_argument("--speed", type=float, default=0.0, help="camera speed factor of the synthetic trajectories (0: the renderer's default; 3 = the bench line's keyframe cadence)")
    ap.add_argument("--device-keyframes", type=int, default=-1, help="1: the keyframe bookkeeping on the device tables, graph cut on the device (the bench line's path); -1: the host layer's defaults")
    args = ap.parse_args()
    import torch
    from rgbd_visualodometry_amd import capi, system, evaluate as ev
    W, H, S, total = 640, 480, args.streams, args.steps + args.warmup
    syn = capi.Synth()
    data = []
    for s in range(S):
        bgr, depth, Twc, ts = syn.render(syn.params(seed=s, speed=args.speed) if args.speed > 0 else syn.params(seed=s), 0, total, threads=min(32, os.cpu_count() or 8))
        db = torch.from_numpy(bgr).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
        data.append((db, dd, Twc, ts))
    torch.cuda.synchronize()
    fb, fd = W * H * 3, W * H * 2
    systems = [system.VoSystem(system.HOST_LIB, width=W, height=H, number_of_features=args.features, max_frames_in_flight=32,
                               backend_lag_frames=args.ba_lag, track_batch=8, map_capacity=1 << 18,
                               **({} if args.device_keyframes < 0 else dict(ba_device_graph=1, map_descriptors_on_device=1, device_keyframes=args.device_keyframes, enable_local_optimization=1, ransac_iterations=100))) for _ in range(S)]
    est = [dict() for _ in range(S)]

    def drive(s, i0, i1):
        db, dd, _, ts = data[s]
        i = i0
        while i < i1:
            n = min(32, i1 - i)
            systems[s].prefetch(ts[i:i + n], [db.data_ptr() + j * fb for j in range(i, i + n)], [dd.data_ptr() + j * fd for j in range(i, i + n)], 3 * W, 2 * W, True)
            for j in range(n):
                ok, T = systems[s].add_prefetched()
                est[s][ts[i + j]] = T
            i += n

    def run_all(i0, i1):
        th = [threading.Thread(target=drive, args=(s, i0, i1)) for s in range(S)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run_all(0, args.warmup)
    el = run_all(args.warmup, total)
    ates = []
    for s in range(S):
        _, _, Twc, ts = data[s]
        gt = {ts[i]: capi.pose12_to_tum(Twc[i]) for i in range(total)}
        ates.append(round(ev.ate(gt, {k: capi.pose12_to_tum(v) for k, v in est[s].items()})["rmse"], 5))
    fps = S * args.steps / el
    print(json.dumps({"metric": "VO frames/sec (640x480 RGB-D), several streams on one GPU", "streams": S, "value": round(fps, 1), "unit": "frames/s",
                      "steps_per_stream": args.steps, "elapsed_s": round(el, 4), "ate_rmse_m": ates, "lost": [x.stats()["lost"] for x in systems],
                      "hbm_frac_whole_frame": round(10.16e6 * fps / 8e12, 5)}))
    for x in systems: x.close()


if __name__ == "__main__":
    main()
