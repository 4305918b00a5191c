"""Experiment: S independent streams on one MI355X, grouped (shared launch chains) or not; prints where the host threads spend time."""
import argparse, json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")            # as bench.py (profiles/r05_hw_queues_ab.txt); read by the HIP runtime when it starts
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,4,8")
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--speed", type=float, default=3.0)
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--lag", type=int, default=8)
    ap.add_argument("--same-seed", action="store_true")
    ap.add_argument("--modes", default="group,separate")
    ap.add_argument("--host-graph", action="store_true")
    ap.add_argument("--host-keyframes", action="store_true")
    ap.add_argument("--profile", action="store_true", help="per-kernel HIP-event timing (all contexts and engines)")
    args = ap.parse_args()
    import torch
    from rgbd_visualodometry_amd import capi, system
    W, H, total = 640, 480, args.frames
    syn = capi.Synth()
    smax = max(int(v) for v in args.streams.split(","))
    data = []
    for s in range(smax):
        bgr, depth, Twc, ts = syn.render(syn.params(seed=0 if args.same_seed else s, speed=args.speed), 0, total, threads=min(32, os.cpu_count() or 8))
        data.append((torch.from_numpy(bgr).cuda(), torch.from_numpy(depth.view(np.int16)).cuda(), ts))
    torch.cuda.synchronize()
    fb, fd = W * H * 3, W * H * 2
    opts = dict(width=W, height=H, number_of_features=2000, max_frames_in_flight=32, backend_lag_frames=args.lag, track_batch=8, map_capacity=1 << 19,
                enable_local_optimization=0 if args.no_ba else 1, ba_device_graph=0 if args.host_graph else 1, map_descriptors_on_device=1, device_keyframes=0 if (args.host_graph or args.host_keyframes) else 1)
    for S in [int(v) for v in args.streams.split(",")]:
        for mode in args.modes.split(","):
            grp = system.StreamGroup(system.HOST_LIB, 0, 128) if mode == "group" else None
            syss = [system.VoSystem(system.HOST_LIB, **opts) for _ in range(S)]
            if grp:
                for x in syss:
                    grp.join(x)
            bar = threading.Barrier(S + 1)
            if args.profile:
                import ctypes as C
                L = capi.load(capi.HIP_LIB); h0 = C.c_void_p(syss[0].context_handle()); L.check(L.lib.vo_profile_enable(h0, 1))

            def drive(k, i0, i1):
                db, dd, ts = data[k]
                i = i0
                while i < i1:
                    n = min(32, i1 - i)
                    syss[k].prefetch(ts[i:i + n], [db.data_ptr() + j * fb for j in range(i, i + n)], [dd.data_ptr() + j * fd for j in range(i, i + n)], 3 * W, 2 * W, True)
                    for _ in range(n):
                        syss[k].add_prefetched()
                    i += n

            def run(k):
                drive(k, 0, 32); syss[k].flush()
                bar.wait()
                drive(k, 32, total); syss[k].flush()
            ths = [threading.Thread(target=run, args=(k,)) for k in range(S)]
            for t in ths: t.start()
            bar.wait()
            t0 = time.perf_counter()
            for t in ths: t.join()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = [x.stats() for x in syss]
            gs = grp.stats() if grp else None
            print(json.dumps({"streams": S, "mode": mode, "fps": round(S * (total - 32) / el, 1), "elapsed_ms": round(el * 1e3, 1), "group": gs,
                              "kf": [x["keyframes"] for x in st], "ba_runs": [x["ba_runs"] for x in st], "lost": [x["lost"] for x in st],
                              "ms_track": [round(x["ms_track"]) for x in st], "ms_keyframe": [round(x["ms_keyframe"]) for x in st], "ms_backend": [round(x["ms_backend"]) for x in st],
                              "ms_extract": [round(x["ms_extract"]) for x in st]}), flush=True)
            if args.profile:
                names = (C.c_char * 48 * 96)(); ms = np.zeros(96); calls = np.zeros(96, dtype=np.int64); nn = C.c_int()
                L.check(L.lib.vo_profile_read(h0, C.cast(names, C.c_void_p), ms.ctypes.data, calls.ctypes.data, 96, C.byref(nn)))
                L.check(L.lib.vo_profile_enable(h0, 0))
                rows = sorted(((names[j].value.decode(), float(ms[j]), int(calls[j])) for j in range(nn.value)), key=lambda r: -r[1])
                for n_, t_, c_ in rows:
                    print("    %-20s total %8.2f ms  launches %6d  avg %8.2f us" % (n_, t_, c_, 1e3 * t_ / max(1, c_)), flush=True)
            for x in syss: x.close()
            if grp: grp.close()


if __name__ == "__main__":
    main()
