"""Soak: one stream over N frames; every `every` frames prints frames/s of the last stretch and what the resident graph cut visited
(vo_ba_resident_window is per BA context: read through VO_TRACE's cut average instead) -- the cut's cost must not grow with the run."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6000)
    ap.add_argument("--every", type=int, default=1000)
    ap.add_argument("--map-capacity", type=int, default=1 << 22, help="initial device map slots (the map doubles when a keyframe may not fit)")
    args = ap.parse_args()
    import torch
    from rgbd_visualodometry_amd import capi, system
    W, H = 640, 480
    syn = capi.Synth()
    chunk = 1000                                            # rendered and uploaded in pieces: 1.5 MB per frame
    sp = syn.params(seed=0, speed=3.0)
    s = system.VoSystem(system.HOST_LIB, width=W, height=H, number_of_features=2000, max_frames_in_flight=32, enable_local_optimization=1, backend_lag_frames=8,
                        track_batch=8, map_capacity=args.map_capacity, ransac_iterations=100, ba_device_graph=1, map_descriptors_on_device=1, device_keyframes=0 if os.environ.get("VO_SOAK_HOST_KEYFRAMES") else 1)
    fb, fd = W * H * 3, W * H * 2
    done, t_last, n_last = 0, time.perf_counter(), 0
    while done < args.frames:
        n = min(chunk, args.frames - done)
        bgr, depth, Twc, stamps = syn.render(sp, done, n, threads=min(32, os.cpu_count() or 8))
        db = torch.from_numpy(bgr).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        i = 0
        while i < n:
            m = min(32, n - i)
            s.prefetch(stamps[i:i + m], [db.data_ptr() + (i + j) * fb for j in range(m)], [dd.data_ptr() + (i + j) * fd for j in range(m)], 3 * W, 2 * W, True)
            for j in range(m):
                s.add_prefetched()
            i += m
        s.flush(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done += n
        st = s.stats()
        print("frames %6d: %7.1f frames/s over the last %d, keyframes %d, lost %d, map points %s" % (done, n / dt, n, st.get("keyframes", -1), st.get("lost", -1), st.get("map_points", "?")), flush=True)
    s.close()


if __name__ == "__main__":
    main()
