"""Local-BA engine batching experiment (developer tool): T host threads each solve the same bench-shaped problem repeatedly through
their own context; the engine steps the problems that are in flight together (blockIdx.z = problem).  Prints, per T, the wall time
per BA, problems per step launch and the average duration of the step kernels (HIP events around each launch, VO_BA_ENGINES=1).

    VO_BA_ENGINES=1 python scripts/bench_ba_batch.py --threads 1,2,4,8"""
import argparse, ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench_ba import make_problem


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    from rgbd_visualodometry_amd import capi
    H = capi.load(os.environ.get("VO_HIP_LIB", capi.HIP_LIB))
    p = H.default_params(map_capacity=1024)
    for T in [int(v) for v in args.threads.split(",")]:
        probs = [make_problem(p, 50, 24, 9000, 16, 11 + k) for k in range(T)]
        ctxs = [H.context(p) for _ in range(T)]
        for k in range(T):
            ctxs[k].local_ba(probs[k][0], 24, probs[k][1], probs[k][2], probs[k][3], probs[k][4])      # allocations
        for c in ctxs: H.check(H.lib.vo_profile_enable(c.h, 1))
        bar = threading.Barrier(T + 1)

        def run(k):
            bar.wait()
            for _ in range(args.reps):
                ctxs[k].local_ba(probs[k][0], 24, probs[k][1], probs[k][2], probs[k][3], probs[k][4])
        ths = [threading.Thread(target=run, args=(k,)) for k in range(T)]
        for t in ths: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in ths: t.join()
        el = time.perf_counter() - t0
        acc = {}
        for c in ctxs:                                       # a launch is booked on the context of the first active slot
            names = (C.c_char * 48 * 96)(); ms = np.zeros(96); calls = np.zeros(96, dtype=np.int64); nn = C.c_int()
            H.check(H.lib.vo_profile_read(c.h, C.cast(names, C.c_void_p), ms.ctypes.data, calls.ctypes.data, 96, C.byref(nn)))
            H.check(H.lib.vo_profile_enable(c.h, 0))
            for i in range(nn.value):
                n = names[i].value.decode()
                if n.startswith("k_ba"):
                    a = acc.setdefault(n, [0.0, 0]); a[0] += ms[i]; a[1] += int(calls[i])
        ker = {n: (round(1e3 * a[0] / max(1, a[1]), 1), a[1]) for n, a in acc.items()}
        print(json.dumps({"threads": T, "ms_per_ba_wall": round(1e3 * el / args.reps, 3), "ba_per_s": round(T * args.reps / el, 1), "kernels_avg_us_calls": ker}), flush=True)
        for c in ctxs: c.close()


if __name__ == "__main__":
    main()
