"""Local-BA micro-benchmark (developer tool): times vo_local_ba on synthetic covisibility-window problems of the bench workload's
shape (about 24 free + 26 fixed keyframes, 9000 points, 70-90 k observations) and checks the result against the CPU restatement.

    python scripts/bench_ba.py [--reps 20] [--oracle]

Prints one JSON line per problem shape: wall ms per BA, LM iterations, chi2, max |pose - oracle|."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def expso3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_problem(p, nP, nfree, nX, window, seed):
    """Poses along a path; point k is seen by a contiguous window of poses (a covisibility graph like the VO's)."""
    rng = np.random.default_rng(seed)
    poses = []
    for j in range(nP):
        R = expso3(rng.normal(size=3) * 0.05 + 1e-9)
        c = np.array([0.04 * j, 0.0, 0.0]) + rng.normal(size=3) * 0.02
        poses.append(np.concatenate([R.ravel(), -R @ c]))
    poses = np.array(poses)
    X = rng.uniform(-2, 2, size=(nX, 3)) + np.array([1.0, 0, 5.0])
    ep, el, uv = [], [], []
    for k in range(nX):
        a = rng.integers(0, nP)
        m = max(2, int(rng.integers(2, window)))
        for j in range(a, min(nP, a + m)):
            R, t = poses[j][:9].reshape(3, 3), poses[j][9:]
            pc = R @ X[k] + t
            o = rng.normal(size=2) * 0.3 + (rng.uniform(size=2) < 0.02) * 15.0
            ep.append(j); el.append(k); uv.append([p.fx * pc[0] / pc[2] + p.cx + o[0], p.fy * pc[1] / pc[2] + p.cy + o[1]])
    poses0 = poses.copy()
    for j in range(nfree):
        poses0[j][:9] = (expso3(rng.normal(size=3) * 0.004) @ poses[j][:9].reshape(3, 3)).ravel()
        poses0[j][9:] += rng.normal(size=3) * 0.01
    X0 = X + rng.normal(size=X.shape) * 0.03
    return poses0, X0, np.array(ep, np.int32), np.array(el, np.int32), np.array(uv, np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--shapes", default="bench")
    ap.add_argument("--its", default="10,10", help="LM iterations of the robust and the plain round")
    args = ap.parse_args()
    from rgbd_visualodometry_amd import capi
    H = capi.load(os.environ.get("VO_HIP_LIB", capi.HIP_LIB))          # e.g. csrc/build/libvo_hip_stamps.so
    p = H.default_params(map_capacity=1024)
    shapes = {"bench": [(50, 24, 9000, 16), (30, 17, 6000, 14), (60, 30, 9000, 20)], "small": [(18, 16, 300, 18), (6, 4, 400, 6)],
              "config5": [(26, 21, 9000, 26)]}[args.shapes]
    itr, itp = [int(x) for x in args.its.split(",")]
    O = None
    if args.oracle:
        from oracle import ORACLE_LIB
        O = capi.load(ORACLE_LIB)
    for (nP, nfree, nX, window) in shapes:
        prob = make_problem(p, nP, nfree, nX, window, 11)
        ctx = H.context(p)
        out = ctx.local_ba(prob[0], nfree, prob[1], prob[2], prob[3], prob[4], it_robust=itr, it_plain=itp)          # warm-up (allocations)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = ctx.local_ba(prob[0], nfree, prob[1], prob[2], prob[3], prob[4], it_robust=itr, it_plain=itp)
        ms = (time.perf_counter() - t0) * 1e3 / args.reps
        ph, xh, fh, rh = out
        rec = {"poses": nP, "free": nfree,
               "its": args.its, "points": nX, "edges": int(len(prob[2])), "ms_per_ba": round(ms, 3), "lm_iters": rh.lm_iters, "chi2_initial": rh.chi2_initial, "chi2_final": rh.chi2_final,
               "culled": int((fh != 0).sum())}
        ctx.close()
        if O is not None:
            octx = O.context(p)
            po, xo, fo, ro = octx.local_ba(prob[0], nfree, prob[1], prob[2], prob[3], prob[4], it_robust=itr, it_plain=itp)
            octx.close()
            rec.update({"oracle_lm_iters": ro.lm_iters, "oracle_chi2_final": ro.chi2_final, "max_pose_diff": float(np.abs(ph - po).max()),
                        "max_point_diff": float(np.abs(xh - xo).max()), "flags_equal": bool(np.array_equal(fh, fo))})
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
