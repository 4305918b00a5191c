#!/usr/bin/env python3
"""Generate golden vectors for the TUM evaluation tools (associate / ATE / RPE).

Run ONLY in the build container, where the reference checkout is mounted at
/root/reference:  python tests/golden/make_golden_eval.py

It imports the reference's own tools/associate.py, tools/evaluate_ate.py and
tools/evaluate_rpe.py, feeds them seeded inputs and stores inputs + outputs as
JSON under tests/golden/.  Only data is committed; the reference sources never
enter this repository.
"""
import json
import os
import sys
import warnings

import numpy as np

REF_TOOLS = "/root/reference/tools"
sys.path.insert(0, REF_TOOLS)
warnings.simplefilter("ignore")
import associate as ref_associate          # noqa: E402
import evaluate_ate as ref_ate             # noqa: E402
import evaluate_rpe as ref_rpe             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def smooth_traj(rng, n, t0=1305031100.0, dt=1.0 / 30.0, jitter=0.0):
    """TUM-format rows: stamp tx ty tz qx qy qz qw along a smooth random walk."""
    rows = []
    p = rng.normal(size=3) * 0.1
    v = rng.normal(size=3) * 0.05
    q = rand_quat(rng)
    for i in range(n):
        v += rng.normal(size=3) * 0.01
        p = p + v * dt
        dq = np.concatenate([rng.normal(size=3) * 0.01, [1.0]])
        # quaternion product (x,y,z,w)
        x1, y1, z1, w1 = q
        x2, y2, z2, w2 = dq
        q = np.array([
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
        q /= np.linalg.norm(q)
        stamp = t0 + i * dt + (rng.uniform(-jitter, jitter) if jitter else 0.0)
        rows.append([round(stamp, 6)] + [float(x) for x in p] + [float(x) for x in q])
    return rows


def perturb(rng, rows, sigma_t, sigma_r, stamp_shift):
    out = []
    for r in rows:
        p = np.array(r[1:4]) + rng.normal(size=3) * sigma_t
        q = np.array(r[4:8]) + rng.normal(size=4) * sigma_r
        q /= np.linalg.norm(q)
        out.append([round(r[0] + stamp_shift, 6)] + [float(x) for x in p] + [float(x) for x in q])
    return out


def make_associate(rng):
    cases = []
    for n1, n2, off, maxd in [(40, 37, 0.0, 0.02), (25, 60, 0.013, 0.02), (30, 30, -0.2, 0.05), (5, 5, 0.0, 0.0001)]:
        a = sorted(set(round(float(x), 6) for x in 100.0 + np.cumsum(rng.uniform(0.02, 0.045, size=n1))))
        b = sorted(set(round(float(x), 6) for x in 100.0 + np.cumsum(rng.uniform(0.02, 0.045, size=n2)) + rng.uniform(-0.01, 0.01)))
        first = {s: ["a%d" % i] for i, s in enumerate(a)}
        second = {s: ["b%d" % i] for i, s in enumerate(b)}
        m = ref_associate.associate(first, second, off, maxd)
        cases.append({"first": a, "second": b, "offset": off, "max_difference": maxd,
                      "matches": [[float(x), float(y)] for x, y in m]})
    return cases


def make_ate(rng):
    cases = []
    for n, sig in [(50, 0.01), (200, 0.05), (12, 0.0), (80, 0.3)]:
        model = rng.normal(size=(3, n)).cumsum(axis=1) * 0.05
        # rigid transform + noise
        q = rand_quat(rng)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        t = rng.normal(size=(3, 1))
        data = R @ model + t + rng.normal(size=(3, n)) * sig
        rot, trans, err = ref_ate.align(np.matrix(model), np.matrix(data))
        err = np.asarray(err)
        cases.append({
            "model": model.tolist(), "data": data.tolist(),
            "rot": np.asarray(rot).tolist(), "trans": np.asarray(trans).ravel().tolist(),
            "trans_error": err.tolist(),
            "rmse": float(np.sqrt(np.dot(err, err) / len(err))),
            "mean": float(np.mean(err)), "median": float(np.median(err)),
            "std": float(np.std(err)), "min": float(np.min(err)), "max": float(np.max(err))})
    return cases


def rows_to_traj(rows):
    return dict((r[0], ref_rpe.transform44(r)) for r in rows)


def make_rpe(rng):
    cases = []
    for n, unit, delta, sig_t, sig_r, shift in [(120, "s", 1.0, 0.01, 0.002, 0.003),
                                                  (90, "f", 5, 0.02, 0.004, 0.0),
                                                  (60, "s", 0.5, 0.0, 0.0, 0.0)]:
        gt = smooth_traj(rng, n, jitter=0.002)
        est = perturb(rng, gt[5:-3], sig_t, sig_r, shift)
        res = ref_rpe.evaluate_trajectory(rows_to_traj(gt), rows_to_traj(est), 10000, True, delta, unit, 0.0, 1.0)
        res = np.array(res)
        trans = res[:, 4]
        rot = res[:, 5]
        cases.append({
            "gt": gt, "est": est, "fixed_delta": True, "delta": delta, "delta_unit": unit,
            "offset": 0.0, "scale": 1.0, "max_pairs": 10000,
            "result": res.tolist(),
            "trans_rmse": float(np.sqrt(np.dot(trans, trans) / len(trans))),
            "trans_mean": float(np.mean(trans)), "trans_median": float(np.median(trans)),
            "trans_std": float(np.std(trans)), "trans_min": float(np.min(trans)), "trans_max": float(np.max(trans)),
            "rot_rmse_deg": float(np.sqrt(np.dot(rot, rot) / len(rot)) * 180.0 / np.pi),
            "rot_mean_deg": float(np.mean(rot) * 180.0 / np.pi)})
    # all-pairs (deterministic because len < sqrt(max_pairs))
    gt = smooth_traj(rng, 30)
    est = perturb(rng, gt, 0.01, 0.002, 0.0)
    res = ref_rpe.evaluate_trajectory(rows_to_traj(gt), rows_to_traj(est), 10000, False, 1.0, "s", 0.0, 1.0)
    res = np.array(res)
    cases.append({"gt": gt, "est": est, "fixed_delta": False, "delta": 1.0, "delta_unit": "s",
                  "offset": 0.0, "scale": 1.0, "max_pairs": 10000, "result": res.tolist(),
                  "trans_rmse": float(np.sqrt(np.dot(res[:, 4], res[:, 4]) / len(res)))})
    # transform44 spot vectors
    t44 = []
    for _ in range(6):
        row = [0.0] + list(rng.normal(size=3)) + list(rand_quat(rng) * rng.uniform(0.5, 2.0))
        t44.append({"row": [float(x) for x in row], "matrix": ref_rpe.transform44(row).tolist()})
    return cases, t44


def main():
    rng = np.random.default_rng(20261003)
    rpe_cases, t44 = make_rpe(rng)
    out = {
        "generator": "tests/golden/make_golden_eval.py (imports /root/reference/tools/{associate,evaluate_ate,evaluate_rpe}.py)",
        "associate": make_associate(rng),
        "ate": make_ate(rng),
        "rpe": rpe_cases,
        "transform44": t44,
    }
    path = os.path.join(HERE, "eval_tools_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
