"""TUM-benchmark trajectory metrics: timestamp association, ATE and RPE.

Behavioural restatement (own code, ndarray based) of the reference's offline
evaluation scripts so that the numbers quoted for this framework are computed
the way the reference computes its own:

* ``associate``            <- /root/reference/tools/associate.py:71-101
* ``horn_align`` / ``ate`` <- /root/reference/tools/evaluate_ate.py:47-79, :129-162
* ``pose_matrix``          <- /root/reference/tools/evaluate_rpe.py:46-74
* ``rpe``                  <- /root/reference/tools/evaluate_rpe.py:204-297

Pinned by tests/golden/eval_tools_golden.json (vectors captured from the
reference scripts themselves by tests/golden/make_golden_eval.py).
"""
from __future__ import annotations

import math
import random
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

_TINY = np.finfo(float).eps * 4.0


# --------------------------------------------------------------------------- #
# file formats
# --------------------------------------------------------------------------- #
def read_stamped_file(path: str) -> Dict[float, List[str]]:
    """``stamp d1 d2 ...`` per line, ``#`` comments, ',' and tabs are blanks."""
    out: Dict[float, List[str]] = {}
    with open(path) as fh:
        for raw in fh.read().replace(",", " ").replace("\t", " ").split("\n"):
            if not raw or raw[0] == "#":
                continue
            tok = [t for t in (s.strip() for s in raw.split(" ")) if t]
            if len(tok) > 1:
                out[float(tok[0])] = tok[1:]
    return out


def read_trajectory(path: str) -> Dict[float, np.ndarray]:
    """TUM trajectory file -> {stamp: 4x4}; all-zero quaternions and NaN rows skipped."""
    traj: Dict[float, np.ndarray] = {}
    for stamp, tok in read_stamped_file(path).items():
        vals = [float(t) for t in tok]
        if len(vals) < 7 or vals[3:7] == [0, 0, 0, 0] or any(math.isnan(v) for v in vals):
            continue
        traj[stamp] = pose_matrix([stamp] + vals[:7])
    return traj


def write_trajectory(path: str, rows: Iterable[Sequence[float]]) -> None:
    """Rows ``stamp tx ty tz qx qy qz qw`` with the reference's 2 header lines (run_vo.cpp:69-70)."""
    with open(path, "w") as fh:
        fh.write("# estimated trajectory format\n# timestamp tx ty tz qx qy qz qw\n")
        for r in rows:
            fh.write("%.6f %s\n" % (r[0], " ".join("%.9g" % v for v in r[1:8])))


# --------------------------------------------------------------------------- #
# association
# --------------------------------------------------------------------------- #
def associate(first: Iterable[float], second: Iterable[float], offset: float = 0.0,
              max_difference: float = 0.02) -> List[Tuple[float, float]]:
    """Greedy closest-stamp pairing: candidates |a-(b+offset)| < max_difference,
    taken in order of increasing (difference, a, b); each stamp used once."""
    free_a = set(first)
    free_b = set(second)
    cand = sorted((abs(a - (b + offset)), a, b) for a in free_a for b in free_b
                  if abs(a - (b + offset)) < max_difference)
    pairs = []
    for _, a, b in cand:
        if a in free_a and b in free_b:
            free_a.discard(a)
            free_b.discard(b)
            pairs.append((a, b))
    pairs.sort()
    return pairs


# --------------------------------------------------------------------------- #
# ATE
# --------------------------------------------------------------------------- #
def horn_align(model: np.ndarray, data: np.ndarray):
    """Rigid (no scale) least-squares alignment of ``model`` (3xn) onto ``data`` (3xn).
    Returns (R 3x3, t 3, per-point residual norm n)."""
    model = np.asarray(model, dtype=np.float64)
    data = np.asarray(data, dtype=np.float64)
    mc = model.mean(axis=1, keepdims=True)
    dc = data.mean(axis=1, keepdims=True)
    cov = (model - mc) @ (data - dc).T          # sum of outer(model_i, data_i)
    u, _, vh = np.linalg.svd(cov.T)
    s = np.eye(3)
    if np.linalg.det(u) * np.linalg.det(vh) < 0:
        s[2, 2] = -1.0
    rot = u @ s @ vh
    trans = dc - rot @ mc
    resid = rot @ model + trans - data
    return rot, trans.ravel(), np.sqrt((resid * resid).sum(axis=0))


def _stats(err: np.ndarray) -> Dict[str, float]:
    return {"rmse": float(math.sqrt(float(err @ err) / len(err))), "mean": float(err.mean()),
            "median": float(np.median(err)), "std": float(err.std()),
            "min": float(err.min()), "max": float(err.max()), "pairs": int(len(err))}


def ate(gt: Dict[float, Sequence[float]], est: Dict[float, Sequence[float]], offset: float = 0.0,
        max_difference: float = 0.02, scale: float = 1.0) -> Dict[str, float]:
    """Absolute trajectory error. ``gt``/``est`` map stamp -> (tx,ty,tz,...)."""
    pairs = associate(gt.keys(), est.keys(), offset, max_difference)
    if len(pairs) < 2:
        raise ValueError("fewer than two associated poses")
    g = np.array([[float(v) for v in gt[a][:3]] for a, _ in pairs]).T
    e = np.array([[float(v) * scale for v in est[b][:3]] for _, b in pairs]).T
    _, _, err = horn_align(e, g)
    return _stats(err)


# --------------------------------------------------------------------------- #
# RPE
# --------------------------------------------------------------------------- #
def pose_matrix(row: Sequence[float]) -> np.ndarray:
    """(stamp,tx,ty,tz,qx,qy,qz,qw) -> 4x4; quaternion normalised on the fly."""
    t = row[1:4]
    q = np.array(row[4:8], dtype=np.float64)
    n = float(q @ q)
    m = np.eye(4)
    m[:3, 3] = t
    if n < _TINY:
        return m
    q = q * math.sqrt(2.0 / n)
    o = np.outer(q, q)
    m[0, :3] = (1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3])
    m[1, :3] = (o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3])
    m[2, :3] = (o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1])
    return m


def _closest(sorted_vals: Sequence[float], t: float) -> int:
    """Bisection that remembers the closest probe (ties keep the earlier probe)."""
    lo, hi = 0, len(sorted_vals)
    best, bestd = 0, abs(sorted_vals[0] - t)
    while lo < hi:
        mid = (lo + hi) // 2
        d = abs(sorted_vals[mid] - t)
        if d < bestd:
            best, bestd = mid, d
        if sorted_vals[mid] == t:
            return mid
        if sorted_vals[mid] > t:
            hi = mid
        else:
            lo = mid + 1
    return best


def _rel(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return np.linalg.inv(a) @ b


def _angle(m: np.ndarray) -> float:
    return float(np.arccos(min(1.0, max(-1.0, (np.trace(m[:3, :3]) - 1.0) / 2.0))))


def _path_index(traj: Dict[float, np.ndarray], unit: str) -> List[float]:
    keys = sorted(traj)
    acc, out = 0.0, [0.0]
    for k0, k1 in zip(keys[:-1], keys[1:]):
        step = _rel(traj[k1], traj[k0])
        if unit == "m":
            acc += float(np.linalg.norm(step[:3, 3]))
        else:
            acc += _angle(step) * (180.0 / math.pi if unit == "deg" else 1.0)
        out.append(acc)
    return out


def rpe(traj_gt: Dict[float, np.ndarray], traj_est: Dict[float, np.ndarray], max_pairs: int = 10000,
        fixed_delta: bool = False, delta: float = 1.0, delta_unit: str = "s", offset: float = 0.0,
        scale: float = 1.0, rng: random.Random | None = None) -> List[List[float]]:
    """Relative pose error rows [t_est0, t_est1, t_gt0, t_gt1, trans_err, rot_err].

    Unlike the reference script, the distance/angle delta units work (the reference
    raises on dict_keys.sort(), evaluate_rpe.py:179-194)."""
    sg = sorted(traj_gt)
    se = sorted(traj_est)
    seen = []
    for t in se:
        tg = sg[_closest(sg, t + offset)]
        back = se[_closest(se, tg - offset)]
        if back not in seen:
            seen.append(back)
    if len(seen) < 2:
        raise ValueError("timestamp overlap too small")

    if delta_unit == "s":
        index = se
    elif delta_unit in ("m", "rad", "deg"):
        index = _path_index(traj_est, delta_unit)
    elif delta_unit == "f":
        index = list(range(len(se)))
    else:
        raise ValueError("unknown delta unit %r" % delta_unit)

    n = len(se)
    rng = rng or random
    if not fixed_delta:
        if max_pairs == 0 or n < math.sqrt(max_pairs):
            pairs = [(i, j) for i in range(n) for j in range(n)]
        else:
            pairs = [(rng.randint(0, n - 1), rng.randint(0, n - 1)) for _ in range(max_pairs)]
    else:
        pairs = []
        for i in range(n):
            j = _closest(index, index[i] + delta)
            if j != n - 1:
                pairs.append((i, j))
        if max_pairs != 0 and len(pairs) > max_pairs:
            pairs = rng.sample(pairs, max_pairs)

    tol = 2.0 * float(np.median(np.diff(sg)))
    rows = []
    for i, j in pairs:
        e0, e1 = se[i], se[j]
        g0 = sg[_closest(sg, e0 + offset)]
        g1 = sg[_closest(sg, e1 + offset)]
        if abs(g0 - (e0 + offset)) > tol or abs(g1 - (e1 + offset)) > tol:
            continue
        d_est = _rel(traj_est[e1], traj_est[e0]).copy()
        d_est[:3, 3] *= scale
        err = _rel(d_est, _rel(traj_gt[g1], traj_gt[g0]))
        rows.append([e0, e1, g0, g1, float(np.linalg.norm(err[:3, 3])), _angle(err)])
    if len(rows) < 2:
        raise ValueError("no comparable pose pairs")
    return rows


def rpe_summary(rows: List[List[float]]) -> Dict[str, float]:
    r = np.asarray(rows)
    t = _stats(r[:, 4])
    a = _stats(r[:, 5] * 180.0 / math.pi)
    return {"trans_" + k: v for k, v in t.items()} | {"rot_deg_" + k: v for k, v in a.items()}
