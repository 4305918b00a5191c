"""Independent numpy re-derivations of the hot path's numerics, used to pin BOTH implementations of the
C-ABI (the CPU oracle in CPU CI, the HIP kernels under -m gpu) to the published definitions instead of to
each other.  Nothing here imports or mirrors oracle/ code: every check restates the textbook / OpenCV-3.1 /
g2o definition in a few lines of numpy and compares through the C-ABI (include/vo_hip.h).

  gray / pyramid      Rec.601 luma and bilinear resize as float formulas, +-1 LSB          (cv::cvtColor, cv::resize)
  FAST-9/16           score from the definition (9 contiguous ring pixels), strict 3x3 NMS (cv::FAST)
  Harris              (ab - c^2) - 0.04 (a + b)^2 over a 7x7 block of Sobel products        (cv::ORB HarrisResponses)
  retain-best         2 x quota by FAST score, then quota by Harris                        (cv::ORB)
  orientation         intensity centroid over the radius-15 disc
  blur                7x7 sigma-2 Gaussian as float convolution, +-1 LSB                   (cv::GaussianBlur)
  rBRIEF              steered tests on the blurred level, bit i -> byte i / 8, bit i % 8
  depth               src/frame.cpp:43-67
  match               frustum + view-angle filter src/frame.cpp:70-91, brute-force Hamming, gate src/frontend.cpp:190-211
  pose LM             first step from finite-difference Jacobians; converged minimum vs scipy; g2o_types.h:47-108
  local BA            first step of the dense (un-reduced) normal equations from finite-difference Jacobians;
                      stationarity of the final state; g2o_types.h:111-179, src/backend.cpp:140-172
"""
import os
import re

import numpy as np
import pytest

import ref_model as rm
from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBS = [pytest.param(ORACLE_LIB, id="cpu-oracle"), pytest.param(capi.HIP_LIB, id="hip", marks=pytest.mark.gpu)]
FX, FY, CX, CY = np.float32(517.3), np.float32(516.5), np.float32(318.6), np.float32(255.3)
K4 = np.array([FX, FY, CX, CY], dtype=np.float64)        # float intrinsics promoted to double (src/camera.cpp:29-32)
DELTA = np.sqrt(7.815)


@pytest.fixture(scope="module")
def frame():
    syn = capi.Synth()
    bgr, depth, Twc, _ = syn.render(syn.params(seed=9), 0, 3, threads=8)
    return bgr, depth, Twc


@pytest.fixture(scope="module")
def pattern():
    txt = open(os.path.join(ROOT, "include", "vo_brief_pattern.h")).read()
    body = txt[txt.index("VO_BRIEF_PATTERN_INIT {"):]
    nums = [int(v) for v in re.findall(r"-?\d+", body[:body.index("}\nstatic")])]
    assert len(nums) == 1024
    return np.array(nums).reshape(256, 4)


# --------------------------------------------------------------------------------------------------------------------
# ORB
# --------------------------------------------------------------------------------------------------------------------
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_score_map(img):
    """Corner score by definition: the largest t such that 9 contiguous ring pixels are all > p + t or all < p - t."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - c for dx, dy in RING])      # 16 x (h-6) x (w-6)
    best = np.full(c.shape, -10 ** 6, np.int32)
    for s in range(16):
        arc = d[[(s + k) % 16 for k in range(9)]]
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    out = np.zeros((h, w), np.int32)
    out[3:h - 3, 3:w - 3] = best - 1            # all 9 differ by more than t  <=>  t <= min - 1
    return out


def harris_float(img, x, y):
    I = img.astype(np.int64)
    a = b = c = 0
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            yy, xx = y + dy, x + dx
            ix = (I[yy, xx + 1] - I[yy, xx - 1]) * 2 + (I[yy - 1, xx + 1] - I[yy - 1, xx - 1]) + (I[yy + 1, xx + 1] - I[yy + 1, xx - 1])
            iy = (I[yy + 1, xx] - I[yy - 1, xx]) * 2 + (I[yy + 1, xx - 1] - I[yy - 1, xx - 1]) + (I[yy + 1, xx + 1] - I[yy - 1, xx + 1])
            a += ix * ix; b += iy * iy; c += ix * iy
    scale = 1.0 / (4 * 7 * 255.0)
    return (float(a) * float(b) - float(c) * float(c) - 0.04 * float(a + b) ** 2) * scale ** 4


def umax_table():
    hp = 15
    um = np.zeros(hp + 2, int)
    vmax, vmin = int(np.floor(hp * np.sqrt(2) / 2 + 1)), int(np.ceil(hp * np.sqrt(2) / 2))
    for v in range(vmax + 1):
        um[v] = int(np.rint(np.sqrt(hp * hp - v * v)))
    v0 = 0
    for v in range(hp, vmin - 1, -1):
        while um[v0] == um[v0 + 1]:
            v0 += 1
        um[v] = v0
        v0 += 1
    return um


@pytest.mark.parametrize("lib", LIBS)
def test_pyramid_matches_float_formulas(lib, frame):
    bgr, depth, _ = frame
    L = capi.load(lib)
    ctx = L.context(L.default_params(n_features=1000))
    ctx.upload(0, bgr[0], depth[0]); ctx.orb(0, 1)
    g0 = ctx.fetch_level(0, 0).astype(np.float64)
    want = 0.114 * bgr[0][..., 0] + 0.587 * bgr[0][..., 1] + 0.299 * bgr[0][..., 2]
    assert np.abs(g0 - want).max() <= 0.5 + 2e-3            # fixed-point luma = round(float luma) up to the 14-bit coefficients
    prev = g0
    for l in range(1, 8):
        w, h, _ = ctx.level_size(l)
        got = ctx.fetch_level(0, l).astype(np.float64)
        sh, sw = prev.shape
        fx = (np.arange(w) + 0.5) * (sw / w) - 0.5
        fy = (np.arange(h) + 0.5) * (sh / h) - 0.5
        x0 = np.floor(fx).astype(int); ax = fx - x0
        y0 = np.floor(fy).astype(int); ay = fy - y0
        ax[x0 < 0] = 0; x0c = np.clip(x0, 0, sw - 1); x1c = np.clip(x0 + 1, 0, sw - 1); ax[x0 >= sw - 1] = 0
        y0c = np.clip(y0, 0, sh - 1); y1c = np.clip(y0 + 1, 0, sh - 1)
        top = prev[y0c][:, x0c] * (1 - ax) + prev[y0c][:, x1c] * ax
        bot = prev[y1c][:, x0c] * (1 - ax) + prev[y1c][:, x1c] * ax
        want = top * (1 - ay)[:, None] + bot * ay[:, None]
        err = np.abs(got - want)
        assert err.max() <= 1.0 and err.mean() < 0.35, "level %d: bilinear resize off by more than the fixed-point truncation (max %.2f mean %.2f)" % (l, err.max(), err.mean())
        prev = got
    ctx.close()


@pytest.mark.parametrize("lib", LIBS)
def test_blur_matches_float_gaussian(lib, frame):
    bgr, depth, _ = frame
    L = capi.load(lib)
    ctx = L.context(L.default_params(n_features=500))
    ctx.upload(0, bgr[1], depth[1]); ctx.orb(0, 1)
    k = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 4.0); k /= k.sum()
    # OpenCV 3.1 filters 8-bit images with the taps rounded to 8 fractional bits (18 34 49 55 49 34 18: they add up to 257,
    # a gain of (257/256)^2 over both passes), accumulates exactly and rounds once at the end
    kq = np.rint(k * 256) / 256
    assert list(np.rint(k * 256).astype(int)) == [18, 34, 49, 55, 49, 34, 18]
    for l in (0, 3, 7):
        img = ctx.fetch_level(0, l).astype(np.float64)
        p = np.pad(img, 3, mode="reflect")                  # BORDER_REFLECT_101
        got = ctx.fetch_blur_level(0, l).astype(np.float64)
        for taps, tol in ((kq, 0.5 + 1e-9), (k, 0.5 + 255 * ((257 / 256) ** 2 - 1) + 0.2)):
            hz = sum(taps[i] * p[:, i:i + img.shape[1]] for i in range(7))
            want = np.minimum(255.0, sum(taps[i] * hz[i:i + img.shape[0], :] for i in range(7)))
            assert np.abs(got - want).max() <= tol, "level %d" % l
    ctx.close()


@pytest.mark.parametrize("lib", LIBS)
def test_detection_selection_orientation_descriptor_from_definitions(lib, frame, pattern):
    bgr, depth, _ = frame
    L = capi.load(lib)
    N = 1000
    ctx = L.context(L.default_params(n_features=N))
    ctx.upload(0, bgr[0], depth[0]); ctx.orb(0, 1)
    kps, desc = ctx.orb_fetch(0)
    assert len(kps) == N
    um = umax_table()
    n_desc_checked = 0
    for l in range(8):
        w, h, quota = ctx.level_size(l)
        img = ctx.fetch_level(0, l)
        blur = ctx.fetch_blur_level(0, l)
        sc = fast_score_map(img)
        corner = np.where(sc >= 20, np.minimum(sc, 255), 0)
        # strict 3x3 non-max suppression on the score map (non-corners count as 0)
        p = np.pad(corner, 1)
        nb = np.stack([p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)])
        keep = (corner > 0) & (corner > nb.max(0))
        keep[:31, :] = False; keep[h - 31:, :] = False; keep[:, :31] = False; keep[:, w - 31:] = False      # edgeThreshold 31
        ys, xs = np.nonzero(keep)
        scores = corner[ys, xs]
        # retainBest(2 * quota) by FAST score: ties at the cut are kept
        if len(ys) > 2 * quota:
            cut = np.sort(scores)[::-1][2 * quota - 1]
            sel = scores >= cut
            ys, xs, scores = ys[sel], xs[sel], scores[sel]
        R = np.array([harris_float(img, int(x), int(y)) for x, y in zip(xs, ys)])
        got = kps[kps["octave"] == l]
        assert len(got) == min(quota, len(R))
        s = np.float32(np.float64(np.float32(1.2)) ** l)
        gx = np.rint(got["x"] / s).astype(int); gy = np.rint(got["y"] / s).astype(int)
        assert np.allclose(got["x"], gx.astype(np.float32) * s) and np.allclose(got["size"], 31 * s)
        cand = {(int(x), int(y)): r for x, y, r in zip(xs, ys, R)}
        assert all((int(x), int(y)) in cand for x, y in zip(gx, gy)), "level %d: a keypoint is not a FAST/NMS survivor" % l
        Rg = np.array([cand[(int(x), int(y))] for x, y in zip(gx, gy)])
        assert np.allclose(got["response"], Rg, rtol=2e-5, atol=1e-12), "level %d: Harris response" % l
        # retainBest(quota) by Harris: everything strictly above the weakest selected response is selected
        order = np.sort(R)[::-1]
        if len(R) > quota:
            thr = order[quota - 1]
            tol = 1e-9 * max(1.0, abs(thr))
            assert Rg.min() >= thr - tol
            must = {(int(x), int(y)) for x, y, r in zip(xs, ys, R) if r > thr + tol}
            assert must <= {(int(x), int(y)) for x, y in zip(gx, gy)}
        assert np.all(np.diff(got["response"]) <= 0)
        # orientation + descriptor for a sample of this level's keypoints
        idx = np.nonzero(kps["octave"] == l)[0]
        I = img.astype(np.int64)
        for j in idx[:: max(1, len(idx) // 12)]:
            x, y = int(np.rint(kps["x"][j] / s)), int(np.rint(kps["y"][j] / s))
            m10 = m01 = 0
            for v in range(-15, 16):
                u = np.arange(-um[abs(v)], um[abs(v)] + 1)
                row = I[y + v, x + u]
                m10 += int((u * row).sum()); m01 += int(v * row.sum())
            ang = np.degrees(np.arctan2(m01, m10)) % 360.0
            da = abs(kps["angle"][j] - ang)
            assert min(da, 360 - da) < 0.35                 # fastAtan2 is a 0.3-degree polynomial
            n = np.hypot(m10, m01)
            cs, sn = (m10 / n, m01 / n) if n > 0 else (1.0, 0.0)
            B = blur.astype(np.int32)
            x1 = np.rint(pattern[:, 0] * cs - pattern[:, 1] * sn).astype(int); y1 = np.rint(pattern[:, 0] * sn + pattern[:, 1] * cs).astype(int)
            x2 = np.rint(pattern[:, 2] * cs - pattern[:, 3] * sn).astype(int); y2 = np.rint(pattern[:, 2] * sn + pattern[:, 3] * cs).astype(int)
            bits = (B[y + y1, x + x1] < B[y + y2, x + x2]).astype(np.uint8)
            want = np.packbits(bits, bitorder="little")
            assert np.array_equal(desc[j], want), "descriptor of keypoint %d (level %d)" % (j, l)
            n_desc_checked += 1
            # Frame::GetDepth (src/frame.cpp:43-67)
            px, py = int(np.rint(kps["x"][j])), int(np.rint(kps["y"][j]))
            d = int(depth[0][py, px])
            if d == 0:
                for ddx, ddy in ((-1, 0), (0, -1), (1, 0), (0, 1)):
                    if 0 <= px + ddx < 640 and 0 <= py + ddy < 480 and depth[0][py + ddy, px + ddx] != 0:
                        d = int(depth[0][py + ddy, px + ddx]); break
            assert kps["depth_raw"][j] == d
    assert n_desc_checked >= 80
    ctx.close()


# --------------------------------------------------------------------------------------------------------------------
# matching
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lib", LIBS)
def test_candidate_filter_and_matching_from_definitions(lib, frame):
    bgr, depth, Twc = frame
    L = capi.load(lib)
    p = L.default_params(n_features=800, map_capacity=4096)
    ctx = L.context(p)
    ctx.upload(0, bgr[0], depth[0]); ctx.orb(0, 1)
    kps, desc = ctx.orb_fetch(0)
    rng = np.random.default_rng(1)
    # a map from this frame's keypoints (identity pose), descriptors partly corrupted, some points behind / outside / oblique
    ok = kps["depth_raw"] > 0
    z = kps["depth_raw"][ok] / 5000.0
    X = np.stack([(kps["x"][ok] - K4[2]) * z / K4[0], (kps["y"][ok] - K4[3]) * z / K4[1], z], 1)
    D = desc[ok].copy()
    n = len(X)
    flip = rng.integers(0, 256, size=(n, 12))
    for i in range(n):
        k = rng.integers(0, 13)
        for b in flip[i, :k]:
            D[i, b // 8] ^= 1 << (b % 8)
    nrm = X / np.linalg.norm(X, axis=1, keepdims=True)
    tilt = rng.random(n) < 0.2
    nrm[tilt] = rm.so3_log(np.eye(3))[None] + np.array([0.9, 0.0, 0.436])     # 64 degrees off the viewing ray for most points
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    flags = (rng.random(n) < 0.05).astype(np.uint8)
    X[rng.random(n) < 0.05] *= -1.0                          # behind the camera
    idx = np.arange(n, dtype=np.int32)
    ctx.map_upsert(idx, X, nrm, D, flags)
    order = rng.permutation(n).astype(np.int32)
    ctx.map_set_active(order)
    T = rm.se3_exp(np.array([0.03, -0.02, 0.01, 0.01, -0.02, 0.015]))
    got, n_cand, min_dist = ctx.match(0, T)
    # definition: src/frame.cpp:70-91
    R, t = T[:9].reshape(3, 3), T[9:]
    C = -R.T @ t
    cand = []
    for q in order:
        if flags[q] & 1:
            continue
        pc = R @ X[q] + t
        if not pc[2] > 0:
            continue
        u, v = K4[0] * pc[0] / pc[2] + K4[2], K4[1] * pc[1] / pc[2] + K4[3]
        if u < 0 or u >= 640 or v < 0 or v >= 480:
            continue
        d = (X[q] - C) / np.linalg.norm(X[q] - C)
        ang = np.arccos(np.clip(d @ nrm[q], -1, 1))
        if ang > np.pi / 6:
            continue
        cand.append(int(q))
    assert n_cand == len(cand) and 50 < len(cand) < n
    bits_f = np.unpackbits(desc, axis=1)
    want = []
    for q in cand:
        dist = (np.unpackbits(D[q])[None, :] != bits_f).sum(1)
        kp = int(np.argmin(dist))                            # first minimum wins
        want.append((q, kp, int(dist[kp])))
    mn = min(w[2] for w in want)
    assert min_dist == mn
    gate = max(np.float32(mn) * np.float32(2.0), np.float32(30.0))
    want = [w for w in want if np.float32(w[2]) <= gate]     # src/frontend.cpp:196,:206
    assert [(int(m["map_index"]), int(m["kp_index"]), int(m["distance"])) for m in got] == want
    ctx.close()


# --------------------------------------------------------------------------------------------------------------------
# pose-only LM and local BA: g2o semantics re-derived with finite differences
# --------------------------------------------------------------------------------------------------------------------
def proj(T, X):
    pc = T[:9].reshape(3, 3) @ X + T[9:]
    return np.array([K4[0] * pc[0] / pc[2] + K4[2], K4[1] * pc[1] / pc[2] + K4[3]])


def huber_rho(e2, delta):
    return e2 if e2 <= delta * delta else 2 * np.sqrt(e2) * delta - delta * delta


def pose_cost(T, X, uv, robust, active=None):
    c = 0.0
    for i in range(len(X)):
        if active is not None and not active[i]:
            continue
        e = uv[i] - proj(T, X[i])
        c += huber_rho(e @ e, DELTA) if robust else e @ e
    return c


def num_jac_pose(T, X, eps=1e-6):
    """d(error)/d(xi) for error = uv - proj(exp(xi) T X), xi = [translation, rotation] (g2o_types.h:56-60,:83)."""
    J = np.zeros((2, 6))
    for a in range(6):
        d = np.zeros(6); d[a] = eps
        J[:, a] = -(proj(rm.compose(rm.se3_exp(d), T), X) - proj(rm.compose(rm.se3_exp(-d), T), X)) / (2 * eps)
    return J


def make_pnp(rng, n, noise=0.4, outliers=0):
    T = rm.se3_exp(np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.1, 3)]))
    X = np.zeros((n, 3), np.float32); uv = np.zeros((n, 2), np.float32)
    R, t = T[:9].reshape(3, 3), T[9:]
    for i in range(n):
        pc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(2.0, 6.0)])
        X[i] = R.T @ (pc - t)
        uv[i] = proj(T, X[i].astype(np.float64)) + rng.normal(0, noise, 2)
    for i in rng.choice(n, outliers, replace=False):
        uv[i] += rng.choice([-1, 1], 2) * rng.uniform(6, 25, 2)
    return T, X, uv


@pytest.mark.parametrize("lib", LIBS)
def test_pose_lm_first_step_from_finite_difference_jacobians(lib):
    """One LM iteration = solve (H + lambda I) dx = b with H = sum w J^T J, b = -sum w J^T e, lambda0 = 1e-5 max diag H,
    update T <- exp(dx) T.  J by central differences here; analytic (g2o_types.h:86-100) in the implementations."""
    L = capi.load(lib)
    rng = np.random.default_rng(21)
    for trial, (n, outl) in enumerate([(40, 0), (200, 30), (6, 0)]):
        Ttrue, X, uv = make_pnp(rng, n, outliers=outl)
        T0 = rm.compose(rm.se3_exp(np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)])), Ttrue)
        ctx = L.context(L.default_params(n_features=64, map_capacity=max(64, n)))
        ctx.matches_set(X, uv)
        # a RANSAC hypothesis near the truth makes every non-outlier an inlier: the inlier list defines the LM's edges
        _, inl, _, _, _ = ctx.pnp_ransac(T0, n_hyp=64, seed=5 + trial)
        assert len(inl) >= n - outl - 3
        Xd, uvd = X[inl].astype(np.float64), uv[inl].astype(np.float64)
        H = np.zeros((6, 6)); b = np.zeros(6)
        for i in range(len(inl)):
            e = uvd[i] - proj(T0, Xd[i])
            J = num_jac_pose(T0, Xd[i])
            e2 = e @ e
            w = 1.0 if e2 <= DELTA ** 2 else DELTA / np.sqrt(e2)      # Huber: rho'(e2)
            H += w * J.T @ J; b -= w * J.T @ e
        lam = 1e-5 * np.abs(np.diag(H)).max()
        dx = np.linalg.solve(H + lam * np.eye(6), b)
        want = rm.compose(rm.se3_exp(dx), T0)
        assert pose_cost(want, Xd, uvd, True) < pose_cost(T0, Xd, uvd, True)
        got, _, iters = ctx.pose_lm(T0, it_robust=1, it_plain=0)
        assert iters == 1
        assert np.abs(got - want).max() < 2e-7 * (1 + np.abs(dx).max() / 1e-2), "LM step differs from the finite-difference normal equations"
        ctx.close()


@pytest.mark.parametrize("lib", LIBS)
def test_pose_lm_converges_to_the_minimum_scipy_finds(lib):
    """10 robust + 10 plain iterations (src/frontend.cpp:291-329): round 1 minimises the Huber cost, edges with chi2 > 1 leave,
    round 2 minimises the plain cost of the rest; the final mask marks chi2 <= 1 under the final pose."""
    from scipy.optimize import minimize
    L = capi.load(lib)
    rng = np.random.default_rng(33)
    Ttrue, X, uv = make_pnp(rng, 300, noise=0.55, outliers=40)
    T0 = rm.compose(rm.se3_exp(np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.005, 3)])), Ttrue)
    ctx = L.context(L.default_params(n_features=64, map_capacity=512))
    ctx.matches_set(X, uv)
    _, inl, _, _, _ = ctx.pnp_ransac(T0, n_hyp=32, seed=9)
    Xd, uvd = X[inl].astype(np.float64), uv[inl].astype(np.float64)
    got, mask, iters = ctx.pose_lm(T0)
    assert 2 <= iters <= 20

    def minimise(Tstart, robust, active):
        f = lambda d: pose_cost(rm.compose(rm.se3_exp(d), Tstart), Xd, uvd, robust, active)
        r = minimize(f, np.zeros(6), method="BFGS", options={"gtol": 1e-10})
        r = minimize(f, r.x, method="Nelder-Mead", options={"xatol": 1e-12, "fatol": 1e-14, "maxiter": 4000})
        return rm.compose(rm.se3_exp(r.x), Tstart)

    T1 = minimise(T0, True, None)
    chi = np.array([np.sum((uvd[i] - proj(T1, Xd[i])) ** 2) for i in range(len(Xd))])
    act = ~(chi > 1.0)
    T2 = minimise(T1, False, act)
    assert np.abs(got - T2).max() < 5e-6
    chi2 = np.array([np.sum((uvd[i] - proj(got, Xd[i])) ** 2) for i in range(len(Xd))])
    safe = np.abs(chi2 - 1.0) > 1e-4
    assert np.array_equal(mask[safe].astype(bool), chi2[safe] <= 1.0)
    assert 150 < mask.sum() < len(mask) - 20
    ctx.close()


def ba_unpack(poses, points, n_free, dx):
    P = poses.copy(); Xn = points.copy()
    for j in range(n_free):
        P[j] = rm.compose(rm.se3_exp(dx[6 * j:6 * j + 6]), poses[j])
    Xn += dx[6 * n_free:].reshape(-1, 3)
    return P, Xn


def ba_residuals(poses, points, ep, el, uv):
    return np.array([uv[e] - proj(poses[ep[e]], points[el[e]]) for e in range(len(ep))])


def make_ba(rng, n_poses, n_free, n_points, noise=0.3):
    Ts = [rm.se3_exp(np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.08, 3)])) for _ in range(n_poses)]
    X = rng.uniform(-1.5, 1.5, (n_points, 3)) + [0, 0, 5]
    ep, el, uv = [], [], []
    for k in range(n_points):                               # edges sorted by point, as the graph cut emits them
        for j in range(n_poses):
            if rng.random() < 0.8:
                ep.append(j); el.append(k); uv.append(proj(Ts[j], X[k]) + rng.normal(0, noise, 2))
    poses0 = np.array([rm.compose(rm.se3_exp(np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.004, 3)])), T) if j < n_free else T
                       for j, T in enumerate(Ts)])
    pts0 = X + rng.normal(0, 0.02, X.shape)
    return poses0, pts0, np.array(ep, np.int32), np.array(el, np.int32), np.array(uv, np.float32)


@pytest.mark.parametrize("lib", LIBS)
def test_local_ba_first_step_equals_dense_normal_equations(lib):
    """g2o's BlockSolver solves the Schur-reduced system; the step is the solution of the full (H + lambda I) [dp; dx] = b.
    Here the full system is assembled densely from finite-difference Jacobians (pose: exp(d) T, g2o_types.h:119-124;
    point: additive, :138-143) and solved with numpy."""
    L = capi.load(lib)
    rng = np.random.default_rng(40)
    for n_poses, n_free, n_points in [(4, 2, 30), (3, 3, 20)]:
        poses, pts, ep, el, uv = make_ba(rng, n_poses, n_free, n_points)
        uvd = uv.astype(np.float64)
        nv = 6 * n_free + 3 * n_points
        H = np.zeros((nv, nv)); b = np.zeros(nv)
        eps = 1e-6
        for e in range(len(ep)):
            j, k = ep[e], el[e]
            r = uvd[e] - proj(poses[j], pts[k])
            e2 = r @ r
            w = 1.0 if e2 <= DELTA ** 2 else DELTA / np.sqrt(e2)
            cols, J = [], []
            if j < n_free:
                Jp = np.zeros((2, 6))
                for a in range(6):
                    d = np.zeros(6); d[a] = eps
                    Jp[:, a] = -(proj(rm.compose(rm.se3_exp(d), poses[j]), pts[k]) - proj(rm.compose(rm.se3_exp(-d), poses[j]), pts[k])) / (2 * eps)
                cols += list(range(6 * j, 6 * j + 6)); J.append(Jp)
            Jx = np.zeros((2, 3))
            for a in range(3):
                d = np.zeros(3); d[a] = eps
                Jx[:, a] = -(proj(poses[j], pts[k] + d) - proj(poses[j], pts[k] - d)) / (2 * eps)
            cols += list(range(6 * n_free + 3 * k, 6 * n_free + 3 * k + 3)); J.append(Jx)
            Jf = np.hstack(J)
            H[np.ix_(cols, cols)] += w * Jf.T @ Jf
            b[cols] -= w * Jf.T @ r
        lam = 1e-5 * np.abs(np.diag(H)).max()
        dx = np.linalg.solve(H + lam * np.eye(nv), b)
        wantP, wantX = ba_unpack(poses, pts, n_free, dx)
        ctx = L.context(L.default_params(n_features=64, map_capacity=64))
        po, pt, fl, res = ctx.local_ba(poses, n_free, pts, ep, el, uv, it_robust=1, it_plain=0, chi2_th=1e9)
        assert res.lm_iters == 1
        scale = 1 + np.abs(dx).max() / 1e-2
        assert np.abs(po - wantP[:n_free]).max() < 1e-6 * scale and np.abs(pt - wantX).max() < 1e-6 * scale
        ctx.close()


@pytest.mark.parametrize("lib", LIBS)
def test_local_ba_reaches_a_stationary_point_and_culls_by_chi2(lib):
    """After 10 + 10 iterations (src/backend.cpp:140-172) the gradient of the plain cost over the surviving edges vanishes
    (finite differences), culled edges are exactly those with chi2 > th after round 1 / round 2."""
    L = capi.load(lib)
    rng = np.random.default_rng(44)
    poses, pts, ep, el, uv = make_ba(rng, 4, 3, 30, noise=0.25)
    uv[5] += (30, -20); uv[17] += (-15, 25)                 # two gross outliers
    ctx = L.context(L.default_params(n_features=64, map_capacity=64))
    po, pt, fl, res = ctx.local_ba(poses, 3, pts, ep, el, uv, chi2_th=4.0)
    P = poses.copy(); P[:3] = po
    uvd = uv.astype(np.float64)
    r = ba_residuals(P, pt, ep, el, uvd)
    chi = (r * r).sum(1)
    assert fl[5] & 3 and fl[17] & 3
    keep = (fl & 3) == 0
    safe = np.abs(chi - 4.0) > 1e-3
    assert np.all(chi[keep & safe] <= 4.0)
    assert res.chi2_final < res.chi2_initial and abs(res.chi2_final - chi[keep].sum()) < 1e-6 * max(1.0, chi[keep].sum())
    # stationarity over the edges that took part in round 2 (bit0 = culled after round 1)
    act = (fl & 1) == 0

    def cost(dx):
        Pn, Xn = ba_unpack(P, pt, 3, dx)
        rr = ba_residuals(Pn, Xn, ep, el, uvd)
        return ((rr * rr).sum(1) * act).sum()

    nv = 18 + 3 * len(pt)
    g = np.zeros(nv)
    for a in range(nv):
        d = np.zeros(nv); d[a] = 1e-6
        g[a] = (cost(d) - cost(-d)) / 2e-6
    c0 = cost(np.zeros(nv))
    # Gauss-Newton decrement of the remaining gradient is tiny compared with the cost
    assert np.abs(g).max() < 2e-2 * max(1.0, c0), "gradient %.3g at cost %.3g" % (np.abs(g).max(), c0)
    ctx.close()


@pytest.mark.parametrize("lib", LIBS)
def test_batched_triangulation_against_svd(lib):
    """vo_triangulate_batch (reference include/myslam/util.h:16-34 for many points): smallest right singular vector of the DLT
    matrix, success iff sigma4 / sigma3 < 1e-2 -- against numpy's SVD."""
    L = capi.load(lib)
    rng = np.random.default_rng(12)
    vs, Ts, xys, want, okw, ratios = [0], [], [], [], [], []
    for i in range(400):
        n = int(rng.integers(1, 7))                        # single-view points take part too: never a success
        X = rng.uniform(-1, 1, 3) + [0, 0, 4]
        noise = 0.0 if i % 4 == 0 else 10 ** rng.uniform(-5, -1.5)
        Tl, pl = [], []
        for _ in range(n):
            T = rm.se3_exp(np.concatenate([rng.normal(0, 0.4, 3), rng.normal(0, 0.15, 3)]))
            pc = T[:9].reshape(3, 3) @ X + T[9:]
            Tl.append(T); pl.append([pc[0] / pc[2] + rng.normal(0, noise), pc[1] / pc[2] + rng.normal(0, noise), 1.0])
        Ts += Tl; xys += [p[:2] for p in pl]; vs.append(vs[-1] + n)
        if n >= 2:
            x, ok, sv = rm.triangulate(Tl, pl)
            want.append(x); okw.append(ok); ratios.append(sv[3] / sv[2])
        else:
            want.append(np.zeros(3)); okw.append(False); ratios.append(1.0)
    ctx = L.context(L.default_params(n_features=64, map_capacity=64))
    xyz, ok = ctx.triangulate_batch(vs, np.array(Ts), np.array(xys))
    ctx.close()
    ratios = np.array(ratios); okw = np.array(okw)
    safe = np.abs(ratios - 1e-2) > 1e-6
    assert np.array_equal(ok[safe], okw[safe]) and 100 < ok.sum() < 390
    good = ratios < 0.5
    err = np.abs(xyz[good] - np.array(want)[good]).max(axis=1) * (1 - ratios[good])
    assert err.max() < 1e-6
