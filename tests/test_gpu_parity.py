"""GPU parity: every stage of the HIP path (through the C-ABI) against the CPU oracle on identical
seeded inputs.  Integer / index / byte results must be bit-exact; the LM / BA results (block
reductions, f64 atomics: different summation order) agree to the tolerance written in each test."""
import numpy as np
import pytest

from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi

pytestmark = pytest.mark.gpu

IDENT = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


def inv12(T):
    R = T[:9].reshape(3, 3)
    t = T[9:]
    return np.concatenate([R.T.ravel(), -R.T @ t])


@pytest.fixture(scope="module")
def frames():
    syn = capi.Synth()
    sp = syn.params(seed=11)
    bgr, depth, Twc, ts = syn.render(sp, 0, 10, threads=8)
    return bgr, depth, Twc, ts


@pytest.fixture(scope="module")
def libs():
    return capi.load(capi.HIP_LIB), capi.load(ORACLE_LIB)


def make_ctx(L, **kw):
    p = L.default_params(**kw)
    return L.context(p), p


def seed_map(ctx, p, kps, desc, T_wc):
    ok = kps["depth_raw"] > 0
    z = kps["depth_raw"][ok] / 5000.0
    pc = np.stack([(kps["x"][ok] - p.cx) * z / p.fx, (kps["y"][ok] - p.cy) * z / p.fy, z], 1)
    R, t = T_wc[:9].reshape(3, 3), T_wc[9:]
    pw = pc @ R.T + t
    nrm = pw - t
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    idx = np.arange(len(pw), dtype=np.int32)
    ctx.map_upsert(idx, pw, nrm, desc[ok], np.zeros(len(pw), np.uint8))
    ctx.map_set_active(idx)
    return len(pw)


@pytest.mark.parametrize("nfeat", [500, 2000])
def test_orb_bit_exact(frames, libs, nfeat):
    bgr, depth, _, _ = frames
    H, O = libs
    assert H.backend == "hip-gfx950" and O.backend == "cpu-oracle"
    ch, _ = make_ctx(H, n_features=nfeat, max_frames=3)
    co, _ = make_ctx(O, n_features=nfeat, max_frames=3)
    for s, f in enumerate((0, 4, 9)):
        ch.upload(s, bgr[f], depth[f])
        co.upload(s, bgr[f], depth[f])
    ch.orb(0, 3)            # one batched launch chain for the three slots
    co.orb(0, 3)
    for s in range(3):
        for l in range(8):
            assert np.array_equal(ch.fetch_level(s, l), co.fetch_level(s, l)), "pyramid level %d of slot %d differs" % (l, s)
        kh, dh = ch.orb_fetch(s)
        ko, do = co.orb_fetch(s)
        assert len(kh) == len(ko) == nfeat
        for field in ("x", "y", "size", "octave", "class_id", "depth_raw"):
            assert np.array_equal(kh[field], ko[field]), "keypoint field %s differs (slot %d)" % (field, s)
        assert np.array_equal(dh, do), "descriptors differ (slot %d): %d rows" % (s, int((dh != do).any(axis=1).sum()))
        np.testing.assert_allclose(kh["angle"], ko["angle"], atol=2e-3)         # float polynomial, degrees
        np.testing.assert_allclose(kh["response"], ko["response"], rtol=1e-5)


@pytest.mark.parametrize("w,h,nfeat", [(517, 391, 600), (322, 250, 300), (639, 477, 1500)])
def test_orb_bit_exact_at_odd_sizes(frames, libs, w, h, nfeat):
    """Image sizes that are multiples of nothing: partial tiles of every ORB kernel (FAST 64 x 32, blur 128 x 16, the pyramid's 5 x 8 split or its
    level-by-level fallback), level widths that end inside a dword, keypoints next to the 31-pixel border."""
    bgr, depth, _, _ = frames
    H, O = libs
    kw = dict(width=w, height=h, n_features=nfeat, max_frames=2, cx=w / 2.0, cy=h / 2.0)
    ch, _ = make_ctx(H, **kw)
    co, _ = make_ctx(O, **kw)
    for s, f in enumerate((1, 7)):
        oy, ox = (bgr[f].shape[0] - h) // 2, (bgr[f].shape[1] - w) // 2
        b = np.ascontiguousarray(bgr[f][oy:oy + h, ox:ox + w]); d = np.ascontiguousarray(depth[f][oy:oy + h, ox:ox + w])
        ch.upload(s, b, d); co.upload(s, b, d)
    ch.orb(0, 2); co.orb(0, 2)
    for s in range(2):
        for l in range(8):
            assert np.array_equal(ch.fetch_level(s, l), co.fetch_level(s, l)), "pyramid level %d of slot %d differs" % (l, s)
            assert np.array_equal(ch.fetch_blur_level(s, l), co.fetch_blur_level(s, l)), "blurred level %d of slot %d differs" % (l, s)
        kh, dh = ch.orb_fetch(s)
        ko, do = co.orb_fetch(s)
        assert len(kh) == len(ko) and len(kh) > nfeat // 2
        for field in ("x", "y", "size", "octave", "class_id", "depth_raw"):
            assert np.array_equal(kh[field], ko[field]), "keypoint field %s differs (slot %d)" % (field, s)
        assert np.array_equal(dh, do), "descriptors differ (slot %d): %d rows" % (s, int((dh != do).any(axis=1).sum()))


def test_orb_batch_with_frame_to_xcd_affinity_is_bit_exact(frames, libs):
    """Batches of >= 8 frame slots send every frame's workgroups to one XCD (vo_orb.hip header); same bits as frame-by-frame."""
    bgr, depth, _, _ = frames
    H, O = libs
    ctx, _ = make_ctx(H, n_features=1200, max_frames=10)
    octx, _ = make_ctx(O, n_features=1200, max_frames=1)
    for s in range(10):
        ctx.upload(s, bgr[s], depth[s])
    ctx.orb(0, 10)
    for s in (0, 3, 7, 8, 9):
        octx.upload(0, bgr[s], depth[s]); octx.orb(0, 1)
        kh, dh = ctx.orb_fetch(s)
        ko, do = octx.orb_fetch(0)
        assert len(kh) == len(ko) == 1200 and np.array_equal(dh, do)
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id", "depth_raw"):
            assert np.array_equal(kh[f], ko[f]), (s, f)
        for l in (0, 4, 7):
            assert np.array_equal(ctx.fetch_level(s, l), octx.fetch_level(0, l)) and np.array_equal(ctx.fetch_blur_level(s, l), octx.fetch_blur_level(0, l))


def test_match_bit_exact(frames, libs):
    bgr, depth, Twc, _ = frames
    out = []
    for L in libs:
        ctx, p = make_ctx(L, n_features=1000, max_frames=2, map_capacity=8192)
        ctx.upload(0, bgr[0], depth[0]); ctx.upload(1, bgr[6], depth[6])
        ctx.orb(0, 2)
        k0, d0 = ctx.orb_fetch(0)
        seed_map(ctx, p, k0, d0, Twc[0])
        m, ncand, mind = ctx.match(1, inv12(Twc[1]), 2.0, 30.0)
        out.append((m, ncand, mind))
    (mh, ch, dh), (mo, co, do) = out
    assert ch == co and dh == do and len(mh) == len(mo) > 100
    for f in ("map_index", "kp_index", "distance"):
        assert np.array_equal(mh[f], mo[f]), "match field %s differs" % f


def synth_corr(rng, n, outlier_frac, p, noise=0.3):
    X = rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 4.5])
    w = rng.normal(size=3) * 0.05
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    t = rng.normal(size=3) * 0.1
    pc = X @ R.T + t
    uv = np.stack([p.fx * pc[:, 0] / pc[:, 2] + p.cx, p.fy * pc[:, 1] / pc[:, 2] + p.cy], 1) + rng.normal(size=(n, 2)) * noise
    bad = rng.uniform(size=n) < outlier_frac
    uv[bad] = rng.uniform([0, 0], [640, 480], size=(int(bad.sum()), 2))
    return X.astype(np.float32), uv.astype(np.float32), np.concatenate([R.ravel(), t])


# the last case is BASELINE config 5 / 3 scale: 28 000 matches scored by 2048 hypotheses
@pytest.mark.parametrize("n,n_hyp,outl", [(300, 100, 0.3), (2000, 256, 0.5), (40, 64, 0.1), (5, 16, 0.0), (28000, 2048, 0.4)])
def test_ransac_counts_bit_exact_and_lm_close(libs, n, n_hyp, outl):
    rng = np.random.default_rng(n)
    out = []
    for L in libs:
        ctx, p = make_ctx(L, map_capacity=max(4096, n + 64), max_hypotheses=max(2048, n_hyp))
        X, uv, Tgt = synth_corr(np.random.default_rng(n), n, outl, p)
        ctx.matches_set(X, uv)
        T, inl, counts, iters, best = ctx.pnp_ransac(IDENT, n_hyp=n_hyp, seed=77)
        T2, mask, lm_it = ctx.pose_lm(T)
        out.append((T, inl, counts, iters, best, T2, mask, Tgt))
    h, o = out
    assert np.array_equal(h[2], o[2]), "per-hypothesis inlier counts differ: %d of %d" % (int((h[2] != o[2]).sum()), n_hyp)
    assert h[3] == o[3] and h[4] == o[4]
    assert np.array_equal(h[1], o[1])
    np.testing.assert_allclose(h[0], o[0], atol=1e-12)      # same hypothesis, same arithmetic
    np.testing.assert_allclose(h[5], o[5], atol=1e-9)       # LM: summation order differs
    assert np.array_equal(h[6], o[6])
    if n >= 40:
        assert np.abs(o[5] - o[7]).max() < 0.02             # and both found the true pose


def test_track_frame_matches_oracle(frames, libs):
    bgr, depth, Twc, _ = frames
    out = []
    for L in libs:
        ctx, p = make_ctx(L, n_features=2000, max_frames=2, map_capacity=8192)
        ctx.upload(0, bgr[0], depth[0]); ctx.upload(1, bgr[8], depth[8])
        ctx.orb(0, 2)
        k0, d0 = ctx.orb_fetch(0)
        seed_map(ctx, p, k0, d0, Twc[0])
        res, m = ctx.track(1, inv12(Twc[0]), L.default_track_params(n_hyp=128))
        out.append((res, m))
    (rh, mh), (ro, mo) = out
    for f in ("n_candidates", "n_matches", "n_ransac_inliers", "n_lm_inliers", "min_distance", "ransac_iters", "best_hypothesis"):
        assert getattr(rh, f) == getattr(ro, f), f
    assert rh.status == 0
    for f in ("map_index", "kp_index", "distance", "flags"):
        assert np.array_equal(mh[f], mo[f]), f
    np.testing.assert_allclose(np.array(rh.T_cw), np.array(ro.T_cw), atol=1e-9)
    assert np.abs(np.array(ro.T_cw) - inv12(Twc[8])).max() < 0.02


def test_config5_sizes_orb_and_tracking_match_oracle(libs):
    """BASELINE.json config 5 sizes: 1280x960 frames (fr1 intrinsics x2), 8000 features, 2048 RANSAC hypotheses.
    ORB stays bit-exact, the tracking chain gives the same integer results and the same pose."""
    syn = capi.Synth()
    sp = syn.params(seed=5, width=1280, height=960, fx=2 * 517.3, fy=2 * 516.5, cx=2 * 318.6, cy=2 * 255.3)
    bgr, depth, Twc, _ = syn.render(sp, 0, 6, threads=8)
    out = []
    for L in libs:
        ctx, p = make_ctx(L, width=1280, height=960, fx=sp.fx, fy=sp.fy, cx=sp.cx, cy=sp.cy, n_features=8000, max_frames=2,
                          map_capacity=16384, max_hypotheses=2048)
        ctx.upload(0, bgr[0], depth[0]); ctx.upload(1, bgr[5], depth[5])
        ctx.orb(0, 2)
        k0, d0 = ctx.orb_fetch(0)
        k1, d1 = ctx.orb_fetch(1)
        seed_map(ctx, p, k0, d0, Twc[0])
        res, m = ctx.track(1, inv12(Twc[0]), L.default_track_params(n_hyp=2048))
        out.append((k0, d0, k1, d1, res, m))
    (k0h, d0h, k1h, d1h, rh, mh), (k0o, d0o, k1o, d1o, ro, mo) = out
    assert len(k0h) == len(k0o) == 8000
    for kh, ko, dh, do in ((k0h, k0o, d0h, d0o), (k1h, k1o, d1h, d1o)):
        for field in ("x", "y", "size", "octave", "class_id", "depth_raw"):
            assert np.array_equal(kh[field], ko[field]), field
        assert np.array_equal(dh, do)
    for f in ("n_candidates", "n_matches", "n_ransac_inliers", "n_lm_inliers", "min_distance", "ransac_iters", "best_hypothesis"):
        assert getattr(rh, f) == getattr(ro, f), f
    assert rh.status == 0 and rh.n_ransac_inliers > 1000
    for f in ("map_index", "kp_index", "distance", "flags"):
        assert np.array_equal(mh[f], mo[f]), f
    np.testing.assert_allclose(np.array(rh.T_cw), np.array(ro.T_cw), atol=1e-9)


# D = 6 nfree: 24 (one 16-column panel + partial), 96 (full panels only), 180 and 186 (the second-generation solver with its last row block in global memory), 192 (first generation's LDS-resident limit),
# 216 (> limit: matrix in L2), 600 and 900 (k_ba_chol16g on the sizes between the tests of round 5 and the 960 the ABI admits; host-built pair lists);
# 1300 points -> > 20000 edges (threaded pair-list build); shuffle: edges not sorted by point
@pytest.mark.parametrize("nP,nX,nfree,shuffle", [(6, 400, 4, False), (18, 300, 16, False), (34, 500, 30, False), (35, 400, 31, False), (34, 300, 32, True),
                                                  (40, 500, 36, False), (34, 1300, 30, False), (26, 9000, 21, False),      # config-5 scale, ~160 k edges
                                                  (108, 400, 100, False), (158, 300, 150, False)])
@pytest.mark.parametrize("fuse", ["1", "0"])                 # 1: Cholesky + update in one launch (k_ba_cholup, the default), 0: launched apart (k_ba_chol16v2, k_ba_upchi2)
def test_local_ba_matches_oracle(libs, nP, nX, nfree, shuffle, fuse, monkeypatch):
    monkeypatch.setenv("VO_BA_FUSE_MAX", fuse)             # (read per chunk of steps: csrc/vo_ba.hip, ba_engine_enqueue)
    rng = np.random.default_rng(5)

    def expso3(w):
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K

    H, O = libs
    p = O.default_params()
    poses = []
    for j in range(nP):
        R = expso3(rng.normal(size=3) * 0.1)
        c = rng.normal(size=3) * 0.5
        poses.append(np.concatenate([R.ravel(), -R @ c]))
    poses = np.array(poses)
    X = rng.uniform(-2, 2, size=(nX, 3)) + np.array([0, 0, 5.0])
    ep, el, uv = [], [], []
    for k in range(nX):
        for j in range(nP):
            if rng.uniform() < 0.7:
                R, t = poses[j][:9].reshape(3, 3), poses[j][9:]
                pc = R @ X[k] + t
                o = rng.normal(size=2) * 0.3 + (rng.uniform(size=2) < 0.02) * 15.0
                ep.append(j); el.append(k); uv.append([p.fx * pc[0] / pc[2] + p.cx + o[0], p.fy * pc[1] / pc[2] + p.cy + o[1]])
    poses0 = poses.copy()
    for j in range(nfree):
        poses0[j][:9] = (expso3(rng.normal(size=3) * 0.01) @ poses[j][:9].reshape(3, 3)).ravel()
        poses0[j][9:] += rng.normal(size=3) * 0.02
    X0 = X + rng.normal(size=X.shape) * 0.05
    if shuffle:
        perm = rng.permutation(len(ep))
        ep = [ep[i] for i in perm]; el = [el[i] for i in perm]; uv = [uv[i] for i in perm]
    res = []
    for L in (H, O):
        ctx, _ = make_ctx(L, map_capacity=1024)
        res.append(ctx.local_ba(poses0, nfree, X0, ep, el, np.array(uv, dtype=np.float32)))
        ctx.close()
    (ph, xh, fh, rh), (po, xo, fo, ro) = res
    # the gain-ratio test near convergence is noise-limited: one LM iteration more or less is legitimate
    assert abs(rh.lm_iters - ro.lm_iters) <= (0 if nfree < 10 else 2)
    assert np.array_equal(fh, fo)
    tol = 1e-8 if nfree < 10 else 1e-6
    np.testing.assert_allclose(ph, po, atol=tol)
    np.testing.assert_allclose(xh, xo, atol=10 * tol)
    assert ro.chi2_final < 0.01 * ro.chi2_initial
    assert np.abs(po - poses[:nfree]).max() < 0.01


def test_hip_rejects_bad_arguments(libs):
    H, _ = libs
    p = H.default_params(width=32)
    with pytest.raises(capi.VoError):
        H.context(p)
    with pytest.raises(capi.VoError):       # k_describe copies the 39 x 40 window around a keypoint: keypoints must keep 20 pixels from the border
        H.context(H.default_params(edge_threshold=19))
    ctx, _ = make_ctx(H)
    with pytest.raises(capi.VoError):
        ctx.orb(0, 1)                       # no frame bound yet -> VO_E_STATE
    with pytest.raises(capi.VoError):
        ctx.map_set_active(np.array([1 << 30], dtype=np.int32))


def test_track_batch_equals_sequential_calls(frames, libs):
    """vo_track_batch on frames that share prior + map == one vo_track_frame per frame (same kernels, lanes)."""
    bgr, depth, Twc, _ = frames
    H, _ = libs
    ctx, p = make_ctx(H, n_features=1000, max_frames=5, map_capacity=8192, max_track_batch=4)
    for s in range(5):
        ctx.upload(s, bgr[2 * s], depth[2 * s])
    ctx.orb(0, 5)
    k0, d0 = ctx.orb_fetch(0)
    seed_map(ctx, p, k0, d0, Twc[0])
    tp = H.default_track_params()
    seeds = [101, 202, 303, 404]
    single = []
    for j, s in enumerate((1, 2, 3, 4)):
        tp.seed = seeds[j]
        single.append(ctx.track(s, inv12(Twc[0]), tp, cap=4096))
    res, ms = ctx.track_batch([1, 2, 3, 4], inv12(Twc[0]), tp, seeds, cap=4096)
    for j in range(4):
        rs, m1 = single[j]
        for f in ("n_candidates", "n_matches", "n_ransac_inliers", "n_lm_inliers", "min_distance", "ransac_iters", "best_hypothesis", "lm_iters"):
            assert getattr(res[j], f) == getattr(rs, f), (j, f)
        assert np.array_equal(np.array(res[j].T_cw), np.array(rs.T_cw))
        assert np.array_equal(ms[j], m1)
    # the deferred form (no match copy in the batch call, per-lane fetch afterwards) returns the same records
    res2, ms2 = ctx.track_batch_deferred([1, 2, 3, 4], inv12(Twc[0]), tp, seeds, cap=4096)
    for j in range(4):
        assert np.array_equal(np.array(res2[j].T_cw), np.array(res[j].T_cw)) and np.array_equal(ms2[j], ms[j])
    # the two-halves form (what the front-end's track-ahead uses): _end without a chain is a state error, then same records
    with pytest.raises(capi.VoError):
        ctx.track_batch_end(4)
    n = ctx.track_batch_begin([1, 2, 3, 4], inv12(Twc[0]), tp, seeds, cap=4096)
    with pytest.raises(capi.VoError):
        ctx.track_batch([1, 2], inv12(Twc[0]), tp, seeds[:2], cap=4096)      # the chain in flight owns the lane buffers
    res3, ms3 = ctx.track_batch_end(n, cap=4096)
    for j in range(4):
        for f in ("n_candidates", "n_matches", "n_ransac_inliers", "n_lm_inliers", "min_distance", "ransac_iters", "best_hypothesis", "lm_iters"):
            assert getattr(res3[j], f) == getattr(res[j], f), (j, f)
        assert np.array_equal(np.array(res3[j].T_cw), np.array(res[j].T_cw)) and np.array_equal(ms3[j], ms[j])


def test_vo_system_gpu_matches_oracle_trajectory(frames):
    """End to end through the libmyslam-style host layer: HIP path (look-ahead ORB, speculative batches) vs oracle."""
    from rgbd_visualodometry_amd import system, evaluate as ev
    bgr, depth, Twc, ts = frames
    n = len(ts)

    def run(lib, **opt):
        s = system.VoSystem(lib, number_of_features=800, **opt)
        look = opt.get("max_frames_in_flight", 1)
        poses, i = [], 0
        while i < n:
            k = min(look, n - i)
            s.prefetch(ts[i:i + k], [bgr[j].ctypes.data for j in range(i, i + k)], [depth[j].ctypes.data for j in range(i, i + k)],
                       bgr[0].strides[0], depth[0].strides[0], False)
            for _ in range(k):
                poses.append(s.add_prefetched()[1])
            i += k
        return np.array(poses), s.stats()

    po, so = run(ORACLE_LIB)
    ph, sh = run(system.HOST_LIB, max_frames_in_flight=5, track_batch=4)
    assert so["keyframes"] == sh["keyframes"] and so["map_points"] == sh["map_points"]
    np.testing.assert_allclose(ph, po, atol=1e-7)
    gt = {ts[i]: capi.pose12_to_tum(Twc[i]) for i in range(n)}
    a_h = ev.ate(gt, {ts[i]: capi.pose12_to_tum(ph[i]) for i in range(n)})["rmse"]
    a_o = ev.ate(gt, {ts[i]: capi.pose12_to_tum(po[i]) for i in range(n)})["rmse"]
    assert a_h <= 1.05 * a_o + 1e-6 and a_o < 0.05


def test_vo_system_config5_sizes_gpu_matches_oracle():
    """BASELINE config-5 sizes end to end (1280x960, 8000 features, 2048 hypotheses, local BA on): the HIP path with look-ahead
    ORB, speculative batches and an overlapped BA reproduces the oracle trajectory of the same schedule."""
    from rgbd_visualodometry_amd import system
    syn = capi.Synth()
    sp = syn.params(seed=9, width=1280, height=960, fx=2 * 517.3, fy=2 * 516.5, cx=2 * 318.6, cy=2 * 255.3)
    n = 10
    bgr, depth, Twc, ts = syn.render(sp, 0, n, threads=8)

    def run(lib, **opt):
        s = system.VoSystem(lib, width=1280, height=960, fx=sp.fx, fy=sp.fy, cx=sp.cx, cy=sp.cy, number_of_features=8000,
                            ransac_iterations=2048, backend_lag_frames=2, map_capacity=1 << 17, **opt)
        look = opt.get("max_frames_in_flight", 1)
        poses, i = [], 0
        while i < n:
            k = min(look, n - i)
            s.prefetch(ts[i:i + k], [bgr[j].ctypes.data for j in range(i, i + k)], [depth[j].ctypes.data for j in range(i, i + k)],
                       bgr[0].strides[0], depth[0].strides[0], False)
            for _ in range(k):
                poses.append(s.add_prefetched()[1])
            i += k
        return np.array(poses), s.stats()

    po, so = run(ORACLE_LIB)
    ph, sh = run(system.HOST_LIB, max_frames_in_flight=5, track_batch=4)
    assert so["keyframes"] == sh["keyframes"] >= 2 and so["map_points"] == sh["map_points"] and so["lost"] == sh["lost"] == 0
    assert so["ba_runs"] == sh["ba_runs"] >= 1 and so["ba_edges"] == sh["ba_edges"]
    np.testing.assert_allclose(ph, po, atol=1e-6)


def test_triangulate_all_and_reobservation_pass_match_oracle(frames):
    """SURVEY 8f-4 through the host layer: batched triangulation of every eligible point (vo_triangulate_batch) and the
    reference's disabled re-observation pass (src/frontend.cpp:408-463: re-detect each local keyframe, match the new map points,
    count the gated matches) give the same counters and the same trajectory on both implementations."""
    from rgbd_visualodometry_amd import system
    bgr, depth, Twc, ts = frames
    out = []
    for lib in (system.HOST_LIB, ORACLE_LIB):
        s = system.VoSystem(lib, number_of_features=600, triangulate_all=1, reobserve_new_mappoints=1, keyframe_rotation=0.01, keyframe_translation=0.01)
        poses = [s.add_frame(ts[i], bgr[i], depth[i])[1] for i in range(len(ts))]
        out.append((np.array(poses), s.stats()))
        s.close()
    (ph, sh), (po, so) = out
    assert sh["keyframes"] == so["keyframes"] >= 4
    assert sh["reobserved_matches"] == so["reobserved_matches"] > 100
    assert sh["triangulated"] == so["triangulated"]
    np.testing.assert_allclose(ph, po, atol=1e-6)


def test_device_resident_graph_cut_system_matches_oracle(frames):
    """SURVEY 8f-2 end to end: local BA graphs cut on the device from the resident observation table, overlapped back-end,
    look-ahead and speculative batches -- HIP against the CPU restatement of the same path, and against the host graph cut."""
    from rgbd_visualodometry_amd import system
    bgr, depth, Twc, ts = frames
    n = len(ts)

    def run(lib, **opt):
        s = system.VoSystem(lib, number_of_features=700, keyframe_rotation=0.02, keyframe_translation=0.02, backend_lag_frames=3, max_frames_in_flight=5, track_batch=4, **opt)
        poses, i = [], 0
        while i < n:
            k = min(5, n - i)
            s.prefetch(ts[i:i + k], [bgr[j].ctypes.data for j in range(i, i + k)], [depth[j].ctypes.data for j in range(i, i + k)], bgr[0].strides[0], depth[0].strides[0], False)
            for _ in range(k):
                poses.append(s.add_prefetched()[1])
            i += k
        s.flush()
        st = s.stats()
        s.close()
        return np.array(poses), st
    pd, sd = run(system.HOST_LIB, ba_device_graph=1, map_descriptors_on_device=1)      # both halves of SURVEY 8f-2 on
    po, so = run(ORACLE_LIB, ba_device_graph=1, map_descriptors_on_device=1)
    ph, sh = run(system.HOST_LIB)
    for k in ("keyframes", "ba_runs", "map_points", "ba_points", "ba_edges", "ba_poses", "ba_fixed"):
        assert sd[k] == so[k] == sh[k], k
    assert sd["ba_runs"] >= 2
    np.testing.assert_allclose(pd, po, atol=1e-6)
    np.testing.assert_allclose(pd, ph, atol=1e-6)
    # track-ahead (the default of this configuration: the next frames' chain is launched at the keyframe by vo_track_batch_begin and collected by
    # the next AddFrame) only moves work in time: without it the frames are tracked by the same lanes with the same seeds
    import os
    os.environ["VO_TRACK_AHEAD"] = "0"
    try:
        pn, sn = run(system.HOST_LIB, ba_device_graph=1, map_descriptors_on_device=1)
    finally:
        del os.environ["VO_TRACK_AHEAD"]
    assert sn["keyframes"] == sd["keyframes"] and sn["ba_runs"] == sd["ba_runs"]
    np.testing.assert_allclose(pn, pd, atol=1e-6)      # (the local BA's atomic sums are not ordered: two runs agree to rounding, not to the bit -- the tolerance of the comparisons above)


def test_map_points_created_from_frame_keypoints_on_the_device(frames, libs):
    """vo_map_upsert_from_frame copies descriptor rows on the device: the map it builds matches like one built from fetched
    descriptors (reference src/frontend.cpp:372-406 copies the keypoint's descriptor row into the new point)."""
    bgr, depth, Twc, ts = frames
    out = []
    for L in libs:
        for from_frame in (False, True):
            ctx, p = make_ctx(L, n_features=700, max_frames=2, map_capacity=8192)
            ctx.upload(0, bgr[0], depth[0]); ctx.orb(0, 1)
            kps, desc = ctx.orb_fetch(0)
            ok = np.nonzero(kps["depth_raw"] > 0)[0]
            z = kps["depth_raw"][ok] / 5000.0
            pc = np.stack([(kps["x"][ok] - p.cx) * z / p.fx, (kps["y"][ok] - p.cy) * z / p.fy, z], 1)
            R, t = Twc[0][:9].reshape(3, 3), Twc[0][9:]
            pw = pc @ R.T + t
            nrm = pw - t; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            idx = np.arange(len(ok), dtype=np.int32)[::-1].copy()          # slots in reverse keypoint order
            if from_frame:
                ctx.map_upsert_from_frame(0, ok, idx, pw, nrm, np.zeros(len(ok), np.uint8))
            else:
                ctx.map_upsert(idx, pw, nrm, desc[ok], np.zeros(len(ok), np.uint8))
            ctx.map_set_active(np.sort(idx))
            ctx.upload(1, bgr[2], depth[2]); ctx.orb(1, 1)
            m, ncand, mind = ctx.match(1, inv12(Twc[2]), 2.0, 30.0)
            out.append((m["map_index"].copy(), m["kp_index"].copy(), m["distance"].copy(), ncand, mind))
            ctx.close()
    for o in out[1:]:
        assert len(o[0]) > 200 and o[3] == out[0][3] and o[4] == out[0][4]
        for a, b in zip(o[:3], out[0][:3]):
            assert np.array_equal(a, b)


def test_degenerate_frames_behave_like_the_oracle(frames, libs):
    """Edge cases of the chain: a textureless frame (no FAST corner survives), an empty tracking map, a tracking map the
    frame cannot see -- same counts and status on both sides, no fault."""
    bgr, depth, Twc, _ = frames
    flat = np.full_like(bgr[0], 127)
    out = []
    for L in libs:
        ctx, p = make_ctx(L, n_features=500, max_frames=3, map_capacity=4096)
        ctx.upload(0, flat, depth[0]); ctx.upload(1, bgr[0], depth[0]); ctx.upload(2, bgr[3], depth[3])
        ctx.orb(0, 3)
        kf, df = ctx.orb_fetch(0)
        k1, d1 = ctx.orb_fetch(1)
        tp = L.default_track_params(n_hyp=64)
        r_empty, m_empty = ctx.track(2, inv12(Twc[0]), tp)                    # nothing in the map yet
        seed_map(ctx, p, k1, d1, Twc[0])
        r_flat, m_flat = ctx.track(0, inv12(Twc[0]), tp)                      # map, but a frame without keypoints
        away = inv12(Twc[0]).copy(); away[:9] = np.array([-1, 0, 0, 0, 1, 0, 0, 0, -1.0])     # looking the other way
        r_away, m_away = ctx.track(2, away, tp)
        out.append((len(kf), r_empty, len(m_empty), r_flat, len(m_flat), r_away, len(m_away)))
    h, o = out
    assert h[0] == o[0] == 0
    for a, b in ((h[1], o[1]), (h[3], o[3]), (h[5], o[5])):
        for f in ("status", "n_candidates", "n_matches", "n_ransac_inliers", "n_lm_inliers"):
            assert getattr(a, f) == getattr(b, f), f
    assert h[2] == o[2] == 0 and h[4] == o[4] == 0 and h[6] == o[6]
    assert h[1].n_ransac_inliers == 0 and h[3].n_ransac_inliers == 0


@pytest.mark.gpu
def test_orb_cut_bin_overflow_is_bit_exact(libs):
    from test_oracle import tie_image
    bgr, depth = tie_image()
    H, O = libs
    res = []
    for L in (H, O):
        ctx, _ = make_ctx(L, n_features=500, max_frames=1)
        ctx.upload(0, bgr, depth); ctx.orb(0, 1)
        res.append(ctx.orb_fetch(0))
        ctx.close()
    (kh, dh), (ko, do) = res
    assert len(kh) == len(ko) and int((kh["octave"] == 0).sum()) == 109      # level 0 fills its quota out of ~3500 ties (the coarse levels of this image have too few corners)
    for field in ("x", "y", "octave", "class_id"):
        assert np.array_equal(kh[field], ko[field]), field
    assert np.array_equal(dh, do)


def test_frame_upload_keeps_the_callers_strides(frames, libs):
    """vo_frame_upload takes any row stride >= the tight one (the slot keeps it: one contiguous copy per image) and a slot may be
    re-uploaded with a different stride; the ORB result does not depend on it."""
    import ctypes as C
    bgr, depth, _, _ = frames
    H, _ = libs
    c, _ = make_ctx(H, n_features=500, max_frames=2)
    c.upload(0, bgr[3], depth[3])                                        # tight: 1920 / 1280 bytes per row
    pb = np.zeros((480, 2048), np.uint8); pb[:, :1920] = bgr[3].reshape(480, 1920)
    pd = np.zeros((480, 704), np.uint16); pd[:, :640] = depth[3]
    c.upload(1, bgr[5], depth[5])                                        # slot 1: tight first, then padded rows (a larger image buffer)
    H.check(H.lib.vo_frame_upload(c.h, 1, C.c_void_p(pb.ctypes.data), 2048, C.c_void_p(pd.ctypes.data), 1408), "vo_frame_upload")
    c.orb(0, 2)
    k0, d0 = c.orb_fetch(0)
    k1, d1 = c.orb_fetch(1)
    for field in ("x", "y", "octave", "depth_raw"):
        assert np.array_equal(k0[field], k1[field]), field
    assert np.array_equal(d0, d1) and len(k0) == 500
    c.close()


def test_preloaded_frames_equal_uploaded_ones(frames, libs):
    """vo_frames_preload copies a look-ahead batch ahead of time on the context's copy stream into one of two slabs (one 2-D copy per image
    kind when the host frames are evenly spaced); the vo_frame_upload calls that follow from the same page-locked buffers take the frames
    from there.  Three rounds through the same slots (the slabs take turns), unevenly spaced frames (per-frame copies), a slot that is
    preloaded from one buffer and uploaded from another (falls back to the copy), and pageable memory (preload is a no-op): the ORB
    results are those of plain uploads."""
    import ctypes as C
    bgr, depth, _, _ = frames
    H, _ = libs
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    fb, fd, nfr = 640 * 480 * 3, 640 * 480 * 2, 6
    pb, pd = C.c_void_p(), C.c_void_p()
    assert hip.hipHostMalloc(C.byref(pb), fb * nfr, 0) == 0 and hip.hipHostMalloc(C.byref(pd), fd * nfr, 0) == 0
    C.memmove(pb, np.ascontiguousarray(bgr[:nfr]).ctypes.data, fb * nfr)
    C.memmove(pd, np.ascontiguousarray(depth[:nfr]).ctypes.data, fd * nfr)
    bp = lambda i: pb.value + i * fb
    dp = lambda i: pd.value + i * fd
    c, _ = make_ctx(H, n_features=500, max_frames=4)
    r, _ = make_ctx(H, n_features=500, max_frames=4)
    rounds = [([0, 1, 2, 3], [0, 1, 2, 3]),                 # evenly spaced: one 2-D copy per image kind
              ([4, 5, 0, 2], [4, 5, 1, 2]),                 # uneven spacing: per-frame copies; slot 2 is preloaded with frame 0 but uploaded from frame 1's buffers
              ([2, 3, 4, 5], [2, 3, 4, 5])]                 # the first slab again
    for rnd, (pre, ids) in enumerate(rounds):
        c.preload_ptrs(0, [bp(i) for i in pre], 1920, [dp(i) for i in pre], 1280)
        for s, i in enumerate(ids):
            c.upload_ptr(s, bp(i), 1920, dp(i), 1280)
            r.upload(s, bgr[i], depth[i])
        c.orb(0, 4); r.orb(0, 4)
        for s in range(4):
            kc, dc = c.orb_fetch(s); kr, dr = r.orb_fetch(s)
            for field in ("x", "y", "octave", "depth_raw"):
                assert np.array_equal(kc[field], kr[field]), (rnd, s, field)
            assert np.array_equal(dc, dr), (rnd, s)
    c.preload_ptrs(0, [bgr[3].ctypes.data], 1920, [depth[3].ctypes.data], 1280)      # pageable: nothing happens
    c.upload(0, bgr[3], depth[3]); r.upload(0, bgr[3], depth[3])
    c.orb(0, 1); r.orb(0, 1)
    assert np.array_equal(c.orb_fetch(0)[1], r.orb_fetch(0)[1])
    c.close(); r.close()
    hip.hipHostFree(pb); hip.hipHostFree(pd)


def test_frame_upload_reads_no_byte_behind_the_last_row(frames, libs):
    """A column slice of a wider image (a cv::Mat ROI): rows are padded, but the padding behind the LAST row is not the caller's.
    The images are placed so that their last pixel row ends exactly at a PROT_NONE guard page: an upload that copies stride * H
    bytes faults, one that copies (H - 1) * stride + row bytes does not."""
    import ctypes as C
    import mmap
    bgr, depth, _, _ = frames
    H, _ = libs
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE

    def guarded_view(img2d, stride):                                       # img2d: (rows, row_bytes) uint8
        rows, rb = img2d.shape
        need = (rows - 1) * stride + rb
        npages = (need + page - 1) // page
        mm = mmap.mmap(-1, (npages + 1) * page)
        base = C.addressof(C.c_char.from_buffer(mm))
        assert libc.mprotect(base + npages * page, page, 0) == 0            # PROT_NONE behind the image
        start = npages * page - need
        buf = np.frombuffer(mm, dtype=np.uint8, count=npages * page)
        for r in range(rows):
            buf[start + r * stride: start + r * stride + rb] = img2d[r]
        return mm, base + start

    c, _ = make_ctx(H, n_features=500, max_frames=2)
    c.upload(0, bgr[3], depth[3])
    mb, pb = guarded_view(bgr[3].reshape(480, 1920), 2048)
    md, pd = guarded_view(depth[3].view(np.uint8).reshape(480, 1280), 1408)
    H.check(H.lib.vo_frame_upload(c.h, 1, C.c_void_p(pb), 2048, C.c_void_p(pd), 1408), "vo_frame_upload")
    c.orb(0, 2)
    k0, d0 = c.orb_fetch(0)
    k1, d1 = c.orb_fetch(1)
    for field in ("x", "y", "octave", "depth_raw"):
        assert np.array_equal(k0[field], k1[field]), field
    assert np.array_equal(d0, d1) and len(k0) == 500
    c.close()


@pytest.mark.parametrize("mfma", ["0", "1"])
def test_both_matching_kernels_give_the_oracles_matches(frames, libs, mfma, monkeypatch):
    """k_match (vector ALU) and k_match_mfma (int8 matrix cores) are interchangeable bit for bit: the launcher picks by problem size; here
    each is forced in turn (VO_MATCH_MFMA, read per launch) on a frame with a partial last keypoint tile."""
    monkeypatch.setenv("VO_MATCH_MFMA", mfma)
    bgr, depth, Twc, _ = frames
    H, O = libs
    out = {}
    for name, L in (("hip", H), ("oracle", O)):
        c, p = make_ctx(L, n_features=777, max_frames=2, map_capacity=4096)          # 777 keypoints: 12 full tiles of 64 + 9
        c.upload(0, bgr[0], depth[0]); c.upload(1, bgr[4], depth[4])
        c.orb(0, 2)
        k0, d0 = c.orb_fetch(0)
        n = seed_map(c, p, k0, d0, Twc[0])
        r, m = c.track(1, inv12(Twc[0]), L.default_track_params())
        out[name] = (n, r.n_candidates, r.n_matches, m)
        c.close()
    assert out["hip"][:3] == out["oracle"][:3] and out["hip"][1] > 300
    assert np.array_equal(out["hip"][3], out["oracle"][3])
