"""Every environment switch the product reads (rgbd_visualodometry_amd/csrc + host/src: VO_TRACE, VO_BA_FUSE_MAX, VO_BA_ENGINES, VO_MATCH_MFMA,
VO_LM_WGS, VO_OBS_CAP, VO_OBS_CAP0, VO_TEST_FAIL_CUT_AT, VO_TRACK_AHEAD) selects a code path; each path is named in a test.  This file holds the
ones no other test sets: the -m gpu cases compare the HIP path under the switch with the CPU oracle, the CPU case counts the
switches so that a new one cannot arrive without a test."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IDENT = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
SWITCHES = {"VO_TRACE", "VO_BA_FUSE_MAX", "VO_BA_ENGINES", "VO_MATCH_MFMA", "VO_LM_WGS", "VO_OBS_CAP", "VO_OBS_CAP0", "VO_TEST_FAIL_CUT_AT", "VO_TRACK_AHEAD", "VO_BA_STEP_MODE"}


def test_the_product_reads_only_switches_that_a_test_names():
    """grep getenv over csrc/ and host/src/: at most 12 call sites, every name in SWITCHES, every name of SWITCHES that occurs set by some test."""
    pkg = os.path.join(ROOT, "rgbd_visualodometry_amd")
    found, sites = set(), 0
    for sub in ("csrc", os.path.join("host", "src"), os.path.join("host", "myslam"), os.path.join("host", "app")):
        d = os.path.join(pkg, sub)
        for fn in sorted(os.listdir(d)):
            if not fn.endswith((".hip", ".h", ".cpp")):
                continue
            for line in open(os.path.join(d, fn), errors="replace"):
                if "getenv" in line:
                    sites += 1
                    found.update(re.findall(r'getenv\("([A-Z0-9_]+)"\)', line))
    assert sites <= 12, sites
    assert found <= SWITCHES, found - SWITCHES
    tests = "".join(open(os.path.join(ROOT, "tests", f)).read() for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py") and f != "test_switches.py")
    tests += "".join(l for l in open(__file__) if not l.startswith("SWITCHES = {"))      # this file without the list itself
    for name in found:
        assert re.search(r'setenv\("%s"|\b%s=|environ\["%s"\] =' % (name, name, name), tests), "no test sets %s" % name


def _trace_run(lib_expr):
    code = ("import numpy as np\nfrom rgbd_visualodometry_amd import capi, system\nfrom oracle import ORACLE_LIB\n"
            "syn = capi.Synth(); sp = syn.params(seed=11); bgr, depth, Twc, ts = syn.render(sp, 0, 8, threads=4)\n"
            "s = system.VoSystem(%s, number_of_features=400, keyframe_rotation=0.01, keyframe_translation=0.01)\n"
            "for i in range(8): s.add_frame(ts[i], bgr[i], depth[i])\n"
            "print('keyframes', s.stats()['keyframes'])\n" % lib_expr)
    env = dict(os.environ, VO_TRACE="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout, r.stderr


def test_vo_trace_prints_the_host_layers_lines_on_the_restatement():
    """VO_TRACE is read once per process (vo_trace_level): a child process with it set prints the [vo_trace] lines of the host layer."""
    out, err = _trace_run("ORACLE_LIB")
    assert "keyframes" in out and "[vo_trace] frontend ms" in err


@pytest.mark.gpu
def test_vo_trace_prints_the_librarys_lines_on_the_hip_path():
    out, err = _trace_run("system.HOST_LIB")
    assert "keyframes" in out and "[vo_trace] frontend ms" in err and "[vo_trace] BA engine" in err


def _synth_corr(rng, n, outlier_frac, p, noise=0.3):
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
    return X.astype(np.float32), uv.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("wgs,n", [("2", 700), ("4", 2500), ("3", 40)])
def test_pose_lm_over_several_workgroups_matches_the_oracle(wgs, n, monkeypatch):
    """k_pose_lm's several-workgroup form (per-pass partial sums, last arrival finishes the 6x6 solve; the launcher takes it above 6000 inliers) forced
    at small sizes by VO_LM_WGS (read per launch): same poses to 1e-9, same inlier mask."""
    monkeypatch.setenv("VO_LM_WGS", wgs)
    out = []
    for path in (capi.HIP_LIB, ORACLE_LIB):
        L = capi.load(path)
        ctx = L.context(L.default_params(map_capacity=4096))
        X, uv = _synth_corr(np.random.default_rng(n), n, 0.3, L.default_params())
        ctx.matches_set(X, uv)
        T, inl, counts, iters, best = ctx.pnp_ransac(IDENT, n_hyp=100, seed=5)
        T2, mask, lm_it = ctx.pose_lm(T)
        out.append((T2, mask, inl))
        ctx.close()
    h, o = out
    assert np.array_equal(h[2], o[2]) and np.array_equal(h[1], o[1])
    np.testing.assert_allclose(h[0], o[0], atol=1e-9)


def _ba_problem(rng, nP, nX, nfree, p):
    poses = np.tile(IDENT, (nP, 1)); poses[:, 9] = -0.06 * np.arange(nP)
    X = rng.uniform(-1.5, 1.5, (nX, 3)) + [0.3, 0, 5]
    ep, el, uv = [], [], []
    for k in range(nX):
        for j in range(nP):
            if (k + j) % 4 == 0:
                continue
            pc = X[k] + poses[j][9:]
            ep.append(j); el.append(k); uv.append([p.fx * pc[0] / pc[2] + p.cx + rng.normal() * 0.3, p.fy * pc[1] / pc[2] + p.cy + rng.normal() * 0.3])
    poses0 = poses.copy(); poses0[:nfree, 9:] += rng.normal(size=(nfree, 3)) * 0.01
    return poses0, X + rng.normal(size=X.shape) * 0.03, np.array(ep, np.int32), np.array(el, np.int32), np.array(uv, np.float32)


@pytest.mark.gpu
def test_two_ba_engines_solve_what_one_solves(monkeypatch):
    """VO_BA_ENGINES=2 (read when a device's first context is created): contexts are bound to the engines in turn; two contexts solving two different
    problems at the same time through two engines give the oracle's results."""
    import threading
    monkeypatch.setenv("VO_BA_ENGINES", "2")
    H, O = capi.load(capi.HIP_LIB), capi.load(ORACLE_LIB)
    p = O.default_params()
    probs = [_ba_problem(np.random.default_rng(s), 8 + 2 * s, 500 + 100 * s, 5 + s, p) for s in range(2)]
    ctxs = [H.context(H.default_params(map_capacity=1024)) for _ in probs]
    res = [None] * len(probs)

    def work(i):
        res[i] = ctxs[i].local_ba(probs[i][0], 5 + i, *probs[i][1:])
    for rep in range(2):                                        # the second pass finds both engines created
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(probs))]
        [t.start() for t in th]; [t.join() for t in th]
        for i, pr in enumerate(probs):
            oc = O.context(O.default_params(map_capacity=1024))
            po, xo, fo, ro = oc.local_ba(pr[0], 5 + i, *pr[1:])
            oc.close()
            ph, xh, fh, rh = res[i]
            assert np.array_equal(fh, fo)
            np.testing.assert_allclose(ph, po, atol=1e-8); np.testing.assert_allclose(xh, xo, atol=1e-7)
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_full_observation_table_on_the_device_falls_back_to_the_host_graph_cut(monkeypatch, capfd):
    """VO_OBS_CAP on the HIP path (tests/test_oracle.py holds the restatement's twin): when a keyframe no longer fits the observation table the
    back-end goes back to cutting its graphs on the host (one line on stderr) and the stream goes on with the same trajectory."""
    from rgbd_visualodometry_amd import system
    syn = capi.Synth()
    bgr, depth, Twc, ts = syn.render(syn.params(seed=3), 0, 24, threads=8)
    kw = dict(number_of_features=500, keyframe_rotation=0.02, keyframe_translation=0.02, backend_lag_frames=3, max_frames_in_flight=8, track_batch=4)

    def run(**opt):
        s = system.VoSystem(system.HOST_LIB, **kw, **opt)
        poses, i = [], 0
        while i < len(ts):
            k = min(8, len(ts) - i)
            s.prefetch(ts[i:i + k], [bgr[j].ctypes.data for j in range(i, i + k)], [depth[j].ctypes.data for j in range(i, i + k)], bgr[0].strides[0], depth[0].strides[0], False)
            for _ in range(k):
                poses.append(s.add_prefetched()[1])
            i += k
        return np.array(poses), s.stats()
    host, sh = run()
    monkeypatch.setenv("VO_OBS_CAP", "2000")                # the first keyframes fit (500 observations each), a later one does not
    dev, sd = run(ba_device_graph=1)
    assert "device observation table full" in capfd.readouterr().err
    assert sd["keyframes"] == sh["keyframes"] >= 5 and sd["ba_runs"] == sh["ba_runs"] and sd["lost"] == 0
    np.testing.assert_allclose(dev, host, atol=1e-5)
