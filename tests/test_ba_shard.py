"""e-3 (BASELINE config 5): the local BA of reference src/backend.cpp:19-195 sharded over ranks by point, with an all-reduce of the
reduced system per LM step (include/vo_hip.h: vo_set_ba_shard).  Here the ranks are THREADS of one process, each with a context of its own; the exchange
is a barrier-synchronised sum (tests/test_shard_gloo.py runs the same thing over a real gloo collective in two processes).  The bar: every rank ends
with the un-sharded solve's result -- identical edge flags, poses to 1e-6 -- and all ranks with bit-identical poses (they stayed in lockstep)."""
import threading

import numpy as np
import pytest

from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi

IDENT = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


def ba_problem(rng, nP, nX, nfree, p, gross=0.03):
    poses = np.tile(IDENT, (nP, 1)); poses[:, 9] = -0.06 * np.arange(nP); poses[:, 10] = 0.02 * np.sin(np.arange(nP))
    X = rng.uniform(-1.5, 1.5, (nX, 3)) + [0.3, 0, 5]
    ep, el, uv = [], [], []
    for k in range(nX):
        for j in range(nP):
            if (k + 3 * j) % 4 == 0:
                continue
            pc = X[k] + poses[j][9:]
            o = rng.normal(size=2) * 0.3 + (rng.uniform(size=2) < gross) * 12.0
            ep.append(j); el.append(k); uv.append([p.fx * pc[0] / pc[2] + p.cx + o[0], p.fy * pc[1] / pc[2] + p.cy + o[1]])
    poses0 = poses.copy(); poses0[:nfree, 9:] += rng.normal(size=(nfree, 3)) * 0.01
    return poses0, X + rng.normal(size=X.shape) * 0.03, np.array(ep, np.int32), np.array(el, np.int32), np.array(uv, np.float32)


class ThreadRanks:
    """An in-place SUM over `world` threads: every rank enters with its array, all leave with the sum (what an all-reduce does)."""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = [0] * world
        self.sizes = [[] for _ in range(world)]

    def exchange(self, rank):
        def f(a):
            self.calls[rank] += 1; self.sizes[rank].append(len(a))
            self.slots[rank] = a.copy()
            self.bar.wait()
            tot = np.sum(self.slots, axis=0)
            self.bar.wait()
            a[:] = tot
        return f


def sharded(lib_path, world, pr, nfree):
    L = capi.load(lib_path)
    tr = ThreadRanks(world)
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            ctx = L.context(L.default_params(map_capacity=1024))
            ctx.set_ba_shard(r, world, tr.exchange(r))
            out[r] = ctx.local_ba(pr[0], nfree, *pr[1:])
            ctx.close()
        except Exception as e:                                  # a failing rank must not leave the others at the barrier
            err[r] = e
            tr.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not any(err), err
    return out, tr


def check(lib_path, world, nP, nX, nfree, seed=7):
    L = capi.load(lib_path)
    pr = ba_problem(np.random.default_rng(seed), nP, nX, nfree, L.default_params())
    ctx = L.context(L.default_params(map_capacity=1024))
    p0, x0, f0, r0 = ctx.local_ba(pr[0], nfree, *pr[1:])
    ctx.close()
    out, tr = sharded(lib_path, world, pr, nfree)
    D = 6 * nfree
    for r in range(world):
        p1, x1, f1, r1 = out[r]
        assert np.array_equal(f1, f0), "edge flags of rank %d differ from the un-sharded solve's: %d" % (r, int((f1 != f0).sum()))
        np.testing.assert_allclose(p1, p0, atol=1e-6); np.testing.assert_allclose(x1, x0, atol=1e-5)
        assert abs(r1.lm_iters - r0.lm_iters) <= 2
        assert abs(r1.chi2_final - r0.chi2_final) <= 1e-6 * max(1.0, r0.chi2_final) and abs(r1.chi2_initial - r0.chi2_initial) <= 1e-9 * max(1.0, r0.chi2_initial)
        assert np.array_equal(p1, out[0][0]) and np.array_equal(x1, out[0][1])      # lockstep: the ranks' results are the same bits
    assert (f0 != 0).sum() > 0
    assert len(set(tr.calls)) == 1 and (tr.calls[0] - 1) % 3 == 0 and tr.calls[0] >= 3 * 10 + 1
    assert D * D + D in tr.sizes[0]                              # the reduced system S, b_s: SURVEY 8e's all-reduce
    return tr


@pytest.mark.parametrize("world,nP,nX,nfree", [(2, 9, 400, 6), (3, 26, 300, 21), (4, 12, 200, 3)])
def test_ba_edge_shard_equals_the_unsharded_solve_on_the_restatement(world, nP, nX, nfree):
    check(ORACLE_LIB, world, nP, nX, nfree)


@pytest.mark.gpu
@pytest.mark.parametrize("world,nP,nX,nfree", [(2, 9, 400, 6), (3, 26, 300, 21), (2, 40, 500, 36), (4, 12, 200, 3)])
def test_ba_edge_shard_equals_the_unsharded_solve_hip(world, nP, nX, nfree):
    """The HIP ranks (k_ba_shard_* + the launch-per-phase step with S in global memory, D = 18 ... 216) against the un-sharded engine's solve -- and that against
    the restatement's un-sharded solve, so a bug common to both HIP forms cannot hide."""
    check(capi.HIP_LIB, world, nP, nX, nfree)
    H, O = capi.load(capi.HIP_LIB), capi.load(ORACLE_LIB)
    pr = ba_problem(np.random.default_rng(7), nP, nX, nfree, O.default_params())
    a = H.context(H.default_params(map_capacity=1024)); b = O.context(O.default_params(map_capacity=1024))
    ph, xh, fh, rh = a.local_ba(pr[0], nfree, *pr[1:]); po, xo, fo, ro = b.local_ba(pr[0], nfree, *pr[1:])
    a.close(); b.close()
    assert np.array_equal(fh, fo)
    np.testing.assert_allclose(ph, po, atol=1e-6)


@pytest.mark.gpu
def test_the_device_graph_cut_refuses_a_sharded_context():
    H = capi.load(capi.HIP_LIB)
    c = H.context(H.default_params(map_capacity=64)); t = H.context(H.default_params(map_capacity=64))
    c.set_ba_shard(0, 2, lambda a: None)
    with pytest.raises(capi.VoError):
        c.local_ba_resident(t, [0])
    c.set_ba_shard(0, 1, None)
    c.close(); t.close()
