"""ATE / RPE / associate restatement vs vectors captured from the reference's tools/*.py."""
import json
import os

import numpy as np
import pytest

from rgbd_visualodometry_amd import evaluate as ev

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_tools_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_associate_matches_reference(gold):
    for c in gold["associate"]:
        got = ev.associate(c["first"], c["second"], c["offset"], c["max_difference"])
        assert [[a, b] for a, b in got] == c["matches"]


def test_horn_alignment_and_ate_stats(gold):
    for c in gold["ate"]:
        rot, trans, err = ev.horn_align(np.array(c["model"]), np.array(c["data"]))
        np.testing.assert_allclose(rot, c["rot"], atol=1e-10)
        np.testing.assert_allclose(trans, c["trans"], atol=1e-10)
        np.testing.assert_allclose(err, c["trans_error"], atol=1e-10)
        st = ev._stats(err)
        for k in ("rmse", "mean", "median", "std", "min", "max"):
            assert st[k] == pytest.approx(c[k], abs=1e-10)


def test_pose_matrix(gold):
    for c in gold["transform44"]:
        np.testing.assert_allclose(ev.pose_matrix(c["row"]), c["matrix"], atol=1e-13)


def test_rpe_rows_and_summary(gold):
    for c in gold["rpe"]:
        gt = {r[0]: ev.pose_matrix(r) for r in c["gt"]}
        est = {r[0]: ev.pose_matrix(r) for r in c["est"]}
        rows = ev.rpe(gt, est, c["max_pairs"], c["fixed_delta"], c["delta"], c["delta_unit"], c["offset"], c["scale"])
        ref = np.array(c["result"])
        got = np.array(rows)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got[:, :4], ref[:, :4], atol=0)
        np.testing.assert_allclose(got[:, 4], ref[:, 4], atol=1e-9)
        # arccos near 1 amplifies rounding: sqrt(eps)-level agreement
        np.testing.assert_allclose(got[:, 5], ref[:, 5], atol=5e-8)
        s = ev.rpe_summary(rows)
        assert s["trans_rmse"] == pytest.approx(c["trans_rmse"], abs=1e-9)
        if "rot_rmse_deg" in c:
            assert s["rot_deg_rmse"] == pytest.approx(c["rot_rmse_deg"], abs=1e-5)


def test_ate_end_to_end_files(tmp_path, gold):
    c = gold["rpe"][0]
    g = tmp_path / "gt.txt"
    e = tmp_path / "est.txt"
    ev.write_trajectory(str(g), c["gt"])
    ev.write_trajectory(str(e), c["est"])
    gt = ev.read_stamped_file(str(g))
    est = ev.read_stamped_file(str(e))
    st = ev.ate(gt, est)
    assert st["pairs"] > 50 and 0 < st["rmse"] < 0.05
    assert len(ev.read_trajectory(str(g))) == len(c["gt"])


def test_rpe_distance_units_work():
    rows = [[float(i), 0.1 * i, 0, 0, 0, 0, 0, 1] for i in range(30)]
    tr = {r[0]: ev.pose_matrix(r) for r in rows}
    out = ev.rpe(tr, tr, fixed_delta=True, delta=0.5, delta_unit="m")
    assert max(r[4] for r in out) < 1e-12
