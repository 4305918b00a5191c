"""Driver-level timing: run_vo on a synthetic TUM-format PNG dataset, sequential vs look-ahead mode.
Writes N frames to a temp dir, runs host/app/run_vo twice, reports wall time per frame including PNG decode."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd_visualodometry_amd import capi, dataset, evaluate as ev
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "rgbd_visualodometry_amd", "host", "app", "run_vo")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
syn = capi.Synth()
bgr, depth, Twc, ts = syn.render(syn.params(seed=0), 0, n)
with tempfile.TemporaryDirectory() as tmp:
    dataset.write_tum_dataset(tmp, bgr, depth, ts, Twc)
    gt = ev.read_stamped_file(os.path.join(tmp, "groundtruth.txt"))
    for name, over in (("sequential (reference driver loop)", {}), ("lookahead 32, 8 decode threads, track_batch 8, BA lag 8",
                       dict(lookahead_frames=32, decode_threads=8, track_batch=8, backend_lag_frames=8))):
        cfg, out = os.path.join(tmp, "cfg.yaml"), os.path.join(tmp, "traj.txt")
        dataset.write_config(cfg, tmp, out, number_of_features=2000, **over)
        t0 = time.perf_counter()
        r = subprocess.run([BIN, cfg], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        el = time.perf_counter() - t0
        assert r.returncode == 0, r.stdout[-1000:]
        traj = ev.read_stamped_file(out)
        mean_ms = [l for l in r.stdout.splitlines() if "mean AddFrame" in l]
        print("%-60s %5d frames  %.1f ms/frame wall (decode + track + process start), ATE %.4f | %s" % (name, len(traj), 1e3 * el / max(1, len(traj)), ev.ate(gt, traj)["rmse"], mean_ms[0] if mean_ms else ""))
