"""Average HIP-event duration of named kernels in the bench's own timing pass: python scripts/kernel_avg.py k_match_mfma k_pose_lm ... (a 96-step bench run)"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "96", "--warmup", "32", "--no-cpu-baseline", "--no-latency-mode", "--multi-streams="], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
d = json.load(open(os.path.join(ROOT, "bench_detail.json")))
k = d["roofline"]["kernels"]
print("frames/s %.1f" % d["value"], " ".join("%s %.2f us (%d)" % (n, k[n]["avg_us"], k[n]["launches"]) for n in sys.argv[1:] if n in k))
