# GPU box: which timed step of `bench.py --mode train` is the slow one?
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
env = dict(os.environ, PNYOLO_BENCH_STEP_TIMES="1")
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--steps", "30"] + sys.argv[1:], capture_output=True, text=True, env=env)
d = json.loads(out.stdout.strip().splitlines()[-1])
print("ms/step %.2f" % d["ms_per_step"], " per step:", " ".join("%.1f" % x for x in d.get("step_ms", [])))
