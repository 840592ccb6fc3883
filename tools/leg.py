"""python tools/leg.py <legs>: ms/step of the named bench.py model legs (prints a dict)."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--models", sys.argv[1], "--no-cpu"],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print({k: round(v["ms_per_step"], 2) for k, v in d["models"].items()}, flush=True)
