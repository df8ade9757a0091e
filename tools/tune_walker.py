"""Times the bench frame under different walker tunables (profiling helper, not a test).
usage: tune_walker.py NAME=v1,v2,... [NAME=...]  -- runs the cartesian product, each in a child process."""
import itertools, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
for combo in itertools.product(*[v for _, v in axes]):
    env = dict(os.environ)
    for (n, _), v in zip(axes, combo):
        env[n] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(dict(zip([n for n, _ in axes], combo)), "ms_per_step", j["ms_per_step"], "Mrays/s", j["value"], flush=True)
    except Exception as e:
        print(combo, "FAILED", out.stderr[-300:], flush=True)
