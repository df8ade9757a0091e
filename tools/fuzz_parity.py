"""Randomised GPU-vs-oracle frames (validation helper; a seeded subset runs in tests/test_gpu_fuzz.py).
usage: fuzz_parity.py [first_seed] [n_cases] [analytic|rays|grazing]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (one HIP runtime per process: torch's first)
import ray_tracer_amd as rt
from fuzz_cases import random_case, random_analytic_case, random_rays_case, random_grazing_case

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t_start = time.time()
for seed in range(first, first + count):
    kind = sys.argv[3] if len(sys.argv) > 3 else "frames"
    ok, desc = {"frames": random_case, "analytic": random_analytic_case, "rays": random_rays_case, "grazing": random_grazing_case}[kind](rt, seed)
    bad += 0 if ok else 1
    print(desc, flush=True)
print("%d cases, %d mismatches, %.0f s" % (count, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
