#!/usr/bin/env python3
"""Per-kernel issue figures from the SQ pass of tools/profile_round.sh (profiling helper, not a test).

usage: sq_summary.py gpurun_out/prof_<tag>/sq ["bench workload args"] > profiles/<tag>_sq_counters.txt"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
order = []
for row in csv.DictReader(open(files[0])):
    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mi355rt::", "").replace("mi355rt::", "")
    if "kernel" not in k or "at::" in k:
        continue
    if k not in acc:
        order.append(k)
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    calls[k].add(row["Dispatch_Id"])
print("# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES "
      "-- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --frames-in-flight 1 " + (sys.argv[2] if len(sys.argv) > 2 else ""))
print("# per launch (averaged over the launches sampled); walk_kernel<1,*> is the one instrumented frame; SIMD VALU busy = "
      "4 cycles x SQ_INSTS_VALU / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 shader engines)")
for k in order:
    n = len(calls[k])
    o = {a: b / n for a, b in acc[k].items()}
    waves = max(o.get("SQ_WAVES", 0.0), 1.0)
    print("%-22s launches=%d waves=%d VALU insts/wave=%d  lane utilisation of VALU=%.3f  wave cycles waiting=%.3f  SALU insts/wave=%d  SIMD VALU busy=%.2f"
          % (k[:22], n, waves, o["SQ_INSTS_VALU"] / waves, o["SQ_THREAD_CYCLES_VALU"] / max(o["SQ_ACTIVE_INST_VALU"] * 64.0, 1.0),
             o["SQ_WAIT_ANY"] / max(o["SQ_WAVE_CYCLES"], 1.0), o["SQ_INSTS_SALU"] / waves,
             4.0 * o["SQ_INSTS_VALU"] / 1024.0 / max(o["SQ_BUSY_CYCLES"] / 32.0, 1.0)))
