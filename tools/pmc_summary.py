"""Summarises a rocprofv3 --pmc counter_collection.csv per kernel (profiling helper, not a test)."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    files = glob.glob(d + "/*/*counter_collection.csv")
    if not files:
        print(d, "no file")
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void mi355rt::", "").replace("mi355rt::", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
    for k, v in acc.items():
        if "kernel" not in k or "at::" in k:
            continue
        n = len(calls[k])
        out = {a: b / n for a, b in v.items()}
        extra = {}
        if "SQ_THREAD_CYCLES_VALU" in out and out.get("SQ_ACTIVE_INST_VALU"):
            extra["valu_lane_util"] = out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_ACTIVE_INST_VALU"] * 64)
        if "SQ_WAIT_ANY" in out and out.get("SQ_WAVE_CYCLES"):
            extra["wait_frac"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in out:
            extra["l2_hit"] = out["TCC_HIT_sum"] / max(out["TCC_HIT_sum"] + out["TCC_MISS_sum"], 1)
        print("%-28s calls=%d " % (k[:28], n) + " ".join("%s=%.3g" % (a.replace("SQ_", "").replace("_sum", ""), b) for a, b in sorted(out.items()))
              + " | " + " ".join("%s=%.3f" % kv for kv in extra.items()))
