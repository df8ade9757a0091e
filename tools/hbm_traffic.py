#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of bench.py into per-kernel HBM bytes per launch.

usage: hbm_traffic.py KEY FETCH_DIR WRITE_DIR [TCC_DIR [SQ_DIR]] [--json PATH] > summary; also merges the result into
PATH (default profiles/hbm_traffic.json) under KEY (bench.py reads roofline.traffic, valu_busy, lane_util and limiter from
profiles/hbm_traffic.json -- and only while the kernel sources' hash stored with them is the hash of the tree it runs from).

Rules (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE are collected in
separate passes (3 + 2 TCC slots), both are in KiB, and on gfx950 FETCH_SIZE tallies each 128-byte
memory-side read request as 64 bytes, so it is doubled; WRITE_SIZE is taken as read.  The walkers gather
80-byte node records and 48-byte group faces (16 bytes per lane, uncoalesced): the guide calls widths other
than a coalesced stream uncalibrated, so the doubled figure is an upper bound for them."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"walk_kernel<0, 0": "walk_primary", "walk_kernel<0, 3": "walk_primary", "walk_kernel<0, 2": "walk_shadow", "walk_kernel<0, 1": "walk_shadow",
         "generate_kernel": "generate", "shade_kernel<0,": "shade", "shade_kernel<1,": "shade", "phong_kernel": "phong", "resolve_kernel": "resolve"}


def per_kernel(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        for pat, name in NAMES.items():
            if pat in row["Kernel_Name"]:
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[name].add(row["Dispatch_Id"])
    return {k: {c: v / len(calls[k]) for c, v in cs.items()} for k, cs in acc.items()}, {k: len(v) for k, v in calls.items()}


def main():
    argv = sys.argv[1:]
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if "--json" in argv:
        i = argv.index("--json")
        path = argv[i + 1]
        del argv[i:i + 2]
    key, fetch_dir, write_dir = argv[:3]
    fetch, n = per_kernel(fetch_dir)
    write, _ = per_kernel(write_dir)
    tcc = per_kernel(argv[3])[0] if len(argv) > 3 else {}
    sq = per_kernel(argv[4])[0] if len(argv) > 4 else {}
    out = {}
    for k in fetch:
        rd = 2.0 * fetch[k]["FETCH_SIZE"] * 1024.0
        wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        out[k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "fetch_size_kib_raw": fetch[k]["FETCH_SIZE"],
                  "launches_sampled": n[k]}
        t = tcc.get(k)
        if t:
            out[k]["tcc_ea0_rdreq"] = t.get("TCC_EA0_RDREQ_sum")
            out[k]["tcc_ea0_rdreq_32b"] = t.get("TCC_EA0_RDREQ_32B_sum")
            hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
            out[k]["l2_hit_rate"] = round(hit / max(hit + miss, 1.0), 4)
        q = sq.get(k)
        if q:   # as tools/sq_summary.py: SIMD VALU busy = 4 cycles x SQ_INSTS_VALU / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 shader engines)
            out[k]["valu_busy"] = round(4.0 * q["SQ_INSTS_VALU"] / 1024.0 / max(q["SQ_BUSY_CYCLES"] / 32.0, 1.0), 3)
            out[k]["lane_util"] = round(q["SQ_THREAD_CYCLES_VALU"] / max(q["SQ_ACTIVE_INST_VALU"] * 64.0, 1.0), 3)
            out[k]["limiter"] = "valu-issue" if out[k]["valu_busy"] >= 0.8 else ("hbm" if (rd + wr) > 0 and out[k].get("l2_hit_rate", 1.0) < 0.5 else "latency")
        print("%-14s launches=%-3d read=%8.1f MB (2 x FETCH_SIZE)  write=%8.1f MB  total=%8.1f MB%s"
              % (k, n[k], rd / 1e6, wr / 1e6, (rd + wr) / 1e6, "  L2 hit %.3f" % out[k]["l2_hit_rate"] if t else ""))
    allj = json.load(open(path)) if os.path.exists(path) else {}
    import hashlib
    import time
    h = hashlib.sha1()
    for name in ("trace_kernels.hip", "device_util.h", "layout.h", "walk_args.h"):   # bench.py kernel_sources_hash()
        h.update(open(os.path.join(ROOT, "ray-tracer_amd", "csrc", name), "rb").read())
    out["kernel_sources_sha1"] = h.hexdigest()
    out["taken"] = time.strftime("%Y-%m-%d")
    allj[key] = out
    json.dump(allj, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
