#!/bin/bash
# one pass of texture-addresser / L1 counters over the bench (GPU box): tools/ta_pass.sh <outdir-name> "<counters>" [env assignments...]
name=$1; counters=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$name
rm -rf $out; mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $counters -d $out --output-format csv -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --frames-in-flight 1 > $out/bench.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
find $out -name "*agent_info.csv" -delete
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "walk_kernel" not in k: continue
        k = k.split("(")[0][-40:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == list(acc[k].keys())[0]: n[k] += 1
for k, d in acc.items():
    print(k, "launches", n[k], " ".join("%s=%.4g" % (c, v / max(n[k], 1)) for c, v in d.items()))
PY
