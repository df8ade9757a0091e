#!/bin/bash
# Here (not on the GPU box): copies the summaries tools/profile_round.sh left in gpurun_out/prof_<tag>/ into profiles/<tag>_*
# and merges its counters into profiles/hbm_traffic.json (what bench.py's roofline line reads).
# usage: tools/keep_profile.sh <tag>      e.g. tools/keep_profile.sh r03_dragon
tag=${1:?tag}; R=$(cd $(dirname $0)/.. && pwd); src=$R/gpurun_out/prof_$tag
[ -f $src/hbm_traffic.json ] || { echo "$src/hbm_traffic.json missing"; exit 1; }
for f in bench.json bench_under_rocprof.json hbm_traffic.txt kernel_stats.csv sq_counters.txt; do cp $src/$f $R/profiles/${tag}_$f; done
python3 - "$src/hbm_traffic.json" "$R/profiles/hbm_traffic.json" <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
try:
    old = json.load(open(sys.argv[2]))
except OSError:
    old = {}
old.update(new)
json.dump(old, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print("merged", sorted(new))
PY
