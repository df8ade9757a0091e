#!/bin/bash
# GPU box: every rank's share of an N-rank frame rendered in turn on one GPU (bench.py --emulate-world): per-rank times, their
# max, and the same with two frames in flight.  usage: tools/shares.sh <tag> [bench.py workload arguments ...]; SHARES="2 4 8"
tag=${1:-dragon}; shift
mkdir -p gpurun_out/shares
for n in ${SHARES:-1 2 4 8}; do
  if [ $n = 1 ]; then
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-10} --warmup 2 "$@" > gpurun_out/shares/${tag}_n1.json 2> gpurun_out/shares/${tag}_n1.err || { echo FAIL $n; tail -3 gpurun_out/shares/${tag}_n1.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/shares/${tag}_n1.json"))
print("$tag N=1  ms %.3f  sustained %s" % (d["step_gpu_ms"]["render"]["median"], (d.get("pipelined") or {}).get("ms_per_step")))
PY
    continue
  fi
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras --emulate-world $n --steps ${STEPS:-10} --warmup 2 "$@" > gpurun_out/shares/${tag}_n$n.json 2> gpurun_out/shares/${tag}_n$n.err || { echo FAIL $n; tail -3 gpurun_out/shares/${tag}_n$n.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/shares/${tag}_n$n.json"))
print("$tag N=$n  max %.3f mean %.3f  sustained max %s  ranks %s" % (d["max_ms"], d["mean_ms"], d.get("max_sustained_ms"), d["rank_ms"]))
PY
done
