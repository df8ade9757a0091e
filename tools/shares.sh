#!/bin/bash
# shares of the bench frame: one frame at a time and two in flight, fused / unfused
mkdir -p gpurun_out/shares
for n in ${SHARES:-2 4 8}; do
  for f in d 0 1; do
    envs=""; [ $f != d ] && envs="MI355RT_FUSE_SHADOW=$f"
    env $envs timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --emulate-world $n --steps 20 --warmup 3 > gpurun_out/shares/n${n}_f$f.json 2> gpurun_out/shares/n${n}_f$f.err || { echo FAIL $n $f; tail -3 gpurun_out/shares/n${n}_f$f.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/shares/n${n}_f$f.json"))
p=d.get("pipelined") or {}
print("N=$n fuse=$f  ms %.3f  pipelined %s" % (d["ms_per_step"], p.get("ms_per_step")))
PY
  done
done
