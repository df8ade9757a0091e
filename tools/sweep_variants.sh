#!/bin/bash
# GPU box: bench each build/variants/<name>.so given as name:blocks_per_cu[:ENV=VAL,...]
mkdir -p gpurun_out
for spec in "$@"; do
  IFS=: read name bpc envs <<< "$spec"
  lib=$PWD/build/variants/$name.so
  [ "$name" = base ] && lib=$PWD/ray-tracer_amd/libmi355rt.so
  extra=$(echo "$envs" | tr ',' ' ')
  env MI355RT_LIB=$lib MI355RT_BLOCKS_PER_CU=$bpc $extra timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --frames-in-flight 1 --steps ${BENCH_STEPS:-10} --warmup 2 $BENCH_EXTRA > gpurun_out/sweep_$name.json 2> gpurun_out/sweep_$name.err || { echo "$spec FAILED"; tail -3 gpurun_out/sweep_$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_$name.json"))
k=d["roofline"]["frame"]["kernels"]
print("%-28s value %7.1f ms %.3f primary %.3f shadow %.3f" % ("$spec", d["value"], d["ms_per_step"], k["walk_primary"]["ms_per_frame"], k.get("walk_shadow", {"ms_per_frame": 0.0})["ms_per_frame"]))
PY
done
