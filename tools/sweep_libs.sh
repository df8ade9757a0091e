#!/bin/bash
# GPU box: the bench frame and rank 0's 1/8 share of it for each build/variants/<name>.so ("base" = the in-tree library)
mkdir -p gpurun_out/sweep
for name in "$@"; do
  lib=$PWD/build/variants/$name.so
  [ "$name" = base ] && lib=$PWD/ray-tracer_amd/libmi355rt.so
  MI355RT_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps ${BENCH_STEPS:-10} --warmup 2 $BENCH_EXTRA > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.err || { echo "$name FAILED"; tail -3 gpurun_out/sweep/$name.err; exit 1; }
  MI355RT_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps ${BENCH_STEPS:-10} --warmup 2 --emulate-world 8 --emulate-rank 0 $BENCH_EXTRA > gpurun_out/sweep/${name}_s8.json 2> gpurun_out/sweep/${name}_s8.err || { echo "$name share FAILED"; tail -3 gpurun_out/sweep/${name}_s8.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep/$name.json")); e=json.load(open("gpurun_out/sweep/${name}_s8.json"))
k=d["roofline"]["frame"]["kernels"]
print("%-12s frame %.3f ms (median %.3f) walker %.3f  2 in flight %.3f | 1/8 share %.3f sustained %.3f" % ("$name", d["ms_per_step"], d["step_gpu_ms"]["render"]["median"], k["walk_primary"]["ms_per_frame"], d["pipelined"]["ms_per_step"], e["max_ms"], e.get("max_sustained_ms") or 0))
PY
done
