#!/bin/bash
mkdir -p gpurun_out/bpc
for n in 8 4; do
for b in 0 2 3 4 5 6; do
  envs="X=0"; [ $b != 0 ] && envs="MI355RT_BLOCKS_PER_CU=$b"
  env $envs timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --frames-in-flight 1 --emulate-world $n --steps 20 --warmup 3 > gpurun_out/bpc/n${n}_$b.json 2> gpurun_out/bpc/n${n}_$b.err || { echo FAIL; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bpc/n${n}_$b.json"))
print("N=$n bpc=$b ms %.3f %s" % (d["ms_per_step"], {k:round(x["ms_per_frame"],3) for k,x in d["roofline"]["frame"]["kernels"].items()}))
PY
done
done
