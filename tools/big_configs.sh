#!/bin/bash
mkdir -p gpurun_out/blk
for v in "$@"; do
for model in "thai --material mirror-floor" "lucy"; do
  name=${model%% *}
  MI355RT_LIB=$PWD/build/variants/$v.so timeout -k 10 280 python bench.py --no-cpu-baseline --no-extras --frames-in-flight 1 --steps 5 --warmup 2 --model $model --width 3840 --height 2160 --spp 16 > gpurun_out/blk/${name}_$v.json 2> gpurun_out/blk/${name}_$v.err || { echo FAIL $name $v; tail -3 gpurun_out/blk/${name}_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/blk/${name}_$v.json"))
print("$v $name ms %.3f  kernels %s" % (d["ms_per_step"], {k:round(x["ms_per_frame"],2) for k,x in d["roofline"]["frame"]["kernels"].items()}))
PY
done
done
