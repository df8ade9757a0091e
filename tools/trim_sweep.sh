#!/bin/bash
# GPU box: launch trimming settings on the 4K x 16 spp configurations (env assignments, comma separated, one run each)
mkdir -p gpurun_out/trim
for model in "thai --material mirror-floor" "lucy"; do
  name=${model%% *}
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $(echo "$envs" | tr ',' ' ') timeout -k 10 280 python bench.py --no-cpu-baseline --no-extras --frames-in-flight 1 --steps 5 --warmup 2 --model $model --width 3840 --height 2160 --spp 16 > gpurun_out/trim/${name}_$i.json 2> gpurun_out/trim/${name}_$i.err || { echo FAIL $name $envs; tail -3 gpurun_out/trim/${name}_$i.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/trim/${name}_$i.json"))
print("$name %-40s ms %.3f  %s" % ("$envs", d["ms_per_step"], {k:round(x["ms_per_frame"],2) for k,x in d["roofline"]["frame"]["kernels"].items() if k.startswith("walk")}))
PY
  done
done
