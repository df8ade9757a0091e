#!/bin/bash
# GPU box: the 4K x 16 spp configurations at other pass sizes (MI355RT_PASS_SLOTS_LOG2)
mkdir -p gpurun_out/pass
for model in "thai --material mirror-floor" "lucy"; do
  name=${model%% *}
  for l in 24 25 26 27; do
    MI355RT_PASS_SLOTS_LOG2=$l timeout -k 10 280 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --model $model --width 3840 --height 2160 --spp 16 > gpurun_out/pass/${name}_$l.json 2> gpurun_out/pass/${name}_$l.err || { echo FAIL $name $l; tail -3 gpurun_out/pass/${name}_$l.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/pass/${name}_$l.json"))
print("$name log2 $l  ms %.3f  pipelined %s  kernels %s" % (d["ms_per_step"], (d.get("pipelined") or {}).get("ms_per_step"), {k:round(v["ms_per_frame"],2) for k,v in d["roofline"]["frame"]["kernels"].items()}))
PY
  done
done
