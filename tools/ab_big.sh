#!/bin/bash
# GPU box: the 4K x 16 spp configurations (burst form) for each build/variants/<name>.so ("base" = the in-tree library)
mkdir -p gpurun_out/ab
for name in "$@"; do
  lib=$PWD/build/variants/$name.so
  [ "$name" = base ] && lib=$PWD/ray-tracer_amd/libmi355rt.so
  for cfg in "lucy:--model lucy" "thai:--model thai --material mirror-floor"; do
    tag=${cfg%%:*}; extra=${cfg#*:}
    MI355RT_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps ${BENCH_STEPS:-6} --warmup 2 --width 3840 --height 2160 --spp 16 $extra > gpurun_out/ab/${name}_$tag.json 2> gpurun_out/ab/${name}_$tag.err || { echo "$name $tag FAILED"; tail -3 gpurun_out/ab/${name}_$tag.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/${name}_$tag.json"))
k=d["roofline"]["frame"]["kernels"]
print("%-10s %-5s frame %.2f ms  " % ("$name", "$tag", d["ms_per_step"]) + "  ".join("%s %.2f" % (n, v["ms_per_frame"]) for n, v in k.items()))
PY
  done
done
