#!/bin/bash
# GPU box: parity tests, then the bench line (no CPU baseline).  usage: tools/gpu_check.sh [tag]
tag=${1:-check}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
tail -4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 240 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], {k:v["ms_per_frame"] for k,v in d["roofline"]["frame"]["kernels"].items()})
PY
