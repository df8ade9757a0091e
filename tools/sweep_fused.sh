#!/bin/bash
# the fused walker on the bench frame and on one rank's share of it: one line per setting
set -u
out=gpurun_out/sweep_fused.txt
: > $out
run() {
   echo "== $*" >> $out
   env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --frames-in-flight 1 --steps 20 --warmup 3 $EXTRA 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], {k: v['ms_per_frame'] for k, v in j['roofline']['frame']['kernels'].items()})" >> $out || exit 1
}
EXTRA=""
run MI355RT_FUSE_SHADOW=1
run MI355RT_AGE_MIN_FUSED=8
run MI355RT_AGE_MIN_FUSED=16
run MI355RT_AGE_MIN_FUSED=32
run MI355RT_AGE_MIN_FUSED=64
EXTRA="--emulate-world 8"
run MI355RT_FUSE_SHADOW=1
run MI355RT_FUSE_SHADOW=1 MI355RT_AGE_MIN_FUSED=8
run MI355RT_FUSE_SHADOW=1 MI355RT_AGE_MIN_FUSED=32
run MI355RT_AGE_MIN_FUSED=8
EXTRA="--emulate-world 4"
run MI355RT_FUSE_SHADOW=1
run MI355RT_FUSE_SHADOW=1 MI355RT_AGE_MIN_FUSED=16
cat $out
