#!/bin/bash
# GPU box: one launch for primary + shadow rays (MODE_FUSED) against two, on the bench frame and on rank 0's share of
# an N-rank run ("Shadow rays inside the primary launch").  One line per setting.
set -u
out=gpurun_out/sweep_fused.txt
: > $out
run() {
   echo "== $EXTRA $*" >> $out
   env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --frames-in-flight 1 --steps 20 --warmup 3 $EXTRA 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], {k: v['ms_per_frame'] for k, v in j['roofline']['frame']['kernels'].items()})" >> $out || exit 1
}
for w in 1 2 4 8; do
   EXTRA="--emulate-world $w"
   run MI355RT_FUSE_SHADOW=1
   run MI355RT_FUSE_SHADOW=0
done
cat $out
