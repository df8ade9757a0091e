#!/bin/bash
# one SQ counter pass of the bench (GPU box): tools/sq_pass.sh <outdir-name> [env assignments...]
name=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$name
rm -rf $out; mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES -d $out --output-format csv -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --frames-in-flight 1 > $out/bench.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
find $out -name "*agent_info.csv" -delete
python3 $R/tools/sq_summary.py $out | grep walk_kernel
