#!/bin/bash
# GPU box: the measurements DESIGN.md section 5 quotes, into gpurun_out/prof_<tag>/ (copy the summaries into profiles/).
# usage: tools/profile_round.sh <tag>.  One rocprofv3 run per counter pass (PMC slots: 8 SQ, 4 TCC; FETCH_SIZE takes 3, WRITE_SIZE 2).
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --frames-in-flight 1 > $out/bench_under_rocprof.json 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
            "sq:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctrs -d $out/$name --output-format csv -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --frames-in-flight 1 > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
  find $out/$name -name "*agent_info.csv" -delete
done
find $out/stats -name "*kernel_trace.csv" -delete
find $out/stats -name "*agent_info.csv" -delete
ls -R $out | head -40
