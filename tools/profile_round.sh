#!/bin/bash
# GPU box: the measurements DESIGN.md section 5 quotes, into gpurun_out/prof_<tag>/ (copy the summaries into profiles/).
# usage: tools/profile_round.sh <tag> <traffic key> [bench.py workload arguments ...]
#   e.g. tools/profile_round.sh r02_dragon dragon_1920x1080_4spp_n1
#        tools/profile_round.sh r02_lucy lucy_3840x2160_16spp_n1 --model lucy --width 3840 --height 2160 --spp 16
# One rocprofv3 run per counter pass (PMC slots: 8 SQ, 4 TCC; FETCH_SIZE takes 3 of them, WRITE_SIZE 2).
tag=${1:-r03}; key=${2:-dragon_1920x1080_4spp_n1}; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
steps=${PROFILE_STEPS:-20}
timeout -k 10 400 python3 $R/bench.py --steps $steps --warmup 3 "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extras --frames-in-flight 1 "$@" > $out/bench_under_rocprof.json 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
            "sq:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -s KILL 200 rocprofv3 --pmc $ctrs -d $out/$name --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --frames-in-flight 1 "$@" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
  find $out/$name -name "*agent_info.csv" -delete
done
python3 $R/tools/hbm_traffic.py $key $out/fetch $out/write $out/tcc $out/sq --json $out/hbm_traffic.json > $out/hbm_traffic.txt || exit 1
python3 $R/tools/sq_summary.py $out/sq "$*" > $out/sq_counters.txt || exit 1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
# the raw per-dispatch tables are large: only the summaries travel back
rm -rf $out/fetch $out/write $out/tcc $out/sq $out/stats
cat $out/hbm_traffic.txt $out/sq_counters.txt
