for b in 2 3 4 6; do MI355RT_BLOCKS_PER_CU=$b timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --emulate-world 8 --emulate-rank 0 > gpurun_out/sweep/bpc$b.json 2>/dev/null; python -c "
import json; e=json.load(open('gpurun_out/sweep/bpc$b.json')); print('blocks/CU $b: 1/8 share', e['max_ms'], 'sustained', e.get('max_sustained_ms'))"; done
