set -x
timeout 900 python -m pytest tests/test_gpu_query_x3.py -m gpu -q -x 2>&1 | tail -15
for v in 0 1 0 1; do
ES_X3R=$v timeout 300 python bench.py --no-cpu-baseline --split-precision --headline-only --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('X3R=$v', round(b['ms_per_step'],3), round(b['value']), {s['kernel']:(s['ms_per_step'],s['tflops']) for s in b['kernel_symbols'] if 'x3' in s['kernel'] or 'query' in s['kernel']})"
done
