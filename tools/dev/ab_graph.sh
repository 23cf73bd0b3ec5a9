for r in 1 2 3; do
 for v in "" "--graph"; do
  python bench.py --no-cpu-baseline --headline-only --steps 40 --warmup 8 $v 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', round(b['ms_per_step'],3))"
 done
done
