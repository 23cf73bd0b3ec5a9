#!/bin/bash
# Dev tool: A/B two builds of the library on the same GPU box.  usage: bash tools/dev/ab.sh [rounds] [extra bench flags]
# expects endosurf_amd/lib/variant_A.so and variant_B.so; alternates them and prints ms per step of bench.py
L=endosurf_amd/lib
R=${1:-3}; shift
for r in $(seq $R); do
  for v in ${VARIANTS:-A B}; do
    cp $L/variant_$v.so $L/libendosurf_hip.so
    python bench.py --no-cpu-baseline --headline-only --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(b['ms_per_step'],3), round(b['value']), {s['kernel']:s['ms_per_step'] for s in b['kernel_symbols'] if True})"
  done
done
