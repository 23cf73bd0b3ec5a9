#!/bin/bash
# Dev tool: A/B two builds of the library on the same GPU box.  usage: bash tools/ab.sh [rounds]
# expects endosurf_amd/lib/variant_A.so and variant_B.so; alternates them and prints ms per step of bench.py
L=endosurf_amd/lib
for r in $(seq ${1:-3}); do
  for v in A B; do
    cp $L/variant_$v.so $L/libendosurf_hip.so
    python bench.py --no-cpu-baseline --steps 30 --warmup 5 | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(b['ms_per_step'],3), round(b['value']), round(b['roofline']['frac'],3), {k:v for k,v in b['kernel_ms_per_step'].items() if 'sdf' in k and 'query' not in k})"
  done
done
