#!/bin/bash
# Dev tool: A/B two builds of the library in split-precision mode on the same GPU box.  usage: bash tools/dev/ab_split.sh [rounds]
L=endosurf_amd/lib
for r in $(seq ${1:-2}); do
  for v in ${VARIANTS:-A B}; do
    cp $L/variant_$v.so $L/libendosurf_hip.so
    python bench.py --no-cpu-baseline --split-precision --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(b['ms_per_step'],3), round(b['value']), {s['kernel']:(s['ms_per_step'],s['tflops']) for s in b['kernel_symbols'] if 'x3' in s['kernel']})"
  done
done
