#!/bin/bash
# Dev tool: A/B builds of the library on the front end of a training step (tools/dev/front_end_times.py) -- expects
# endosurf_amd/lib/variant_<X>.so for every X in $VARIANTS (default "A B"); prints the chain / launch timings of each, twice
L=endosurf_amd/lib
for r in 1 2; do
  for v in ${VARIANTS:-A B}; do
    cp $L/variant_$v.so $L/libendosurf_hip.so
    python tools/dev/front_end_times.py > /dev/null 2>&1
    python - <<P
import json
d = json.load(open('gpurun_out/front_end_times.json'))
print('$v', {k: round(x, 4) for k, x in d.items() if isinstance(x, float) and ('query16' in k or 'chain' in k or 'concurrent_ms' in k)})
P
  done
done
