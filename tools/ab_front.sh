#!/bin/bash
# Dev tool: A/B two builds of the library on the front end (tools/front_end_times.py) -- expects endosurf_amd/lib/variant_{A,B}.so
L=endosurf_amd/lib
for r in 1 2; do for v in ${VARIANTS:-A B}; do cp $L/variant_$v.so $L/libendosurf_hip.so; python tools/front_end_times.py 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.rindex('{\n \"march'):]) if False else None
" ; python - <<P
import json
d=json.load(open('gpurun_out/front_end_times.json'))
print('$v', {k:round(v,4) for k,v in d.items() if isinstance(v,float) and ('query16' in k or 'chain' in k or 'concurrent_ms' in k)})
P
done; done
