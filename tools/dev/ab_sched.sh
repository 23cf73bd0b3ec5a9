#!/bin/bash
L=endosurf_amd/lib
for v in ${VARIANTS:-A B}; do cp $L/variant_$v.so $L/libendosurf_hip.so; echo -n "$v "; python tools/dev/front_sched.py 2>/dev/null | tail -1; done
