#!/bin/bash
# Dev tool: build instrumented / ablated variants of csrc/query_x3r.hip (here, no GPU needed), then `bash tools/dev/xr_profile.sh run` on the GPU box.
cd "$(dirname "$0")/.."
L=endosurf_amd/lib; B=endosurf_amd/build; S=endosurf_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -Wno-unused-variable -Wno-pass-failed"
if [ "$1" != run ]; then
  python -m endosurf_amd.build >/dev/null || exit 1
  for v in "prof1:-DXR_PROFILE=1" "prof2:-DXR_PROFILE=2" "nopin:-DXR_PROFILE=1 -DXR_NO_PIN" "novalu:-DXR_PROFILE=1 -DXR_NO_VALU" "nodma:-DXR_PROFILE=1 -DXR_NO_DMA" "nobar:-DXR_PROFILE=1 -DXR_NO_BARRIER" "nodmabar:-DXR_PROFILE=1 -DXR_NO_DMA -DXR_NO_BARRIER" "nothing:-DXR_PROFILE=1 -DXR_NO_DMA -DXR_NO_BARRIER -DXR_NO_VALU"; do
    name=${v%%:*}; defs=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS $defs -c $S/query_x3r.hip -o /tmp/xr_$name.o || exit 1
    objs=$(ls $B/*.o | grep -v query_x3r.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/xr_$name.o -o $L/xr_$name.so || exit 1
    echo built $L/xr_$name.so
  done
  exit 0
fi
cp $L/libendosurf_hip.so /tmp/orig.so
for name in ${VARIANTS:-prof1 prof2 novalu nodma nobar nodmabar nothing}; do
  cp $L/xr_$name.so $L/libendosurf_hip.so
  echo "== $name"; python tools/dev/xr_profile.py
done
cp /tmp/orig.so $L/libendosurf_hip.so
