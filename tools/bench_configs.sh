#!/bin/bash
# The other bench.py lines of a round (GPU box):  bash tools/bench_configs.sh <tag>
#   profiles/<tag>_cfg3.json / _cfg4.json / _cfg5_frame.json / _forward.json  and the OPT-IN split-precision lines _split.json / _forward_split.json / _cfg5_frame_split.json
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p profiles gpurun_out
python bench.py --config 3 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_cfg3.json
python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_cfg4.json
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_cfg5_frame.json
python bench.py --mode forward --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_forward.json
python bench.py --split-precision --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_split.json
python bench.py --mode forward --split-precision --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_forward_split.json
python bench.py --config 5 --split-precision --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_cfg5_frame_split.json
python bench.py --config 3 --split-precision --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_cfg3_split.json
python bench.py --config 4 --split-precision --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${TAG}_cfg4_split.json
cp profiles/${TAG}_cfg3.json profiles/${TAG}_cfg4.json profiles/${TAG}_cfg5_frame.json profiles/${TAG}_forward.json profiles/${TAG}_split.json profiles/${TAG}_forward_split.json profiles/${TAG}_cfg5_frame_split.json profiles/${TAG}_cfg3_split.json profiles/${TAG}_cfg4_split.json gpurun_out/
for f in cfg3 cfg4 cfg5_frame forward split forward_split cfg5_frame_split cfg3_split cfg4_split; do python - <<PY
import json
b=json.load(open("profiles/${TAG}_$f.json"))
r=b.get("roofline") or {}
print("$f", round(b["ms_per_step"],3), "ms", round(b["value"]), b["unit"], "| dominant", r.get("kernel"), round(r.get("frac",0),3), "| e2e", round((r.get("end_to_end") or {}).get("frac",0),3),
      "| exit", (b["config"].get("with_early_exit") or {}).get("ms_per_step"))
PY
done
