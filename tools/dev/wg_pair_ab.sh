#!/bin/bash
# A/B of the k-block pairing of the weight-gradient tasks (DEAD_ENDS C5): run with a library built -DES_WG_NOPAIR and with the release build.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/wgpair_$1
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --headline-only"
$B --steps 20 --warmup 4 | tail -1 > $R/gpurun_out/wgpair_$1.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python - <<PY
import csv, glob, json
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_wgrad" in r["Kernel_Name"]:
        k = r["Kernel_Name"][:24]
        a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
d = json.load(open("$R/gpurun_out/wgpair_$1.json"))
print("$1", "ms_per_step", round(d["ms_per_step"], 3), {k: v for k, v in d["kernel_ms_per_step"].items() if "wgrad" in k})
print({k: round(v[0] / v[1] * 64 / 1e9, 3) for k, v in acc.items()}, "GB fetched per launch (FETCH_SIZE x 64 B)")
PY
rm -rf $O
