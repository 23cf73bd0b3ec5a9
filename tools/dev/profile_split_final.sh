set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out profiles
python bench.py --split-precision --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/r03_split.json
python bench.py --config 3 --split-precision --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/r03_cfg3_split.json
python bench.py --config 4 --split-precision --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/r03_cfg4_split.json
bash tools/dev/profile_mode.sh r03_split --split-precision > gpurun_out/prof_split.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_split_stats -o train -- python $R/bench.py --no-cpu-baseline --headline-only --split-precision --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
cp $(find gpurun_out/prof_split_stats -name "*kernel_stats.csv" | head -1) profiles/r03_split_kernel_stats.csv
cp endosurf_amd/lib/libendosurf_hip.so endosurf_amd/lib/variant_B.so
VARIANTS="B" bash tools/dev/pmc_probe.sh > profiles/r03_wgrad_x3_probe_pmc.txt 2>&1
export PYTHONPATH=$R; for f in randn relu zeros; do python tools/dev/wgrad_x3_probe.py 1654784 $f 2>&1 | tail -2; done > profiles/r03_wgrad_x3_probe.txt
cp profiles/r03_split.json profiles/r03_cfg3_split.json profiles/r03_cfg4_split.json profiles/r03_split_pmc_summary.json profiles/r03_split_kernel_stats.csv profiles/r03_wgrad_x3_probe_pmc.txt profiles/r03_wgrad_x3_probe.txt gpurun_out/
for f in split cfg3_split cfg4_split; do python -c "
import json; b=json.load(open('profiles/r03_$f.json')); print('$f', round(b['ms_per_step'],3), round(b['value']))"; done
head -8 profiles/r03_split_kernel_stats.csv | cut -c1-160; cat profiles/r03_wgrad_x3_probe.txt
