cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_split
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o p -- python $R/bench.py --no-cpu-baseline --split-precision --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $O/b -o p -- python $R/bench.py --no-cpu-baseline --split-precision --steps 2 --warmup 1 > /dev/null 2>&1
python - <<PY
import csv, collections, glob, statistics as st
for d in ("a","b"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print("no file", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(f[0])):
        k = x["Kernel_Name"].split("(")[0].replace("void ","").replace("es::","")
        if "x3" in k or "k_query_sdf<" in k:
            acc[(k, x["Grid_Size"])][x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, c in acc.items():
        print(d, k, {n: round(st.mean(v)) for n, v in c.items()})
PY
