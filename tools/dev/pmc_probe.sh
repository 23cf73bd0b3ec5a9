#!/bin/bash
# Dev tool: SQ counters of the bare split-precision weight-gradient GEMM (tools/dev/wgrad_x3_probe.py) for library variants.
# usage: VARIANTS="B C E" bash tools/dev/pmc_probe.sh   -> gpurun_out/pmc_probe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; L=$R/endosurf_amd/lib; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-B}; do
  cp $L/variant_$v.so $L/libendosurf_hip.so
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
              "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pp -o p -- python $R/tools/dev/wgrad_x3_probe.py > /dev/null 2>&1
    python - "$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pp/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for x in csv.DictReader(open(f[0])):
    if 'k_wgrad_x3' in x['Kernel_Name']: acc[x['Counter_Name']].append(float(x['Counter_Value']))
print(sys.argv[1], {k: round(sum(v) / len(v)) for k, v in acc.items()}, len(next(iter(acc.values()), [])))
PY
  done
done
cp $L/variant_B.so $L/libendosurf_hip.so
