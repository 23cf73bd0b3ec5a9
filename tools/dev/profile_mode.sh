#!/bin/bash
# PMC summary of any bench.py mode on the GPU box:  bash tools/dev/profile_mode.sh <tag> <bench.py arguments ...>
#   profiles/<tag>_pmc_summary.json   three separate --pmc passes (SQ counters | FETCH_SIZE | WRITE_SIZE), tools/pmc_summary.py
# e.g.  bash tools/dev/profile_mode.sh r02e_forward_split --mode forward --split-precision
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O $R/profiles
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcA -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcC -o p -- $B > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $TAG $(dirname $(find $O/pmcA -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/pmcB -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/pmcC -name "*counter_collection.csv" | head -1)) | head -12
cp profiles/${TAG}_pmc_summary.json gpurun_out/
