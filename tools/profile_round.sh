#!/bin/bash
# Collect the profiles/ artefacts of a round on the GPU box:  bash tools/profile_round.sh <tag>
#   <tag>_train_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 2`
#   <tag>_pmc_summary.json         three separate --pmc passes (SQ counters | FETCH_SIZE | WRITE_SIZE), tools/pmc_summary.py
#   <tag>_bench.json               the bench.py line of the same build
#   <tag>_plain_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the REFERENCE trainer's own loop through the drop-in
#                                  (`python bench.py --refseq-only plain`: renderer(rays) -> errorondepth -> surface_neighbour_error, torch Adam, loss.item())
#   <tag>_plain_bench.json         its record (ms per step, host issue, library calls per step)
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O $R/profiles
cd /tmp && export TMPDIR=/tmp
# --headline-only: every launch of the profiled process belongs to the headline workload, so that rocprof's per-symbol averages are
# directly comparable with roofline.avg_launch_ms of the bench line
B="python $R/bench.py --no-cpu-baseline --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o train -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcA -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcC -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -o plain -- python $R/bench.py --refseq-only plain --steps 10 --warmup 3 > /dev/null 2>&1
cd $R
cp $(find $O/plain -name "*kernel_stats.csv" | head -1) profiles/${TAG}_plain_kernel_stats.csv
python bench.py --refseq-only plain --steps 30 --warmup 5 | tail -1 > profiles/${TAG}_plain_bench.json
python bench.py --refseq-only logging --steps 30 --warmup 5 | tail -1 > profiles/${TAG}_plain_logging_bench.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}_train_kernel_stats.csv
python tools/pmc_summary.py $TAG $(dirname $(find $O/pmcA -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/pmcB -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/pmcC -name "*counter_collection.csv" | head -1)) | head -14
python bench.py --steps 30 --warmup 5 | tail -1 > profiles/${TAG}_bench.json
cp profiles/${TAG}_train_kernel_stats.csv profiles/${TAG}_pmc_summary.json profiles/${TAG}_bench.json profiles/${TAG}_plain_kernel_stats.csv profiles/${TAG}_plain_bench.json profiles/${TAG}_plain_logging_bench.json gpurun_out/
head -c 600 profiles/${TAG}_bench.json
rm -rf $O
