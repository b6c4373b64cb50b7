#!/bin/bash
# Dev tool: regenerate the round's rocprof / PMC summaries under gpurun_out/prof/ (run on the GPU box from the repo root;
# copy the r03_* files into profiles/ afterwards).  One bench.py command per pass, all with the same saved tuning.
# usage: tools/refresh_profiles.sh [math=bf16x6] [batch=32] [tag=r03]
set -u
MATH=${1:-bf16x6}; BATCH=${2:-32}; TAG=${3:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/prof; mkdir -p $O
B="--math $MATH --batch $BATCH --no-secondary --no-cpu-baseline --tuning $O/${TAG}_bench_tuning.json"
rm -f $O/${TAG}_bench_tuning.json
python bench.py --steps 1 --warmup 0 --iters 4 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$R/$O/stats" -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 $B > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.err
cp $O/stats/s_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv; rm -rf $O/stats
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$O/pmcA" -o a -- python3 bench.py --steps 1 --warmup 0 --iters 4 $B > $O/pmcA.log 2> $O/pmcA.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$O/pmcB" -o b -- python3 bench.py --steps 1 --warmup 0 --iters 4 $B > $O/pmcB.log 2> $O/pmcB.err
python tools/pmc_traffic.py $O/pmcA $O/pmcB $BATCH $O/${TAG}_pmc_hbm_traffic.json $O/pmcA.log $MATH 16 > $O/${TAG}_pmc_hbm_traffic.txt
rm -rf $O/pmcA $O/pmcB
echo "traffic passes done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$R/$O/pm" -o p -- python3 bench.py --steps 1 --warmup 0 --iters 6 $B > $O/pm.log 2>&1
python tools/pmc_mfma.py $O/pm $O/${TAG}_pmc_mfma_util.json > $O/${TAG}_pmc_mfma_util.txt; rm -rf $O/pm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/$O/pw" -o p -- python3 bench.py --steps 1 --warmup 0 --iters 6 $B > $O/pw.log 2>&1
python tools/pmc_waves.py $O/pw $O/${TAG}_pmc_wave_states.json > $O/${TAG}_pmc_wave_states.txt; rm -rf $O/pw
echo "pmc passes done"
# (per-layer views map kernels to layers by launch order: the side stream is switched off for this pass only)
export IVF_OVERLAP=0
rocprofv3 --kernel-trace -d "$R/$O/ktrace" -o k --output-format csv -- python3 bench.py --steps 1 --warmup 0 --iters 12 $B > $O/ktrace.log 2>&1
unset IVF_OVERLAP
python tools/iter_breakdown.py $O/ktrace/k_kernel_trace.csv > $O/${TAG}_iteration_breakdown.txt
python tools/trace_convs.py $O/ktrace/k_kernel_trace.csv $BATCH 40 $MATH > $O/${TAG}_per_layer_iteration_B${BATCH}.txt 2>&1
rm -rf $O/ktrace
tail -c 600 $O/${TAG}_bench_under_rocprof.json; echo; head -30 $O/${TAG}_iteration_breakdown.txt
