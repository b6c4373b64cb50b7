#!/bin/bash
# Dev tool: regenerate the round's rocprof / PMC summaries under gpurun_out/ (run on the GPU box from the repo root;
# copy the r02_* files into profiles/ afterwards).  One bench.py command per pass, all with the same saved tuning.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
B="--no-secondary --no-cpu-baseline --tuning gpurun_out/bench_tuning.json"
rm -f gpurun_out/bench_tuning.json
python bench.py --steps 1 --warmup 0 --iters 4 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/stats" -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 $B > gpurun_out/r02_bench_under_rocprof.json 2> gpurun_out/stats.err
cp gpurun_out/stats/s_kernel_stats.csv gpurun_out/r02_bench_kernel_stats.csv; rm -rf gpurun_out/stats
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmcA" -o a -- python3 bench.py --steps 1 --warmup 0 --iters 4 $B > gpurun_out/pmcA.log 2> gpurun_out/pmcA.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmcB" -o b -- python3 bench.py --steps 1 --warmup 0 --iters 4 $B > gpurun_out/pmcB.log 2> gpurun_out/pmcB.err
python tools/pmc_traffic.py gpurun_out/pmcA gpurun_out/pmcB 64 gpurun_out/r02_pmc_hbm_traffic.json gpurun_out/pmcA.log > /dev/null
rm -rf gpurun_out/pmcA gpurun_out/pmcB
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pm" -o p -- python3 bench.py --steps 1 --warmup 0 --iters 6 $B > gpurun_out/pm.log 2>&1
python tools/pmc_mfma.py gpurun_out/pm gpurun_out/r02_pmc_mfma_util.json > gpurun_out/r02_pmc_mfma_util.txt; rm -rf gpurun_out/pm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/pw" -o p -- python3 bench.py --steps 1 --warmup 0 --iters 6 $B > gpurun_out/pw.log 2>&1
python tools/pmc_waves.py gpurun_out/pw gpurun_out/r02_pmc_wave_states.json > gpurun_out/r02_pmc_wave_states.txt; rm -rf gpurun_out/pw
rocprofv3 --kernel-trace -d "$R/gpurun_out/ktrace" -o k --output-format csv -- python3 bench.py --steps 1 --warmup 0 --iters 12 $B > gpurun_out/ktrace.log 2>&1
python tools/iter_breakdown.py gpurun_out/ktrace/k_kernel_trace.csv > gpurun_out/r02_iteration_breakdown.txt; rm -rf gpurun_out/ktrace
tail -c 400 gpurun_out/r02_bench_under_rocprof.json; echo; head -3 gpurun_out/r02_iteration_breakdown.txt
