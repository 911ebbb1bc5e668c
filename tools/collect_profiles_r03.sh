#!/bin/bash
# Profiles of the default bench workload (192 pairs per step, float16-piece arithmetic) on a GPU box, run from the repository root:
#   1. kernel trace + stats  -> gpurun_out/r03_bench_kernel_stats_top45.csv, r03_steady_state_step_breakdown.txt
#   2. FETCH_SIZE and WRITE_SIZE in passes of their own, mean per launch and kernel -> r03_pmc_{FETCH,WRITE}_SIZE.json
#   3. matrix-core busy cycles per kernel -> r03_pmc_mfma.json
#   4. kernel trace of the reference's full argument set -> r03_full_args_step_breakdown.txt
# tools/update_profiles_r03.py folds them into profiles/.
set -e -o pipefail
R=$(pwd)
QUIET="--no-cpu-baseline --no-config-legs --no-mixed-precision-leg --no-upload-leg"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 3 --warmup 1 $QUIET > $R/gpurun_out/r03_prof_bench.log 2>&1
python3 $R/tools/step_breakdown.py $R/gpurun_out/prof_bench 45 > $R/gpurun_out/r03_steady_state_step_breakdown.txt
f=$(ls $R/gpurun_out/prof_bench/*/*_kernel_stats.csv | head -1)
head -46 $f | cut -c1-400 > $R/gpurun_out/r03_bench_kernel_stats_top45.csv
rm -rf $R/gpurun_out/prof_bench
echo "trace done"
# every kernel under its own (templated) name; tools/update_profiles_r03.py groups them by entry point
KERNELS=()
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcb_$c -- python3 $R/bench.py --steps 1 --warmup 0 $QUIET > $R/gpurun_out/r03_pmcb_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcb_$c "${KERNELS[@]}" > $R/gpurun_out/r03_pmc_$c.json
  rm -rf $R/gpurun_out/pmcb_$c
  echo "$c done"
done
timeout -s KILL 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcb_mfma -- python3 $R/bench.py --steps 1 --warmup 0 $QUIET > $R/gpurun_out/r03_pmcb_mfma.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcb_mfma "${KERNELS[@]}" > $R/gpurun_out/r03_pmc_mfma.json
rm -rf $R/gpurun_out/pmcb_mfma
echo "mfma done"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_full -- python3 $R/bench.py --stereo full --steps 2 --warmup 1 $QUIET > $R/gpurun_out/r03_prof_full.log 2>&1
python3 $R/tools/step_breakdown.py $R/gpurun_out/prof_full 45 > $R/gpurun_out/r03_full_args_step_breakdown.txt
rm -rf $R/gpurun_out/prof_full
echo "all done"
