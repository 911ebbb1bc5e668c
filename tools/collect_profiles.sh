#!/bin/bash
# Collect the per-round profiles on a GPU box (run from the repository root):
#   kernel trace + stats of the default bench, and HBM-traffic counters (FETCH_SIZE and
#   WRITE_SIZE in passes of their own) over the hand-written kernels at the bench's shapes.
# Outputs land under gpurun_out/; tools/pmc_summary.py and tools/step_breakdown.py reduce them.
set -e -o pipefail
R=$(pwd)
B=${1:-16}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-mixed-precision-leg > $R/gpurun_out/prof_bench.log 2>&1
echo "trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${c}_encoders -- python3 $R/tools/profile_kernels.py --batch $B --reps 2 --only encoders > $R/gpurun_out/pmc_${c}_encoders.log 2>&1
  echo "$c encoders done"
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${c}_corr -- python3 $R/tools/profile_kernels.py --batch $B --reps 3 --only corr2d,pyr2d,lookup2d,lookup1d,corr1d,pyr1d > $R/gpurun_out/pmc_${c}_corr.log 2>&1
  echo "$c corr done"
done
# the Winograd convolutions: traffic as the mean over all their launches of a bench step, MFMA
# busy cycles at two 3x3 layer shapes and one GRU gate convolution
cd $R && bash tools/pmc_bench_kernel.sh "wino4_kernel wino1d_kernel" wino && cd /tmp
PMCW="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for shape in "16 64 480 640 64" "16 256 120 160 128" "32 384 60 80 256 3 5"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc $PMCW --output-format csv -d $R/gpurun_out/pmc_w1_$tag -- python3 $R/tools/wino_one.py $shape > $R/gpurun_out/pmc_w1.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_w1_$tag wino4_kernel wino1d_kernel > $R/gpurun_out/pmc_w1_$tag.json
  rm -rf $R/gpurun_out/pmc_w1_$tag
done
echo "winograd done"
python3 $R/tools/step_breakdown.py $R/gpurun_out/prof_bench 45 > $R/gpurun_out/step_breakdown.txt
for c in FETCH_SIZE WRITE_SIZE; do
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${c}_encoders instnorm_kernel affine_kernel add_relu_kernel > $R/gpurun_out/pmc_${c}_encoders.json
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${c}_corr corr_abt_kernel\<128 corr_abt_kernel\<160 pyramid2d pyramid1d lookup2d lookup1d > $R/gpurun_out/pmc_${c}_corr.json
done
f=$(ls $R/gpurun_out/prof_bench/*/*_kernel_stats.csv | head -1)
head -46 $f | cut -c1-400 > $R/gpurun_out/bench_kernel_stats_top45.csv
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_*_encoders $R/gpurun_out/pmc_*_corr
echo "all done"
