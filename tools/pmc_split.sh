#!/bin/bash
# MFMA-busy and wait counters of the split-bf16 direct 3x3 kernel at a few layer shapes.
# Run from the repository root on a GPU box; JSON summaries land in gpurun_out/.
set -e -o pipefail
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for shape in "16 64 480 640 64" "64 128 120 160 128" "64 256 60 80 192"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_split_$tag -- python3 $R/tools/wino_one.py $shape 3 1 > $R/gpurun_out/pmc_split.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_split_$tag conv3x3_bf16x3_kernel > $R/gpurun_out/pmc_split_$tag.json
  f=$(ls $R/gpurun_out/pmc_split_$tag/*/*_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep conv3x3_bf16x3_kernel $f | awk -F, '{print $(NF-1)-$(NF-2)}' | tr '\n' ' ' > $R/gpurun_out/pmc_split_$tag.dur || true
  rm -rf $R/gpurun_out/pmc_split_$tag
done
echo done
