#!/bin/bash
# Two counter passes over the F(4x4,3x3) Winograd kernel at one layer shape (run from the repository root).
set -e -o pipefail
R=$(pwd)
shape=${1:-"16 256 120 160 128"}
tag=$(echo $shape | tr ' ' '_')
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_w4a_$tag -- python3 $R/tools/wino_one.py $shape 3 4 > $R/gpurun_out/pmc_w4a.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_w4a_$tag wino4_kernel > $R/gpurun_out/pmc_w4a_$tag.json
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_w4b_$tag -- python3 $R/tools/wino_one.py $shape 3 4 > $R/gpurun_out/pmc_w4b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_w4b_$tag wino4_kernel > $R/gpurun_out/pmc_w4b_$tag.json
rm -rf $R/gpurun_out/pmc_w4a_$tag $R/gpurun_out/pmc_w4b_$tag
echo done
