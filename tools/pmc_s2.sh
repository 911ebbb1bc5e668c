#!/bin/bash
# Counter passes over the streaming 3x3 stride-2 kernel at one layer shape (run from the repository root).
set -e -o pipefail
R=$(pwd)
shape=${1:-"32 64 480 640 64"}
tag=$(echo $shape | tr ' ' '_')
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_s2a_$tag -- python3 $R/tools/wino_one.py $shape 3 32 > $R/gpurun_out/pmc_s2a.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s2a_$tag conv3s2_kernel > $R/gpurun_out/pmc_s2a_$tag.json
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_s2b_$tag -- python3 $R/tools/wino_one.py $shape 3 32 > $R/gpurun_out/pmc_s2b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s2b_$tag conv3s2_kernel > $R/gpurun_out/pmc_s2b_$tag.json
timeout -s KILL 120 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_s2c_$tag -- python3 $R/tools/wino_one.py $shape 3 32 > $R/gpurun_out/pmc_s2c.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s2c_$tag conv3s2_kernel > $R/gpurun_out/pmc_s2c_$tag.json
rm -rf $R/gpurun_out/pmc_s2a_$tag $R/gpurun_out/pmc_s2b_$tag $R/gpurun_out/pmc_s2c_$tag
echo done
