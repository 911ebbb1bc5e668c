#!/bin/bash
# Counter passes over the fused stride-2 pair (conv_bf16x3_kernel<..., Geo<3, 3, 2, 1, 32>, false, 2, false, true>) at one layer
# shape, one block of counters per run (run from the repository root): tools/pmc_s2_shortcut.sh "N C H W K"
set -e -o pipefail
R=$(pwd)
shape=${1:-"64 64 480 640 96"}
tag=$(echo $shape | tr ' ' '_')
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/s2_shortcut_probe.py $shape > $R/gpurun_out/s2sc_$tag.txt
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_s2sc_$name -- python3 $R/tools/s2_shortcut_probe.py $shape > $R/gpurun_out/pmc_s2sc_$name.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s2sc_$name "false, 2, false, true" > $R/gpurun_out/pmc_s2sc_${name}_$tag.json
  rm -rf $R/gpurun_out/pmc_s2sc_$name
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run mem FETCH_SIZE GRBM_GUI_ACTIVE
run wr WRITE_SIZE GRBM_GUI_ACTIVE
echo done
