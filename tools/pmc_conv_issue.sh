#!/bin/bash
# Issue-side counters of the split 3x3 kernel on one layer shape (tools/bin/conv_timeline built with -DVO_TOKEN=0, i.e. the
# production kernel plus its stamps): how busy each instruction category keeps a SIMD.  One block of counters per run.
set -e -o pipefail
R=$(pwd)
BIN=${1:-$R/tools/bin/conv_timeline0}
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_ci_$name -- $BIN > $R/gpurun_out/pmc_ci_$name.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_ci_$name "conv_bf16x3_kernel" > $R/gpurun_out/pmc_ci_$name.json
  rm -rf $R/gpurun_out/pmc_ci_$name
}
run a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run c SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
echo done
