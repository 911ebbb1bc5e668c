set -e -o pipefail
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for shape in "16 64 480 640 64" "16 256 120 160 128"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_w1_$tag -- python3 $R/tools/wino_one.py $shape > $R/gpurun_out/pmc_w1.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_w2_$tag -- python3 $R/tools/wino_one.py $shape > $R/gpurun_out/pmc_w2.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_w1_$tag wino_kernel > $R/gpurun_out/pmc_w1_$tag.json
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_w2_$tag wino_kernel > $R/gpurun_out/pmc_w2_$tag.json
  rm -rf $R/gpurun_out/pmc_w1_$tag $R/gpurun_out/pmc_w2_$tag
done
echo ok
