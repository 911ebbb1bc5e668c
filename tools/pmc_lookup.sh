#!/bin/bash
# Fabric traffic of the row-major and the tiled 2-D lookup (FETCH_SIZE and WRITE_SIZE in passes of their own).
set -e -o pipefail
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_lk_$c -- python3 $R/tools/lookup_probe.py 64 3 > $R/gpurun_out/pmc_lk_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_lk_$c "lookup2d_kernel" "lookup2d_tiled_kernel" "pyramid2d_kernel" "pyramid2d_tiled_kernel" > $R/gpurun_out/r02_pmc_lookup_$c.json
  rm -rf $R/gpurun_out/pmc_lk_$c
done
timeout -k 10 100 python3 $R/tools/lookup_probe.py 64 5
timeout -k 10 100 python3 $R/tools/lookup_probe.py 192 5
