#!/bin/bash
# HBM-traffic counters of one hand-written kernel over a short run of the default bench:
#   tools/pmc_bench_kernel.sh "<kernel-name-substring> ..." <tag>   ->  gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE.json
set -e -o pipefail
R=$(pwd); K=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcb_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mixed-precision-leg > $R/gpurun_out/pmcb_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcb_$c $K > $R/gpurun_out/pmc_${TAG}_$c.json
  rm -rf $R/gpurun_out/pmcb_$c
  echo "$c done"
done
