#!/bin/bash
# A/B of two builds of the library on one box: gpurun_ab/lib_old.so and gpurun_ab/lib_new.so take turns as
# visual_odometry_amd/lib/libvo_hip.so under the same command (default: the quiet bench), two rounds each.
set -e
CMD=${1:-"python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-legs --no-upload-leg"}
for round in 1 2; do
  for v in old new; do
    cp gpurun_ab/lib_$v.so visual_odometry_amd/lib/libvo_hip.so
    timeout -k 10 300 $CMD > gpurun_out/ab_$v.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_$v.json")); k = d["kernels"]
print("$v", round(d["value"], 1), {n: round(k[n]["ms_per_step"], 1) for n in k if k[n]["ms_per_step"] > 40}, flush=True)
PY
  done
done
