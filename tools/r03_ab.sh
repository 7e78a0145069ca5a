#!/bin/bash
# A/B of one environment switch on the default bench (same box, back to back): tools/r03_ab.sh NAME VAR
# writes gpurun_out/r03/ab_NAME_{on,off}.json and prints the two lines side by side
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
name=$1; var=$2; shift 2
for arm in off on off on; do
  if [ $arm = on ]; then export $var=1; else unset $var; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --no-secondary --no-mixed-dpi --host-steps 0 "$@" > gpurun_out/r03/ab_${name}_${arm}.json 2> gpurun_out/r03/ab_${name}_${arm}.err || { echo "bench $arm failed"; tail -5 gpurun_out/r03/ab_${name}_${arm}.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03/ab_${name}_${arm}.json") if l.startswith("{")][-1])
print("$var=$arm", "pages/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],1), "igemm iso", round(d["roofline"]["isolated"]["achieved"]),
      "isolated ms", {k: round(v,1) for k,v in d["kernels_ms_per_step_isolated"].items()}, "det", round(d["detector_ms_per_step_alone"],1), "rec", round(d["recognizer_ms_per_step_alone"],1))
PY
done
