#!/bin/bash
# engine_api leg of bench.py over batch shapes: first batch x page batch x phase gate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for cfg in ${CFGS:-8_32 32_32 64_64}; do
  set -- ${cfg/_/ }
  for gate in ""; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-mixed-dpi --host-steps 0 --stream-pages 0 --no-kernel-timing --engine-first-batch $1 --engine-page-batch $2 $gate > gpurun_out/r03/es.json 2> gpurun_out/r03/es.err || { echo fail; tail -3 gpurun_out/r03/es.err; exit 1; }
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03/es.json") if l.startswith("{")][-1])
e=d["engine_api"]["fixed_lines"]
print("first $1 batch $2 gate '$gate':", "engine", round(e["value"],1), "s/call", round(e["s_per_call"],3), [[w[0][0],w[1],int(w[2]),int(w[3])] for w in e["timeline_ms"]])
PY
  done
done
