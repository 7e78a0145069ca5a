#!/bin/bash
# engine_api leg of bench.py with the streaming recognizer: first batch x stream batch (MARIE_STREAM_BATCH read by bench)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for cfg in ${CFGS:-8_16 8_8 16_16 8_24}; do
  set -- ${cfg/_/ }
  MARIE_ENGINE_STREAM_BATCH=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-mixed-dpi --host-steps 0 --stream-pages 0 --no-kernel-timing --engine-first-batch $1 > gpurun_out/r03/es.json 2> gpurun_out/r03/es.err || { echo fail; tail -5 gpurun_out/r03/es.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03/es.json") if l.startswith("{")][-1])
e=d["engine_api"]["fixed_lines"]
print("first $1 stream batch $2:", "engine", round(e["value"],1), "s/call", round(e["s_per_call"],3), [[w[0][0],w[1],int(w[2]),int(w[3])] for w in e["timeline_ms"]])
PY
done
