#!/bin/bash
# engine_api: streaming recognizer (encode per batch, one beam search) against per-batch recognizer calls, same box, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-mixed-dpi --host-steps 0 --stream-pages 0 --no-kernel-timing $EXTRA > gpurun_out/r03/es.json 2> gpurun_out/r03/es.err || { echo fail; tail -5 gpurun_out/r03/es.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03/es.json") if l.startswith("{")][-1])
e=d["engine_api"]["fixed_lines"]
print("$name:", "engine", round(e["value"],1), "s/call", round(e["s_per_call"],3), "last span ends", int(e["timeline_ms"][-1][3]))
PY
}
for rep in 1 2; do
  EXTRA="--engine-first-batch 8" run "streaming 8/8" MARIE_ENGINE_STREAM_BATCH=8
  EXTRA="--engine-first-batch 8" run "streaming 8/16" MARIE_ENGINE_STREAM_BATCH=16
  EXTRA="--engine-first-batch 32 --engine-page-batch 32" run "per-batch 32/32" MARIE_ENGINE_NO_STREAM=1
  EXTRA="--engine-first-batch 8 --engine-page-batch 32" run "per-batch 8/32/24" MARIE_ENGINE_NO_STREAM=1
done
