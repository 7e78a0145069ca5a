cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for cfg in "8 8 8" "8 28 32" "8 56 64" "16 24 32"; do
  set -- $cfg
  MARIE_ENGINE_STREAM_BATCH=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-mixed-dpi --host-steps 0 --stream-pages 0 --no-kernel-timing --engine-first-batch $1 --det-batch $3 > gpurun_out/r03/es.json 2> gpurun_out/r03/es.err || { echo fail; tail -5 gpurun_out/r03/es.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03/es.json") if l.startswith("{")][-1])
e=d["engine_api"]["fixed_lines"]
print("first $1 stream batch $2 det batch $3:", "engine", round(e["value"],1), "s/call", round(e["s_per_call"],3), [[w[0][0],w[1],int(w[2]),int(w[3])] for w in e["timeline_ms"]])
PY
done
