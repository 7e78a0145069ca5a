#!/bin/bash
# round 3, first GPU call: tests touched by the gate / constructor / advisor changes + phase-align A/B of the bench
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_bench_launch.py tests/test_pipeline_gpu.py tests/test_trocr_gpu.py tests/test_content_gpu.py tests/test_overlay_gpu.py -x -q -m gpu > gpurun_out/r03/t1.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r03/t1.log
tail -5 gpurun_out/r03/t1.log
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 6 > gpurun_out/r03/b_aligned.json 2> gpurun_out/r03/b_aligned.err
echo "bench aligned rc=$?"
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 6 --no-phase-align --no-secondary --no-mixed-dpi --host-steps 0 > gpurun_out/r03/b_race.json 2> gpurun_out/r03/b_race.err
echo "bench race rc=$?"
python - <<'PY'
import json
for n in ("b_aligned","b_race"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03/{n}.json") if l.startswith("{")][-1])
        print(n, "pages/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],1), "igemm in situ", round(d["roofline"]["achieved"]), "iso", round(d["roofline"]["isolated"]["achieved"]),
              "engine", d.get("engine_api",{}).get("fixed_lines",{}).get("value"), "stream", (d.get("stream") or {}).get("value"))
        print("   kernels", {k: round(v,1) for k,v in d["kernels_ms_per_step"].items()})
    except Exception as e:
        print(n, "ERR", e)
PY
