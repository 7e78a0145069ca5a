cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_bc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-mixed-dpi --host-steps 0 --no-kernel-timing > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_bc/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in __import__("os").environ.get("PROF_KERNELS", "beam_candidates,decode_attn,layernorm2").split(",")):
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
PY
