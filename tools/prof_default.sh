# rocprofv3 --kernel-trace --stats of the default bench command (and of --serial): per-kernel device time for profiles/rNN.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
for mode in default serial; do
  extra=""; [ "$mode" = serial ] && extra="--serial"
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-mixed-dpi --host-steps 0 --no-kernel-timing --stream-pages 0 $extra > gpurun_out/prof/${mode}_bench.json 2> gpurun_out/prof/${mode}_bench.err
  echo "$mode rc=$?"
  cp $(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${mode}_kernel_stats.csv
done
python3 - <<'PY'
import csv
for mode in ("default", "serial"):
    rows = list(csv.DictReader(open(f"gpurun_out/prof/{mode}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(mode, "total device ms", tot / 1e6)
    for r in rows[:16]:
        print("   %-62.62s calls %6s total %9.2f ms avg %9.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", ""), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
