#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py --serial` with and without one environment switch: tools/prof_serial_env.sh NAME VAR
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
name=$1; var=$2
for arm in off on; do
  if [ $arm = on ]; then export $var=1; else unset $var; fi
  rm -rf /tmp/prof_${name}_$arm
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${name}_$arm -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-mixed-dpi --host-steps 0 --no-kernel-timing --serial > gpurun_out/prof/${name}_${arm}_bench.json 2> gpurun_out/prof/${name}_${arm}_bench.err
  echo "$arm rc=$?"
  cp $(find /tmp/prof_${name}_$arm -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${name}_${arm}_kernel_stats.csv
done
python3 - <<PY
import csv
for arm in ("off", "on"):
    rows = list(csv.DictReader(open(f"gpurun_out/prof/${name}_{arm}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("$var", arm, "total device ms", tot / 1e6)
    for r in rows[:14]:
        print("   %-74.74s calls %6s total %9.2f ms avg %9.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", ""), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
