#!/bin/bash
# per-kernel device time of cross_attn bench variants: tools/ubench/prof_cab.sh bin1 bin2 ... (run from the repo root on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/ubench
for b in "$@"; do
  rm -rf /tmp/prof_$b
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$b -- ./bin/$b 2560 3 16 577 768 20 > /tmp/prof_$b.out 2>&1
  tail -2 /tmp/prof_$b.out
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof_$b/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>1: print("   $b %-50.50s calls %5s avg %8.1f us"%(r["Name"], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
