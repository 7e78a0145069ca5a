cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/ubench
for v in "$@"; do
  echo "== $v"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- ./bin/cab_$v ${CAB_ARGS:-1280 3 16 577 768 20} 2>/dev/null | grep -E "us per|checksum"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("   %-46.46s calls %5s avg %8.1f us  min %8.1f" % (r["Name"].replace("(anonymous namespace)::","").replace("_ZN12_GLOBAL__N_1",""), r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
