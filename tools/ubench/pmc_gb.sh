# Build the variants here first (cd tools/ubench && ./build_gb.sh prod [-D...]).  HBM-side traffic of gemm_bench variants: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/ubench
for v in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pg_${v}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_${v}_$c -- ./bin/gb_$v 369280 2 > /tmp/pg_${v}_$c.out 2>/dev/null
  done
  python3 - $v <<'PY'
import csv, sys, glob, collections
v = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pg_{v}_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"]:
            per.setdefault(r["Dispatch_Id"], 0.0)
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    res[c] = list(per.values())
names = ["qkv", "proj+res", "fc1", "fc2+res", "1536->2304", "3072->2304", "768->768", "3072->768"]
out = []
for i, n in enumerate(names):
    k = i * 4 + 3          # 2 warm-up + 2 timed launches per shape: the last one
    if k < len(res["FETCH_SIZE"]):
        out.append("%s rd %.2f wr %.2f" % (n, res["FETCH_SIZE"][k] * 2 * 1024 / 1e9, res["WRITE_SIZE"][k] * 1024 / 1e9))
print(v, "(GB; FETCH_SIZE x 2 for 16-byte lanes, counters in KiB):", " | ".join(out))
PY
done
