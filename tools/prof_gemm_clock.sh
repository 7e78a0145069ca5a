# Effective clock and MFMA-busy share of conv_igemm<256> on the ViT GEMM shapes: rocprofv3 --pmc passes over tools/ubenc./bin/gb_prod
# (counters only with --kernel-trace, one group per pass).  Build the binary here first: (cd tools/ubench && ./build_gb.sh prod).  GRBM_GUI_ACTIVE / 8 / wall = clock; SQ_VALU_MFMA_BUSY_CYCLES summed over
# the 1024 SIMDs / (1024 x GRBM_GUI_ACTIVE / 8) = share of cycles the matrix cores are busy.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/ubench
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/gclk_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/gclk_$c -- ./bin/gb_prod 1477120 3 > /tmp/gclk_$c.out 2>/tmp/gclk_$c.err
  echo "$c rc=$?"
  cp $(find /tmp/gclk_$c -name "*counter_collection.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/prof/gclk_$c.csv
  cp $(find /tmp/gclk_$c -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/prof/gclk_trace_$c.csv
done
python3 - <<'PY'
import csv, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof/"
def load(c):
    vals = collections.OrderedDict()
    for r in csv.DictReader(open(root + f"gclk_{c}.csv")):
        if "conv_igemm" not in r["Kernel_Name"]: continue
        vals.setdefault(r["Dispatch_Id"], 0.0)
        vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    dur = {}
    for r in csv.DictReader(open(root + f"gclk_trace_{c}.csv")):
        if "conv_igemm" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return vals, dur
g, gd = load("GRBM_GUI_ACTIVE")
m, md = load("SQ_VALU_MFMA_BUSY_CYCLES")
gk, mk = list(g), list(m)
print("dispatch  wall_ms  clock_GHz   mfma_busy_share")
for i in range(min(len(gk), len(mk))):
    wall = gd[gk[i]]
    clock = g[gk[i]] / 8 / wall
    share = m[mk[i]] / (1024 * (md[mk[i]] * clock))      # the MFMA pass's own wall time x this pass's clock
    print(f"{i:3d}  {wall*1e3:8.3f}  {clock/1e9:6.3f}  {share:6.3f}")
PY
