# HBM-side traffic per kernel family: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only with --kernel-trace, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes) over one warm-up + one timed step of the default bench configuration.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-mixed-dpi --host-steps 0 --no-kernel-timing --stream-pages 0 > gpurun_out/prof/pmc_$c.json 2> gpurun_out/prof/pmc_$c.err
  echo "$c rc=$?"
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" gpurun_out/prof/pmc_traffic.json "bench.py --steps 1 --warmup 1 (2 passes of the default 64-page dit_trocr step)" '{"workload": "dit_trocr", "pages": 64, "det_batch": 8, "decode_len": 15, "model": "base", "det_passes": 1, "precision": "f16"}'
