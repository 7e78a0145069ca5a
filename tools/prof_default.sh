cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_c
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-mixed-dpi --host-steps 0 --no-kernel-timing > gpurun_out/prof_c.json 2> gpurun_out/prof_c.err
echo rc=$?
find gpurun_out/prof_c -name "*kernel_stats.csv" | head
