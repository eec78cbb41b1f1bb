set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --skip-cpu > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in c4 c1; do
  rm -rf /tmp/sk_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk_$c -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py $c > /dev/null 2> $OUT/sk_$c.err
  F=$(find /tmp/sk_$c -name "st_kernel_stats.csv" | head -1)
  T=$(find /tmp/sk_$c -name "st_kernel_trace.csv" | head -1)
  TOP=60 python $GRAFT_REPO_ROOT/tools/step_kernels.py --summarise $F > $OUT/step_kernels_$c.txt
  python $GRAFT_REPO_ROOT/tools/step_kernels.py --by-grid $T chain >> $OUT/step_kernels_$c.txt
  python $GRAFT_REPO_ROOT/tools/step_kernels.py --by-grid $T head >> $OUT/step_kernels_$c.txt
done
cd $GRAFT_REPO_ROOT
python tools/probes/parity_depth.py c1 > $OUT/parity_depth_c1.txt 2>&1
python tools/cpu_threads.py c4 8 16 32 64 > $OUT/cpu_threads_c4.txt 2>&1
SECONDS_PER=4 python tools/cpu_threads.py c1 1 4 8 16 32 > $OUT/cpu_threads_c1.txt 2>&1
ls -la $OUT
