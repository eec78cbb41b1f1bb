set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in c4; do
  rm -rf /tmp/sk_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/sk_$c -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py $c > /dev/null 2> $OUT/sk_$c.err
  T=$(find /tmp/sk_$c -name "st_kernel_trace.csv" | head -1)
  TOP=45 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T > $OUT/step_kernels_$c.txt
  python $GRAFT_REPO_ROOT/tools/step_kernels.py --by-grid $T chain >> $OUT/step_kernels_$c.txt
done
