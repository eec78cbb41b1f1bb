#!/bin/bash
# per-kernel census of a train step in the ORDERED mode: which sites still serialise?   bash tools/det_census.sh c3 rnvp_img fpp_img c4:512
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  c=${spec%%:*}; b=${spec#*:}; [ "$b" = "$spec" ] && b=""
  rm -rf /tmp/skd_$c$b
  NF_DETERMINISTIC=1 NF_BATCH=$b NF_STEPS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/skd_$c$b -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py $c > /dev/null 2>&1
  T=$(find /tmp/skd_$c$b -name "st_kernel_trace.csv" | head -1)
  echo "=== $spec (NF_DETERMINISTIC=1)"
  NF_STEPS=2 TOP=10 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T 2>&1 | cut -c1-140
done
