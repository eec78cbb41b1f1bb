# in-graph timeline of a C4 step with the heads' forward in the chain prologue (on) and on its own launches (off)
cd /tmp && export TMPDIR=/tmp
for on in 1 0; do
  rm -rf /tmp/tlf_$on
  NF_HEAD_AB=$on rocprofv3 --kernel-trace --output-format csv -d /tmp/tlf_$on -o st -- python -c "
import importlib, os, sys, runpy
L = importlib.import_module('normalizing-flows-pytorch_amd.layers'); L.HEAD_IN_CHAIN = bool(int(os.environ['NF_HEAD_AB']))
sys.argv = ['step_only.py', 'c4', '6']
runpy.run_path(os.environ['GRAFT_REPO_ROOT'] + '/tools/probes/step_only.py', run_name='__main__')" > /dev/null 2>&1
  T=$(find /tmp/tlf_$on -name "st_kernel_trace.csv" | head -1)
  echo "=== HEAD_IN_CHAIN = $on"
  python $GRAFT_REPO_ROOT/tools/probes/step_timeline.py $T | grep -E "step:|chain_fwd|chain_bwd|glow_head_w"
done
