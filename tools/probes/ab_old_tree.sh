# same-box A/B of two trees: the repository and a second checkout under _ab_old/ (a scratch `git worktree add _ab_old <commit>`, built there, removed afterwards);
#   HEAD_IN_CHAIN=0|1 sets layers.HEAD_IN_CHAIN in both
for rep in 1 2 3; do
  for d in . _ab_old; do
    (cd $GRAFT_REPO_ROOT/$d; python -c "
import importlib, os, sys, runpy
L = importlib.import_module('normalizing-flows-pytorch_amd.layers'); L.HEAD_IN_CHAIN = bool(int(os.environ.get('HEAD_IN_CHAIN', '1')))
sys.argv = ['bench.py', '--config', 'c4', '--skip-cpu', '--steps', '40', '--warmup', '5']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['value'], d['ms_per_step'])")
  done
done
