for rep in 1 2; do
for on in True False; do
python -c "
import importlib,sys,runpy
NF=importlib.import_module('normalizing-flows-pytorch_amd.functional'); NF.HEAD_BWD_IN_CHAIN=$on
sys.argv=['bench.py','--config','c4','--skip-cpu','--steps','40','--warmup','5']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$on', d['value'], d['ms_per_step'])"
done; done
