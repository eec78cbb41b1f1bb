# image RealNVP train step with the flow BatchNorm head in one persistent launch (on) and as statistics + apply launches (off), same box
for rep in 1 2; do
for on in True False; do
python -c "
import importlib,sys,runpy
NF=importlib.import_module('normalizing-flows-pytorch_amd.functional'); NF.FLOWBN_FUSED=$on
sys.argv=['bench.py','--config','rnvp_img','--skip-cpu','--steps','30','--warmup','5']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FLOWBN_FUSED $on', d['value'], d['ms_per_step'])"
done; done
