# weight gradients of the 8 x 8 level on the bulk kernel (bf16 x 3, nf_conv_bulk_config min_pixels = 4097) against the default (conv_bn.hip, fp32 MFMA)
for rep in 1 2; do
for px in 16385 4097; do
python -c "
import importlib,sys,runpy
N=importlib.import_module('normalizing-flows-pytorch_amd._native'); N.load().nf_conv_bulk_config(-1, $px, -1)
sys.argv=['bench.py','--config','c4','--skip-cpu','--steps','30','--warmup','5']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_pixels $px', d['value'], d['ms_per_step'])"
done; done
