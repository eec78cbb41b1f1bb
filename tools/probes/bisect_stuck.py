"""which elements leave the 25-step bisection of the Flow++ inverse with a bracket >= 1e-4 (the reference's 'stuck' rule,
modules.py:205)?  Random coupling parameters of the C3 shape through nf_mixlog_coupling_inv; prints the flag and the brackets."""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = pkg._native
B, K = 65536, 8
torch.manual_seed(0)
z = torch.randn(B, 2, device='cuda')
params = torch.randn(B, 2 + 3 * K, device='cuda') * 0.5
a, c = torch.full((1, ), 0.1, device='cuda'), torch.zeros(1, device='cuda')
y, ld = torch.empty_like(z), torch.zeros(B, device='cuda')
scratch = torch.empty(3 * B, device='cuda')
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
N.call('nf_mixlog_coupling_inv', z.data_ptr(), params.data_ptr(), a.data_ptr(), c.data_ptr(), y.data_ptr(), ld.data_ptr(),
       scratch.data_ptr(), flag.data_ptr(), K, N.SPLIT_1D, 0, B, 2, 1, 1, N.stream())
torch.cuda.synchronize()
lo, hi, tg = scratch[:B], scratch[B:2 * B], scratch[2 * B:]
w = hi - lo
bad = (~(w.abs() < 1e-4)).nonzero().flatten()
print('flag', int(flag), 'stuck elements', bad.numel(), 'of', B)
for i in bad[:10].tolist():
    print(i, 'lo %.6f hi %.6f width %.3e target %.9f' % (float(lo[i]), float(hi[i]), float(w[i]), float(tg[i])))
print('nan x:', int(torch.isnan(y).sum()))
