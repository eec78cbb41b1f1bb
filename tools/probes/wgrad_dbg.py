"""mapping probe of the large-batch weight-gradient kernel: G = one-hot (channel oc0, sample b, pixel y, x), activations encode
(channel, row, column) -> g_weff[oc0][ic][tap] must read act[ic][y + dy][x + dx]
    python tools/probes/wgrad_dbg.py H W B b y x"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = pkg._native
fc = importlib.import_module('normalizing-flows-pytorch_amd.fused_conv')
N.load()
H, W, B, b, y, x = (int(v) for v in sys.argv[1:7])
I = 32
dev = 'cuda'
N.call('nf_conv_bulk_config', 1, 0, 0)
act = torch.zeros(B, I, H, W, device=dev)
for ic in range(I):
    for yy in range(H):
        act[:, ic, yy, :] = ic * 1.0 + (yy * W + torch.arange(W, device=dev)) / 1024.0
act += torch.arange(B, device=dev).view(B, 1, 1, 1) * 100.0
oc0 = 5
G = torch.zeros(B, 32, H, W, device=dev)
G[b, oc0, y, x] = 1.0
slabs = int(N.load().nf_conv_wgrad_slabs(B, H, W, 1))
region = torch.full((slabs * 32 * I * 9, ), float('nan'), device=dev)
gb = torch.zeros(8 * 256, device=dev)
arr = (fc.ConvBwdDesc * 1)()
d = fc._desc(fc.ConvBwdDesc, in_=act, weight=torch.zeros(32, I, 3, 3, device=dev), g_skip=G, g_weff=region, g_bias=gb)
ctypes.memmove(ctypes.addressof(arr), ctypes.addressof(d), ctypes.sizeof(fc.ConvBwdDesc))
N.call('nf_conv_bn_wgrad_multi', ctypes.addressof(arr), 1, B, I, 32, H, W, 3, N.stream())
g_w = torch.empty(32, I, 3, 3, device=dev)
fc._slab_sum([(region, g_w, g_w.numel(), g_w.numel(), slabs, False, 9)])
torch.cuda.synchronize()
want = torch.nn.grad.conv2d_weight(act.double(), (32, I, 3, 3), G.double(), padding=1)
print('slabs', slabs, 'max err', float((g_w.double() - want).abs().max()), 'bias', gb.view(8, 256).sum(0)[:8].tolist())
for ic in (0, 1, 7, 31):
    print('ic', ic, 'got', [round(v, 4) for v in g_w[oc0, ic].flatten().tolist()])
    print('     want', [round(v, 4) for v in want[oc0, ic].flatten().tolist()])
other = g_w.clone(); other[oc0] = 0
print('other oc max', float(other.abs().max()), 'at', (other.abs() == other.abs().max()).nonzero()[:3].tolist())
