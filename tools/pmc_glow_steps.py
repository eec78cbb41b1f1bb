"""
Workload for the rocprofv3 --pmc passes of C2's backward as it runs in the train step (B = 4096): S single-step launches of
k_mlp_chain_bwd<1> in deferred-fold mode + one k_glow_fold_all (nf_glow_flow_steps_bwd), after a calibration copy of known
size.  FETCH_SIZE / WRITE_SIZE in SEPARATE runs, kernel-trace only:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fetch -- python tools/pmc_glow_steps.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o write -- python tools/pmc_glow_steps.py
    python tools/pmc_probe.py --summarise out/fetch_counter_collection.csv out/write_counter_collection.csv
"""
import ctypes
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    pkg = importlib.import_module('normalizing-flows-pytorch_amd')
    F = importlib.import_module('normalizing-flows-pytorch_amd.fused')
    N = pkg._native
    N.load()
    dev = torch.device('cuda', 0)
    x = torch.randn(2 ** 26, device=dev)                       # 256 MiB read + 256 MiB write: calibration
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    del x, y
    B, D, S, reps = int(os.environ.get('NF_PMC_B', 4096)), 2, 32, 5
    steps, sinks, keep = [], [], []
    for i in range(S):
        k = pkg.AffineCoupling((D, ), odd=bool(i & 1)).to(dev).train()
        a, c = pkg.ActNorm((D, )).to(dev), pkg.InvertibleConv1x1(D).to(dev)
        h = [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
        m = F._mlp_tensors(k.net)
        steps.append((int(i & 1), h, m))
        sinks.append([torch.zeros_like(t) for t in F._glow_step_learnables(h, m)])
        keep.append((k, a, c))
    table = F._glow_flow_table(steps, sinks, D, dev)
    host = ctypes.addressof(F._GLOW_FLOW_HOST[table.data_ptr()])
    z, gy = torch.randn(B, D, device=dev), torch.randn(B, D, device=dev)
    ld = torch.zeros(B, device=dev)
    ys, gzs = torch.empty(S, B, D, device=dev), torch.empty(S, B, D, device=dev)
    saves = torch.empty(S, N.header_constant('NF_GLOW_FLOW_SAVE_FLOATS'), device=dev)
    nws = N.header_constant('NF_MLP_WS_FLOATS')
    N.call('nf_glow_flow_steps_fwd', host, S, z.data_ptr(), ys.data_ptr(), ld.data_ptr(), saves.data_ptr(),
           torch.zeros(S * nws, device=dev).data_ptr(), B, D, 1, 1.0e-5, 0.1, 1.0e-5, N.stream())
    slabs, rec = F._glow_steps_scratch(S, (B + 127) // 128, dev)
    for _ in range(reps):
        ws = torch.zeros(S * nws, device=dev)
        N.call('nf_glow_flow_steps_bwd', host, table.data_ptr(), S, z.data_ptr(), ys.data_ptr(), gy.data_ptr(), None, gzs.data_ptr(),
               saves.data_ptr(), 1, ws.data_ptr(), slabs.data_ptr(), rec.data_ptr(), B, D, 1, 1.0e-5, 1.0e-5, N.stream())
    torch.cuda.synchronize()
    assert N.persistent_timeouts() == 0


if __name__ == '__main__':
    main()
