"""
Workload for the rocprofv3 --pmc passes of C2's and C5's backward as they run in the train step: C2 (B = 4096) S = 32
single-step launches of k_mlp_chain_bwd<1> in deferred-fold mode + one k_glow_fold_all (nf_glow_flow_steps_bwd); C5
(B = 16384) S = 10 launches of k_maf_step_bwd in deferred-fold mode + one k_maf_fold_all; after a calibration copy of known
size.  FETCH_SIZE / WRITE_SIZE in SEPARATE runs, kernel-trace only:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fetch -- python tools/pmc_deferred_steps.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o write -- python tools/pmc_deferred_steps.py
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
    # ---- C5: MAF steps
    B, D, S = 16384, 2, 10
    bn = pkg.BatchNorm((D, ), affine=False).to(dev).train()
    ar = pkg.AutoregressiveTransfrom(D).to(dev).train()
    ms, mt = ar.net_s.draw_masks(dev), ar.net_t.draw_masks(dev)
    head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, ar.perm, ar.s_log_scale, ar.s_bias]
    made = F._made_tensors(ar.net_s, ms) + F._made_tensors(ar.net_t, mt)
    z, gy = torch.randn(B, D, device=dev), torch.randn(B, D, device=dev)
    y, ld, gz = torch.empty_like(z), torch.zeros(B, device=dev), torch.empty_like(z)
    save = torch.empty(N.header_constant('NF_MAF_SAVE_FLOATS'), device=dev)
    htab, mtab = F._ptr_table([t.detach() for t in head]), F._ptr_table([t.detach() for t in made])
    nws = N.header_constant('NF_MAF_WS_FLOATS')
    N.call('nf_maf_step_fwd', z.data_ptr(), y.data_ptr(), ld.data_ptr(), ctypes.addressof(htab), ctypes.addressof(mtab),
           save.data_ptr(), torch.zeros(nws, device=dev).data_ptr(), B, D, 1.0e-5, 0.1, 1.0e-5, N.stream())
    dst = [torch.zeros_like(t) for t in F._made_learnables(made[:27]) + F._made_learnables(made[27:])]
    gtab = F._ptr_table(dst)
    ga, gc = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    blocks = (B + 127) // 128
    slabs, rec = F._maf_steps_scratch(S, blocks, dev)
    nsl, nrec = blocks * N.header_constant('NF_MAF_SLAB_WG_FLOATS'), blocks * N.header_constant('NF_MAF_HEAD_REC_WG')
    pm, pg = F._ptr_table([t.detach() for t in made] * S), F._ptr_table(dst * S)
    pa, pc = F._ptr_table([ga] * S), F._ptr_table([gc] * S)
    for _ in range(reps):
        for i in range(S):
            ws = torch.zeros(nws, device=dev)
            N.call('nf_maf_step_bwd_partial', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                   ctypes.addressof(mtab), save.data_ptr(), ctypes.addressof(gtab), ws.data_ptr(), slabs.data_ptr() + 4 * i * nsl,
                   rec.data_ptr() + 4 * i * nrec, B, D, N.stream())
        N.call('nf_maf_fold_all', ctypes.addressof(pm), ctypes.addressof(pg), ctypes.addressof(pa), ctypes.addressof(pc), S,
               slabs.data_ptr(), rec.data_ptr(), blocks, D, N.stream())
    torch.cuda.synchronize()
    assert N.persistent_timeouts() == 0


if __name__ == '__main__':
    main()
