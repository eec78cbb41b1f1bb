"""
Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in SEPARATE runs, kernel-trace only):
a calibration copy of known size, then the dominant kernel of the bench workload (k_linear_bn_bwd) at the bench shape
(N = 4096) and at an asymptotic shape (N = 2^21), then the fused coupling.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fetch -- python tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o write -- python tools/pmc_probe.py
    python tools/pmc_probe.py --summarise out/fetch_counter_collection.csv out/write_counter_collection.csv
"""
import csv
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(paths):
    agg = {}
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r.get('Kernel_Name') or r.get('Kernel Name') or ''
            key = (name.split('(')[0][:60], r['Counter_Name'], r.get('Grid_Size', ''))
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    print('%-60s %-12s %-10s %8s %14s' % ('kernel', 'counter', 'grid', 'calls', 'avg value'))
    for (k, c, g), (n, v) in sorted(agg.items()):
        if k.startswith('k_') or 'copy' in k.lower() or 'void k_' in k or any(t in k for t in ('k_mlp', 'k_flowpp', 'k_maf', 'k_glow', 'k_conv')):
            print('%-60s %-12s %-10s %8d %14.2f' % (k, c, g, n, v / n))


def main():
    import torch
    pkg = importlib.import_module('normalizing-flows-pytorch_amd')
    F = importlib.import_module('normalizing-flows-pytorch_amd.fused')
    N = pkg._native
    N.load()
    dev = 'cuda'
    x = torch.randn(2 ** 26, device=dev)                       # 256 MiB read + 256 MiB write: calibration
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    del x, y
    for Nr, reps in ((4096, 20), (2 ** 21, 3)):
        xin, gn_src, out, gn_out = (torch.randn(Nr, 32, device=dev) for _ in range(4))
        Wt, wg = torch.randn(32, 32, device=dev) * 0.2, torch.rand(32, device=dev) + 0.5
        gam, bet = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
        ws = torch.zeros(64, 32, device=dev)
        ws[24] += 1
        gweff = torch.empty(F.bwd_slabs(Nr) * 1024, device=dev)
        d = F._desc(F.LinearBwdDesc, in_=xin, weight=Wt, weight_g=wg, bn_gamma=gam, bn_beta=bet, bn_save_mean=ws[16],
                    bn_save_invstd=ws[24], gn_src=gn_src, out=out, cbn_gamma=gam, cbn_save_mean=ws[16],
                    cbn_save_invstd=ws[24], cbn_sum_g=ws[32], cbn_sum_gx=ws[40], g_bias=ws[48], g_weff=gweff, gn_out=gn_out,
                    sum_g=ws[0], sum_gx=ws[8])
        for _ in range(reps):
            F._launch_bwd([d], Nr, 32, 32)
        torch.cuda.synchronize()
    for B, reps in ((4096, 20), (2 ** 24, 3)):
        z, params = torch.randn(B, 2, device=dev), torch.randn(B, 2, device=dev)
        a, c = torch.full((1, ), 0.5, device=dev), torch.zeros(1, device=dev)
        yv, ld = torch.empty_like(z), torch.zeros(B, device=dev)
        for _ in range(reps):
            N.call('nf_affine_coupling_fwd', z.data_ptr(), params.data_ptr(), params.data_ptr() + 4, 2, a.data_ptr(),
                   c.data_ptr(), yv.data_ptr(), ld.data_ptr(), 0, 0, 0, B, 2, 1, 1, N.stream())
        torch.cuda.synchronize()
    # the persistent flow-step backward kernels (dominant kernels of C2 and C5) at the bench shapes, and the Flow++
    # conditioner backward (C3)
    import ctypes
    dv = torch.device(dev, 0)
    for Nr, reps in ((4096, 20), ):
        D = 2
        a, c, k = pkg.ActNorm((D, )).to(dev), pkg.InvertibleConv1x1(D).to(dev), pkg.AffineCoupling((D, )).to(dev).train()
        head = [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
        mts = F._mlp_tensors(k.net)
        lh = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
        lm = list(mts[:18]) + [t for j in range(5) for t in mts[18 + 5 * j:18 + 5 * j + 2]]
        dh, dm = [torch.zeros_like(t) for t in lh], [torch.zeros_like(t) for t in lm]
        htab, mtab = F._ptr_table([t.detach() for t in head]), F._ptr_table([t.detach() for t in mts])
        hg, mg = F._ptr_table(dh), F._ptr_table(dm)
        z, gy = torch.randn(Nr, D, device=dev), torch.randn(Nr, D, device=dev)
        y, ld, gz = torch.empty_like(z), torch.zeros(Nr, device=dev), torch.empty_like(z)
        save = torch.empty(N.header_constant('NF_REALNVP_SAVE_FLOATS'), device=dev)
        nws = N.header_constant('NF_MLP_WS_FLOATS')
        N.call('nf_glow_step_vec_fwd', z.data_ptr(), y.data_ptr(), ld.data_ptr(), ctypes.addressof(htab), ctypes.addressof(mtab),
               save.data_ptr(), torch.zeros(nws, device=dev).data_ptr(), Nr, D, 0, 1, 1.0e-5, 0.1, 1.0e-5, N.stream())
        slabs = F._mlp_slabs(dv)
        for _ in range(reps):
            ws = torch.zeros(nws, device=dev)
            N.call('nf_glow_step_vec_bwd', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                   ctypes.addressof(mtab), save.data_ptr(), ctypes.addressof(hg), ctypes.addressof(mg), 1, ws.data_ptr(),
                   slabs.data_ptr(), Nr, D, 0, 1, 1.0e-5, 1.0e-5, N.stream())
        torch.cuda.synchronize()
    for Nr, reps in ((16384, 20), ):
        D = 2
        bn = pkg.BatchNorm((D, ), affine=False).to(dev).train()
        ar = pkg.AutoregressiveTransfrom(D).to(dev).train()
        ms, mt = ar.net_s.draw_masks(dv), ar.net_t.draw_masks(dv)
        head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, ar.perm, ar.s_log_scale,
                ar.s_bias]
        made = F._made_tensors(ar.net_s, ms) + F._made_tensors(ar.net_t, mt)
        z, gy = torch.randn(Nr, D, device=dev), torch.randn(Nr, D, device=dev)
        y, ld, gz = torch.empty_like(z), torch.zeros(Nr, device=dev), torch.empty_like(z)
        save = torch.empty(N.header_constant('NF_MAF_SAVE_FLOATS'), device=dev)
        htab, mtab = F._ptr_table([t.detach() for t in head]), F._ptr_table([t.detach() for t in made])
        nws = N.header_constant('NF_MAF_WS_FLOATS')
        N.call('nf_maf_step_fwd', z.data_ptr(), y.data_ptr(), ld.data_ptr(), ctypes.addressof(htab), ctypes.addressof(mtab),
               save.data_ptr(), torch.zeros(nws, device=dev).data_ptr(), Nr, D, 1.0e-5, 0.1, 1.0e-5, N.stream())
        learn = F._made_learnables(made[:27]) + F._made_learnables(made[27:])
        dst = [torch.zeros_like(t) for t in learn]
        gtab = F._ptr_table(dst)
        ga, gc = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        slabs = F._maf_slabs(dv)
        for _ in range(reps):
            ws = torch.zeros(nws, device=dev)
            N.call('nf_maf_step_bwd', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                   ctypes.addressof(mtab), save.data_ptr(), ctypes.addressof(gtab), ga.data_ptr(), gc.data_ptr(), ws.data_ptr(),
                   slabs.data_ptr(), Nr, D, N.stream())
        torch.cuda.synchronize()
    for Nr, reps in ((65536, 10), ):
        layer = pkg.MixLogAttnCoupling((2, ), n_mixtures=8).to(dev)
        ts, F_ = F._flowpp_tensors(layer.net)
        O = ts[13].shape[0]
        xin, gout = torch.randn(Nr, 1, device=dev), torch.randn(Nr, O, device=dev)
        gx = torch.empty_like(xin)
        dst = [torch.zeros_like(t) for t in ts]
        d = [t.data_ptr() for t in dst]
        d[7] += 4 * 2 * F_ * 32
        d[8] += 4 * 2 * F_
        wsb = F.flowpp_bwd_workspace(dv)
        args = F._flowpp_fwd_args(ts, F_)
        for _ in range(reps):
            N.call('nf_flowpp_cond_bwd', xin.data_ptr(), *args, gout.data_ptr(), gx.data_ptr(), *d, wsb.data_ptr(), 1, 1, 1, 1, 0, Nr, 1, O,
                   N.stream())
        torch.cuda.synchronize()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
        summarise(sys.argv[2:])
    else:
        main()
