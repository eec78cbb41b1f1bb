"""
Asymptotic sweep: every hot-path kernel at a size large enough to be bandwidth-bound (well past the 256 MiB
Infinity Cache where it matters), HIP events on the launch stream, algorithmic bytes from DESIGN.md section 3.

    python tools/kernel_sweep.py [--scale 1.0]  > profiles/rNN_kernel_sweep.txt
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = pkg._native
NF = pkg.functional
F = importlib.import_module('normalizing-flows-pytorch_amd.fused')
N.load()
DEV = 'cuda'
PEAK, COPY = 8000.0, 6290.0


def timeit(fn, reps=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    st = torch.cuda.current_stream()
    s.record(st)
    for _ in range(reps):
        fn()
    e.record(st)
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


ONLY = None


TRANS_PEAK = 20.0e12   # v_exp_f32 / v_log_f32 / v_rcp_f32 lane-operations per second chip-wide (tools/probes/vexp_rate_probe.hip, profiles/rNN_vexp_rate.txt)


def report(name, nbytes, fn, reps=20, trans=None):
    """``trans``: hardware transcendental lane-operations the launch executes (the mixture kernels): printed against TRANS_PEAK"""
    if ONLY and not any(o in name for o in ONLY):
        return
    t = timeit(fn, reps)
    gbs = nbytes / t / 1e9
    tr = '' if trans is None else '  %5.2f T transcendental/s = %4.1f%% of %.0f T/s' % (trans / t / 1e12, 100 * trans / t / TRANS_PEAK, TRANS_PEAK / 1e12)
    print('%-46s %9.1f us %9.1f MB %8.1f GB/s  %5.1f%% of 8.0 TB/s  %5.1f%% of 6.29 TB/s copy%s' %
          (name, t * 1e6, nbytes / 1e6, gbs, 100 * gbs / PEAK, 100 * gbs / COPY, tr))


def st():
    return torch.cuda.current_stream().cuda_stream


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--only', default='', help='comma-separated substrings of the row names to measure')
    a = ap.parse_args()
    sc = a.scale
    global ONLY
    ONLY = [o for o in a.only.split(',') if o] or None
    print('# kernel sweep on', torch.cuda.get_device_name(0), '| scale', sc)
    one, zero = torch.full((1, ), 0.5, device=DEV), torch.zeros(1, device=DEV)
    g2 = torch.zeros(2, device=DEV)

    # ---- float4 copy yard-stick -------------------------------------------------------------------------------------
    n = int(2 ** 28 * sc)
    x = torch.empty(n, device=DEV)
    y = torch.empty(n, device=DEV)
    report('torch copy_ (yard-stick)', 8 * n, lambda: y.copy_(x))
    del x, y

    # ---- affine coupling ----------------------------------------------------------------------------------------------
    for name, shape, mode in [('1d D=2', (int(2 ** 24 * sc), 2), N.SPLIT_1D),
                              ('checker (3,32,32)', (int(8192 * sc), 3, 32, 32), N.SPLIT_CHECKER),
                              ('channel (12,16,16)', (int(8192 * sc), 12, 16, 16), N.SPLIT_CHANNEL),
                              ('checker (48,8,8)', (int(8192 * sc), 48, 8, 8), N.SPLIT_CHECKER)]:
        z = torch.randn(shape, device=DEV)
        B = shape[0]
        C, H, W = (shape[1], 1, 1) if len(shape) == 2 else shape[1:]
        params = torch.randn(NF._half_shape(z, mode)[:1] + (2 * NF._half_shape(z, mode)[1], ) + NF._half_shape(z, mode)[2:],
                             device=DEV) * 0.3
        nh = z[0].numel() // 2
        yv, ld = torch.empty_like(z), torch.zeros(B, device=DEV)
        gz, gp = torch.empty_like(z), torch.empty_like(params)
        report('affine_coupling_fwd ' + name, z.numel() * 12 + B * 8,
               lambda: N.call('nf_affine_coupling_fwd', z.data_ptr(), params.data_ptr(), params.data_ptr() + 4 * nh, 2 * nh,
                              one.data_ptr(), zero.data_ptr(), yv.data_ptr(), ld.data_ptr(), mode, 0, 0, B, C, H, W, st()))
        report('affine_coupling_bwd ' + name, z.numel() * 16 + B * 4,
               lambda: N.call('nf_affine_coupling_bwd', yv.data_ptr(), ld.data_ptr(), z.data_ptr(), params.data_ptr(),
                              params.data_ptr() + 4 * nh, 2 * nh, one.data_ptr(), zero.data_ptr(), gz.data_ptr(),
                              gp.data_ptr(), gp.data_ptr() + 4 * nh, g2.data_ptr(), g2.data_ptr() + 4, mode, 0, B, C, H, W,
                              st()))
        half = torch.empty(NF._half_shape(z, mode), device=DEV)
        report('half_gather ' + name, z.numel() * 4,
               lambda: N.call('nf_half_gather', z.data_ptr(), half.data_ptr(), 1, mode, 0, B, C, H, W, st()))
        del z, params, yv, gz, gp, half

    # ---- ActNorm / flow-BN / stats ---------------------------------------------------------------------------------------
    for name, shape in [('2d', (int(2 ** 24 * sc), 2)), ('(12,16,16)', (int(8192 * sc), 12, 16, 16)),
                        ('(48,8,8)', (int(8192 * sc), 48, 8, 8))]:
        x = torch.randn(shape, device=DEV)
        B, C, P = NF._bcp(x)
        yv, ld = torch.empty_like(x), torch.zeros(B, device=DEV)
        ls, bs = torch.randn(C, device=DEV) * 0.1, torch.randn(C, device=DEV)
        gab = torch.zeros(2 * C, device=DEV)
        report('actnorm_fwd ' + name, x.numel() * 8,
               lambda: N.call('nf_chan_affine_fwd', 0, x.data_ptr(), ls.data_ptr(), bs.data_ptr(), None, None, yv.data_ptr(),
                              ld.data_ptr(), 0, B, C, P, st()))
        report('actnorm_bwd ' + name, x.numel() * 12,
               lambda: N.call('nf_chan_affine_bwd', 0, yv.data_ptr(), ld.data_ptr(), x.data_ptr(), ls.data_ptr(),
                              bs.data_ptr(), None, None, yv.data_ptr(), gab.data_ptr(), gab.data_ptr() + 4 * C, B, C, P, st()))
        report('chan_sum ' + name, x.numel() * 4, lambda: N.call('nf_chan_sum', x.data_ptr(), gab.data_ptr(), B, C, P, st()))
        del x, yv

    # ---- invertible 1x1 -----------------------------------------------------------------------------------------------------
    for C, P, B in [(2, 1, int(2 ** 24 * sc)), (3, 1024, int(8192 * sc)), (12, 256, int(8192 * sc)), (48, 64, int(8192 * sc))]:
        z = torch.randn(B, C, P, device=DEV)
        W = torch.linalg.qr(torch.randn(C, C))[0].to(DEV).contiguous()
        yv, ld, lsv = torch.empty_like(z), torch.zeros(B, device=DEV), torch.zeros(C, device=DEV)
        gW = torch.zeros(C, C, device=DEV)
        report('invconv_apply C=%d P=%d' % (C, P), z.numel() * 8,
               lambda: N.call('nf_invconv_apply', z.data_ptr(), W.data_ptr(), 0, yv.data_ptr(), ld.data_ptr(), lsv.data_ptr(),
                              1.0, B, C, P, st()))
        report('invconv_wgrad C=%d P=%d' % (C, P), z.numel() * 8,
               lambda: N.call('nf_invconv_wgrad', yv.data_ptr(), z.data_ptr(), gW.data_ptr(), B, C, P, st()))
        del z, yv

    # ---- fused Glow head for 9 .. 64 channels (ActNorm + 1x1 + gather, MFMA) ------------------------------------------------------
    for C, Hh, B in [(12, 16, int(8192 * sc)), (48, 8, int(8192 * sc))]:
        x = torch.randn(B, C, Hh, Hh, device=DEV)
        W = torch.linalg.qr(torch.randn(C, C))[0].to(DEV).contiguous()
        ls, bs, lsv = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        h, z1c, ld = torch.empty_like(x), torch.empty(B, C // 2, Hh, Hh, device=DEV), torch.zeros(B, device=DEV)
        gx, acc = torch.empty_like(x), torch.zeros(C * C + 2 * C, device=DEV)
        report('glow_head_w_fwd (%d,%d,%d)' % (C, Hh, Hh), x.numel() * 10,
               lambda: N.call('nf_glow_head_w_fwd', x.data_ptr(), ls.data_ptr(), bs.data_ptr(), W.data_ptr(), lsv.data_ptr(), h.data_ptr(),
                              z1c.data_ptr(), ld.data_ptr(), 2, 0, B, C, Hh, Hh, st()))
        report('glow_head_w_bwd (%d,%d,%d)' % (C, Hh, Hh), x.numel() * 12,
               lambda: N.call('nf_glow_head_w_bwd', h.data_ptr(), ld.data_ptr(), x.data_ptr(), ls.data_ptr(), bs.data_ptr(), W.data_ptr(),
                              gx.data_ptr(), acc.data_ptr() + 4 * C * C, acc.data_ptr() + 4 * C * C + 4 * C, acc.data_ptr(), B, C, Hh, Hh,
                              st()))
        del x, h, z1c, gx

    # ---- fused Glow head (2-D) ------------------------------------------------------------------------------------------------
    B = int(2 ** 24 * sc)
    z = torch.randn(B, 2, device=DEV)
    layer = pkg.InvertibleConv1x1(2).to(DEV)
    ls, bs = torch.zeros(2, device=DEV), torch.zeros(2, device=DEV)
    h, z1c, Wm, ld = torch.empty_like(z), torch.empty(B, 1, device=DEV), torch.empty(2, 2, device=DEV), torch.zeros(B, device=DEV)
    report('glow_head_fwd 2d', B * (8 + 8 + 4),
           lambda: N.call('nf_glow_head_fwd', z.data_ptr(), ls.data_ptr(), bs.data_ptr(), layer.P.data_ptr(),
                          layer.L.data_ptr(), layer.U.data_ptr(), layer.L_mask.data_ptr(), layer.U_mask.data_ptr(),
                          layer.sign_s.data_ptr(), layer.log_s.data_ptr(), h.data_ptr(), z1c.data_ptr(), Wm.data_ptr(),
                          ld.data_ptr(), 0, 0, B, 2, 1, 1, st()))
    acc = torch.zeros(16, device=DEV)
    report('glow_head_bwd 2d', B * (8 + 4 + 8 + 8),
           lambda: N.call('nf_glow_head_bwd', h.data_ptr(), z1c.data_ptr(), ld.data_ptr(), z.data_ptr(), ls.data_ptr(),
                          bs.data_ptr(), Wm.data_ptr(), h.data_ptr(), acc.data_ptr(), acc.data_ptr() + 8, acc.data_ptr() + 16,
                          acc.data_ptr() + 24, 0, 0, B, 2, 1, 1, st()))
    del z, h, z1c

    # ---- logit / squeeze ----------------------------------------------------------------------------------------------------------
    B = int(8192 * sc)
    x = torch.rand(B, 3, 32, 32, device=DEV)
    yv, ld = torch.empty_like(x), torch.zeros(B, device=DEV)
    report('logit_fwd (3,32,32)', x.numel() * 8,
           lambda: N.call('nf_logit_fwd', x.data_ptr(), yv.data_ptr(), ld.data_ptr(), 0.01, 0, B, 3072, st()))
    report('logit_bwd (3,32,32)', x.numel() * 12,
           lambda: N.call('nf_logit_bwd', yv.data_ptr(), ld.data_ptr(), x.data_ptr(), yv.data_ptr(), 0.01, B, 3072, st()))
    report('squeeze2d (3,32,32)', x.numel() * 8, lambda: N.call('nf_squeeze2d', x.data_ptr(), yv.data_ptr(), B, 3, 32, 32, st()))
    del x, yv

    # ---- mixture-of-logistics coupling -------------------------------------------------------------------------------------------------
    B, K = int(2 ** 21 * sc), 8
    z = torch.randn(B, 2, device=DEV)
    params = torch.randn(B, 2 + 3 * K, device=DEV) * 0.5
    yv, ld = torch.empty_like(z), torch.zeros(B, device=DEV)
    gz, gp = torch.empty_like(z), torch.empty_like(params)
    scratch, flag = torch.empty(3 * B, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    nb = B * ((4 + 3 * K) * 4 + 8)
    report('mixlog_coupling_fwd 2d K=8', nb + B * 8,
           lambda: N.call('nf_mixlog_coupling_fwd', z.data_ptr(), params.data_ptr(), one.data_ptr(), zero.data_ptr(),
                          yv.data_ptr(), ld.data_ptr(), K, 1e-5, 0, 0, B, 2, 1, 1, st()), trans=B * (4 * 8 + 8))      # (round 6 row kernels: 4 K + 8 per row)
    report('mixlog_coupling_bwd 2d K=8', 2 * nb + B * 12,
           lambda: N.call('nf_mixlog_coupling_bwd', yv.data_ptr(), ld.data_ptr(), z.data_ptr(), params.data_ptr(),
                          one.data_ptr(), zero.data_ptr(), gz.data_ptr(), gp.data_ptr(), g2.data_ptr(), g2.data_ptr() + 4, K,
                          1e-5, 0, 0, B, 2, 1, 1, st()), trans=B * (4 * 8 + 10))
    report('mixlog_coupling_inv 2d K=8 (25 it)', nb + B * 8 + B * 24,
           lambda: N.call('nf_mixlog_coupling_inv', yv.data_ptr(), params.data_ptr(), one.data_ptr(), zero.data_ptr(),
                          gz.data_ptr(), ld.data_ptr(), scratch.data_ptr(), flag.data_ptr(), K, 0, 0, B, 2, 1, 1, st()), reps=5, trans=B * (2 * 8 + 25 * 2 * 8 + 4 * 8 + 8))
    del z, params, yv, gz, gp, scratch

    # ---- fused linear + BatchNorm (MFMA) ------------------------------------------------------------------------------------------------
    Nr = int(2 ** 21 * sc)
    x = torch.randn(Nr, 32, device=DEV)
    out, res = torch.empty_like(x), torch.randn(Nr, 32, device=DEV)
    Wt, g, b = torch.randn(32, 32, device=DEV) * 0.2, torch.rand(32, device=DEV) + 0.5, torch.randn(32, device=DEV) * 0.1
    gamma, beta = torch.rand(32, device=DEV) + 0.5, torch.randn(32, device=DEV) * 0.1
    ws = torch.zeros(64, 32, device=DEV)
    ws[8] += Nr
    rm, rv, nbt = torch.zeros(32, device=DEV), torch.ones(32, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    d = F._desc(F.LinearDesc, in_=x, weight=Wt, weight_g=g, bias=b, residual=res, out=out, bn_gamma=gamma, bn_beta=beta,
                bn_sum=ws[0], bn_sqsum=ws[8], bn_center=b, bn_running_mean=rm, bn_running_var=rv, bn_num_batches=nbt,
                bn_save_mean=ws[16], bn_save_invstd=ws[24], stat_sum=ws[32], stat_sqsum=ws[40])
    report('linear_bn_fwd 32x32 (+BN,+WN,+res,+stats)', Nr * 32 * 4 * 3, lambda: F._launch_fwd([d], Nr, 32, 32, 1))
    flop = 2.0 * Nr * 32 * 32
    t = timeit(lambda: F._launch_fwd([d], Nr, 32, 32, 1))
    print('%-46s %9.2f TFLOP/s fp32 MFMA (peak 157.3)' % ('  ... same launch as GEMM rate', flop / t / 1e12))
    slabs = F.bwd_slabs(Nr)
    gn, gst = torch.empty_like(x), torch.empty_like(x)
    gweff = torch.empty(slabs * 1024, device=DEV)
    acc = torch.zeros(64, 32, device=DEV)
    db = F._desc(F.LinearBwdDesc, in_=x, weight=Wt, weight_g=g, bn_gamma=gamma, bn_beta=beta, bn_save_mean=ws[16],
                 bn_save_invstd=ws[24], gn_src=res, out=out, g_skip=res, cbn_gamma=gamma, cbn_save_mean=ws[16],
                 cbn_save_invstd=ws[24], cbn_sum_g=acc[0], cbn_sum_gx=acc[8], g_store=gst, g_bias=acc[16], g_weff=gweff,
                 gn_out=gn, sum_g=acc[24], sum_gx=acc[32])
    report('linear_bn_bwd 32x32 (all terms)', Nr * 32 * 4 * 6, lambda: F._launch_bwd([db], Nr, 32, 32))
    del x, out, res, gn, gst

    # ---- NLL / Adam ---------------------------------------------------------------------------------------------------------------------
    B = int(2 ** 24 * sc)
    z, ld, loss = torch.randn(B, 2, device=DEV), torch.zeros(B, device=DEV), torch.zeros((), device=DEV)
    report('nll_loss 2d', B * 12, lambda: N.call('nf_nll_loss', z.data_ptr(), ld.data_ptr(), loss.data_ptr(), B, 2, st()))
    n = int(8380754 * max(sc, 1.0))
    p, gr, m, v = (torch.randn(n, device=DEV) for _ in range(4))
    v.abs_()
    stp, lr = torch.zeros(1, dtype=torch.int32, device=DEV), torch.full((1, ), 1e-4, device=DEV)
    report('adam_step (Glow-CIFAR: 8.38 M params)', n * 28,
           lambda: N.call('nf_adam_step', p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), stp.data_ptr(),
                          lr.data_ptr(), 0.9, 0.999, 1e-8, 0.0, 1.0, n, st()))


if __name__ == '__main__':
    main()
