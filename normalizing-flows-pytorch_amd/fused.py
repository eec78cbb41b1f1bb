"""
Fused conditioner networks on the fp32-MFMA linear + BatchNorm kernels (csrc/linear_bn.hip, C ABI
``nf_linear_bn_fwd / nf_linear_bn_bwd / nf_weight_grad_finalize``):

  * ``mlp_forward``  -- the MLP conditioner of AffineCoupling (flows/modules.py:393-413): 6 launches forward,
                        6 + 1 backward, instead of ~60 / ~150 framework kernels;
  * ``made_forward`` -- MAF's pair of MADE nets (flows/maf.py:49-64, :103-104), s-net and t-net in the SAME launches:
                        4 forward, 4 + 1 backward.

Both are exact restatements of the module math in training mode (batch statistics, gradients through the
statistics, running-statistics bookkeeping) and in evaluation mode (running statistics).
"""
import ctypes
import os as _os

import torch

from . import _native as N
from . import workspace as WS

BN_EPS, BN_MOMENTUM, WN_EPS = 1.0e-5, 0.1, 1.0e-5
H = 32  # hidden width of every reference conditioner (base_filters=32)
R = 8   # NF_STAT_REPL of include/nfhip.h: replicas of every atomically accumulated 32-vector
WS_ROWS = 2 * R + 2   # per BatchNorm workspace rows of 32: sum[R], sqsum[R], save_mean, save_invstd

_LIN_FIELDS = ['in_', 'weight', 'weight_g', 'mask', 'bias', 'residual', 'out', 'bn_gamma', 'bn_beta', 'bn_sum', 'bn_sqsum',
               'bn_center', 'bn_running_mean', 'bn_running_var', 'bn_num_batches', 'bn_save_mean', 'bn_save_invstd',
               'stat_sum', 'stat_sqsum']
_BWD_FIELDS = ['in_', 'weight', 'weight_g', 'mask', 'bn_gamma', 'bn_beta', 'bn_save_mean', 'bn_save_invstd', 'g_direct',
               'g_skip', 'gn_src', 'out', 'cbn_gamma', 'cbn_save_mean', 'cbn_save_invstd', 'cbn_sum_g', 'cbn_sum_gx',
               'g_store', 'g_bias', 'g_weff', 'gn_out', 'sum_g', 'sum_gx']
_WG_FIELDS = ['g_weff', 'weight', 'weight_g', 'mask', 'g_weight', 'g_weight_g', 'vec_src0', 'vec_dst0', 'vec_src1',
              'vec_dst1']
_WG_INTS = ['vec_n0', 'vec_n1', 'I', 'O', 'n_slabs', 'accumulate', 'vec_repl', 'reserved']


class LinearDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in _LIN_FIELDS]


class LinearBwdDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in _BWD_FIELDS]


class CopyDesc(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('n', ctypes.c_int64)]


class WnDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in ('v', 'g', 'w', 'g_w', 'g_v', 'g_g')] + \
               [(f, ctypes.c_int) for f in ('O', 'M', 'accumulate', 'reserved')]


class WeightGradDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in _WG_FIELDS] + [(f, ctypes.c_int) for f in _WG_INTS]


def _p(t):
    return None if t is None else t.data_ptr()


def _desc(cls, **kw):
    d = cls()
    for k, v in kw.items():
        setattr(d, k, _p(v) if isinstance(v, torch.Tensor) or v is None else v)
    return d


def _launch_fwd(descs, Nrows, I, O, training):
    arr = (LinearDesc * len(descs))(*descs)
    N.call('nf_linear_bn_fwd', ctypes.addressof(arr), len(descs), Nrows, I, O, int(training), BN_EPS, BN_MOMENTUM, WN_EPS,
           N.stream())


def _launch_bwd(descs, Nrows, I, O):
    arr = (LinearBwdDesc * len(descs))(*descs)
    N.call('nf_linear_bn_bwd', ctypes.addressof(arr), len(descs), Nrows, I, O, WN_EPS, N.stream())


def _launch_wgrad(descs):
    arr = (WeightGradDesc * len(descs))(*descs)
    N.call('nf_weight_grad_finalize', ctypes.addressof(arr), len(descs), WN_EPS, N.stream())


def bwd_slabs(Nrows):
    return int(N.load().nf_linear_bwd_slabs(Nrows))


def _finalize_jobs(lin_jobs, bn_jobs, slabs, direct):
    """descriptor list of nf_weight_grad_finalize.  lin_jobs: (g_weff, weight, weight_g|None, mask|None, g_bias_src, dst
    triple (g_weight, g_weight_g|None, g_bias)); bn_jobs: (sum_gx, sum_g, dst pair (g_gamma, g_beta))."""
    descs = []
    for g_weff, W, Wg, M, gb_src, (dW, dWg, dB) in lin_jobs:
        descs.append(_desc(WeightGradDesc, g_weff=g_weff, weight=W, weight_g=Wg, mask=M, g_weight=dW, g_weight_g=dWg,
                           vec_src0=gb_src, vec_dst0=dB, vec_n0=W.shape[0], I=W.shape[1], O=W.shape[0], n_slabs=slabs,
                           accumulate=int(direct), vec_repl=R))
    for s_gx, s_g, (dG, dBt) in bn_jobs:
        descs.append(_desc(WeightGradDesc, vec_src0=s_gx, vec_dst0=dG, vec_n0=dG.numel(), vec_src1=s_g, vec_dst1=dBt,
                           vec_n1=dBt.numel(), I=1, O=1, n_slabs=0, accumulate=int(direct), vec_repl=R))
    return descs


# ----------------------------------------------------------------------------------------------------------------------
# MLP conditioner
# ----------------------------------------------------------------------------------------------------------------------
class _FusedMLP(torch.autograd.Function):
    """x -> WN0 -> [BN,ReLU,WN, BN,ReLU,WN, +skip] * n_blocks -> BN,ReLU,WN_out.

    tensors: per linear (v, g, bias) * (2*n_blocks + 2), then per BatchNorm (gamma, beta, running_mean, running_var,
    num_batches_tracked) * (2*n_blocks + 1)."""

    @staticmethod
    def forward(ctx, x, training, n_blocks, *tensors):
        nl, nb = 2 * n_blocks + 2, 2 * n_blocks + 1
        lin = [tensors[3 * i:3 * i + 3] for i in range(nl)]
        bns = [tensors[3 * nl + 5 * i:3 * nl + 5 * i + 5] for i in range(nb)]
        x = x.contiguous()
        Nrows, I0 = x.shape
        O_out = lin[-1][0].shape[0]
        dev = x.device
        ws = WS.zeros(nb * WS_ROWS * H, dev).view(nb, WS_ROWS, H)            # [sum*R, sqsum*R, save_mean, save_invstd]
        acts = [torch.empty(Nrows, H, dtype=torch.float32, device=dev) for _ in range(nb)]
        out = torch.empty(Nrows, O_out, dtype=torch.float32, device=dev)

        def bn_kw(j):
            g, b, rm, rv, nbt = bns[j]
            return dict(bn_gamma=g, bn_beta=b, bn_sum=ws[j, 0], bn_sqsum=ws[j, R], bn_center=lin[j][2], bn_running_mean=rm,
                        bn_running_var=rv, bn_num_batches=nbt, bn_save_mean=ws[j, 2 * R], bn_save_invstd=ws[j, 2 * R + 1])

        # K0
        _launch_fwd([_desc(LinearDesc, in_=x, weight=lin[0][0], weight_g=lin[0][1], bias=lin[0][2], out=acts[0],
                           stat_sum=ws[0, 0], stat_sqsum=ws[0, R])], Nrows, I0, H, training)
        for j in range(1, nb):                    # linear j consumes acts[j-1] through BatchNorm j-1
            res = acts[j - 2] if j % 2 == 0 else None      # second linear of a residual block adds the block input
            _launch_fwd([_desc(LinearDesc, in_=acts[j - 1], weight=lin[j][0], weight_g=lin[j][1], bias=lin[j][2],
                               residual=res, out=acts[j], stat_sum=ws[j, 0], stat_sqsum=ws[j, R], **bn_kw(j - 1))],
                        Nrows, H, H, training)
        _launch_fwd([_desc(LinearDesc, in_=acts[nb - 1], weight=lin[nl - 1][0], weight_g=lin[nl - 1][1],
                           bias=lin[nl - 1][2], out=out, **bn_kw(nb - 1))], Nrows, H, O_out, training)
        ctx.save_for_backward(x, ws, *acts, *[t for l in lin for t in l[:2]], *[t for b in bns for t in b[:2]])
        ctx.meta = (n_blocks, Nrows, I0, O_out, bool(training))
        from .functional import _sinks
        ctx.sinks = _sinks(*[t for l in lin for t in l], *[t for b in bns for t in b[:2]])
        return out

    @staticmethod
    def backward(ctx, g_out):
        n_blocks, Nrows, I0, O_out, training = ctx.meta
        nl, nb = 2 * n_blocks + 2, 2 * n_blocks + 1
        saved = ctx.saved_tensors
        x, ws = saved[0], saved[1]
        acts = saved[2:2 + nb]
        vg = saved[2 + nb:2 + nb + 2 * nl]
        gb = saved[2 + nb + 2 * nl:]
        V = [vg[2 * i] for i in range(nl)]
        G = [vg[2 * i + 1] for i in range(nl)]
        gamma = [gb[2 * i] for i in range(nb)]
        beta = [gb[2 * i + 1] for i in range(nb)]
        dev = x.device
        g_out = g_out.contiguous()
        # g_weff: per-workgroup slabs (written, reduced by the finalize launch); zero-initialised atomically
        # accumulated vectors: g_bias per linear, (sum_g, sum_gx) per BatchNorm
        slabs = bwd_slabs(Nrows)
        g_weff = list(torch.empty(nl, slabs * H * H, dtype=torch.float32, device=dev).unbind(0))
        acc = WS.zeros(nl * R * H + nb * 2 * R * H, dev)
        g_bias = [acc[i * R * H:(i + 1) * R * H] for i in range(nl)]          # R replicas of 32
        sums = acc[nl * R * H:].view(nb, 2, R * H)
        gn = [torch.empty(Nrows, H, dtype=torch.float32, device=dev) for _ in range(nb)]
        g_x = torch.empty(Nrows, I0, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        G_skip = None                              # assembled gradient of the latest residual-stream tensor

        def in_bn(j):
            return dict(bn_gamma=gamma[j], bn_beta=beta[j], bn_save_mean=ws[j, 2 * R], bn_save_invstd=ws[j, 2 * R + 1])

        def cons_bn(j):                            # evaluation mode: statistics are constants -> no mean terms
            return dict(cbn_gamma=gamma[j], cbn_save_mean=ws[j, 2 * R], cbn_save_invstd=ws[j, 2 * R + 1],
                        cbn_sum_g=sums[j, 0] if training else None, cbn_sum_gx=sums[j, 1] if training else None)

        # last linear: G = g_out
        _launch_bwd([_desc(LinearBwdDesc, in_=acts[nb - 1], weight=V[nl - 1], weight_g=G[nl - 1], g_direct=g_out,
                           g_bias=g_bias[nl - 1], g_weff=g_weff[nl - 1], gn_out=gn[nb - 1], sum_g=sums[nb - 1, 0],
                           sum_gx=sums[nb - 1, 1], **in_bn(nb - 1))], Nrows, H, O_out)
        for j in range(nb - 1, 0, -1):             # linear j produced acts[j]; its consumer BatchNorm is j
            is_stream = (j % 2 == 0)               # acts[j] is a residual-stream tensor (block output)
            store = torch.empty(Nrows, H, dtype=torch.float32, device=dev) if is_stream else None
            _launch_bwd([_desc(LinearBwdDesc, in_=acts[j - 1], weight=V[j], weight_g=G[j], gn_src=gn[j], out=acts[j],
                               g_skip=G_skip if is_stream else None, g_store=store, g_bias=g_bias[j], g_weff=g_weff[j],
                               gn_out=gn[j - 1], sum_g=sums[j - 1, 0], sum_gx=sums[j - 1, 1], **in_bn(j - 1),
                               **cons_bn(j))], Nrows, H, H)
            if is_stream:
                G_skip = store
        _launch_bwd([_desc(LinearBwdDesc, in_=x, weight=V[0], weight_g=G[0], gn_src=gn[0], out=acts[0], g_skip=G_skip,
                           g_bias=g_bias[0], g_weff=g_weff[0], gn_out=g_x, **cons_bn(0))], Nrows, I0, H)
        direct = ctx.sinks is not None
        if direct:                                 # accumulate straight into the .grad buffers (GradBucket-tagged)
            dst_lin = [tuple(ctx.sinks[3 * i:3 * i + 3]) for i in range(nl)]
            dst_bn = [tuple(ctx.sinks[3 * nl + 2 * j:3 * nl + 2 * j + 2]) for j in range(nb)]
        else:
            dst_lin = [(torch.empty_like(V[i]), torch.empty_like(G[i]),
                        torch.empty(V[i].shape[0], dtype=torch.float32, device=dev)) for i in range(nl)]
            dst_bn = [(torch.empty(H, dtype=torch.float32, device=dev), torch.empty(H, dtype=torch.float32, device=dev))
                      for _ in range(nb)]
        _launch_wgrad(_finalize_jobs([(g_weff[i], V[i], G[i], None, g_bias[i], dst_lin[i]) for i in range(nl)],
                                     [(sums[j, 1], sums[j, 0], dst_bn[j]) for j in range(nb)], slabs, direct))
        if direct:
            return (g_x, None, None) + (None, ) * (3 * nl + 5 * nb)
        grads = []
        for i in range(nl):
            grads += list(dst_lin[i])
        for j in range(nb):
            grads += [dst_bn[j][0], dst_bn[j][1], None, None, None]        # g_gamma, g_beta
        return (g_x, None, None) + tuple(grads)


# ----------------------------------------------------------------------------------------------------------------------
# MLP conditioner, persistent single-launch form (csrc/mlp_chain.hip) for N <= NF_MLP_MAX_ROWS
# ----------------------------------------------------------------------------------------------------------------------
def _mlp_modules(mlp):
    lins = [mlp.in_block[0]]
    bns = []
    for blk in mlp.mid_block:
        bns += [blk.net[0], blk.net[3]]
        lins += [blk.net[2], blk.net[5]]
    bns.append(mlp.out_block[0])
    lins.append(mlp.out_block[2])
    return lins, bns


def _mlp_tensors(mlp):
    lins, bns = _mlp_modules(mlp)
    tensors = []
    for wn in lins:
        m = wn.module
        tensors += [m.weight_v, m.weight_g, m.bias]
    for bn in bns:
        tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
    return tensors


def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def mlp_chain_usable(mlp, x):
    return (len(mlp.mid_block) == 2 and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and 0 < x.shape[0] <= N.mlp_max_rows())


def mlp_chain_forward_nograd(mlp, x, training):
    """forward only (no autograd graph); returns (out, save_stats)."""
    ts = _mlp_tensors(mlp)
    x = x.contiguous()
    Nrows, I0 = x.shape
    O_out = ts[15].shape[0]
    out = torch.empty(Nrows, O_out, dtype=torch.float32, device=x.device)
    save = torch.empty(5, 2, H, dtype=torch.float32, device=x.device)
    ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), x.device)
    tab = _ptr_table([t.detach() for t in ts])
    N.call('nf_mlp_chain_fwd', N.ptr(x), ctypes.addressof(tab), N.ptr(out), N.ptr(save), N.ptr(ws), Nrows, I0, O_out,
           int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
    return out, save


def mlp_forward(mlp, x, chain=None):
    """``mlp``: conditioners.MLP with weight_norm=True.  Returns the conditioner output (N, out_channels).
    chain: None = the persistent single-launch kernels when the batch fits them, else one launch per linear."""
    if chain is None:
        chain = mlp_chain_usable(mlp, x)
    tensors = _mlp_tensors(mlp)
    if chain:
        return _FusedMLPChain.apply(x, mlp.training, *tensors)
    return _FusedMLP.apply(x, mlp.training, len(mlp.mid_block), *tensors)


_MLP_SLABS = {}


def _mlp_slabs(device):
    """scratch of nf_mlp_chain_bwd (per-workgroup weight-gradient partials); launches on a stream are ordered, so one
    buffer per device serves every layer."""
    t = _MLP_SLABS.get(device)
    if t is None:
        t = _MLP_SLABS[device] = torch.empty(N.header_constant('NF_MLP_BWD_SLAB_FLOATS'), dtype=torch.float32, device=device)
    return t


class _FusedMLPChain(torch.autograd.Function):
    """the MLP conditioner as one persistent launch per direction (csrc/mlp_chain.hip); same tensor list as _FusedMLP."""

    @staticmethod
    def forward(ctx, x, training, *tensors):
        nl, nb = 6, 5
        x = x.contiguous()
        Nrows, I0 = x.shape
        O_out = tensors[15].shape[0]
        dev = x.device
        out = torch.empty(Nrows, O_out, dtype=torch.float32, device=dev)
        save = torch.empty(nb, 2, H, dtype=torch.float32, device=dev)
        ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), dev)
        tab = _ptr_table(tensors)
        N.call('nf_mlp_chain_fwd', N.ptr(x), ctypes.addressof(tab), N.ptr(out), N.ptr(save), N.ptr(ws), Nrows, I0, O_out,
               int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
        ctx.save_for_backward(x, save, *tensors)
        ctx.meta = (Nrows, I0, O_out, bool(training))
        from .functional import _sinks
        learn = list(tensors[:3 * nl]) + [t for j in range(nb) for t in tensors[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
        ctx.sinks = _sinks(*learn)
        return out

    @staticmethod
    def backward(ctx, g_out):
        nl, nb = 6, 5
        Nrows, I0, O_out, training = ctx.meta
        x, save, *tensors = ctx.saved_tensors
        dev = x.device
        g_out = g_out.contiguous()
        learn = list(tensors[:3 * nl]) + [t for j in range(nb) for t in tensors[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
        direct = ctx.sinks is not None
        dst = ctx.sinks if direct else [torch.empty_like(t) for t in learn]
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), dev)
        tab, gtab = _ptr_table(tensors), _ptr_table(dst)
        N.call('nf_mlp_chain_bwd', N.ptr(x), ctypes.addressof(tab), N.ptr(save), N.ptr(g_out), _p(g_x), ctypes.addressof(gtab),
               int(direct), N.ptr(ws), N.ptr(_mlp_slabs(dev)), Nrows, I0, O_out, int(training), BN_EPS, WN_EPS, N.stream())
        if direct:
            return (g_x, None) + (None, ) * len(tensors)
        grads = list(dst[:3 * nl])
        for j in range(nb):
            grads += [dst[3 * nl + 2 * j], dst[3 * nl + 2 * j + 1], None, None, None]
        return (g_x, None) + tuple(grads)


# ----------------------------------------------------------------------------------------------------------------------
# MADE pair (MAF)
# ----------------------------------------------------------------------------------------------------------------------
class _FusedMADEPair(torch.autograd.Function):
    """two MADE nets on the same input, same launches.  tensors per net: weights*(nh+1), biases*(nh+1),
    masks*(nh+1), then per BatchNorm (gamma, beta, running_mean, running_var, num_batches_tracked)*nh."""

    @staticmethod
    def forward(ctx, z, training, nh, *tensors):
        per = 3 * (nh + 1) + 5 * nh
        nets = [tensors[i * per:(i + 1) * per] for i in range(2)]
        z = z.contiguous()
        Nrows, D = z.shape
        dev = z.device
        ws = WS.zeros(2 * nh * WS_ROWS * H, dev).view(2, nh, WS_ROWS, H)
        acts = [[torch.empty(Nrows, H, dtype=torch.float32, device=dev) for _ in range(nh)] for _ in range(2)]
        outs = [torch.empty(Nrows, D, dtype=torch.float32, device=dev) for _ in range(2)]

        def W(n, l): return nets[n][l]
        def Bs(n, l): return nets[n][(nh + 1) + l]
        def M(n, l): return nets[n][2 * (nh + 1) + l]
        def BN(n, j): return nets[n][3 * (nh + 1) + 5 * j:3 * (nh + 1) + 5 * j + 5]

        def bn_kw(n, j):
            g, b, rm, rv, nbt = BN(n, j)
            return dict(bn_gamma=g, bn_beta=b, bn_sum=ws[n, j, 0], bn_sqsum=ws[n, j, R], bn_center=Bs(n, j),
                        bn_running_mean=rm, bn_running_var=rv, bn_num_batches=nbt, bn_save_mean=ws[n, j, 2 * R],
                        bn_save_invstd=ws[n, j, 2 * R + 1])

        _launch_fwd([_desc(LinearDesc, in_=z, weight=W(n, 0), mask=M(n, 0), bias=Bs(n, 0), out=acts[n][0],
                           stat_sum=ws[n, 0, 0], stat_sqsum=ws[n, 0, R]) for n in range(2)], Nrows, D, H, training)
        for l in range(1, nh):
            _launch_fwd([_desc(LinearDesc, in_=acts[n][l - 1], weight=W(n, l), mask=M(n, l), bias=Bs(n, l), out=acts[n][l],
                               stat_sum=ws[n, l, 0], stat_sqsum=ws[n, l, R], **bn_kw(n, l - 1)) for n in range(2)],
                        Nrows, H, H, training)
        _launch_fwd([_desc(LinearDesc, in_=acts[n][nh - 1], weight=W(n, nh), mask=M(n, nh), bias=Bs(n, nh), out=outs[n],
                           **bn_kw(n, nh - 1)) for n in range(2)], Nrows, H, D, training)
        keep = [z, ws]
        for n in range(2):
            keep += acts[n]
            keep += [W(n, l) for l in range(nh + 1)] + [M(n, l) for l in range(nh + 1)]
            keep += [BN(n, j)[0] for j in range(nh)] + [BN(n, j)[1] for j in range(nh)]
        ctx.save_for_backward(*keep)
        ctx.meta = (nh, Nrows, D, bool(training))
        from .functional import _sinks
        sink_in = []
        for n in range(2):
            sink_in += [W(n, l) for l in range(nh + 1)] + [Bs(n, l) for l in range(nh + 1)]
            for j in range(nh):
                sink_in += list(BN(n, j)[:2])
        ctx.sinks = _sinks(*sink_in)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, g_s, g_t):
        nh, Nrows, D, training = ctx.meta
        saved = ctx.saved_tensors
        z, ws = saved[0], saved[1]
        per = nh + 2 * (nh + 1) + 2 * nh
        dev = z.device
        nets = []
        for n in range(2):
            blk = saved[2 + n * per:2 + (n + 1) * per]
            nets.append(dict(acts=blk[:nh], W=blk[nh:2 * nh + 1], M=blk[2 * nh + 1:3 * nh + 2],
                             gamma=blk[3 * nh + 2:4 * nh + 2], beta=blk[4 * nh + 2:5 * nh + 2]))
        g_outs = [g_s.contiguous(), g_t.contiguous()]
        slabs = bwd_slabs(Nrows)
        gw_all = torch.empty(2, nh + 1, slabs * H * H, dtype=torch.float32, device=dev)
        per_acc = (nh + 1) * R * H + nh * 2 * R * H
        acc = WS.zeros(2 * per_acc, dev).view(2, per_acc)
        g_weff, g_bias, sums = [], [], []
        for n in range(2):
            g_weff.append([gw_all[n, l] for l in range(nh + 1)])
            g_bias.append([acc[n, l * R * H:(l + 1) * R * H] for l in range(nh + 1)])
            sums.append(acc[n, (nh + 1) * R * H:].view(nh, 2, R * H))
        gn = [[torch.empty(Nrows, H, dtype=torch.float32, device=dev) for _ in range(nh)] for _ in range(2)]
        g_z = [torch.empty(Nrows, D, dtype=torch.float32, device=dev) for _ in range(2)]

        def in_bn(n, j):
            return dict(bn_gamma=nets[n]['gamma'][j], bn_beta=nets[n]['beta'][j], bn_save_mean=ws[n, j, 2 * R],
                        bn_save_invstd=ws[n, j, 2 * R + 1])

        def cons_bn(n, j):
            return dict(cbn_gamma=nets[n]['gamma'][j], cbn_save_mean=ws[n, j, 2 * R], cbn_save_invstd=ws[n, j, 2 * R + 1],
                        cbn_sum_g=sums[n][j, 0] if training else None, cbn_sum_gx=sums[n][j, 1] if training else None)

        _launch_bwd([_desc(LinearBwdDesc, in_=nets[n]['acts'][nh - 1], weight=nets[n]['W'][nh], mask=nets[n]['M'][nh],
                           g_direct=g_outs[n], g_bias=g_bias[n][nh], g_weff=g_weff[n][nh], gn_out=gn[n][nh - 1],
                           sum_g=sums[n][nh - 1, 0], sum_gx=sums[n][nh - 1, 1], **in_bn(n, nh - 1)) for n in range(2)],
                    Nrows, H, D)
        for l in range(nh - 1, 0, -1):
            _launch_bwd([_desc(LinearBwdDesc, in_=nets[n]['acts'][l - 1], weight=nets[n]['W'][l], mask=nets[n]['M'][l],
                               gn_src=gn[n][l], out=nets[n]['acts'][l], g_bias=g_bias[n][l], g_weff=g_weff[n][l],
                               gn_out=gn[n][l - 1], sum_g=sums[n][l - 1, 0], sum_gx=sums[n][l - 1, 1], **in_bn(n, l - 1),
                               **cons_bn(n, l)) for n in range(2)], Nrows, H, H)
        _launch_bwd([_desc(LinearBwdDesc, in_=z, weight=nets[n]['W'][0], mask=nets[n]['M'][0], gn_src=gn[n][0],
                           out=nets[n]['acts'][0], g_bias=g_bias[n][0], g_weff=g_weff[n][0], gn_out=g_z[n],
                           **cons_bn(n, 0)) for n in range(2)], Nrows, D, H)
        direct = ctx.sinks is not None
        per_sink = 2 * (nh + 1) + 2 * nh
        lin_jobs, bn_jobs, ret = [], [], []
        for n in range(2):
            if direct:
                sk = ctx.sinks[n * per_sink:(n + 1) * per_sink]
                dW, dB, dBN = sk[:nh + 1], sk[nh + 1:2 * nh + 2], sk[2 * nh + 2:]
            else:
                dW = [torch.empty_like(nets[n]['W'][l]) for l in range(nh + 1)]
                dB = [torch.empty(nets[n]['W'][l].shape[0], dtype=torch.float32, device=dev) for l in range(nh + 1)]
                dBN = [torch.empty(H, dtype=torch.float32, device=dev) for _ in range(2 * nh)]
            for l in range(nh + 1):
                lin_jobs.append((g_weff[n][l], nets[n]['W'][l], None, nets[n]['M'][l], g_bias[n][l], (dW[l], None, dB[l])))
            for j in range(nh):
                bn_jobs.append((sums[n][j, 1], sums[n][j, 0], (dBN[2 * j], dBN[2 * j + 1])))
            ret += list(dW) + list(dB) + [None] * (nh + 1)
            for j in range(nh):
                ret += [dBN[2 * j], dBN[2 * j + 1], None, None, None]
        _launch_wgrad(_finalize_jobs(lin_jobs, bn_jobs, slabs, direct))
        g_in = g_z[0] + g_z[1] if ctx.needs_input_grad[0] else None
        if direct:
            return (g_in, None, None) + (None, ) * len(ret)
        return (g_in, None, None) + tuple(ret)


def made_pair_forward(net_s, net_t, z, masks_s, masks_t):
    """s_raw, t = MADE_s(z), MADE_t(z) with the given per-layer mask tensors (device, fp32)."""
    nh = net_s.num_hidden
    tensors = []
    for net, masks in ((net_s, masks_s), (net_t, masks_t)):
        tensors += list(net.weights) + list(net.biases) + list(masks)
        for bn in net.bnorms:
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
    training = net_s.training
    return _FusedMADEPair.apply(z, training, nh, *tensors)


# ----------------------------------------------------------------------------------------------------------------------
# Flow++ conditioner (density data): the whole gated-attention stack in one launch
# ----------------------------------------------------------------------------------------------------------------------
def flowpp_cond_fusable(net, x):
    """net: the nn.Sequential of MixLogAttnCoupling for len(dims) == 1 (coupling.py:142-149)."""
    try:
        first, gated, ln1, attn, ln2, last = net
    except (TypeError, ValueError):
        return False
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and isinstance(first, torch.nn.Linear)
            and first.in_features <= 4 and first.out_features == H and attn.filters == H and attn.channels == H
            and isinstance(last, torch.nn.Linear) and last.out_features <= 64 and tuple(ln1.normalized_shape) == (H, ))


def _flowpp_tensors(net):
    first, gated, ln1, attn, ln2, last = net
    F_ = attn.filters
    return [first.weight, first.bias, gated.op.weight, gated.op.bias, ln1.weight, ln1.bias, attn.pos_emb,
            attn.conv1.weight, attn.conv1.bias, attn.conv2.weight, attn.conv2.bias, ln2.weight, ln2.bias, last.weight,
            last.bias], F_


def _flowpp_fwd_args(ts, F_):
    (W0, b0, Wg, bg, l1g, l1b, pos, c1w, c1b, c2w, c2b, l2g, l2b, W5, b5) = [t.detach() for t in ts]
    # the single-position attention only sees conv1's Q rows [2F:3F] (the softmax over one key is identically 1)
    return [N.ptr(W0), N.ptr(b0), N.ptr(Wg), N.ptr(bg), N.ptr(l1g), N.ptr(l1b), N.ptr(pos), c1w.data_ptr() + 4 * 2 * F_ * H,
            c1b.data_ptr() + 4 * 2 * F_, N.ptr(c2w), N.ptr(c2b), N.ptr(l2g), N.ptr(l2b), N.ptr(W5), N.ptr(b5)]


_FPP_WS = {}


def flowpp_bwd_workspace(device):
    """the partial-sum slabs of nf_flowpp_cond_bwd; one per device: backward launches on a stream are ordered, the next
    call may overwrite what the previous call's second kernel has already folded."""
    ws = _FPP_WS.get(device)
    if ws is None:
        ws = _FPP_WS[device] = torch.empty(N.header_constant('NF_FLOWPP_BWD_WS_FLOATS'), dtype=torch.float32, device=device)
    return ws


class FlowppFinDesc(ctypes.Structure):
    """nf_flowpp_fin_desc of include/nfhip.h"""
    _fields_ = [(f, ctypes.c_void_p) for f in ('workspace', 'g_W0', 'g_b0', 'g_Wg', 'g_bg', 'g_ln1_g', 'g_ln1_b', 'g_pos', 'g_Wq',
                                                'g_bq', 'g_W2', 'g_b2', 'g_ln2_g', 'g_ln2_b', 'g_W5', 'g_b5', 'g_scale', 'g_bias',
                                                'next_log_scale', 'g_next_log_scale', 'g_next_bias')] + \
               [('odd', ctypes.c_int), ('reserved', ctypes.c_int)]


class FlowppDefer:
    """Deferred slab finalizes of the fused Flow++ steps (nf_flowpp_vec_step_bwd phase 1 + nf_flowpp_vec_step_finalize).
    A training step opens it around loss.backward() (FlowTrainer does): every step's backward then keeps its slab workspace
    and queues a descriptor instead of launching its own finalize, ``flush`` folds all of them in one launch per eight
    steps (C3: 32 finalize launches of 7 us + a launch gap each -> 4 launches).  Closed (the default) nothing is deferred:
    a bare ``loss.backward()`` outside a trainer cannot leave gradients unfolded."""

    def __init__(self):
        self.active = False
        self.queue = []         # (desc, K, N, tensors kept alive)
        self.pool = {}          # device -> [workspace tensors]

    def begin(self):
        self.queue = []
        self.active = FLOWPP_DEFER

    def workspace(self, device):
        pool = self.pool.setdefault(device, [])
        i = len(self.queue)
        while len(pool) <= i:
            pool.append(torch.empty(N.header_constant('NF_FLOWPP_BWD_WS_FLOATS'), dtype=torch.float32, device=device))
        return pool[i]

    def flush(self):
        """fold everything queued (also on the error path: the queue never survives a step)"""
        q, self.queue, self.active = self.queue, [], False
        i = 0
        while i < len(q):
            j = i
            while j < len(q) and q[j][1:3] == q[i][1:3]:      # one call per run of equal (K, N)
                j += 1
            arr = (FlowppFinDesc * (j - i))(*[e[0] for e in q[i:j]])
            N.call('nf_flowpp_vec_step_finalize', ctypes.addressof(arr), j - i, q[i][1], q[i][2], N.stream())
            i = j


FLOWPP_DEFER = True             # (internal: tests compare the deferred finalize with the per-step one)
FPP_DEFER = FlowppDefer()


class _FusedFlowppCond(torch.autograd.Function):
    """x (N, I0) -> the (N, O) coupling parameters; one launch forward, one launch backward (csrc/flowpp_cond.hip)."""

    @staticmethod
    def forward(ctx, x, F_, *ts):
        for t in ts:
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise RuntimeError('fused Flow++ conditioner needs contiguous fp32 device parameters')
        x = x.contiguous()
        Nrows, I0 = x.shape
        O = ts[13].shape[0]
        out = torch.empty(Nrows, O, dtype=torch.float32, device=x.device)
        N.call('nf_flowpp_cond_fwd', N.ptr(x), *_flowpp_fwd_args(ts, F_), N.ptr(out), I0, 1, Nrows, I0, O, N.stream())
        ctx.save_for_backward(x, *ts)
        ctx.F_ = F_
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, *ts = ctx.saved_tensors
        F_ = ctx.F_
        Nrows, I0 = x.shape
        O = ts[13].shape[0]
        g_out = g_out.contiguous()
        from .functional import _sinks
        sinks = _sinks(*ts)
        if sinks is not None:
            dst, direct = sinks, True
        else:                                                    # handed to autograd, which may keep them: not arena memory
            flat = torch.zeros(sum(t.numel() for t in ts), dtype=torch.float32, device=x.device)
            dst, o = [], 0
            for t in ts:
                dst.append(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()
            direct = False
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        d = [t.data_ptr() for t in dst]
        d[7] += 4 * 2 * F_ * H                                   # conv1 gradient rows [2F:3F]; V / K rows stay exactly zero
        d[8] += 4 * 2 * F_
        N.call('nf_flowpp_cond_bwd', N.ptr(x), *_flowpp_fwd_args(ts, F_), N.ptr(g_out), _p(g_x), *d,
               N.ptr(flowpp_bwd_workspace(x.device)), I0, 1, I0, 1, 0, Nrows, I0, O, N.stream())
        if direct:
            return (g_x, None) + (None, ) * len(ts)
        return (g_x, None) + tuple(dst)


def flowpp_cond_forward(net, x):
    """the (N, O) output of the density Flow++ conditioner ``net`` (see flowpp_cond_fusable)."""
    ts, F_ = _flowpp_tensors(net)
    return _FusedFlowppCond.apply(x, F_, *ts)


def flowpp_cond_forward_nograd(net, x):
    (W0, b0, Wg, bg, l1g, l1b, pos, c1w, c1b, c2w, c2b, l2g, l2b, W5, b5), F_ = _flowpp_tensors(net)
    x = x.contiguous()
    Nrows, I0 = x.shape
    O = W5.shape[0]
    out = torch.empty(Nrows, O, dtype=torch.float32, device=x.device)
    N.call('nf_flowpp_cond_fwd', N.ptr(x), N.ptr(W0.detach()), N.ptr(b0.detach()), N.ptr(Wg.detach()), N.ptr(bg.detach()),
           N.ptr(l1g.detach()), N.ptr(l1b.detach()), N.ptr(pos.detach()), c1w.data_ptr() + 4 * 2 * F_ * H,
           c1b.data_ptr() + 4 * 2 * F_, N.ptr(c2w.detach()), N.ptr(c2b.detach()), N.ptr(l2g.detach()), N.ptr(l2b.detach()),
           N.ptr(W5.detach()), N.ptr(b5.detach()), N.ptr(out), I0, 1, Nrows, I0, O, N.stream())
    return out


# ----------------------------------------------------------------------------------------------------------------------
# one whole Glow flow step on vector data (ActNorm -> invertible 1x1 -> affine coupling with the MLP conditioner)
# ----------------------------------------------------------------------------------------------------------------------
def glow_step_vec_usable(z, mlp):
    """dims = (D,) with D in (2, 4) and a batch the persistent kernels hold (csrc/mlp_chain.hip, GLOW variant)."""
    return (z.is_cuda and z.dim() == 2 and z.dtype == torch.float32 and z.shape[1] in (2, 4) and mlp.fused
            and len(mlp.mid_block) == 2 and 0 < z.shape[0] <= N.mlp_max_rows())


class _GlowStepVec(torch.autograd.Function):
    """(y, ld) = coupling(invconv(actnorm(z))) in one launch, its autograd in one launch.
    head: log_scale, bias, P, L, U, L_mask, U_mask, sign_s, log_s, s_log_scale, s_bias; then the 43 MLP tensors."""

    @staticmethod
    def forward(ctx, z, ld, odd, training, *tensors):
        head, mlp = tensors[:11], tensors[11:]
        z = z.contiguous()
        Nrows, D = z.shape
        y = torch.empty_like(z)
        save = torch.empty(5, 2, H, dtype=torch.float32, device=z.device)
        ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), z.device)
        htab, mtab = _ptr_table(head), _ptr_table(mlp)
        N.call('nf_glow_step_vec_fwd', N.ptr(z), N.ptr(y), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab),
               N.ptr(save), N.ptr(ws), Nrows, D, int(odd), int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
        ctx.save_for_backward(z, save, *tensors)
        ctx.meta = (int(odd), bool(training))
        from .functional import _sinks
        nl, nb = 6, 5
        learn_h = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
        learn_m = list(mlp[:3 * nl]) + [t for j in range(nb) for t in mlp[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
        ctx.sinks = _sinks(*(learn_h + learn_m))
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        nl, nb = 6, 5
        odd, training = ctx.meta
        z, save, *tensors = ctx.saved_tensors
        head, mlp = tensors[:11], tensors[11:]
        Nrows, D = z.shape
        dev = z.device
        g_y = g_y.contiguous()
        g_ld = None if g_ld is None else g_ld.contiguous()
        learn_h = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
        learn_m = list(mlp[:3 * nl]) + [t for j in range(nb) for t in mlp[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
        direct = ctx.sinks is not None
        dst = ctx.sinks if direct else [torch.empty_like(t) for t in learn_h + learn_m]
        g_z = torch.empty_like(z)
        ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), dev)
        htab, mtab = _ptr_table(head), _ptr_table(mlp)
        hg, mg = _ptr_table(dst[:7]), _ptr_table(dst[7:])
        N.call('nf_glow_step_vec_bwd', N.ptr(z), N.ptr(g_y), _p(g_ld), N.ptr(g_z), ctypes.addressof(htab), ctypes.addressof(mtab),
               N.ptr(save), ctypes.addressof(hg), ctypes.addressof(mg), int(direct), N.ptr(ws), N.ptr(_mlp_slabs(dev)), Nrows, D,
               odd, int(training), BN_EPS, WN_EPS, N.stream())
        if direct:
            return (g_z, g_ld, None, None) + (None, ) * len(tensors)
        gh = [dst[0], dst[1], None, dst[2], dst[3], None, None, None, dst[4], dst[5], dst[6]]
        gm = list(dst[7:7 + 3 * nl])
        for j in range(nb):
            gm += [dst[7 + 3 * nl + 2 * j], dst[7 + 3 * nl + 2 * j + 1], None, None, None]
        return (g_z, g_ld, None, None) + tuple(gh) + tuple(gm)


def glow_step_vec(z, ld, actnorm, conv, coupling):
    """ActNorm ``actnorm`` -> InvertibleConv1x1 ``conv`` -> AffineCoupling ``coupling`` on (N, D) data, fused."""
    from .functional import _owned_ld
    head = [actnorm.log_scale, actnorm.bias, conv.P, conv.L, conv.U, conv.L_mask, conv.U_mask, conv.sign_s, conv.log_s,
            coupling.s_log_scale, coupling.s_bias]
    return _GlowStepVec.apply(z, _owned_ld(ld), int(coupling.odd), coupling.net.training, *(head + _mlp_tensors(coupling.net)))


# ----------------------------------------------------------------------------------------------------------------------
# a whole flow of fused vector Glow steps in one launch per direction (csrc/mlp_chain.hip: k_glow_flow_fwd / _bwd)
# ----------------------------------------------------------------------------------------------------------------------
# whole-flow launches: 'auto' = batches of at most GLOW_FLOW_AUTO_ROWS rows, where they are a measured win over the per-step
# launches with the deferred fold.  Round 3: up to 1024 rows (from 32 workgroups on the in-kernel exchanges were slower than the
# launch gaps they replace: B = 4096 1.91 -> 1.94 ms).  Round 4, after the whole-flow backward got its next-step loads a step ahead
# (tools/probes/flow_rows_sweep.sh, profiles/r04_flow_rows_sweep.txt, ms per train step whole-flow | steps): Glow 2-D B = 2048
# 1.809 | 1.875, 4096 (C2) 1.888 | 1.935, 8192 2.192 | 2.207, 16384 2.820 | 2.780; RealNVP 2-D 2048 1.804 | 1.959, 4096 1.879 | 2.025,
# 16384 2.880 | 2.936.  DESIGN.md section 3.11.  '1' / '0' / 'steps' force a path.
GLOW_FLOW = _os.environ.get('NF_GLOW_FLOW', 'auto')
GLOW_FLOW_AUTO_ROWS = 8192
# larger batches: the same run as ONE autograd node of S single-step launches per direction whose backward defers every
# step's grid barrier + gradient fold to one launch at the end (nf_glow_flow_steps_*; C2 at B = 2048 / 4096 / 16384:
# 1.77 -> 1.63 / 1.88 -> 1.71 / 3.02 -> 2.50 ms per train step)
GLOW_FLOW_STEPS = True          # (internal: deferred fold of the per-step launches)
# the whole-flow backward leaves every step's weight-gradient slabs behind and ONE launch folds them all (nf_*_flow_vec_bwd_deferred):
# the in-kernel fold is 8.4 of a step's 34 us at two workgroups (C1 1.89 -> 1.6 ms per train step); FLOW_DEFER_FOLD = False: in-kernel
FLOW_DEFER_FOLD = True          # (internal: deferred fold inside the whole-flow launch)
# Device tables of per-step pointer records, keyed by the addresses they contain (if a later model lands on the same addresses
# the entry is, by construction, still correct).  A captured hipGraph (FlowTrainer._capture) has the table's address baked into its
# kernel arguments, so an entry that was looked up or created WHILE A STREAM WAS CAPTURING is pinned for the life of the process;
# everything else (eager models that come and go: tests, bench.py running several workloads) is least-recently-used beyond
# 32 entries (~25 KB each for 32 steps).
def _capturing():
    try:
        return bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
    except Exception:
        return False


class _FlowTableCache:
    def __init__(self, limit):
        import collections
        self.limit = int(limit)
        self.entries = collections.OrderedDict()      # key -> [table, pinned]
        self.host = {}                                # device table pointer -> host copy of the same records

    def get(self, key):
        e = self.entries.get(key)
        if e is None:
            return None
        self.entries.move_to_end(key)
        if _capturing():
            e[1] = True
        return e[0]

    def __setitem__(self, key, table):
        self.entries[key] = [table, _capturing()]
        if len(self.entries) > self.limit:
            for k in list(self.entries):
                if len(self.entries) <= self.limit:
                    break
                t, pinned = self.entries[k]
                if not pinned and k != key:
                    del self.entries[k]
                    self.host.pop(t.data_ptr(), None)

    def __len__(self):
        return len(self.entries)


_GLOW_FLOW_TABLES = _FlowTableCache(32)
_GLOW_FLOW_HOST = _GLOW_FLOW_TABLES.host          # device table pointer -> the host copy of the same records
_GLOW_FLOW_SLABS = {}
_GRAPH_SEEN = [False]      # a hipGraph capture has asked for the deferred-fold scratch at least once (only then can a graph hold its address)


def _glow_steps_scratch(S, blocks, device):
    """(slabs_all, head_rec) of the deferred fold: a slab region per step and workgroup, kept across calls."""
    key = ('steps', device)
    n = S * blocks * N.header_constant('NF_MLP_BWD_SLAB_WG_FLOATS')
    t = _GLOW_FLOW_SLABS.get(key)
    if torch.cuda.is_current_stream_capturing():
        _GRAPH_SEEN[0] = True
    if t is None or t[0].numel() < n or t[1].numel() < S * blocks * 64:
        if t is not None and torch.cuda.is_current_stream_capturing() is False and not _GRAPH_SEEN[0]:
            pass                                   # nothing captured yet can hold its address: the outgrown pair is simply freed
        elif t is not None:
            # an outgrown pair may be in the kernel arguments of a captured hipGraph (the whole-flow backward with its deferred fold):
            # it is retired, not freed -- a later replay must not scribble over whatever the allocator placed there
            _GLOW_FLOW_SLABS.setdefault(('retired', device), []).append(t)
        t = _GLOW_FLOW_SLABS[key] = (torch.empty(n, dtype=torch.float32, device=device),
                                     torch.empty(S * blocks * 64, dtype=torch.float32, device=device))
    return t


def _glow_flow_slabs(device):
    t = _GLOW_FLOW_SLABS.get(device)
    if t is None:
        t = _GLOW_FLOW_SLABS[device] = torch.empty(2 * N.header_constant('NF_MLP_BWD_SLAB_FLOATS'), dtype=torch.float32,
                                                   device=device)
    return t


def _glow_step_learnables(head, mlp):
    nl, nb = 6, 5
    learn_h = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
    learn_m = list(mlp[:3 * nl]) + [t for j in range(nb) for t in mlp[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
    return learn_h + learn_m


def _glow_flow_table(steps, sinks, D, device):
    """device array of the steps' pointer records (nf_glow_flow_pack), cached while every pointer stays where it is.
    steps: [(odd, head tensors, mlp tensors)]; sinks: per step the 35 gradient buffers (or None: forward only)."""
    key = (device, D, tuple(int(odd) for odd, _, _ in steps),
           tuple(t.data_ptr() for _, h, m in steps for t in list(h) + list(m)),
           None if sinks is None else tuple(t.data_ptr() for g in sinks for t in g))
    hit = _GLOW_FLOW_TABLES.get(key)
    if hit is not None:
        return hit
    lib = N.load()
    nbytes = int(lib.nf_glow_flow_step_bytes())
    host = (ctypes.c_ubyte * (nbytes * len(steps)))()
    for i, (odd, head, mlp) in enumerate(steps):
        htab, mtab = _ptr_table(head), _ptr_table(mlp)
        if sinks is None:
            hg = mg = None
        else:
            hgt, mgt = _ptr_table(sinks[i][:7]), _ptr_table(sinks[i][7:])
            hg, mg = ctypes.addressof(hgt), ctypes.addressof(mgt)
        N.call('nf_glow_flow_pack', ctypes.addressof(host) + i * nbytes, ctypes.addressof(htab), ctypes.addressof(mtab), hg, mg,
               D, int(odd))
    table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
    _GLOW_FLOW_TABLES[key] = table
    _GLOW_FLOW_HOST[table.data_ptr()] = host
    return table


class _GlowFlowVec(torch.autograd.Function):
    """(y, ld) of S consecutive [ActNorm, InvertibleConv1x1, AffineCoupling] steps on (N, D) data: one launch forward, one
    backward.  tensors: per step the 11 head tensors and the 43 MLP tensors of _GlowStepVec.  Needs the gradient sinks of
    a GradBucket (the Compose peephole checks)."""

    @staticmethod
    def forward(ctx, z, ld, odds, training, per_step, *tensors):
        S = len(odds)
        per = 11 + 43
        steps = [(odds[i], tensors[per * i:per * i + 11], tensors[per * i + 11:per * (i + 1)]) for i in range(S)]
        from .functional import _sinks
        sinks = [_sinks(*_glow_step_learnables(h, m)) for _, h, m in steps]
        if any(g is None for g in sinks):
            raise RuntimeError('glow_flow_vec needs direct gradient sinks (GradBucket) for every parameter')
        z = z.contiguous()
        Nrows, D = z.shape
        dev = z.device
        table = _glow_flow_table(steps, sinks, D, dev)
        ys = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
        saves = torch.empty(S, N.header_constant('NF_GLOW_FLOW_SAVE_FLOATS'), dtype=torch.float32, device=dev)
        ws = WS.zeros(S * N.header_constant('NF_MLP_WS_FLOATS'), dev)
        ctx.host = _GLOW_FLOW_HOST[table.data_ptr()] if per_step else None      # (kept: the cache may be recycled before backward)
        if per_step:
            N.call('nf_glow_flow_steps_fwd', ctypes.addressof(ctx.host), S, N.ptr(z), N.ptr(ys), N.ptr(ld),
                   N.ptr(saves), N.ptr(ws), Nrows, D, int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
        else:
            N.call('nf_glow_flow_vec_fwd', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(ld), N.ptr(saves), N.ptr(ws), Nrows, D,
                   int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
        ctx.save_for_backward(z, ys, saves, table)
        ctx.meta = (S, bool(training), len(tensors), bool(per_step))
        ctx.mark_dirty(ld)
        return ys[S - 1], ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        S, training, n_tensors, per_step = ctx.meta
        z, ys, saves, table = ctx.saved_tensors
        Nrows, D = z.shape
        dev = z.device
        g_y = g_y.contiguous()
        g_ld = None if g_ld is None else g_ld.contiguous()
        gzs = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
        ws = WS.zeros(S * N.header_constant('NF_MLP_WS_FLOATS'), dev)
        if per_step:
            rpb = N.header_constant('NF_MLP_ROWS_PER_BLOCK')
            slabs, rec = _glow_steps_scratch(S, (Nrows + rpb - 1) // rpb, dev)
            N.call('nf_glow_flow_steps_bwd', ctypes.addressof(ctx.host), table.data_ptr(), S, N.ptr(z),
                   N.ptr(ys), N.ptr(g_y), _p(g_ld), N.ptr(gzs), N.ptr(saves), 1, N.ptr(ws), N.ptr(slabs), N.ptr(rec), Nrows, D,
                   int(training), BN_EPS, WN_EPS, N.stream())
        elif FLOW_DEFER_FOLD:                  # whole-flow launch, every step's gradient fold in ONE launch behind it
            rpb = N.header_constant('NF_MLP_ROWS_PER_BLOCK')
            slabs, rec = _glow_steps_scratch(S, (Nrows + rpb - 1) // rpb, dev)
            N.call('nf_glow_flow_vec_bwd_deferred', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(g_y), _p(g_ld), N.ptr(gzs),
                   N.ptr(saves), 1, N.ptr(ws), N.ptr(slabs), N.ptr(rec), Nrows, D, int(training), BN_EPS, WN_EPS, N.stream())
        else:
            N.call('nf_glow_flow_vec_bwd', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(g_y), _p(g_ld), N.ptr(gzs), N.ptr(saves),
                   1, N.ptr(ws), N.ptr(_glow_flow_slabs(dev)), Nrows, D, int(training), BN_EPS, WN_EPS, N.stream())
        return (gzs[0], g_ld, None, None, None) + (None, ) * n_tensors


def glow_flow_vec_usable(z, steps):
    """steps: [(actnorm, conv, coupling)] -- at least two fused-step-capable steps whose parameters all have direct sinks."""
    from .functional import grad_sink
    on = _flow_on(z) or _glow_steps_on(z)
    if not on or len(steps) < 2 or len(steps) > N.header_constant('NF_GLOW_FLOW_MAX_STEPS') or not torch.is_grad_enabled():
        return False
    for a, c, k in steps:
        if not glow_step_vec_usable(z, k.net):
            return False
        head = [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
        if any(grad_sink(t) is None for t in _glow_step_learnables(head, _mlp_tensors(k.net))):
            return False
    return True


def glow_flow_vec(z, ld, steps):
    from .functional import _owned_ld
    tensors = []
    for a, c, k in steps:
        tensors += [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
        tensors += _mlp_tensors(k.net)
    odds = tuple(int(k.odd) for _, _, k in steps)
    return _GlowFlowVec.apply(z, _owned_ld(ld), odds, steps[0][2].net.training, not _flow_on(z), *tensors)


def glow_flow_nograd_usable(z, steps):
    """density evaluation (log_py under no_grad): the run of steps in ONE launch when that launch has no grid exchanges to pay
    for (evaluation-mode conditioners) or the batch is small (training-mode statistics, see GLOW_FLOW)"""
    if torch.is_grad_enabled() or len(steps) < 2 or len(steps) > N.header_constant('NF_GLOW_FLOW_MAX_STEPS') or GLOW_FLOW == '0':
        return False
    training = steps[0][2].net.training
    if training and not _flow_on(z):
        return False
    return all(a.initialized and glow_step_vec_usable(z, k.net) and k.net.training == training for a, c, k in steps)


def glow_flow_vec_nograd(z, ld, steps):
    from .functional import _owned_ld
    z = z.contiguous()
    Nrows, D = z.shape
    dev = z.device
    S = len(steps)
    training = bool(steps[0][2].net.training)
    recs = [(int(k.odd), [t.detach() for t in (a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s,
                                                 k.s_log_scale, k.s_bias)], [t.detach() for t in _mlp_tensors(k.net)])
            for a, c, k in steps]
    table = _glow_flow_table(recs, None, D, dev)
    ld = _owned_ld(ld)
    ys = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
    saves = torch.empty(S, N.header_constant('NF_GLOW_FLOW_SAVE_FLOATS'), dtype=torch.float32, device=dev)
    nws = N.header_constant('NF_MLP_WS_FLOATS')
    ws = WS.zeros(S * nws, dev) if training else torch.empty(S * nws, dtype=torch.float32, device=dev)
    N.call('nf_glow_flow_vec_fwd', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(ld), N.ptr(saves), N.ptr(ws), Nrows, D,
           int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
    return ys[S - 1], ld


# ---- the INVERSE of vector Glow steps (sampling: net.backward), one launch per step or per run --------------------------------
GLOW_INVERSE = True             # (internal: fused inverse / no-grad launches)


def _glow_inverse_head(a, c, k):
    """the 11 head tensors of _GlowStepVec with the pivots' row-swap matrix in the P slot (nf_glow_step_vec_inv)"""
    return [a.log_scale, a.bias, c._pivot_matrix().contiguous(), c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale,
            k.s_bias]


def glow_inverse_usable(y, steps):
    """steps: [(actnorm, conv, coupling)] of fused-step-capable Glow steps on (N, 2 | 4) data; no autograd through the inverse
    (InvertibleConv1x1.backward runs under no_grad as well)."""
    if not (GLOW_INVERSE and steps and len(steps) <= N.header_constant('NF_GLOW_FLOW_MAX_STEPS')):
        return False
    for a, c, k in steps:
        if not (a.initialized and glow_step_vec_usable(y, k.net) and k.net.training == steps[0][2].net.training):
            return False
    return True


def glow_flow_vec_inverse(y, ld, steps):
    """y -> z through the inverses of ``steps`` (given in FORWARD order; applied last to first), ld -= the run's log-det:
    one launch for the whole run in evaluation mode or on small batches (csrc/mlp_chain.hip: k_glow_flow_inv), one per step
    otherwise (k_mlp_chain_fwd<1, true>)."""
    with torch.no_grad():
        y = y.contiguous()
        Nrows, D = y.shape
        dev = y.device
        S = len(steps)
        training = bool(steps[0][2].net.training)
        ld = ld.clone()
        nws = N.header_constant('NF_MLP_WS_FLOATS')
        nsave = N.header_constant('NF_GLOW_FLOW_SAVE_FLOATS')
        one_launch = S >= 2 and (not training or _flow_on(y))       # no exchanges in evaluation mode: nothing to lose
        ws = WS.zeros(S * nws, dev) if training else torch.empty(nws, dtype=torch.float32, device=dev)
        saves = torch.empty(S, nsave, dtype=torch.float32, device=dev)
        if one_launch:
            recs = [(int(k.odd), [t.detach() for t in _glow_inverse_head(a, c, k)], [t.detach() for t in _mlp_tensors(k.net)])
                    for a, c, k in steps]
            table = _glow_flow_table(recs, None, D, dev)
            zs = torch.empty(2, Nrows, D, dtype=torch.float32, device=dev)
            N.call('nf_glow_flow_vec_inv', table.data_ptr(), S, N.ptr(y), N.ptr(zs), N.ptr(ld), N.ptr(saves),
                   N.ptr(ws), Nrows, D, int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
            return zs[0], ld
        cur = y
        for i in range(S - 1, -1, -1):
            a, c, k = steps[i]
            z = torch.empty_like(cur)
            htab = _ptr_table([t.detach() for t in _glow_inverse_head(a, c, k)])
            mtab = _ptr_table([t.detach() for t in _mlp_tensors(k.net)])
            N.call('nf_glow_step_vec_inv', N.ptr(cur), N.ptr(z), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab),
                   N.ptr(saves[i]), N.ptr(ws[i * nws:(i + 1) * nws]) if training else N.ptr(ws), Nrows, D, int(k.odd), int(training),
                   BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
            cur = z
        return cur, ld


def _realnvp_step_learnables(head, mlp):
    nl, nb = 6, 5
    return [head[6], head[7]] + list(mlp[:3 * nl]) + [t for j in range(nb) for t in mlp[3 * nl + 5 * j:3 * nl + 5 * j + 2]]


def _realnvp_flow_table(steps, sinks, D, device):
    """steps: [(odd, flow_bn_eps, flow_bn_momentum, 8 head tensors, 43 MLP tensors)]; sinks: per step the 30 gradient buffers."""
    key = ('realnvp', device, D, tuple((int(o), float(e), float(m)) for o, e, m, _, _ in steps),
           tuple(t.data_ptr() for _, _, _, h, m in steps for t in list(h) + list(m)), tuple(t.data_ptr() for g in sinks for t in g))
    hit = _GLOW_FLOW_TABLES.get(key)
    if hit is not None:
        return hit
    nbytes = int(N.load().nf_glow_flow_step_bytes())
    host = (ctypes.c_ubyte * (nbytes * len(steps)))()
    for i, (odd, eps, mom, head, mlp) in enumerate(steps):
        htab, mtab, mgt = _ptr_table(head), _ptr_table(mlp), _ptr_table(sinks[i][2:])
        N.call('nf_realnvp_flow_pack', ctypes.addressof(host) + i * nbytes, ctypes.addressof(htab), ctypes.addressof(mtab),
               N.ptr(sinks[i][0]), N.ptr(sinks[i][1]), ctypes.addressof(mgt), D, int(odd), float(eps), float(mom))
    table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
    _GLOW_FLOW_TABLES[key] = table
    _GLOW_FLOW_HOST[table.data_ptr()] = host
    return table


class _RealNVPFlowVec(torch.autograd.Function):
    """S consecutive [flow BatchNorm (training, affine=False), AffineCoupling] steps on (N, D) data: one launch per direction.
    tensors: per step the 8 head tensors and the 43 MLP tensors of _RealNVPStepVec."""

    @staticmethod
    def forward(ctx, z, ld, metas, per_step, *tensors):
        S = len(metas)
        per = 8 + 43
        steps = [metas[i] + (tensors[per * i:per * i + 8], tensors[per * i + 8:per * (i + 1)]) for i in range(S)]
        from .functional import _sinks
        sinks = [_sinks(*_realnvp_step_learnables(h, m)) for _, _, _, h, m in steps]
        if any(g is None for g in sinks):
            raise RuntimeError('realnvp_flow_vec needs direct gradient sinks (GradBucket) for every parameter')
        z = z.contiguous()
        Nrows, D = z.shape
        dev = z.device
        table = _realnvp_flow_table(steps, sinks, D, dev)
        ys = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
        # (S statistics records, then -- for the shapes the one-workgroup kernels take -- the stash of BatchNorm inputs: the library says how much)
        saves = torch.empty(S * int(N.load().nf_realnvp_flow_save_floats(Nrows, D)), dtype=torch.float32, device=dev)
        ws = WS.zeros(S * N.header_constant('NF_MLP_WS_FLOATS'), dev)
        ctx.host = _GLOW_FLOW_HOST[table.data_ptr()] if per_step else None
        if per_step:       # one launch per step, the backward's gradient folds deferred to one launch (as for the Glow steps)
            N.call('nf_realnvp_flow_steps_fwd', ctypes.addressof(ctx.host), S, N.ptr(z), N.ptr(ys), N.ptr(ld), N.ptr(saves), N.ptr(ws),
                   Nrows, D, BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
        else:
            N.call('nf_realnvp_flow_vec_fwd', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(ld), N.ptr(saves), N.ptr(ws), Nrows, D,
                   BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
        ctx.save_for_backward(z, ys, saves, table)
        ctx.meta = (S, len(tensors))
        ctx.mark_dirty(ld)
        return ys[S - 1], ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        S, n_tensors = ctx.meta
        z, ys, saves, table = ctx.saved_tensors
        Nrows, D = z.shape
        dev = z.device
        g_y = g_y.contiguous()
        g_ld = None if g_ld is None else g_ld.contiguous()
        gzs = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
        ws = WS.zeros(S * N.header_constant('NF_MLP_WS_FLOATS'), dev)
        if ctx.host is not None:
            rpb = N.header_constant('NF_MLP_ROWS_PER_BLOCK')
            slabs, rec = _glow_steps_scratch(S, (Nrows + rpb - 1) // rpb, dev)
            N.call('nf_realnvp_flow_steps_bwd', ctypes.addressof(ctx.host), table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(g_y),
                   _p(g_ld), N.ptr(gzs), N.ptr(saves), 1, N.ptr(ws), N.ptr(slabs), N.ptr(rec), Nrows, D, BN_EPS, WN_EPS, N.stream())
        elif FLOW_DEFER_FOLD:
            regions = int(N.load().nf_realnvp_flow_bwd_regions(Nrows, D))      # (the one-workgroup kernel leaves its own number of partials)
            slabs, rec = _glow_steps_scratch(S, regions, dev)
            N.call('nf_realnvp_flow_vec_bwd_deferred', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(g_y), _p(g_ld), N.ptr(gzs),
                   N.ptr(saves), 1, N.ptr(ws), N.ptr(slabs), N.ptr(rec), Nrows, D, BN_EPS, WN_EPS, N.stream())
        else:
            N.call('nf_realnvp_flow_vec_bwd', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(g_y), _p(g_ld), N.ptr(gzs), N.ptr(saves),
                   1, N.ptr(ws), N.ptr(_glow_flow_slabs(dev)), Nrows, D, BN_EPS, WN_EPS, N.stream())
        return (gzs[0], g_ld, None, None) + (None, ) * n_tensors


def _flow_on(z):
    return GLOW_FLOW is True or GLOW_FLOW == '1' or (GLOW_FLOW == 'auto' and z.shape[0] <= GLOW_FLOW_AUTO_ROWS)


def _glow_steps_on(z):
    """per-step launches + deferred fold: where the whole-flow launch is off and the slabs are more than two (the atomic fold
    of one or two workgroups has no barrier to defer)."""
    if z.shape[0] <= 2 * N.header_constant('NF_MLP_ROWS_PER_BLOCK'):
        return False
    return GLOW_FLOW == 'steps' or (GLOW_FLOW == 'auto' and GLOW_FLOW_STEPS and not _flow_on(z))


def realnvp_flow_vec_usable(z, steps):
    """steps: [(flow BatchNorm, AffineCoupling)] -- at least two fused-step-capable steps, every parameter with a direct sink."""
    from .functional import grad_sink
    if not (_flow_on(z) or _glow_steps_on(z)) or len(steps) < 2 or len(steps) > N.header_constant('NF_GLOW_FLOW_MAX_STEPS') \
            or not torch.is_grad_enabled():
        return False
    for bn, k in steps:
        if not realnvp_step_vec_usable(z, bn, k.net):
            return False
        head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, k.s_log_scale, k.s_bias]
        if any(grad_sink(t) is None for t in _realnvp_step_learnables(head, _mlp_tensors(k.net))):
            return False
    return True


def realnvp_flow_vec(z, ld, steps):
    from .functional import _owned_ld
    tensors, metas = [], []
    for bn, k in steps:
        tensors += [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, k.s_log_scale, k.s_bias]
        tensors += _mlp_tensors(k.net)
        metas.append((int(k.odd), float(bn.eps), float(bn.momentum)))
    return _RealNVPFlowVec.apply(z, _owned_ld(ld), tuple(metas), not _flow_on(z), *tensors)


FBN_RUNNING, FBN_BATCH_BUFFERS = -1.0, -2.0              # include/nfhip.h: NF_FBN_RUNNING, NF_FBN_BATCH_BUFFERS


# ---- MAF steps in evaluation mode (density evaluation under no_grad) and their inverse (sampling) ---------------------------------
def _maf_struct_ok(z, bn, ar):
    return (z.is_cuda and z.dim() == 2 and z.dtype == torch.float32 and 1 <= z.shape[1] <= 4 and ar.net_s.num_hidden == 3
            and ar.net_s.base_filters == H and ar.net_t.num_hidden == 3 and bn.training == ar.net_s.training == ar.net_t.training
            and 0 < z.shape[0] <= N.maf_max_rows())


def _maf_tables(bn, ar, ms, mt):
    head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, ar.perm, ar.s_log_scale, ar.s_bias]
    made = _made_tensors(ar.net_s, ms) + _made_tensors(ar.net_t, mt)
    return _ptr_table([t.detach() for t in head]), _ptr_table([t.detach() for t in made])


def maf_step_eval_usable(z, bn, ar):
    return GLOW_INVERSE and not torch.is_grad_enabled() and not bn.training and _maf_struct_ok(z, bn, ar)


def maf_step_eval(z, ld, bn, ar):
    """[flow BatchNorm, AutoregressiveTransfrom] in evaluation mode under no_grad: one launch, no grid exchange"""
    from .functional import _owned_ld
    z = z.contiguous()
    Nrows, D = z.shape
    ms = ar.net_s.draw_masks(z.device)                       # same RNG order as the reference: s-net, then t-net
    mt = ar.net_t.draw_masks(z.device)
    htab, mtab = _maf_tables(bn, ar, ms, mt)
    ld = _owned_ld(ld)
    y = torch.empty_like(z)
    ws = torch.empty(N.header_constant('NF_MAF_WS_FLOATS'), dtype=torch.float32, device=z.device)
    save = torch.empty(N.header_constant('NF_MAF_SAVE_FLOATS'), dtype=torch.float32, device=z.device)
    N.call('nf_maf_step_fwd', N.ptr(z), N.ptr(y), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab), N.ptr(save), N.ptr(ws),
           Nrows, D, float(bn.eps), FBN_RUNNING, BN_EPS, N.stream())
    return y, ld


def maf_step_inverse_usable(y, bn, ar):
    """D <= 2: the MADE masks of such nets do not depend on the draw, so the D passes of the reference's inverse (which draws
    anew in every pass) share one set"""
    return GLOW_INVERSE and _maf_struct_ok(y, bn, ar) and y.shape[1] <= 2 and (y.shape[0] > 1 or not bn.training)


def maf_step_inverse(y, ld, bn, ar):
    """the inverse of [flow BatchNorm, AutoregressiveTransfrom] in one launch (maf.py:108-119, modules.py:309-322)"""
    with torch.no_grad():
        y = y.contiguous()
        Nrows, D = y.shape
        for _ in range(D):                                   # the reference draws per pass and net: same RNG consumption
            ms = ar.net_s.draw_masks(y.device)
            mt = ar.net_t.draw_masks(y.device)
        htab, mtab = _maf_tables(bn, ar, ms, mt)
        training = bool(bn.training)
        nws = D * N.header_constant('NF_MAF_WS_FLOATS')
        ws = WS.zeros(nws, y.device) if training else torch.empty(nws, dtype=torch.float32, device=y.device)
        ld = ld.clone()
        z = torch.empty_like(y)
        N.call('nf_maf_step_inv', N.ptr(y), N.ptr(z), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab), N.ptr(ws), Nrows, D,
               int(training), BN_EPS, N.stream())
        return z, ld


# ---- RealNVP runs in evaluation mode (density evaluation under no_grad) and their inverse (sampling) ------------------------------


def _realnvp_const_table(steps, mode, D, device):
    """records of [(flow BatchNorm, AffineCoupling)] whose flow BatchNorm reads constants (mode: FBN_RUNNING | FBN_BATCH_BUFFERS)"""
    heads = [[t.detach() for t in (bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var,
                                   k.s_log_scale, k.s_bias)] for bn, k in steps]
    mlps = [[t.detach() for t in _mlp_tensors(k.net)] for _, k in steps]
    key = ('realnvp-const', mode, device, D, tuple(int(k.odd) for _, k in steps),
           tuple(t.data_ptr() for h, m in zip(heads, mlps) for t in h + m))
    hit = _GLOW_FLOW_TABLES.get(key)
    if hit is not None:
        return hit
    nbytes = int(N.load().nf_glow_flow_step_bytes())
    host = (ctypes.c_ubyte * (nbytes * len(steps)))()
    for i, (bn, k) in enumerate(steps):
        htab, mtab = _ptr_table(heads[i]), _ptr_table(mlps[i])
        N.call('nf_realnvp_flow_pack', ctypes.addressof(host) + i * nbytes, ctypes.addressof(htab), ctypes.addressof(mtab), None, None,
               None, D, int(k.odd), float(bn.eps), float(mode))
    table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
    _GLOW_FLOW_TABLES[key] = table
    return table


def _realnvp_const_usable(z, steps):
    if not (GLOW_INVERSE and steps and len(steps) <= N.header_constant('NF_GLOW_FLOW_MAX_STEPS')):
        return False
    training = steps[0][0].training
    return all(glow_step_vec_usable(z, k.net) and bn.training == training and k.net.training == training and z.shape[0] > 1
               for bn, k in steps)


def realnvp_eval_usable(z, steps):
    """[(flow BatchNorm, AffineCoupling)] in evaluation mode under no_grad: the run in one launch, no grid exchange"""
    return (not torch.is_grad_enabled()) and _realnvp_const_usable(z, steps) and not steps[0][0].training and GLOW_FLOW != '0'


def realnvp_flow_vec_eval(z, ld, steps):
    from .functional import _owned_ld
    z = z.contiguous()
    Nrows, D = z.shape
    dev = z.device
    S = len(steps)
    table = _realnvp_const_table(steps, FBN_RUNNING, D, dev)
    ld = _owned_ld(ld)
    ys = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
    saves = torch.empty(S, N.header_constant('NF_REALNVP_SAVE_FLOATS'), dtype=torch.float32, device=dev)
    ws = torch.empty(N.header_constant('NF_MLP_WS_FLOATS') * S, dtype=torch.float32, device=dev)
    N.call('nf_realnvp_flow_vec_fwd_eval', table.data_ptr(), S, N.ptr(z), N.ptr(ys), N.ptr(ld), N.ptr(saves), N.ptr(ws), Nrows, D,
           BN_EPS, WN_EPS, N.stream())
    return ys[S - 1], ld


def realnvp_inverse_usable(y, steps):
    return _realnvp_const_usable(y, steps)


def realnvp_flow_vec_inverse(y, ld, steps):
    """y -> z through the inverses of ``steps`` (forward order given, applied last to first); training mode: the flow BatchNorm
    inverts with its batch buffers, the conditioner normalises with batch statistics (modules.py:309-322, coupling.py:115-122)."""
    with torch.no_grad():
        y = y.contiguous()
        Nrows, D = y.shape
        dev = y.device
        S = len(steps)
        training = bool(steps[0][0].training)
        ld = ld.clone()
        nws = N.header_constant('NF_MLP_WS_FLOATS')
        saves = torch.empty(S, N.header_constant('NF_REALNVP_SAVE_FLOATS'), dtype=torch.float32, device=dev)
        ws = WS.zeros(S * nws, dev) if training else torch.empty(S * nws, dtype=torch.float32, device=dev)
        if not training or _flow_on(y):
            table = _realnvp_const_table(steps, FBN_BATCH_BUFFERS if training else FBN_RUNNING, D, dev)
            zs = torch.empty(2, Nrows, D, dtype=torch.float32, device=dev)
            N.call('nf_realnvp_flow_vec_inv', table.data_ptr(), S, N.ptr(y), N.ptr(zs), N.ptr(ld), N.ptr(saves), N.ptr(ws), Nrows, D,
                   int(training), BN_EPS, BN_MOMENTUM, WN_EPS, N.stream())
            return zs[0], ld
        cur = y
        for i in range(S - 1, -1, -1):
            bn, k = steps[i]
            z = torch.empty_like(cur)
            htab = _ptr_table([t.detach() for t in (bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean,
                                                    bn.running_var, k.s_log_scale, k.s_bias)])
            mtab = _ptr_table([t.detach() for t in _mlp_tensors(k.net)])
            N.call('nf_realnvp_step_vec_inv', N.ptr(cur), N.ptr(z), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab),
                   N.ptr(saves[i]), N.ptr(ws[i * nws:(i + 1) * nws]), Nrows, D, int(k.odd), int(training), BN_EPS, BN_MOMENTUM, WN_EPS,
                   N.stream())
            cur = z
        return cur, ld


# ----------------------------------------------------------------------------------------------------------------------
# whole Flow++ coupling on vector data: conditioner (strided read of the conditioning half) + mixture-of-logistics coupling
# ----------------------------------------------------------------------------------------------------------------------
FLOWPP_FUSED_BWD = True      # tests flip it to compare against the separate coupling / conditioner backward launches


class _FlowppCouplingVec(torch.autograd.Function):
    """(y, ld) = MixLogAttnCoupling.forward for dims = (D,): 2 launches forward (no gather), 3 backward (the conditioner's
    input gradient is added in place into the coupling's, no scatter / add).  tensors: a_log_scale, a_bias, then the 15
    conditioner tensors of _flowpp_tensors.  post = (log_scale, bias) of the NEXT step's ActNorm (D = 2 only): applied by the
    same coupling launches (nf_flowpp_vec_couple_fwd / _bwd), the pair then returns the ActNorm's output."""

    @staticmethod
    def forward(ctx, z, ld, K, eps, odd, F_, n_post, *tensors):
        post = tensors[:n_post]
        a, c = tensors[n_post:n_post + 2]
        ts = tensors[n_post + 2:]
        z = z.contiguous()
        Nrows, D = z.shape
        I0 = ts[0].shape[1]
        O = ts[13].shape[0]
        sel1 = 1 ^ int(odd)                                   # conditioning half = elements 2 e + sel1 (squeeze.py:68-69)
        params = torch.empty(Nrows, O, dtype=torch.float32, device=z.device)
        y = torch.empty_like(z)
        if D == 2 and K <= 8 and FLOWPP_FUSED_BWD:           # conditioner + coupling (+ next ActNorm) in one launch
            N.call('nf_flowpp_vec_step_fwd', N.ptr(z), *_flowpp_fwd_args(ts, F_), N.ptr(a), N.ptr(c),
                   N.ptr(post[0]) if n_post else None, N.ptr(post[1]) if n_post else None, N.ptr(params), N.ptr(y), N.ptr(ld), K,
                   float(eps), int(odd), Nrows, N.stream())
        else:
            N.call('nf_flowpp_cond_fwd', z.data_ptr() + 4 * sel1, *_flowpp_fwd_args(ts, F_), N.ptr(params), D, 2, Nrows, I0, O,
                   N.stream())
        if D == 2 and K <= 8 and FLOWPP_FUSED_BWD:
            pass
        elif n_post:
            N.call('nf_flowpp_vec_couple_fwd', N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c), N.ptr(post[0]), N.ptr(post[1]),
                   N.ptr(y), N.ptr(ld), K, float(eps), int(odd), Nrows, N.stream())
        else:
            N.call('nf_mixlog_coupling_fwd', N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c), N.ptr(y), N.ptr(ld), K, float(eps),
                   N.SPLIT_1D, int(odd), Nrows, D, 1, 1, N.stream())
        ctx.save_for_backward(z, params, *tensors)
        ctx.meta = (K, float(eps), int(odd), F_, n_post)
        from .functional import _sinks
        ctx.sinks_ac = _sinks(a, c)
        ctx.sinks_net = _sinks(*ts)
        ctx.sinks_post = _sinks(*post) if n_post else None
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, params, *tensors = ctx.saved_tensors
        K, eps, odd, F_, n_post = ctx.meta
        post = tensors[:n_post]
        a, c = tensors[n_post:n_post + 2]
        ts = tensors[n_post + 2:]
        Nrows, D = z.shape
        I0 = ts[0].shape[1]
        O = ts[13].shape[0]
        sel1 = 1 ^ odd
        dev = z.device
        g_y, g_ld = g_y.contiguous(), g_ld.contiguous()
        g_z = torch.empty_like(z)
        if ctx.sinks_ac is not None:
            pa, pc, ga, gc = ctx.sinks_ac[0].data_ptr(), ctx.sinks_ac[1].data_ptr(), None, None
        else:
            g_ac = torch.zeros(2, dtype=torch.float32, device=dev)
            pa, pc = g_ac.data_ptr(), g_ac.data_ptr() + 4
            ga, gc = g_ac[0:1].view_as(a), g_ac[1:2].view_as(c)
        gpost = ()
        pls = pb = None
        if n_post:
            if ctx.sinks_post is not None:
                pls, pb, gpost = ctx.sinks_post[0].data_ptr(), ctx.sinks_post[1].data_ptr(), (None, None)
            else:
                g_n = torch.zeros(2, D, dtype=torch.float32, device=dev)
                pls, pb = g_n[0].data_ptr(), g_n[1].data_ptr()
                gpost = (g_n[0].view_as(post[0]), g_n[1].view_as(post[1]))
        if ctx.sinks_net is not None:
            dst, direct = ctx.sinks_net, True
        else:
            flat = torch.zeros(sum(t.numel() for t in ts), dtype=torch.float32, device=dev)
            dst, o = [], 0
            for t in ts:
                dst.append(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()
            direct = False
        d = [t.data_ptr() for t in dst]
        d[7] += 4 * 2 * F_ * H
        d[8] += 4 * 2 * F_
        if D == 2 and K <= 8 and FLOWPP_FUSED_BWD:
            # the coupling's backward runs inside the conditioner's backward kernel: the (N, 2 + 3K) gradient never exists
            defer = (FPP_DEFER.active and direct and ctx.sinks_ac is not None and (not n_post or ctx.sinks_post is not None))
            wsb = FPP_DEFER.workspace(dev) if defer else flowpp_bwd_workspace(dev)
            N.call('nf_flowpp_vec_step_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(params), *_flowpp_fwd_args(ts, F_),
                   N.ptr(a), N.ptr(c), N.ptr(post[0]) if n_post else None, N.ptr(post[1]) if n_post else None, N.ptr(g_z), *d,
                   pa, pc, pls, pb, N.ptr(wsb), K, eps, odd, Nrows, 1 if defer else 0, N.stream())
            if defer:                                         # the finalize of this step runs with everybody else's (flush)
                desc = FlowppFinDesc(N.ptr(wsb), *d, pa, pc, N.ptr(post[0]) if n_post else None, pls, pb, odd, 0)
                FPP_DEFER.queue.append((desc, K, Nrows, (wsb, post, dst)))
        else:
            g_p = torch.empty_like(params)
            if n_post:
                N.call('nf_flowpp_vec_couple_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c),
                       N.ptr(post[0]), N.ptr(post[1]), N.ptr(g_z), N.ptr(g_p), pa, pc, pls, pb, K, eps, odd, Nrows, N.stream())
            else:
                N.call('nf_mixlog_coupling_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c),
                       N.ptr(g_z), N.ptr(g_p), pa, pc, K, eps, N.SPLIT_1D, odd, Nrows, D, 1, 1, N.stream())
            N.call('nf_flowpp_cond_bwd', z.data_ptr() + 4 * sel1, *_flowpp_fwd_args(ts, F_), N.ptr(g_p), g_z.data_ptr() + 4 * sel1,
                   *d, N.ptr(flowpp_bwd_workspace(dev)), D, 2, D, 2, 1, Nrows, I0, O, N.stream())
        gnet = (None, ) * len(ts) if direct else tuple(dst)
        return (g_z, g_ld, None, None, None, None, None) + gpost + (ga, gc) + gnet


def flowpp_post_actnorm_usable(z, coupling, actnorm):
    """the (coupling, next ActNorm) pair of a 2-feature Flow++ density flow runs as the coupling's own launches"""
    return (z.dim() == 2 and z.shape[1] == 2 and z.shape[0] > 0 and coupling.n_mixtures <= 8 and actnorm.initialized
            and actnorm.log_scale.numel() == 2 and flowpp_cond_fusable(coupling.net, z[:, :1]))


def flowpp_coupling_vec(z, ld, coupling, post=None):
    """MixLogAttnCoupling.forward on (N, D) data with the fused conditioner (see flowpp_cond_fusable); post = the next flow
    step's (initialised) ActNorm, applied in the same launches when flowpp_post_actnorm_usable."""
    from .functional import _owned_ld
    ts, F_ = _flowpp_tensors(coupling.net)
    extra = () if post is None else (post.log_scale, post.bias)
    return _FlowppCouplingVec.apply(z, _owned_ld(ld), coupling.n_mixtures, coupling.logit_eps, int(coupling.odd), F_,
                                    len(extra), *extra, coupling.a_log_scale, coupling.a_bias, *ts)


# ----------------------------------------------------------------------------------------------------------------------
# one whole MAF flow step on vector data: flow BatchNorm -> permutation -> MADE pair -> affine transform
# ----------------------------------------------------------------------------------------------------------------------
def maf_step_usable(z, bn, ar):
    """training-mode [flow BatchNorm(affine=False), AutoregressiveTransfrom] on (N, D <= 4) data, csrc/made_chain.hip."""
    return (z.is_cuda and z.dim() == 2 and z.dtype == torch.float32 and 1 <= z.shape[1] <= 4 and bn.training
            and ar.net_s.training and ar.net_s.num_hidden == 3 and ar.net_s.base_filters == H and ar.net_t.num_hidden == 3
            and 1 < z.shape[0] <= N.maf_max_rows())


_MAF_SLABS = {}


def _maf_slabs(device):
    t = _MAF_SLABS.get(device)
    if t is None:
        t = _MAF_SLABS[device] = torch.empty(N.header_constant('NF_MAF_BWD_SLAB_FLOATS'), dtype=torch.float32, device=device)
    return t


def _made_tensors(net, masks):
    ts = []
    for l in range(net.num_hidden + 1):
        ts += [net.weights[l], masks[l], net.biases[l]]
    for bn in net.bnorms:
        ts += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
    return ts


def _made_learnables(ts):
    """(weight, bias) * 4 then (gamma, beta) * 3 of one net's 27-tensor list."""
    out = []
    for l in range(4):
        out += [ts[3 * l], ts[3 * l + 2]]
    for j in range(3):
        out += [ts[12 + 5 * j], ts[12 + 5 * j + 1]]
    return out


class _MAFStepVec(torch.autograd.Function):
    """head: flow-BN log_gamma, beta, batch_mean, batch_var, running_mean, running_var, perm, s_log_scale, s_bias;
    then the 27 tensors of net s and of net t (weight, mask, bias per layer; BatchNorm1d tensors)."""

    @staticmethod
    def forward(ctx, z, ld, bn_eps, bn_momentum, *tensors):
        head, made = tensors[:9], tensors[9:]
        z = z.contiguous()
        Nrows, D = z.shape
        y = torch.empty_like(z)
        save = torch.empty(N.header_constant('NF_MAF_SAVE_FLOATS'), dtype=torch.float32, device=z.device)
        ws = WS.zeros(N.header_constant('NF_MAF_WS_FLOATS'), z.device)
        htab, mtab = _ptr_table(head), _ptr_table(made)
        N.call('nf_maf_step_fwd', N.ptr(z), N.ptr(y), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab), N.ptr(save),
               N.ptr(ws), Nrows, D, float(bn_eps), float(bn_momentum), BN_EPS, N.stream())
        ctx.save_for_backward(z, save, *tensors)
        from .functional import _sinks
        learn = _made_learnables(made[:27]) + _made_learnables(made[27:]) + [head[7], head[8]]
        ctx.sinks = _sinks(*learn)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, save, *tensors = ctx.saved_tensors
        head, made = tensors[:9], tensors[9:]
        Nrows, D = z.shape
        dev = z.device
        g_y = g_y.contiguous()
        g_ld = None if g_ld is None else g_ld.contiguous()
        learn = _made_learnables(made[:27]) + _made_learnables(made[27:]) + [head[7], head[8]]
        direct = ctx.sinks is not None
        dst = ctx.sinks if direct else [torch.zeros_like(t) for t in learn]      # the kernel accumulates by atomics
        g_z = torch.empty_like(z)
        ws = WS.zeros(N.header_constant('NF_MAF_WS_FLOATS'), dev)
        htab, mtab, gtab = _ptr_table(head), _ptr_table(made), _ptr_table(dst[:28])
        N.call('nf_maf_step_bwd', N.ptr(z), N.ptr(g_y), _p(g_ld), N.ptr(g_z), ctypes.addressof(htab), ctypes.addressof(mtab),
               N.ptr(save), ctypes.addressof(gtab), N.ptr(dst[28]), N.ptr(dst[29]), N.ptr(ws), N.ptr(_maf_slabs(dev)), Nrows, D,
               N.stream())
        if direct:
            return (g_z, g_ld, None, None) + (None, ) * len(tensors)
        gh = [None] * 7 + [dst[28], dst[29]]
        gm = []
        for n in range(2):
            d = dst[14 * n:14 * n + 14]
            for l in range(4):
                gm += [d[2 * l], None, d[2 * l + 1]]
            for j in range(3):
                gm += [d[8 + 2 * j], d[8 + 2 * j + 1], None, None, None]
        return (g_z, g_ld, None, None) + tuple(gh) + tuple(gm)


def maf_step_vec(z, ld, bn, ar):
    """[flow BatchNorm ``bn`` (training, affine=False), AutoregressiveTransfrom ``ar``] on (N, D) data, fused."""
    from .functional import _owned_ld
    ms = ar.net_s.draw_masks(z.device)                       # same RNG order as the reference: s-net, then t-net
    mt = ar.net_t.draw_masks(z.device)
    head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, ar.perm, ar.s_log_scale,
            ar.s_bias]
    return _MAFStepVec.apply(z, _owned_ld(ld), bn.eps, bn.momentum, *(head + _made_tensors(ar.net_s, ms) + _made_tensors(ar.net_t, mt)))


# a run of MAF steps as ONE autograd node: the step launches of _MAFStepVec, the backward's in deferred-fold mode -- every step
# leaves its weight-gradient slabs behind and nf_maf_fold_all folds them all in one launch (C5: 0.725 -> 0.628 ms per train step)
MAF_FLOW = _os.environ.get('NF_MAF_FLOW', '1') != '0'


def _maf_steps_scratch(S, blocks, device):
    key = ('maf_steps', device)
    n = S * blocks * N.header_constant('NF_MAF_SLAB_WG_FLOATS')
    m = S * blocks * N.header_constant('NF_MAF_HEAD_REC_WG')
    t = _MAF_SLABS.get(key)
    if t is None or t[0].numel() < n or t[1].numel() < m:
        t = _MAF_SLABS[key] = (torch.empty(n, dtype=torch.float32, device=device), torch.empty(m, dtype=torch.float32, device=device))
    return t


def _maf_step_learnables(head, made):
    return _made_learnables(made[:27]) + _made_learnables(made[27:]) + [head[7], head[8]]


class _MAFFlowVec(torch.autograd.Function):
    """S consecutive [flow BatchNorm (training, affine=False), AutoregressiveTransfrom] steps on (N, D) data.
    metas: per step (flow_bn_eps, flow_bn_momentum); tensors: per step the 9 head + 54 MADE tensors of _MAFStepVec."""

    @staticmethod
    def forward(ctx, z, ld, metas, *tensors):
        S, per = len(metas), 9 + 54
        from .functional import _sinks
        z = z.contiguous()
        Nrows, D = z.shape
        dev = z.device
        ys = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
        saves = torch.empty(S, N.header_constant('NF_MAF_SAVE_FLOATS'), dtype=torch.float32, device=dev)
        nws = N.header_constant('NF_MAF_WS_FLOATS')
        ws = WS.zeros(S * nws, dev)
        sinks = []
        for i in range(S):
            head, made = tensors[per * i:per * i + 9], tensors[per * i + 9:per * (i + 1)]
            g = _sinks(*_maf_step_learnables(head, made))
            if g is None:
                raise RuntimeError('maf_flow_vec needs direct gradient sinks (GradBucket) for every parameter')
            sinks.append(g)
            htab, mtab = _ptr_table(head), _ptr_table(made)
            N.call('nf_maf_step_fwd', N.ptr(z) if i == 0 else N.ptr(ys[i - 1]), N.ptr(ys[i]), N.ptr(ld), ctypes.addressof(htab),
                   ctypes.addressof(mtab), N.ptr(saves[i]), N.ptr(ws[i * nws:(i + 1) * nws]), Nrows, D, float(metas[i][0]),
                   float(metas[i][1]), BN_EPS, N.stream())
        ctx.save_for_backward(z, ys, saves, *tensors)
        ctx.sinks = sinks
        ctx.mark_dirty(ld)
        return ys[S - 1], ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, ys, saves, *tensors = ctx.saved_tensors
        S, per = ys.shape[0], 9 + 54
        Nrows, D = z.shape
        dev = z.device
        g_y = g_y.contiguous()
        g_ld = None if g_ld is None else g_ld.contiguous()
        gzs = torch.empty(S, Nrows, D, dtype=torch.float32, device=dev)
        nws = N.header_constant('NF_MAF_WS_FLOATS')
        ws = WS.zeros(S * nws, dev)
        blocks = -(-Nrows // N.header_constant('NF_MAF_ROWS_PER_BLOCK'))
        nsl, nrec = blocks * N.header_constant('NF_MAF_SLAB_WG_FLOATS'), blocks * N.header_constant('NF_MAF_HEAD_REC_WG')
        slabs, rec = _maf_steps_scratch(S, blocks, dev)
        all_made, all_grads, all_a, all_c = [], [], [], []
        for i in range(S - 1, -1, -1):
            head, made = tensors[per * i:per * i + 9], tensors[per * i + 9:per * (i + 1)]
            dst = ctx.sinks[i]
            htab, mtab, gtab = _ptr_table(head), _ptr_table(made), _ptr_table(dst[:28])
            N.call('nf_maf_step_bwd_partial', N.ptr(z) if i == 0 else N.ptr(ys[i - 1]), N.ptr(g_y) if i == S - 1 else N.ptr(gzs[i + 1]),
                   _p(g_ld), N.ptr(gzs[i]), ctypes.addressof(htab), ctypes.addressof(mtab), N.ptr(saves[i]), ctypes.addressof(gtab),
                   N.ptr(ws[i * nws:(i + 1) * nws]), N.ptr(slabs[i * nsl:(i + 1) * nsl]), N.ptr(rec[i * nrec:(i + 1) * nrec]), Nrows, D,
                   N.stream())
        for i in range(S):
            head, made = tensors[per * i:per * i + 9], tensors[per * i + 9:per * (i + 1)]
            all_made += list(made)
            all_grads += list(ctx.sinks[i][:28])
            all_a.append(ctx.sinks[i][28])
            all_c.append(ctx.sinks[i][29])
        pm, pg, pa, pc = _ptr_table(all_made), _ptr_table(all_grads), _ptr_table(all_a), _ptr_table(all_c)
        N.call('nf_maf_fold_all', ctypes.addressof(pm), ctypes.addressof(pg), ctypes.addressof(pa), ctypes.addressof(pc), S, N.ptr(slabs),
               N.ptr(rec), blocks, D, N.stream())
        return (gzs[0], g_ld, None) + (None, ) * len(tensors)


def maf_flow_vec_usable(z, steps):
    """steps: [(flow BatchNorm, AutoregressiveTransfrom)] -- at least two fused-step-capable steps, every parameter with a sink."""
    from .functional import grad_sink
    if not MAF_FLOW or len(steps) < 2 or not torch.is_grad_enabled():
        return False
    for bn, ar in steps:
        if not maf_step_usable(z, bn, ar):
            return False
        ps = [p for net in (ar.net_s, ar.net_t) for p in list(net.weights) + list(net.biases)]
        ps += [t for net in (ar.net_s, ar.net_t) for b in net.bnorms for t in (b.weight, b.bias)] + [ar.s_log_scale, ar.s_bias]
        if any(grad_sink(t) is None for t in ps):
            return False
    return True


def maf_flow_vec(z, ld, steps):
    from .functional import _owned_ld
    tensors, metas = [], []
    for bn, ar in steps:
        ms = ar.net_s.draw_masks(z.device)                   # same RNG order as the reference: per step s-net, then t-net
        mt = ar.net_t.draw_masks(z.device)
        tensors += [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, ar.perm, ar.s_log_scale,
                    ar.s_bias] + _made_tensors(ar.net_s, ms) + _made_tensors(ar.net_t, mt)
        metas.append((float(bn.eps), float(bn.momentum)))
    return _MAFFlowVec.apply(z, _owned_ld(ld), tuple(metas), *tensors)


# ----------------------------------------------------------------------------------------------------------------------
# one whole RealNVP flow step on vector data: flow BatchNorm (batch statistics) -> affine coupling with the MLP conditioner
# ----------------------------------------------------------------------------------------------------------------------
def realnvp_step_vec_usable(z, bn, mlp):
    return (glow_step_vec_usable(z, mlp) and bn.training and mlp.training and z.shape[0] > 1)


class _RealNVPStepVec(torch.autograd.Function):
    """head: flow-BN log_gamma, beta, batch_mean, batch_var, running_mean, running_var, s_log_scale, s_bias; then the 43 MLP
    tensors."""

    @staticmethod
    def forward(ctx, z, ld, odd, bn_eps, bn_momentum, *tensors):
        head, mlp = tensors[:8], tensors[8:]
        z = z.contiguous()
        Nrows, D = z.shape
        y = torch.empty_like(z)
        save = torch.empty(N.header_constant('NF_REALNVP_SAVE_FLOATS'), dtype=torch.float32, device=z.device)
        ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), z.device)
        htab, mtab = _ptr_table(head), _ptr_table(mlp)
        N.call('nf_realnvp_step_vec_fwd', N.ptr(z), N.ptr(y), N.ptr(ld), ctypes.addressof(htab), ctypes.addressof(mtab),
               N.ptr(save), N.ptr(ws), Nrows, D, int(odd), float(bn_eps), float(bn_momentum), BN_EPS, BN_MOMENTUM, WN_EPS,
               N.stream())
        ctx.save_for_backward(z, save, *tensors)
        ctx.odd = int(odd)
        from .functional import _sinks
        nl, nb = 6, 5
        learn = [head[6], head[7]] + list(mlp[:3 * nl]) + [t for j in range(nb) for t in mlp[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
        ctx.sinks = _sinks(*learn)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        nl, nb = 6, 5
        z, save, *tensors = ctx.saved_tensors
        head, mlp = tensors[:8], tensors[8:]
        Nrows, D = z.shape
        dev = z.device
        g_y = g_y.contiguous()
        g_ld = None if g_ld is None else g_ld.contiguous()
        learn = [head[6], head[7]] + list(mlp[:3 * nl]) + [t for j in range(nb) for t in mlp[3 * nl + 5 * j:3 * nl + 5 * j + 2]]
        direct = ctx.sinks is not None
        dst = ctx.sinks if direct else [torch.empty_like(t) for t in learn]
        g_z = torch.empty_like(z)
        ws = WS.zeros(N.header_constant('NF_MLP_WS_FLOATS'), dev)
        htab, mtab, mg = _ptr_table(head), _ptr_table(mlp), _ptr_table(dst[2:])
        N.call('nf_realnvp_step_vec_bwd', N.ptr(z), N.ptr(g_y), _p(g_ld), N.ptr(g_z), ctypes.addressof(htab),
               ctypes.addressof(mtab), N.ptr(save), N.ptr(dst[0]), N.ptr(dst[1]), ctypes.addressof(mg), int(direct), N.ptr(ws),
               N.ptr(_mlp_slabs(dev)), Nrows, D, ctx.odd, BN_EPS, WN_EPS, N.stream())
        if direct:
            return (g_z, g_ld, None, None, None) + (None, ) * len(tensors)
        gh = [None] * 6 + [dst[0], dst[1]]
        gm = list(dst[2:2 + 3 * nl])
        for j in range(nb):
            gm += [dst[2 + 3 * nl + 2 * j], dst[2 + 3 * nl + 2 * j + 1], None, None, None]
        return (g_z, g_ld, None, None, None) + tuple(gh) + tuple(gm)


def realnvp_step_vec(z, ld, bn, coupling):
    """[flow BatchNorm ``bn`` (training, affine=False), AffineCoupling ``coupling``] on (N, D) data, fused."""
    from .functional import _owned_ld
    head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, coupling.s_log_scale,
            coupling.s_bias]
    return _RealNVPStepVec.apply(z, _owned_ld(ld), int(coupling.odd), bn.eps, bn.momentum, *(head + _mlp_tensors(coupling.net)))


# ----------------------------------------------------------------------------------------------------------------------
# weight normalisation of every weight-normed layer of a model in a handful of launches (csrc/weight_norm.hip)
# ----------------------------------------------------------------------------------------------------------------------
WN_MAX = N.header_constant("NF_WN_MAX_LAYERS")      # layers per weight-norm launch (round 6: 896, one launch for a CIFAR Glow's 966 is two)


class _WeightNormMulti(torch.autograd.Function):
    """(w_0, w_1, ...) = weight_norm(v_k, g_k) for tensors = v_0, g_0, v_1, g_1, ...; ceil(n / 64) launches each way."""

    @staticmethod
    def forward(ctx, eps, *tensors):
        vs, gs = tensors[0::2], tensors[1::2]
        outs = [torch.empty_like(v) for v in vs]
        for k0 in range(0, len(vs), WN_MAX):
            descs = [_desc(WnDesc, v=vs[k], g=gs[k], w=outs[k], O=vs[k].shape[0], M=vs[k].numel() // vs[k].shape[0])
                     for k in range(k0, min(k0 + WN_MAX, len(vs)))]
            arr = (WnDesc * len(descs))(*descs)
            N.call('nf_weight_norm_fwd', ctypes.addressof(arr), len(descs), float(eps), N.stream())
        ctx.save_for_backward(*tensors)
        ctx.eps = float(eps)
        from .functional import _sinks
        ctx.sinks = _sinks(*tensors)
        from .fused_conv import CONV_DEFER
        CONV_DEFER.arm(outs)                      # consumers may hand back gradients that are filled when backward() flushes
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_ws):
        from .fused_conv import CONV_DEFER
        CONV_DEFER.flush()                        # the deferred weight-gradient passes of the image conditioners fill g_ws
        tensors = ctx.saved_tensors
        vs, gs = tensors[0::2], tensors[1::2]
        direct = ctx.sinks is not None
        dst = ctx.sinks if direct else [torch.empty_like(t) for t in tensors]
        live = [k for k in range(len(vs)) if g_ws[k] is not None]
        gws = {k: g_ws[k].contiguous() for k in live}
        for k0 in range(0, len(live), WN_MAX):
            ks = live[k0:k0 + WN_MAX]
            descs = [_desc(WnDesc, v=vs[k], g=gs[k], g_w=gws[k], g_v=dst[2 * k], g_g=dst[2 * k + 1], O=vs[k].shape[0],
                           M=vs[k].numel() // vs[k].shape[0], accumulate=int(direct)) for k in ks]
            arr = (WnDesc * len(descs))(*descs)
            N.call('nf_weight_norm_bwd', ctypes.addressof(arr), len(descs), ctx.eps, N.stream())
        if direct:
            return (None, ) + (None, ) * len(tensors)
        out = []
        for k in range(len(vs)):
            out += [dst[2 * k], dst[2 * k + 1]] if k in gws else [None, None]
        return (None, ) + tuple(out)


def weight_norm_all(wn_modules):
    """effective weights of the given conditioners.WeightNorm modules, stashed on them for the forward pass under way."""
    if not wn_modules:
        return
    eps = wn_modules[0].eps
    tensors = []
    for m in wn_modules:
        tensors += [m.module.weight_v, m.module.weight_g]
    outs = _WeightNormMulti.apply(eps, *tensors)
    for m, w in zip(wn_modules, outs):
        m._w_eff = w
    from .fused_conv import pack_conv_weights
    pack_conv_weights(wn_modules, outs)               # the chain kernels' LDS images of these weights (three-way bf16 split)


# ----------------------------------------------------------------------------------------------------------------------
# the PLU weight W = P L' U' of every invertible 1x1 convolution of a model, and its autograd, in batched launches
# ----------------------------------------------------------------------------------------------------------------------
class PluDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in ('P', 'L', 'U', 'L_mask', 'U_mask', 'sign_s', 'log_s', 'W', 'g_W', 'g_ld', 'g_L',
                                                'g_U', 'g_log_s')] + \
               [('B', ctypes.c_int64), ('C', ctypes.c_int), ('accumulate', ctypes.c_int), ('pixels', ctypes.c_float),
                ('reserved', ctypes.c_int)]


def _plu_launch(name, descs):
    step = N.header_constant('NF_PLU_MAX_LAYERS')
    for k0 in range(0, len(descs), step):
        chunk = descs[k0:k0 + step]
        arr = (PluDesc * len(chunk))(*chunk)
        N.call(name, ctypes.addressof(arr), len(chunk), N.stream())


class _PLUWeightsMulti(torch.autograd.Function):
    """W_i = P_i (L_i o mask + I) (U_i o mask + diag(sign_s exp(log_s))) for n layers; tensors: per layer P, L, U, L_mask, U_mask,
    sign_s, log_s.  The backward needs every layer's log-det gradient: the applications (functional._InvConvApplyW) leave it
    in ``holder``."""

    @staticmethod
    def forward(ctx, holder, *tensors):
        n = len(tensors) // 7
        Ws, descs = [], []
        for i in range(n):
            P, L, U, Lm, Um, sg, ls = tensors[7 * i:7 * i + 7]
            W = torch.empty_like(L)
            Ws.append(W)
            descs.append(_desc(PluDesc, P=P, L=L, U=U, L_mask=Lm, U_mask=Um, sign_s=sg, log_s=ls, W=W, C=L.shape[0]))
        _plu_launch('nf_invconv_weight_fwd_multi', descs)
        ctx.save_for_backward(*tensors)
        ctx.holder = holder
        from .functional import _sinks
        ctx.sinks = [_sinks(tensors[7 * i + 1], tensors[7 * i + 2], tensors[7 * i + 6]) for i in range(n)]
        return tuple(Ws)

    @staticmethod
    def backward(ctx, *g_Ws):
        tensors = ctx.saved_tensors
        n = len(tensors) // 7
        holder = ctx.holder
        from .functional import flush_head_params
        flush_head_params(holder)                 # the heads' deferred parameter gradients fill g_Ws
        descs, grads = [], [None]
        for i in range(n):
            P, L, U, Lm, Um, sg, ls = tensors[7 * i:7 * i + 7]
            if g_Ws[i] is None or holder.meta[i] is None:
                grads += [None] * 7
                continue
            B, Px = holder.meta[i]
            direct = ctx.sinks[i] is not None
            gL, gU, gls = ctx.sinks[i] if direct else (torch.empty_like(L), torch.empty_like(U), torch.empty_like(ls))
            descs.append(_desc(PluDesc, P=P, L=L, U=U, L_mask=Lm, U_mask=Um, sign_s=sg, log_s=ls, g_W=g_Ws[i].contiguous(),
                               g_ld=holder.g_ld[i], g_L=gL, g_U=gU, g_log_s=gls, B=B, C=L.shape[0], accumulate=int(direct),
                               pixels=float(Px)))
            grads += [None, None, None, None, None, None, None] if direct else [None, gL, gU, None, None, None, gls]
        if descs:
            _plu_launch('nf_invconv_weight_bwd_multi', descs)
        return tuple(grads)


def plu_weights_all(convs):
    """W of the given layers.InvertibleConv1x1 modules in batched launches, stashed on them (with the holder the batched
    backward reads) for the forward pass under way."""
    if not convs:
        return
    from .functional import PluHolder
    holder = PluHolder(len(convs))
    tensors = []
    for c in convs:
        tensors += [c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s]
    Ws = _PLUWeightsMulti.apply(holder, *tensors)
    for i, (c, W) in enumerate(zip(convs, Ws)):
        c._W_eff = (W, holder, i)

