"""
torch.autograd.Function wrappers over the C ABI of libnfhip.so -- the only place where the product path touches
the kernels.  Forward-direction ops carry hand-written autograd (backward kernels); inverse-direction ops
(sampling, ``layer.backward`` in the reference's naming) run without building a graph, like the reference's
own ``torch.no_grad`` inverse of the invertible 1x1 convolution (flows/modules.py:485).

``ld`` (``log_df_dz``) is updated IN PLACE and returned, as the reference does (coupling.py:110,
modules.py:249,305,480); only the returned tensor is contractual.
"""
import ctypes

import torch

from . import _native as N
from . import workspace as WS

MODE_OF = {'1d': N.SPLIT_1D, 'checkerboard': N.SPLIT_CHECKER, 'channelwise': N.SPLIT_CHANNEL, 'none': N.SPLIT_NONE}


def _bchw(z):
    """(B, C, H, W) view of the problem: 2-D data is (B, D, 1, 1)."""
    if z.dim() == 2:
        return z.shape[0], z.shape[1], 1, 1
    if z.dim() == 4:
        return tuple(z.shape)
    raise ValueError('flow tensors are (B, D) or (B, C, H, W), got shape %s' % (tuple(z.shape), ))


def _half_shape(z, mode):
    B, C, H, W = _bchw(z)
    if mode == N.SPLIT_1D:
        return (B, C // 2)
    if mode == N.SPLIT_CHECKER:
        return (B, 2 * C, H // 2, W // 2)
    if mode == N.SPLIT_CHANNEL:
        return (B, C // 2, H, W)
    return tuple(z.shape)


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


def grad_sink(p):
    """the buffer a hand-written backward may accumulate into directly instead of returning a temporary: ``p.grad`` of a
    parameter that a GradBucket tagged (``_nf_direct_grad``).  Same ``+=`` semantics as autograd's AccumulateGrad, one
    launch less per parameter.  None -> return the gradient to autograd as usual."""
    if not getattr(p, '_nf_direct_grad', False):
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.device != p.device:
        return None
    return g


def _sinks(*params):
    """all-or-nothing list of gradient sinks for the parameters of one op."""
    out = [grad_sink(p) for p in params]
    return out if all(t is not None for t in out) else None


def _owned_ld(ld, *others):
    """ld is mutated in place; a leaf that requires grad (never the case in the models) gets copied first."""
    if ld.requires_grad and ld.is_leaf:
        return ld.clone()
    return ld


# ----------------------------------------------------------------------------------------------------------------------
# index maps
# ----------------------------------------------------------------------------------------------------------------------
class _HalfGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, which, mode, odd):
        z = _contig(z)
        B, C, H, W = _bchw(z)
        out = torch.empty(_half_shape(z, mode), dtype=z.dtype, device=z.device)
        N.call('nf_half_gather', N.ptr(z), N.ptr(out), which, mode, int(odd), B, C, H, W, N.stream())
        ctx.meta = (which, mode, int(odd), tuple(z.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        which, mode, odd, shape = ctx.meta
        g = _contig(g)
        full = torch.empty(shape, dtype=g.dtype, device=g.device)
        B, C, H, W = _bchw(full)
        N.call('nf_half_scatter', N.ptr(g), N.ptr(full), which, mode, odd, B, C, H, W, N.stream())
        return full, None, None, None


def half_gather(z, which, mode, odd):
    """contiguous z0 (which=0, transformed half) or z1 (which=1, conditioning half) of the split map."""
    return _HalfGather.apply(z, which, mode, odd)


class _SpaceDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, to_depth):
        z = _contig(z)
        B, C, H, W = z.shape
        if to_depth:
            out = torch.empty((B, 4 * C, H // 2, W // 2), dtype=z.dtype, device=z.device)
            N.call('nf_squeeze2d', N.ptr(z), N.ptr(out), B, C, H, W, N.stream())
        else:
            out = torch.empty((B, C // 4, 2 * H, 2 * W), dtype=z.dtype, device=z.device)
            N.call('nf_unsqueeze2d', N.ptr(z), N.ptr(out), B, C // 4, 2 * H, 2 * W, N.stream())
        ctx.to_depth = to_depth
        return out

    @staticmethod
    def backward(ctx, g):
        return _SpaceDepth.apply(g, not ctx.to_depth), None


def squeeze2d(z):
    if z.dim() != 4 or z.shape[2] % 2 or z.shape[3] % 2:
        raise ValueError('squeeze2d needs (B, C, even H, even W)')
    return _SpaceDepth.apply(z, True)


def unsqueeze2d(z):
    if z.dim() != 4 or z.shape[1] % 4:
        raise ValueError('unsqueeze2d needs (B, 4C, h, w)')
    return _SpaceDepth.apply(z, False)


# ----------------------------------------------------------------------------------------------------------------------
# affine coupling
# ----------------------------------------------------------------------------------------------------------------------
class _AffineCoupling(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, t, s_raw, pbs, a, c, ld, mode, odd):
        B, C, H, W = _bchw(z)
        y = torch.empty_like(z)
        N.call('nf_affine_coupling_fwd', N.ptr(z), N.ptr(t), N.ptr(s_raw), pbs, N.ptr(a), N.ptr(c), N.ptr(y),
               N.ptr(ld), mode, int(odd), 0, B, C, H, W, N.stream())
        ctx.save_for_backward(z, t, s_raw, a, c)
        ctx.meta = (pbs, mode, int(odd))
        ctx.sinks = _sinks(a, c)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, t, s_raw, a, c = ctx.saved_tensors
        pbs, mode, odd = ctx.meta
        B, C, H, W = _bchw(z)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_z = torch.empty_like(z)
        g_t = torch.empty_like(t)
        g_s = torch.empty_like(s_raw)
        if ctx.sinks is not None:
            pa, pc, ga, gc = ctx.sinks[0].data_ptr(), ctx.sinks[1].data_ptr(), None, None
        else:
            g_ac = WS.zeros(2, z.device)
            pa, pc = g_ac.data_ptr(), g_ac.data_ptr() + 4
            ga, gc = g_ac[0:1].view_as(a), g_ac[1:2].view_as(c)
        N.call('nf_affine_coupling_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(t), N.ptr(s_raw), pbs, N.ptr(a),
               N.ptr(c), N.ptr(g_z), N.ptr(g_t), N.ptr(g_s), pa, pc, mode, odd, B, C, H, W, N.stream())
        return g_z, g_t, g_s, None, ga, gc, g_ld, None, None


class _AffineCouplingPacked(torch.autograd.Function):
    """conditioner output packed as one tensor (B, 2*Ch, h, w): t = params[:, :Ch], s_raw = params[:, Ch:]."""

    @staticmethod
    def forward(ctx, z, params, a, c, ld, mode, odd):
        B, C, H, W = _bchw(z)
        n_half = params.numel() // (2 * B) if B else 0
        y = torch.empty_like(z)
        N.call('nf_affine_coupling_fwd', N.ptr(z), N.ptr(params), params.data_ptr() + 4 * n_half, 2 * n_half, N.ptr(a),
               N.ptr(c), N.ptr(y), N.ptr(ld), mode, int(odd), 0, B, C, H, W, N.stream())
        ctx.save_for_backward(z, params, a, c)
        ctx.meta = (n_half, mode, int(odd))
        ctx.sinks = _sinks(a, c)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, params, a, c = ctx.saved_tensors
        n_half, mode, odd = ctx.meta
        B, C, H, W = _bchw(z)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_z = torch.empty_like(z)
        g_p = torch.empty_like(params)
        if ctx.sinks is not None:
            pa, pc, ga, gc = ctx.sinks[0].data_ptr(), ctx.sinks[1].data_ptr(), None, None
        else:
            g_ac = WS.zeros(2, z.device)
            pa, pc = g_ac.data_ptr(), g_ac.data_ptr() + 4
            ga, gc = g_ac[0:1].view_as(a), g_ac[1:2].view_as(c)
        N.call('nf_affine_coupling_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(params),
               params.data_ptr() + 4 * n_half, 2 * n_half, N.ptr(a), N.ptr(c), N.ptr(g_z), N.ptr(g_p),
               g_p.data_ptr() + 4 * n_half, pa, pc, mode, odd, B, C, H, W, N.stream())
        return g_z, g_p, ga, gc, g_ld, None, None


def affine_coupling(z, params, s_log_scale, s_bias, ld, mode, odd, inverse=False):
    """AbstractCoupling.forward/backward around AffineCoupling._transform/_inverse_transform, given ``params`` =
    conditioner(z1).  (flows/coupling.py:32-43, :104-122)"""
    z, params = _contig(z), _contig(params)
    half = _half_shape(z, mode)
    if tuple(params.shape) != (half[0], 2 * half[1]) + tuple(half[2:]):
        raise ValueError('conditioner output has shape %s, expected %s' % (tuple(params.shape),
                                                                         (half[0], 2 * half[1]) + tuple(half[2:])))
    ld = _owned_ld(ld)
    if not inverse:
        return _AffineCouplingPacked.apply(z, params, s_log_scale, s_bias, ld, mode, odd)
    with torch.no_grad():
        B, C, H, W = _bchw(z)
        n_half = params.numel() // (2 * B) if B else 0
        y = torch.empty_like(z)
        N.call('nf_affine_coupling_fwd', N.ptr(z), N.ptr(params), params.data_ptr() + 4 * n_half, 2 * n_half,
               N.ptr(s_log_scale), N.ptr(s_bias), N.ptr(y), N.ptr(ld), mode, int(odd), 1, B, C, H, W, N.stream())
    return y, ld


def affine_transform(z, s_raw, t, s_log_scale, s_bias, ld, inverse=False):
    """un-split affine transform with separate scale / shift tensors (MAF: flows/maf.py:103-106, :114-115)."""
    z, s_raw, t = _contig(z), _contig(s_raw), _contig(t)
    ld = _owned_ld(ld)
    pbs = s_raw.numel() // s_raw.shape[0] if s_raw.shape[0] else 0
    if not inverse:
        return _AffineCoupling.apply(z, t, s_raw, pbs, s_log_scale, s_bias, ld, N.SPLIT_NONE, False)
    with torch.no_grad():
        B, C, H, W = _bchw(z)
        y = torch.empty_like(z)
        N.call('nf_affine_coupling_fwd', N.ptr(z), N.ptr(t), N.ptr(s_raw), pbs, N.ptr(s_log_scale), N.ptr(s_bias),
               N.ptr(y), N.ptr(ld), N.SPLIT_NONE, 0, 1, B, C, H, W, N.stream())
    return y, ld


# ----------------------------------------------------------------------------------------------------------------------
# per-channel affine bijectors (ActNorm, flow BatchNorm) and their statistics
# ----------------------------------------------------------------------------------------------------------------------
def _bcp(x):
    B, C = x.shape[0], x.shape[1]
    P = 1
    for d in x.shape[2:]:
        P *= int(d)
    return B, C, P


class _ChanAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, x, ld, p0, p1, p2, p3):
        B, C, P = _bcp(x)
        y = torch.empty_like(x)
        N.call('nf_chan_affine_fwd', op, N.ptr(x), N.ptr(p0), N.ptr(p1), N.ptr(p2), N.ptr(p3), N.ptr(y), N.ptr(ld), 0,
               B, C, P, N.stream())
        ctx.op = op
        ctx.save_for_backward(x, p0, p1, p2, p3)
        ctx.sinks = _sinks(p0, p1) if op == N.OP_ACTNORM else (_sinks(p2, p3) if p2 is not None else None)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        x, p0, p1, p2, p3 = ctx.saved_tensors
        op = ctx.op
        B, C, P = _bcp(x)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_x = torch.empty_like(x)
        pa, pb = (p0, p1) if op == N.OP_ACTNORM else (p2, p3)
        want = ctx.needs_input_grad[3 if op == N.OP_ACTNORM else 5]
        ga = gb = None
        if want and ctx.sinks is not None:
            ga_ptr, gb_ptr = ctx.sinks[0].data_ptr(), ctx.sinks[1].data_ptr()
        elif want:
            g_ab = WS.zeros(2 * pa.numel(), x.device).view((2, ) + tuple(pa.shape))
            ga_ptr, gb_ptr = g_ab.data_ptr(), g_ab.data_ptr() + 4 * pa.numel()
            ga, gb = g_ab[0], g_ab[1]
        else:
            ga_ptr, gb_ptr = None, None
        N.call('nf_chan_affine_bwd', op, N.ptr(g_y), N.ptr(g_ld), N.ptr(x), N.ptr(p0), N.ptr(p1), N.ptr(p2), N.ptr(p3),
               N.ptr(g_x), ga_ptr, gb_ptr, B, C, P, N.stream())
        if op == N.OP_ACTNORM:
            return None, g_x, g_ld, ga, gb, None, None
        return None, g_x, g_ld, None, None, ga, gb


def chan_affine(op, x, ld, p0, p1, p2=None, p3=None, inverse=False):
    x = _contig(x)
    ld = _owned_ld(ld)
    if not inverse:
        return _ChanAffine.apply(op, x, ld, p0, p1, p2, p3)
    with torch.no_grad():
        B, C, P = _bcp(x)
        y = torch.empty_like(x)
        N.call('nf_chan_affine_fwd', op, N.ptr(x), N.ptr(p0), N.ptr(p1), N.ptr(p2), N.ptr(p3), N.ptr(y), N.ptr(ld), 1,
               B, C, P, N.stream())
    return y, ld


def chan_stats(x):
    """per-channel (sum, sum of squared deviations) over batch and pixels, two passes (no E[x^2]-E[x]^2)."""
    x = _contig(x)
    B, C, P = _bcp(x)
    st = torch.zeros((2, C), dtype=x.dtype, device=x.device)
    N.call('nf_chan_sum', N.ptr(x), st.data_ptr(), B, C, P, N.stream())
    N.call('nf_chan_sqdev', N.ptr(x), st.data_ptr(), st.data_ptr() + 4 * C, B, C, P, N.stream())
    return st, B * P


def actnorm_init_(x, log_scale, bias, eps):
    """data-dependent ActNorm initialisation, writes the parameters in place (flows/modules.py:238-244)."""
    with torch.no_grad():
        st, n = chan_stats(x)
        C = st.shape[1]
        N.call('nf_actnorm_init_finalize', st.data_ptr(), st.data_ptr() + 4 * C, N.ptr(log_scale.data),
               N.ptr(bias.data), float(eps), n, C, N.stream())


def flowbn_update_(x, batch_mean, batch_var, running_mean, running_var, eps, momentum):
    """train-mode statistics of the flow BatchNorm, all buffers written in place (flows/modules.py:284-294)."""
    with torch.no_grad():
        st, n = chan_stats(x)
        C = st.shape[1]
        N.call('nf_flowbn_finalize', st.data_ptr(), st.data_ptr() + 4 * C, N.ptr(batch_mean), N.ptr(batch_var),
               N.ptr(running_mean), N.ptr(running_var), float(eps), float(momentum), n, C, N.stream())


# ----------------------------------------------------------------------------------------------------------------------
# invertible 1x1 convolution
# ----------------------------------------------------------------------------------------------------------------------
class _InvConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, W, ld, log_s):
        B, C, P = _bcp(z)
        y = torch.empty_like(z)
        N.call('nf_invconv_apply', N.ptr(z), N.ptr(W), 0, N.ptr(y), N.ptr(ld), N.ptr(log_s), 1.0, B, C, P, N.stream())
        ctx.save_for_backward(z, W)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, W = ctx.saved_tensors
        B, C, P = _bcp(z)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_z = g_W = g_ls = None
        if ctx.needs_input_grad[0]:
            g_z = torch.empty_like(z)
            N.call('nf_invconv_apply', N.ptr(g_y), N.ptr(W), 1, N.ptr(g_z), None, None, 0.0, B, C, P, N.stream())
        if ctx.needs_input_grad[1]:
            g_W = torch.zeros_like(W)
            N.call('nf_invconv_wgrad', N.ptr(g_y), N.ptr(z), N.ptr(g_W), B, C, P, N.stream())
        if ctx.needs_input_grad[3]:
            g_ls = (g_ld.sum() * float(P)).expand(C)
        return g_z, g_W, g_ld, g_ls


def invconv(z, W, ld, log_s):
    """forward 1x1 convolution y = W z per pixel, ld += P * sum(log_s)  (flows/modules.py:475-480)."""
    return _InvConv.apply(_contig(z), _contig(W), _owned_ld(ld), _contig(log_s))


def invconv_inverse(y, W_inv, ld, log_s):
    """z = W^-1 y per pixel, ld -= P * sum(log_s)  (flows/modules.py:484-497), no autograd like the reference."""
    with torch.no_grad():
        y, W_inv = _contig(y), _contig(W_inv)
        ld = _owned_ld(ld)
        B, C, P = _bcp(y)
        z = torch.empty_like(y)
        N.call('nf_invconv_apply', N.ptr(y), N.ptr(W_inv), 0, N.ptr(z), N.ptr(ld), N.ptr(_contig(log_s)), -1.0, B, C, P,
               N.stream())
    return z, ld


class _InvConvPLU(torch.autograd.Function):
    """whole InvertibleConv1x1.forward: PLU assembly + per-pixel mat-vec + log-det, 2 launches forward, 4 backward."""

    @staticmethod
    def forward(ctx, z, ld, P, L, U, L_mask, U_mask, sign_s, log_s):
        B, C, Px = _bcp(z)
        W = torch.empty((C, C), dtype=z.dtype, device=z.device)
        N.call('nf_invconv_weight_fwd', N.ptr(P), N.ptr(L), N.ptr(U), N.ptr(L_mask), N.ptr(U_mask), N.ptr(sign_s),
               N.ptr(log_s), N.ptr(W), C, N.stream())
        y = torch.empty_like(z)
        N.call('nf_invconv_apply', N.ptr(z), N.ptr(W), 0, N.ptr(y), N.ptr(ld), N.ptr(log_s), 1.0, B, C, Px, N.stream())
        ctx.save_for_backward(z, W, P, L, U, L_mask, U_mask, sign_s, log_s)
        ctx.sinks = _sinks(L, U, log_s)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, W, P, L, U, L_mask, U_mask, sign_s, log_s = ctx.saved_tensors
        B, C, Px = _bcp(z)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_z = None
        if ctx.needs_input_grad[0]:
            g_z = torch.empty_like(z)
            N.call('nf_invconv_apply', N.ptr(g_y), N.ptr(W), 1, N.ptr(g_z), None, None, 0.0, B, C, Px, N.stream())
        g_W = WS.zeros(W.numel(), W.device).view_as(W)
        N.call('nf_invconv_wgrad', N.ptr(g_y), N.ptr(z), N.ptr(g_W), B, C, Px, N.stream())
        direct = ctx.sinks is not None
        g_L, g_U, g_ls = ctx.sinks if direct else (torch.empty_like(L), torch.empty_like(U), torch.empty_like(log_s))
        N.call('nf_invconv_weight_bwd', N.ptr(g_W), N.ptr(P), N.ptr(L), N.ptr(U), N.ptr(L_mask), N.ptr(U_mask),
               N.ptr(sign_s), N.ptr(log_s), N.ptr(g_ld), N.ptr(g_L), N.ptr(g_U), N.ptr(g_ls), int(direct), C, B, Px,
               N.stream())
        if direct:
            return g_z, g_ld, None, None, None, None, None, None, None
        return g_z, g_ld, None, g_L, g_U, None, None, None, g_ls


PLU_MAX_C = 64


class PluHolder:
    """what the per-layer applications leave for the batched PLU backward of the same pass: their incoming log-det gradient
    (g_log_s has a pixels * sum_b g_ld term, modules.py:480) and their batch / pixel counts."""

    def __init__(self, n):
        self.g_ld = [None] * n
        self.meta = [None] * n
        self.pending = []       # heads whose parameter gradients wait for the batched launch (_GlowHeadW.backward -> flush_head_params)


class _InvConvApplyW(torch.autograd.Function):
    """InvertibleConv1x1.forward with the weight W = P L' U' computed elsewhere (fused.plu_weights_all: every layer of the model
    in a few launches): per-pixel mat-vec + log-det forward, transposed mat-vec + weight gradient backward."""

    @staticmethod
    def forward(ctx, z, ld, W, log_s, holder, idx):
        B, C, Px = _bcp(z)
        y = torch.empty_like(z)
        N.call('nf_invconv_apply', N.ptr(z), N.ptr(W), 0, N.ptr(y), N.ptr(ld), N.ptr(log_s), 1.0, B, C, Px, N.stream())
        ctx.save_for_backward(z, W)
        ctx.holder, ctx.idx = holder, idx
        holder.meta[idx] = (B, Px)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, W = ctx.saved_tensors
        B, C, Px = _bcp(z)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_z = None
        if ctx.needs_input_grad[0]:
            g_z = torch.empty_like(z)
            N.call('nf_invconv_apply', N.ptr(g_y), N.ptr(W), 1, N.ptr(g_z), None, None, 0.0, B, C, Px, N.stream())
        g_W = WS.zeros(W.numel(), W.device).view_as(W)
        N.call('nf_invconv_wgrad', N.ptr(g_y), N.ptr(z), N.ptr(g_W), B, C, Px, N.stream())
        ctx.holder.g_ld[ctx.idx] = g_ld
        return g_z, g_ld, g_W, None, None, None


def invconv_apply_w(z, ld, W, log_s, holder, idx):
    return _InvConvApplyW.apply(_contig(z), _owned_ld(ld), W, log_s, holder, idx)


def invconv_plu(z, ld, P, L, U, L_mask, U_mask, sign_s, log_s):
    """InvertibleConv1x1.forward from its stored PLU parameters (flows/modules.py:470-482)."""
    return _InvConvPLU.apply(_contig(z), _owned_ld(ld), P, L, U, L_mask, U_mask, sign_s, log_s)


FLOWBN_FUSED = True           # (internal: image data's flow BatchNorm head in one persistent launch; False = statistics + apply launches)


class _FlowBNHead(torch.autograd.Function):
    """training-mode flow BatchNorm (affine=False) + optional conditioning-half gather: 2 launches forward, 1 backward."""

    @staticmethod
    def forward(ctx, x, ld, log_gamma, beta, batch_mean, batch_var, running_mean, running_var, eps, momentum, mode, odd,
                gather):
        B, C, H, W = _bchw(x)
        y = torch.empty_like(x)
        z1c = torch.empty(_half_shape(x, mode), dtype=x.dtype, device=x.device) if gather else None
        nws = int(N.load().nf_flowbn_head_fused_ws_floats(B, C, H, W)) if (FLOWBN_FUSED and x.dim() == 4) else 0
        if nws > 0:
            # statistics and apply in ONE persistent launch: the workgroups exchange their per-channel sums (csrc/flowbn_head.hip)
            ws = WS.zeros(nws, x.device)
            N.call('nf_flowbn_head_fused', N.ptr(x), N.ptr(log_gamma), N.ptr(beta), N.ptr(batch_mean), N.ptr(batch_var),
                   N.ptr(running_mean), N.ptr(running_var), float(eps), float(momentum), N.ptr(y), N.ptr(z1c), N.ptr(ld), N.ptr(ws),
                   mode, int(odd), B, C, H, W, N.stream())
        else:
            ws = WS.zeros(3 * C, x.device)
            N.call('nf_flowbn_stats', N.ptr(x), N.ptr(running_mean), N.ptr(ws), B, C, H * W, N.stream())
            N.call('nf_flowbn_head_fwd', N.ptr(x), N.ptr(ws), N.ptr(log_gamma), N.ptr(beta), N.ptr(batch_mean),
                   N.ptr(batch_var), N.ptr(running_mean), N.ptr(running_var), float(eps), float(momentum), N.ptr(y),
                   N.ptr(z1c), N.ptr(ld), mode, int(odd), B, C, H, W, N.stream())
        ctx.save_for_backward(batch_var, log_gamma)
        ctx.meta = (mode, int(odd), tuple(x.shape), gather)
        ctx.mark_dirty(ld)
        ctx.set_materialize_grads(False)          # the gathered half has no gradient of its own under the fused image coupling
        if gather:
            return y, z1c, ld
        return y, ld

    @staticmethod
    def backward(ctx, *grads):
        batch_var, log_gamma = ctx.saved_tensors
        mode, odd, shape, gather = ctx.meta
        if gather:
            g_h, g_z1c, g_ld = grads
            g_z1c = _contig(g_z1c) if g_z1c is not None else None
        else:
            (g_h, g_ld), g_z1c = grads, None
        if g_h is None:                            # (grads are not materialised: an unused output arrives as None)
            g_h = torch.zeros(shape, dtype=batch_var.dtype, device=batch_var.device)
        g_h = _contig(g_h)
        B, C, H, W = shape if len(shape) == 4 else (shape[0], shape[1], 1, 1)
        g_x = torch.empty_like(g_h)
        N.call('nf_flowbn_head_bwd', N.ptr(g_h), N.ptr(g_z1c), N.ptr(batch_var), N.ptr(log_gamma), N.ptr(g_x), mode, odd,
               B, C, H, W, N.stream())
        return (g_x, g_ld) + (None, ) * 11


def flowbn_head(x, ld, bn, mode=N.SPLIT_NONE, odd=False, gather=False):
    """train-mode ``BatchNorm.forward`` of a flow BatchNorm with affine=False (+ the split gather of the coupling that
    follows): (y, [z1c,] ld).  flows/modules.py:283-307, flows/coupling.py:33."""
    return _FlowBNHead.apply(_contig(x), _owned_ld(ld), bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var,
                             bn.running_mean, bn.running_var, bn.eps, bn.momentum, mode, odd, gather)


class _GlowHead(torch.autograd.Function):
    """ActNorm + invertible 1x1 (PLU assembled in-kernel) + conditioning-half gather: 1 launch forward, 2 backward."""

    @staticmethod
    def forward(ctx, z, ld, log_scale, bias, P, L, U, L_mask, U_mask, sign_s, log_s, mode, odd, bwd_defer=False, defer=False):
        B, C, H, W = _bchw(z)
        ctx.bwd_defer = bool(bwd_defer)
        h = torch.empty_like(z)
        z1c = torch.empty(_half_shape(z, mode), dtype=z.dtype, device=z.device)
        Wm = torch.empty((C, C), dtype=z.dtype, device=z.device)
        if defer:
            # (as in _GlowHeadW.forward: the coupling's chain launch performs this head in its prologue -- nf_cc_head_small_fwd)
            PENDING_HEADS[h.data_ptr()] = (z, log_scale, bias, None, log_s, h, z1c, ld, mode, int(odd), (P, L, U, L_mask, U_mask, sign_s, Wm))
        else:
            N.call('nf_glow_head_fwd', N.ptr(z), N.ptr(log_scale), N.ptr(bias), N.ptr(P), N.ptr(L), N.ptr(U), N.ptr(L_mask),
                   N.ptr(U_mask), N.ptr(sign_s), N.ptr(log_s), N.ptr(h), N.ptr(z1c), N.ptr(Wm), N.ptr(ld), mode, int(odd), B,
                   C, H, W, N.stream())
        ctx.save_for_backward(z, log_scale, bias, Wm, P, L, U, L_mask, U_mask, sign_s, log_s)
        ctx.meta = (mode, int(odd))
        ctx.sinks = _sinks(log_scale, bias, L, U, log_s)
        ctx.mark_dirty(ld)
        ctx.set_materialize_grads(False)          # the gathered half has no gradient of its own under the fused image coupling
        return h, z1c, ld

    @staticmethod
    def backward(ctx, g_h, g_z1c, g_ld):
        z, log_scale, bias, Wm, P, L, U, L_mask, U_mask, sign_s, log_s = ctx.saved_tensors
        mode, odd = ctx.meta
        B, C, H, W = _bchw(z)
        if g_h is None:                            # (grads are not materialised: an unused output arrives as None)
            g_h = torch.zeros_like(z)
        if g_ld is None:
            g_ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        g_h, g_z1c, g_ld = _contig(g_h), (_contig(g_z1c) if g_z1c is not None else None), _contig(g_ld)
        g_z = torch.empty_like(z)
        direct = ctx.sinks is not None
        tmp = WS.zeros(C * C + 4 + (0 if direct else 2 * C), z.device)
        g_W, sum_gld = tmp[:C * C], tmp[C * C:C * C + 1]
        if direct:
            g_ls, g_b, g_L, g_U, g_logs = ctx.sinks
        else:
            g_ls, g_b = tmp[C * C + 4:C * C + 4 + C].view_as(log_scale), tmp[C * C + 4 + C:].view_as(bias)
            g_L, g_U, g_logs = torch.empty_like(L), torch.empty_like(U), torch.empty_like(log_s)
        from .fused_conv import CONV_DEFER
        if direct and CONV_DEFER.active and HEAD_PARAMS_DEFER:
            # the data gradient now; the parameter sums of all such heads in one launch at the end-of-pass flush, in front of their PLU jobs
            if ctx.bwd_defer and HEAD_BWD_IN_CHAIN and g_z1c is None and z.dim() == 4 and 2 <= C <= 4:
                # (as in _GlowHeadW.backward: the previous step's chain launch computes g_z in its prologue -- nf_cc_head_small_bwd)
                PENDING_HEAD_BWD[g_z.data_ptr()] = (g_h, log_scale, Wm, g_z, (mode, odd))
            else:
                N.call('nf_glow_head_bwd_data', N.ptr(g_h), N.ptr(g_z1c), N.ptr(log_scale), N.ptr(Wm), N.ptr(g_z), mode, odd, B, C, H, W, N.stream())
            CONV_DEFER.head_jobs.append(((mode, B, C, H, W), g_h, g_z1c, g_ld, z, log_scale, bias, Wm, g_ls, g_b, g_W, sum_gld, int(odd)))
        else:
            N.call('nf_glow_head_bwd', N.ptr(g_h), N.ptr(g_z1c), N.ptr(g_ld), N.ptr(z), N.ptr(log_scale), N.ptr(bias),
                   N.ptr(Wm), N.ptr(g_z), N.ptr(g_ls), N.ptr(g_b), N.ptr(g_W), N.ptr(sum_gld), mode, odd, B, C, H, W,
                   N.stream())
        if direct and CONV_DEFER.active:
            # inside a trainer step the PLU backward (8 us per layer on the backward pass's latency chain, nothing but Adam waits for
            # it) joins the end-of-pass flush: all queued layers in one nf_invconv_weight_bwd_multi launch per 24
            CONV_DEFER.plu_jobs.append((g_W, P, L, U, L_mask, U_mask, sign_s, log_s, sum_gld, g_L, g_U, g_logs, C, H * W))
        else:
            N.call('nf_invconv_weight_bwd', N.ptr(g_W), N.ptr(P), N.ptr(L), N.ptr(U), N.ptr(L_mask), N.ptr(U_mask),
                   N.ptr(sign_s), N.ptr(log_s), N.ptr(sum_gld), N.ptr(g_L), N.ptr(g_U), N.ptr(g_logs), int(direct), C, 1,
                   H * W, N.stream())
        if direct:
            return (g_z, g_ld) + (None, ) * 13
        return g_z, g_ld, g_ls, g_b, None, g_L, g_U, None, None, None, g_logs, None, None, None, None


class _GlowHeadW(torch.autograd.Function):
    """ActNorm + invertible 1x1 with its weight W given assembled (fused.plu_weights_all) + conditioning-half gather on 9 .. 64
    channels of image data: one MFMA launch per direction (csrc/glow_head_mfma.hip)."""

    @staticmethod
    def forward(ctx, x, ld, log_scale, bias, W, log_s, holder, idx, mode, odd, defer=False, bwd_defer=False):
        B, C, H, Wd = x.shape
        ctx.bwd_defer = bool(bwd_defer)
        h = torch.empty_like(x)
        z1c = torch.empty(_half_shape(x, mode), dtype=x.dtype, device=x.device)
        if defer:
            # the launch is left to the coupling's chain kernel, whose prologue does this head's work (csrc/conv_chain.hip:
            # nf_cc_head_fwd): it finds the operands under the address of h, which it receives as its z
            PENDING_HEADS[h.data_ptr()] = (x, log_scale, bias, W, log_s, h, z1c, ld, mode, int(odd), None)
        else:
            N.call('nf_glow_head_w_fwd', N.ptr(x), N.ptr(log_scale), N.ptr(bias), N.ptr(W), N.ptr(log_s), N.ptr(h), N.ptr(z1c),
                   N.ptr(ld), mode, int(odd), B, C, H, Wd, N.stream())
        ctx.save_for_backward(x, log_scale, bias, W)
        ctx.holder, ctx.idx, ctx.meta = holder, idx, (mode, int(odd))
        holder.meta[idx] = (B, H * Wd)
        ctx.sinks = _sinks(log_scale, bias)
        ctx.mark_dirty(ld)
        ctx.set_materialize_grads(False)          # the gathered half has no gradient of its own under the fused image coupling
        return h, z1c, ld

    @staticmethod
    def backward(ctx, g_h, g_z1c, g_ld):
        x, log_scale, bias, W = ctx.saved_tensors
        mode, odd = ctx.meta
        B, C, H, Wd = x.shape
        if g_h is None:
            g_h = torch.zeros_like(x)
        if g_ld is None:
            g_ld = torch.zeros(B, dtype=x.dtype, device=x.device)
        g_h, g_ld = _contig(g_h), _contig(g_ld)
        if g_z1c is not None:                       # (a coupling that was not fused into its conditioner's launches: image Flow++)
            full = torch.empty_like(x)
            N.call('nf_half_scatter_add', N.ptr(_contig(g_z1c)), N.ptr(g_h), N.ptr(full), 1, mode, odd, B, C, H, Wd, N.stream())
            g_h = full
        g_x = torch.empty_like(x)
        direct = ctx.sinks is not None
        tmp = WS.zeros(C * C + (0 if direct else 2 * C), x.device)
        g_W = tmp[:C * C].view_as(W)
        if HEAD_PARAMS_DEFER and direct and ctx.needs_input_grad[4]:
            # only g_x is on the way of the backward pass: the contractions over the batch (g_W, g_log_scale, g_bias) of all heads of one
            # shape run in one launch where the pass ends -- in front of the PLU backward that reads g_W (fused._PLUWeightsMulti.backward).
            # What they read is kept alive here: g_h, g_ld (nothing writes to a gradient autograd has handed over), x.
            from .fused_conv import CONV_DEFER
            if ctx.bwd_defer and HEAD_BWD_IN_CHAIN and CONV_DEFER.active:
                # x is the output of a fused image coupling: g_x is the g_y of that coupling's chain launch, which autograd runs next and
                # which computes it in its own prologue (csrc/conv_chain.hip: nf_cc_head_bwd) -- no launch here.  Inside a trainer step
                # only: the flush at the end of the pass performs whatever was not picked up (fused_conv.ConvDefer.flush).
                PENDING_HEAD_BWD[g_x.data_ptr()] = (g_h, log_scale, W, g_x, None)
            else:
                N.call('nf_glow_head_w_bwd_data', N.ptr(g_h), N.ptr(log_scale), N.ptr(W), N.ptr(g_x), B, C, H, Wd, N.stream())
            ctx.holder.pending.append(((B, C, H, Wd), g_h, g_ld, x, log_scale, bias, W, ctx.sinks[0], ctx.sinks[1], g_W))
            ctx.holder.g_ld[ctx.idx] = g_ld
            return g_x, g_ld, None, None, g_W, None, None, None, None, None, None, None
        if direct:
            p_ls, p_b, g_ls, g_b = ctx.sinks[0].data_ptr(), ctx.sinks[1].data_ptr(), None, None
        else:
            g_ls, g_b = tmp[C * C:C * C + C].view_as(log_scale), tmp[C * C + C:].view_as(bias)
            p_ls, p_b = g_ls.data_ptr(), g_b.data_ptr()
        N.call('nf_glow_head_w_bwd', N.ptr(g_h), N.ptr(g_ld), N.ptr(x), N.ptr(log_scale), N.ptr(bias), N.ptr(W), N.ptr(g_x), p_ls,
               p_b, N.ptr(g_W), B, C, H, Wd, N.stream())
        ctx.holder.g_ld[ctx.idx] = g_ld
        return g_x, g_ld, g_ls, g_b, g_W, None, None, None, None, None, None, None


HEAD_PARAMS_DEFER = True      # (internal constant: tests flip it to compare the two forms of the head backward)
HEAD_BWD_IN_CHAIN = True      # (internal: the head's data gradient in the prologue of the previous coupling's backward chain launch)
PENDING_HEAD_BWD = {}         # address of g_x -> (g_h, log_scale, W, g_x, small) of a head whose data gradient the next chain launch computes
                              # (small: None for the MFMA head, (mode, odd) for the 2 .. 4 channel head with its saved weight)


def flush_pending_head_bwd(entry):
    """the data gradient of a head left to a chain launch (``entry`` of PENDING_HEAD_BWD) on its own kernel after all"""
    g_h, log_scale, W, g_x, small = entry
    B, C, H, Wd = g_x.shape
    if small is None:
        N.call('nf_glow_head_w_bwd_data', N.ptr(g_h), N.ptr(log_scale), N.ptr(W), N.ptr(g_x), B, C, H, Wd, N.stream())
    else:
        N.call('nf_glow_head_bwd_data', N.ptr(g_h), None, N.ptr(log_scale), N.ptr(W), N.ptr(g_x), small[0], small[1], B, C, H, Wd, N.stream())


def flush_all_pending_head_bwd():
    """end of a trainer's backward pass: nothing may be left.  An entry that no chain launch picked up means the gradient autograd handed
    to the coupling was not the tensor the head returned (the coupling's output has a second consumer, so autograd summed two gradients
    first; or only part of the graph was differentiated): the pass consumed a tensor that had not been computed -- loud, not silent."""
    n = len(PENDING_HEAD_BWD)
    while PENDING_HEAD_BWD:
        flush_pending_head_bwd(PENDING_HEAD_BWD.pop(next(iter(PENDING_HEAD_BWD))))
    if n:
        raise RuntimeError('%d Glow head(s) left their data gradient to the previous coupling\'s backward launch, which never asked for it: '
                           'the output of a fused image coupling must feed the next flow step only (functional.HEAD_BWD_IN_CHAIN = False '
                           'restores the stand-alone launches)' % n)


class GlowHeadParamsDesc(ctypes.Structure):
    """include/nfhip.h: nf_glow_head_params_desc"""
    _fields_ = [(f, ctypes.c_void_p) for f in ('g_h', 'g_ld', 'x', 'act_log_scale', 'act_bias', 'W', 'g_log_scale', 'g_bias', 'g_W')]


class GlowHeadSmallParamsDesc(ctypes.Structure):
    """include/nfhip.h: nf_glow_head_small_params_desc"""
    _fields_ = [(f, ctypes.c_void_p) for f in ('g_h', 'g_z1c', 'g_ld', 'z', 'log_scale', 'bias', 'W_saved', 'g_log_scale', 'g_bias', 'g_W',
                                                'sum_g_ld')] + [('odd', ctypes.c_int), ('reserved', ctypes.c_int)]


def launch_small_head_params(jobs):
    """the parameter sums of the C <= 4 heads queued by _GlowHead.backward (fused_conv.ConvDefer.flush)"""
    step = N.header_constant('NF_GLOW_HEAD_MULTI_MAX')
    groups = {}
    for e in jobs:
        groups.setdefault(e[0], []).append(e)
    for (mode, B, C, H, W), es in groups.items():
        for k0 in range(0, len(es), step):
            chunk = es[k0:k0 + step]
            arr = (GlowHeadSmallParamsDesc * len(chunk))()
            for i, (_, g_h, g_z1c, g_ld, z, ls, bs, Wm, g_ls, g_b, g_W, sum_gld, odd) in enumerate(chunk):
                d = arr[i]
                d.g_h, d.g_z1c, d.g_ld, d.z = g_h.data_ptr(), (g_z1c.data_ptr() if g_z1c is not None else None), g_ld.data_ptr(), z.data_ptr()
                d.log_scale, d.bias, d.W_saved = ls.data_ptr(), bs.data_ptr(), Wm.data_ptr()
                d.g_log_scale, d.g_bias, d.g_W, d.sum_g_ld, d.odd = g_ls.data_ptr(), g_b.data_ptr(), g_W.data_ptr(), sum_gld.data_ptr(), odd
            N.call('nf_glow_head_bwd_params_multi', ctypes.addressof(arr), len(chunk), mode, B, C, H, W, N.stream())


def flush_head_params(holder):
    """the parameter gradients of the heads queued on ``holder``: heads of one shape share launches of NF_GLOW_HEAD_MULTI_MAX"""
    pend, holder.pending = holder.pending, []
    if not pend:
        return
    step = N.header_constant('NF_GLOW_HEAD_MULTI_MAX')
    groups = {}
    for e in pend:
        groups.setdefault(e[0], []).append(e)
    for (B, C, H, Wd), es in groups.items():
        for k0 in range(0, len(es), step):
            chunk = es[k0:k0 + step]
            arr = (GlowHeadParamsDesc * len(chunk))()
            for i, (_, g_h, g_ld, x, ls, bs, W, s_ls, s_b, g_W) in enumerate(chunk):
                d = arr[i]
                d.g_h, d.g_ld, d.x, d.act_log_scale, d.act_bias, d.W = g_h.data_ptr(), g_ld.data_ptr(), x.data_ptr(), ls.data_ptr(), bs.data_ptr(), W.data_ptr()
                d.g_log_scale, d.g_bias, d.g_W = s_ls.data_ptr(), s_b.data_ptr(), g_W.data_ptr()
            N.call('nf_glow_head_w_bwd_params_multi', ctypes.addressof(arr), len(chunk), B, C, H, Wd, N.stream())


PENDING_HEADS = {}      # address of h -> operands of a head forward that its coupling's chain launch performs (_GlowHeadW.forward(defer=True))


def flush_pending_head(h):
    """launch the head whose output buffer is ``h`` on its own kernel, if it is still pending (a coupling that did not take the chain
    launch after all); returns True if there was one"""
    pend = PENDING_HEADS.pop(h.data_ptr(), None)
    if pend is None:
        return False
    x, log_scale, bias, W, log_s, h_, z1c, ld, mode, odd, small = pend
    B, C, H, Wd = x.shape
    if small is None:
        N.call('nf_glow_head_w_fwd', N.ptr(x), N.ptr(log_scale), N.ptr(bias), N.ptr(W), N.ptr(log_s), N.ptr(h_), N.ptr(z1c), N.ptr(ld), mode,
               odd, B, C, H, Wd, N.stream())
    else:
        P, L, U, Lm, Um, sign_s, Wm = small
        N.call('nf_glow_head_fwd', N.ptr(x), N.ptr(log_scale), N.ptr(bias), N.ptr(P), N.ptr(L), N.ptr(U), N.ptr(Lm), N.ptr(Um), N.ptr(sign_s),
               N.ptr(log_s), N.ptr(h_), N.ptr(z1c), N.ptr(Wm), N.ptr(ld), mode, odd, B, C, H, Wd, N.stream())
    return True


def glow_head_w_usable(z, mode):
    return (z.is_cuda and z.dim() == 4 and z.dtype == torch.float32
            and bool(N.load().nf_glow_head_w_usable(z.shape[0], z.shape[1], z.shape[2], z.shape[3], int(mode))))


def glow_head_w(z, ld, log_scale, bias, W, log_s, holder, idx, mode, odd, defer=False, bwd_defer=False):
    """(h, z1c, ld): ActNorm.forward -> InvertibleConv1x1.forward (weight given) -> conditioning half of the split, fused.
    defer: no launch -- the caller guarantees that the coupling's chain launch follows and performs it (fused_conv._cn_forward).
    bwd_defer: the caller guarantees that z IS the output of a fused image coupling (fused_conv._FusedConvCoupling) of the same
    shape: inside a trainer step the backward leaves this head's data gradient to that coupling's backward launch."""
    return _GlowHeadW.apply(_contig(z), _owned_ld(ld), log_scale, bias, W, log_s, holder, idx, mode, odd, defer, bwd_defer)


HEAD_MAX_C = 4


def glow_head(z, ld, log_scale, bias, P, L, U, L_mask, U_mask, sign_s, log_s, mode, odd, bwd_defer=False, defer=False):
    """(h, z1c, ld): ActNorm.forward -> InvertibleConv1x1.forward -> conditioning half of the split, fused (C <= 4).
    defer, bwd_defer: as in glow_head_w."""
    return _GlowHead.apply(_contig(z), _owned_ld(ld), log_scale, bias, P, L, U, L_mask, U_mask, sign_s, log_s, mode, odd, bwd_defer, defer)


def from_fused_coupling(z):
    """z is the (first) output of a fused image coupling's chain launch (fused_conv._FusedConvCoupling): the gradient a head returns for it
    is consumed by that launch's backward and by nothing else"""
    fn = z.grad_fn
    return fn is not None and type(fn).__name__ == '_FusedConvCouplingBackward' and not z._backward_hooks and z.is_contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# logit
# ----------------------------------------------------------------------------------------------------------------------
class _Logit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ld, eps):
        B = x.shape[0]
        n = x.numel() // B if B else 1
        y = torch.empty_like(x)
        N.call('nf_logit_fwd', N.ptr(x), N.ptr(y), N.ptr(ld), float(eps), 0, B, n, N.stream())
        ctx.save_for_backward(x)
        ctx.eps = float(eps)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        (x, ) = ctx.saved_tensors
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_x = torch.empty_like(x)
        N.call('nf_logit_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(x), N.ptr(g_x), ctx.eps, x.shape[0], x.numel() // max(x.shape[0], 1),
               N.stream())
        return g_x, g_ld, None


def logit(x, ld, eps, inverse=False):
    x = _contig(x)
    ld = _owned_ld(ld)
    if not inverse:
        return _Logit.apply(x, ld, eps)
    with torch.no_grad():
        y = torch.empty_like(x)
        N.call('nf_logit_fwd', N.ptr(x), N.ptr(y), N.ptr(ld), float(eps), 1, x.shape[0], x.numel() // max(x.shape[0], 1), N.stream())
    return y, ld


# ----------------------------------------------------------------------------------------------------------------------
# Sigmoid / Tanh / Arctanh (flows/modules.py:125-183) and Squeeze1d / Unsqueeze1d (flows/squeeze.py:114-151)
# ----------------------------------------------------------------------------------------------------------------------
BIJ_SIGMOID, BIJ_SIGMOID_INV, BIJ_TANH, BIJ_ARCTANH = 0, 1, 2, 3


class _Bijector(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ld, kind):
        B = x.shape[0]
        n = x.numel() // B if B else 1
        y = torch.empty_like(x)
        N.call('nf_bijector_fwd', N.ptr(x), N.ptr(y), N.ptr(ld), kind, B, n, N.stream())
        ctx.save_for_backward(x)
        ctx.kind = kind
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        (x, ) = ctx.saved_tensors
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_x = torch.empty_like(x)
        N.call('nf_bijector_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(x), N.ptr(g_x), ctx.kind, x.shape[0], x.numel() // max(x.shape[0], 1),
               N.stream())
        return g_x, g_ld, None


def bijector(x, ld, kind):
    """one elementwise bijector pass, ld[b] += its per-sample log-det.  The modules' forward directions (sigmoid, tanh, arctanh) build an
    autograd graph; the Sigmoid module's inverse (kind BIJ_SIGMOID_INV) builds none, like every inverse direction here."""
    x = _contig(x)
    ld = _owned_ld(ld)
    if kind != BIJ_SIGMOID_INV and torch.is_grad_enabled() and (x.requires_grad or ld.requires_grad):
        return _Bijector.apply(x, ld, kind)
    with torch.no_grad():
        y = torch.empty_like(x)
        N.call('nf_bijector_fwd', N.ptr(x), N.ptr(y), N.ptr(ld), kind, x.shape[0], x.numel() // max(x.shape[0], 1), N.stream())
    return y, ld


class _Squeeze1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, odd, inverse):
        z = _contig(z)
        out = torch.empty_like(z)
        N.call('nf_squeeze1d', N.ptr(z), N.ptr(out), z.shape[0], z.shape[1], int(odd), int(inverse), N.stream())
        ctx.meta = (int(odd), int(inverse))
        return out

    @staticmethod
    def backward(ctx, g):
        odd, inverse = ctx.meta
        g = _contig(g)
        out = torch.empty_like(g)
        N.call('nf_squeeze1d', N.ptr(g), N.ptr(out), g.shape[0], g.shape[1], odd, 1 - inverse, N.stream())
        return out, None, None


def squeeze1d(z, odd=False, inverse=False):
    """(B, D) -> cat(z[:, odd::2], z[:, 1 - odd::2]) (flows/squeeze.py:63-83, :124-127); inverse=True: the inverse map."""
    if z.dim() != 2 or z.shape[1] % 2:
        raise ValueError('squeeze1d takes (B, D) data with even D, got %s' % (tuple(z.shape), ))
    return _Squeeze1d.apply(z, bool(odd), bool(inverse))


# ----------------------------------------------------------------------------------------------------------------------
# Flow++ mixture-of-logistics coupling
# ----------------------------------------------------------------------------------------------------------------------
class _MixLogCoupling(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, params, a, c, ld, K, eps, mode, odd):
        B, C, H, W = _bchw(z)
        y = torch.empty_like(z)
        N.call('nf_mixlog_coupling_fwd', N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c), N.ptr(y), N.ptr(ld), K, float(eps),
               mode, int(odd), B, C, H, W, N.stream())
        ctx.save_for_backward(z, params, a, c)
        ctx.meta = (K, float(eps), mode, int(odd))
        ctx.sinks = _sinks(a, c)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        z, params, a, c = ctx.saved_tensors
        K, eps, mode, odd = ctx.meta
        B, C, H, W = _bchw(z)
        g_y, g_ld = _contig(g_y), _contig(g_ld)
        g_z = torch.empty_like(z)
        g_p = torch.empty_like(params)
        if ctx.sinks is not None and z.dim() == 4:
            # image data inside a trainer step: the coupling's scale / shift gradients leave as per-workgroup partial sums and are folded
            # with the conditioners' slabs where the pass ends (fused_flowpp_img.FlowppImgDefer) -- no same-address atomics
            from .fused_flowpp_img import FPP_IMG_DEFER
            n = int(N.load().nf_mixlog_bwd_blocks(K, mode, B, C, H, W)) if FPP_IMG_DEFER.active else 0
            if n > 0:
                part = torch.empty(2 * n, dtype=torch.float32, device=z.device)
                N.call('nf_mixlog_coupling_bwd_partials', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c), N.ptr(g_z),
                       N.ptr(g_p), N.ptr(part), K, eps, mode, odd, B, C, H, W, N.stream())
                FPP_IMG_DEFER.sums.append((part[:n], ctx.sinks[0], 1, 1, n, True, 1))
                FPP_IMG_DEFER.sums.append((part[n:], ctx.sinks[1], 1, 1, n, True, 1))
                return g_z, g_p, None, None, g_ld, None, None, None, None
        if ctx.sinks is not None:
            pa, pc, ga, gc = ctx.sinks[0].data_ptr(), ctx.sinks[1].data_ptr(), None, None
        else:
            g_ac = WS.zeros(2, z.device)
            pa, pc = g_ac.data_ptr(), g_ac.data_ptr() + 4
            ga, gc = g_ac[0:1].view_as(a), g_ac[1:2].view_as(c)
        N.call('nf_mixlog_coupling_bwd', N.ptr(g_y), N.ptr(g_ld), N.ptr(z), N.ptr(params), N.ptr(a), N.ptr(c),
               N.ptr(g_z), N.ptr(g_p), pa, pc, K, eps, mode, odd, B, C, H, W, N.stream())
        return g_z, g_p, ga, gc, g_ld, None, None, None, None


def mixlog_coupling(z, params, a_log_scale, a_bias, ld, n_mixtures, mode, odd, inverse=False, logit_eps=1.0e-5):
    """MixLogAttnCoupling._transform / _inverse_transform + split + merge given ``params`` = conditioner(z1)
    (flows/coupling.py:172-210).  The inverse reproduces the reference's 25-or-100 iteration bisection rule."""
    z, params = _contig(z), _contig(params)
    half = _half_shape(z, mode)
    want = (half[0], (2 + 3 * n_mixtures) * half[1]) + tuple(half[2:])
    if tuple(params.shape) != want:
        raise ValueError('conditioner output has shape %s, expected %s' % (tuple(params.shape), want))
    ld = _owned_ld(ld)
    if not inverse:
        return _MixLogCoupling.apply(z, params, a_log_scale, a_bias, ld, n_mixtures, logit_eps, mode, odd)
    with torch.no_grad():
        B, C, H, W = _bchw(z)
        y = torch.empty_like(z)
        n_el = z.numel() // 2
        scratch = torch.empty(3 * max(n_el, 1), dtype=z.dtype, device=z.device)
        flag = torch.empty(1, dtype=torch.int32, device=z.device)
        N.call('nf_mixlog_coupling_inv', N.ptr(z), N.ptr(params), N.ptr(a_log_scale), N.ptr(a_bias), N.ptr(y), N.ptr(ld),
               N.ptr(scratch), N.ptr(flag), n_mixtures, mode, int(odd), B, C, H, W, N.stream())
    return y, ld


# ----------------------------------------------------------------------------------------------------------------------
# standalone MixLogCDF (flows/modules.py:186-212)
# ----------------------------------------------------------------------------------------------------------------------
def _mixcdf_shapes(x, log_pi):
    B = x.shape[0]
    n = x.numel() // B if B else 0
    if log_pi.dim() != x.dim() + 1 or log_pi.shape[0] != B or tuple(log_pi.shape[2:]) != tuple(x.shape[1:]):
        raise ValueError('MixLogCDF: x %s needs log_pi / mu / s of shape (B, K) + x.shape[1:], got %s'
                         % (tuple(x.shape), tuple(log_pi.shape)))
    return B, n, log_pi.shape[1]


class _MixLogCDF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, log_pi, mu, s, ld):
        B, n, K = _mixcdf_shapes(x, log_pi)
        out = torch.empty_like(x)
        N.call('nf_mixlogcdf_fwd', N.ptr(x), N.ptr(log_pi), N.ptr(mu), N.ptr(s), N.ptr(out), N.ptr(ld), K, B, n, N.stream())
        ctx.save_for_backward(x, log_pi, mu, s)
        ctx.mark_dirty(ld)
        return out, ld

    @staticmethod
    def backward(ctx, g_out, g_ld):
        x, log_pi, mu, s = ctx.saved_tensors
        B, n, K = _mixcdf_shapes(x, log_pi)
        g_out, g_ld = _contig(g_out), _contig(g_ld)
        g_x, g_lp, g_mu, g_s = torch.empty_like(x), torch.empty_like(log_pi), torch.empty_like(mu), torch.empty_like(s)
        N.call('nf_mixlogcdf_bwd', N.ptr(g_out), N.ptr(g_ld), N.ptr(x), N.ptr(log_pi), N.ptr(mu), N.ptr(s), N.ptr(g_x),
               N.ptr(g_lp), N.ptr(g_mu), N.ptr(g_s), K, B, n, N.stream())
        return g_x, g_lp, g_mu, g_s, g_ld


def mixlogcdf(x, log_pi, mu, s, ld, inverse=False):
    """MixLogCDF.forward / .backward (flows/modules.py:190-212): ``log_pi`` is already log-softmaxed over the mixture axis.
    Unlike the in-place transforms, the reference returns a NEW log-det tensor here (modules.py:194): so does this."""
    x, log_pi, mu, s = _contig(x), _contig(log_pi), _contig(mu), _contig(s)
    if mu.shape != log_pi.shape or s.shape != log_pi.shape:
        raise ValueError('MixLogCDF: log_pi, mu and s must have one shape')
    ld = ld.clone()
    if not inverse:
        return _MixLogCDF.apply(x, log_pi, mu, s, ld)
    with torch.no_grad():
        B, n, K = _mixcdf_shapes(x, log_pi)
        out = torch.empty_like(x)
        scratch = torch.empty(2 * max(x.numel(), 1), dtype=x.dtype, device=x.device)
        flag = torch.empty(1, dtype=torch.int32, device=x.device)
        N.call('nf_mixlogcdf_inv', N.ptr(x), N.ptr(log_pi), N.ptr(mu), N.ptr(s), N.ptr(out), N.ptr(ld), N.ptr(scratch),
               N.ptr(flag), K, B, n, N.stream())
    return out, ld


# ----------------------------------------------------------------------------------------------------------------------
# NLL under the standard-normal prior (training harness, main.py:49-51, :85)
# ----------------------------------------------------------------------------------------------------------------------
class _NLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, ld):
        B = z.shape[0]
        D = z.numel() // B
        loss = WS.zeros_owned((), z.device) if z.dtype == torch.float32 else torch.zeros((), dtype=z.dtype, device=z.device)
        N.call('nf_nll_loss', N.ptr(z), N.ptr(ld), N.ptr(loss), B, D, N.stream())
        ctx.save_for_backward(z)
        return loss

    @staticmethod
    def backward(ctx, g):
        (z, ) = ctx.saved_tensors
        B = z.shape[0]
        D = z.numel() // B
        g = _contig(g)
        g_z = torch.empty_like(z)
        g_ld = torch.empty(B, dtype=z.dtype, device=z.device)
        N.call('nf_nll_loss_bwd', N.ptr(z), N.ptr(g), N.ptr(g_z), N.ptr(g_ld), B, D, N.stream())
        return g_z, g_ld


def nll_loss(z, ld):
    """-mean_b( log N(z_b; 0, I) + ld_b ): one reduction kernel (the reference builds a D x D MultivariateNormal)."""
    return _NLL.apply(_contig(z), _contig(ld))
