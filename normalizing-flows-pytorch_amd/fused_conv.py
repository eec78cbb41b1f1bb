"""
The image conditioner ConvNet (flows/modules.py:416-438) on the fused convolution + BatchNorm2d kernels of
csrc/conv_bn.hip (C ABI ``nf_conv_bn_fwd / nf_conv_bn_bwd / nf_slab_sum``): 6 launches forward, 6 + 1 backward, instead
of ~25 / ~60 MIOpen + ATen kernels (convolution, bias add, BatchNorm, ReLU, residual add; their backward twins, the
layout transposes MIOpen's weight-gradient kernels need and a reduction per bias).

Exact restatement of the module math in training mode (batch statistics, gradients through the statistics, running
statistics bookkeeping) and in evaluation mode.  The weight-norm arithmetic stays where it is: every convolution of a
model gets its effective weight from ONE ``nf_weight_norm_fwd`` launch per pass (fused.weight_norm_all), the kernels
here consume effective weights and return the gradient with respect to them.
"""
import ctypes
import functools

import torch

from . import _native as N
from . import workspace as WS
from .fused import BN_EPS, BN_MOMENTUM, H, R, _desc

WS_ROWS = 2 * R + 2      # per BatchNorm: sum[R], sqsum[R], save_mean, save_invstd (rows of 32)
GB = 256                 # replica stride of the bias-gradient accumulators (nfhip.h: nf_conv_bwd_desc.g_bias)

_FWD_FIELDS = ['in_', 'weight', 'bias', 'residual', 'out', 'bn_gamma', 'bn_beta', 'bn_sum', 'bn_sqsum', 'bn_center',
               'bn_running_mean', 'bn_running_var', 'bn_num_batches', 'bn_save_mean', 'bn_save_invstd', 'stat_sum',
               'stat_sqsum', 'wpk']
_BWD_FIELDS = ['in_', 'weight', 'bn_gamma', 'bn_beta', 'bn_save_mean', 'bn_save_invstd', 'g_direct', 'g_skip', 'gn_src',
               'out', 'cbn_gamma', 'cbn_save_mean', 'cbn_save_invstd', 'cbn_sum_g', 'cbn_sum_gx', 'g_store', 'g_bias',
               'g_weff', 'gn_out', 'sum_g', 'sum_gx', 'wpk']


class ConvDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in _FWD_FIELDS] + [('valid_h', ctypes.c_int), ('valid_w', ctypes.c_int)]


class ConvBwdDesc(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in _BWD_FIELDS] + [('valid_h', ctypes.c_int), ('valid_w', ctypes.c_int)]


class SlabSumDesc(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('n', ctypes.c_int64), ('stride', ctypes.c_int64),
                ('n_slabs', ctypes.c_int), ('accumulate', ctypes.c_int), ('taps', ctypes.c_int), ('reserved', ctypes.c_int)]


class ConvNetDesc(ctypes.Structure):
    """nf_convnet_desc of include/nfhip.h (csrc/conv_chain.hip: the whole conditioner in one persistent launch)"""
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p * 6), ('b', ctypes.c_void_p * 6), ('gamma', ctypes.c_void_p * 5),
                ('beta', ctypes.c_void_p * 5), ('rmean', ctypes.c_void_p * 5), ('rvar', ctypes.c_void_p * 5),
                ('nbt', ctypes.c_void_p * 5), ('acts', ctypes.c_void_p * 5), ('out', ctypes.c_void_p),
                ('save_mean', ctypes.c_void_p * 5), ('save_invstd', ctypes.c_void_p * 5), ('ws_zero', ctypes.c_void_p),
                ('cp_z', ctypes.c_void_p), ('cp_y', ctypes.c_void_p), ('cp_ld', ctypes.c_void_p), ('cp_a', ctypes.c_void_p),
                ('cp_c', ctypes.c_void_p), ('cp_mode', ctypes.c_int), ('cp_odd', ctypes.c_int), ('cp_C', ctypes.c_int),
                ('cp_inverse', ctypes.c_int), ('wpk', ctypes.c_void_p * 6),
                ('hd_x', ctypes.c_void_p), ('hd_ls', ctypes.c_void_p), ('hd_bias', ctypes.c_void_p), ('hd_W', ctypes.c_void_p),
                ('hd_log_s', ctypes.c_void_p), ('hd_x1', ctypes.c_void_p),
                ('hs_P', ctypes.c_void_p), ('hs_L', ctypes.c_void_p), ('hs_U', ctypes.c_void_p), ('hs_Lm', ctypes.c_void_p),
                ('hs_Um', ctypes.c_void_p), ('hs_sign', ctypes.c_void_p), ('hs_Wout', ctypes.c_void_p),
                ('ws_gen', ctypes.c_int), ('ws_reserved', ctypes.c_int)]


class ConvNetBwdDesc(ctypes.Structure):
    """nf_convnet_bwd_desc of include/nfhip.h (the data gradient of the whole conditioner in one persistent launch)"""
    _fields_ = [('w', ctypes.c_void_p * 6), ('gamma', ctypes.c_void_p * 5), ('beta', ctypes.c_void_p * 5),
                ('save_mean', ctypes.c_void_p * 5), ('save_invstd', ctypes.c_void_p * 5), ('acts', ctypes.c_void_p * 5),
                ('g_out', ctypes.c_void_p), ('gn', ctypes.c_void_p * 5), ('sum_g', ctypes.c_void_p * 5),
                ('sum_gx', ctypes.c_void_p * 5), ('g_store', ctypes.c_void_p * 2), ('g_x', ctypes.c_void_p),
                ('ws_zero', ctypes.c_void_p),
                ('cp_g_y', ctypes.c_void_p), ('cp_g_ld', ctypes.c_void_p), ('cp_z', ctypes.c_void_p), ('cp_out', ctypes.c_void_p),
                ('cp_a', ctypes.c_void_p), ('cp_c', ctypes.c_void_p), ('cp_g_z', ctypes.c_void_p), ('cp_g_out', ctypes.c_void_p),
                ('cp_g_a', ctypes.c_void_p), ('cp_g_c', ctypes.c_void_p), ('cp_mode', ctypes.c_int), ('cp_odd', ctypes.c_int),
                ('cp_C', ctypes.c_int), ('ws_gen', ctypes.c_int), ('g_gamma', ctypes.c_void_p * 5), ('g_beta', ctypes.c_void_p * 5),
                ('wpk', ctypes.c_void_p * 6), ('hd_g_h', ctypes.c_void_p), ('hd_W', ctypes.c_void_p), ('hd_ls', ctypes.c_void_p)]


class ConvPackDesc(ctypes.Structure):
    """nf_conv_pack_desc of include/nfhip.h"""
    _fields_ = [('w', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('O', ctypes.c_int), ('I', ctypes.c_int), ('ksize', ctypes.c_int),
                ('reserved', ctypes.c_int)]


CONV_PACK_ON = True


def pack_conv_weights(wn_modules, w_effs):
    """The chain kernels read their weights as LDS images of the three-way bf16 split (csrc/conv_chain.hip): written here once per
    pass for every weight-normed convolution whose shape the kernels take (nf_conv_weight_pack, 64 layers per launch), into one
    persistent buffer per module (static addresses: hipGraph-safe).  A conditioner uses its images only while they belong to the
    effective weights of the pass under way (``_w_pack_of is _w_eff``)."""
    if not CONV_PACK_ON:
        return
    lib = N.load()
    img = N.header_constant('NF_CONV_PACK_IMAGE_FLOATS')
    step = N.header_constant('NF_CONV_PACK_MAX_LAYERS')
    descs = []
    for m, w in zip(wn_modules, w_effs):
        if not (w.is_cuda and w.dim() == 4 and w.dtype == torch.float32 and w.is_contiguous()):
            continue
        n = getattr(m, '_w_pack_n', None)
        if n is None:
            n = m._w_pack_n = int(lib.nf_conv_weight_pack_images(w.shape[0], w.shape[1], w.shape[2]))
        if n == 0:
            continue
        buf = getattr(m, '_w_pack', None)
        if buf is None or buf.device != w.device:
            buf = m._w_pack = torch.empty(n * img, dtype=torch.float32, device=w.device)
        m._w_pack_of = w
        m._w_pack_gen = getattr(m, '_w_pack_gen', 0) + 1        # a backward that saved an earlier pass's images must not read these
        descs.append(ConvPackDesc(w.data_ptr(), buf.data_ptr(), w.shape[0], w.shape[1], w.shape[2], 0))
    for k0 in range(0, len(descs), step):
        chunk = descs[k0:k0 + step]
        arr = (ConvPackDesc * len(chunk))(*chunk)
        N.call('nf_conv_weight_pack', ctypes.addressof(arr), len(chunk), N.stream())


def _convnet_packs(net):
    """the six weight-image buffers of a ConvNet if all of them belong to the pass under way, else None"""
    convs, _ = _convnet_modules(net)
    packs = []
    for c in convs:
        w = getattr(c, '_w_eff', None)
        if w is None or getattr(c, '_w_pack_of', None) is not w:
            return None
        packs.append(_Pack(c._w_pack, c, c._w_pack_gen))
    return tuple(packs)


class _Pack:
    """a module's weight-image buffer as of one pass: the buffer is persistent (static address under hipGraph replay) and is REWRITTEN by
    the next pass's pack_conv_weights, so a backward that runs after a later forward (forward, step, forward, backward of the first)
    must not read it -- ``current()`` says whether the images still belong to the weights this pass saved"""
    __slots__ = ('buf', 'module', 'gen')

    def __init__(self, buf, module, gen):
        self.buf, self.module, self.gen = buf, module, gen

    def data_ptr(self):
        return self.buf.data_ptr()

    def current(self):
        return getattr(self.module, '_w_pack_gen', None) == self.gen and self.module._w_pack is self.buf


CONV_CHAIN_ON = __import__('os').environ.get('NF_CONV_CHAIN', '1') != '0'
CONV_CHAIN_BWD_ON = True
CONV_COUPLING_ON = True
CONV_COUPLING_MIN_PX = 0


@functools.lru_cache(maxsize=None)
def _chain_ws_floats(B, I0, O_out, Hh, Ww):
    return int(N.load().nf_convnet_chain_ws_floats(B, I0, O_out, Hh, Ww))


def _chain_usable(B, I0, O_out, Hh, Ww):
    return CONV_CHAIN_ON and bool(N.load().nf_convnet_chain_usable(B, I0, O_out, Hh, Ww))


def _fwd(shape, I, O, k, training, **kw):
    B, Hh, Ww = shape
    d = _desc(ConvDesc, **kw)
    N.call('nf_conv_bn_fwd', ctypes.addressof(d), B, I, O, Hh, Ww, k, int(training), BN_EPS, BN_MOMENTUM, N.stream())


def _bwd(shape, I, O, k, **kw):
    B, Hh, Ww = shape
    d = _desc(ConvBwdDesc, **kw)
    N.call('nf_conv_bn_bwd', ctypes.addressof(d), B, I, O, Hh, Ww, k, N.stream())


@functools.lru_cache(maxsize=None)
def _wgrad_slabs(B, Hh, Ww, n_layers):
    return int(N.load().nf_conv_wgrad_slabs(B, Hh, Ww, n_layers))


def _slab_sum(jobs):
    arr = (SlabSumDesc * len(jobs))()
    for i, (src, dst, n, stride, n_slabs, acc, taps) in enumerate(jobs):
        arr[i].src, arr[i].dst, arr[i].n, arr[i].stride = src.data_ptr(), dst.data_ptr(), n, stride
        arr[i].n_slabs, arr[i].accumulate, arr[i].taps = n_slabs, int(acc), taps
    N.call('nf_slab_sum', ctypes.addressof(arr), len(jobs), N.stream())


def _slab_sum_all(jobs):
    step = N.header_constant('NF_SLAB_SUM_MAX')
    for k0 in range(0, len(jobs), step):
        _slab_sum(jobs[k0:k0 + step])


class ConvDefer:
    """Deferred weight gradients of the fused image conditioners.  The weight-gradient tiles, slabs and bias sums of a
    convolution's backward are on the latency chain of its launch (~10 of ~36 us on 8 .. 128 workgroups), yet only the
    weight-norm backward at the very end of the pass and Adam wait for them.  Inside a trainer step whose effective weights
    come from fused.weight_norm_all (which flushes before it reads the gradients), a conditioner's backward launches only
    the data-gradient passes (nf_conv_bn_bwd with g_weff = NULL) and queues the rest; ``flush`` runs the queued layers
    sixteen per launch (nf_conv_bn_wgrad_multi), where they fill the machine, then their slab sums.  Closed (the default)
    nothing is deferred."""

    def __init__(self):
        self.active = False
        self.armed = False
        self.layers = []        # (key, desc kwargs, dst weight gradient, n_slabs)
        self.sums = []          # slab-sum jobs that read accumulators the deferred passes fill, or that can wait as well
        self.scratch = {}
        self.plu_jobs = []      # PLU weight backward of the C <= 4 heads (functional._GlowHead): batched at the flush
        self.head_jobs = []     # ... and their parameter sums, which those PLU jobs read (functional.launch_small_head_params)

    def begin(self):
        self.layers, self.sums = [], []
        self.active, self.armed = CONV_DEFER_ON, False
        self.plu_jobs = []
        self.head_jobs = []

    def arm(self, weights):
        """fused.weight_norm_all: these effective weights' gradients are consumed by a backward that flushes first"""
        if self.active:
            self.armed = True
            for w in weights:
                w._nf_deferred_grad_ok = True

    def usable(self, weights):
        return self.active and self.armed and all(getattr(w, '_nf_deferred_grad_ok', False) for w in weights)

    SCRATCH_RING = 3            # scratch buffers in rotation: the slab sums of up to SCRATCH_RING - 1 launches share one nf_slab_sum

    def _slab_scratch(self, n, device, slot=0):
        t = self.scratch.get((device, slot))
        if t is None or t.numel() < n:
            t = self.scratch[(device, slot)] = torch.empty(n, dtype=torch.float32, device=device)
        return t

    TABLE_MIN = 17              # more layers of one shape than the by-value launch takes: ONE launch over a device table of descriptors
    TABLE_MAX = 512             # (x NF_STAT_REPL chains of the ordered mode = its 4 096 turn words)
    TABLE_TILES = 64            # ... and at most this many 128-pixel tiles per workgroup
    TABLE_WGS = 256             # workgroups the table launch aims at: ONE per layer once there are 256 layers (C4, ms per step at 256 / 512 / 1024 /
                                # 2048 / 4096: 22.41 / 22.42 / 22.53 / 22.89 / 23.47 -- every extra slab is a 36 KB write, a fold and fewer tiles per prologue)

    def launch_layers(self, layers):
        """the weight-gradient passes of ``layers`` (entries as in self.layers) and their slab sums.  Up to sixteen layers of one shape
        travel in the kernel arguments of one launch (nf_conv_bn_wgrad_multi); MORE of one shape -- config 4 queues 320 hidden layers per
        resolution -- run as ONE launch over a descriptor table in device memory (nf_conv_bn_wgrad_table, round 6): sixteen per launch
        were 2 .. 8 tiles per workgroup, i.e. prologue + slab write + launch gap."""
        step = N.header_constant('NF_CONV_WGRAD_MAX')
        sum_max = N.header_constant('NF_SLAB_SUM_MAX')
        groups = {}
        for e in layers:
            groups.setdefault(e[0], []).append(e)
        pending, launches = [], 0        # slab-sum jobs of the launches whose scratch buffers are still untouched
        for key, es in groups.items():
            (B, Hh, Ww), I, O, k = key[:4]           # (key[4]: the valid extent of maps in power-of-two storage, one per launch)
            table = WGRAD_TABLE and len(es) >= self.TABLE_MIN
            per_launch = self.TABLE_MAX if table else step
            for k0 in range(0, len(es), per_launch):
                chunk = es[k0:k0 + per_launch]
                if table:
                    tiles = (B * Hh * Ww + 127) // 128
                    # (... and no workgroup with more than 64 tiles: at config 4's literal batch a 16 x 16 layer is 1 024 tiles, which one
                    #  workgroup per layer would walk in two uneven rounds over the 256 compute units)
                    slabs = max(1, min(tiles, 128, max(-(-self.TABLE_WGS // len(chunk)), -(-tiles // self.TABLE_TILES))))
                else:
                    slabs = _wgrad_slabs(B, Hh, Ww, len(chunk))       # (per launch: one workgroup per compute unit over all its layers)
                per = [slabs * e[2].numel() for e in chunk]
                dev = chunk[0][2].device
                if pending and (launches % self.SCRATCH_RING == 0 or len(pending) + 2 * len(chunk) > sum_max):
                    _slab_sum_all(pending)           # before this launch takes the oldest scratch buffer (stream order)
                    pending, launches = [], 0
                scratch = self._slab_scratch(sum(per), dev, launches % self.SCRATCH_RING)
                launches += 1
                arr = (ConvBwdDesc * len(chunk))()
                jobs, off = [], 0
                for i, e in enumerate(chunk):
                    _, kw, g_w = e[:3]
                    region = scratch[off:off + per[i]]
                    off += per[i]
                    d = _desc(ConvBwdDesc, g_weff=region, **kw)
                    ctypes.memmove(ctypes.addressof(arr) + i * ctypes.sizeof(ConvBwdDesc), ctypes.addressof(d), ctypes.sizeof(ConvBwdDesc))
                    jobs.append((region, g_w, g_w.numel(), g_w.numel(), slabs, False, k * k))
                    if len(e) > 4 and e[4] is not None:      # the layer's bias sums (filled by this very launch) ride the same slab sum
                        jobs.append(e[4])
                if table:
                    tab = torch.empty(len(chunk) * ctypes.sizeof(ConvBwdDesc), dtype=torch.uint8, device=dev)
                    N.call('nf_conv_bn_wgrad_table', ctypes.addressof(arr), tab.data_ptr(), len(chunk), slabs, B, I, O, Hh, Ww, k, N.stream())
                else:
                    N.call('nf_conv_bn_wgrad_multi', ctypes.addressof(arr), len(chunk), B, I, O, Hh, Ww, k, N.stream())
                pending += jobs
        if pending:
            _slab_sum_all(pending)

    def flush(self):
        layers, sums = self.layers, self.sums
        self.layers, self.sums, self.active, self.armed = [], [], False, False
        plu, self.plu_jobs = self.plu_jobs, []
        heads, self.head_jobs = self.head_jobs, []
        from .functional import flush_all_pending_head_bwd
        flush_all_pending_head_bwd()            # (a head gradient that no chain launch picked up)
        if heads:
            from .functional import launch_small_head_params
            launch_small_head_params(heads)
        if plu:
            from .fused import PluDesc, _plu_launch
            from .fused import _desc as _fdesc
            _plu_launch('nf_invconv_weight_bwd_multi',
                        [_fdesc(PluDesc, g_W=gW, P=P, L=L, U=U, L_mask=Lm, U_mask=Um, sign_s=sg, log_s=ls, g_ld=sgld, g_L=gL, g_U=gU,
                                g_log_s=gls, B=1, C=C, accumulate=1, pixels=float(px))
                         for gW, P, L, U, Lm, Um, sg, ls, sgld, gL, gU, gls, C, px in plu])
        if not layers and not sums:
            return
        self.launch_layers(layers)
        _slab_sum_all(sums)


# internal constants (tests flip them to compare the paths; they are not environment switches any more -- round 5 pruned the switchboard:
# every setting measured slower than the default in rounds 2 - 4 lost its switch, the side-stream overlap of the weight-gradient launches
# lost its code as well: profiles/r04_overlap_ab.txt)
WGRAD_FROM_STORE = True
WGRAD_TABLE = True         # False: at most NF_CONV_WGRAD_MAX layers per weight-gradient launch (the round-2 .. 5 form; tests compare the two)
CONV_DEFER_ON = True
CONV_DEFER = ConvDefer()


CHAIN_SLOTS_SHARED = True      # (internal: False = a fresh zeroed slot buffer per chain launch, the round 2 .. 5 form; tests compare)


def _chain_slots(n, dev):
    """(exchange slots of one chain launch, its generation number).  Inside a trainer step every launch shares ONE buffer from the
    step's zero arena -- zeroed once where the step begins -- and comes with its own generation (nf_convnet_desc.ws_gen): the launches
    are serialised on the stream and a launch takes only its own tags for an arrival.  A fresh buffer per launch (5.6 MB each at the
    16 x 16 levels) was 795 MB of memset per config-4 step.  Outside a step: fresh zeros, generation 0."""
    A = WS.ARENA
    if not (CHAIN_SLOTS_SHARED and A.active and A.buf is not None and A.buf.device == dev):
        return WS.zeros(n, dev), 0
    t = A.step_state.get('chain_slots')
    if t is None or t.numel() < n:
        t = A.step_state['chain_slots'] = WS.zeros(n, dev)
    A.step_state['chain_gen'] = gen = A.step_state.get('chain_gen', 0) + 1
    if gen >= (1 << 28):                                      # (8 gen + 7 stays a 32-bit tag)
        return WS.zeros(n, dev), 0
    return t, gen


def _convnet_modules(net):
    convs = [net.in_block[0]]
    bns = []
    for blk in net.mid_block:
        bns += [blk.net[0], blk.net[3]]
        convs += [blk.net[2], blk.net[5]]
    bns.append(net.out_block[0])
    convs.append(net.out_block[2])
    return convs, bns


def _pow2_ceil(n):
    return 1 << max(int(n) - 1, 0).bit_length()


@functools.lru_cache(maxsize=None)
def _storage_dims(B, I, O_out, Hv, Wv):
    """(Hs, Ws, masked): the extent the per-layer kernels run a (Hv, Wv) map at.  The kernels tile maps whose sides are powers
    of two (the CIFAR pyramid: 16, 8, 4); any other map (the 28 x 28 pyramid: 14, 7) is held in the next power-of-two storage
    with its valid extent in the descriptors (nf_conv_desc.valid_h / valid_w): the dead border reads as zero padding and stays
    out of every sum.  None when neither fits."""
    lib = N.load()
    ok = lambda h, w: bool(lib.nf_conv_bn_usable(B, I, H, h, w, 3) and lib.nf_conv_bn_usable(B, H, O_out, h, w, 1))
    if ok(Hv, Wv):
        return Hv, Wv, False
    Hs, Ws = _pow2_ceil(Hv), _pow2_ceil(Wv)
    return (Hs, Ws, True) if ok(Hs, Ws) else None


def convnet_usable(net, x):
    """ConvNet with two residual blocks of 32 filters on an input whose spatial size tiles into the kernels' 128-pixel
    groups (every level of the reference's CIFAR pyramid does) or fits such a map with a dead border (the 28 x 28 pyramid)."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return False
    return _convnet_usable_shape(net, tuple(x.shape), padded=True)


def _convnet_usable_shape(net, shape, padded=False):
    if not (shape[0] > 0 and len(net.mid_block) == 2):
        return False
    convs, _ = _convnet_modules(net)
    c0 = getattr(convs[0], 'module', convs[0])
    c5 = getattr(convs[-1], 'module', convs[-1])
    if c0.out_channels != H or c5.in_channels != H or c0.kernel_size != (3, 3) or c5.kernel_size != (1, 1):
        return False
    B, I, Hh, Ww = shape
    dims = _storage_dims(B, I, c5.out_channels, Hh, Ww)
    return dims is not None and (padded or not dims[2])


def _convnet_tensors(net):
    convs, bns = _convnet_modules(net)
    tensors = []
    for c in convs:
        if hasattr(c, 'effective_weight'):            # conditioners.WeightNorm
            tensors += [c.effective_weight(), c.module.bias]
        else:
            tensors += [c.weight, c.bias]
    for bn in bns:
        tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
    return tensors


def _cn_forward(ctx, x, training, defer, tensors, cpl=None, packs=None):
    """forward of the conditioner; ``cpl`` = (z, ld, a, c, mode, odd, inverse): the affine coupling it parameterises rides the
    epilogue of the chain launch (returns y; ld is updated in place), else returns the conditioner's output."""
    nl, nb = 6, 5
    conv = [tensors[2 * i:2 * i + 2] for i in range(nl)]
    bns = [tensors[2 * nl + 5 * i:2 * nl + 5 * i + 5] for i in range(nb)]
    x = x.contiguous()
    B, I0, Hv, Wv = x.shape
    O_out = conv[-1][0].shape[0]
    Hh, Ww, masked = _storage_dims(B, I0, O_out, Hv, Wv)
    vk = dict(valid_h=Hv, valid_w=Wv) if masked else {}
    if masked:                                # power-of-two storage; the kernels never read the dead border of the activations
        xs = x.new_zeros(B, I0, Hh, Ww)
        xs[:, :, :Hv, :Wv] = x
        x = xs
    shape = (B, Hh, Ww)
    dev = x.device
    ws = WS.zeros(nb * WS_ROWS * H, dev).view(nb, WS_ROWS, H)
    acts = [torch.empty(B, H, Hh, Ww, dtype=torch.float32, device=dev) for _ in range(nb)]
    out = torch.empty(B, O_out, Hh, Ww, dtype=torch.float32, device=dev)
    w = [c[0].contiguous() for c in conv]
    y = None

    def bn_kw(j):
        g, b, rm, rv, nbt = bns[j]
        return dict(bn_gamma=g, bn_beta=b, bn_sum=ws[j, 0], bn_sqsum=ws[j, R], bn_center=conv[j][1], bn_running_mean=rm,
                    bn_running_var=rv, bn_num_batches=nbt, bn_save_mean=ws[j, 2 * R], bn_save_invstd=ws[j, 2 * R + 1])

    def stats(j):                             # evaluation mode normalises with running statistics: no batch sums
        return dict(stat_sum=ws[j, 0], stat_sqsum=ws[j, R]) if training else {}

    if not masked and _chain_usable(B, I0, O_out, Hh, Ww):   # the whole conditioner: ONE persistent launch (csrc/conv_chain.hip)
        d = ConvNetDesc()
        d.x = x.data_ptr()
        for i in range(nl):
            d.w[i], d.b[i] = w[i].data_ptr(), conv[i][1].data_ptr()
        for j in range(nb):
            g_, b_, rm, rv, nbt = bns[j]
            d.gamma[j], d.beta[j], d.rmean[j], d.rvar[j] = g_.data_ptr(), b_.data_ptr(), rm.data_ptr(), rv.data_ptr()
            d.nbt[j] = nbt.data_ptr() if nbt is not None else None
            d.acts[j] = acts[j].data_ptr()
            d.save_mean[j], d.save_invstd[j] = ws[j, 2 * R].data_ptr(), ws[j, 2 * R + 1].data_ptr()
        d.out = out.data_ptr()
        if packs is not None:
            for i in range(nl):
                d.wpk[i] = packs[i].data_ptr()
        # (the slots' tensor must outlive every allocation up to the launch: a freed block is handed to the next torch.empty)
        nws = _chain_ws_floats(B, I0, O_out, Hh, Ww)
        need = training or nws > N.header_constant('NF_CONVNET_WS_FLOATS')           # (halo hand-over: in evaluation mode too)
        slots, d.ws_gen = _chain_slots(nws, dev) if need else (None, 0)
        d.ws_zero = slots.data_ptr() if need else None
        if cpl is not None:
            z, ld, a, c, mode, odd, inverse = cpl
            y = torch.empty_like(z)
            d.cp_z, d.cp_y, d.cp_ld, d.cp_a, d.cp_c = z.data_ptr(), y.data_ptr(), ld.data_ptr(), a.data_ptr(), c.data_ptr()
            d.cp_mode, d.cp_odd, d.cp_C, d.cp_inverse = int(mode), int(odd), z.shape[1], int(inverse)
            from . import functional as NF
            pend = NF.PENDING_HEADS.pop(z.data_ptr(), None)
            if pend is not None:
                # the step's ActNorm + 1 x 1 convolution were left to this launch (functional._GlowHeadW.forward(defer=True)): z is the
                # head's OUTPUT buffer, x its conditioning half -- both written by the prologue
                hx, hls, hb, hW, hlog_s, h_, z1c_, ld_, hmode, hodd, hsmall = pend
                if (h_.data_ptr() == z.data_ptr() and z1c_.data_ptr() == x.data_ptr() and ld_.data_ptr() == ld.data_ptr() and not inverse
                        and (hmode, hodd) == (int(mode), int(odd))):
                    d.hd_x, d.hd_ls, d.hd_bias, d.hd_log_s = hx.data_ptr(), hls.data_ptr(), hb.data_ptr(), hlog_s.data_ptr()
                    d.hd_x1 = x.data_ptr()
                    if hsmall is None:
                        d.hd_W = hW.data_ptr()
                    else:                      # 2 .. 4 channels: the weight is assembled in the launch from its PLU factors
                        (d.hs_P, d.hs_L, d.hs_U, d.hs_Lm, d.hs_Um, d.hs_sign, d.hs_Wout) = [t.data_ptr() for t in hsmall]
                else:                          # (not the tensors this launch works on: the head runs on its own kernel first)
                    NF.PENDING_HEADS[z.data_ptr()] = pend
                    NF.flush_pending_head(z)
        N.call('nf_convnet_chain_fwd', ctypes.addressof(d), B, I0, O_out, Hh, Ww, int(training), BN_EPS, BN_MOMENTUM, N.stream())
    else:
        if cpl is not None:
            raise RuntimeError('the fused coupling needs the chain kernel (coupling_fusable was not consulted)')
        # (large batches: the 3x3 layers run on csrc/conv_bulk.hip, which reads the pass's weight images when they exist)
        pk = (lambda i: packs[i].buf) if packs is not None else (lambda i: None)
        _fwd(shape, I0, H, 3, training, in_=x, weight=w[0], bias=conv[0][1], out=acts[0], wpk=pk(0), **stats(0), **vk)
        for j in range(1, nb):                # convolution j consumes acts[j-1] through BatchNorm j-1
            res = acts[j - 2] if j % 2 == 0 else None
            _fwd(shape, H, H, 3, training, in_=acts[j - 1], weight=w[j], bias=conv[j][1], residual=res, out=acts[j], wpk=pk(j),
                 **stats(j), **bn_kw(j - 1), **vk)
        _fwd(shape, H, O_out, 1, training, in_=acts[nb - 1], weight=w[nl - 1], bias=conv[nl - 1][1], out=out,
             **bn_kw(nb - 1), **vk)
    from .functional import _sinks
    extra = ()
    if cpl is not None:
        extra = (cpl[0], out, cpl[2], cpl[3])
        ctx.cpl_meta = (int(cpl[4]), int(cpl[5]))
        ctx.cpl_sinks = _sinks(cpl[2], cpl[3])
    ctx.save_for_backward(x, ws, *acts, *w, *[t for b in bns for t in b[:2]], *extra)
    ctx.meta = (shape, I0, O_out, bool(training))
    ctx.valid = (Hv, Wv) if masked else None
    ctx.packs = packs
    ctx.sinks = _sinks(*[c[1] for c in conv], *[t for b in bns for t in b[:2]])
    ctx.defer = bool(defer) and ctx.sinks is not None
    if masked:
        out = out[:, :, :Hv, :Wv].contiguous()
    return out if cpl is None else y


class _FusedConvNet(torch.autograd.Function):
    """x -> conv0 -> [BN,ReLU,conv, BN,ReLU,conv, +skip] * 2 -> BN,ReLU,conv1x1.

    tensors: per convolution (effective weight, bias) * 6, then per BatchNorm (gamma, beta, running_mean, running_var,
    num_batches_tracked) * 5."""

    @staticmethod
    def forward(ctx, x, training, defer, packs, *tensors):
        return _cn_forward(ctx, x, training, defer, tensors, packs=packs)

    @staticmethod
    def backward(ctx, g_out):
        lead, grads = _cn_backward(ctx, g_out)
        return (lead[0], None, None, None) + grads


class _FusedConvCoupling(torch.autograd.Function):
    """the affine coupling of an image model WITH its conditioner: y, ld = coupling(z, ConvNet(x)) where x is the untouched half of
    z (gathered by the caller: the fused Glow head produces it anyway).  One launch per direction: the transform, the merge and the
    log-det sums ride the epilogue of the conditioner's output convolution; backward, the coupling's gradient is formed on the way
    into the first transposed convolution and the conditioner's input gradient lands directly in the gradient of z -- so the
    gradient of x is reported as None (it is contained in that of z)."""

    @staticmethod
    def forward(ctx, z, x, ld, a, c, mode, odd, training, defer, packs, *tensors):
        y = _cn_forward(ctx, x, training, defer, tensors, cpl=(z, ld, a, c, mode, odd, 0), packs=packs)
        ctx.mark_dirty(ld)
        return y, ld

    @staticmethod
    def backward(ctx, g_y, g_ld):
        (g_z, g_a, g_c), grads = _cn_backward(ctx, None, cpl_grads=(g_y.contiguous(), g_ld.contiguous()))
        return (g_z, None, g_ld, g_a, g_c, None, None, None, None, None) + grads


def _cn_backward(ctx, g_out, cpl_grads=None):
    """backward of _cn_forward: returns (leading gradients, parameter gradients).  Leading = (g_x,) for the bare conditioner,
    (g_z, g_a, g_c) with the fused coupling (``cpl_grads`` = (g_y, g_ld); g_x is folded into g_z)."""
    shape, I0, O_out, training = ctx.meta
    nl, nb = 6, 5
    saved = ctx.saved_tensors
    x, ws = saved[0], saved[1]
    acts = saved[2:2 + nb]
    w = saved[2 + nb:2 + nb + nl]
    gb = saved[2 + nb + nl:2 + nb + nl + 2 * nb]
    cpl = saved[2 + nb + nl + 2 * nb:] if cpl_grads is not None else None      # (z, out, a, c)
    gamma = [gb[2 * i] for i in range(nb)]
    beta = [gb[2 * i + 1] for i in range(nb)]
    dev = x.device
    B, Hh, Ww = shape
    valid = getattr(ctx, 'valid', None)
    vk = dict(valid_h=valid[0], valid_w=valid[1]) if valid else {}
    if valid:                                  # (the dead border of the gradient is never read)
        gs = g_out.new_zeros(B, O_out, Hh, Ww)      # (zero border: a kernel that forgets the mask reads zeros, not garbage)
        gs[:, :, :valid[0], :valid[1]] = g_out
        g_out = gs
    g_out = g_out.contiguous() if cpl is None else torch.empty_like(cpl[1])    # fused coupling: written by the chain launch
    slabs = int(N.load().nf_conv_bwd_slabs(B, Hh, Ww))
    chained = CONV_CHAIN_BWD_ON and not valid and _chain_usable(B, I0, O_out, Hh, Ww)
    g_weff = [None] * nl if ((ctx.defer and CONV_DEFER.active) or chained) else \
        [torch.empty(slabs, t.numel(), dtype=torch.float32, device=dev) for t in w]
    acc = WS.zeros(nl * R * GB + nb * 2 * R * H, dev)
    g_bias = [acc[i * R * GB:(i + 1) * R * GB] for i in range(nl)]
    sums = acc[nl * R * GB:].view(nb, 2, R * H)
    gn = [torch.empty_like(acts[0]) for _ in range(nb)]
    g_x = torch.empty_like(x) if (cpl is None and ctx.needs_input_grad[0]) else None
    G_skip = None

    def in_bn(j):
        return dict(bn_gamma=gamma[j], bn_beta=beta[j], bn_save_mean=ws[j, 2 * R], bn_save_invstd=ws[j, 2 * R + 1])

    def cons_bn(j):                            # evaluation mode: statistics are constants -> no mean terms
        return dict(cbn_gamma=gamma[j], cbn_save_mean=ws[j, 2 * R], cbn_save_invstd=ws[j, 2 * R + 1],
                    cbn_sum_g=sums[j, 0] if training else None, cbn_sum_gx=sums[j, 1] if training else None)

    defer = ctx.defer and CONV_DEFER.active
    # the data gradient of the whole conditioner in ONE persistent launch (csrc/conv_chain.hip); the weight passes follow as
    # deferred (or, outside a trainer step, immediate) nf_conv_bn_wgrad_multi launches over the same descriptors
    if cpl is not None and not chained:
        raise RuntimeError('the fused coupling needs the chain kernels in both directions')
    queued = []
    stores = [torch.empty_like(acts[0]) for _ in range(2)] if chained else None

    def layer(I, O, k, i, **kw):
        """convolution i's backward: both passes now, or the data pass now (unless the chain launch covers it) and the weight
        pass queued"""
        kw.update(vk)
        if not defer and not chained:
            _bwd(shape, I, O, k, g_bias=g_bias[i], g_weff=g_weff[i], **kw)
            return
        if not chained:
            _bwd(shape, I, O, k, **kw)
        wkw = {f: v for f, v in kw.items() if f not in ('g_store', 'gn_out', 'sum_g', 'sum_gx')}
        if WGRAD_FROM_STORE and kw.get('g_store') is not None:
            # the data pass has left this layer's G (= g_skip + BatchNorm backward of gn_src at out) in g_store: the weight pass reads
            # that ONE tensor as its plain gradient instead of assembling G again from three (the pass is HBM-bound at large batches)
            wkw = {f: v for f, v in wkw.items() if f not in ('gn_src', 'out', 'g_skip') and not f.startswith('cbn_')}
            wkw['g_direct'] = kw['g_store']
        wkw['g_bias'] = g_bias[i]
        queued.append(((shape, I, O, k, valid), wkw, i))

    if chained:
        d = ConvNetBwdDesc()
        for i in range(nl):
            d.w[i] = w[i].data_ptr()
        for j in range(nb):
            d.gamma[j], d.beta[j] = gamma[j].data_ptr(), beta[j].data_ptr()
            d.save_mean[j], d.save_invstd[j] = ws[j, 2 * R].data_ptr(), ws[j, 2 * R + 1].data_ptr()
            d.acts[j], d.gn[j] = acts[j].data_ptr(), gn[j].data_ptr()
            d.sum_g[j], d.sum_gx[j] = sums[j, 0].data_ptr(), sums[j, 1].data_ptr()
        d.g_out = g_out.data_ptr()
        d.g_store[0], d.g_store[1] = stores[0].data_ptr(), stores[1].data_ptr()
        d.g_x = g_x.data_ptr() if g_x is not None else None
        if getattr(ctx, 'packs', None) is not None and all(p.current() for p in ctx.packs):
            for i in range(nl):                     # (stale images: the kernel splits the saved weights itself, bitwise the same)
                d.wpk[i] = ctx.packs[i].data_ptr()
        slots, d.ws_gen = _chain_slots(_chain_ws_floats(B, I0, O_out, Hh, Ww), dev)      # (kept alive up to the launch, see the forward)
        d.ws_zero = slots.data_ptr()
        if ctx.sinks is not None:                   # BatchNorm parameter gradients: added by the launch itself
            for j in range(nb):
                d.g_gamma[j], d.g_beta[j] = ctx.sinks[nl + 2 * j].data_ptr(), ctx.sinks[nl + 2 * j + 1].data_ptr()
        if cpl is not None:
            z, out, a, c = cpl
            g_y, g_ld = cpl_grads
            g_z = torch.empty_like(z)
            if ctx.cpl_sinks is not None:
                pa, pc, g_a, g_c = ctx.cpl_sinks[0].data_ptr(), ctx.cpl_sinks[1].data_ptr(), None, None
            else:
                g_ac = WS.zeros(2, dev)
                pa, pc = g_ac.data_ptr(), g_ac.data_ptr() + 4
                g_a, g_c = g_ac[0:1].view_as(a), g_ac[1:2].view_as(c)
            d.g_out = None
            d.cp_g_y, d.cp_g_ld, d.cp_z, d.cp_out = g_y.data_ptr(), g_ld.data_ptr(), z.data_ptr(), out.data_ptr()
            d.cp_a, d.cp_c, d.cp_g_z, d.cp_g_out, d.cp_g_a, d.cp_g_c = a.data_ptr(), c.data_ptr(), g_z.data_ptr(), g_out.data_ptr(), pa, pc
            d.cp_mode, d.cp_odd = ctx.cpl_meta
            d.cp_C = z.shape[1]
            from . import functional as NF
            pend = NF.PENDING_HEAD_BWD.pop(g_y.data_ptr(), None)
            if pend is not None:
                # g_y has not been computed yet: it is the data gradient of the NEXT step's head, left to this launch's prologue
                # (functional._GlowHeadW.backward)
                hg, hls, hW, hgx, small = pend
                Cz = z.shape[1]
                fits = (9 <= Cz <= 64) if small is None else (2 <= Cz <= 4)       # (MFMA head | thread-per-pixel head with its saved weight)
                if hgx.shape == z.shape and hg.shape == z.shape and hg.is_contiguous() and fits:
                    d.hd_g_h, d.hd_W, d.hd_ls = hg.data_ptr(), hW.data_ptr(), hls.data_ptr()
                else:
                    NF.flush_pending_head_bwd(pend)
        N.call('nf_convnet_chain_bwd', ctypes.addressof(d), B, I0, O_out, Hh, Ww, int(training), N.stream())

    layer(H, O_out, 1, nl - 1, in_=acts[nb - 1], weight=w[nl - 1], g_direct=g_out, gn_out=gn[nb - 1], sum_g=sums[nb - 1, 0],
          sum_gx=sums[nb - 1, 1], **in_bn(nb - 1))
    cpk = getattr(ctx, 'packs', None)
    pk = (lambda i: cpk[i].buf) if (cpk is not None and all(p.current() for p in cpk)) else (lambda i: None)
    for j in range(nb - 1, 0, -1):             # convolution j produced acts[j]; its consumer BatchNorm is j
        is_stream = (j % 2 == 0)
        store = (stores[(nb - 1 - j) // 2] if chained else torch.empty_like(acts[0])) if is_stream else None
        layer(H, H, 3, j, in_=acts[j - 1], weight=w[j], gn_src=gn[j], out=acts[j], g_skip=G_skip if is_stream else None,
              g_store=store, gn_out=gn[j - 1], sum_g=sums[j - 1, 0], sum_gx=sums[j - 1, 1], wpk=pk(j), **in_bn(j - 1), **cons_bn(j))
        if is_stream:
            G_skip = store
    layer(I0, H, 3, 0, in_=x, weight=w[0], gn_src=gn[0], out=acts[0], g_skip=G_skip, gn_out=g_x, wpk=pk(0), **cons_bn(0))

    direct = ctx.sinks is not None
    g_w = [torch.empty_like(t) for t in w]
    if direct:
        d_bias = ctx.sinks[:nl]
        d_bn = [tuple(ctx.sinks[nl + 2 * j:nl + 2 * j + 2]) for j in range(nb)]
    else:
        d_bias = [torch.empty(t.shape[0], dtype=torch.float32, device=dev) for t in w]
        d_bn = [(torch.empty(H, dtype=torch.float32, device=dev), torch.empty(H, dtype=torch.float32, device=dev))
                for _ in range(nb)]
    jobs = [(g_weff[i], g_w[i], w[i].numel(), w[i].numel(), slabs, False, w[i].shape[2] * w[i].shape[3]) for i in range(nl)]
    jobs += [(g_bias[i], d_bias[i], w[i].shape[0], GB, R, direct, 1) for i in range(nl)]
    bn_jobs = []
    if not (chained and direct):                # (the chain launch adds them into the sinks itself)
        for j in range(nb):
            bn_jobs.append((sums[j, 1], d_bn[j][0], H, H, R, direct, 1))       # g_gamma = sum g * xhat
            bn_jobs.append((sums[j, 0], d_bn[j][1], H, H, R, direct, 1))       # g_beta  = sum g
    if defer:
        for key, wkw, i in queued:             # (the tensors in wkw keep every operand alive until the flush)
            CONV_DEFER.layers.append((key, wkw, g_w[i], slabs, jobs[nl + i]))
        CONV_DEFER.sums += bn_jobs
    elif chained:
        CONV_DEFER.launch_layers([(key, wkw, g_w[i], slabs, jobs[nl + i]) for key, wkw, i in queued])
        if bn_jobs:
            _slab_sum(bn_jobs)
    else:
        _slab_sum(jobs + bn_jobs)
    grads = []
    for i in range(nl):
        grads += [g_w[i], None if direct else d_bias[i]]
    for j in range(nb):
        grads += [None if direct else d_bn[j][0], None if direct else d_bn[j][1], None, None, None]
    if valid and g_x is not None:
        g_x = g_x[:, :, :valid[0], :valid[1]].contiguous()
    return ((g_x, ) if cpl is None else (g_z, g_a, g_c)), tuple(grads)


def convnet_forward(net, x):
    """``net``: conditioners.ConvNet; returns the conditioner output (B, out_channels, H, W)."""
    tensors = _convnet_tensors(net)
    return _FusedConvNet.apply(x, net.training, CONV_DEFER.usable(tensors[0:12:2]), _convnet_packs(net), *tensors)


def coupling_fusable(net, z, mode):
    """the affine coupling (split map ``mode``) of an image tensor z can ride its conditioner's chain launches, both directions"""
    if not (CONV_COUPLING_ON and CONV_CHAIN_ON and CONV_CHAIN_BWD_ON and getattr(net, 'fused', False)):
        return False
    if not (z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and z.is_contiguous()):
        return False
    B, C, Hf, Wf = z.shape
    if mode == N.SPLIT_CHANNEL and C % 2 == 0:
        half = (B, C // 2, Hf, Wf)
    elif mode == N.SPLIT_CHECKER and Hf % 2 == 0 and Wf % 2 == 0:
        half = (B, 2 * C, Hf // 2, Wf // 2)
    else:
        return False
    from .conditioners import _sync_on
    if _sync_on() or not _convnet_usable_shape(net, half):
        return False
    convs, _ = _convnet_modules(net)
    c0 = getattr(convs[0], 'module', convs[0])
    c5 = getattr(convs[-1], 'module', convs[-1])
    if B * half[2] * half[3] < CONV_COUPLING_MIN_PX:       # (experiment knob: a level whose launch has very few workgroups)
        return False
    return c0.in_channels == half[1] and c5.out_channels == 2 * half[1] and _chain_usable(B, half[1], 2 * half[1], half[2], half[3])


def head_in_chain_ok(net, z, mode):
    """the forward of the Glow head in front of this coupling can ride the coupling's chain launch (csrc/conv_chain.hip: nf_cc_head_fwd):
    the coupling takes that launch, 9 .. 64 channels, and the conditioning half of one tile fits the head's LDS buffer"""
    if not coupling_fusable(net, z, mode):
        return False
    B, C, Hf, Wf = z.shape
    if not (2 <= C <= 4 or 9 <= C <= 64) or (Hf * Wf) % 16:      # (thread-per-pixel head | MFMA head)
        return False
    I0, Hh, Ww = (C // 2, Hf, Wf) if mode == N.SPLIT_CHANNEL else (2 * C, Hf // 2, Wf // 2)
    blocks = int(N.load().nf_convnet_chain_blocks(B, I0, 2 * I0, Hh, Ww))
    if blocks <= 0:
        return False
    px = -(-B * Hh * Ww // blocks)                           # pixels of a tile
    px = 1 << (px - 1).bit_length()
    if Hh * Ww > px:                                         # a sample over two tiles: its rows + one halo row each side
        return Hh * Ww == 2 * px and Wf >= 16 and I0 * (px // Ww + 2) * Ww <= 3840
    return (px // (Hh * Ww)) * I0 * Hh * Ww <= 3840


def convnet_coupling(net, x, z, ld, a, c, mode, odd, inverse=False):
    """y, ld = AffineCoupling(z; net(x)) with x = the untouched half of z, in the conditioner's own launch (coupling_fusable)."""
    tensors = _convnet_tensors(net)
    if inverse and torch.is_grad_enabled() and (z.requires_grad or x.requires_grad or ld.requires_grad):
        # the inverse-direction kernels build no graph: say so instead of returning detached results
        raise NotImplementedError('differentiating through the inverse direction of a fused image coupling is not supported '
                                  '(the reference never does: sampling runs under no_grad, main.py:109-116)')
    if inverse or not torch.is_grad_enabled():
        with torch.no_grad():
            class _Ctx:                                   # no graph: nothing is kept
                def save_for_backward(self, *t):
                    pass
            y = _cn_forward(_Ctx(), x, net.training, False, tensors, cpl=(z, ld, a, c, mode, odd, int(inverse)), packs=_convnet_packs(net))
        return y, ld
    return _FusedConvCoupling.apply(z, x, ld, a, c, mode, odd, net.training, CONV_DEFER.usable(tensors[0:12:2]), _convnet_packs(net),
                                    *tensors)
