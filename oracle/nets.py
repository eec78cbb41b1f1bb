"""
Conditioner networks of the coupling / autoregressive layers, restated functionally over a
``state_dict`` (``sd``: name -> tensor, reference key names).  ``p`` is always the key prefix of
the sub-module, ending with a dot.

These are NOT transform kernels, but the transforms cannot be exercised end to end without them:
MLP / ConvNet produce (t, s_raw) for AffineCoupling, the gated-attention stack produces the
mixture parameters of Flow++, MADE produces s and t of MAF.

BatchNorm running statistics inside ``sd`` are updated IN PLACE in training mode, exactly like the
``nn.BatchNorm*`` modules of the reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

WN_EPS = 1.0e-5      # flows/weight_norm.py:9


def wn_weight(sd, p):
    """w = v * g / (||v||_{dim 0} + eps), norm over dim 0 (flows/weight_norm.py:35-41)."""
    v, g = sd[p + 'weight_v'], sd[p + 'weight_g']
    return v * (g / (torch.norm(v, dim=0) + WN_EPS)).expand_as(v)


def wn_linear(x, sd, p):
    return F.linear(x, wn_weight(sd, p), sd[p + 'bias'])


def wn_conv(x, sd, p, padding):
    return F.conv2d(x, wn_weight(sd, p), sd[p + 'bias'], stride=1, padding=padding)


def batch_norm(x, sd, p, training):
    """nn.BatchNorm1d/2d defaults: eps 1e-5, momentum 0.1, affine, track_running_stats."""
    if training and (p + 'num_batches_tracked') in sd:
        sd[p + 'num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                        training, 0.1, 1.0e-5)


def _res_block(x, sd, p, training, conv):
    """ResBlockLinear / ResBlock2d with in == out channels, i.e. identity bridge (modules.py:342-390)."""
    lin = (lambda h, q: wn_conv(h, sd, q, 1)) if conv else (lambda h, q: wn_linear(h, sd, q))
    h = torch.relu(batch_norm(x, sd, p + 'net.0.', training))
    h = lin(h, p + 'net.2.module.')
    h = torch.relu(batch_norm(h, sd, p + 'net.3.', training))
    h = lin(h, p + 'net.5.module.')
    return x + h


def mlp(x, sd, p, training, n_blocks=2):
    """MLP(in, out, base_filters=32, n_blocks=2, weight_norm=True)  (modules.py:393-413)."""
    h = wn_linear(x, sd, p + 'in_block.0.module.')
    for i in range(n_blocks):
        h = _res_block(h, sd, p + 'mid_block.%d.' % i, training, conv=False)
    h = torch.relu(batch_norm(h, sd, p + 'out_block.0.', training))
    return wn_linear(h, sd, p + 'out_block.2.module.')


def convnet(x, sd, p, training, n_blocks=2):
    """ConvNet: 3x3 in-conv, two residual blocks, 1x1 out-conv (modules.py:416-438)."""
    h = wn_conv(x, sd, p + 'in_block.0.module.', 1)
    for i in range(n_blocks):
        h = _res_block(h, sd, p + 'mid_block.%d.' % i, training, conv=True)
    h = torch.relu(batch_norm(h, sd, p + 'out_block.0.', training))
    return wn_conv(h, sd, p + 'out_block.2.module.', 0)


# ---- Flow++ conditioner (coupling.py:142-166, modules.py:500-578) -------------------------------------------------------

def _gated(x, sd, p, conv):
    """GatedLinear / GatedConv2d (modules.py:500-538): concat-ELU, op, concat-ELU, gate, residual."""
    C = x.shape[1]
    h = F.elu(torch.cat([x, -x], dim=1))
    if conv:
        h = F.conv2d(h, sd[p + 'op.weight'], sd[p + 'op.bias'], stride=1, padding=1)
    else:
        h = F.linear(h, sd[p + 'op.weight'], sd[p + 'op.bias'])
    h = F.elu(torch.cat([h, -h], dim=1))
    y, a = torch.split(h, C, dim=1)
    return x + y * torch.sigmoid(a)


def gated_attn(x, sd, p, filters, heads=4):
    """GatedAttn (modules.py:541-578); note the reference names the projections V, K, Q in that split order."""
    shape = x.shape
    B, C = shape[0], shape[1]
    D = filters // heads
    xr = (x + sd[p + 'pos_emb']).reshape(B, C, -1)
    prm = F.conv1d(xr, sd[p + 'conv1.weight'], sd[p + 'conv1.bias']).reshape(B, 3 * heads, D, -1)
    V, K, Q = torch.split(prm, heads, dim=1)
    Wt = torch.matmul(V.permute(0, 1, 3, 2), K) / math.sqrt(D)
    Wt = F.softmax(Wt, dim=2)
    A = torch.matmul(Q, Wt).reshape(B, C, -1)
    y = F.conv1d(A, sd[p + 'conv2.weight'], sd[p + 'conv2.bias'])
    y, a = torch.split(y, C, dim=1)
    return x + (y * torch.sigmoid(a)).reshape(shape)


def flowpp_net(x, sd, p, conv, base_filters=32):
    """the nn.Sequential of MixLogAttnCoupling (coupling.py:142-166)."""
    if conv:
        h = F.conv2d(x, sd[p + '0.weight'], sd[p + '0.bias'], stride=1, padding=1)
    else:
        h = F.linear(x, sd[p + '0.weight'], sd[p + '0.bias'])
    h = _gated(h, sd, p + '1.', conv)
    ln_shape = tuple(sd[p + '2.weight'].shape)
    h = F.layer_norm(h, ln_shape, sd[p + '2.weight'], sd[p + '2.bias'], 1.0e-5)
    h = gated_attn(h, sd, p + '3.', base_filters)
    h = F.layer_norm(h, ln_shape, sd[p + '4.weight'], sd[p + '4.bias'], 1.0e-5)
    if conv:
        return F.conv2d(h, sd[p + '5.weight'], sd[p + '5.bias'], stride=1, padding=1)
    return F.linear(h, sd[p + '5.weight'], sd[p + '5.bias'])


# ---- MADE (maf.py:9-85) -------------------------------------------------------------------------------------------------

def made_masks(D, num_hidden, base_filters=32, rng=None):
    """_create_masks (maf.py:66-85).  Draws from ``rng.randint`` exactly like the reference draws from the
    global ``np.random`` on EVERY forward; for D == 2 the draw is degenerate (all hidden degrees 0)."""
    rng = np.random if rng is None else rng
    m_prev = np.arange(D)
    hidden = [D] + [base_filters] * num_hidden
    masks = []
    for in_dims, out_dims in zip(hidden[:-1], hidden[1:]):
        min_k = min(int(m_prev.min()), D - 2)
        m = rng.randint(min_k, D - 1, size=(out_dims))
        masks.append(torch.from_numpy((m_prev[None, :] <= m[:, None]).astype(np.float32)))   # M[k,:] = m_prev <= m[k]
        m_prev = m
    M = np.zeros((D, hidden[-1]), dtype=np.float32)
    for k in range(hidden[-1]):
        M[m_prev[k] + 1:, k] = 1.0                                                           # maf.py:81-82
    masks.append(torch.from_numpy(M))
    return masks


def made(z, sd, p, num_hidden, training, masks):
    """MADE.forward with use_companion=False (maf.py:49-64)."""
    h = z
    for i in range(num_hidden):
        h = F.linear(h, sd[p + 'weights.%d' % i] * masks[i], sd[p + 'biases.%d' % i])
        h = torch.relu(batch_norm(h, sd, p + 'bnorms.%d.' % i, training))
    return F.linear(h, sd[p + 'weights.%d' % num_hidden] * masks[-1], sd[p + 'biases.%d' % num_hidden])
