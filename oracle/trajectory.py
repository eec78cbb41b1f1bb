"""
TEST INFRASTRUCTURE (never imported by the product path): a training trajectory of the oracle -- the reference's
``Model.train_on_batch`` (main.py:78-92: forward, NLL under N(0, I), autograd backward, Adam lr 1e-4) run for a few steps on
one resident batch, recording what every step saw BEFORE its optimizer update: z, log-det, loss and every gradient.

Run in float32 this is the reference's CPU path; run in float64 it is the yard-stick that says how far the float32 CPU path
itself is from exact arithmetic at a given depth (the parity tests at the BASELINE sizes bound the GPU error by the 1e-5
bar plus a small multiple of that measured distance: a 32-step flow of BatchNorm-conditioned couplings amplifies rounding,
and two correct fp32 implementations cannot agree better than either agrees with the exact result).
"""
import torch

from . import models as om
from . import transforms as tf


def cast_state(sd, dtype):
    """deep copy of a state_dict with the floating-point tensors in ``dtype`` (integer pivots etc. unchanged)."""
    return {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()) for k, v in sd.items()}


def run(kind, dims, datatype, layers, sd0, y, steps, mixtures=None, dtype=torch.float32, lr=1.0e-4, record=None,
        actnorm_initialized=False):
    """``steps`` train steps from ``sd0`` on the batch ``y``.  Returns (records, oracle); records[i] for the 1-based step
    numbers in ``record`` (default: all) = dict(z, ld, loss, grads{name: tensor}) as seen by step i before its Adam update."""
    sd = cast_state(sd0, dtype)
    ora = om.FlowOracle(kind, dims, datatype, layers, sd, mixtures=mixtures, training=True,
                        actnorm_initialized=actnorm_initialized).requires_grad_(True)
    params = ora.parameters()
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    yy = y.detach().cpu().to(dtype)
    out = {}
    for i in range(1, steps + 1):
        opt.zero_grad(set_to_none=True)
        z, ld = ora.forward(yy)
        loss = tf.nll_loss(z, ld)
        loss.backward()
        if record is None or i in record:
            out[i] = dict(z=z.detach().clone(), ld=ld.detach().clone(), loss=loss.detach().clone(),
                          grads={k: v.grad.detach().clone() for k, v in params.items() if v.grad is not None})
        if i < steps:
            opt.step()
    return out, ora


def forward_only(kind, dims, datatype, layers, sd, y, mixtures=None, dtype=torch.float32, training=True):
    """(z, ld) of one forward pass over the state ``sd`` (ActNorm treated as initialised: the state is a trained one)."""
    ora = om.FlowOracle(kind, dims, datatype, layers, cast_state(sd, dtype), mixtures=mixtures, training=training,
                        actnorm_initialized=True)
    with torch.no_grad():
        return ora.forward(y.detach().cpu().to(dtype))


def run_slice(kind, dims, datatype, layers, sd, a, b, z_in, ld_in, mixtures=None, dtype=torch.float64, y=None):
    """layers[a:b] of the stack (``FlowOracle.plan`` indices = positions in ``net.layers``) on (z_in, ld_in), then the NLL of
    the slice's own output: returns dict(z, ld, loss, g_in, grads{name}).  With ``y`` given instead of ``z_in``, the slice input
    is computed first: layers[:a] on ``y`` in the same dtype (no graph)."""
    ora = om.FlowOracle(kind, dims, datatype, layers, cast_state(sd, dtype), mixtures=mixtures, training=True,
                        actnorm_initialized=True).requires_grad_(True)
    if y is not None:
        with torch.no_grad():
            z_in = y.detach().cpu().to(dtype)
            ld_in = torch.zeros(z_in.shape[0], dtype=dtype)
            for L in ora.plan[:a]:
                z_in, ld_in = ora._apply(L, z_in, ld_in, False)
    z0 = z_in.detach().cpu().to(dtype).clone().requires_grad_(True)
    z, ld = z0, ld_in.detach().cpu().to(dtype).clone()
    for L in ora.plan[a:b]:
        z, ld = ora._apply(L, z, ld, False)
    loss = tf.nll_loss(z, ld)
    loss.backward()
    pre = tuple(L['prefix'] for L in ora.plan[a:b])
    grads = {k: v.grad.detach().clone() for k, v in ora.parameters().items() if v.grad is not None and k.startswith(pre)}
    return dict(z_in=z0.detach().clone(), ld_in=ld_in.detach().clone(), z=z.detach().clone(), ld=ld.detach().clone(),
                loss=loss.detach().clone(), g_in=z0.grad.detach().clone(), grads=grads)
