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
