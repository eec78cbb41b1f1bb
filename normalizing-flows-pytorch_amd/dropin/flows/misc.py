"""flows.misc of the reference: the NaN/Inf forward hook (flows/misc.py:28-52), restated without its typo."""
import torch


def anomaly_hook(self, inputs, outputs):
    def bad(t):
        return isinstance(t, torch.Tensor) and t.is_floating_point() and not bool(torch.isfinite(t).all())
    outs = outputs if isinstance(outputs, (tuple, list)) else (outputs, )
    for i, o in enumerate(outs):
        if bad(o):
            raise RuntimeError('Found NaN/Inf in output %d of %s' % (i, self.__class__.__name__))
