"""flows.misc of the reference: the NaN/Inf forward hook (flows/misc.py:28-52), restated without its typo."""
import torch


def anomaly_hook(self, inputs, outputs):
    def bad(t):
        return isinstance(t, torch.Tensor) and t.is_floating_point() and not bool(torch.isfinite(t).all())
    outs = outputs if isinstance(outputs, (tuple, list)) else (outputs, )
    for i, o in enumerate(outs):
        if bad(o):
            raise RuntimeError('Found NaN/Inf in output %d of %s' % (i, self.__class__.__name__))


def __getattr__(name):
    """Names the engine does not replace (helpers such as flows/misc.py's free functions) come from the reference checkout."""
    from . import reference_module
    try:
        return getattr(reference_module('misc'), name)
    except ImportError as e:
        raise AttributeError('flows.misc has no %r in the engine and no reference checkout is reachable (%s)' % (name, e))
