"""
Zero-initialised scratch for the backward accumulators and the BatchNorm statistics of one training step.

Every fused op needs a few hundred zeroed floats (atomic accumulators).  Allocating them with ``torch.zeros`` costs
one fill launch per op -- hundreds per step at ~4 us each, in a regime that is purely launch-bound.  ``ZeroArena`` is
a bump allocator over ONE device buffer that the trainer zeroes once per step (one memset node in the hipGraph);
outside a trainer step it is inactive and callers fall back to ``torch.zeros``.
"""
import torch


class ZeroArena:
    def __init__(self):
        self.buf = None
        self.off = 0
        self.high = 0
        self.active = False

    def begin(self, device):
        """start a step: (re)zero the arena; sized from the demand observed in earlier steps."""
        want = max(int(self.high * 1.25), 1 << 14)
        if self.buf is None or self.buf.device != device or (self.buf.numel() < self.high and not _capturing()):
            self.buf = torch.zeros(want, dtype=torch.float32, device=device)
        else:
            self.buf.zero_()
        self.off = 0
        self.active = True

    def end(self):
        self.high = max(self.high, self.off)
        self.active = False

    def zeros(self, n, device):
        """a zero-filled fp32 vector of n elements (16-byte aligned slice of the arena when a step is open)."""
        n_pad = (int(n) + 3) & ~3
        if self.active and self.buf is not None and self.buf.device == device:
            o = self.off
            self.off += n_pad
            if o + n_pad <= self.buf.numel():
                return self.buf[o:o + n]
        return torch.zeros(n, dtype=torch.float32, device=device)


def _capturing():
    try:
        return torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


ARENA = ZeroArena()


def zeros(n, device):
    return ARENA.zeros(n, device)
