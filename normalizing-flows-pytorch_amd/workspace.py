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
        self.last = 0           # elements the previous step handed out
        self.zeroed = 0         # elements of buf that are zero for the step under way
        self.high = 0
        self.active = False
        self.retired = []       # outgrown buffers a hipGraph was captured against: its kernel arguments still point into them
        self.captured = False   # a capture happened while the CURRENT buffer was the arena
        self.step_state = {}    # per-step values of the arena's clients (fused_conv._chain_slots: one exchange-slot buffer per step)

    def begin(self, device):
        """start a step: zero as much of the arena as the PREVIOUS step used (+ 25 %).  Sizing by the all-time maximum would make a
        small model pay for a large one that ran earlier in the process (bench.py: RealNVP moons after Glow-CIFAR: a 0.9 GB memset,
        0.2 ms, in front of every 1.9 ms step)."""
        need = max(int(self.last * 1.25), 1 << 14)
        if self.buf is None or self.buf.device != device or (self.buf.numel() < need and not _capturing()):
            if self.buf is not None and self.captured:   # only a buffer some captured graph replays against must outlive its use
                self.retired.append(self.buf)
            self.buf = torch.zeros(need, dtype=torch.float32, device=device)
            self.captured = False
            self.zeroed = need
        else:
            self.zeroed = min(need, self.buf.numel())
            if self.buf.is_cuda:
                from . import _native as N
                N.call('nf_zero_fill', self.buf.data_ptr(), self.zeroed, N.stream())
            else:
                self.buf[:self.zeroed].zero_()
        if _capturing():
            self.captured = True
        self.off = 0
        self.active = True
        self.step_state = {}

    def end(self):
        self.last = self.off
        self.high = max(self.high, self.off)
        self.active = False

    def zeros(self, n, device):
        """a zero-filled fp32 vector of n elements (16-byte aligned slice of the arena when a step is open)."""
        n_pad = (int(n) + 3) & ~3
        if self.active and self.buf is not None and self.buf.device == device:
            o = self.off
            self.off += n_pad                        # (counted even when it does not fit: the next step's arena is sized from it)
            if o + n_pad <= self.zeroed:
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


def zeros_owned(shape, device):
    """zero-filled fp32 tensor of ``shape`` for a value that travels through autograd (the log-det accumulator the layers update in place,
    the loss): inside a trainer step a piece of the arena -- as a tensor of its own over the arena's storage, NOT a view of the arena
    tensor (autograd would rebase the whole arena behind an in-place update of a view) -- else ``torch.zeros``.  It lives until the next
    step's memset: FlowTrainer hands a copy to its caller outside hipGraph capture."""
    n = 1
    for d in shape:
        n *= int(d)
    if ARENA.active and ARENA.buf is not None and ARENA.buf.device == device:
        t = ARENA.zeros(n, device)
        if t.untyped_storage().data_ptr() == ARENA.buf.untyped_storage().data_ptr():
            return torch.empty(0, dtype=torch.float32, device=device).set_(t.untyped_storage(), t.storage_offset(), tuple(shape))
        return t.reshape(tuple(shape)) if len(shape) != 1 else t
    return torch.zeros(tuple(shape), dtype=torch.float32, device=device)


def in_arena(t):
    return ARENA.buf is not None and t.is_cuda == ARENA.buf.is_cuda and t.untyped_storage().data_ptr() == ARENA.buf.untyped_storage().data_ptr()
