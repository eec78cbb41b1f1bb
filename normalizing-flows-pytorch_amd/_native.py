"""
ctypes binding of libnfhip.so (C ABI declared in include/nfhip.h).

There is NO fallback: if the library is missing or a tensor is not on the GPU, the call raises.  torch is used
only for device memory and the current HIP stream; the library itself links nothing of torch.
"""
import ctypes
import functools
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libnfhip.so')
HEADER = os.path.join(HERE, '..', 'include', 'nfhip.h')
if not os.path.exists(HEADER):
    HEADER = os.path.join(HERE, 'nfhip.h')               # the copy _build.build() leaves next to the library

# enum mirrors of include/nfhip.h
SPLIT_1D, SPLIT_CHECKER, SPLIT_CHANNEL, SPLIT_NONE = 0, 1, 2, 3
OP_ACTNORM, OP_FLOWBN = 0, 1

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_CTYPES = {'const float*': _P, 'float*': _P, 'int*': _P, 'char*': _P, 'nf_stream_t': _P, 'int': _I, 'int64_t': _L,
           'float': _F}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def header_prototypes(path=HEADER):
    """parse `int nf_xxx(args);` prototypes out of include/nfhip.h -> {name: [ctypes...]} (single source of truth)."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(nf_\w+)\s*\(([^)]*)\)\s*;', text):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                ty = a.rsplit(' ', 1)[0] if not a.endswith('*') else a
                ty = ty.replace(' *', '*')
                if ty.endswith('*'):
                    types.append(_P)
                elif ty in _CTYPES:
                    types.append(_CTYPES[ty])
                else:
                    raise NativeLibraryError('unknown C type %r in prototype of %s' % (ty, name))
        protos[name] = types
    return protos


@functools.lru_cache(maxsize=None)
def _header_constants():
    """every ``#define NAME <integer expression>`` of include/nfhip.h (integer literals, + * and parentheses only), parsed ONCE:
    the fused Functions ask for these on every call (a regex scan of the 40 KB header cost ~250 us each time)."""
    out = {}
    for m in re.finditer(r'^#define\s+(\w+)\s+([0-9\s\*\+\(\)]+?)\s*(/\*.*)?$', open(HEADER).read(), re.M):
        try:
            out[m.group(1)] = int(eval(m.group(2), {'__builtins__': {}}, {}))   # the character class admits arithmetic only
        except SyntaxError:
            pass
    return out


def header_constant(name):
    """integer value of a ``#define NAME <expr>`` in include/nfhip.h."""
    try:
        return _header_constants()[name]
    except KeyError:
        raise NativeLibraryError('include/nfhip.h does not define %s as an integer expression' % name)


def load():
    """dlopen libnfhip.so (after torch, so that it binds to the HIP runtime torch already loaded)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError('libnfhip.so is not built (%s): run `python __graft_entry__.py` or '
                                     '`__graft_entry__.build()`; there is no CPU fallback' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in header_prototypes().items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise NativeLibraryError('libnfhip.so does not export %s (stale build?)' % name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = lib               # no HIP runtime call here: the sticky error word is armed per DEVICE, lazily (_arm_device)
    return _lib


def is_built():
    return os.path.exists(LIB_PATH)


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """device pointer of a contiguous fp32/int32 GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeLibraryError('nfhip kernels need a GPU (HIP) tensor, got device %s; the MI355X engine has no CPU '
                                 'path' % t.device)
    if not t.is_contiguous():
        raise NativeLibraryError('nfhip kernels need contiguous tensors')
    if t.dtype not in (torch.float32, torch.int32):
        raise NativeLibraryError('nfhip kernels are fp32 (int32 for flags), got %s' % t.dtype)
    if t.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's stream and read that device's copy of the library's globals (spin limit, error word,
        # deterministic mode): a tensor of another GPU would be dereferenced from the wrong device
        raise NativeLibraryError('tensor on cuda:%s but the current device is cuda:%d: one process per GPU (torch.cuda.set_device) or '
                                 'wrap the call in torch.cuda.device(...)' % (t.device.index, torch.cuda.current_device()))
    return t.data_ptr()


class PersistentKernelTimeout(NativeLibraryError):
    """a software grid exchange of a persistent kernel gave up waiting: some launch since the last reset produced garbage
    (its workgroups were not co-resident: GPU shared with another job, CU mask, work on another stream)."""


_err_word = None      # ctypes view of the pinned host word the kernels set when a spin loop gives up (nf_persistent_config)
_armed = set()        # device ordinals whose copies of the spin limit / error-word pointer have been written


def _arm_device():
    """nf_persistent_config writes device globals (hipMemcpyToSymbol): they exist once per DEVICE, so every device a kernel is
    launched on is armed when the first launch on it happens -- not at load(), which under torchrun runs before
    torch.cuda.set_device(local_rank) and would arm (and open a context on) GPU 0 from every rank."""
    global _err_word
    dev = torch.cuda.current_device()
    if dev not in _armed:
        p = ctypes.c_void_p()
        rc = load().nf_persistent_config(SPIN_LIMIT, 0, ctypes.byref(p))
        if rc != 0:
            raise NativeLibraryError('nf_persistent_config failed with code %d on device %d' % (rc, dev))
        _err_word = ctypes.c_uint.from_address(p.value)     # ONE pinned, portable host word shared by all devices
        if _det:
            rc = load().nf_deterministic(1)                  # (device globals as well: per device)
            if rc != 0:
                raise NativeLibraryError('nf_deterministic failed with code %d on device %d' % (rc, dev))
        _armed.add(dev)


SPIN_LIMIT = 1 << 22        # poll budget of one spin loop of a persistent kernel (persistent_reset(spin_limit=...) changes it)
_det = os.environ.get('NF_DETERMINISTIC', '0') == '1'   # deterministic mode wanted (applied to every device when it is armed)


def deterministic(on=None):
    """deterministic mode of the kernels (csrc/nf_det.h, include/nfhip.h:nf_deterministic): batch sums that meet at one address by
    float atomics are added in a fixed order, two runs from identical inputs are bit-identical.  ``deterministic()`` returns the
    current setting, ``deterministic(True / False)`` switches it (synchronises the device).  NF_DETERMINISTIC=1 switches it on at
    start-up.  A verification mode: the ordered tails of the kernels serialise."""
    global _det
    if on is None:
        return _det
    _det = bool(on)
    if torch.cuda.is_available():
        for dev in sorted(_armed):
            with torch.cuda.device(dev):
                rc = load().nf_deterministic(1 if _det else 0)
                if rc != 0:
                    raise NativeLibraryError('nf_deterministic failed with code %d on device %d' % (rc, dev))
    return _det


def deterministic_timeouts():
    """turnstile waits of the deterministic mode that gave up (must be 0); synchronises the device."""
    v = ctypes.c_int(0)
    rc = load().nf_deterministic_timeouts(ctypes.byref(v))
    if rc != 0:
        raise NativeLibraryError('nf_deterministic_timeouts failed with code %d' % rc)
    return int(v.value)


def _error_word():
    if not torch.cuda.is_available():
        return None
    _arm_device()
    return _err_word


def check_persistent():
    """raise if any persistent kernel launched so far gave up on a grid exchange (no device synchronisation: the word is
    pinned host memory, so a failure surfaces at the first check after the kernel hit it)."""
    w = _error_word()
    if w is not None and w.value != 0:
        raise PersistentKernelTimeout(
            'a persistent kernel timed out in a grid-wide exchange (%d spin loops gave up): its workgroups were not co-resident, '
            'the results since are invalid.  Is the GPU shared or CU-masked?  NF_GLOW_FLOW=0 NF_MAF_FLOW=0 select the '
            'multi-launch paths; _native.persistent_reset() clears the flag' % persistent_timeouts())


def persistent_reset(spin_limit=None):
    """clear the counters and the error word (synchronises); optionally set the poll budget of the spin loops."""
    global _err_word
    p = ctypes.c_void_p()
    lim = int(spin_limit) if spin_limit is not None else SPIN_LIMIT
    rc = load().nf_persistent_config(lim, 1, ctypes.byref(p))
    if rc != 0:
        raise NativeLibraryError('nf_persistent_config failed with code %d' % rc)
    _err_word = ctypes.c_uint.from_address(p.value)
    _armed.add(torch.cuda.current_device())


@functools.lru_cache(maxsize=None)
def persistent_capacity():
    """(mlp_blocks, maf_blocks): workgroups of the persistent kernel families the device holds at once (occupancy x CUs)."""
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    rc = load().nf_persistent_capacity(ctypes.byref(a), ctypes.byref(b))
    if rc != 0:
        raise NativeLibraryError('nf_persistent_capacity failed with code %d' % rc)
    return int(a.value), int(b.value)


def mlp_max_rows():
    """largest batch the MLP-chain family of persistent kernels takes: the header's cap, lowered to what is co-resident"""
    return min(header_constant('NF_MLP_MAX_ROWS'),
               min(header_constant('NF_MLP_MAX_BLOCKS'), persistent_capacity()[0]) * header_constant('NF_MLP_ROWS_PER_BLOCK'))


def maf_max_rows():
    return min(header_constant('NF_MAF_MAX_ROWS'),
               min(header_constant('NF_MAF_MAX_BLOCKS'), persistent_capacity()[1]) * header_constant('NF_MAF_ROWS_PER_BLOCK'))


_timing = None        # measurement hook (bench.py): {'name': entry point, 'match': f(args) -> bool, 'events': [(start, stop)]}


class timed_launches:
    """context: every call of the C-ABI entry point ``name`` whose arguments satisfy ``match`` is bracketed by two HIP events
    recorded on the launch stream (the current torch stream = the stream handed to the launcher), so that a real train step can
    report the duration of its dominant kernel AS THE STEP LAUNCHES IT.  ``mean_us()`` after a synchronise."""

    def __init__(self, name, match=None):
        self.rec = {'name': name, 'match': match, 'events': []}

    def __enter__(self):
        global _timing
        self._old, _timing = _timing, self.rec
        return self

    def __exit__(self, *exc):
        global _timing
        _timing = self._old
        return False

    def durations_us(self):
        torch.cuda.synchronize()
        return [a.elapsed_time(b) * 1e3 for a, b in self.rec['events']]


def call(name, *args):
    if torch.cuda.current_device() not in _armed:
        _arm_device()
    t = _timing
    if t is not None and t['name'] == name and (t['match'] is None or t['match'](args)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(load(), name)(*args)
        b.record()
        t['events'].append((a, b))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise NativeLibraryError('%s failed with code %d' % (name, rc))
    w = _err_word
    if w is not None and w.value != 0:
        check_persistent()


def persistent_timeouts():
    """spin loops of the persistent kernels that gave up since load / the last reset (must be 0); synchronises the device."""
    import ctypes as _c
    v = _c.c_int(0)
    rc = load().nf_persistent_timeouts(_c.byref(v))
    if rc != 0:
        raise NativeLibraryError('nf_persistent_timeouts failed with code %d' % rc)
    return int(v.value)
