"""
ctypes binding of libnfhip.so (C ABI declared in include/nfhip.h).

There is NO fallback: if the library is missing or a tensor is not on the GPU, the call raises.  torch is used
only for device memory and the current HIP stream; the library itself links nothing of torch.
"""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libnfhip.so')
HEADER = os.path.join(HERE, '..', 'include', 'nfhip.h')

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


def header_constant(name):
    """integer value of a ``#define NAME <expr>`` in include/nfhip.h (integer literals, + * and parentheses only)."""
    m = re.search(r'^#define\s+%s\s+([0-9\s\*\+\(\)]+?)\s*(/\*.*)?$' % re.escape(name), open(HEADER).read(), re.M)
    if m is None:
        raise NativeLibraryError('include/nfhip.h does not define %s as an integer expression' % name)
    return int(eval(m.group(1), {'__builtins__': {}}, {}))       # the character class above admits arithmetic only


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
        _lib = lib
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
    return t.data_ptr()


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise NativeLibraryError('%s failed with code %d' % (name, rc))


def persistent_timeouts():
    """spin loops of the persistent kernels that gave up since load (must be 0); synchronises the device."""
    import ctypes as _c
    v = _c.c_int(0)
    call('nf_persistent_timeouts', _c.byref(v))
    return int(v.value)
