"""
CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950 without a GPU, loads, and exports
every symbol include/nfhip.h declares (no compute calls here); the product path refuses CPU tensors loudly.
"""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nfhip.h')


@pytest.fixture(scope='module')
def built(pkg):
    path = pkg.build()
    assert os.path.exists(path)
    return path


def test_library_exports_every_declared_symbol(pkg, built):
    protos = pkg._native.header_prototypes(HEADER)
    assert len(protos) >= 20
    lib = ctypes.CDLL(built)
    for name in protos:
        assert hasattr(lib, name), 'libnfhip.so does not export %s' % name
    out = subprocess.run(['nm', '-D', '--defined-only', built], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (nf_\w+)', out))
    assert exported == set(protos), 'header and library disagree: %s' % (exported ^ set(protos))


def test_header_is_plain_c(built):
    """the boundary is a C ABI: the header must compile as C (no torch / C++ types in the signatures)."""
    src = '#include "%s"\nint main(void) { return nf_version == 0; }\n' % HEADER
    r = subprocess.run(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', '-'], input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_version_and_argument_errors_without_gpu(pkg, built):
    lib = pkg._native.load()
    assert lib.nf_version() >= 100
    # argument validation happens on the host before any launch: odd feature count is rejected
    rc = lib.nf_half_gather(None, None, 0, 0, 0, 4, 3, 1, 1, None)
    assert rc == 10001
    rc = lib.nf_affine_coupling_fwd(None, None, None, 0, None, None, None, None, 7, 0, 0, 4, 2, 1, 1, None)
    assert rc == 10001


def test_gfx950_code_object(built, tmp_path):
    # llvm-objdump --offloading EXTRACTS every code object next to the file it is given: run it on a temporary copy, so that the
    # package directory (which travels to the GPU box) stays free of the ~9 MB of libnfhip.so.N.hipv4-* dumps
    import shutil
    work = tmp_path / 'libnfhip.so'
    shutil.copyfile(built, work)
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', str(work)], capture_output=True, text=True,
                       cwd=str(tmp_path))
    if r.returncode != 0:
        pytest.skip('llvm-objdump --offloading unavailable')
    assert 'gfx950' in r.stdout
    assert not [f for f in os.listdir(os.path.dirname(built)) if f.startswith('libnfhip.so.')], 'code-object dumps in the package'


def test_no_cpu_fallback(pkg, built):
    from types import SimpleNamespace as NS
    net = pkg.Glow((2, ), '2d', NS(layers=1))
    with pytest.raises(RuntimeError, match='no CPU'):
        net(torch.randn(8, 2))
    with pytest.raises(RuntimeError):
        pkg.functional.logit(torch.rand(2, 3), torch.zeros(2), 0.01)


def test_product_does_not_import_oracle():
    pk = os.path.join(ROOT, 'normalizing-flows-pytorch_amd')
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
                assert 'from oracle' not in text and 'import oracle' not in text, f


def test_ctypes_struct_layout_matches_header(pkg):
    """fused.py mirrors the three descriptor structs by hand: field order and count must equal the header's."""
    fused = __import__('importlib').import_module(pkg.__name__ + '.fused')
    text = re.sub(r'/\*.*?\*/', ' ', open(HEADER).read(), flags=re.S)
    for struct, cls in [('nf_linear_desc', fused.LinearDesc), ('nf_linear_bwd_desc', fused.LinearBwdDesc),
                        ('nf_weight_grad_desc', fused.WeightGradDesc), ('nf_wn_desc', fused.WnDesc)]:
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (struct, struct), text, flags=re.S).group(1)
        names = [re.sub(r'\W', '', f.strip().split()[-1]) for f in body.split(';') if f.strip()]
        got = [n.rstrip('_') for n, _ in cls._fields_]
        assert got == names, (struct, got, names)
        for (n, ty), decl in zip(cls._fields_, [f.strip() for f in body.split(';') if f.strip()]):
            assert ('*' in decl) == (ty is ctypes.c_void_p), decl
            if '*' not in decl:
                assert decl.startswith('int ') and ty is ctypes.c_int, decl


def test_committed_pmc_summary_names_current_kernels():
    """bench.pmc_traffic reads the newest profiles/rNN_pmc.json: every kernel it holds traffic for must still exist in csrc/ (a
    stale file from an earlier round would silently attach old counters to new launches)."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc.json')))
    if not files:
        pytest.skip('no PMC summary committed yet')
    src = ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'normalizing-flows-pytorch_amd', 'csrc', '*.hip')))
    data = json.load(open(files[-1]))
    kernels = [k for k, v in data.items() if isinstance(v, dict) and k.startswith('k_')]
    assert kernels, files[-1]
    for k in kernels:
        assert re.search(r'\b%s\b' % re.escape(k), src), '%s holds counters of %s, which no longer exists' % (os.path.basename(files[-1]), k)
        for key, e in data[k].items():
            assert key.split(':')[0].isdigit() and e.get('traffic_bytes', 1) > 0, (k, key)
    # the default bench line's C4 kernel must be covered once the round's collection has run
    bench_src = open(os.path.join(ROOT, 'bench.py')).read()
    for k in re.findall(r"pmc_traffic\('(k_\w+)'", bench_src):
        assert re.search(r'\b%s\b' % re.escape(k), src), k


def test_convnet_descriptor_layouts_match_header(pkg):
    """fused_conv.py mirrors the chain kernels' descriptors (arrays of pointers + a few ints) by hand"""
    fc = __import__('importlib').import_module(pkg.__name__ + '.fused_conv')
    text = re.sub(r'/\*.*?\*/', ' ', open(HEADER).read(), flags=re.S)
    for struct, cls in [('nf_convnet_desc', fc.ConvNetDesc), ('nf_convnet_bwd_desc', fc.ConvNetBwdDesc)]:
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (struct, struct), text, flags=re.S).group(1)
        want = []
        for decl in [f.strip() for f in body.split(';') if f.strip()]:
            ptr = '*' in decl
            for part in decl.split(','):                         # "int cp_mode, cp_odd, cp_C, cp_inverse"
                m = re.search(r'(\w+)\s*(?:\[(\d+)\])?\s*$', part.strip())
                want.append((m.group(1), int(m.group(2)) if m.group(2) else 0, ptr))
        got = []
        for n, ty in cls._fields_:
            if hasattr(ty, '_length_'):
                got.append((n, ty._length_, ty._type_ is ctypes.c_void_p))
            else:
                got.append((n, 0, ty is ctypes.c_void_p))
        assert got == want, (struct, [g for g, w in zip(got, want) if g != w], len(got), len(want))


def test_conv_wgrad_slabs_host_logic(pkg):
    """nf_conv_wgrad_slabs (host only): one workgroup per compute unit over a launch -- 256 / layers slabs per layer, never more than the
    layer's 128-pixel tiles or the 128-slab cap; 0 for an empty problem."""
    import ctypes
    lib = pkg._native.load()
    f = lib.nf_conv_wgrad_slabs
    f.argtypes, f.restype = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int], ctypes.c_int
    assert f(64, 16, 16, 16) == 16          # 128 tiles, 16 layers
    assert f(64, 8, 8, 16) == 16            # 32 tiles
    assert f(64, 4, 4, 16) == 8             # 8 tiles in all
    assert f(64, 16, 16, 4) == 64
    assert f(64, 16, 16, 1) == 128          # the cap of nf_conv_bwd_slabs
    assert f(512, 32, 32, 16) == 16
    assert f(0, 16, 16, 16) == 0 and f(64, 16, 16, 0) == 0
