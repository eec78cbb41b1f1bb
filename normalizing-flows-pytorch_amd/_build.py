"""
Builds libnfhip.so IN-TREE from csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU).
Objects are compiled in parallel and only when a source (or a header) is newer than its object.

    python -m <pkg>._build        or        __graft_entry__.build()
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libnfhip.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: libnfhip.so cannot be built (ROCm toolchain required)')


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """compile every csrc/*.hip and link libnfhip.so; returns its path."""
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    sources = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    if not sources:
        raise RuntimeError('no HIP sources under ' + CSRC)
    jobs = []
    for src in sources:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (os.path.basename(src), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in sources]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link of libnfhip.so failed:\n' + r.stderr)
    # the C header is the single source of the ctypes prototypes and constants: keep a copy next to the library, so that the
    # package directory alone (without the repository's include/) is loadable
    src = os.path.join(HERE, '..', 'include', 'nfhip.h')
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(HERE, 'nfhip.h'))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
