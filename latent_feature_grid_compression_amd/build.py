"""Build liblfgc.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m latent_feature_grid_compression_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
OBJ_DIR = os.path.join(CSRC, 'build')
LIB_PATH = os.path.join(PKG_DIR, 'liblfgc.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-ffp-contract=off', '-fno-slp-vectorize', '-fPIC', '-std=c++17', '-Wno-unused-result']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _fingerprint() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(PKG_DIR, '..', 'include')):
        for f in sorted(os.listdir(root)):
            if f.endswith(('.hip', '.h')):
                with open(os.path.join(root, f), 'rb') as fh:
                    h.update(f.encode())
                    h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str, obj_dir: str = None, extra=()) -> str:
    obj = os.path.join(obj_dir or OBJ_DIR, os.path.basename(src)[:-4] + '.o')
    cmd = [_hipcc()] + FLAGS + list(extra) + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, 'fingerprint.txt')
    fp = _fingerprint()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read() == fp:
        if verbose:
            print('[lfgc.build] up to date:', LIB_PATH)
        return LIB_PATH
    srcs = sources()
    if verbose:
        print('[lfgc.build] compiling %d HIP sources for %s' % (len(srcs), ARCH))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB_PATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    with open(stamp, 'w') as f:
        f.write(fp)
    if verbose:
        print('[lfgc.build] built', LIB_PATH)
    return LIB_PATH


def build_variant(out_path: str, defines, verbose: bool = True, flags=()) -> str:
    """Diagnostics: build a separate library with extra -D flags (e.g. LFGC_ABLATE=1) next to the product one."""
    obj_dir = out_path + '.objs'
    os.makedirs(obj_dir, exist_ok=True)
    extra = ['-D' + d for d in defines] + list(flags)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, obj_dir, extra), sources()))
    r = subprocess.run([_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', out_path] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('[lfgc.build] built variant', out_path, defines)
    return out_path


if __name__ == '__main__':
    build(force='--force' in sys.argv)
