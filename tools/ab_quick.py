"""Quick A/B libraries: recompile only the named translation units with extra -D flags and link them with the product
build's other objects (latent_feature_grid_compression_amd/csrc/build/*.o must be up to date: run build first).

    python tools/ab_quick.py lfgc_fwd16_ch32 name1=DEF1,DEF2 name2= ...      -> tools/ab_libs/liblfgc_ab_<name>.so
Run them with  LFGC_AB_DIR=$PWD/tools/ab_libs python tools/ab_forward.py run  on the GPU box (built libraries travel).
"""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latent_feature_grid_compression_amd import build as B      # noqa: E402

OUT = os.path.join(ROOT, 'tools', 'ab_libs')


def one(spec, tus):
    name, _, defs = spec.partition('=')
    extra = ['-D' + d for d in defs.split(',') if d]
    odir = os.path.join(OUT, name + '.objs')
    os.makedirs(odir, exist_ok=True)
    objs = []
    for src in B.sources():
        base = os.path.basename(src)[:-4]
        if base in tus:
            objs.append(B._compile(src, odir, extra))
        else:
            objs.append(os.path.join(B.OBJ_DIR, base + '.o'))
    lib = os.path.join(OUT, 'liblfgc_ab_%s.so' % name)
    r = subprocess.run([B._hipcc(), '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    subprocess.run(['rm', '-rf', odir])
    print('built', lib, extra, flush=True)


if __name__ == '__main__':
    tus = set(sys.argv[1].split(','))
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(lambda s: one(s, tus), sys.argv[2:]))
