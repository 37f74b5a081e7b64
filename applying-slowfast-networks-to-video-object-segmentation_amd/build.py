"""Compile libsfvos.so (the C-ABI HIP library, include/sfvos.h) for gfx950 with hipcc.

In-tree build: the .so lands next to the sources so it travels with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libsfvos.so')
SOURCES = ['runtime.hip', 'elementwise.hip', 'batchnorm.hip', 'conv3d.hip', 'wgrad.hip', 'lateral.hip', 'lateral_wgrad.hip', 'wgrad_t1.hip', 'maskhead.hip', 'roialign.hip']
HEADERS = ['common.h', 'elt_util.h']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False, diag=False):
    """diag=True: libsfvos_diag.so, compiled with -DSFVOS_DIAG (timing-only switches and tuning overrides read from the
    environment; tools/diag only, never loaded by the package unless SFVOS_LIB names it)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = FLAGS + (['-DSFVOS_DIAG'] if diag else [])
    lib = LIB.replace('libsfvos.so', 'libsfvos_diag.so') if diag else LIB
    deps = [os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'sfvos.h'))

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.diag.o' if diag else '.o'))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + flags + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(lib):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, diag='--diag' in sys.argv))
