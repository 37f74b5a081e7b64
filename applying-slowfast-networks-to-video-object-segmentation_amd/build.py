"""Compile libsfvos.so (the C-ABI HIP library, include/sfvos.h) for gfx950 with hipcc.

In-tree build: the .so lands next to the sources so it travels with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libsfvos.so')
SOURCES = ['runtime.hip', 'elementwise.hip', 'batchnorm.hip', 'conv3d.hip', 'wgrad.hip', 'lateral.hip', 'maskhead.hip']
HEADERS = ['common.h', 'elt_util.h']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    deps = [os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'sfvos.h'))

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=7) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
