"""Build libsidlsg_hip.so (gfx950) in-tree with plain hipcc.  `python -m sid_lsg_amd.csrc.build`."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ['gemm.hip', 'norm.hip', 'attention.hip', 'elementwise.hip', 'optim.hip', 'fp32.hip', 'trace.hip']
LIB = os.path.join(PKG, 'libsidlsg_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-unused-result']
# Per-file extras.  attention.hip: the softmax works on MFMA results with VALU ops; with the default heuristics the
# accumulators live in AGPRs and every tile pays ~110 v_accvgpr_read/write moves in a VALU-bound loop.
# -fno-honor-nans: no canonicalising v_max x,x in front of every fmaxf on an MFMA result (infinities stay honoured).
EXTRA = {'attention.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form', '-fno-honor-nans']}


def digest():
    h = hashlib.md5()
    for f in SOURCES + ['common.h', 'gemm_p8.h', 'bias_act_kernel.h']:
        with open(os.path.join(HERE, f), 'rb') as fh:
            h.update(fh.read())
    h.update((' '.join(FLAGS) + repr(sorted(EXTRA.items()))).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = LIB + '.md5'
    d = digest()
    if not force and os.path.isfile(LIB) and os.path.isfile(stamp) and open(stamp).read().strip() == d:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ['-c', os.path.join(HERE, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(d)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
