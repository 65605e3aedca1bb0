"""`get_plugin(module_name, sources, **build_kwargs)` -- the reference's plugin loader seam
(torch_utils/custom_ops.py:46-124, boundary B3) re-designed for ROCm: sources are .hip files with
`extern "C"` entry points, compiled in-tree by hipcc for gfx950 into `<module_name>.so` and returned
as a ctypes library handle.

Kept from the reference contract: process-global cache keyed by module name; rebuild only when the
md5 digest of the sources changes; a cross-process lock so N ranks build once (reference: FileBaton,
custom_ops.py:95-105); `verbosity` in {'none','brief','full'}; failures propagate as exceptions.
Not kept: pybind/torch::Tensor signatures -- the boundary here is a C ABI (plain pointers + sizes).
"""
import ctypes
import fcntl
import hashlib
import os
import subprocess

verbosity = 'brief'
_cached_plugins = dict()
_DEFAULT_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-munsafe-fp-atomics']


def get_plugin(module_name, sources, build_directory=None, extra_cflags=(), headers=(), **build_kwargs):
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if verbosity != 'none':
        print(f'Setting up HIP plugin "{module_name}"... ', end='' if verbosity == 'brief' else '\n', flush=True)
    try:
        sources = [os.path.abspath(s) for s in sources]
        build_directory = build_directory or os.path.dirname(sources[0])
        h = hashlib.md5()
        for f in list(sources) + [os.path.abspath(x) for x in headers]:
            with open(f, 'rb') as fh:
                h.update(fh.read())
        h.update(' '.join(list(_DEFAULT_FLAGS) + list(extra_cflags)).encode())
        so = os.path.join(build_directory, f'{module_name}.so')
        stamp = so + '.md5'
        with open(os.path.join(build_directory, f'.{module_name}.lock'), 'w') as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)   # one rank builds, the others wait here
            try:
                fresh = os.path.isfile(so) and os.path.isfile(stamp) and open(stamp).read().strip() == h.hexdigest()
                if not fresh:
                    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + _DEFAULT_FLAGS + list(extra_cflags) + sources + ['-o', so]
                    if verbosity == 'full':
                        print(' '.join(cmd), flush=True)
                    subprocess.check_call(cmd)
                    with open(stamp, 'w') as f:
                        f.write(h.hexdigest())
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
        module = ctypes.CDLL(so)
    except Exception:
        if verbosity == 'brief':
            print('Failed!')
        raise
    if verbosity != 'none':
        print('Done.' if verbosity == 'brief' else f'Done setting up HIP plugin "{module_name}".')
    _cached_plugins[module_name] = module
    return module
