"""ctypes binding of libsidlsg_hip.so (C ABI declared in include/sidlsg_hip.h).

The argument types are parsed from the header itself, so Python and C cannot drift.
There is NO fallback: if the library is missing or a kernel call fails, this raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'sidlsg_hip.h')
# SIDLSG_LIB: an alternative build of the same C ABI (A/B measurements of kernel variants inside one GPU session)
LIB_PATH = os.environ.get('SIDLSG_LIB') or os.path.join(_HERE, 'libsidlsg_hip.so')

_CT = {'int': ctypes.c_int, 'float': ctypes.c_float, 'long long': ctypes.c_longlong}


def parse_header(path=HEADER):
    """-> {name: [ctypes arg types]} for every `int sidlsg_*(...)` prototype."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(sidlsg_\w+)\s*\(([^)]*)\)\s*;', src):
        args = []
        for a in [x.strip() for x in m.group(2).split(',') if x.strip() and x.strip() != 'void']:
            if '*' in a:
                args.append(ctypes.c_void_p)
            else:
                ty = re.sub(r'\b\w+$', '', a).replace('const', '').strip()
                args.append(_CT[ty])
        protos[m.group(1)] = args
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.isfile(LIB_PATH):
                raise RuntimeError(f'{LIB_PATH} not built: run `python -m sid_lsg_amd.csrc.build` '
                                   '(or __graft_entry__.build()).  There is no non-HIP fallback.')
            # torch first: it brings its own libamdhip64; loading our library before it would pull a second HIP runtime
            # (/opt/rocm) into the process, and kernels registered with one runtime cannot touch the other's memory
            # (seen as hipErrorNoDevice from the first launch when build() and smoke() ran in one process)
            import torch  # noqa: F401
            dll = ctypes.CDLL(LIB_PATH)
            for name, args in self.protos.items():
                fn = getattr(dll, name)
                fn.argtypes = args
                fn.restype = ctypes.c_int
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith('sidlsg_'):
            fn = getattr(self.load(), name)

            def call(*a):
                rc = fn(*a)
                if rc != 0:
                    raise RuntimeError(f'{name} failed with code {rc}')
            call.raw = fn
            self.__dict__[name] = call
            return call
        raise AttributeError(name)


lib = _Lib()
