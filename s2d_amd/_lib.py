"""ctypes binding of libs2d_hip.so.  include/s2d_hip.h is the single source of truth: prototypes are
parsed from it, so the header, the library and this binding cannot drift apart.

The product path has no CPU fallback: if the library is missing or a symbol is absent this module
raises, loudly."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, "..", "include", "s2d_hip.h")
LIBPATH = os.environ.get("S2D_HIP_LIB") or os.path.join(_HERE, "csrc", "libs2d_hip.so")   # override: kernel experiments only

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
           "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "hipStream_t": ctypes.c_void_p,
           "unsigned": ctypes.c_uint, "uint8_t": ctypes.c_uint8, "uint32_t": ctypes.c_uint32}


def parse_header(path=HEADER):
    """-> {name: [ctypes arg types]} for every `int s2d_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(s2d_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    types.append(_CTYPES[base])
        protos[name] = (types, ret)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIBPATH):
            raise RuntimeError(f"{LIBPATH} not built: run `python -m s2d_amd.build` (there is no CPU fallback)")
        import torch  # noqa: F401  (loads the process's libamdhip64 first so both share one HIP runtime)
        self._dll = ctypes.CDLL(LIBPATH)
        self.protos = parse_header()
        for name, (types, ret) in self.protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the library lacks a declared symbol
            fn.argtypes = types
            fn.restype = ctypes.c_long if ret == "long" else ctypes.c_int
            setattr(self, "_raw_" + name, fn)

    def call(self, name, *args):
        fn = getattr(self, "_raw_" + name)
        conv = []
        for a in args:
            if a is None:
                conv.append(None)
            elif hasattr(a, "data_ptr"):
                conv.append(a.data_ptr())
            elif hasattr(a, "ctypes"):  # host numpy array
                conv.append(a.ctypes.data)
            else:
                conv.append(a)
        rc = fn(*conv)
        if self.protos[name][1] == "long" or name == "s2d_abi_version":
            return rc
        if rc != 0:
            raise RuntimeError(f"{name} failed with code {rc}")


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
