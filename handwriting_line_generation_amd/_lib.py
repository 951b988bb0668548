"""ctypes binding of libhwg_hip.so (the C-ABI declared in include/hwg.h).

The argument/return types are derived by parsing the header itself, so the binding cannot drift
from the declared ABI. There is no CPU fallback: if the shared library is missing the import of
this module fails, and calling any kernel without a HIP device raises.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must be loaded first: its bundled HIP runtime has to be the one libhwg_hip.so binds to)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_DIR = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("HWG_LIB_OVERRIDE") or os.path.join(_PKG_DIR, "libhwg_hip.so")   # override: tuning builds only
HEADER_PATH = os.path.join(_REPO_DIR, "include", "hwg.h")


class ConvDesc(ctypes.Structure):
    """mirror of `hwg_conv_desc`"""
    _fields_ = [(n, ctypes.c_int) for n in (
        "N", "H", "W", "C", "K", "R", "S", "stride_h", "stride_w", "pad_h", "pad_w",
        "dil_h", "dil_w", "P", "Q", "transposed")]


_SCALARS = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
}


def _ctype_of(decl):
    decl = decl.strip()
    if decl == "void":
        return None
    if "*" in decl:
        if "hwg_conv_desc" in decl:
            return ctypes.POINTER(ConvDesc)
        if decl.replace("const", "").replace(" ", "") == "char*":
            return ctypes.c_char_p
        return ctypes.c_void_p
    # strip the parameter name
    words = decl.replace("const ", "").split()
    for n in (3, 2, 1):
        t = " ".join(words[:n])
        if t in _SCALARS and len(words) >= n:
            return _SCALARS[t]
    raise ValueError("hwg.h: cannot map C type of %r" % decl)


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in hwg.h"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"typedef\s+(struct|enum)[^{]*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"enum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(hwg_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _SCALARS[ret.replace("const ", "").strip()]
        argtypes = []
        if args and args != "void":
            argtypes = [_ctype_of(a) for a in args.split(",")]
        decls[name] = (restype, argtypes)
    return decls


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libhwg_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C handwriting_line_generation_amd/csrc`). There is no CPU fallback." % LIB_PATH)

_dll = ctypes.CDLL(LIB_PATH)
DECLS = parse_header()
for _name, (_res, _args) in DECLS.items():
    _fn = getattr(_dll, _name)  # AttributeError here == header declares a symbol the library lacks
    _fn.restype = _res
    _fn.argtypes = _args


class HwgError(RuntimeError):
    pass


def last_error():
    return _dll.hwg_last_error().decode("utf-8", "replace")


def _conv(a):
    # torch tensors -> raw device pointers; keep everything else
    dp = getattr(a, "data_ptr", None)
    if dp is not None:
        return dp()
    return a


_FN = {name: getattr(_dll, name) for name in DECLS}


def call(name, *args):
    """Call a status-returning entry point; raises HwgError with the library's message on failure."""
    rc = _FN[name](*[a.data_ptr() if hasattr(a, "data_ptr") else a for a in args])
    if rc != 0:
        raise HwgError("%s failed (%d): %s" % (name, rc, last_error()))


def query(name, *args):
    """Call a value-returning entry point (workspace sizes, version)."""
    return getattr(_dll, name)(*[_conv(a) for a in args])


def device_ok():
    return bool(_dll.hwg_device_ok())


def abi_version():
    return int(_dll.hwg_abi_version())
