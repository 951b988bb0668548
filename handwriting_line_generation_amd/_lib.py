"""ctypes binding of libhwg_hip.so (the C-ABI declared in include/hwg.h).

The argument/return types are derived by parsing the header itself, so the binding cannot drift
from the declared ABI. There is no CPU fallback: if the shared library is missing the import of
this module fails, and calling any kernel without a HIP device raises.
"""
import ctypes
import sys
import os
import re

import torch  # noqa: F401  (must be loaded first: its bundled HIP runtime has to be the one libhwg_hip.so binds to)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_DIR = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("HWG_LIB_OVERRIDE") or os.path.join(_PKG_DIR, "libhwg_hip.so")   # override: tuning builds only
HEADER_PATH = os.path.join(_REPO_DIR, "include", "hwg.h")


class ConvDesc(ctypes.Structure):
    """mirror of `hwg_conv_desc`; `.ptr` is its address (what the entry points take), valid while the object lives"""
    _fields_ = [(n, ctypes.c_int) for n in (
        "N", "H", "W", "C", "K", "R", "S", "stride_h", "stride_w", "pad_h", "pad_w",
        "dil_h", "dil_w", "P", "Q", "transposed")]

    @property
    def ptr(self):
        return ctypes.addressof(self)


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

# Calls go through generated CPython thunks (_hwgcall.so, tools/gen_pycall.py: one METH_FASTCALL function per declared entry point, typed
# against hwg.h at compile time) instead of ctypes' per-argument conversion: ~1 us instead of 5-9 us per call, ~450 calls per training
# step. The thunks are handed the addresses of the symbols of the library loaded above.
_PYCALL_PATH = os.path.join(_PKG_DIR, "_hwgcall.so")
import importlib.util as _ilu  # noqa: E402

_hwgcall = None
try:
    if os.environ.get("HWG_NO_THUNKS"):
        raise ImportError("disabled by HWG_NO_THUNKS")
    if not os.path.exists(_PYCALL_PATH):
        raise ImportError("not built")
    _spec = _ilu.spec_from_file_location("handwriting_line_generation_amd._hwgcall", _PYCALL_PATH)
    _hwgcall = _ilu.module_from_spec(_spec)
    _spec.loader.exec_module(_hwgcall)
    for _name in DECLS:
        _hwgcall.bind(_name, ctypes.cast(getattr(_dll, _name), ctypes.c_void_p).value)
except Exception as _e:  # noqa: BLE001 - e.g. thunks built for another interpreter (Python.h of a different version), or not built at all
    import warnings
    warnings.warn("handwriting_line_generation_amd: the call thunks (_hwgcall.so) are unusable (%r); falling back to ctypes calls, ~5 us slower "
                  "per kernel launch - rebuild with `make -C handwriting_line_generation_amd/csrc PYTHON=%s`" % (_e, sys.executable))
    _hwgcall = None


def _ctypes_entry(fn):
    """ctypes fallback for one entry point: tensors -> their device address, everything else as it is (argtypes do the rest)"""
    # (descriptors travel as addresses - ConvDesc.ptr - like every other pointer: typed struct pointers become void*)
    fn.argtypes = [ctypes.c_void_p if (isinstance(t, type) and issubclass(t, ctypes._Pointer)) else t for t in (fn.argtypes or [])]

    # the same range checks the thunks make in C: ctypes would wrap an out-of-range Python int into the C type silently
    limits = [(-(1 << 31), (1 << 31) - 1) if t is ctypes.c_int else (0, (1 << 64) - 1) if t in (ctypes.c_size_t, ctypes.c_ulonglong)
              else (-(1 << 63), (1 << 63) - 1) if t is ctypes.c_longlong else None for t in fn.argtypes]

    def call(*args):
        conv = []
        for a, lim in zip(args, limits):
            a = a.data_ptr() if hasattr(a, "data_ptr") else a
            if lim is not None and isinstance(a, int) and not (lim[0] <= a <= lim[1]):
                raise OverflowError("%s: integer argument %d does not fit the C type" % (getattr(fn, "__name__", "hwg call"), a))
            conv.append(a)
        return fn(*conv)
    return call


class HwgError(RuntimeError):
    pass


_FN = {name: (getattr(_hwgcall, name) if _hwgcall is not None else _ctypes_entry(getattr(_dll, name))) for name in DECLS}


def last_error():
    e = _FN["hwg_last_error"]()
    return e.decode() if isinstance(e, bytes) else e


def call(name, *args):
    """Call a status-returning entry point; raises HwgError with the library's message on failure. Arguments: None (NULL), ints,
    floats, addresses, or anything with `.data_ptr()` (torch tensors)."""
    rc = _FN[name](*args)
    if rc != 0:
        raise HwgError("%s failed (%d): %s" % (name, rc, last_error()))


def query(name, *args):
    """Call a value-returning entry point (workspace sizes, version)."""
    return _FN[name](*args)


def device_ok():
    return bool(_FN["hwg_device_ok"]())


def abi_version():
    return int(_FN["hwg_abi_version"]())
