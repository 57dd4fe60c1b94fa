"""Loader of the native C-ABI library ``libgsplat_hip.so`` (ctypes).

Counterpart of the reference's ``gsplat/cuda/_backend.py:79-141`` (which finds or
JIT-builds a pybind11/torch extension).  Here the native side is a plain shared
library with a flat C ABI (``include/gsplat_hip.h``); this module

* loads it from the package tree (``gscodec_studio_amd/csrc/libgsplat_hip.so``),
* derives every function's ctypes prototype by parsing the header, so the header is
  the single source of truth for the ABI,
* exposes ``call(name, *args)`` which passes tensors as raw device pointers, appends
  nothing implicitly, and turns a non-zero status into ``RuntimeError`` with the
  library's ``gs_last_error()`` message.

There is deliberately NO fallback: if the library is missing or a symbol is absent the
import of the op fails loudly (a CPU/eager fallback would silently void every parity
and performance claim made for the HIP path).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Optional, Tuple

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_DIR = os.path.dirname(_PKG_DIR)



def _find_header() -> str:
    """The ABI header: $GSPLAT_HIP_HEADER, the copy the build puts inside the package (``<pkg>/include``, what an
    installed / vendored package carries), or the repository's ``include/`` (source checkout)."""
    cands = [os.environ.get("GSPLAT_HIP_HEADER"), os.path.join(_PKG_DIR, "include", "gsplat_hip.h"),
             os.path.join(_REPO_DIR, "include", "gsplat_hip.h")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    return cands[-1]


HEADER_PATH = _find_header()
LIB_PATH = os.environ.get("GSPLAT_HIP_LIB", os.path.join(_PKG_DIR, "csrc", "libgsplat_hip.so"))

_SCALARS = {
    "int32_t": ctypes.c_int32,
    "uint32_t": ctypes.c_uint32,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "gs_stream_t": ctypes.c_void_p,
}


def _ctype_of(decl: str):
    """Map one C parameter / return declaration to a ctypes type."""
    decl = decl.strip()
    if "*" in decl:
        if decl.replace("const", "").strip().startswith("char"):
            return ctypes.c_char_p
        return ctypes.c_void_p
    base = decl.replace("const", "").split()[0]
    return _SCALARS[base]


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object], List[str]]]:
    """Return {function name: (restype, [argtypes], [arg names])} for every prototype."""
    with open(path, "r") as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)  # strip comments
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*", " ", src, flags=re.M)  # strip preprocessor lines
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(gs_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = ret.replace('extern "C"', "").strip()
        args = " ".join(args.split())
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                argtypes.append(_ctype_of(a))
                argnames.append(re.findall(r"(\w+)\s*$", a)[0])
        protos[name] = (_ctype_of(ret), argtypes, argnames)
    return protos


def header_abi_version(path: str = HEADER_PATH) -> int:
    """GS_ABI_VERSION as the header declares it."""
    with open(path, "r") as f:
        m = re.search(r"^\s*#\s*define\s+GS_ABI_VERSION\s+(\d+)", f.read(), flags=re.M)
    if m is None:
        raise ImportError(f"gscodec_studio_amd: {path} does not define GS_ABI_VERSION")
    return int(m.group(1))


def header_hash(path: str = HEADER_PATH) -> int:
    """First 8 bytes (big-endian) of the SHA-256 of the header file: what the Makefile compiles into gs_header_hash()."""
    import hashlib

    with open(path, "rb") as f:
        return int(hashlib.sha256(f.read()).hexdigest()[:16], 16)


_LIB: Optional[ctypes.CDLL] = None
_PROTOS: Optional[Dict] = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the native library with prototypes attached."""
    global _LIB, _PROTOS
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"gscodec_studio_amd: native library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C gscodec_studio_amd/csrc`. There is no CPU fallback."
        )
    if not os.path.exists(HEADER_PATH):
        raise ImportError(
            f"gscodec_studio_amd: ABI header gsplat_hip.h not found (looked at $GSPLAT_HIP_HEADER, "
            f"{os.path.join(_PKG_DIR, 'include')}, {os.path.join(_REPO_DIR, 'include')}); the ctypes prototypes are "
            "derived from it. `make -C gscodec_studio_amd/csrc` copies it into the package.")
    _PROTOS = parse_header()
    L = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes, _) in _PROTOS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:  # declared in the header but not exported
            raise ImportError(f"gscodec_studio_amd: {LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    # the header the prototypes were parsed from and the library must be the SAME revision of the ABI: the version the
    # header declares, and the hash of the header file the library was compiled against (a changed argument list behind
    # an unchanged version number would otherwise be called through shifted arguments)
    want = header_abi_version()
    if L.gs_version() != want:
        raise ImportError(f"gscodec_studio_amd: ABI version mismatch: {LIB_PATH} reports {L.gs_version()}, "
                          f"{HEADER_PATH} declares {want}; rebuild with `make -C gscodec_studio_amd/csrc`")
    if L.gs_header_hash() != header_hash():
        raise ImportError(f"gscodec_studio_amd: {LIB_PATH} was compiled against a different gsplat_hip.h than {HEADER_PATH} "
                          f"(hash {L.gs_header_hash():016x} != {header_hash():016x}); rebuild with `make -C gscodec_studio_amd/csrc`")
    _LIB = L
    return L


def prototypes() -> Dict:
    lib()
    return _PROTOS


def ptr(t) -> Optional[int]:
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def call(name: str, *args) -> None:
    """Call an ``int32_t``-status entry point; raise RuntimeError on failure."""
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        msg = L.gs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (status {rc}): {msg}")


def query(name: str, *args):
    """Call a value-returning helper (e.g. gs_sort_temp_bytes)."""
    return getattr(lib(), name)(*args)
