"""ctypes binding of libsparse_rx.so (C ABI declared in include/sparse_rx.h).

The library is the product: there is no CPU fallback.  If it is missing or cannot be loaded every entry point
raises ``SparseRxUnavailable`` -- loudly, never a silent eager path.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from typing import Optional

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.path.join(_PKG_DIR, "libsparse_rx.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")
SOURCES = ["wave_kernel.hip", "sparse_rx.hip", "dense.hip"]   # one translation unit each, compiled in parallel
SRC_PATH = os.path.join(CSRC_DIR, "wave_kernel.hip")            # the dominant kernel's source (bench.py hashes it)
INCLUDE_DIR = os.path.join(_ROOT, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-std=c++17"]

SRX_VAL_F32, SRX_VAL_F16 = 0, 1


class SparseRxUnavailable(RuntimeError):
    """libsparse_rx.so is not built / not loadable (the HIP engine is mandatory)."""


class SparseRxError(RuntimeError):
    """A C-ABI call returned a negative status."""


class IndexDesc(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("val_type", ctypes.c_int32), ("n_docs", ctypes.c_int64),
                ("vocab", ctypes.c_int64), ("nnz", ctypes.c_int64), ("n_blocks", ctypes.c_int64), ("doc_base", ctypes.c_int64),
                ("tile_log2", ctypes.c_int32), ("n_tiles", ctypes.c_int32), ("unit_tiles", ctypes.c_int32),
                ("reserved0", ctypes.c_int32), ("term_ptr", ctypes.c_void_p), ("post", ctypes.c_void_p),
                ("tile_skip", ctypes.c_void_p), ("idf", ctypes.c_void_p), ("term_bound", ctypes.c_void_p),
                ("post16", ctypes.c_void_p)]


class SearchOpts(ctypes.Structure):
    _fields_ = [("supertile_log2", ctypes.c_int32), ("target_blocks", ctypes.c_int32), ("profile", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("unit_tiles", ctypes.c_int32)]


# every symbol include/sparse_rx.h declares: name -> (restype, argtypes)
_VP, _I32, _I64, _DBL = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
SYMBOLS = {
    "srx_version": (ctypes.c_int, []),
    "srx_last_error": (ctypes.c_char_p, []),
    "srx_limits": (ctypes.c_int, [ctypes.POINTER(_I32)]),
    "srx_device_count": (ctypes.c_int, []),
    "srx_index_create": (ctypes.c_int, [ctypes.POINTER(IndexDesc), ctypes.POINTER(_VP)]),
    "srx_index_destroy": (None, [_VP]),
    "srx_index_set_opts": (ctypes.c_int, [_VP, ctypes.POINTER(SearchOpts)]),
    "srx_search_workspace_bytes": (_I64, [_VP, _I32, _I32]),
    "srx_search": (ctypes.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_search_after": (ctypes.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_search_after_packed": (ctypes.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_merge_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "srx_merge_topk": (ctypes.c_int, [_I32, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_merge_topk_packed": (ctypes.c_int, [_I32, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_search_packed": (ctypes.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, _I64, _VP]),
    "srx_merge_topk_packed_out": (ctypes.c_int, [_I32, _VP, _I32, _I32, _I32, _VP, _VP, _I64, _VP]),
    "srx_dense_workspace_bytes": (_I64, [_I32, _I64, _I32]),
    "srx_dense_search_i8": (ctypes.c_int, [_I32, _VP, _VP, _I64, _I32, _VP, _VP, _I32, _I32, _I64, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_dense_packed_bytes": (_I64, [_I64, _I32]),
    "srx_dense_pack_i8": (ctypes.c_int, [_I32, _VP, _I64, _I32, _VP, _VP]),
    "srx_dense_search_i8_packed": (ctypes.c_int, [_I32, _VP, _VP, _I64, _I32, _VP, _VP, _I32, _I32, _I64, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_dense_f32_workspace_bytes": (_I64, [_I32, _I64, _I32]),
    "srx_dense_search_f32": (ctypes.c_int, [_I32, _VP, _I64, _I32, _VP, _I32, _I32, _I64, _VP, _VP, _VP, _VP, _I64, _VP, ctypes.c_float]),
    "srx_dense_search_u8": (ctypes.c_int, [_I32, _VP, _VP, _I64, _I32, _VP, _I32, _I32, _I64, _VP, _VP, _VP, _VP, _I64, _VP]),
    "srx_build_impacts": (ctypes.c_int, [_I32, _VP, _VP, _VP, _I64, _DBL, _DBL, _DBL, _VP, _VP]),
    "srx_build_tile_skip": (ctypes.c_int, [_I32, _VP, _VP, _I64, _I32, _I32, _VP, _VP]),
    "srx_memcpy_async": (ctypes.c_int, [_VP, _VP, _I64, _VP]),
    "srx_auto_unit_tiles": (_I32, [_I64, _I64, _I64, _I32]),
    "srx_build_blocks": (ctypes.c_int, [_I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _I32, _I32, _I32, _VP, _VP, _VP, _I64, _VP]),
    "srx_build_compact": (ctypes.c_int, [_I32, _I32, _VP, _I64, _I32, _I32, _VP, _VP]),
    "srx_build_term_bounds": (ctypes.c_int, [_I32, _I32, _VP, _VP, _I64, _VP, _I32, _VP, _VP, _VP]),
    "srx_build_sum_duplicates": (ctypes.c_int, [_I32, _VP, _I64, _VP, _VP, _VP]),
    "srx_profile_read": (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_float)]),
}

_lib: Optional[ctypes.CDLL] = None


def kernel_sources_sha256() -> str:
    """sha256 over the sparse path's kernel sources (wave_kernel.hip, sparse_rx.hip, srx_common.h): identifies the build a
    profile was taken from."""
    import hashlib
    h = hashlib.sha256()
    for f in ["wave_kernel.hip", "sparse_rx.hip", "srx_common.h"]:
        with open(os.path.join(CSRC_DIR, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _deps():
    return [os.path.join(CSRC_DIR, f) for f in SOURCES] + [os.path.join(CSRC_DIR, "srx_common.h"),
                                                            os.path.join(INCLUDE_DIR, "sparse_rx.h")]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with hipcc (one object per source, in parallel) and link them into
    <package>/libsparse_rx.so (in-tree)."""
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(p) for p in _deps()):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise SparseRxUnavailable("hipcc not found: cannot build libsparse_rx.so")
    objdir = os.path.join(CSRC_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(os.path.join(CSRC_DIR, "srx_common.h")), os.path.getmtime(os.path.join(INCLUDE_DIR, "sparse_rx.h")))
    procs, objs = [], []
    for f in SOURCES:
        src, obj = os.path.join(CSRC_DIR, f), os.path.join(objdir, f.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
            continue
        cmd = [hipcc, *HIPCC_FLAGS, f"-I{INCLUDE_DIR}", f"-I{CSRC_DIR}", "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        procs.append((f, subprocess.Popen(cmd)))
    for f, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, f"hipcc -c {f}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", LIB_PATH, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library (symbols typed).  Raises SparseRxUnavailable if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SparseRxUnavailable(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SparseRxUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            f = getattr(L, name)
        except AttributeError as e:
            raise SparseRxUnavailable(f"{LIB_PATH} does not export {name}") from e
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = lib().srx_last_error()
        text = msg.decode("utf-8", "replace") if msg else ""
        if rc == -1:
            raise ValueError(f"{what}: {text}")
        raise SparseRxError(f"{what} failed ({rc}): {text}")
    return rc


def limits():
    out = (ctypes.c_int32 * 4)()
    check(lib().srx_limits(out), "srx_limits")
    return {"max_k": out[0], "max_tile_log2": out[1], "hash_cap": out[2], "threads": out[3]}
