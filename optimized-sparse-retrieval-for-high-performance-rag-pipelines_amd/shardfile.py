"""Native on-disk format of one doc-range shard of the device index.

The reference only caches its host-side CSR (``.rag_cache/*.npz``, evaluate_rag_pipeline.py:280-312; read/written by
``registry.save_index_npz`` / ``load_index_npz``).  A rank of a 10 M-doc deployment should not redo the CSR -> CSC
transposition, impacts and skip table at start-up, so the device index itself can be stored:

    magic "SRXSHARD" | u32 version | u32 header_len | header JSON (utf-8) | padding to 4096 |
    raw little-endian arrays, each starting at a multiple of 4096: term_ptr i64[V+1], post_doc i32[nnz+PAD],
    post_val f32|f16[nnz+PAD], tile_skip i32[V*(n_tiles+1)], idf f32[V], term_bound f32[V*4] and fine_bound f32[V*14]
    (both optional)

The header carries the dims, dtypes, byte offsets and a CRC-32 of every array.  Loading memory-maps the file and streams
each array to the GPU in bounded chunks (host memory stays small at 10^9 postings); nothing in the file is executed."""
import json
import os
import struct
import zlib
from typing import Dict

import numpy as np

MAGIC = b"SRXSHARD"
VERSION = 1
ALIGN = 4096
_ARRAYS = ("term_ptr", "post_doc", "post_val", "tile_skip", "idf", "term_bound", "fine_bound")
_DTYPES = {"int64": np.int64, "int32": np.int32, "float32": np.float32, "float16": np.float16}


def _crc(a: np.ndarray, chunk: int = 1 << 26) -> int:
    flat = a.reshape(-1).view(np.uint8)
    c = 0
    for i in range(0, flat.size, chunk):
        c = zlib.crc32(flat[i: i + chunk], c)
    return c & 0xFFFFFFFF


def write_shard_file(path: str, arrays: Dict[str, np.ndarray], meta: Dict) -> None:
    """arrays: host arrays named as in _ARRAYS (term_bound optional); meta: n_docs, vocab, nnz, doc_base, tile_log2, ..."""
    entries, off = {}, 0
    for name in _ARRAYS:
        a = arrays.get(name)
        if a is None:
            continue
        a = np.ascontiguousarray(a)
        if a.dtype.name not in _DTYPES:
            raise ValueError(f"{name}: unsupported dtype {a.dtype}")
        entries[name] = {"dtype": a.dtype.name, "count": int(a.size), "offset": off, "crc32": _crc(a)}
        off += (a.nbytes + ALIGN - 1) // ALIGN * ALIGN
    header = json.dumps({"meta": {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in meta.items()},
                         "arrays": entries}, sort_keys=True).encode("utf-8")
    pre = len(MAGIC) + 8 + len(header)
    data_start = (pre + ALIGN - 1) // ALIGN * ALIGN
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, len(header)))
        f.write(header)
        f.write(b"\0" * (data_start - pre))
        for name, e in entries.items():
            a = np.ascontiguousarray(arrays[name])
            assert f.tell() == data_start + e["offset"]
            a.reshape(-1).view(np.uint8).tofile(f)
            pad = (-a.nbytes) % ALIGN
            if pad:
                f.write(b"\0" * pad)
    os.replace(tmp, path)


def read_shard_file(path: str, verify: bool = True):
    """-> (meta dict, {name: read-only np.memmap}).  Raises ValueError on a malformed / corrupted file."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(len(MAGIC) + 8)
        if len(head) < len(MAGIC) + 8 or head[: len(MAGIC)] != MAGIC:
            raise ValueError(f"{path}: not a sparse-rx shard file")
        version, hlen = struct.unpack("<II", head[len(MAGIC):])
        if version != VERSION:
            raise ValueError(f"{path}: shard file version {version}, expected {VERSION}")
        if hlen > size:
            raise ValueError(f"{path}: truncated header")
        hdr = json.loads(f.read(hlen).decode("utf-8"))
    pre = len(MAGIC) + 8 + hlen
    data_start = (pre + ALIGN - 1) // ALIGN * ALIGN
    out = {}
    for name, e in hdr["arrays"].items():
        if name not in _ARRAYS or e["dtype"] not in _DTYPES:
            raise ValueError(f"{path}: unknown array {name!r} / dtype {e['dtype']!r}")
        dt = np.dtype(_DTYPES[e["dtype"]])
        start = data_start + int(e["offset"])
        if int(e["count"]) < 0 or start + int(e["count"]) * dt.itemsize > size:
            raise ValueError(f"{path}: array {name} runs past the end of the file")
        a = np.memmap(path, dtype=dt, mode="r", offset=start, shape=(int(e["count"]),)) if e["count"] else np.zeros(0, dt)
        if verify and _crc(np.asarray(a)) != int(e["crc32"]):
            raise ValueError(f"{path}: checksum mismatch in {name}")
        out[name] = a
    return hdr["meta"], out
