"""Native on-disk format of one doc-range shard of the device index.

The reference only caches its host-side CSR (``.rag_cache/*.npz``, evaluate_rag_pipeline.py:280-312; read/written by
``registry.save_index_npz`` / ``load_index_npz``).  A rank of a 10 M-doc deployment should not redo the CSR -> CSC
transposition, impacts and skip table at start-up, so the device index itself can be stored:

    magic "SRXSHARD" | u32 version | u32 header_len | u32 crc32(header JSON) | header JSON (utf-8) | padding to 4096 |
    raw little-endian arrays, each starting at a multiple of 4096: term_ptr i64[V+1] (padded positions), post
    i32[(n_blocks+PAD)*words] (blocks of 4 postings, docs and values side by side: include/sparse_rx.h), tile_skip
    i32[V*(n_tiles+1)], idf f32[V], term_bound f32[V*4] and fine_bound f32[V*14] (both optional)

The header carries the dims, dtypes, byte offsets and a CRC-32 of every array, and is itself covered by a CRC-32.
Reading validates the header against itself BEFORE anything reaches the GPU -- the kernels index the arrays with these
dims and receive raw pointers without sizes, so a damaged or mismatched header must become a ``ValueError`` here, never
an out-of-bounds device read: every array length against the dims (term_ptr V+1, idf V, (n_blocks + pad) blocks of 8 or
6 words, skip table V x (n_tiles+1), bounds V x 4 / V x 14), term_ptr a non-decreasing table of multiples of 4 from 0 to
4 n_blocks, the skip rows (non-decreasing, block-aligned at unit boundaries, ending at the term's padded posting count),
offsets non-negative, aligned and non-overlapping.  With ``verify`` (default) also the array checksums and the O(nnz) scan
that every doc id lies inside [-2017, n_docs) (negative = sentinel); ``verify=False`` TRUSTS the posting bytes themselves
(a doc id outside the shard would index the kernels' LDS tables out of range).  Loading memory-maps the file and streams each array to the GPU in
bounded chunks (host memory stays small at 10^9 postings); nothing in the file is executed."""
import json
import os
import struct
import zlib
from typing import Dict

import numpy as np

MAGIC = b"SRXSHARD"
VERSION = 3
ALIGN = 4096
_ARRAYS = ("term_ptr", "post", "tile_skip", "idf", "term_bound", "fine_bound")
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
    pre = len(MAGIC) + 12 + len(header)
    data_start = (pre + ALIGN - 1) // ALIGN * ALIGN
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<III", VERSION, len(header), zlib.crc32(header) & 0xFFFFFFFF))
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


FINE_KS_LEN = 14  # len(DeviceIndex.FINE_KS)
_EXPECT_DTYPE = {"term_ptr": "int64", "post": "int32", "tile_skip": "int32", "idf": "float32", "term_bound": "float32",
                 "fine_bound": "float32"}


def _int_meta(path, meta, key, lo, hi):
    v = meta.get(key)
    if isinstance(v, bool) or not isinstance(v, int) or not (lo <= v <= hi):
        raise ValueError(f"{path}: header field {key!r} = {v!r} is missing or out of range [{lo}, {hi}]")
    return v


def read_shard_file(path: str, verify: bool = True):
    """-> (meta dict, {name: read-only np.memmap}).  Raises ValueError on a malformed, corrupted or self-inconsistent
    file (see the module docstring for what is checked)."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(len(MAGIC) + 12)
        if len(head) < len(MAGIC) + 12 or head[: len(MAGIC)] != MAGIC:
            raise ValueError(f"{path}: not a sparse-rx shard file")
        version, hlen, hcrc = struct.unpack("<III", head[len(MAGIC):])
        if version != VERSION:
            raise ValueError(f"{path}: shard file version {version}, expected {VERSION}")
        if hlen > size:
            raise ValueError(f"{path}: truncated header")
        raw = f.read(hlen)
        if len(raw) != hlen or (zlib.crc32(raw) & 0xFFFFFFFF) != hcrc:
            raise ValueError(f"{path}: header checksum mismatch")
        try:
            hdr = json.loads(raw.decode("utf-8"))
            meta, entries = hdr["meta"], hdr["arrays"]
            assert isinstance(meta, dict) and isinstance(entries, dict)
        except Exception as e:
            raise ValueError(f"{path}: malformed header ({e})") from None
    pre = len(MAGIC) + 12 + hlen
    data_start = (pre + ALIGN - 1) // ALIGN * ALIGN
    # ---- dims ----
    n_docs = _int_meta(path, meta, "n_docs", 1, 0x7FFFFFFE)
    vocab = _int_meta(path, meta, "vocab", 1, 1 << 40)
    nnz = _int_meta(path, meta, "nnz", 0, 1 << 40)
    n_blocks = _int_meta(path, meta, "n_blocks", 0, 1 << 40)
    tile_log2 = _int_meta(path, meta, "tile_log2", 6, 14)
    unit_tiles = _int_meta(path, meta, "unit_tiles", 1, 64)
    val_type = _int_meta(path, meta, "val_type", 0, 1)
    pad = _int_meta(path, meta, "block_pad", 1, 1 << 20)
    _int_meta(path, meta, "doc_base", 0, 1 << 62)
    if n_blocks * 4 < nnz:
        raise ValueError(f"{path}: n_blocks = {n_blocks} cannot hold nnz = {nnz} postings")
    words = 8 if val_type == 0 else 6
    n_tiles = (n_docs + (1 << tile_log2) - 1) >> tile_log2
    want = {"term_ptr": vocab + 1, "post": (n_blocks + pad) * words, "tile_skip": vocab * (n_tiles + 1),
            "idf": vocab, "term_bound": vocab * 4, "fine_bound": vocab * FINE_KS_LEN}
    for name in ("term_ptr", "post", "tile_skip", "idf"):
        if name not in entries:
            raise ValueError(f"{path}: array {name} is missing")
    out, spans = {}, []
    for name, e in entries.items():
        if name not in _ARRAYS or not isinstance(e, dict) or e.get("dtype") not in _DTYPES:
            raise ValueError(f"{path}: unknown array {name!r} / dtype {e.get('dtype') if isinstance(e, dict) else e!r}")
        if e["dtype"] != _EXPECT_DTYPE[name]:
            raise ValueError(f"{path}: array {name} has dtype {e['dtype']}")
        dt = np.dtype(_DTYPES[e["dtype"]])
        count, off = e.get("count"), e.get("offset")
        if not isinstance(count, int) or not isinstance(off, int) or isinstance(count, bool) or isinstance(off, bool):
            raise ValueError(f"{path}: array {name}: count / offset are not integers")
        if count != want[name]:
            raise ValueError(f"{path}: array {name} holds {count} elements, the header dims require {want[name]}")
        if off < 0 or off % ALIGN:
            raise ValueError(f"{path}: array {name}: offset {off} is negative or not {ALIGN}-aligned")
        start = data_start + off
        if start + count * dt.itemsize > size:
            raise ValueError(f"{path}: array {name} runs past the end of the file")
        spans.append((start, start + count * dt.itemsize, name))
        out[name] = np.memmap(path, dtype=dt, mode="r", offset=start, shape=(count,)) if count else np.zeros(0, dt)
    spans.sort()
    for (a0, a1, an), (b0, b1, bn) in zip(spans, spans[1:]):
        if b0 < a1:
            raise ValueError(f"{path}: arrays {an} and {bn} overlap")
    # ---- the offsets every kernel trusts ----
    tp = np.asarray(out["term_ptr"])
    if int(tp[0]) != 0 or int(tp[-1]) != 4 * n_blocks or np.any(np.diff(tp) < 0) or np.any(tp & 3):
        raise ValueError(f"{path}: term_ptr is not a non-decreasing table of block-aligned positions from 0 to 4 * n_blocks = {4 * n_blocks}")
    # tile_skip rows index `post` from inside the kernels: their structure is checked on every load, O(vocab * n_tiles) like
    # the term_ptr check above (only the O(nnz) doc-id scan and the checksums are left to `verify`)
    ts = np.asarray(out["tile_skip"]).reshape(vocab, n_tiles + 1)
    if (np.any(ts[:, 0] != 0) or np.any(np.diff(ts, axis=1) < 0) or np.any(ts[:, -1].astype(np.int64) != np.diff(tp))
            or np.any(ts[:, ::unit_tiles] & 3)):
        raise ValueError(f"{path}: tile_skip rows are not non-decreasing, block-aligned at unit boundaries and ending at the term's padded posting count")
    if verify:
        for name, e in entries.items():
            if _crc(np.asarray(out[name])) != int(e.get("crc32", -1)):
                raise ValueError(f"{path}: checksum mismatch in {name}")
        chunk = 1 << 22  # blocks per pass
        blocks = out["post"].reshape(n_blocks + pad, words)
        for i in range(0, n_blocks + pad, chunk):
            docs = np.asarray(blocks[i: i + chunk, :4])
            if docs.size and (int(docs.min()) < -1 - 32 * 63 or int(docs.max()) >= n_docs):
                raise ValueError(f"{path}: post holds doc ids outside [-2017, {n_docs})")
    return meta, out
