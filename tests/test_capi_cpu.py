"""CPU-side checks: the C-ABI library loads and exports every symbol include/sparse_rx.h declares (no compute
calls without a GPU), argument validation that does not touch the device, and the host logic."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import sparse_rx
from sparse_rx import _capi


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sparse_rx.h")).read()
    declared = set(re.findall(r"\b(srx_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libsparse_rx.so does not export {name}"
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    assert L.srx_version() == 301


def test_limits_and_error_strings():
    lim = _capi.limits()
    assert lim == {"max_k": 1024, "max_tile_log2": 14, "hash_cap": 4096, "threads": 256}
    L = _capi.lib()
    assert L.srx_index_create(None, None) == -1
    assert b"null" in L.srx_last_error()
    assert L.srx_search(None, None, None, None, 1, 10, None, None, None, None, 0, None) == -1
    with pytest.raises(ValueError):
        _capi.check(L.srx_merge_workspace_bytes(1, 0, 10), "srx_merge_workspace_bytes")
    # argument checks of the round-2 entry points (all return before anything touches a device)
    assert L.srx_build_compact(0, 0, None, 10, 14, 3, None, None) == -1 and b"srx_build_compact" in L.srx_last_error()
    assert L.srx_build_compact(0, 0, 1 << 20, 10, 14, 4, 1 << 21, None) == -1 and b"49152" in L.srx_last_error()   # 4 x 16384 docs per unit
    assert L.srx_build_compact(0, 7, 1 << 20, 10, 14, 3, 1 << 21, None) == -1 and b"val_type" in L.srx_last_error()
    assert L.srx_build_term_bounds(0, 0, None, 1 << 20, 10, 1 << 21, 4, 1 << 22, 1 << 23, None) == -1 and b"srx_build_term_bounds" in L.srx_last_error()
    assert L.srx_build_term_bounds(0, 0, 1 << 19, 1 << 20, 10, 1 << 21, 65, 1 << 22, 1 << 23, None) == -1 and b"nk" in L.srx_last_error()
    assert L.srx_build_term_bounds(0, 7, 1 << 19, 1 << 20, 10, 1 << 21, 4, 1 << 22, 1 << 23, None) == -1 and b"val_type" in L.srx_last_error()
    assert L.srx_build_sum_duplicates(0, None, 5, 1 << 20, 1 << 21, None) == -1 and b"srx_build_sum_duplicates" in L.srx_last_error()
    assert L.srx_build_sum_duplicates(0, None, 0, None, None, None) == 0
    assert L.srx_dense_packed_bytes(1000, 768) == 1024 * 768 and L.srx_dense_packed_bytes(33, 32) == 64 * 32
    assert L.srx_dense_packed_bytes(10, 48) == -1 and b"srx_dense_packed_bytes" in L.srx_last_error()
    assert L.srx_dense_pack_i8(0, None, 10, 64, 1 << 20, None) == -1 and b"srx_dense_pack_i8" in L.srx_last_error()
    assert L.srx_dense_pack_i8(0, (1 << 20) + 8, 10, 64, 1 << 21, None) == -1 and b"aligned" in L.srx_last_error()
    assert L.srx_dense_search_i8_packed(0, 1 << 20, 1 << 21, 10, 48, 1 << 22, 1 << 23, 1, 5, 0, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 20, None) == -1
    assert L.srx_dense_search_u8(0, 1 << 20, None, 10, 64, 1 << 21, 1, 5, 0, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 20, None) == -1
    assert L.srx_dense_search_u8(0, 1 << 20, 1 << 26, 10, 48, 1 << 21, 1, 5, 0, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 20, None) == -1
    assert b"multiple of 64" in L.srx_last_error()
    assert L.srx_merge_workspace_bytes(10, 8, 100) == 0        # 800 candidates fit one workgroup
    assert L.srx_merge_workspace_bytes(10, 8, 1000) > 0        # 8000 do not: tree merge needs scratch


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sparse_rx.SparseRxUnavailable):
        sparse_rx.DeviceIndex.from_csr(np.array([0, 1]), np.array([0], np.int32), np.array([1.0], np.float32),
                                       np.array([1.0], np.float32), doc_lengths=np.array([1.0], np.float32))
    svc = sparse_rx.RetrievalService()
    with pytest.raises(ValueError, match="BM25 index not built"):
        svc.search_bm25({"a": "b"})
    with pytest.raises(ValueError, match="Empty corpus"):
        svc.build_bm25_index({})
    with pytest.raises(sparse_rx.SparseRxUnavailable):
        svc.build_bm25_index({"d": {"text": "hello world"}})


def test_product_never_imports_oracle():
    pkg = os.path.dirname(os.path.abspath(sparse_rx._capi.__file__))
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, fn), encoding="utf-8").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f"{fn} imports oracle"
                assert "liboracle" not in src, f"{fn} references liboracle"


def test_host_index_matches_reference_state(golden_dir):
    """build_host_index == the state RetrievalService.build_bm25_index left on the reference (bit-equal)."""
    z = np.load(os.path.join(golden_dir, "text_small.npz"))
    j = json.load(open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8"))
    h = sparse_rx.build_host_index(j["corpus"])
    assert h.doc_ids == list(z["doc_ids"])
    assert [t for t, _ in sorted(h.vocabulary.items(), key=lambda kv: kv[1])] == list(z["vocabulary"])
    assert np.array_equal(h.indptr, z["tf_indptr"]) and np.array_equal(h.indices, z["tf_indices"])
    assert np.array_equal(h.data, z["tf_data"]) and h.data.dtype == np.float32
    assert np.array_equal(h.doc_lengths, z["doc_lengths"])
    assert np.array_equal(h.idf.view(np.uint32), z["idf"].view(np.uint32))
    assert h.avgdl == float(z["avgdl"])


def test_encode_queries_matches_reference_query_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "text_small.npz"))
    j = json.load(open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8"))
    h = sparse_rx.build_host_index(j["corpus"])
    qids = list(z["score_qids"])
    q_ptr, q_term, q_w = sparse_rx.encode_queries([j["queries"][q] for q in qids], h.vocabulary)
    assert np.array_equal(q_ptr, z["score_q_ptr"]) and np.array_equal(q_term, z["score_q_term"])
    assert np.array_equal(q_w, z["score_q_weight"])
    # blank / OOV-only / punctuation-only queries encode to empty rows (the reference returns {} for them)
    p, t, w = sparse_rx.encode_queries(["", "   ", "zzzunknown qqqmissing", "?!", "w5 notaword w6"], h.vocabulary)
    assert list(np.diff(p)) == [0, 0, 0, 0, 2]


def test_tokenizer_cases():
    """Probed behaviour of re.findall(r'\\b\\w+\\b', text.lower())."""
    tk = sparse_rx.tokenize
    assert tk("Don't") == ["don", "t"]
    assert tk("U.S.A.") == ["u", "s", "a"]
    assert tk("3.14") == ["3", "14"]
    assert tk("café_1") == ["café_1"]
    assert tk("") == []


def test_npz_cache_schema_roundtrip(golden_dir, tmp_path):
    """save_index_npz writes the reference's .rag_cache schema (evaluate_rag_pipeline.py:280-296) and load_index_npz
    reads it back without pickle."""
    j = json.load(open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8"))
    h = sparse_rx.build_host_index(j["corpus"])
    p = tmp_path / "bm25_index_test.npz"
    sparse_rx.save_index_npz(p, h)
    z = np.load(p, allow_pickle=False)
    assert set(z.files) == {"tf_data", "tf_indices", "tf_indptr", "tf_shape", "doc_lengths", "idf", "vocabulary", "doc_ids", "avgdl"}
    assert z["tf_data"].dtype == np.float32 and z["tf_indices"].dtype == np.int32 and z["tf_shape"].dtype == np.int64
    h2 = sparse_rx.load_index_npz(p)
    assert h2.doc_ids == h.doc_ids and h2.vocabulary == h.vocabulary
    assert np.array_equal(h2.data, h.data) and np.array_equal(h2.indices, h.indices) and np.array_equal(h2.indptr, h.indptr)
    assert np.array_equal(h2.idf.view(np.uint32), h.idf.view(np.uint32)) and h2.avgdl == np.float32(h.avgdl)


def _consistent_shard(rng, V=37, n_docs=5000, tile_log2=11, unit_tiles=2, val_dtype=np.float16):
    """Arrays of one small shard in the blocked layout that satisfy every invariant read_shard_file checks."""
    from parity import np_build_blocks
    rows, cols = [], []
    for t in range(V):
        docs = np.sort(rng.choice(n_docs, rng.integers(0, 60), replace=False))
        rows += docs.tolist()
        cols += [t] * len(docs)
    order = np.lexsort((cols, rows))
    rows, cols = np.array(rows)[order], np.array(cols)[order]
    indptr = np.zeros(n_docs + 1, np.int64)
    indptr[1:] = np.cumsum(np.bincount(rows, minlength=n_docs))
    vals = (rng.integers(1, 9, len(cols)) / 4).astype(np.float32)  # exact in fp16
    term_ptr, post, skip, n_blocks = np_build_blocks(indptr, cols.astype(np.int32), vals, n_docs, V, tile_log2, unit_tiles, val_dtype)
    arrays = {"term_ptr": term_ptr, "post": post, "tile_skip": skip, "idf": rng.random(V).astype(np.float32),
              "term_bound": rng.random(V * 4).astype(np.float32)}
    meta = {"n_docs": n_docs, "vocab": V, "nnz": len(cols), "n_blocks": n_blocks, "doc_base": 123456789012, "tile_log2": tile_log2,
            "unit_tiles": unit_tiles, "val_type": 1 if val_dtype == np.float16 else 0, "block_pad": 256}
    return arrays, meta


def test_shard_file_roundtrip_and_corruption(tmp_path):
    """Native shard file: header + aligned raw arrays; roundtrip is bit-exact, corruption is detected."""
    from sparse_rx import shardfile
    rng = np.random.default_rng(3)
    arrays, meta = _consistent_shard(rng)
    p = str(tmp_path / "shard0.srx")
    shardfile.write_shard_file(p, arrays, meta)
    m2, a2 = shardfile.read_shard_file(p)
    assert m2 == meta
    assert set(a2) == set(arrays)
    for k in arrays:
        assert a2[k].dtype == arrays[k].dtype and np.array_equal(np.asarray(a2[k]).view(np.uint8), arrays[k].view(np.uint8)), k
    # optional array absent
    del arrays["term_bound"]
    shardfile.write_shard_file(p, arrays, meta)
    assert "term_bound" not in shardfile.read_shard_file(p)[1]
    # flip one payload byte -> checksum mismatch; truncate -> error; wrong magic -> error
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) - 4096 + 10] ^= 0x40  # inside the last array (idf), whose padded block ends the file
    open(p, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        shardfile.read_shard_file(p)
    open(p, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        shardfile.read_shard_file(p)
    open(p, "wb").write(b"NOTSHARD" + bytes(100))
    with pytest.raises(ValueError, match="not a sparse-rx shard file"):
        shardfile.read_shard_file(p)


def _rewrite_header(path, edit):
    """Apply `edit(header_dict)` to a shard file's JSON header and write it back WITH a matching header CRC (a header
    that is wrong but internally well-formed: what the structural validation has to catch)."""
    import struct
    import zlib
    from sparse_rx import shardfile
    raw = open(path, "rb").read()
    n0 = len(shardfile.MAGIC)
    version, hlen, _ = struct.unpack("<III", raw[n0:n0 + 12])
    hdr = json.loads(raw[n0 + 12:n0 + 12 + hlen].decode())
    edit(hdr)
    new = json.dumps(hdr, sort_keys=True).encode()
    data_start_old = (n0 + 12 + hlen + 4095) // 4096 * 4096
    data_start_new = (n0 + 12 + len(new) + 4095) // 4096 * 4096
    assert data_start_new == data_start_old
    out = raw[:n0] + struct.pack("<III", version, len(new), zlib.crc32(new) & 0xFFFFFFFF) + new
    out += b"\0" * (data_start_new - len(out)) + raw[data_start_old:]
    open(path, "wb").write(out)


def test_shard_file_header_is_validated(tmp_path):
    """A damaged or mismatched header must raise ValueError before anything is uploaded: the kernels index the arrays
    with the header's dims and get raw pointers without sizes (round-1 advisor finding)."""
    from sparse_rx import shardfile
    rng = np.random.default_rng(4)
    arrays, meta = _consistent_shard(rng, val_dtype=np.float32)
    good = str(tmp_path / "good.srx")
    shardfile.write_shard_file(good, arrays, meta)
    shardfile.read_shard_file(good)
    p = str(tmp_path / "bad.srx")

    def case(edit, match, verify=True):
        import shutil
        shutil.copy(good, p)
        _rewrite_header(p, edit)
        with pytest.raises(ValueError, match=match):
            shardfile.read_shard_file(p, verify=verify)

    # a single flipped header byte (no CRC fix-up) is caught by the header checksum
    raw = bytearray(open(good, "rb").read())
    raw[30] ^= 0x01
    open(p, "wb").write(raw)
    with pytest.raises(ValueError, match="header checksum"):
        shardfile.read_shard_file(p)
    # dims that disagree with the arrays (each would be an out-of-bounds device read)
    case(lambda h: h["meta"].__setitem__("vocab", meta["vocab"] + 1), "header dims require", verify=False)
    case(lambda h: h["meta"].__setitem__("n_docs", meta["n_docs"] * 4), "header dims require", verify=False)   # more tiles -> longer skip rows
    case(lambda h: h["meta"].__setitem__("n_blocks", meta["n_blocks"] + 8), "header dims require", verify=False)
    case(lambda h: h["meta"].__setitem__("nnz", 4 * meta["n_blocks"] + 1), "cannot hold", verify=False)
    case(lambda h: h["meta"].__setitem__("val_type", 1), "header dims require", verify=False)   # 6-word blocks: another array size
    case(lambda h: h["meta"].__setitem__("unit_tiles", 0), "out of range", verify=False)
    case(lambda h: h["meta"].__setitem__("tile_log2", 9), "header dims require", verify=False)
    case(lambda h: h["meta"].__setitem__("tile_log2", 40), "out of range", verify=False)
    case(lambda h: h["meta"].pop("n_docs"), "missing or out of range", verify=False)
    case(lambda h: h["meta"].__setitem__("nnz", -1), "out of range", verify=False)
    # offsets: negative, unaligned, overlapping, past the end
    case(lambda h: h["arrays"]["idf"].__setitem__("offset", -4096), "negative or not", verify=False)
    case(lambda h: h["arrays"]["idf"].__setitem__("offset", 100), "negative or not", verify=False)
    case(lambda h: h["arrays"]["idf"].__setitem__("offset", h["arrays"]["term_ptr"]["offset"]), "overlap", verify=False)
    case(lambda h: h["arrays"]["idf"].__setitem__("offset", 1 << 40), "past the end", verify=False)
    case(lambda h: h["arrays"]["post"].__setitem__("dtype", "float32"), "dtype", verify=False)
    case(lambda h: h["arrays"].pop("tile_skip"), "missing", verify=False)
    # contents the kernels trust: term_ptr, doc ids, skip rows (the last two need the data pass of verify=True)
    def swap(a, b):
        def f(h):
            h["arrays"][a]["offset"], h["arrays"][b]["offset"] = h["arrays"][b]["offset"], h["arrays"][a]["offset"]
        return f
    bad_tp = dict(arrays)
    bad_tp["term_ptr"] = arrays["term_ptr"].copy()
    bad_tp["term_ptr"][5] = bad_tp["term_ptr"][6] + 4  # decreasing
    shardfile.write_shard_file(p, bad_tp, meta)
    with pytest.raises(ValueError, match="term_ptr"):
        shardfile.read_shard_file(p, verify=False)
    bad_doc = dict(arrays)
    bad_doc["post"] = arrays["post"].copy()
    bad_doc["post"][3] = meta["n_docs"] + 7   # a doc slot of block 0
    shardfile.write_shard_file(p, bad_doc, meta)
    with pytest.raises(ValueError, match="doc ids outside"):
        shardfile.read_shard_file(p)
    bad_ts = dict(arrays)
    bad_ts["tile_skip"] = arrays["tile_skip"].copy()
    bad_ts["tile_skip"][-1] += 5  # the last term's row no longer ends at its posting count
    shardfile.write_shard_file(p, bad_ts, meta)
    with pytest.raises(ValueError, match="tile_skip"):
        shardfile.read_shard_file(p)


def test_query_batch_validation():
    """The host-array entry point checks srx_search's preconditions before launching (round-1 advisor finding): an
    out-of-range term id would be an out-of-bounds device read, a repeated term breaks one-posting-per-(doc, term)."""
    from sparse_rx.index import validate_query_batch
    ok = (np.array([0, 2, 2, 5], np.int32), np.array([3, 1, 9, 0, 4], np.int32), np.ones(5, np.float32))
    validate_query_batch(*ok, vocab=10)  # terms need not be ascending (token order is legal)
    validate_query_batch(np.array([0], np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), vocab=10)
    for q_ptr, q_term, msg in ((np.array([1, 2, 5]), ok[1], "start at 0"), (np.array([0, 3, 2, 5]), ok[1], "non-decreasing"),
                               (ok[0], np.array([3, 1, 10, 0, 4]), "out of range"), (ok[0], np.array([3, 1, -1, 0, 4]), "out of range"),
                               (ok[0], np.array([3, 3, 9, 0, 4]), "same term twice"), (ok[0], np.array([3, 1, 9, 0, 9]), "same term twice"),
                               (np.array([0, 2, 2, 7]), ok[1], "shorter")):
        with pytest.raises(ValueError, match=msg):
            validate_query_batch(q_ptr.astype(np.int32), q_term.astype(np.int32), ok[2], vocab=10)
    validate_query_batch(ok[0], np.array([3, 1, 3, 0, 4], np.int32), ok[2], vocab=10)  # the same term in DIFFERENT queries is fine


def test_encode_queries_token_order(golden_dir):
    """order="token": first-occurrence order of the in-vocabulary terms -- the pipeline twin's ``relevant_terms``
    (evaluate_rag_pipeline.py:360-370), pinned by the token-ordered lists the reference run recorded."""
    p = np.load(os.path.join(golden_dir, "pipeline_small.npz"))
    j = json.load(open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8"))
    h = sparse_rx.build_host_index(j["corpus"])
    qids = [str(q) for q in p["bm25_qids"]]
    q_ptr, q_term, q_w = sparse_rx.encode_queries([j["queries"][q] for q in qids], h.vocabulary, order="token")
    assert np.array_equal(q_ptr, p["bm25_q_ptr"]) and np.array_equal(q_term, p["bm25_q_term"]) and np.array_equal(q_w, p["bm25_q_weight"])
    a = sparse_rx.encode_queries(["w9 w8 w7 w9"], h.vocabulary, order="token")
    b = sparse_rx.encode_queries(["w9 w8 w7 w9"], h.vocabulary)
    assert list(a[1]) == [h.vocabulary[w] for w in ("w9", "w8", "w7")] and list(b[1]) == sorted(a[1])
    assert sorted(zip(a[1], a[2])) == list(zip(b[1], b[2]))
    with pytest.raises(ValueError):
        sparse_rx.encode_queries(["w1"], h.vocabulary, order="nope")
