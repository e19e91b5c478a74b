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
    assert L.srx_version() == 100


def test_limits_and_error_strings():
    lim = _capi.limits()
    assert lim == {"max_k": 1024, "max_tile_log2": 14, "hash_cap": 4096, "threads": 256}
    L = _capi.lib()
    assert L.srx_index_create(None, None) == -1
    assert b"null" in L.srx_last_error()
    assert L.srx_search(None, None, None, None, 1, 10, None, None, None, None, 0, None) == -1
    with pytest.raises(ValueError):
        _capi.check(L.srx_merge_workspace_bytes(1, 0, 10), "srx_merge_workspace_bytes")
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


def test_shard_file_roundtrip_and_corruption(tmp_path):
    """Native shard file: header + aligned raw arrays; roundtrip is bit-exact, corruption is detected."""
    from sparse_rx import shardfile
    rng = np.random.default_rng(3)
    V, nnz, n_tiles = 37, 1000, 3
    arrays = {"term_ptr": np.sort(rng.integers(0, nnz, V + 1)).astype(np.int64),
              "post_doc": rng.integers(0, 5000, nnz + 16).astype(np.int32),
              "post_val": rng.random(nnz + 16).astype(np.float16),
              "tile_skip": rng.integers(0, 99, V * (n_tiles + 1)).astype(np.int32),
              "idf": rng.random(V).astype(np.float32),
              "term_bound": rng.random(V * 4).astype(np.float32)}
    meta = {"n_docs": 5000, "vocab": V, "nnz": nnz, "doc_base": 123456789012, "tile_log2": 11, "post_pad": 16}
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
