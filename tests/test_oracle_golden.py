"""Pins oracle/ (the CPU restatement) against outputs of the reference itself (tests/golden)."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import np_oracle
from parity import assert_ranked_equal, assert_canonical_order


@pytest.fixture(scope="module")
def zipf(golden_dir):
    return np.load(os.path.join(golden_dir, "csr_zipf.npz"))


@pytest.fixture(scope="module")
def text(golden_dir):
    z = np.load(os.path.join(golden_dir, "text_small.npz"))
    with open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8") as f:
        j = json.load(f)
    return z, j


def _q(z, q, p="q_"):
    lo, hi = z[p + "ptr"][q], z[p + "ptr"][q + 1]
    return z[p + "term"][lo:hi], z[p + "weight"][lo:hi]


def test_bm25_scores_bit_exact_zipf(zipf):
    z = zipf
    for q in range(z["bm25_full"].shape[0]):
        t, w = _q(z, q)
        s = oracle.bm25_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], t, w,
                               float(z["k1"]), float(z["b"]), float(z["avgdl"]))
        assert np.array_equal(s.view(np.uint32), z["bm25_full"][q].view(np.uint32)), q
        s2 = np_oracle.bm25_scores_taat(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], t, w,
                                        float(z["k1"]), float(z["b"]), float(z["avgdl"]))
        assert np.array_equal(s2.view(np.uint32), z["bm25_full"][q].view(np.uint32)), q


def test_tfidf_scores_bit_exact_zipf(zipf):
    z = zipf
    for q in range(z["tfidf_full"].shape[0]):
        t, w = _q(z, q)
        s = oracle.tfidf_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["idf_tfidf"], t, w)
        assert np.array_equal(s.view(np.uint32), z["tfidf_full"][q].view(np.uint32)), q
        s2 = np_oracle.tfidf_scores_taat(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["idf_tfidf"], t, w)
        assert np.array_equal(s2.view(np.uint32), z["tfidf_full"][q].view(np.uint32)), q


def test_bm25_scores_bit_exact_text(text):
    z, _ = text
    for q in range(z["full_scores"].shape[0]):
        t, w = _q(z, q, "score_q_")
        s = oracle.bm25_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], t, w,
                               float(z["k1"]), float(z["b"]), float(z["avgdl"]))
        assert np.array_equal(s.view(np.uint32), z["full_scores"][q].view(np.uint32)), str(z["score_qids"][q])


@pytest.mark.parametrize("k", [10, 100])
def test_search_batch_matches_reference_topk(zipf, k):
    z = zipf
    od, osc, oc = oracle.search_batch(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"],
                                      z["q_ptr"], z["q_term"], z["q_weight"], k, float(z["k1"]), float(z["b"]),
                                      float(z["avgdl"]))
    ed, es, ec = z[f"top{k}_doc"], z[f"top{k}_score"], z[f"top{k}_count"]
    assert np.array_equal(oc, ec)
    for q in range(len(oc)):
        t, w = _q(z, q)
        full = oracle.bm25_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], t, w,
                                  float(z["k1"]), float(z["b"]), float(z["avgdl"]))
        c = oc[q]
        assert_ranked_equal(od[q, :c], osc[q, :c], ed[q, :c], es[q, :c], k=k, full_scores=full, label=f"q{q}")
        assert_canonical_order(od[q, :c], osc[q, :c], label=f"q{q}")
        assert np.all(od[q, c:] == -1)


def test_topk_protocol_cases():
    """The reference's own top-k acceptance protocol (tests/topk_selection.py:298-379): cases
    (n, k, dist), reference = argsort(-scores)[:k]; here at the stricter bar (exact order mod ties)."""
    rng = np.random.default_rng(7)
    cases = [(100, 10, "normal"), (1000, 50, "uniform"), (500, 5, "zipf"), (200, 100, "bimodal"), (50, 50, "normal"),
             (30, 100, "uniform"), (1, 1, "normal")]
    for n, k, dist in cases:
        if dist == "normal":
            s = rng.normal(size=n)
        elif dist == "uniform":
            s = rng.uniform(size=n)
        elif dist == "zipf":
            s = rng.zipf(1.5, size=n).astype(np.float64)  # many exact ties
        else:
            s = np.concatenate([rng.normal(0, 1, n // 2), rng.normal(5, 1, n - n // 2)])
        s = s.astype(np.float32)
        idx, sc = oracle.topk(s, k)
        ref = np.argsort(-s, kind="stable")[:k]
        assert_ranked_equal(idx, sc, ref, s[ref], k=k, full_scores=s, label=f"{n},{k},{dist}")
        assert_canonical_order(idx, sc)
        i2, s2 = np_oracle.topk_ranked(s, k)
        assert np.array_equal(i2, idx) and np.array_equal(s2, sc)


def test_fp64_variant_is_close(zipf):
    """Numba-faithful fp64 restatement vs the fp32 path: diagnostic bound."""
    z = zipf
    for q in range(8):
        t, w = _q(z, q)
        a = oracle.bm25_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], t, w,
                               float(z["k1"]), float(z["b"]), float(z["avgdl"]))
        b = oracle.bm25_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], t, w,
                               float(z["k1"]), float(z["b"]), float(z["avgdl"]), fp64=True)
        assert np.max(np.abs(a - b)) < 1e-4


def test_empty_query_and_empty_rows(zipf):
    z = zipf
    q_ptr = np.array([0, 0, 1], dtype=np.int32)
    od, osc, oc = oracle.search_batch(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], q_ptr,
                                      np.array([5], np.int32), np.array([1.0], np.float32), 10, 1.2, 0.75,
                                      float(z["avgdl"]))
    assert oc[0] == 0 and np.all(od[0] == -1)
    assert oc[1] > 0


# ---------------------------------------------------------------------------------------------------------------
# pipeline twin (evaluate_rag_pipeline.py:162-479): its NumPy fallback accumulates in QUERY-TOKEN order
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pipeline(golden_dir):
    p = np.load(os.path.join(golden_dir, "pipeline_small.npz"))
    with open(os.path.join(golden_dir, "pipeline_small.json"), encoding="utf-8") as f:
        return p, json.load(f)


@pytest.mark.parametrize("name,tfidf", [("bm25", False), ("splade", True)])
def test_given_order_scores_bit_exact_pipeline(text, pipeline, name, tfidf):
    """oracle_scores_given_order == _numpy_score_documents of the reference's OptimizedRetriever, bit for bit; and the
    ascending-term order does NOT reproduce it on some queries (why the order is part of the contract)."""
    z, _ = text
    p, _ = pipeline
    ptr = p[f"{name}_q_ptr"]
    differ_sorted = 0
    for q in range(len(ptr) - 1):
        t, w = p[f"{name}_q_term"][ptr[q]:ptr[q + 1]], p[f"{name}_q_weight"][ptr[q]:ptr[q + 1]]
        args = (z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], p[f"{name}_idf"])
        s = oracle.scores_given_order(*args, t, w, 1.2, 0.75, float(p[f"{name}_avgdl"]), tfidf=tfidf)
        exp = p[f"{name}_full_scores"][q]
        assert np.array_equal(s.view(np.uint32), exp.view(np.uint32)), str(p[f"{name}_qids"][q])
        o = np.argsort(t)
        s2 = oracle.scores_given_order(*args, t[o], w[o], 1.2, 0.75, float(p[f"{name}_avgdl"]), tfidf=tfidf)
        # the ascending-order sum is what simd_bm25_score / simd_tfidf_score (CSR row order) give
        s3 = (oracle.tfidf_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], p[f"{name}_idf"], t, w) if tfidf else
              oracle.bm25_scores(*args, t, w, 1.2, 0.75, float(p[f"{name}_avgdl"])))
        assert np.array_equal(s2.view(np.uint32), s3.view(np.uint32))
        differ_sorted += not np.array_equal(s2.view(np.uint32), exp.view(np.uint32))
    assert differ_sorted > 0


@pytest.mark.parametrize("name,mode,k1,b", [("bm25", oracle.MODE_BM25_F32_GIVEN_ORDER, 1.2, 0.75),
                                            ("bm25_custom", oracle.MODE_BM25_F32_GIVEN_ORDER, 1.6, 0.8),
                                            ("splade", oracle.MODE_TFIDF_F32_GIVEN_ORDER, 1.2, 0.75),
                                            ("dpr", oracle.MODE_TFIDF_F32_GIVEN_ORDER, 1.2, 0.75)])
def test_search_batch_given_order_matches_pipeline_twin_results(text, pipeline, name, mode, k1, b):
    """oracle.search_batch in the given-order modes against the ``search`` results the reference's OptimizedRetriever
    produced (k = 5, 50), modulo the reference's unspecified order inside exact ties."""
    import sparse_rx
    z, j = text
    _, pj = pipeline
    h = sparse_rx.build_host_index(j["corpus"], idf_kind="bm25" if name.startswith("bm25") else "tfidf")
    qids = [q for q, t in j["queries"].items()]
    q_ptr, q_term, q_w = sparse_rx.encode_queries([j["queries"][q] for q in qids], h.vocabulary, order="token")
    row = {d: i for i, d in enumerate(h.doc_ids)}
    for k in ("5", "50"):
        od, osc, oc = oracle.search_batch(h.indptr, h.indices, h.data, h.doc_lengths, h.idf, q_ptr, q_term, q_w, int(k), k1, b,
                                          h.avgdl, mode=mode)
        exp = pj[name]["results"][k]
        assert list(exp.keys()) == qids
        for i, qid in enumerate(qids):
            e = exp[qid]
            c = oc[i]
            assert_ranked_equal(od[i, :c], osc[i, :c], [row[d] for d in e], np.array(list(e.values()), np.float32), k=int(k),
                                label=f"{name} k={k} {qid}")


def test_reference_written_npz_cache_loads(text, golden_dir):
    """load_index_npz reads the ``.rag_cache/*.npz`` files the REFERENCE wrote (``_save_cached_index``,
    evaluate_rag_pipeline.py:280-296; copied to tests/golden/ref_cache by make_golden.py) without pickle, and the
    index state equals what build_host_index computes from the text (bit-equal arrays)."""
    import sparse_rx
    z, j = text
    for fn, kind in (("bm25_index_4619a0fc.npz", "bm25"), ("splade_index_4619a0fc.npz", "tfidf")):
        h = sparse_rx.load_index_npz(os.path.join(golden_dir, "ref_cache", fn))
        g = sparse_rx.build_host_index(j["corpus"], idf_kind=kind)
        assert h.doc_ids == g.doc_ids and h.vocabulary == g.vocabulary
        assert np.array_equal(h.indptr, g.indptr) and np.array_equal(h.indices, g.indices) and np.array_equal(h.data, g.data)
        assert np.array_equal(h.doc_lengths, g.doc_lengths) and h.avgdl == g.avgdl
        assert np.array_equal(np.asarray(h.idf).view(np.uint32), g.idf.view(np.uint32))
        assert h.idf.dtype == np.float32 and h.data.dtype == np.float32 and h.indices.dtype == np.int32


def test_c1_fiqa_shaped_plumbing():
    """BASELINE config C1 without a GPU (the reference's own CPU-runnable case): FiQA-shaped synthetic text through the
    full host path (tokenise, vocabulary, CSR, idf, avgdl), scored by the C oracle (doc-at-a-time full CSR scan) and
    cross-checked against the independent NumPy term-at-a-time restatement, k = 10."""
    import sparse_rx
    from sparse_rx import synth
    corpus, queries = synth.fiqa_shaped_text()
    assert len(corpus) == synth.C1_DOCS == 57_638 and len(queries) == 100
    h = sparse_rx.build_host_index(corpus)
    assert h.n_docs == 57_638 and 70_000 < h.vocab_size <= 80_000 and 120 < h.avgdl < 140
    assert (h.idf < 0).sum() > 0  # hot words with df > N/2: negative idf must be accumulated, not skipped
    q_ptr, q_term, q_w = sparse_rx.encode_queries(list(queries.values()), h.vocabulary)
    assert q_ptr[-1] > 500
    od, osc, oc = oracle.search_batch(h.indptr, h.indices, h.data, h.doc_lengths, h.idf, q_ptr, q_term, q_w, 10, 1.2, 0.75, h.avgdl)
    assert np.all(oc <= 10) and oc.max() == 10
    for q in (0, 1, 17, 99):
        t, w = q_term[q_ptr[q]:q_ptr[q + 1]], q_w[q_ptr[q]:q_ptr[q + 1]]
        s = np_oracle.bm25_scores_taat(h.indptr, h.indices, h.data, h.doc_lengths, h.idf, t, w, 1.2, 0.75, h.avgdl)
        idx, sc = np_oracle.topk_ranked(s, 10)
        keep = sc > 0
        assert np.array_equal(od[q, :oc[q]], idx[keep]) and np.array_equal(osc[q, :oc[q]].view(np.uint32), sc[keep].view(np.uint32))


def _deep_fixture(golden_dir):
    """tests/golden/text_deep.npz (written by the reference's search_bm25 at top_k = 1500 / 5000 on 2600 docs):
    (corpus dict, queries dict, {k: {qid: (doc rows, scores)}})."""
    z = np.load(os.path.join(golden_dir, "text_deep.npz"))
    texts = bytes(z["texts"]).decode("utf-8").split("\n")
    corpus = {f"doc{i}": {"text": t} for i, t in enumerate(texts)}
    queries = {str(q): str(t) for q, t in zip(z["qids"], z["qtexts"])}
    exp = {k: {q: (z[f"k{k}_{q}_doc"], z[f"k{k}_{q}_score"]) for q in queries} for k in (1500, 5000)}
    return corpus, queries, exp


def test_deep_ranking_matches_reference(golden_dir):
    """top_k beyond 1024 and top_k >= n_docs (retrieval.py:272-284): the oracle's unbounded top-k reproduces the rows the
    reference returned (modulo its unspecified order inside exact ties)."""
    import sparse_rx
    corpus, queries, exp = _deep_fixture(golden_dir)
    hi = sparse_rx.build_host_index(corpus)
    q = sparse_rx.encode_queries(list(queries.values()), hi.vocabulary)
    assert hi.n_docs == 2600
    for k in (1500, 5000):
        kk = min(k, hi.n_docs)
        d, s, c = oracle.search_batch(hi.indptr, hi.indices, hi.data, hi.doc_lengths, hi.idf, q[0], q[1], q[2], kk, 1.2, 0.75, hi.avgdl)
        for i, qid in enumerate(queries):
            ed, es = exp[k][qid]
            assert c[i] == len(ed), (k, qid)
            assert_ranked_equal(d[i, : c[i]], s[i, : c[i]], ed, es, k=kk, label=f"deep k={k} {qid}")
    assert max(len(v[0]) for v in exp[5000].values()) > 1024  # the fixture does reach past the engine's list capacity
