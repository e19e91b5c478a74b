"""Dense INT8 side: the oracle and the host-side quantizers against the fixture produced by the
reference's QuantizedEmbeddingRetriever (tests/golden/make_golden.py dense), and -- on the GPU -- srx_dense_search_i8
against the oracle (bit-exact) and against the reference's recorded search results."""
import json
import os

import numpy as np
import pytest

import sparse_rx
from oracle import np_oracle


@pytest.fixture(scope="module")
def dense_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "dense_int8.npz"))
    j = json.load(open(os.path.join(golden_dir, "dense_int8.json")))
    return z, j


def test_quantizers_match_reference(dense_golden):
    z, _ = dense_golden
    q, s = sparse_rx.quantize_symmetric(z["emb"])
    assert q.dtype == np.int8 and np.array_equal(q, z["corpus_int8"])
    assert s.dtype == np.float32 and np.array_equal(s.view(np.uint32), z["corpus_scales"].view(np.uint32))
    for i in range(len(z["qemb"])):
        q8, qs = sparse_rx.quantize_query_symmetric(z["qemb"][i])
        assert np.array_equal(q8, z["query_int8"][i]) and np.float32(qs).view(np.uint32) == z["query_scales"][i].view(np.uint32)


def test_oracle_similarities_and_topk_match_reference(dense_golden):
    z, j = dense_golden
    sims = np_oracle.int8_similarities(z["query_int8"], z["corpus_int8"], z["query_scales"], z["corpus_scales"])
    assert np.array_equal(sims.view(np.uint32), z["similarities"].view(np.uint32))  # bit for bit
    for k_s, res in j["results"].items():
        k = int(k_s)
        d, s, n = np_oracle.dense_topk(sims, k)
        for i, qid in enumerate(j["qids"]):
            exp = res[qid]
            got_ids = [j["doc_ids"][x] for x in d[i, : n[i]]]
            # same scores rank by rank; ids equal except inside groups of exactly tied scores (reference order unspecified)
            assert [float(x) for x in s[i, : n[i]]] == list(exp.values())
            assert set(got_ids) == set(exp) or len(set(exp.values())) < len(exp)
            for did, sc in exp.items():
                if list(exp.values()).count(sc) == 1:
                    assert got_ids[list(exp.values()).index(sc)] == did


def _rand_case(rng, n_docs, dim, nq):
    c = rng.integers(-127, 128, (n_docs, dim)).astype(np.int8)
    q = rng.integers(-127, 128, (nq, dim)).astype(np.int8)
    cs = (rng.random(n_docs) + 0.01).astype(np.float32)
    qs = (rng.random(nq) + 0.01).astype(np.float32) / 127
    return c, cs, q, qs


@pytest.mark.gpu
def test_dense_int8_matches_oracle_bit_exact():
    rng = np.random.default_rng(77)
    # small corpora take the score-matrix path; >= 65 536 docs the filtered path (sample threshold + fused filter)
    for n_docs, dim, nq, k in ((1000, 32, 5, 10), (4133, 96, 33, 100), (20_000, 384, 70, 7), (777, 48, 3, 1000), (50_000, 768, 260, 100),
                               (130, 1024, 64, 128), (150_001, 64, 300, 100), (90_000, 128, 17, 1000)):
        c, cs, q, qs = _rand_case(rng, n_docs, dim, nq)
        if n_docs > 1000:
            c[5] = c[6]  # exact score ties
        ix = sparse_rx.DenseInt8Index(c, cs, doc_base=1000)
        kk = min(k, n_docs)
        d, s, n = ix.search(q, qs, kk)
        ed, es, en = np_oracle.dense_topk(np_oracle.int8_similarities(q, c, qs, cs), kk)
        ed = np.where(ed >= 0, ed + 1000, -1)
        assert np.array_equal(n, en), (n_docs, dim, nq, k)
        assert np.array_equal(s.view(np.uint32), es.view(np.uint32)), (n_docs, dim, nq, k)
        assert np.array_equal(d, ed), (n_docs, dim, nq, k)
        # the row-major corpus (what the reference holds) through srx_dense_search_i8: the same rows as the fragment-ordered default
        d2, s2, n2 = sparse_rx.DenseInt8Index(c, cs, doc_base=1000, packed=False).search(q, qs, kk)
        assert np.array_equal(n2, n) and np.array_equal(d2, d) and np.array_equal(s2.view(np.uint32), s.view(np.uint32)), (n_docs, dim)


@pytest.mark.gpu
def test_dense_int8_mfma_layout_exact_integers():
    """Asymmetric integer data through the MFMA path: unit scales make the score the integer dot product itself."""
    n_docs, dim, nq = 200, 64, 40
    c = np.zeros((n_docs, dim), np.int8)
    q = np.zeros((nq, dim), np.int8)
    for i in range(n_docs):
        c[i, i % dim] = 1 + (i % 100)           # doc i: one non-zero at column i % dim
    for j in range(nq):
        q[j, :] = (np.arange(dim) % 7) + 1 + (j % 5)  # asymmetric in (query, column)
    ix = sparse_rx.DenseInt8Index(c, np.ones(n_docs, np.float32))
    d, s, n = ix.search(q, np.ones(nq, np.float32), n_docs)
    exp = (q.astype(np.int32) @ c.T.astype(np.int32)).astype(np.float32)
    for j in range(nq):
        got = np.zeros(n_docs, np.float32)
        got[d[j, : n[j]]] = s[j, : n[j]]
        assert np.array_equal(got, exp[j]), j


@pytest.mark.gpu
def test_dense_int8_reference_fixture_end_to_end(dense_golden):
    z, j = dense_golden
    qi = sparse_rx.QuantizedEmbeddingIndex()
    qi.build(j["doc_ids"], z["emb"])
    qembs = {qid: z["qemb"][i] for i, qid in enumerate(j["qids"])}
    for k_s, res in j["results"].items():
        got = qi.search(qembs, top_k=int(k_s))
        for qid in j["qids"]:
            exp = res[qid]
            assert list(got[qid].values()) == list(exp.values())  # same scores rank by rank (python floats of the same f32)
            for did, sc in exp.items():
                if list(exp.values()).count(sc) == 1:
                    assert list(got[qid])[list(exp.values()).index(sc)] == did


@pytest.mark.gpu
def test_dense_int8_candidate_overflow_falls_back():
    """Degenerate score distributions (every doc identical: all scores tie at the threshold) overflow the filtered path's
    candidate buffers; those queries are re-ranked through the score matrix and the result is still exact -- mixed
    with queries that do not overflow."""
    rng = np.random.default_rng(5)
    n_docs, dim, nq, k = 70_000, 32, 6, 50
    c = np.tile(rng.integers(1, 100, (1, dim)).astype(np.int8), (n_docs, 1))
    cs = np.ones(n_docs, np.float32)
    c[::7] = rng.integers(-127, 128, (len(c[::7]), dim)).astype(np.int8)  # some variety: not every query overflows
    q = rng.integers(-127, 128, (nq, dim)).astype(np.int8)
    q[0] = 1
    q[3] = np.abs(q[3])  # positive queries: tens of thousands of identical positive scores
    qs = np.full(nq, 1.0 / 127, np.float32)
    ix = sparse_rx.DenseInt8Index(c, cs)
    d, s, n = ix.search(q, qs, k)
    ed, es, en = np_oracle.dense_topk(np_oracle.int8_similarities(q, c, qs, cs), k)
    assert np.array_equal(n, en) and np.array_equal(s.view(np.uint32), es.view(np.uint32)) and np.array_equal(d, ed)


@pytest.mark.gpu
def test_dense_int8_screen_edge_cases():
    """The fused filter screens accumulator rows in fp32 before the exact fp64 chain (csrc/dense.hip): cases that sit on the
    screen's edges -- few distinct dot products (whole groups of docs tie with the threshold exactly), scales spread over 40
    orders of magnitude, zero / negative / denormal scales (no screen for such a query; a negative scale flips the sign of the
    scores: negative dot products then rank) -- must still be the oracle's rows bit for bit."""
    rng = np.random.default_rng(2027)
    for n_docs, dim, nq, k, lohi in ((80_000, 32, 40, 100, 3), (70_001, 64, 33, 10, 2), (66_000, 768, 36, 64, 128)):
        c = rng.integers(-lohi + 1, lohi, (n_docs, dim)).astype(np.int8)
        q = rng.integers(-lohi + 1, lohi, (nq, dim)).astype(np.int8)
        for scales in ("flat", "wild"):
            if scales == "flat":
                cs = np.ones(n_docs, np.float32)
                qs = np.ones(nq, np.float32)
            else:
                cs = (10.0 ** rng.integers(-18, 18, n_docs)).astype(np.float32) * (rng.random(n_docs) + 0.5).astype(np.float32)
                qs = (10.0 ** rng.integers(-18, 18, nq)).astype(np.float32)
                cs[::11] = 0.0
                cs[3::13] *= -1.0
                cs[5::17] = np.float32(1e-41)  # denormal
            qs[1] = 0.0
            qs[2] = -qs[2]
            qs[4] = np.float32(1e-42)
            ix = sparse_rx.DenseInt8Index(c, cs)
            d, s, n = ix.search(q, qs, k)
            with np.errstate(over="ignore", under="ignore"):
                ed, es, en = np_oracle.dense_topk(np_oracle.int8_similarities(q, c, qs, cs), k)
            assert np.array_equal(n, en), (n_docs, dim, scales)
            assert np.array_equal(s.view(np.uint32), es.view(np.uint32)), (n_docs, dim, scales)
            assert np.array_equal(d, ed), (n_docs, dim, scales)


@pytest.mark.gpu
def test_dense_f32_search_by_vector():
    """srx_dense_search_f32 against np.dot(embedding_index, query_vector) (retrieval.py:411-423) -- the call the
    reference makes; its BLAS summation order is unspecified, so: scores within 1e-5 relative of the float64 value
    (north_star allows 1e-4), ranking correct up to swaps inside that tolerance, only scores > 0, rows padded."""
    rng = np.random.default_rng(9)
    for n_docs, dim, nq, k in ((3000, 64, 1, 10), (20_000, 384, 5, 100), (5000, 100, 9, 7), (70_000, 768, 2, 1000), (400, 1024, 3, 500)):
        emb = rng.standard_normal((n_docs, dim)).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        ix = sparse_rx.DenseF32Index(emb, doc_base=7)
        kk = min(k, n_docs)
        d, s, n = ix.search(q, kk)
        exact = (q.astype(np.float64) @ emb.astype(np.float64).T)
        blas = np.stack([np.dot(emb, q[i]) for i in range(nq)])  # what the reference computes
        tol = 1e-5 * np.abs(emb.astype(np.float64)) @ np.abs(q.astype(np.float64)).T  # per (doc, query) bound, [n_docs, nq]
        for i in range(nq):
            cnt = int(n[i])
            pos = np.sort(exact[i][exact[i] > 0])[::-1]
            assert abs(cnt - min(kk, len(pos))) <= 1  # a score within rounding of 0 may fall on either side
            docs = d[i, :cnt] - 7
            assert len(set(docs.tolist())) == cnt and docs.min() >= 0 and docs.max() < n_docs
            assert np.all(np.abs(s[i, :cnt] - exact[i][docs]) <= tol[docs, i] + 1e-12)
            assert np.all(np.abs(s[i, :cnt] - blas[i][docs]) <= 2 * tol[docs, i] + 1e-12)
            assert np.all(s[i, : cnt - 1] >= s[i, 1:cnt])  # ranked
            if cnt:  # nothing better than the worst returned score was left out (beyond the tolerance)
                left_out = np.setdiff1d(np.arange(n_docs), docs)
                if len(left_out) and cnt == kk:
                    assert exact[i][left_out].max() <= s[i, cnt - 1] + 2 * tol[left_out, i].max()
            assert np.all(d[i, cnt:] == -1) and np.all(s[i, cnt:] == 0)


@pytest.mark.gpu
def test_service_search_by_vector(tmp_path):
    """RetrievalService.search_by_vector mirror: embeddings given directly or memory-mapped from embedding_path with the
    row count of doc_ids (retrieval.py:320-339), result list as in :425-436; ValueError without an embedding index."""
    rng = np.random.default_rng(3)
    corpus = {f"d{i}": {"text": f"word{i} common"} for i in range(50)}
    emb = rng.standard_normal((50, 96)).astype(np.float32)
    emb[7] = 0.0  # a score of exactly 0: returned under the default min_score = 0.0 (retrieval.py:425-427), after the positives
    qv = rng.standard_normal(96).astype(np.float32)
    exp = np.dot(emb, qv)
    order = [i for i in np.argsort(-exp) if exp[i] > 0][:10]
    n_pos = int((exp > 0).sum())
    svc = sparse_rx.RetrievalService()
    svc.build_bm25_index(corpus)
    with pytest.raises(ValueError, match="No embedding index available"):
        svc.search_by_vector(qv)
    svc.set_embeddings(emb)
    got = svc.search_by_vector(qv, k=10)
    assert [r["doc_id"] for r in got] == [f"d{i}" for i in order]
    assert np.allclose([r["score"] for r in got], exp[order], rtol=1e-5, atol=1e-6)
    assert len(svc.search_by_vector(qv, k=10, min_score=float(exp[order[3]]) - 1e-6)) == 4
    # min_score <= 0: zero and negative scores come back too, down to the threshold, in the reference's order
    full = [int(i) for i in np.argsort(-exp, kind="stable")]
    got0 = svc.search_by_vector(qv, k=50)  # default min_score = 0.0
    assert [r["doc_id"] for r in got0] == [f"d{i}" for i in full[: n_pos + 1]] and got0[-1] == {"doc_id": "d7", "score": 0.0}
    gotn = svc.search_by_vector(qv, k=50, min_score=-1e9)
    assert [r["doc_id"] for r in gotn] == [f"d{i}" for i in full]
    assert np.allclose([r["score"] for r in gotn], exp[full], rtol=1e-5, atol=1e-6)
    thr = float(exp[full[n_pos + 5]])
    gott = svc.search_by_vector(qv, k=50, min_score=thr - 1e-6)
    assert [r["doc_id"] for r in gott] == [f"d{i}" for i in full[: n_pos + 6]]
    assert [r["doc_id"] for r in svc.search_by_vector(qv, k=3, min_score=-1e9)] == [f"d{i}" for i in full[:3]]
    p = tmp_path / "emb.bin"
    emb.tofile(p)
    svc2 = sparse_rx.RetrievalService(embedding_path=str(p))
    svc2.build_bm25_index(corpus)
    assert [r["doc_id"] for r in svc2.search_by_vector(qv, k=5)] == [f"d{i}" for i in order[:5]]


def test_simulated_embedding_generators_match_reference(golden_dir):
    """QuantizedEmbeddingRetriever's simulated embeddings (retriever_registry.py:409-433, 526-536) restated: same legacy
    NumPy streams, bit for bit (constructing the class does not touch the GPU)."""
    z = np.load(os.path.join(golden_dir, "dense_synth.npz"))
    r = sparse_rx.QuantizedEmbeddingRetriever("dpr", "fixture", embedding_dim=24)
    syn = r.synthetic_embeddings(137)
    assert syn.dtype == np.float32 and np.array_equal(syn.view(np.uint32), z["synthetic_137x24"].view(np.uint32))
    qe = r.query_embedding_from_seed(int(z["query_seed"]))
    assert qe.dtype == np.float32 and np.array_equal(qe.view(np.uint32), z["query_24"].view(np.uint32))
    assert sparse_rx.QuantizedEmbeddingRetriever("dpr", "x", quantization_method="asymmetric").quantization_method == "asymmetric"
    assert isinstance(sparse_rx.RetrieverRegistry.create({"type": "dpr", "params": {"embedding_dim": 64}}), sparse_rx.QuantizedEmbeddingRetriever)


@pytest.mark.gpu
def test_quantized_embedding_retriever_end_to_end():
    """Registry-created dense retriever: build on simulated embeddings, batched search == per-query oracle on the same
    quantized arrays (INT8 bit-exact; fp32 mode within tolerance of np.dot)."""
    corpus = {f"doc{i}": {"text": f"t{i}"} for i in range(400)}
    queries = {"a": "alpha beta", "b": "", "c": "gamma"}
    r = sparse_rx.RetrieverRegistry.create({"type": "contriever", "params": {"embedding_dim": 96}})
    r.build_index_from_corpus(corpus)
    got = r.search(queries, top_k=7)
    assert got["b"] == {} and set(got) == {"a", "b", "c"}
    for qid in ("a", "c"):
        q8, qs = sparse_rx.quantize_query_symmetric(r._generate_query_embedding(queries[qid]))
        sims = np_oracle.int8_similarities(q8[None, :], r.corpus_embeddings_int8, np.array([qs], np.float32), r.corpus_scales)
        ed, es, en = np_oracle.dense_topk(sims, 7)
        assert list(got[qid]) == [r.doc_ids[i] for i in ed[0, : en[0]]]
        assert [np.float32(v) for v in got[qid].values()] == list(es[0, : en[0]])
    f = sparse_rx.QuantizedEmbeddingRetriever("dpr", "m", embedding_dim=64, use_quantization=False)
    f.build_index_from_corpus(corpus)
    gf = f.search({"a": "alpha beta"}, top_k=5)["a"]
    exp = np.dot(f.corpus_embeddings_fp32, f._generate_query_embedding("alpha beta"))
    order = [i for i in np.argsort(-exp) if exp[i] > 0][:5]
    assert list(gf) == [f.doc_ids[i] for i in order] and np.allclose(list(gf.values()), exp[order], rtol=1e-5, atol=1e-6)


# ---- asymmetric (uint8) scheme: retriever_registry.py:449-462 (writer), 486-491 (query), 550-559 (similarity) ----
@pytest.fixture(scope="module")
def asym_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "dense_uint8_asym.npz"))
    j = json.load(open(os.path.join(golden_dir, "dense_uint8_asym.json")))
    e = np.load(os.path.join(golden_dir, "dense_int8.npz"))  # the same embeddings went through both schemes
    return z, j, e


def _same_ranking_within(got_ids, got_scores, exp: dict, rtol=1e-5):
    """Scores agree rank by rank within the fp32 dot's summation-order tolerance; ids agree wherever the expected
    neighbours are further apart than that tolerance (the reference's BLAS order is unspecified).  The de-quantized
    vectors are not normalised (the reader's scale / min indexing makes some rows large), and the error of an fp32 dot
    scales with sum |q_i d_i|, not with the result: the absolute tolerance follows the largest score of the row."""
    ev = np.array(list(exp.values()), dtype=np.float64)
    atol = 1e-6 * max(1.0, float(np.max(np.abs(ev))) if len(ev) else 1.0)
    assert len(got_ids) == len(exp)
    assert np.allclose(np.array(got_scores, dtype=np.float64), ev, rtol=rtol, atol=atol)
    ids = list(exp)
    for r, did in enumerate(ids):
        lo = abs(ev[r] - ev[r - 1]) if r > 0 else np.inf
        hi = abs(ev[r] - ev[r + 1]) if r + 1 < len(ev) else np.inf
        if min(lo, hi) > 4 * (atol + rtol * abs(ev[r])):
            assert got_ids[r] == did


def test_asymmetric_quantizers_match_reference(asym_golden):
    z, _, e = asym_golden
    q, s = sparse_rx.quantize_asymmetric(e["emb"])
    assert q.dtype == np.uint8 and np.array_equal(q, z["corpus_uint8"])
    assert s.dtype == np.float32 and s.shape == (2 * len(q),) and np.array_equal(s.view(np.uint32), z["corpus_scales"].view(np.uint32))
    for i in range(len(e["qemb"])):
        q8, qs = sparse_rx.quantize_query_asymmetric(e["qemb"][i])
        assert q8.dtype == np.uint8 and np.array_equal(q8, z["query_uint8"][i])
        assert np.array_equal(qs.view(np.uint32), z["query_scales"][i].view(np.uint32))


def test_asymmetric_oracle_matches_reference(asym_golden):
    """The restated similarity (the reader's [2 i], [2 i + 1] indexing of the concatenated table included) against the
    rows the reference computed: fp32 BLAS dot vs float64 dot, so a tolerance; and the recorded search results."""
    z, j, _ = asym_golden
    sims = np_oracle.uint8_asymmetric_similarities(z["query_uint8"], z["query_scales"], z["corpus_uint8"], z["corpus_scales"])
    assert np.allclose(sims, z["similarities"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(z["similarities"]))))
    for k_s, res in j["results"].items():
        d, s, n = np_oracle.dense_topk(sims, int(k_s))
        for i, qid in enumerate(j["qids"]):
            _same_ranking_within([j["doc_ids"][x] for x in d[i, : n[i]]], list(s[i, : n[i]]), res[qid])


@pytest.mark.gpu
def test_dense_uint8_asymmetric_reference_fixture(asym_golden):
    """srx_dense_search_u8 on the reference's own quantized arrays against the results its search recorded."""
    z, j, _ = asym_golden
    ix = sparse_rx.DenseUint8Index(z["corpus_uint8"], z["corpus_scales"])
    for k_s, res in j["results"].items():
        d, s, n = ix.search(z["query_uint8"], z["query_scales"], int(k_s))
        for i, qid in enumerate(j["qids"]):
            _same_ranking_within([j["doc_ids"][x] for x in d[i, : n[i]]], list(s[i, : n[i]]), res[qid])


@pytest.mark.gpu
def test_dense_uint8_asymmetric_vs_oracle_and_registry():
    rng = np.random.default_rng(77)
    n_docs, dim, nq, k = 5000, 200, 9, 64   # dim not a multiple of 64: rows are zero-padded on both sides
    emb = rng.standard_normal((n_docs, dim)).astype(np.float32)
    qe = rng.standard_normal((nq, dim)).astype(np.float32)
    c8, cs = sparse_rx.quantize_asymmetric(emb)
    qq = [sparse_rx.quantize_query_asymmetric(x) for x in qe]
    q8, qs = np.stack([a for a, _ in qq]), np.stack([b for _, b in qq])
    ix = sparse_rx.DenseUint8Index(c8, cs)
    d, s, n = ix.search(q8, qs, k)
    sims = np_oracle.uint8_asymmetric_similarities(q8, qs, c8, cs)
    ed, es, en = np_oracle.dense_topk(sims, k)
    assert np.array_equal(n, en)
    for i in range(nq):
        _same_ranking_within(list(d[i, : n[i]]), list(s[i, : n[i]]), {int(a): float(b) for a, b in zip(ed[i, : en[i]], es[i, : en[i]])})
    # registry-created retriever with the asymmetric scheme
    corpus = {f"doc{i}": {"text": f"t{i}"} for i in range(300)}
    r = sparse_rx.RetrieverRegistry.create({"type": "dpr", "params": {"embedding_dim": 96, "quantization_method": "asymmetric"}})
    r.build_index_from_corpus(corpus)
    got = r.search({"a": "alpha beta", "b": ""}, top_k=6)
    assert got["b"] == {}
    a8, asc = sparse_rx.quantize_query_asymmetric(r._generate_query_embedding("alpha beta"))
    sims = np_oracle.uint8_asymmetric_similarities(a8[None, :], asc[None, :], r.corpus_embeddings_int8, r.corpus_scales)
    ed, es, en = np_oracle.dense_topk(sims, 6)
    _same_ranking_within(list(got["a"]), list(got["a"].values()), {r.doc_ids[int(a)]: float(b) for a, b in zip(ed[0, : en[0]], es[0, : en[0]])})
