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
