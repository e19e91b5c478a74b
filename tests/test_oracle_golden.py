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
