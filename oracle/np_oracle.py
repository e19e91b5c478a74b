"""oracle/np_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy term-at-a-time restatement of ``RetrievalService._numpy_bm25_score``
(/root/reference/rag_system/core/retrieval.py:298-318): for term ids ascending
(``np.nonzero(query_tf)[0]``, :302) walk the term's column and do, per posting, in fp32

    numerator   = tf * (k1 + 1)                                          (:314)
    denominator = tf + k1 * (1 - b + b * doc_len / avgdl)                (:315)
    scores[doc] += idf * (numerator / denominator) * query_weight        (:316)

It is an independent second restatement used to cross-check ``bm25_oracle.c`` (doc-at-a-time) and
to pin the "precomputed impact" factorisation the HIP index stores.  Slow: small cases only.
Parity status: PINNED by tests/golden (scores bit-equal to the imported reference).
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix


def impacts_f32(data, row_of_nnz, doc_lengths, k1, b, avgdl) -> np.ndarray:
    """impact = (tf*(k1+1)) / (tf + k1*(1-b+b*len/avgdl)) in fp32, reference operation order."""
    f = np.float32
    tf = np.asarray(data, dtype=f)
    ln = np.asarray(doc_lengths, dtype=f)[row_of_nnz]
    norm = f(k1) * (f(1.0 - b) + (f(b) * ln) / f(avgdl))
    return (tf * f(k1 + 1.0)) / (tf + norm)


def bm25_scores_taat(indptr, indices, data, doc_lengths, idf, q_term, q_w, k1=1.2, b=0.75, avgdl=1.0) -> np.ndarray:
    n = len(indptr) - 1
    V = len(idf)
    m = csr_matrix((np.asarray(data, np.float32), np.asarray(indices), np.asarray(indptr)), shape=(n, V)).tocsc()
    m.sort_indices()
    scores = np.zeros(n, dtype=np.float32)
    f = np.float32
    order = np.argsort(np.asarray(q_term), kind="stable")
    for i in order:
        t = int(q_term[i])
        if t < 0 or t >= V or not (q_w[i] > 0):
            continue
        lo, hi = m.indptr[t], m.indptr[t + 1]
        docs = m.indices[lo:hi]
        imp = impacts_f32(m.data[lo:hi], docs, doc_lengths, k1, b, avgdl)
        scores[docs] += (f(idf[t]) * imp) * f(q_w[i])  # docs unique within a term: plain fancy += is exact
    return scores


def tfidf_scores_taat(indptr, indices, data, idf, q_term, q_w) -> np.ndarray:
    n = len(indptr) - 1
    V = len(idf)
    m = csr_matrix((np.asarray(data, np.float32), np.asarray(indices), np.asarray(indptr)), shape=(n, V)).tocsc()
    m.sort_indices()
    scores = np.zeros(n, dtype=np.float32)
    f = np.float32
    for i in np.argsort(np.asarray(q_term), kind="stable"):
        t = int(q_term[i])
        if t < 0 or t >= V or not (q_w[i] > 0):
            continue
        lo, hi = m.indptr[t], m.indptr[t + 1]
        scores[m.indices[lo:hi]] += (m.data[lo:hi] * f(idf[t])) * f(q_w[i])
    return scores


def topk_ranked(scores: np.ndarray, k: int):
    """(score desc, index asc) top-k by a full lexsort -- the tie contract stated in bm25_oracle.c."""
    s = np.asarray(scores, dtype=np.float32)
    order = np.lexsort((np.arange(len(s)), -s.astype(np.float64)))
    order = order[: min(k, len(s))]
    return order.astype(np.int64), s[order]


def int8_similarities(queries_int8, corpus_int8, query_scales, corpus_scales):
    """retriever_registry.py:538-548 (the NumPy twin of quantized_dot_product_batch :90-117, symmetric scheme):
    similarities[q, d] = f32( int32 dot * query_scale (f32) * doc_scale (f32) ) with NumPy scalar promotion
    (int32 * float32 -> float64, * float32 -> float64, stored into a float32 array)."""
    q = np.asarray(queries_int8).astype(np.int32)
    c = np.asarray(corpus_int8).astype(np.int32)
    dots = q @ c.T  # exact: |dot| <= dim * 127 * 127 < 2^31
    qs = np.asarray(query_scales, dtype=np.float32).astype(np.float64)[:, None]
    cs = np.asarray(corpus_scales, dtype=np.float32).astype(np.float64)[None, :]
    return ((dots.astype(np.float64) * qs) * cs).astype(np.float32)


def dense_topk(similarities, k):
    """Ranked top-k of every row with the engine's contract: score > 0 only (retriever_registry.py:515-519),
    (score desc, doc asc), padded with -1 / 0."""
    sims = np.asarray(similarities, dtype=np.float32)
    nq = sims.shape[0]
    out_d = np.full((nq, k), -1, np.int32)
    out_s = np.zeros((nq, k), np.float32)
    out_n = np.zeros(nq, np.int32)
    for i in range(nq):
        d, s = topk_ranked(sims[i], k)
        keep = s > 0
        d, s = d[keep], s[keep]
        out_d[i, : len(d)] = d
        out_s[i, : len(d)] = s
        out_n[i] = len(d)
    return out_d, out_s, out_n


def f32_similarities(embeddings, queries):
    """retrieval.py:411: similarities = np.dot(self.embedding_index, query_vector), one column per query.  Evaluated in
    float64 and rounded: the reference's fp32 BLAS matvec differs from it by its (unspecified) summation order only."""
    e = np.asarray(embeddings, dtype=np.float32).astype(np.float64)
    q = np.asarray(queries, dtype=np.float32).astype(np.float64)
    return (q @ e.T).astype(np.float32)


def uint8_asymmetric_similarities(queries_uint8, query_scales, corpus_uint8, corpus_scales):
    """retriever_registry.py:550-559 (asymmetric scheme of _numpy_quantized_similarity): for doc i the reader takes
    doc_scale = corpus_scales[2 i], doc_min = corpus_scales[2 i + 1] from the table the writer stored as all scales then
    all mins (:459) -- restated as the reader does it; query_fp32 = u8 * query_scale + query_min and
    doc_fp32 = u8 * doc_scale + doc_min in fp32, similarities[i] = np.dot(query_fp32, doc_fp32).  The dot is evaluated in
    float64 and rounded: the reference's fp32 BLAS dot differs by its (unspecified) summation order only."""
    cs = np.asarray(corpus_scales, dtype=np.float32).reshape(-1)
    c = np.asarray(corpus_uint8)
    n = c.shape[0]
    doc_scale, doc_min = cs[0:2 * n:2], cs[1:2 * n:2]
    doc = c.astype(np.float32) * doc_scale[:, None] + doc_min[:, None]
    qs = np.asarray(query_scales, dtype=np.float32).reshape(-1, 2)
    q = np.asarray(queries_uint8).astype(np.float32) * qs[:, 0:1] + qs[:, 1:2]
    return (q.astype(np.float64) @ doc.astype(np.float64).T).astype(np.float32)
