/*
 * oracle/bm25_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, optional OpenMP) of the reference's sparse scoring + top-k hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (optimized-sparse-retrieval-..._amd/) never imports, links or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against outputs of the
 * reference itself (imported in the build container, NumPy fallback because numba is not
 * installed) through the committed fixtures in tests/golden/ (generator:
 * tests/golden/make_golden.py).
 *
 * Reference functions restated (paths relative to /root/reference):
 *   simd_bm25_score        rag_system/core/retrieval.py:41-76
 *   simd_tfidf_score       rag_system/pipeline/evaluate_rag_pipeline.py:95-121
 *   fast_topk_selection    rag_system/core/retrieval.py:79-92 (and its inline NumPy twin 276-284)
 *   result filtering       rag_system/core/retrieval.py:292-296  (score > 0 only)
 *
 * Numerics: the runnable reference evaluates everything in fp32 (NumPy-2 weak Python scalars):
 *   norm  = f32(k1) * (f32(1-b) + (f32(b)*len)/f32(avgdl))        retrieval.py:58
 *   num   = tf * f32(k1+1)                                         retrieval.py:70
 *   den   = tf + norm                                              retrieval.py:71
 *   score += (idf * (num/den)) * qw   (terms ascending = CSR row order)   retrieval.py:72
 * The *_f64 variant restates what the Numba-compiled path would compute (fp64 intermediates,
 * cast to fp32 on store, retrieval.py:74) -- diagnostic only, it can never be run against the
 * reference here because numba is absent.
 *
 * Build:  gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile)
 * -ffp-contract=off matters: x86-64 gcc would otherwise be free to fuse a*b+c on FMA targets.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Dense query_tf[vocab] built from the sparse (term, weight) list, exactly what
 * retrieval.py:241-249 hands to the kernel. */
static float *dense_query(const int32_t *q_term, const float *q_w, int nt, int64_t vocab) {
    float *qtf = (float *)calloc((size_t)(vocab > 0 ? vocab : 1), sizeof(float));
    if (!qtf) return NULL;
    for (int i = 0; i < nt; ++i)
        if (q_term[i] >= 0 && q_term[i] < vocab) qtf[q_term[i]] = q_w[i];
    return qtf;
}

/* simd_bm25_score, fp32 evaluation (what the importable reference computes).
 * retrieval.py:41-76: prange over docs, serial loop over the row's nnz in CSR order. */
ORACLE_API int oracle_bm25_scores_f32(int64_t n_docs, int64_t vocab, const int64_t *indptr,
                                      const int32_t *indices, const float *data,
                                      const float *doc_lengths, const float *idf,
                                      const int32_t *q_term, const float *q_w, int nt, double k1,
                                      double b, double avgdl, float *scores) {
    float *qtf = dense_query(q_term, q_w, nt, vocab);
    if (!qtf) return -1;
    const float k1f = (float)k1, bf = (float)b, omb = (float)(1.0 - b), k1p1 = (float)(k1 + 1.0),
                avf = (float)avgdl;
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n_docs; ++d) {
        float s = 0.0f;
        const float len = doc_lengths[d];
        const float norm = k1f * (omb + (bf * len) / avf); /* retrieval.py:58 */
        for (int64_t j = indptr[d]; j < indptr[d + 1]; ++j) {
            const int32_t t = indices[j];
            if (t < vocab && qtf[t] > 0.0f) { /* retrieval.py:67 */
                const float tf = data[j];
                const float num = tf * k1p1;  /* :70 */
                const float den = tf + norm;  /* :71 */
                s += (idf[t] * (num / den)) * qtf[t]; /* :72 */
            }
        }
        scores[d] = s;
    }
    free(qtf);
    return 0;
}

/* Numba-faithful restatement: k1, b, avgdl, 1.0 and the accumulator are float64 in Numba's
 * typing, the result is cast on store (retrieval.py:74).  Diagnostic only. */
ORACLE_API int oracle_bm25_scores_f64(int64_t n_docs, int64_t vocab, const int64_t *indptr,
                                      const int32_t *indices, const float *data,
                                      const float *doc_lengths, const float *idf,
                                      const int32_t *q_term, const float *q_w, int nt, double k1,
                                      double b, double avgdl, float *scores) {
    float *qtf = dense_query(q_term, q_w, nt, vocab);
    if (!qtf) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n_docs; ++d) {
        double s = 0.0;
        const double norm = k1 * (1.0 - b + b * (double)doc_lengths[d] / avgdl);
        for (int64_t j = indptr[d]; j < indptr[d + 1]; ++j) {
            const int32_t t = indices[j];
            if (t < vocab && qtf[t] > 0.0f) {
                const double tf = (double)data[j];
                s += (double)idf[t] * ((tf * (k1 + 1.0)) / (tf + norm)) * (double)qtf[t];
            }
        }
        scores[d] = (float)s;
    }
    free(qtf);
    return 0;
}

/* simd_tfidf_score (evaluate_rag_pipeline.py:95-121): contribution tf*idf*qw, fp32. */
ORACLE_API int oracle_tfidf_scores_f32(int64_t n_docs, int64_t vocab, const int64_t *indptr,
                                       const int32_t *indices, const float *data, const float *idf,
                                       const int32_t *q_term, const float *q_w, int nt,
                                       float *scores) {
    float *qtf = dense_query(q_term, q_w, nt, vocab);
    if (!qtf) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n_docs; ++d) {
        float s = 0.0f;
        for (int64_t j = indptr[d]; j < indptr[d + 1]; ++j) {
            const int32_t t = indices[j];
            if (t < vocab && qtf[t] > 0.0f) s += (data[j] * idf[t]) * qtf[t]; /* :117 */
        }
        scores[d] = s;
    }
    free(qtf);
    return 0;
}

/* The pipeline twin's NumPy fallback (evaluate_rag_pipeline.py:436-479, `_numpy_score_documents`): term-at-a-time
 * over `relevant_terms`, which is in QUERY-TOKEN order (first occurrence, :360-370), `scores[docs] += term_scores`.
 * Per doc that is a sequential fp32 sum of its contributions in the GIVEN term order (not ascending term id), so it
 * can differ from simd_bm25_score in the last bit.  Restated doc-at-a-time: for every doc, the query's terms in the
 * given order, each looked up in the (sorted) CSR row.  tfidf != 0: contribution (tf*idf)*qw (:472), else BM25
 * (idf*(num/den))*qw (:455-457, same fp32 operations as retrieval.py:314-316). */
ORACLE_API int oracle_scores_given_order(int tfidf, int64_t n_docs, int64_t vocab, const int64_t *indptr,
                                         const int32_t *indices, const float *data, const float *doc_lengths,
                                         const float *idf, const int32_t *q_term, const float *q_w, int nt,
                                         double k1, double b, double avgdl, float *scores) {
    const float k1f = (float)k1, bf = (float)b, omb = (float)(1.0 - b), k1p1 = (float)(k1 + 1.0),
                avf = (float)avgdl;
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n_docs; ++d) {
        float s = 0.0f;
        const float norm = tfidf ? 0.0f : k1f * (omb + (bf * doc_lengths[d]) / avf);
        const int64_t lo0 = indptr[d], hi0 = indptr[d + 1];
        for (int i = 0; i < nt; ++i) {
            const int32_t t = q_term[i];
            if (t < 0 || t >= vocab) continue;
            int64_t lo = lo0, hi = hi0;
            while (lo < hi) { /* lower_bound in the sorted row */
                const int64_t mid = (lo + hi) >> 1;
                if (indices[mid] < t) lo = mid + 1; else hi = mid;
            }
            if (lo < hi0 && indices[lo] == t) {
                const float tf = data[lo];
                if (tfidf)
                    s += (tf * idf[t]) * q_w[i];
                else
                    s += (idf[t] * ((tf * k1p1) / (tf + norm))) * q_w[i];
            }
        }
        scores[d] = s;
    }
    return 0;
}

/* ---- fast_topk_selection (retrieval.py:79-92) ------------------------------------------
 * The reference partitions with argpartition(-scores, k) and sorts the k survivors with
 * argsort; the order of equal scores is whatever introselect/quicksort leave behind, i.e.
 * unspecified.  This restatement fixes it: rank order is (score descending, index ascending).
 * The selected SET differs from the reference only inside a group of exactly equal scores that
 * straddles rank k, and the ORDER only inside groups of exactly equal scores; the fixtures pin
 * everything outside those groups.  If k >= n every element is returned (full argsort branch,
 * retrieval.py:83-85 / 281-284). */
typedef struct {
    float s;
    int64_t i;
} pair_t;

static inline int better(const pair_t *a, const pair_t *b) { /* a ranks before b */
    return (a->s > b->s) || (a->s == b->s && a->i < b->i);
}

static void sift_down(pair_t *h, int64_t n, int64_t r) { /* min-heap on "better": root = worst */
    for (;;) {
        int64_t c = 2 * r + 1;
        if (c >= n) return;
        if (c + 1 < n && better(&h[c], &h[c + 1])) c = c + 1; /* pick the worse child */
        if (better(&h[r], &h[c])) {
            pair_t t = h[r];
            h[r] = h[c];
            h[c] = t;
            r = c;
        } else
            return;
    }
}

static int cmp_rank(const void *pa, const void *pb) {
    const pair_t *a = (const pair_t *)pa, *b = (const pair_t *)pb;
    if (better(a, b)) return -1;
    if (better(b, a)) return 1;
    return 0;
}

/* out_idx/out_scores hold min(k, n) entries; returns that count. */
ORACLE_API int64_t oracle_topk(const float *scores, int64_t n, int64_t k, int64_t *out_idx,
                               float *out_scores) {
    if (k > n) k = n;
    if (k <= 0) return 0;
    pair_t *h = (pair_t *)malloc((size_t)k * sizeof(pair_t));
    if (!h) return -1;
    for (int64_t i = 0; i < k; ++i) {
        h[i].s = scores[i];
        h[i].i = i;
    }
    for (int64_t r = k / 2 - 1; r >= 0; --r) sift_down(h, k, r);
    for (int64_t i = k; i < n; ++i) {
        pair_t c = {scores[i], i};
        if (better(&c, &h[0])) {
            h[0] = c;
            sift_down(h, k, 0);
        }
    }
    qsort(h, (size_t)k, sizeof(pair_t), cmp_rank);
    for (int64_t i = 0; i < k; ++i) {
        out_idx[i] = h[i].i;
        out_scores[i] = h[i].s;
    }
    free(h);
    return k;
}

/* ---- batched driver: what search_bm25 does per query (retrieval.py:203-296) minus the
 * Python-side tokenisation, cache and doc-id mapping.  mode 0 = BM25 fp32, 1 = tf-idf dot fp32,
 * 2 = BM25 Numba-like fp64, 3 / 4 = BM25 / tf-idf fp32 accumulated in the GIVEN term order (pipeline twin's
 * NumPy fallback).  Outputs are [nq, k], rank ordered, entries with score <= 0 dropped
 * (retrieval.py:295), padded with doc -1 / score 0.  This is also the timed CPU baseline: it
 * keeps the reference's cost model (a full CSR scan per query, prange over docs). */
ORACLE_API int oracle_search_batch(int mode, int64_t n_docs, int64_t vocab, const int64_t *indptr,
                                   const int32_t *indices, const float *data,
                                   const float *doc_lengths, const float *idf, double k1, double b,
                                   double avgdl, const int32_t *q_ptr, const int32_t *q_term,
                                   const float *q_w, int nq, int k, int32_t *out_doc,
                                   float *out_score, int32_t *out_count) {
    float *scores = (float *)malloc((size_t)(n_docs > 0 ? n_docs : 1) * sizeof(float));
    int64_t *idx = (int64_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(int64_t));
    float *sc = (float *)malloc((size_t)(k > 0 ? k : 1) * sizeof(float));
    if (!scores || !idx || !sc) return -1;
    for (int q = 0; q < nq; ++q) {
        const int32_t *qt = q_term + q_ptr[q];
        const float *qw = q_w + q_ptr[q];
        const int nt = q_ptr[q + 1] - q_ptr[q];
        int32_t *od = out_doc + (int64_t)q * k;
        float *os = out_score + (int64_t)q * k;
        for (int j = 0; j < k; ++j) {
            od[j] = -1;
            os[j] = 0.0f;
        }
        out_count[q] = 0;
        if (nt == 0) continue; /* retrieval.py:251-252 */
        int rc;
        if (mode == 0)
            rc = oracle_bm25_scores_f32(n_docs, vocab, indptr, indices, data, doc_lengths, idf, qt,
                                        qw, nt, k1, b, avgdl, scores);
        else if (mode == 1)
            rc = oracle_tfidf_scores_f32(n_docs, vocab, indptr, indices, data, idf, qt, qw, nt,
                                         scores);
        else if (mode == 3 || mode == 4)
            rc = oracle_scores_given_order(mode == 4, n_docs, vocab, indptr, indices, data, doc_lengths, idf, qt, qw,
                                           nt, k1, b, avgdl, scores);
        else
            rc = oracle_bm25_scores_f64(n_docs, vocab, indptr, indices, data, doc_lengths, idf, qt,
                                        qw, nt, k1, b, avgdl, scores);
        if (rc) return rc;
        int64_t m = oracle_topk(scores, n_docs, k, idx, sc);
        if (m < 0) return -1;
        int c = 0;
        for (int64_t j = 0; j < m; ++j)
            if (sc[j] > 0.0f) { /* retrieval.py:295 */
                od[c] = (int32_t)idx[j];
                os[c] = sc[j];
                ++c;
            }
        out_count[q] = c;
    }
    free(scores);
    free(idx);
    free(sc);
    return 0;
}
