#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference is mounted read-only at /root/reference; it does not
exist on the GPU box and nothing under tests/ reads it at test time):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

What is recorded is DATA only -- inputs (texts, CSR arrays, query term lists) and the outputs the
reference produced for them:
  text_small.json      corpus + queries + ``RetrievalService.search_bm25`` results (k = 3, 10, 1000)
  text_small.npz       the index state ``build_bm25_index`` produced (CSR, idf, doc_lengths, avgdl,
                       vocabulary, doc_ids) + full per-query score vectors from ``_numpy_bm25_score``
                       and from the plain-Python ``simd_bm25_score``
  csr_zipf.npz         numeric Zipf CSR pushed through the real ``search_bm25`` (k = 10, 100) and
                       through ``simd_tfidf_score`` (pipeline twin) with its idf variant
  registry_small.json  ``OptimizedBM25Retriever`` (registry twin) results on the same text corpus,
                       bm25 and the registry's tfidf setting (k1=1000, b=0)
  pipeline_small.json  ``OptimizedRetriever`` (pipeline twin, evaluate_rag_pipeline.py:162-479) ``search`` results on the
                       same text corpus for bm25, bm25_custom (k1=1.6, b=0.8), splade and dpr types (the last two score
                       with the tf-idf dot product), k = 5 and 50, fresh build and cache-hit build
  pipeline_small.npz   per-query full score vectors of its ``_numpy_score_documents`` (accumulation in QUERY-TOKEN order,
                       :436-479) for the bm25 and splade types, with the token-ordered (term, weight) lists
  ref_cache/*.npz      the ``.rag_cache/{method}_index_{hash}.npz`` files the reference itself wrote
                       (``_save_cached_index``, :280-296) for the bm25 and splade types -- data files, read with
                       ``allow_pickle=False``
  text_deep.npz        2600 docs through ``search_bm25`` at top_k = 1500 and 5000 (>= n_docs): rankings deeper than the
                       engine's 1024-row lists (run with the argument ``deep`` to refresh only this one)
  dense_uint8_asym.npz/.json  the same retriever with quantization_method="asymmetric" (uint8 + scale / min table)
  dense_int8.npz/.json ``QuantizedEmbeddingRetriever`` (symmetric INT8): quantized corpus, per-query similarity rows
                       and ``search`` results for recorded embeddings (run with the argument ``dense`` to refresh
                       only these)
  dense_synth.npz      its simulated corpus / query embedding generators (seeded NumPy streams)

numba is not installed here, so the reference runs its own NumPy / plain-Python fallbacks
(NUMBA_AVAILABLE=False, retrieval.py:22-33).
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import rag_system.core.retrieval as ref_retrieval  # noqa: E402
from rag_system.core.memory_index import MemoryIndex  # noqa: E402

assert ref_retrieval.NUMBA_AVAILABLE is False


def new_service(tmp):
    p = os.path.join(tmp, "idx.bin")
    MemoryIndex(p, create=True).close()
    return ref_retrieval.RetrievalService(p)


def zipf_text_corpus(rng, n_docs, vocab_words):
    """Text in the style of tests/core_test.py:203-252 (Zipf words, gamma lengths)."""
    V = len(vocab_words)
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    corpus = {}
    for i in range(n_docs):
        ln = int(np.clip(rng.gamma(2.0, 12.0), 3, 80))
        words = rng.choice(V, size=ln, p=p)
        corpus[f"doc{i}"] = {"text": " ".join(vocab_words[w] for w in words), "title": f"Title {i}"}
    return corpus


def make_text_fixture(tmp):
    rng = np.random.default_rng(20250)
    vocab_words = [f"w{i}" for i in range(400)] + ["café", "naïve", "straße", "東京", "данные", "x_1", "e2e"]
    corpus = zipf_text_corpus(rng, 300, vocab_words)
    # edge-case documents (retrieval.py:145: text -> content -> body fallback; title never indexed)
    corpus["edge_content"] = {"content": "Don't panic: the U.S.A. costs 3.14 dollars; CAFÉ café Café!"}
    corpus["edge_body"] = {"body": "İstanbul straße 東京 東京 данные x_1 e2e w1 w1 w1 w2"}
    corpus["edge_empty"] = {"text": ""}
    corpus["edge_title_only"] = {"title": "w1 w2 w3 only a title"}
    corpus["edge_punct"] = {"text": "... !!! ???"}
    corpus["edge_dup_a"] = {"text": "w7 w8 w9 w7"}
    corpus["edge_dup_b"] = {"text": "w7 w8 w9 w7"}  # exact score tie with edge_dup_a
    corpus["edge_long"] = {"text": " ".join(["w3 w4 w5"] * 60)}
    svc = new_service(tmp)
    svc.build_bm25_index(corpus)

    queries = {}
    for i in range(30):
        ln = int(np.clip(rng.gamma(1.5, 2.5), 1, 10))
        words = rng.choice(120, size=ln)  # from the frequent part of the vocabulary
        queries[f"q{i}"] = " ".join(vocab_words[w] for w in words)
    queries.update({
        "q_blank": "   ",
        "q_empty": "",
        "q_oov": "zzzunknown qqqmissing",
        "q_punct": "?!",
        "q_negidf": "w0",                  # df > N/2 -> negative idf -> no positive score
        "q_negidf_mix": "w0 w0 w250 w399",
        "q_repeat": "w7 w7 w7 w8",
        "q_tie": "w9 w8 w7",
        "q_unicode": "CAFÉ 東京 Данные straße",
        "q_apostrophe": "don't u.s.a. 3.14",
        "q_mixed_oov": "w5 notaword w6",
        "q_long": " ".join(f"w{i}" for i in range(0, 140, 2)),  # 70 distinct terms
    })
    results = {}
    for k in (3, 10, 1000):
        svc.clear_cache()
        results[str(k)] = svc.search_bm25(queries, top_k=k)
    # cache-hit path (retrieval.py:216-225) must give the same dict
    again = svc.search_bm25(queries, top_k=1000)
    assert again == results["1000"]

    # full score vectors
    score_q = [q for q in queries if q.startswith("q") and queries[q].strip()]
    full_numpy, full_simd, qterms, qweights, qnames = [], [], [], [], []
    import re
    from collections import Counter
    for qid in score_q:
        toks = re.findall(r"\b\w+\b", queries[qid].lower())
        cnt = Counter(toks)
        qtf = np.zeros(len(svc.vocabulary), dtype=np.float32)
        for t, c in cnt.items():
            if t in svc.vocabulary:
                qtf[svc.vocabulary[t]] = float(c)
        if not qtf.any():
            continue
        s_np = svc._numpy_bm25_score(qtf)
        s_simd = ref_retrieval.simd_bm25_score(qtf, svc.corpus_tf.data, svc.corpus_tf.indices, svc.corpus_tf.indptr,
                                               svc.doc_lengths, svc.idf_weights, svc.k1, svc.b, svc.avgdl)
        assert np.array_equal(s_np, s_simd), qid
        nz = np.nonzero(qtf)[0]
        qnames.append(qid)
        qterms.append(nz.astype(np.int32))
        qweights.append(qtf[nz])
        full_numpy.append(s_np.astype(np.float32))
    q_ptr = np.zeros(len(qterms) + 1, dtype=np.int32)
    q_ptr[1:] = np.cumsum([len(t) for t in qterms])
    vocab_sorted = sorted(svc.vocabulary, key=svc.vocabulary.get)
    np.savez_compressed(
        os.path.join(OUT, "text_small.npz"),
        tf_data=svc.corpus_tf.data, tf_indices=svc.corpus_tf.indices, tf_indptr=svc.corpus_tf.indptr,
        tf_shape=np.array(svc.corpus_tf.shape, dtype=np.int64), doc_lengths=svc.doc_lengths, idf=svc.idf_weights,
        avgdl=np.float64(svc.avgdl), k1=np.float64(svc.k1), b=np.float64(svc.b),
        vocabulary=np.array(vocab_sorted), doc_ids=np.array(svc.doc_ids),
        score_qids=np.array(qnames), score_q_ptr=q_ptr, score_q_term=np.concatenate(qterms),
        score_q_weight=np.concatenate(qweights), full_scores=np.stack(full_numpy),
    )
    with open(os.path.join(OUT, "text_small.json"), "w", encoding="utf-8") as f:
        json.dump({"corpus": corpus, "queries": queries, "results": results,
                   "stats": {k: v for k, v in svc.get_stats().items()}}, f, ensure_ascii=False, indent=0)
    svc.close()
    return corpus, queries


def make_csr_fixture(tmp):
    """Numeric CSR (no text): Zipf term ids, pushed through the REAL search_bm25 by giving the service
    a vocabulary of synthetic tokens t<id> (zero-padded so that sorted order == id order)."""
    rng = np.random.default_rng(20251)
    n_docs, V = 3000, 600
    p = 1.0 / np.arange(1, V + 1) ** 1.0
    p /= p.sum()
    rows, cols, vals = [], [], []
    doc_len = np.zeros(n_docs, dtype=np.float32)
    for d in range(n_docs):
        ln = int(np.clip(rng.gamma(2.0, 15.0), 2, 120))
        terms, counts = np.unique(rng.choice(V, size=ln, p=p), return_counts=True)
        rows += [d] * len(terms)
        cols += terms.tolist()
        vals += counts.astype(np.float32).tolist()
        doc_len[d] = ln
    from scipy.sparse import csr_matrix
    m = csr_matrix((np.array(vals, np.float32), (rows, cols)), shape=(n_docs, V), dtype=np.float32)
    m.sort_indices()
    svc = new_service(tmp)
    svc.corpus_tf = m
    svc.vocabulary = {f"t{i:04d}": i for i in range(V)}
    svc.doc_ids = [str(i) for i in range(n_docs)]
    svc.doc_lengths = doc_len
    df = np.bincount(m.indices, minlength=V)
    svc.idf_weights = np.log((n_docs - df + 0.5) / (df + 0.5)).astype(np.float32)  # retrieval.py:189
    svc.avgdl = float(np.mean(doc_len))                                              # retrieval.py:190
    nq = 48
    qterms, qweights, texts = [], [], {}
    for q in range(nq):
        nt = int(rng.integers(1, 12))
        ts = rng.choice(V, size=nt, p=p)
        terms, counts = np.unique(ts, return_counts=True)
        qterms.append(terms.astype(np.int32))
        qweights.append(counts.astype(np.float32))
        texts[f"{q}"] = " ".join(f"t{t:04d}" for t in ts)
    res = {}
    for k in (10, 100):
        svc.clear_cache()
        out = svc.search_bm25(texts, top_k=k)
        docs = np.full((nq, k), -1, dtype=np.int32)
        scs = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.int32)
        for q in range(nq):
            items = list(out[f"{q}"].items())
            cnt[q] = len(items)
            for j, (d, s) in enumerate(items):
                docs[q, j] = int(d)
                scs[q, j] = np.float32(s)
        res[k] = (docs, scs, cnt)
    # tf-idf twin (pipeline copy): simd_tfidf_score + idf = log(N/(df+1)) (evaluate_rag_pipeline.py:95-121, 273-278)
    cwd = os.getcwd()
    os.chdir(tmp)  # the pipeline module writes .rag_cache/ into the CWD
    with contextlib.redirect_stdout(io.StringIO()):
        import rag_system.pipeline.evaluate_rag_pipeline as ref_pipe
    os.chdir(cwd)
    idf_tfidf = np.log(n_docs / (df + 1)).astype(np.float32)
    tfidf_full = []
    for q in range(16):
        qtf = np.zeros(V, dtype=np.float32)
        qtf[qterms[q]] = qweights[q]
        tfidf_full.append(np.asarray(ref_pipe.simd_tfidf_score(qtf, m.data, m.indices, m.indptr, idf_tfidf),
                                     dtype=np.float32))
    bm25_full = []
    for q in range(16):
        qtf = np.zeros(V, dtype=np.float32)
        qtf[qterms[q]] = qweights[q]
        bm25_full.append(svc._numpy_bm25_score(qtf))
    q_ptr = np.zeros(nq + 1, dtype=np.int32)
    q_ptr[1:] = np.cumsum([len(t) for t in qterms])
    np.savez_compressed(
        os.path.join(OUT, "csr_zipf.npz"),
        tf_data=m.data, tf_indices=m.indices, tf_indptr=m.indptr, tf_shape=np.array(m.shape, dtype=np.int64),
        doc_lengths=doc_len, idf=svc.idf_weights, idf_tfidf=idf_tfidf, avgdl=np.float64(svc.avgdl),
        k1=np.float64(svc.k1), b=np.float64(svc.b),
        q_ptr=q_ptr, q_term=np.concatenate(qterms), q_weight=np.concatenate(qweights),
        top10_doc=res[10][0], top10_score=res[10][1], top10_count=res[10][2],
        top100_doc=res[100][0], top100_score=res[100][1], top100_count=res[100][2],
        bm25_full=np.stack(bm25_full), tfidf_full=np.stack(tfidf_full),
    )
    svc.close()


def make_registry_fixture(tmp, corpus, queries):
    cwd = os.getcwd()
    os.chdir(tmp)
    with contextlib.redirect_stdout(io.StringIO()):
        import rag_system.core.retriever_registry as ref_reg
        out = {}
        for name, cfg in (("bm25", {"type": "bm25", "params": {"k1": 1.2, "b": 0.75}}),
                          ("bm25_msmarco", {"type": "bm25_custom", "params": {"k1": 1.6, "b": 0.8}}),
                          ("tfidf", {"type": "tfidf"})):
            r = ref_reg.RetrieverRegistry.create(cfg)
            r.build_index_from_corpus(corpus)
            out[name] = {"k1": r.k1, "b": r.b, "results": {str(k): r.search(queries, top_k=k) for k in (5, 50)}}
    os.chdir(cwd)
    with open(os.path.join(OUT, "registry_small.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)


def make_pipeline_fixture(tmp, corpus, queries):
    """The pipeline twin: results, token-ordered score vectors, and the index caches it writes."""
    import re
    import shutil
    from collections import Counter
    cwd = os.getcwd()
    os.chdir(tmp)  # the module creates .rag_cache/ in the CWD
    hw = {"memory_gb": 8, "cores": 4}  # its own hard-coded fallback (evaluate_rag_pipeline.py:50-53)
    out, full = {}, {}
    with contextlib.redirect_stdout(io.StringIO()):
        import rag_system.pipeline.evaluate_rag_pipeline as ref_pipe
        assert ref_pipe.NUMBA_AVAILABLE is False
        for name, cfg in (("bm25", {"type": "bm25", "params": {"k1": 1.2, "b": 0.75}}),
                          ("bm25_custom", {"type": "bm25_custom", "params": {"k1": 1.6, "b": 0.8}}),
                          ("splade", {"type": "splade"}),
                          ("dpr", {"type": "dpr"})):
            r = ref_pipe.OptimizedRetriever(cfg, hw)
            r.build_index_from_corpus(corpus)       # fresh build, writes the cache
            res = {str(k): r.search(queries, top_k=k) for k in (5, 50)}
            r2 = ref_pipe.OptimizedRetriever(cfg, hw)
            r2.build_index_from_corpus(corpus)      # loads the cache the first one wrote (:194-196)
            res_cached = {str(k): r2.search(queries, top_k=k) for k in (5, 50)}
            assert res_cached == res, name
            out[name] = {"k1": r.k1, "b": r.b, "results": res}
            if name in ("bm25", "splade"):
                qn, qt, qw, fs = [], [], [], []
                for qid, text in queries.items():
                    toks = re.findall(r"\b\w+\b", text.lower()) if text else []
                    cnt = Counter(toks)
                    rel = [r.vocabulary[t] for t in cnt if t in r.vocabulary]   # token (first-occurrence) order, :360-370
                    if not rel:
                        continue
                    qtf = np.zeros(len(r.vocabulary), dtype=np.float32)
                    for t, c in cnt.items():
                        if t in r.vocabulary:
                            qtf[r.vocabulary[t]] = c
                    qn.append(qid)
                    qt.append(np.array(rel, dtype=np.int32))
                    qw.append(qtf[rel])
                    fs.append(np.asarray(r._numpy_score_documents(qtf, rel), dtype=np.float32))
                ptr = np.zeros(len(qt) + 1, dtype=np.int32)
                ptr[1:] = np.cumsum([len(t) for t in qt])
                full[name] = dict(qids=np.array(qn), q_ptr=ptr, q_term=np.concatenate(qt), q_weight=np.concatenate(qw),
                                  full_scores=np.stack(fs), idf=r.idf, avgdl=np.float64(r.avgdl))
    os.makedirs(os.path.join(OUT, "ref_cache"), exist_ok=True)
    for fn in sorted(os.listdir(".rag_cache")):
        if fn.startswith(("bm25_index_", "splade_index_")):
            shutil.copy(os.path.join(".rag_cache", fn), os.path.join(OUT, "ref_cache", fn))
    os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "pipeline_small.npz"),
                        **{f"{n}_{k}": v for n, d in full.items() for k, v in d.items()})
    with open(os.path.join(OUT, "pipeline_small.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)


def make_dense_fixture():
    """dense_int8.npz: embeddings pushed through the reference's QuantizedEmbeddingRetriever (symmetric INT8):
    ``_quantize_embeddings`` -> corpus_int8 / corpus_scales, and the real ``search`` (query quantization, NumPy
    similarity twin of quantized_dot_product_batch, top-k, score > 0 filter) with the simulated query-embedding
    generator replaced by the recorded query embeddings."""
    with contextlib.redirect_stdout(io.StringIO()):
        import rag_system.core.retriever_registry as ref_reg
        rng = np.random.default_rng(4242)
        n_docs, dim, nq = 300, 48, 12
        emb = rng.standard_normal((n_docs, dim)).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        emb[7] = emb[3]            # exact duplicates: score ties
        emb[11] = 0.0              # an all-zero row (scale clamps to 1e-8)
        qemb = rng.standard_normal((nq, dim)).astype(np.float32)
        qemb /= np.linalg.norm(qemb, axis=1, keepdims=True)
        qemb[2] = emb[3] * 0.5     # a query parallel to the duplicated docs
        r = ref_reg.QuantizedEmbeddingRetriever("dpr", "fixture", embedding_dim=dim)
        r.doc_ids = [f"d{i}" for i in range(n_docs)]
        r.corpus_embeddings_int8, r.corpus_scales = r._quantize_embeddings(emb)
        qtexts = {f"q{i}": f"query {i}" for i in range(nq)}
        lookup = {t: qemb[i] for i, t in enumerate(qtexts.values())}
        r._generate_query_embedding = lambda text: lookup[text]
        sims, qi8, qsc = [], [], []
        for i in range(nq):  # the similarity rows the search ranks (same calls as retriever_registry.py:482-503)
            s = np.max(np.abs(qemb[i]))
            q8 = np.round(qemb[i] / s * 127.0).astype(np.int8)
            qs = np.array([s / 127.0], dtype=np.float32)
            sims.append(r._numpy_quantized_similarity(q8, qs).copy())
            qi8.append(q8)
            qsc.append(qs[0])
        results = {str(k): r.search(qtexts, top_k=k) for k in (5, 20)}
        # the simulated embedding generators (retriever_registry.py:409-433, 526-536); hash(text) is process-dependent, so
        # the seed the reference derived from it in THIS process is recorded next to the vector
        r2 = ref_reg.QuantizedEmbeddingRetriever("dpr", "fixture", embedding_dim=24)
        syn = r2._generate_synthetic_embeddings(137)
        qtext = "what is sparse retrieval"
        qseed = hash(qtext) % (2 ** 31)
        qsyn = r2._generate_query_embedding(qtext)
        # the asymmetric (uint8) scheme: same embeddings through the reference's other branch (:449-462, 486-491, 550-559)
        ra = ref_reg.QuantizedEmbeddingRetriever("dpr", "fixture", embedding_dim=dim, quantization_method="asymmetric")
        ra.doc_ids = list(r.doc_ids)
        ra.corpus_embeddings_int8, ra.corpus_scales = ra._quantize_embeddings(emb)
        ra._generate_query_embedding = lambda text: lookup[text]
        asims, aq8, aqs = [], [], []
        for i in range(nq):
            qmin, qmax = np.min(qemb[i]), np.max(qemb[i])
            qscale = (qmax - qmin) / 255.0
            q8 = np.round((qemb[i] - qmin) / qscale).astype(np.uint8)
            qsc2 = np.array([qscale, qmin], dtype=np.float32)
            asims.append(ra._numpy_quantized_similarity(q8, qsc2).copy())
            aq8.append(q8)
            aqs.append(qsc2)
        aresults = {str(k): ra.search(qtexts, top_k=k) for k in (5, 20)}
    np.savez_compressed(os.path.join(OUT, "dense_uint8_asym.npz"), corpus_uint8=ra.corpus_embeddings_int8,
                        corpus_scales=ra.corpus_scales, query_uint8=np.stack(aq8), query_scales=np.stack(aqs),
                        similarities=np.stack(asims))
    with open(os.path.join(OUT, "dense_uint8_asym.json"), "w", encoding="utf-8") as f:
        json.dump({"doc_ids": ra.doc_ids, "qids": list(qtexts), "results": aresults}, f, indent=0)
    np.savez_compressed(os.path.join(OUT, "dense_synth.npz"), synthetic_137x24=syn, query_seed=np.int64(qseed), query_24=qsyn)
    np.savez_compressed(os.path.join(OUT, "dense_int8.npz"), emb=emb, qemb=qemb, corpus_int8=r.corpus_embeddings_int8,
                        corpus_scales=r.corpus_scales, query_int8=np.stack(qi8), query_scales=np.array(qsc, dtype=np.float32),
                        similarities=np.stack(sims))
    with open(os.path.join(OUT, "dense_int8.json"), "w", encoding="utf-8") as f:
        json.dump({"doc_ids": r.doc_ids, "qids": list(qtexts), "results": results}, f, indent=0)


def make_deep_fixture(tmp):
    """text_deep.npz: a corpus of MORE than 1024 docs through the reference's ``search_bm25`` at top_k beyond 1024 --
    k = 1500 (argpartition + argsort, retrieval.py:276-279) and k = 5000 >= n_docs (the full argsort branch, :281-284).
    Arrays only: the doc texts (newline-joined, utf-8 bytes), the queries, and per (k, query) the returned rows
    (doc row ids in rank order, fp32 scores)."""
    rng = np.random.default_rng(20257)
    vocab_words = [f"w{i}" for i in range(400)]
    corpus = zipf_text_corpus(rng, 2600, vocab_words)
    svc = new_service(tmp)
    svc.build_bm25_index(corpus)
    queries = {"d0": "w8 w9 w10 w11 w12", "d1": "w6 w20 w33 w47", "d2": "w250 w399 w120 w77 w5", "d3": "w0",
               "d4": "w9 w9 w300 w14 w15 w16", "d5": "w7 w13 w21 w34 w55 w89 w144 w233 w377"}
    row = {d: i for i, d in enumerate(svc.doc_ids)}
    out = {"texts": np.frombuffer("\n".join(corpus[d]["text"] for d in svc.doc_ids).encode("utf-8"), dtype=np.uint8),
           "qids": np.array(list(queries)), "qtexts": np.array(list(queries.values()))}
    for k in (1500, 5000):
        svc.clear_cache()
        res = svc.search_bm25(queries, top_k=k)
        for qid in queries:
            out[f"k{k}_{qid}_doc"] = np.array([row[d] for d in res[qid]], dtype=np.int32)
            out[f"k{k}_{qid}_score"] = np.array(list(res[qid].values()), dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "text_deep.npz"), **out)
    svc.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deep":  # only the deep-ranking fixture
        with tempfile.TemporaryDirectory() as tmp:
            make_deep_fixture(tmp)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dense":  # only the dense fixture (the others stay as committed)
        make_dense_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pipeline":  # only the pipeline-twin fixtures, on the committed text corpus
        with open(os.path.join(OUT, "text_small.json"), encoding="utf-8") as f:
            j = json.load(f)
        with tempfile.TemporaryDirectory() as tmp:
            make_pipeline_fixture(tmp, j["corpus"], j["queries"])
        sys.exit(0)
    with tempfile.TemporaryDirectory() as tmp:
        corpus, queries = make_text_fixture(tmp)
        make_csr_fixture(tmp)
        make_registry_fixture(tmp, corpus, queries)
        make_pipeline_fixture(tmp, corpus, queries)
        make_deep_fixture(tmp)
    make_dense_fixture()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))
