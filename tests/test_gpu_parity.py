"""GPU parity tests proper: the HIP path (through the C ABI, libsparse_rx.so) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at larger sizes -- through size-independent
properties.  Bar: doc ids and fp32 score bits EXACTLY equal to the oracle (same canonical tie order: score desc,
doc asc); against reference-produced fixtures, equal modulo the reference's unspecified order inside exact ties
(tests/parity.py).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (checker only)
from parity import assert_canonical_order, assert_ranked_equal  # noqa: E402


@pytest.fixture(scope="module")
def rx():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import sparse_rx
    sparse_rx._capi.lib()  # fails loudly if the HIP library is missing
    return sparse_rx


def _oracle_batch(c, idf, avgdl, q, k, mode=oracle.MODE_BM25_F32, k1=1.2, b=0.75):
    return oracle.search_batch(c.indptr, c.indices, c.data, c.doc_lengths, idf, q[0], q[1], q[2], k, k1, b, avgdl, mode=mode)


def _assert_exact(got, exp, label=""):
    gd, gs, gc = got
    ed, es, ec = exp
    assert np.array_equal(gc, ec), f"{label}: counts differ at {np.flatnonzero(gc != ec)[:8]}: {gc[gc != ec][:8]} vs {ec[gc != ec][:8]}"
    bad = np.flatnonzero((gs.view(np.uint32) != es.view(np.uint32)).any(axis=1) | (gd != ed).any(axis=1))
    if len(bad):
        q = bad[0]
        j = np.flatnonzero((gd[q] != ed[q]) | (gs[q].view(np.uint32) != es[q].view(np.uint32)))[:5]
        raise AssertionError(f"{label}: {len(bad)} queries differ; first q={q} ranks {j}: got {gd[q][j]} {gs[q][j]} exp {ed[q][j]} {es[q][j]}")


def _dev_index(rx, c, idf, avgdl, **kw):
    return rx.DeviceIndex.from_csr(c.indptr, c.indices, c.data, idf, doc_lengths=c.doc_lengths, avgdl=avgdl, **kw)


# ---------------------------------------------------------------------------------------------------------------
# golden fixtures produced by the reference
# ---------------------------------------------------------------------------------------------------------------
def test_text_golden_end_to_end(rx, golden_dir):
    z = np.load(os.path.join(golden_dir, "text_small.npz"))
    with open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8") as f:
        j = json.load(f)
    svc = rx.RetrievalService(device="cuda:0", tile_log2=6)
    with pytest.raises(ValueError, match="BM25 index not built"):
        svc.search_bm25({"q": "x"})
    with pytest.raises(ValueError, match="Empty corpus"):
        svc.build_bm25_index({})
    svc.build_bm25_index(j["corpus"])
    # host index state is bit-equal to the reference's
    assert svc.doc_ids == list(z["doc_ids"])
    assert [t for t, _ in sorted(svc.vocabulary.items(), key=lambda kv: kv[1])] == list(z["vocabulary"])
    assert np.array_equal(svc.host.indptr, z["tf_indptr"]) and np.array_equal(svc.host.indices, z["tf_indices"])
    assert np.array_equal(svc.host.data, z["tf_data"]) and np.array_equal(svc.host.doc_lengths, z["doc_lengths"])
    assert np.array_equal(svc.host.idf.view(np.uint32), z["idf"].view(np.uint32))
    assert svc.avgdl == float(z["avgdl"])
    row = {d: i for i, d in enumerate(svc.doc_ids)}
    qids = list(z["score_qids"])
    for k in ("3", "10", "1000"):
        svc.clear_cache()
        got = svc.search_bm25(j["queries"], top_k=int(k))
        exp = j["results"][k]
        assert list(got.keys()) == list(exp.keys())
        for qid in exp:
            g, e = got[qid], exp[qid]
            full = z["full_scores"][qids.index(qid)] if qid in qids else None
            assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), [row[d] for d in e],
                                np.array(list(e.values()), np.float32), k=min(int(k), len(row)), full_scores=full,
                                label=f"k={k} {qid}")
        again = svc.search_bm25(j["queries"], top_k=int(k))  # cache-hit path
        assert again == got
    st = svc.get_stats()
    for key in ("cache_size", "query_cache_size", "numba_available", "num_docs", "vocab_size", "matrix_density",
                "bm25_memory_mb", "avgdl"):
        assert key in st
    assert st["num_docs"] == j["stats"]["num_docs"] and st["vocab_size"] == j["stats"]["vocab_size"]
    svc.close()


def test_deep_ranking_golden_and_search_after(rx, golden_dir):
    """top_k > 1024 (the engine's list capacity) and top_k >= n_docs (retrieval.py:272-284): served in pages by
    srx_search_after.  (1) against rows the REFERENCE returned at top_k = 1500 / 5000 on a 2600-doc corpus
    (tests/golden/text_deep.npz); (2) against the oracle at k = 5000 / 2048 / 1025 on corpora that take the tier-2
    kernel's hash, flat, wave-dense and block-dense paths; (3) a search-after page equals the matching slice of one deeper
    search, also for k <= 112 (where tier 1 would otherwise serve)."""
    import torch
    from test_oracle_golden import _deep_fixture
    from sparse_rx import synth
    corpus, queries, exp = _deep_fixture(golden_dir)
    for tl in (6, 10):
        svc = rx.RetrievalService(device="cuda:0", tile_log2=tl)
        svc.build_bm25_index(corpus)
        row = {d: i for i, d in enumerate(svc.doc_ids)}
        for k in (1500, 5000):
            svc.clear_cache()
            got = svc.search_bm25(queries, top_k=k)
            for qid in queries:
                ed, es = exp[k][qid]
                g = got[qid]
                assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), ed, es, k=min(k, len(row)),
                                    label=f"deep golden tile={tl} k={k} {qid}")
        svc.close()
    reg = rx.OptimizedBM25Retriever(device="cuda:0", tile_log2=8)  # the registry twin accepts any top_k too; <= 0 gives {}
    reg.build_index_from_corpus(corpus)
    assert reg.search(queries, top_k=0) == {q: {} for q in queries} and reg.search(queries, top_k=-3) == {q: {} for q in queries}
    g = reg.search(queries, top_k=5000)
    assert [len(g[q]) for q in queries] == [len(exp[5000][q][0]) for q in queries]
    reg.close()
    # (2) oracle, several layouts
    c = synth.zipf_corpus_np(60_000, 3_000, 40, seed=77)
    _, idf, avgdl = synth.corpus_stats(c)
    q = synth.queries_np(24, c.vocab, 6, seed=78, dist="zipf")
    for kw in (dict(tile_log2=14), dict(tile_log2=12, unit_tiles=1), dict(tile_log2=9), dict(tile_log2=12, unit_tiles=2)):
        ix = _dev_index(rx, c, idf, avgdl, **kw)
        for k in (5000, 2048, 1025):
            _assert_exact(ix.search(*q, k), _oracle_batch(c, idf, avgdl, q, k), f"deep zipf {kw} k={k}")
        ix.close()
    cs = synth.splade_corpus_np(30_000, 2_000, 80, seed=79)
    ones = np.ones(cs.vocab, dtype=np.float32)
    qs = synth.queries_np(12, cs.vocab, 40, seed=80, dist="zipf", s=0.7, weights="learned")
    for vd, tl, ut in (("f16", 12, 1), ("f32", 13, 0)):
        ix = rx.DeviceIndex.from_csr(cs.indptr, cs.indices, cs.data, ones, mode="dot", val_dtype=vd, tile_log2=tl, unit_tiles=ut)
        _assert_exact(ix.search(*qs, 3000), _oracle_batch(cs, ones, 1.0, qs, 3000, mode=oracle.MODE_TFIDF_F32), f"deep splade {vd} {tl}")
        ix.close()
    # (3) pages by hand on device tensors, small k
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=12)
    dq = tuple(torch.as_tensor(x, device="cuda:0") for x in q)
    d_all, s_all, c_all = (t.cpu().numpy() for t in ix.search_device(*dq, 300))
    for k_page in (100, 64):
        after, got = None, 0
        while got + k_page <= 300:
            d, s, n = ix.search_device(*dq, k_page, after=after)
            torch.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy(), d_all[:, got: got + k_page]) and np.array_equal(s.cpu().numpy().view(np.uint32), s_all[:, got: got + k_page].view(np.uint32))
            full = n == k_page
            after = (torch.where(full, d[:, -1], torch.zeros_like(d[:, -1])).contiguous(), torch.where(full, s[:, -1], torch.zeros_like(s[:, -1])).contiguous())
            got += k_page
    ix.close()


@pytest.mark.parametrize("tile_log2,super_log2,target_blocks", [(6, 0, 0), (8, 8, 4096), (7, 11, 1), (14, 0, 0)])
def test_csr_zipf_golden(rx, golden_dir, tile_log2, super_log2, target_blocks):
    z = np.load(os.path.join(golden_dir, "csr_zipf.npz"))
    ix = rx.DeviceIndex.from_csr(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["idf"], doc_lengths=z["doc_lengths"],
                                 k1=float(z["k1"]), b=float(z["b"]), avgdl=float(z["avgdl"]), tile_log2=tile_log2)
    ix.set_opts(supertile_log2=super_log2, target_blocks=target_blocks)
    for k in (10, 100):
        gd, gs, gc = ix.search(z["q_ptr"], z["q_term"], z["q_weight"], k)
        ed, es, ec = z[f"top{k}_doc"], z[f"top{k}_score"], z[f"top{k}_count"]
        assert np.array_equal(gc, ec)
        for q in range(len(gc)):
            lo, hi = z["q_ptr"][q], z["q_ptr"][q + 1]
            full = oracle.bm25_scores(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"],
                                      z["q_term"][lo:hi], z["q_weight"][lo:hi], float(z["k1"]), float(z["b"]), float(z["avgdl"]))
            c = gc[q]
            assert_ranked_equal(gd[q, :c], gs[q, :c], ed[q, :c], es[q, :c], k=k, full_scores=full, label=f"k={k} q{q}")
            assert_canonical_order(gd[q, :c], gs[q, :c])
            assert np.all(gd[q, c:] == -1) and np.all(gs[q, c:] == 0)
        exp = oracle.search_batch(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["doc_lengths"], z["idf"], z["q_ptr"],
                                  z["q_term"], z["q_weight"], k, float(z["k1"]), float(z["b"]), float(z["avgdl"]))
        _assert_exact((gd, gs, gc), exp, f"zipf k={k}")
    ix.close()


def test_tfidf_dot_mode_golden(rx, golden_dir):
    """simd_tfidf_score twin (evaluate_rag_pipeline.py:95-121): dot mode with idf = log(N/(df+1))."""
    z = np.load(os.path.join(golden_dir, "csr_zipf.npz"))
    ix = rx.DeviceIndex.from_csr(z["tf_indptr"], z["tf_indices"], z["tf_data"], z["idf_tfidf"], mode="dot", tile_log2=8)
    n = z["tfidf_full"].shape[0]
    k = 50
    gd, gs, gc = ix.search(z["q_ptr"][: n + 1], z["q_term"][: z["q_ptr"][n]], z["q_weight"][: z["q_ptr"][n]], k)
    for q in range(n):
        full = z["tfidf_full"][q]  # produced by the reference
        idx, sc = oracle.topk(full, k)
        keep = sc > 0
        c = gc[q]
        assert c == keep.sum()
        assert np.array_equal(gd[q, :c], idx[keep]) and np.array_equal(gs[q, :c].view(np.uint32), sc[keep].view(np.uint32))
    ix.close()


# ---------------------------------------------------------------------------------------------------------------
# seeded synthetic corpora vs the oracle (exact)
# ---------------------------------------------------------------------------------------------------------------
def test_uniform_sparse_hash_path(rx):
    """C2-shaped, scaled down: every unit goes through the LDS hash path; several splits; k=100."""
    from sparse_rx import synth
    c = synth.uniform_corpus_np(200_000, 20_000, 40, seed=20252)
    _, idf, avgdl = synth.corpus_stats(c)
    q = synth.queries_np(256, c.vocab, 8, seed=77)
    for ut in (1, 3, 5, 7):  # indexes padded for units of a non-power-of-two number of tiles (tier 1)
        ixu = _dev_index(rx, c, idf, avgdl, tile_log2=12, unit_tiles=ut)
        assert ixu.unit_tiles == ut
        _assert_exact(ixu.search(*q, 100), _oracle_batch(c, idf, avgdl, q, 100), f"uniform unit_tiles={ut}")
        ixu.close()
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=12)
    for ut in (1, 3):  # search-time override of the unit: served by tier 2 alone, still exact
        ix.set_opts(unit_tiles=ut)
        _assert_exact(ix.search(*q, 100), _oracle_batch(c, idf, avgdl, q, 100), f"uniform override unit_tiles={ut}")
    # target_blocks below the batch size: whole rounds of unsplit queries + a tail cut into 2 / 3 / 4 splits
    for sl, tb, dbg in ((0, 0, 0), (12, 0, 0), (16, 1, 0), (14, 100000, 0), (0, 0, 8), (17, 0, 8), (18, 0, 0), (0, 100, 0), (0, 60, 0),
                        (12, 250, 0), (0, 60, 8)):
        ix.set_opts(supertile_log2=sl, target_blocks=tb, debug=dbg)  # debug=8: everything through the tier-2 block kernel
        for k in (100, 10, 128, 129):
            _assert_exact(ix.search(*q, k), _oracle_batch(c, idf, avgdl, q, k), f"uniform sl={sl} tb={tb} dbg={dbg} k={k}")
    ix.close()


def test_zipf_dense_and_overflow_paths(rx):
    """C5-shaped, scaled down: hot terms (df ~ n_docs, negative idf) force the dense-tile path and the overflow
    packer; ties are abundant."""
    from sparse_rx import synth
    c = synth.zipf_corpus_np(120_000, 5_000, 60, seed=20255)
    df, idf, avgdl = synth.corpus_stats(c)
    assert idf.min() < 0 and df.max() > 0.9 * c.n_docs
    q = synth.queries_np(96, c.vocab, 8, seed=5, dist="zipf")
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=14)
    for sl, dbg, tb in ((0, 0, 0), (14, 0, 0), (17, 0, 0), (15, 8, 0), (0, 0, 40), (14, 0, 90)):  # tb: mixed unsplit / split queries
        ix.set_opts(supertile_log2=sl, debug=dbg, target_blocks=tb)
        for k in (100, 1000, 1):
            _assert_exact(ix.search(*q, k), _oracle_batch(c, idf, avgdl, q, k), f"zipf sl={sl} dbg={dbg} tb={tb} k={k}")
    ix.close()
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=10)  # small tiles: overflow packer groups several tiles per unit
    ix.set_opts(supertile_log2=16)
    _assert_exact(ix.search(*q, 100), _oracle_batch(c, idf, avgdl, q, 100), "zipf packer")
    ix.close()
    for ut in (1, 2):  # wave-level dense tiles with negative idf (sentinel contributions are -0), unmasked / masked form
        ix = _dev_index(rx, c, idf, avgdl, tile_log2=12, unit_tiles=ut)
        for k in (1000, 300):
            _assert_exact(ix.search(*q, k), _oracle_batch(c, idf, avgdl, q, k), f"zipf wave-dense ut={ut} k={k}")
        ix.close()


def test_splade_dot_f16_k1000(rx):
    """C4-shaped, scaled down: learned-sparse weights in fp16, 50 terms / query, k = 1000, dot mode."""
    from sparse_rx import synth
    c = synth.splade_corpus_np(60_000, 3_000, 100, seed=20254)
    idf = np.ones(c.vocab, dtype=np.float32)
    q = synth.queries_np(48, c.vocab, 50, seed=9, dist="zipf", s=0.7, weights="learned")
    exp = _oracle_batch(c, idf, 1.0, q, 1000, mode=oracle.MODE_TFIDF_F32)
    # tiles of <= 4096 docs: the wave-level dense path (LDS float atomics in term order); one-tile units: its unmasked form
    for vd, tl, ut in (("f16", 13, 0), ("f32", 13, 0), ("f16", 12, 1), ("f32", 12, 1), ("f16", 12, 4), ("f16", 10, 1), ("f32", 11, 3)):
        ix = rx.DeviceIndex.from_csr(c.indptr, c.indices, c.data, idf, mode="dot", val_dtype=vd, tile_log2=tl, unit_tiles=ut)
        _assert_exact(ix.search(*q, 1000), exp, f"splade {vd} tile={tl} ut={ut}")
        if tl == 12:
            for dbg in (2048, 4096, 8192):  # block-level dense tiles / masked form / general selection: same rows
                ix.set_opts(debug=dbg)
                _assert_exact(ix.search(*q, 1000), exp, f"splade {vd} tile={tl} ut={ut} debug={dbg}")
        ix.close()


def test_edge_cases(rx):
    from sparse_rx import synth
    c = synth.zipf_corpus_np(5_000, 700, 30, seed=3)
    _, idf, avgdl = synth.corpus_stats(c)
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=8)
    # empty batch
    d, s, n = ix.search(np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), 10)
    assert d.shape == (0, 10)
    # empty queries interleaved, a negative-idf-only query, a 1-term query, a > 256-term query (general path)
    long_terms = np.arange(0, 700, 2, dtype=np.int32)  # 350 distinct terms
    qs = [np.array([], np.int32), np.array([0], np.int32), np.array([699], np.int32), long_terms, np.array([], np.int32),
          np.array([0, 1, 2, 3], np.int32)]
    q_ptr = np.zeros(len(qs) + 1, np.int32)
    q_ptr[1:] = np.cumsum([len(x) for x in qs])
    q_term = np.concatenate(qs)
    q_w = np.ones(len(q_term), np.float32)
    q_w[::3] = 2.0
    q = (q_ptr, q_term, q_w)
    for k in (1, 7, 100, 1024):
        got = ix.search(*q, k)
        _assert_exact(got, _oracle_batch(c, idf, avgdl, q, k), f"edge k={k}")
        assert got[2][0] == 0 and got[2][4] == 0
    with pytest.raises(ValueError):
        ix.search(*q, 0)
    for k in (1025, 2500):  # deeper than the engine's lists: paged with srx_search_after (index.deep_search)
        got = ix.search(*q, k)
        _assert_exact(got, _oracle_batch(c, idf, avgdl, q, k), f"edge deep k={k}")
    # host batches are validated before any launch: out-of-range / repeated terms never reach the kernels
    with pytest.raises(ValueError, match="out of range"):
        ix.search(np.array([0, 2], np.int32), np.array([1, 700], np.int32), np.ones(2, np.float32), 10)
    with pytest.raises(ValueError, match="same term twice"):
        ix.search(np.array([0, 2], np.int32), np.array([5, 5], np.int32), np.ones(2, np.float32), 10)
    # term order inside a query = accumulation order: descending ids == the given-order oracle
    rq = (q_ptr, np.concatenate([x[::-1] for x in qs]), q_w)
    got = ix.search(*rq, 100)
    exp = oracle.search_batch(c.indptr, c.indices, c.data, c.doc_lengths, idf, *rq, 100, 1.2, 0.75, avgdl,
                              mode=oracle.MODE_BM25_F32_GIVEN_ORDER)
    _assert_exact(got, exp, "edge reversed-term order")
    ix.close()
    # n_docs not a multiple of the tile, single doc, all-identical docs (one giant tie group)
    rows = 1000
    indptr = np.arange(rows + 1, dtype=np.int64) * 2
    indices = np.tile(np.array([1, 3], np.int32), rows)
    data = np.ones(2 * rows, np.float32)
    dl = np.full(rows, 2, np.float32)
    idf2 = np.array([0.1, 0.7, 0.2, 0.9, 0.3], np.float32)
    ix = rx.DeviceIndex.from_csr(indptr, indices, data, idf2, doc_lengths=dl, avgdl=2.0, tile_log2=6)
    qq = (np.array([0, 2], np.int32), np.array([1, 3], np.int32), np.array([1, 1], np.float32))
    for k in (10, 64, 1000):
        d, s, n = ix.search(*qq, k)
        assert n[0] == min(k, rows) and np.array_equal(d[0, : n[0]], np.arange(n[0]))  # ties -> ascending doc id
        assert len(set(s[0, : n[0]].tolist())) == 1
    ix.close()


def test_merge_topk_matches_single_shard(rx):
    """Doc-range shards + srx_merge_topk == one index over all docs (the multi-GPU contract, on one device)."""
    import torch
    from sparse_rx import synth
    c = synth.uniform_corpus_np(90_000, 8_000, 30, seed=11)
    _, idf, avgdl = synth.corpus_stats(c)
    q = synth.queries_np(64, c.vocab, 6, seed=13)
    k = 100
    whole = _dev_index(rx, c, idf, avgdl, tile_log2=12)
    exp = whole.search(*q, k)
    whole.close()
    for shards in (2, 3, 8, 50):
        bounds = [(c.n_docs * r) // shards for r in range(shards + 1)]
        parts = []
        for r in range(shards):
            a, b = bounds[r], bounds[r + 1]
            sub_ptr = c.indptr[a: b + 1] - c.indptr[a]
            lo, hi = c.indptr[a], c.indptr[b]
            ix = rx.DeviceIndex.from_csr(sub_ptr, c.indices[lo:hi], c.data[lo:hi], idf, doc_lengths=c.doc_lengths[a:b],
                                         avgdl=avgdl, tile_log2=10, doc_base=a)  # GLOBAL idf / avgdl
            qd = [torch.as_tensor(x, device="cuda:0") for x in q]
            parts.append(ix.search_device(*qd, k))
            torch.cuda.synchronize()
            ix.close()
        in_doc = torch.stack([p[0] for p in parts], dim=1)
        in_score = torch.stack([p[1] for p in parts], dim=1)
        in_count = torch.stack([p[2] for p in parts], dim=1)
        d, s, n = rx.merge_topk_device(in_doc, in_score, in_count, k)
        torch.cuda.synchronize()
        _assert_exact((d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()), exp, f"shards={shards}")
        # the all-gather layout [shards, nq, k]
        d, s, n = rx.merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]),
                                       torch.stack([p[2] for p in parts]), k, gathered=True)
        torch.cuda.synchronize()
        _assert_exact((d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()), exp, f"gathered shards={shards}")
        # the packed single-buffer exchange format [shards, nq, 2k+1]
        from sparse_rx.index import merge_topk_packed_device, pack_results
        packed = torch.stack([pack_results(*p) for p in parts])
        d, s, n = merge_topk_packed_device(packed, k)
        torch.cuda.synchronize()
        _assert_exact((d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()), exp, f"packed shards={shards}")
        # packed rows out as well (srx_merge_topk_packed_out)
        from sparse_rx.index import merge_topk_packed_out_device
        rows = merge_topk_packed_out_device(packed, k)
        torch.cuda.synchronize()
        _assert_exact((rows[:, :k].cpu().numpy(), rows[:, k:2 * k].contiguous().view(torch.float32).cpu().numpy(),
                       rows[:, 2 * k].cpu().numpy()), exp, f"packed-out shards={shards}")
    # srx_search_packed == srx_search, row by row (one split and several splits per query)
    for nq_sub in (len(q[0]) - 1, 3):
        qs = (q[0][: nq_sub + 1], q[1][: q[0][nq_sub]], q[2][: q[0][nq_sub]])
        ix = _dev_index(rx, c, idf, avgdl, tile_log2=10)
        qd = [torch.as_tensor(x, device="cuda:0") for x in qs]
        for tb in (0, 2, 7):  # 2, 7: more queries than the split target -> unsplit rounds (final rows written by tier 1) + a split tail (merge kernel)
            ix.set_opts(target_blocks=tb)
            d, s, n = ix.search_device(*qd, k)
            rows = ix.search_packed_device(*qd, k)
            torch.cuda.synchronize()
            assert torch.equal(rows[:, :k], d) and torch.equal(rows[:, k:2 * k], s.view(torch.int32)) and torch.equal(rows[:, 2 * k], n)
            _assert_exact((d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()), tuple(x[: nq_sub] for x in exp), f"packed tb={tb} nq={nq_sub}")
        ix.close()


def test_term_bounds_kernel_matches_sorted_values(rx):
    """srx_build_term_bounds (the K-th largest stored value of every term for K in FINE_KS, through DeviceIndex._term_bounds)
    against a NumPy sort of every term's values: ties at the ranks, zeros (they do not count: a bound of 0 either way), empty
    terms, terms shorter and longer than 1 024 postings and longer than one pass of the kernel (4 096), f32 and f16; a negative
    value anywhere -> no bounds."""
    import torch
    rng = np.random.default_rng(31)
    lens = np.concatenate([[0, 1, 2, 9, 10, 11, 99, 100, 101, 1000, 1023, 1024, 1025, 4096, 4097, 20_000, 150_000, 0, 3],
                           rng.integers(0, 300, 400)]).astype(np.int64)
    V = len(lens)
    term_ptr = np.zeros(V + 1, np.int64)
    term_ptr[1:] = np.cumsum(lens)
    nnz = int(term_ptr[-1])
    for dt in (np.float32, np.float16):
        vals = rng.random(nnz).astype(np.float32)
        vals[rng.random(nnz) < 0.2] = 0.0
        few = rng.random(nnz) < 0.3
        vals[few] = rng.choice(np.array([0.25, 0.5, 0.75, 1.5], np.float32), int(few.sum()))  # heavy ties
        vals = vals.astype(dt)
        dev = torch.device("cuda:0")
        tp_d = torch.as_tensor(term_ptr, device=dev)
        v_d = torch.as_tensor(vals, device=dev)
        cols = torch.zeros(1, dtype=torch.int32, device=dev)  # only tested for None-ness
        got = rx.DeviceIndex._term_bounds(torch, cols, v_d, tp_d, None, V).cpu().numpy()
        ks = rx.DeviceIndex.FINE_KS
        exp = np.zeros((V, len(ks)), np.float32)
        for t in range(V):
            x = np.sort(vals[term_ptr[t]:term_ptr[t + 1]].astype(np.float32))[::-1]
            for j, K in enumerate(ks):
                if K <= len(x):
                    exp[t, j] = x[K - 1]
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), dt
        v_neg = v_d.clone()
        v_neg[nnz // 2] = -0.5
        assert rx.DeviceIndex._term_bounds(torch, cols, v_neg, tp_d, None, V) is None


def test_blocked_layout_matches_numpy_builder(rx):
    """srx_build_tile_skip + srx_build_blocks (device index construction) against the NumPy restatement of the blocked
    layout (tests/parity.py): padded runs per unit, sentinels (doc -1, value 0), padded skip table and term offsets --
    f32 and f16 values, units of 1 / 3 / 4 tiles, docs not a multiple of the tile, empty terms."""
    from parity import np_build_blocks, np_compact_blocks
    from sparse_rx import synth
    c = synth.zipf_corpus_np(7_013, 300, 20, seed=77)
    idf = np.ones(c.vocab, np.float32)
    for ut in (1, 3, 4):
        for vd, npd in (("f32", np.float32), ("f16", np.float16)):
            ix = rx.DeviceIndex.from_csr(c.indptr, c.indices, c.data, idf, mode="dot", val_dtype=vd, tile_log2=8, unit_tiles=ut)
            tp, post, skip, nb = np_build_blocks(c.indptr, c.indices, c.data, c.n_docs, c.vocab, 8, ut, npd)
            assert ix.n_blocks == nb and ix.nnz == len(c.indices) and ix.unit_tiles == ut
            assert np.array_equal(ix.term_ptr.cpu().numpy(), tp), (ut, vd)
            assert np.array_equal(ix.tile_skip.cpu().numpy(), skip), (ut, vd)
            got = ix.post.cpu().numpy()
            assert got.shape == post.shape and np.array_equal(got, post), (ut, vd)
            # the compact copy tier 1 streams (16-bit unit-local doc ids)
            assert ix.post16 is not None
            assert np.array_equal(ix.post16.cpu().numpy(), np_compact_blocks(post, ut << 8, npd)), (ut, vd)
            ix.close()
    auto = rx.DeviceIndex.from_csr(c.indptr, c.indices, c.data, idf, mode="dot", tile_log2=8)
    assert 1 <= auto.unit_tiles <= 64 and (auto.unit_tiles << 8) <= 49152
    auto.close()
    # a unit of more than 49152 docs has no compact copy: tier 2 serves every query, results stay exact
    big = synth.uniform_corpus_np(70_000, 500, 6, seed=5)
    ixb = rx.DeviceIndex.from_csr(big.indptr, big.indices, big.data, np.ones(big.vocab, np.float32), mode="dot", tile_log2=14, unit_tiles=4)
    assert ixb.post16 is None
    q = synth.queries_np(16, big.vocab, 4, seed=6)
    gd, gs, gc = ixb.search(*q, 10)
    import oracle
    ed, es, ec = oracle.search_batch(big.indptr, big.indices, big.data, None, np.ones(big.vocab, np.float32), q[0], q[1], q[2], 10,
                                     mode=oracle.MODE_TFIDF_F32)
    assert np.array_equal(gc, ec) and np.array_equal(gd, ed) and np.array_equal(gs.view(np.uint32), es.view(np.uint32))
    ixb.close()


def test_impacts_bit_exact(rx):
    """srx_build_impacts == the reference's fp32 expression (retrieval.py:58,70-71) evaluated by NumPy."""
    import torch
    from oracle import np_oracle
    rng = np.random.default_rng(1)
    n, nnz = 5000, 200_000
    tf = rng.integers(1, 40, nnz).astype(np.float32)
    doc = rng.integers(0, n, nnz).astype(np.int32)
    dl = rng.integers(1, 3000, n).astype(np.float32)
    for k1, b, avgdl in ((1.2, 0.75, 137.3), (1.6, 0.8, 55.5), (1000.0, 0.0, 10.0)):
        exp = np_oracle.impacts_f32(tf, doc, dl, k1, b, avgdl)
        out = torch.empty(nnz, dtype=torch.float32, device="cuda:0")
        t = [torch.as_tensor(x, device="cuda:0") for x in (tf, doc, dl)]
        rc = rx._capi.lib().srx_build_impacts(0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), nnz, k1, b, avgdl,
                                              out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), exp.view(np.uint32))


def test_larger_properties(rx):
    """1 M docs (C2 shape): exact vs the oracle on a query sample + size-independent properties on the batch:
    descending order, doc-unique, scores of returned docs recomputed exactly from the CSR rows, idempotence,
    and shard-invariance (k-prefix property: top-10 is the prefix of top-100)."""
    from sparse_rx import synth
    c = synth.uniform_corpus_np(1_000_000, 50_000, 50, seed=20252)
    _, idf, avgdl = synth.corpus_stats(c)
    q = synth.queries_np(1000, c.vocab, 8, seed=20252 + 1)
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=14)
    d100, s100, n100 = ix.search(*q, 100)
    d10, s10, n10 = ix.search(*q, 10)
    assert np.array_equal(d10, d100[:, :10]) and np.array_equal(s10, s100[:, :10])
    again = ix.search(*q, 100)
    assert all(np.array_equal(a, b) for a, b in zip(again, (d100, s100, n100)))
    for qi in range(0, 1000, 50):
        c_ = n100[qi]
        assert_canonical_order(d100[qi, :c_], s100[qi, :c_])
    sample = slice(0, 24)
    qs = (q[0][: 25] - q[0][0], q[1][q[0][0]: q[0][24]], q[2][q[0][0]: q[0][24]])
    exp = _oracle_batch(c, idf, avgdl, qs, 100)
    _assert_exact((d100[sample], s100[sample], n100[sample]), exp, "1M sample")
    ix.close()


def test_c3_full_size_properties(rx):
    """BASELINE config C3 at FULL size (10 M docs x 100 k vocab, 10^9 postings, 10 k queries x 8 terms, k = 100), corpus
    generated on the device like bench.py does.  The oracle's full-CSR scan is bench.py's job at this size (it checks a
    sample of every run); here the size-independent properties: canonical order, doc-unique rows, k-prefix property,
    idempotence, and -- the strong one -- every query of a 96-query sub-batch returns bit-identical rows when searched
    alone (64 doc-range splits + tier-2 lists + merge kernel) and inside the 10 k batch (unsplit, ranked by tier 1)."""
    import torch
    from sparse_rx import synth
    dev = torch.device("cuda:0")
    n_docs, V, nnz_per_doc, seed, nq, k = 10_000_000, 100_000, 100, 20253, 10_000, 100
    rows_l, cols_l, tf_l, dl_l = [], [], [], []
    for ci in range(n_docs // synth.CHUNK_DOCS):
        r, cc, tf, dl = synth.uniform_chunk_torch(ci, synth.CHUNK_DOCS, V, nnz_per_doc, seed, dev)
        rows_l.append(r + ci * synth.CHUNK_DOCS); cols_l.append(cc); tf_l.append(tf); dl_l.append(dl)
    rows, cols, tf, dl = torch.cat(rows_l), torch.cat(cols_l), torch.cat(tf_l), torch.cat(dl_l)
    del rows_l, cols_l, tf_l, dl_l
    df = torch.bincount(cols, minlength=V).cpu().numpy()
    idf = torch.as_tensor(np.log((n_docs - df + 0.5) / (df + 0.5)).astype(np.float32), device=dev)
    avgdl = float(np.mean(dl.cpu().numpy()))
    ix = rx.DeviceIndex.from_coo(rows, cols, tf, idf, n_docs, doc_lengths=dl, avgdl=avgdl, device=dev, tile_log2=14)
    del rows, cols, tf
    assert ix.post16 is not None and ix.nnz > 990_000_000
    q = synth.queries_np(nq, V, 8, seed=seed + 1)
    d, s, n = ix.search(*q, k)
    assert np.all(n == k)                                        # ~80 k matching docs per query
    assert np.all(d >= 0) and np.all(d < n_docs) and np.all(s > 0)
    assert np.all((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (d[:, :-1] < d[:, 1:])))  # (score desc, doc asc), doc-unique
    d2, s2, n2 = ix.search(*q, k)
    assert np.array_equal(d, d2) and np.array_equal(s.view(np.uint32), s2.view(np.uint32)) and np.array_equal(n, n2)
    d10, s10, n10 = ix.search(*q, 10)
    assert np.array_equal(d10, d[:, :10]) and np.array_equal(s10.view(np.uint32), s[:, :10].view(np.uint32))
    lo, hi = 5000, 5096                                          # a sub-batch: other work-item plan (splits + merge)
    qs = ((q[0][lo: hi + 1] - q[0][lo]).astype(np.int32), q[1][q[0][lo]: q[0][hi]], q[2][q[0][lo]: q[0][hi]])
    ds, ss, ns = ix.search(*qs, k)
    assert np.array_equal(ds, d[lo:hi]) and np.array_equal(ss.view(np.uint32), s[lo:hi].view(np.uint32)) and np.array_equal(ns, n[lo:hi])
    ix.set_opts(debug=8)                                         # the same sub-batch through the tier-2 kernel alone
    dt, st, nt_ = ix.search(*qs, k)
    assert np.array_equal(dt, ds) and np.array_equal(st.view(np.uint32), ss.view(np.uint32)) and np.array_equal(nt_, ns)
    ix.close()


def _full_size_index(rx, kind, n_docs, V, nnz_per_doc, seed, want_host_csr, **build_kw):
    """A BASELINE-size corpus generated on the device chunk by chunk (the generators of bench.py), its DeviceIndex, and
    -- for the oracle sample -- the doc-major CSR on the host."""
    import torch
    from sparse_rx import synth
    dev = torch.device("cuda:0")
    gen = {"uniform": synth.uniform_chunk_torch, "zipf": synth.zipf_chunk_torch, "splade": synth.splade_chunk_torch}[kind]
    rows_l, cols_l, tf_l, dl_l = [], [], [], []
    for ci in range(n_docs // synth.CHUNK_DOCS):
        r, cc, tf, dl = gen(ci, synth.CHUNK_DOCS, V, nnz_per_doc, seed, dev)
        rows_l.append(r + ci * synth.CHUNK_DOCS); cols_l.append(cc); tf_l.append(tf); dl_l.append(dl)
    rows, cols, tf, dl = torch.cat(rows_l), torch.cat(cols_l), torch.cat(tf_l), torch.cat(dl_l)
    del rows_l, cols_l, tf_l, dl_l
    if kind == "splade":
        idf_np, avgdl = np.ones(V, dtype=np.float32), 1.0
    else:
        df = torch.bincount(cols, minlength=V).cpu().numpy()
        idf_np = np.log((n_docs - df + 0.5) / (df + 0.5)).astype(np.float32)
        avgdl = float(np.mean(dl.cpu().numpy()))
    host = None
    if want_host_csr:
        indptr = torch.zeros(n_docs + 1, dtype=torch.int64, device=dev)
        indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_docs), 0)
        host = (indptr.cpu().numpy(), cols.cpu().numpy(), tf.cpu().numpy(), dl.cpu().numpy())
        del indptr
    ix = rx.DeviceIndex.from_coo(rows, cols, tf, torch.as_tensor(idf_np, device=dev), n_docs, doc_lengths=dl, avgdl=avgdl, device=dev,
                                 mode="dot" if kind == "splade" else "bm25", val_dtype="f16" if kind == "splade" else "f32", **build_kw)
    del rows, cols, tf
    return ix, idf_np, avgdl, host


def _assert_rows_equal(a, b, label):
    assert np.array_equal(a[2], b[2]), f"{label}: counts"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), f"{label}: rows"


def _sub_batch(q, lo, hi):
    return ((q[0][lo: hi + 1] - q[0][lo]).astype(np.int32), q[1][q[0][lo]: q[0][hi]], q[2][q[0][lo]: q[0][hi]])


def test_c4_full_size(rx):
    """BASELINE config C4 at FULL size: SPLADE-style learned sparse, 5 M docs x 30 k vocab, ~150 nnz / doc, fp16 weights, 1 000
    queries x 50 terms, k = 1000 (bench.py --workload c4: 4096-doc tiles, one-tile units -> the tier-2 kernel's wave-level
    dense tiles).  A 16-query sample against the oracle's full scan, and size-independent properties over the whole batch:
    canonical order, idempotence, the k-prefix property ACROSS tiers (k = 100 is ranked by tier 1 from the compact copy,
    k = 1000 by tier 2 from the canonical blocks), a sub-batch under another work-item plan, the block-level dense tiles."""
    from sparse_rx import synth
    n_docs, V, nq, k = 5_000_000, 30_000, 1_000, 1000
    ix, idf, avgdl, host = _full_size_index(rx, "splade", n_docs, V, 150, 20254, True, tile_log2=12, unit_tiles=1, keep_canonical=False)
    assert ix.post is None  # one copy of the postings, as bench.py runs it: tier 2 reads the compact blocks
    q = synth.queries_np(nq, V, 50, seed=20255, dist="zipf", s=0.7, weights="learned")
    got = ix.search(*q, k)
    d, s, n = got
    assert np.all(n == k) and np.all(d >= 0) and np.all(d < n_docs) and np.all(s > 0)
    assert np.all((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (d[:, :-1] < d[:, 1:])))
    qs16 = _sub_batch(q, 0, 16)
    exp = oracle.search_batch(host[0], host[1], host[2], host[3], idf, qs16[0], qs16[1], qs16[2], k, 1.2, 0.75, avgdl, native=True,
                              mode=oracle.MODE_TFIDF_F32)
    _assert_exact((d[:16], s[:16], n[:16]), exp, "c4 oracle sample")
    del host
    _assert_rows_equal(ix.search(*q, k), got, "c4 idempotence")
    d100, s100, n100 = ix.search(*q, 100)   # tier 1 (k <= 112): other kernel, other copy of the postings
    _assert_rows_equal((d100, s100, n100), (d[:, :100], s[:, :100], np.minimum(n, 100)), "c4 k-prefix across tiers")
    sub = _sub_batch(q, 400, 464)
    _assert_rows_equal(ix.search(*sub, k), (d[400:464], s[400:464], n[400:464]), "c4 sub-batch")
    ix.set_opts(debug=2048)  # block-level dense tiles instead of the wave-level ones
    _assert_rows_equal(ix.search(*sub, k), (d[400:464], s[400:464], n[400:464]), "c4 block-level dense tiles")
    ix.close()


def test_c5_full_size(rx):
    """BASELINE config C5 at FULL size: Zipf(1.0) postings, 10 M docs x 100 k vocab, 256 queries of hot terms (negative idf
    among them), k = 100 -- long posting runs, tier 2's dense tiles.  A 16-query oracle sample + canonical order,
    idempotence, k-prefix, a sub-batch under another plan, and the tier-2 kernel alone against the default two-tier plan."""
    from sparse_rx import synth
    n_docs, V, nq, k = 10_000_000, 100_000, 256, 100
    ix, idf, avgdl, host = _full_size_index(rx, "zipf", n_docs, V, 100, 20255, True, tile_log2=14, keep_canonical=False)
    assert ix.post is None
    q = synth.queries_np(nq, V, 8, seed=20256, dist="zipf", s=1.0)
    got = ix.search(*q, k)
    d, s, n = got
    full = n == k
    assert full.mean() > 0.5 and np.all(d[full] >= 0) and np.all(d < n_docs)
    rows_ok = (s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & ((d[:, :-1] < d[:, 1:]) | (d[:, 1:] < 0)))
    assert np.all(rows_ok)
    qs16 = _sub_batch(q, 0, 16)
    exp = oracle.search_batch(host[0], host[1], host[2], host[3], idf, qs16[0], qs16[1], qs16[2], k, 1.2, 0.75, avgdl, native=True)
    _assert_exact((d[:16], s[:16], n[:16]), exp, "c5 oracle sample")
    del host
    _assert_rows_equal(ix.search(*q, k), got, "c5 idempotence")
    d10, s10, n10 = ix.search(*q, 10)
    _assert_rows_equal((d10, s10, n10), (d[:, :10], s[:, :10], np.minimum(n, 10)), "c5 k-prefix")
    sub = _sub_batch(q, 100, 132)
    _assert_rows_equal(ix.search(*sub, k), (d[100:132], s[100:132], n[100:132]), "c5 sub-batch")
    ix.set_opts(debug=8)  # everything through the tier-2 kernel
    _assert_rows_equal(ix.search(*q, k), got, "c5 tier 2 alone")
    ix.close()


def test_one_copy_of_the_postings(rx, tmp_path):
    """``keep_canonical=False`` / ``drop_canonical()``: the canonical blocks are freed and the tier-2 kernel reads the compact
    copy (16-bit unit-local ids + the unit's first doc).  Every tier-2 path -- hash units, flat tiles, block- and
    wave-level dense tiles (masked / unmasked), the overflow packer, > 256-term queries, search-after pages -- must return
    the oracle's rows bit for bit, alone (debug = 8) and behind tier 1; what needs the canonical blocks raises."""
    from sparse_rx import synth
    cz = synth.zipf_corpus_np(120_000, 5_000, 60, seed=20255)   # hot terms: dense tiles, packer, negative idf
    _, idf, avgdl = synth.corpus_stats(cz)
    qz = synth.queries_np(64, cz.vocab, 8, seed=5, dist="zipf")
    for kw in (dict(tile_log2=14), dict(tile_log2=10), dict(tile_log2=12, unit_tiles=1), dict(tile_log2=12, unit_tiles=2)):
        both = _dev_index(rx, cz, idf, avgdl, **kw)
        one = _dev_index(rx, cz, idf, avgdl, keep_canonical=False, **kw)
        assert one.post is None and one.post16 is not None and one.device_bytes() < 0.6 * both.device_bytes()
        for dbg in (0, 8):
            one.set_opts(debug=dbg)
            for k in (100, 1000, 1, 2500):
                _assert_exact(one.search(*qz, k), _oracle_batch(cz, idf, avgdl, qz, k), f"one copy zipf {kw} dbg={dbg} k={k}")
        with pytest.raises(ValueError, match="canonical"):
            one.set_opts(supertile_log2=16)
        with pytest.raises(ValueError, match="canonical"):
            one.save(str(tmp_path / "x.srx"))
        assert both.drop_canonical() and both.post is None and not both.drop_canonical()
        _assert_exact(both.search(*qz, 100), _oracle_batch(cz, idf, avgdl, qz, 100), f"dropped later {kw}")
        both.close()
        one.close()
    cs = synth.splade_corpus_np(60_000, 3_000, 100, seed=20254)  # 50-term queries, fp16, k = 1000: flat tiles + wave-level dense tiles
    ones = np.ones(cs.vocab, dtype=np.float32)
    qs = synth.queries_np(32, cs.vocab, 50, seed=9, dist="zipf", s=0.7, weights="learned")
    exp = _oracle_batch(cs, ones, 1.0, qs, 1000, mode=oracle.MODE_TFIDF_F32)
    for vd, tl, ut in (("f16", 13, 0), ("f32", 13, 0), ("f16", 12, 1), ("f32", 12, 1), ("f16", 12, 4), ("f16", 10, 1), ("f32", 11, 3)):
        ix = rx.DeviceIndex.from_csr(cs.indptr, cs.indices, cs.data, ones, mode="dot", val_dtype=vd, tile_log2=tl, unit_tiles=ut, keep_canonical=False)
        assert ix.post is None
        _assert_exact(ix.search(*qs, 1000), exp, f"one copy splade {vd} tile={tl} ut={ut}")
        if tl == 12:
            for dbg in (2048, 4096, 8192, 128):
                ix.set_opts(debug=dbg)
                _assert_exact(ix.search(*qs, 1000), exp, f"one copy splade {vd} tile={tl} ut={ut} debug={dbg}")
        ix.close()
    ce = synth.zipf_corpus_np(5_000, 700, 30, seed=3)            # a > 256-term query: the general path
    _, idfe, avgdle = synth.corpus_stats(ce)
    ix = _dev_index(rx, ce, idfe, avgdle, tile_log2=8, keep_canonical=False)
    long_terms = np.arange(0, 700, 2, dtype=np.int32)
    qe = (np.array([0, len(long_terms), len(long_terms) + 3], np.int32), np.concatenate([long_terms, np.array([1, 5, 9], np.int32)]),
          np.ones(len(long_terms) + 3, np.float32))
    for k in (7, 100, 1024):
        _assert_exact(ix.search(*qe, k), _oracle_batch(ce, idfe, avgdle, qe, k), f"one copy general path k={k}")
    ix.close()
    # the API mirror: RetrievalService(one_copy=True) returns the reference's dicts from half the index memory
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(golden, "text_small.json"), encoding="utf-8") as f:
        j = json.load(f)
    sizes = {}
    for one_copy in (False, True):
        svc = rx.RetrievalService(device="cuda:0", tile_log2=6, one_copy=one_copy)
        svc.build_bm25_index(j["corpus"])
        assert (svc.dev.post is None) == one_copy
        sizes[one_copy] = svc.get_stats()["device_index_mb"]
        row = {d: i for i, d in enumerate(svc.doc_ids)}
        for kk in ("3", "10", "1000"):
            got = svc.search_bm25(j["queries"], top_k=int(kk))
            for qid, e in j["results"][kk].items():
                g = got[qid]
                assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), [row[d] for d in e],
                                    np.array(list(e.values()), np.float32), k=min(int(kk), len(row)), label=f"one_copy={one_copy} k={kk} {qid}")
        svc.close()
    assert sizes[True] < sizes[False]


def test_registry_twin_golden(rx, golden_dir, tmp_path):
    """OptimizedBM25Retriever / RetrieverRegistry / OptimizedRetriever against results the reference's registry twin
    produced on the same corpus (tests/golden/registry_small.json): bm25, bm25_custom (k1=1.6, b=0.8), tfidf."""
    with open(os.path.join(golden_dir, "registry_small.json"), encoding="utf-8") as f:
        reg = json.load(f)
    with open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8") as f:
        j = json.load(f)
    cfgs = {"bm25": {"type": "bm25", "params": {"k1": 1.2, "b": 0.75}},
            "bm25_msmarco": {"type": "bm25_custom", "params": {"k1": 1.6, "b": 0.8}},
            "tfidf": {"type": "tfidf"}}
    for name, cfg in cfgs.items():
        cfg = dict(cfg)
        cfg.setdefault("params", {})
        cfg["params"] = dict(cfg["params"], tile_log2=6)
        r = rx.RetrieverRegistry.create(cfg)
        assert (r.k1, r.b) == (reg[name]["k1"], reg[name]["b"])
        r.build_index_from_corpus(j["corpus"])
        row = {d: i for i, d in enumerate(r.doc_ids)}
        for k in ("5", "50"):
            got = r.search(j["queries"], top_k=int(k))
            exp = reg[name]["results"][k]
            assert list(got.keys()) == list(exp.keys())
            for qid in exp:
                g, e = got[qid], exp[qid]
                assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), [row[d] for d in e],
                                    np.array(list(e.values()), np.float32), k=int(k), label=f"{name} k={k} {qid}")
        r.close()
    assert isinstance(rx.RetrieverRegistry.create({"type": "dpr"}), rx.QuantizedEmbeddingRetriever)  # retriever_registry.py:588-592
    assert rx.RetrieverRegistry.create({"type": "dpr", "params": {"quantization_method": "asymmetric"}}).quantization_method == "asymmetric"
    with pytest.raises(ValueError):
        rx.RetrieverRegistry.create({"type": "nope"})


def _assert_dict_results(got, exp, row, k, label):
    assert list(got.keys()) == list(exp.keys()), label
    for qid in exp:
        g, e = got[qid], exp[qid]
        assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), [row[d] for d in e],
                            np.array(list(e.values()), np.float32), k=k, label=f"{label} {qid}")


@pytest.mark.parametrize("name,cfg", [("bm25", {"type": "bm25", "params": {"k1": 1.2, "b": 0.75}}),
                                      ("bm25_custom", {"type": "bm25_custom", "params": {"k1": 1.6, "b": 0.8}}),
                                      ("splade", {"type": "splade"}), ("dpr", {"type": "dpr"})])
def test_pipeline_twin_golden(rx, golden_dir, tmp_path, name, cfg):
    """registry.OptimizedRetriever against the ``search`` results the REFERENCE's OptimizedRetriever produced
    (evaluate_rag_pipeline.py:162-479; tests/golden/pipeline_small.json): bm25 types score with BM25, every other type
    with the tf-idf dot product and idf = log(N/(df+1)).  Scores must be bit-equal -- the twin's NumPy fallback adds a
    doc's contributions in query-token order, which the engine reproduces (accumulation="token") -- and ranks equal
    modulo the reference's unspecified order inside exact ties.  Three ways to the index: built from the text, loaded
    from the cache this build wrote, loaded from the .npz the reference itself wrote."""
    import shutil
    with open(os.path.join(golden_dir, "pipeline_small.json"), encoding="utf-8") as f:
        pj = json.load(f)
    with open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8") as f:
        j = json.load(f)
    hw = {"memory_gb": 8, "cores": 4}
    cache = tmp_path / "cache"
    r = rx.OptimizedRetriever(cfg, hw, tile_log2=6, cache_dir=str(cache))
    assert (r.k1, r.b) == (pj[name]["k1"], pj[name]["b"])
    assert r.mode == ("bm25" if name.startswith("bm25") else "dot")
    r.build_index_from_corpus(j["corpus"])
    row = {d: i for i, d in enumerate(r.doc_ids)}
    written = list(cache.glob(f"{cfg['type']}_index_*.npz"))
    assert [w.name for w in written] == [f"{cfg['type']}_index_4619a0fc.npz"]  # the reference's file name for this corpus (:189-192)
    fresh = {}
    for k in ("5", "50"):
        fresh[k] = r.search(j["queries"], top_k=int(k))
        _assert_dict_results(fresh[k], pj[name]["results"][k], row, int(k), f"{name} fresh k={k}")
        assert r.search(j["queries"], top_k=int(k)) == fresh[k]  # query-cache hits
    r.close()
    r2 = rx.OptimizedRetriever(cfg, hw, tile_log2=6, cache_dir=str(cache))
    r2.build_index_from_corpus(j["corpus"])  # loads the cache written above
    for k in ("5", "50"):
        assert r2.search(j["queries"], top_k=int(k)) == fresh[k]
    r2.close()
    ref_file = os.path.join(golden_dir, "ref_cache", f"{cfg['type']}_index_4619a0fc.npz")
    if os.path.exists(ref_file):  # bm25 and splade: the cache file written by the reference
        cache3 = tmp_path / "cache_ref"
        cache3.mkdir()
        shutil.copy(ref_file, cache3)
        r3 = rx.OptimizedRetriever(cfg, hw, tile_log2=6, cache_dir=str(cache3))
        r3.build_index_from_corpus(j["corpus"])
        assert sorted(x.name for x in cache3.iterdir()) == [os.path.basename(ref_file)]  # used, not rebuilt
        for k in ("5", "50"):
            got = r3.search(j["queries"], top_k=int(k))
            assert got == fresh[k]
            _assert_dict_results(got, pj[name]["results"][k], row, int(k), f"{name} ref-cache k={k}")
        r3.close()
    # accumulation="term" (the Numba kernels' CSR-row order) == the given-order oracle fed ascending terms
    r4 = rx.OptimizedRetriever(cfg, {"memory_gb": 2, "cores": 4}, tile_log2=6, cache_dir=str(cache), accumulation="term")
    assert r4.query_cache is None  # memory_gb <= 4: no query cache, no index cache (:170-173)
    r4.build_index_from_corpus(j["corpus"])
    h = r4.host
    qids = list(j["queries"])
    q = rx.encode_queries([j["queries"][x] for x in qids], h.vocabulary)
    mode = oracle.MODE_BM25_F32 if name.startswith("bm25") else oracle.MODE_TFIDF_F32
    ed, es, ec = oracle.search_batch(h.indptr, h.indices, h.data, h.doc_lengths, h.idf, *q, 50, r4.k1, r4.b, h.avgdl, mode=mode)
    got = r4.search(j["queries"], top_k=50)
    for i, qid in enumerate(qids):
        assert [row[d] for d in got[qid]] == ed[i, :ec[i]].tolist(), qid
        assert np.array_equal(np.array(list(got[qid].values()), np.float32).view(np.uint32), es[i, :ec[i]].view(np.uint32)), qid
    r4.close()


def test_c1_fiqa_shaped_end_to_end(rx):
    """BASELINE config C1: FiQA-shaped synthetic text (57 638 docs, ~80 k Zipf words, ~130 tokens / doc, 100 queries of
    ~10 words; real FiQA is not available offline) through RetrievalService.build_bm25_index / search_bm25(top_k=10),
    compared bit for bit with the oracle run on the host CSR the service built (same canonical tie order)."""
    from sparse_rx import synth
    corpus, queries = synth.fiqa_shaped_text()
    svc = rx.RetrievalService(device="cuda:0")
    svc.build_bm25_index(corpus)
    h = svc.host
    assert h.n_docs == 57_638
    got = svc.search_bm25(queries, top_k=10)
    assert list(got.keys()) == list(queries.keys())
    qids = list(queries)
    q = rx.encode_queries([queries[x] for x in qids], h.vocabulary)
    ed, es, ec = oracle.search_batch(h.indptr, h.indices, h.data, h.doc_lengths, h.idf, *q, 10, 1.2, 0.75, h.avgdl)
    row = {d: i for i, d in enumerate(h.doc_ids)}
    for i, qid in enumerate(qids):
        g = got[qid]
        assert [row[d] for d in g] == ed[i, :ec[i]].tolist(), qid
        assert np.array_equal(np.array(list(g.values()), np.float32).view(np.uint32), es[i, :ec[i]].view(np.uint32)), qid
    # k = 50 (the reference's FiQA experiment, rag_system/configs/paper_results.yaml:11) and the raw arrays
    gd, gs, gc = svc.dev.search(*q, 50)
    _assert_exact((gd, gs, gc), oracle.search_batch(h.indptr, h.indices, h.data, h.doc_lengths, h.idf, *q, 50, 1.2, 0.75, h.avgdl), "c1 k=50")
    svc.close()


def test_fuzz_shapes_vs_oracle(rx):
    """Randomised shapes: query length 1..64 (all lane-group sizes), correlated terms (many multi-term docs, the
    parking / resolve / dense roll-back paths), units of 1..8 tiles, k from 1 to 128 (tier 1) and above (tier 2)."""
    from sparse_rx import synth
    rng = np.random.default_rng(2025)
    for trial in range(40):
        n_docs = int(rng.integers(3_000, 60_000))
        vocab = int(rng.integers(50, 4_000))
        draws = int(rng.integers(5, 60))
        s = float(rng.choice([0.0, 0.6, 1.0, 1.3]))
        c = synth.zipf_corpus_np(n_docs, vocab, draws, seed=1000 + trial, s=s) if s > 0 else \
            synth.uniform_corpus_np(n_docs, vocab, min(draws, vocab), seed=1000 + trial)
        _, idf, avgdl = synth.corpus_stats(c)
        nq = 40
        terms = int(rng.integers(1, 65))
        q = synth.queries_np(nq, vocab, min(terms, vocab), seed=2000 + trial, dist="zipf" if s > 0 else "uniform", s=max(s, 0.5))
        tile_log2 = int(rng.integers(6, 13))
        for _ in range(3):
            ut = int(rng.integers(1, 9))
            k = int(rng.choice([1, 5, 100, 128, 129, 300]))
            ix = _dev_index(rx, c, idf, avgdl, tile_log2=tile_log2, unit_tiles=ut)  # runs padded for units of ut tiles
            ix.set_opts(target_blocks=int(rng.choice([0, 1, 10_000])))
            _assert_exact(ix.search(*q, k), _oracle_batch(c, idf, avgdl, q, k),
                          f"fuzz trial={trial} docs={n_docs} V={vocab} draws={draws} s={s} terms={terms} tile={tile_log2} ut={ut} k={k}")
            ix.close()


def test_shard_file_save_load(rx, tmp_path):
    """DeviceIndex.save -> DeviceIndex.load (native shard file, chunked upload) searches identically (f32 and f16)."""
    import torch
    from sparse_rx import synth
    c = synth.zipf_corpus_np(30_000, 2_000, 30, seed=11)
    _, idf, avgdl = synth.corpus_stats(c)
    q = synth.queries_np(40, c.vocab, 6, seed=12, dist="zipf")
    for mode, vd in (("bm25", "f32"), ("dot", "f16")):
        a = rx.DeviceIndex.from_csr(c.indptr, c.indices, c.data, idf, doc_lengths=c.doc_lengths, avgdl=avgdl, mode=mode,
                                    val_dtype=vd, tile_log2=11, doc_base=777)
        p = str(tmp_path / f"s_{mode}.srx")
        a.save(p)
        a.close()
        b = rx.DeviceIndex.load(p, chunk_bytes=100_000)  # several chunks per array
        assert (b.n_docs, b.vocab, b.doc_base, b.tile_log2, b.nnz) == (c.n_docs, c.vocab, 777, 11, len(c.indices))
        got = b.search(*q, 50)
        b.close()
        # the loaded index against the ORACLE (not against the index it was saved from); stored values of the Zipf
        # corpus (term counts) are exact in fp16, so the f16 shard has the f32 oracle's scores
        ed, es, ec = _oracle_batch(c, idf, avgdl, q, 50, mode=oracle.MODE_BM25_F32 if mode == "bm25" else oracle.MODE_TFIDF_F32)
        ed = np.where(ed >= 0, ed + 777, -1).astype(np.int32)  # doc_base travels in the file
        _assert_exact(got, (ed, es, ec), f"shard file {mode}/{vd}")
    # a header that disagrees with its arrays never reaches the GPU
    import json as _json
    import struct
    import zlib
    raw = open(p, "rb").read()
    hlen = struct.unpack("<I", raw[12:16])[0]
    hdr = _json.loads(raw[20:20 + hlen])
    hdr["meta"]["vocab"] += 3
    new = _json.dumps(hdr, sort_keys=True).encode()
    assert (20 + len(new) + 4095) // 4096 == (20 + hlen + 4095) // 4096
    ds = (20 + hlen + 4095) // 4096 * 4096
    bad = raw[:8] + struct.pack("<III", struct.unpack("<I", raw[8:12])[0], len(new), zlib.crc32(new) & 0xFFFFFFFF) + new
    open(p, "wb").write(bad + b"\0" * (ds - len(bad)) + raw[ds:])
    with pytest.raises(ValueError, match="header dims require"):
        rx.DeviceIndex.load(p)


def test_corpus_wide_bounds_keep_sharded_search_exact(rx):
    """Shards searched with corpus-wide score bounds (combine_term_bounds of the shards' fine tables) may return fewer
    than k rows each, but the merged result equals the single-index result bit for bit; and the bounds do cut the
    per-shard candidate lists."""
    import torch
    from sparse_rx import synth
    from sparse_rx.index import combine_term_bounds, merge_topk_packed_out_device
    c = synth.uniform_corpus_np(120_000, 3_000, 40, seed=21)
    _, idf, avgdl = synth.corpus_stats(c)
    q = synth.queries_np(64, c.vocab, 8, seed=22)
    k = 100
    whole = _dev_index(rx, c, idf, avgdl, tile_log2=12)
    exp = whole.search(*q, k)
    whole.close()
    for shards in (2, 8):
        bounds = [(c.n_docs * r) // shards for r in range(shards + 1)]
        ixs = []
        for r in range(shards):
            a, b = bounds[r], bounds[r + 1]
            lo, hi = c.indptr[a], c.indptr[b]
            ixs.append(rx.DeviceIndex.from_csr(c.indptr[a: b + 1] - lo, c.indices[lo:hi], c.data[lo:hi], idf,
                                               doc_lengths=c.doc_lengths[a:b], avgdl=avgdl, tile_log2=10, doc_base=a))
        table = combine_term_bounds([ix.fine_bound for ix in ixs], shards)
        qd = [torch.as_tensor(x, device="cuda:0") for x in q]
        rows_local = torch.stack([ix.search_packed_device(*qd, k).clone() for ix in ixs])
        for ix in ixs:
            ix.set_term_bound(table)
        rows_global = torch.stack([ix.search_packed_device(*qd, k).clone() for ix in ixs])
        torch.cuda.synchronize()
        for rows, tag in ((rows_local, "local bounds"), (rows_global, "corpus-wide bounds")):
            out = merge_topk_packed_out_device(rows.contiguous(), k)
            torch.cuda.synchronize()
            _assert_exact((out[:, :k].cpu().numpy(), out[:, k:2 * k].contiguous().view(torch.float32).cpu().numpy(),
                           out[:, 2 * k].cpu().numpy()), exp, f"{tag}, shards={shards}")
        n_local, n_global = int(rows_local[:, :, 2 * k].sum()), int(rows_global[:, :, 2 * k].sum())
        assert n_global <= n_local
        if shards == 8:
            assert n_global < n_local  # shards return only what can still reach the corpus-wide top k
        for ix in ixs:
            ix.close()


def test_host_batch_pipeline(rx):
    """HostBatchPipeline (pinned, double-buffered H2D / D2H on copy streams): several different batches in flight,
    every result equal to the oracle's; misuse (unread slot, oversize batch, bad term id) raises."""
    from sparse_rx import synth
    c = synth.uniform_corpus_np(80_000, 6_000, 30, seed=31)
    _, idf, avgdl = synth.corpus_stats(c)
    ix = _dev_index(rx, c, idf, avgdl, tile_log2=11)
    k = 100
    batches = [synth.queries_np(n, c.vocab, t, seed=40 + i) for i, (n, t) in enumerate([(200, 8), (1, 3), (137, 12), (200, 8), (64, 1), (199, 5)])]
    for zero_copy in (False, True):  # explicit H2D copy of the query CSR / kernels reading the pinned buffer in place
        p2 = rx.HostBatchPipeline(ix, 200, 200 * 12, k, depth=3, zero_copy_queries=zero_copy, zero_copy_results=zero_copy)
        tk2 = [p2.submit(*b) for b in batches[:3]]
        for b, t in zip(batches[:3], tk2):
            _assert_exact(tuple(x.copy() for x in p2.result(t)), _oracle_batch(c, idf, avgdl, b, k), f"pipeline zero_copy={zero_copy}")
        p2.close()
    pipe = rx.HostBatchPipeline(ix, 200, 200 * 12, k, depth=2)
    tickets, got = [], []
    for b in batches:
        tickets.append(pipe.submit(*b))
        if len(tickets) == 2:
            got.append(tuple(x.copy() for x in pipe.result(tickets.pop(0))))
    while tickets:
        got.append(tuple(x.copy() for x in pipe.result(tickets.pop(0))))
    for i, (b, g) in enumerate(zip(batches, got)):
        _assert_exact(g, _oracle_batch(c, idf, avgdl, b, k), f"pipeline batch {i}")
    _assert_exact(pipe.search(*batches[2]), _oracle_batch(c, idf, avgdl, batches[2], k), "pipeline sync search")
    t0 = pipe.submit(*batches[0])
    pipe.submit(*batches[1])
    with pytest.raises(RuntimeError, match="unread"):
        pipe.submit(*batches[2])
    pipe.result(t0)
    with pytest.raises(RuntimeError):
        pipe.result(t0)
    with pytest.raises(ValueError, match="larger"):
        pipe.submit(*synth.queries_np(201, c.vocab, 4, seed=1))
    with pytest.raises(ValueError, match="out of range"):
        pipe.submit(np.array([0, 1], np.int32), np.array([6000], np.int32), np.ones(1, np.float32))
    pipe.close()
    ix.close()


def test_sharded_searcher_overlap_on_one_gpu(rx):
    """The N > 1 code path on ONE GPU (RCCL process group of one rank, force_exchange): packed rows written into the
    send buffer, all-to-all / all-gather, packed merge -- with the exchange of batch i overlapping the scoring of batch
    i+1 on a second stream (double-buffered slots, event guards).  Several different batches back to back without a
    synchronisation in between; after wait() every batch equals the oracle."""
    import socket
    import torch
    import torch.distributed as dist
    from sparse_rx import synth
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = synth.uniform_corpus_np(100_000, 5_000, 30, seed=51)
        _, idf, avgdl = synth.corpus_stats(c)
        ix = _dev_index(rx, c, idf, avgdl, tile_log2=11, doc_base=1000)
        k = 50
        batches = [synth.queries_np(300, c.vocab, 6, seed=60 + i) for i in range(5)]
        dev_batches = [[torch.as_tensor(x, device="cuda:0") for x in b] for b in batches]
        exps = []
        for b in batches:
            ed, es, ec = _oracle_batch(c, idf, avgdl, b, k)
            exps.append((np.where(ed >= 0, ed + 1000, -1).astype(np.int32), es, ec))
        for mode in ("a2a", "allgather"):
            for overlap in (True, False):
                s = rx.ShardedSearcher.for_device_index(ix)
                s.force_exchange, s.overlap, s.mode = True, overlap, mode
                outs = [s.search(*qb, k) for qb in dev_batches]  # no synchronisation between the batches
                s.wait()
                torch.cuda.synchronize()
                for i, (o, e) in enumerate(zip(outs, exps)):
                    _assert_exact(tuple(x.contiguous().cpu().numpy() for x in o), e, f"{mode} overlap={overlap} batch {i}")
        ix.close()
        # The drop-in API on the same RCCL group: RetrievalService(sharded=True) builds its doc range through the collectives
        # of distributed.build_sharded_host_index and searches through ShardedSearcher (srx_search_packed -> all-to-all ->
        # srx_merge_topk_packed_out -> all-gather); the dicts must be the reference's single-process results.
        golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
        with open(os.path.join(golden, "text_small.json"), encoding="utf-8") as f:
            j = json.load(f)
        svc = rx.RetrievalService(device="cuda:0", tile_log2=6, sharded=True)
        svc.build_bm25_index(j["corpus"])
        assert svc._be.searcher is not None and svc.get_stats()["n_gpus"] == 1
        row = {d: i for i, d in enumerate(svc.doc_ids)}
        for kk in ("3", "10", "1000"):
            got = svc.search_bm25(j["queries"], top_k=int(kk))
            exp = j["results"][kk]
            assert list(got.keys()) == list(exp.keys())
            for qid in exp:
                g, e = got[qid], exp[qid]
                assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), [row[d] for d in e],
                                    np.array(list(e.values()), np.float32), k=min(int(kk), len(row)), label=f"sharded service k={kk} {qid}")
        svc.close()
        from test_oracle_golden import _deep_fixture
        corpus, queries, exp = _deep_fixture(golden)  # deeper than one page: srx_search_after_packed through the exchange
        reg = rx.OptimizedBM25Retriever(device="cuda:0", tile_log2=8, sharded=True)
        reg.build_index_from_corpus(corpus)
        row = {d: i for i, d in enumerate(reg.doc_ids)}
        got = reg.search(queries, top_k=5000)
        for qid in queries:
            ed, es = exp[5000][qid]
            g = got[qid]
            assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), ed, es, k=len(row), label=f"sharded deep {qid}")
        reg.close()
    finally:
        dist.destroy_process_group()


def test_duplicate_postings_are_summed_like_scipy(rx):
    """A CSR that lists the same (doc, term) twice or three times: SciPy sums duplicates when the reference assembles its
    matrix (retrieval.py:171-175); the device builder merges them the same way, so the search equals the oracle on the
    canonical (summed) CSR."""
    import oracle
    from scipy.sparse import csr_matrix
    from sparse_rx import synth
    rng = np.random.default_rng(9)
    n_docs, V = 3_000, 400
    rows = rng.integers(0, n_docs, 60_000)
    cols = (rng.zipf(1.3, 60_000) - 1) % V          # hot terms: many repeated (doc, term) pairs
    vals = rng.integers(1, 4, 60_000).astype(np.float32)
    o = np.lexsort((np.arange(len(rows)), rows))     # row-major input order, duplicates NOT adjacent inside a row
    rows, cols, vals = rows[o], cols[o], vals[o]
    indptr = np.zeros(n_docs + 1, np.int64)
    indptr[1:] = np.cumsum(np.bincount(rows, minlength=n_docs))
    m = csr_matrix((vals, (rows, cols)), shape=(n_docs, V), dtype=np.float32)  # sums duplicates
    m.sort_indices()
    assert m.nnz < len(vals)
    dl = np.asarray(m.sum(axis=1)).ravel().astype(np.float32)
    df = np.bincount(m.indices, minlength=V)
    idf = np.log((n_docs - df + 0.5) / (df + 0.5)).astype(np.float32)
    avgdl = float(np.mean(dl))
    ix = rx.DeviceIndex.from_csr(indptr, cols.astype(np.int32), vals, idf, doc_lengths=dl, avgdl=avgdl, tile_log2=8)
    assert ix.nnz == m.nnz
    q = synth.queries_np(64, V, 6, seed=3, dist="zipf", s=1.0)
    gd, gs, gc = ix.search(*q, 20)
    ed, es, ec = oracle.search_batch(m.indptr, m.indices, m.data, dl, idf, q[0], q[1], q[2], 20, 1.2, 0.75, avgdl)
    assert np.array_equal(gc, ec) and np.array_equal(gd, ed) and np.array_equal(gs.view(np.uint32), es.view(np.uint32))
    ix.close()
