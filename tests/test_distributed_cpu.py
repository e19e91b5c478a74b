"""N > 1 path under gloo on the CPU (world_size 2, 3 and -- the size of the node the scaling bench runs on -- 8): shard ranges, corpus-wide statistics by all-reduce /
all-gather, the all-gather of per-shard top-k and the exact merge.  The per-shard scoring is done by the CPU oracle
here (tests may use it; the product wiring ShardedSearcher.for_device_index uses the HIP engine and is covered by the
GPU tests + bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import np_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pack(doc, score, count):
    nq, k = doc.shape
    out = torch.empty((nq, 2 * k + 1), dtype=torch.int32)
    out[:, :k] = doc
    out[:, k:2 * k] = score.view(torch.int32)
    out[:, 2 * k] = count
    return out


def _merge_packed(packed, k):
    g_doc = packed[:, :, :k]
    g_score = packed[:, :, k:2 * k].contiguous().view(torch.float32)
    g_count = packed[:, :, 2 * k]
    return _merge_numpy(g_doc, g_score, g_count, k)


def _merge_numpy(g_doc, g_score, g_count, k):
    """Reference merge: union of the per-shard lists, ranked (score desc, doc asc), top k, padded."""
    W, nq, _ = g_doc.shape
    out_d = torch.full((nq, k), -1, dtype=torch.int32)
    out_s = torch.zeros((nq, k), dtype=torch.float32)
    out_c = torch.zeros((nq,), dtype=torch.int32)
    for q in range(nq):
        ds, ss = [], []
        for w in range(W):
            c = int(g_count[w, q])
            ds.append(g_doc[w, q, :c].numpy())
            ss.append(g_score[w, q, :c].numpy())
        d = np.concatenate(ds)
        s = np.concatenate(ss)
        order = np.lexsort((d, -s.astype(np.float64)))[:k]
        out_d[q, : len(order)] = torch.from_numpy(d[order].astype(np.int32))
        out_s[q, : len(order)] = torch.from_numpy(s[order])
        out_c[q] = len(order)
    return out_d, out_s, out_c


def _worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sparse_rx
    from sparse_rx import synth
    c = synth.zipf_corpus_np(4001, 300, 25, seed=5)  # same corpus on every rank; each uses its own row range
    q = synth.queries_np(25, c.vocab, 5, seed=6, dist="zipf")  # not a multiple of the world sizes: padded query blocks
    k = 20
    a, b = sparse_rx.shard_range(c.n_docs, world, rank)
    lo, hi = c.indptr[a], c.indptr[b]
    sub_ptr = c.indptr[a: b + 1] - lo
    # corpus-wide statistics from the shards
    df = torch.from_numpy(np.bincount(c.indices[lo:hi], minlength=c.vocab))
    sparse_rx.global_df(df)
    idf = sparse_rx.bm25_idf_from_df(df, c.n_docs)
    avgdl = sparse_rx.global_avgdl(torch.from_numpy(c.doc_lengths[a:b]), c.n_docs)
    _, idf_ref, avgdl_ref = synth.corpus_stats(c)
    assert np.array_equal(idf.view(np.uint32), idf_ref.view(np.uint32)) and avgdl == avgdl_ref

    def local_search(q_ptr, q_term, q_w, kk):
        d, s, n = oracle.search_batch(sub_ptr, c.indices[lo:hi], c.data[lo:hi], c.doc_lengths[a:b], idf, q_ptr.numpy(),
                                      q_term.numpy(), q_w.numpy(), kk, 1.2, 0.75, avgdl)
        d = np.where(d >= 0, d + a, -1).astype(np.int32)  # doc_base
        return torch.from_numpy(d), torch.from_numpy(s), torch.from_numpy(n)

    if mode.endswith("+packed"):  # the in-place variants the HIP wiring uses (srx_search_packed / srx_merge_topk_packed_out)
        def local_search_packed(q_ptr, q_term, q_w, kk, out):
            out.copy_(_pack(*local_search(q_ptr, q_term, q_w, kk)))
            return out

        def merge_packed_out(packed, kk, out):
            out.copy_(_pack(*_merge_packed(packed, kk)))
            return out

        searcher = sparse_rx.ShardedSearcher(local_search, _pack, _merge_packed, local_search_packed=local_search_packed,
                                             merge_packed_out=merge_packed_out)
    else:
        searcher = sparse_rx.ShardedSearcher(local_search, _pack, _merge_packed)
    searcher.mode = mode.split("+")[0]
    d, s, n = searcher.search(*(torch.from_numpy(x) for x in q), k)
    ed, es, en = oracle.search_batch(c.indptr, c.indices, c.data, c.doc_lengths, idf_ref, q[0], q[1], q[2], k, 1.2, 0.75, avgdl_ref)
    ok = (np.array_equal(n.numpy(), en) and np.array_equal(d.numpy(), ed)
          and np.array_equal(s.numpy().view(np.uint32), es.view(np.uint32)))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "a2a"), (3, "a2a"), (2, "allgather"), (3, "allgather"), (2, "a2a+packed"),
                                        (3, "a2a+packed"), (3, "allgather+packed"), (8, "a2a+packed"), (8, "allgather")])
def test_sharded_search_matches_single_shard(world, mode):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, mode, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {r: True for r in range(world)}


def test_shard_range_partitions():
    import sparse_rx
    for n in (1, 7, 100, 10_000_000):
        for w in (1, 2, 3, 8):
            rs = [sparse_rx.shard_range(n, w, r) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))


def test_combined_term_bounds_are_lower_bounds():
    """combine_term_bounds: for every term and K, the combined value is <= the corpus-wide K-th largest stored value
    (so it is a valid initial threshold on every shard), and on similar shards it beats the per-shard bound."""
    from sparse_rx.index import DeviceIndex, combine_term_bounds
    rng = np.random.default_rng(8)
    V, W = 40, 4
    shards = [[np.sort(rng.random(rng.integers(0, 40) if t < 5 else rng.integers(1500, 3000)).astype(np.float32))[::-1]
               for t in range(V)] for _ in range(W)]  # terms 0-4 are rare (missing ranks), the others similar across shards
    fine = torch.zeros((W, V, len(DeviceIndex.FINE_KS)))
    for r in range(W):
        for t in range(V):
            for j, K in enumerate(DeviceIndex.FINE_KS):
                if len(shards[r][t]) >= K:
                    fine[r, t, j] = float(shards[r][t][K - 1])
    comb = combine_term_bounds(fine, W).numpy()
    better = 0
    for t in range(V):
        allv = np.sort(np.concatenate([shards[r][t] for r in range(W)]))[::-1]
        for j, K in enumerate(DeviceIndex.BOUND_KS):
            true_k = allv[K - 1] if len(allv) >= K else 0.0
            assert comb[t, j] <= true_k, (t, K, comb[t, j], true_k)
            own = max(float(fine[r, t, DeviceIndex.FINE_KS.index(K)]) for r in range(W))
            assert comb[t, j] >= own
            better += comb[t, j] > own
    assert better >= 2 * (V - 5)  # the shared bound is the stronger one for K = 10, 100, 1000 on similar shards


def _bounds_worker(rank, world, port, ret, no_table_rank=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sparse_rx
    from sparse_rx.index import DeviceIndex, combine_term_bounds

    class FakeIndex:  # the collective wiring only needs these members
        def __init__(self, t):
            self.fine_bound, self.table, self.vocab, self.device = t, None, 50, torch.device("cpu")

        def set_term_bound(self, table):
            self.table = table

    tabs = [torch.from_numpy(np.random.default_rng(100 + r).random((50, len(DeviceIndex.FINE_KS))).astype(np.float32)) for r in range(world)]
    if no_table_rank is not None:
        tabs[no_table_rank] = torch.zeros_like(tabs[no_table_rank])  # what a shard without a table contributes
    ix = FakeIndex(None if rank == no_table_rank else tabs[rank])
    sparse_rx.global_term_bounds(ix)  # must not hang when only SOME ranks lack a table (round-1 advisor finding)
    ret[rank] = bool(ix.table is not None and torch.equal(ix.table, combine_term_bounds(tabs, world)))
    dist.destroy_process_group()


@pytest.mark.parametrize("no_table_rank", [None, 1])
def test_global_term_bounds_collective(no_table_rank):
    world = 3
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_bounds_worker, args=(r, world, port, ret, no_table_rank)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=120)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {r: True for r in range(world)}


# ---------------------------------------------------------------------------------------------------------------
# the drop-in API under torch.distributed: RetrievalService / OptimizedBM25Retriever shard by doc range
# ---------------------------------------------------------------------------------------------------------------
def _oracle_searcher_factory(host, doc_base, mode, k1, b, group):
    """TEST backend behind the sharding protocol: the CPU oracle scores this rank's rows (the product wiring puts the HIP
    engine here: backend.SparseBackend.upload)."""
    import sparse_rx
    omode = oracle.MODE_BM25_F32 if mode == "bm25" else oracle.MODE_TFIDF_F32

    def local_search(q_ptr, q_term, q_w, kk, after=None):
        K = kk if after is None else host.n_docs  # a page: rank the shard, keep the rows after the bound
        d, s, n = oracle.search_batch(host.indptr, host.indices, host.data, host.doc_lengths, host.idf, q_ptr.numpy(), q_term.numpy(),
                                      q_w.numpy(), min(K, host.n_docs), k1, b, host.avgdl, mode=omode)
        d = np.where(d >= 0, d + doc_base, -1).astype(np.int32)
        od = np.full((len(n), kk), -1, np.int32)
        os_ = np.zeros((len(n), kk), np.float32)
        on = np.zeros(len(n), np.int32)
        for q in range(len(n)):
            dd, ss = d[q, : n[q]], s[q, : n[q]]
            if after is not None:
                ad, as_ = int(after[0][q]), float(after[1][q])
                keep = (ss < as_) | ((ss == as_) & (dd > ad))
                dd, ss = dd[keep], ss[keep]
            m = min(kk, len(dd))
            od[q, :m], os_[q, :m], on[q] = dd[:m], ss[:m], m
        return torch.from_numpy(od), torch.from_numpy(os_), torch.from_numpy(on)

    return sparse_rx.ShardedSearcher(local_search, _pack, _merge_packed, group)


def _service_worker(rank, world, port, golden_dir, ret):
    import json
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sparse_rx
    from parity import assert_ranked_equal
    from test_oracle_golden import _deep_fixture
    z = np.load(os.path.join(golden_dir, "text_small.npz"))
    with open(os.path.join(golden_dir, "text_small.json"), encoding="utf-8") as f:
        j = json.load(f)
    svc = sparse_rx.RetrievalService(shard_searcher_factory=_oracle_searcher_factory)
    svc.build_bm25_index(j["corpus"])
    # corpus-wide state equals the reference's single-process state; the rows are this rank's slice of it
    a, b = sparse_rx.shard_range(len(j["corpus"]), world, rank)
    assert svc.doc_ids == list(z["doc_ids"]) and svc.avgdl == float(z["avgdl"])
    assert [t for t, _ in sorted(svc.vocabulary.items(), key=lambda kv: kv[1])] == list(z["vocabulary"])
    assert np.array_equal(svc.idf_weights.view(np.uint32), z["idf"].view(np.uint32))
    assert np.array_equal(svc.doc_lengths, z["doc_lengths"]) and np.array_equal(svc.host.doc_lengths, z["doc_lengths"][a:b])
    lo, hi = z["tf_indptr"][a], z["tf_indptr"][b]
    assert np.array_equal(svc.host.indptr, z["tf_indptr"][a: b + 1] - lo)
    assert np.array_equal(svc.host.indices, z["tf_indices"][lo:hi]) and np.array_equal(svc.host.data, z["tf_data"][lo:hi])
    st = svc.get_stats()
    assert st["n_gpus"] == world and st["num_docs"] == len(j["corpus"]) and st["shard_docs"] == b - a
    row = {d: i for i, d in enumerate(svc.doc_ids)}
    qids = list(z["score_qids"])
    for k in ("3", "10", "1000"):
        got = svc.search_bm25(j["queries"], top_k=int(k))
        exp = j["results"][k]
        assert list(got.keys()) == list(exp.keys())
        for qid in exp:
            g, e = got[qid], exp[qid]
            full = z["full_scores"][qids.index(qid)] if qid in qids else None
            assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), [row[d] for d in e],
                                np.array(list(e.values()), np.float32), k=min(int(k), len(row)), full_scores=full, label=f"sharded k={k} {qid}")
        assert svc.search_bm25(j["queries"], top_k=int(k)) == got  # cache hits
    svc.close()
    # the registry twin, and a ranking deeper than one page (2600 docs, k = 1500 / 5000 -> srx_search_after pages)
    corpus, queries, exp = _deep_fixture(golden_dir)
    reg = sparse_rx.OptimizedBM25Retriever(shard_searcher_factory=_oracle_searcher_factory)
    reg.build_index_from_corpus(corpus)
    row = {d: i for i, d in enumerate(reg.doc_ids)}
    for k in (1500, 5000):
        got = reg.search(queries, top_k=k)
        for qid in queries:
            ed, es = exp[k][qid]
            g = got[qid]
            assert_ranked_equal([row[d] for d in g], np.array(list(g.values()), np.float32), ed, es, k=min(k, len(row)), label=f"sharded deep k={k} {qid}")
    reg.close()
    ret[rank] = True
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_retrieval_service_shards_under_torch_distributed(world, golden_dir):
    """RetrievalService.build_bm25_index / search_bm25 (retrieval.py:129-231) called by every rank of a gloo group: the
    corpus-wide vocabulary / idf / avgdl equal the reference's, the rank holds its doc range only, and the dicts every
    rank returns equal the reference's single-process results (tests/golden/text_small.json, text_deep.npz)."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_service_worker, args=(r, world, port, golden_dir, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {r: True for r in range(world)}
