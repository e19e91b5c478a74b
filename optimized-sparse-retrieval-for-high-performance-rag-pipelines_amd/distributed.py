"""Doc-range sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed code at all; this is the build-side addition of row 8e:

  * documents are independent given GLOBAL statistics, so rank r owns the contiguous rows
    [r*n/W, (r+1)*n/W) and builds its inverted index over those rows only, but with the corpus-wide
    df -> idf (retrieval.py:187-189) and avgdl (retrieval.py:190) -- per-shard statistics would change scores;
  * every rank scores the same query batch against its shard (srx_search returns GLOBAL doc ids: doc_base + row);
  * ONE exchange step: all-gather of the per-shard top-k blocks [(doc i32, score f32) x nq x k] + counts
    (nq*k*8 B per rank: 8 MB at 10 k queries, k = 100), then the same exact merge kernel (srx_merge_topk,
    gathered layout) on every rank -> identical result on all ranks and identical to the 1-GPU result.

The payload is tiny against xGMI (7 links x ~153 GB/s per GPU), so the step is latency-bound: one collective per
tensor per batch, never per query.
"""
from __future__ import annotations

import contextlib
from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_docs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous doc-id range [a, b) of `rank` (identical rule on every rank)."""
    return (n_docs * rank) // world, (n_docs * (rank + 1)) // world


def global_df(df_local, group=None):
    """Sum the per-shard document frequencies in place (all-reduce) -> corpus-wide df on every rank."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(df_local, group=group)
    return df_local


def global_avgdl(doc_lengths_local, n_docs_total: int, group=None) -> float:
    """``float(np.mean(doc_lengths))`` over the WHOLE corpus in global doc order (retrieval.py:190): the shards'
    length vectors are all-gathered and reduced on the host by NumPy itself, so the value is the reference's."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return float(np.mean(doc_lengths_local.detach().cpu().numpy().astype(np.float32)))
    world = dist.get_world_size(group)
    sizes = [shard_range(n_docs_total, world, r)[1] - shard_range(n_docs_total, world, r)[0] for r in range(world)]
    m = max(sizes)  # collectives want equal sizes: pad every shard to the largest, cut the padding after the gather
    mine = torch.zeros(m, dtype=doc_lengths_local.dtype, device=doc_lengths_local.device)
    mine[: doc_lengths_local.numel()] = doc_lengths_local
    out = torch.empty(world * m, dtype=doc_lengths_local.dtype, device=doc_lengths_local.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    parts = [out[r * m: r * m + sizes[r]] for r in range(world)]
    return float(np.mean(torch.cat(parts).cpu().numpy().astype(np.float32)))


def global_term_bounds(index, group=None) -> None:
    """Install corpus-wide score bounds on this rank's DeviceIndex: all-gather the shards' fine bound tables (a few MB,
    once, at start-up) and combine them (index.combine_term_bounds).  The search stays exact -- the bounds only have
    to be LOWER bounds of the corpus-wide k-th best score -- but every shard now starts from (nearly) the single-GPU
    threshold instead of its own shard's, so it appends ~1/W of the candidates during warm-up; a shard may then
    return fewer than k rows, which is all the final merge can use."""
    import torch
    import torch.distributed as dist
    from .index import DeviceIndex, combine_term_bounds
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    world = dist.get_world_size(group)
    # The decision to enter the collective must not depend on rank-local state: a shard without a table (no postings,
    # or a negative stored value in dot mode: DeviceIndex._term_bounds) contributes an all-zero one.  0 is a valid lower
    # bound of every K-th largest value, the other shards' own maxima stay valid (docs are disjoint across shards), and
    # a rank that skipped the all-gather would leave the others hanging in it.
    if index.fine_bound is not None:
        mine = index.fine_bound.contiguous()
    else:
        mine = torch.zeros((index.vocab, len(DeviceIndex.FINE_KS)), dtype=torch.float32, device=index.device)
    g = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(g.view(world * mine.shape[0], mine.shape[1]), mine, group=group)
    index.set_term_bound(combine_term_bounds(g, world))


def bm25_idf_from_df(df_global, n_docs_total: int) -> np.ndarray:
    """retrieval.py:187-189 on the corpus-wide df (f64 log, cast to f32)."""
    df = df_global.detach().cpu().numpy() if hasattr(df_global, "detach") else np.asarray(df_global)
    return np.log((n_docs_total - df + 0.5) / (df + 0.5)).astype(np.float32)


def build_sharded_host_index(corpus, idf_kind: str = "bm25", group=None):
    """``build_bm25_index`` (retrieval.py:129-201) under torch.distributed: every rank is handed the SAME corpus dict (the
    reference's API) and does the tokenising / counting of ITS doc range only; the corpus-wide pieces come from three
    small collectives, so that every array equals the matching slice of the single-process index bit for bit:

      * vocabulary -- the ranks' local word sets are all-gathered (objects) and united; ``sorted`` gives the same
        code-point order and therefore the same term ids on every rank (:155);
      * df -> idf  -- all-reduce of the local document frequencies, then the reference's f64 ``log`` (:187-189);
      * avgdl      -- all-gather of the f32 doc lengths in doc order, ``np.mean`` on the host (:190).

    Returns (HostIndex of the local rows [a, b) with GLOBAL term ids / idf / avgdl / vocabulary and ALL doc ids,
    a, n_docs_total, doc_lengths of the whole corpus)."""
    from collections import Counter

    import torch
    import torch.distributed as dist
    from scipy.sparse import csr_matrix

    from .index import HostIndex, bm25_idf, tfidf_idf, tokenize
    if not corpus:
        raise ValueError("Empty corpus provided")  # retrieval.py:133-134
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    doc_ids = list(corpus.keys())  # dict order = row order, identical on every rank
    n_total = len(doc_ids)
    if n_total < world:
        raise ValueError(f"{n_total} docs cannot be sharded over {world} ranks (every shard needs at least one doc)")
    a, b = shard_range(n_total, world, rank)
    doc_tokens = []
    local_vocab = set()
    for doc_id in doc_ids[a:b]:
        doc = corpus[doc_id]
        text = doc.get("text", doc.get("content", doc.get("body", "")))  # :145
        toks = tokenize(text) if text else []
        doc_tokens.append(toks)
        local_vocab.update(toks)
    parts = [None] * world
    dist.all_gather_object(parts, sorted(local_vocab), group=group)
    vocabulary = {t: i for i, t in enumerate(sorted(set().union(*parts)))}  # :155
    V = len(vocabulary)
    doc_lengths = np.zeros(b - a, dtype=np.float32)
    rows, cols, data = [], [], []
    for i, toks in enumerate(doc_tokens):
        doc_lengths[i] = len(toks)  # :165
        for term, cnt in Counter(toks).items():
            rows.append(i)
            cols.append(vocabulary[term])
            data.append(float(cnt))
    m = csr_matrix((data, (rows, cols)), shape=(b - a, max(V, 1)), dtype=np.float32)  # :176-180 on the local rows
    m.sort_indices()
    m.eliminate_zeros()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    df = torch.as_tensor(np.bincount(m.indices, minlength=max(V, 1)).astype(np.int64), device=dev)
    global_df(df, group)
    df = df.cpu().numpy()
    idf = bm25_idf(df, n_total) if idf_kind == "bm25" else tfidf_idf(df, n_total)
    sizes = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world)]
    mx = max(sizes)
    mine = torch.zeros(mx, dtype=torch.float32, device=dev)
    mine[: b - a] = torch.as_tensor(doc_lengths, device=dev)
    allv = torch.empty(world * mx, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(allv, mine, group=group)
    allv = allv.cpu().numpy()
    dl_all = np.concatenate([allv[r * mx: r * mx + sizes[r]] for r in range(world)]).astype(np.float32)
    avgdl = float(np.mean(dl_all))  # :190
    hi = HostIndex(indptr=m.indptr, indices=m.indices, data=m.data, doc_lengths=doc_lengths, idf=idf[:V] if V else idf[:0],
                   avgdl=avgdl, vocabulary=vocabulary, doc_ids=doc_ids)
    return hi, a, n_total, dl_all


class ShardedSearcher:
    """search = local search on this rank's shard -> ONE all-gather of the packed per-shard top-k -> exact merge.

    ``local_search(q_ptr, q_term, q_weight, k) -> (doc i32[nq,k] GLOBAL ids, score f32[nq,k], count i32[nq])``,
    ``pack(doc, score, count) -> i32[nq, 2k+1]`` and ``merge(packed i32[W, nq, 2k+1], k) -> (doc, score, count)`` are
    torch-tensor functions on one device.  The product wiring is :meth:`for_device_index` (HIP engine:
    ``srx_search`` + ``srx_merge_topk_packed``); tests inject CPU callables to exercise the protocol under gloo."""

    def __init__(self, local_search: Callable, pack: Callable, merge: Callable, group=None,
                 local_search_packed: Callable = None, merge_packed_out: Callable = None):
        self.local_search = local_search
        self.pack = pack
        self.merge = merge
        self.group = group
        self._buf = None
        self.mode = "a2a"
        # Optional in-place variants (no packing / unpacking kernels around the exchange):
        # local_search_packed(q_ptr, q_term, q_weight, k, out i32[nq, 2k+1]) and
        # merge_packed_out(packed i32[W, nq, 2k+1], k, out i32[nq, 2k+1]) fill `out` rows [k docs][k score bits][count].
        self.local_search_packed = local_search_packed
        self.merge_packed_out = merge_packed_out

    @classmethod
    def for_device_index(cls, index, group=None) -> "ShardedSearcher":
        from .index import merge_topk_packed_device, merge_topk_packed_out_device, pack_results
        s = cls(index.search_device, pack_results, merge_topk_packed_device, group,
                local_search_packed=index.search_packed_device, merge_packed_out=merge_topk_packed_out_device)
        s.workspace_bytes = index.workspace_bytes  # lets the graph lanes give every lane a workspace of its own
        return s

    def search(self, q_ptr, q_term, q_weight, k: int, chunks: int = 0, q_ptr_host=None, after=None):
        """One batch.  Optionally (chunks > 1) the batch is cut into sub-batches: the all-gather + merge of
        sub-batch i runs on a second HIP stream while sub-batch i+1 is scored (default off: one exchange per batch).  q_ptr_host: the
        same q_ptr on the host (NumPy / CPU tensor), to cut sub-batches without a device-to-host sync."""
        import torch
        import torch.distributed as dist
        exchange = (dist.is_initialized() and dist.get_world_size(self.group) > 1) or getattr(self, "force_exchange", False)
        kw = {} if after is None else {"after": after}  # srx_search_after: the bound is a GLOBAL (doc, score) row, the same on every shard
        if not exchange:
            return self.local_search(q_ptr, q_term, q_weight, k, **kw)
        if after is not None:
            chunks = 1  # deep pages: plain path
        world = dist.get_world_size(self.group)
        nq = q_ptr.shape[0] - 1
        on_gpu = q_ptr.is_cuda
        if chunks <= 0:
            chunks = 1  # measured on one GPU: cutting the batch costs more (under-filled launches) than it can hide
        if (on_gpu and after is None and getattr(self, "graph", False) and self.local_search_packed is not None
                and self.merge_packed_out is not None and getattr(self, "workspace_bytes", None) is not None):
            out = self._search_graph(q_ptr, q_term, q_weight, k, world)
            if out is not None:
                return out
        if (chunks == 1 or not on_gpu) and self.local_search_packed is not None and self.merge_packed_out is not None:
            return self._search_packed(q_ptr, q_term, q_weight, k, world, **kw)
        if chunks == 1 or not on_gpu:
            doc, score, count = self.local_search(q_ptr, q_term, q_weight, k, **kw)
            return self._exchange(doc, score, count, k, world, slot=0)
        # sub-batch boundaries (queries are rows of a CSR: slice q_ptr, rebase, slice terms/weights)
        dev = q_ptr.device
        bounds = [(nq * i) // chunks for i in range(chunks + 1)]
        qp_host = q_ptr_host if q_ptr_host is not None else q_ptr.cpu()  # the term offsets of the cut points
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)
        side = self._side
        side.wait_stream(main)
        outs = []
        for i in range(chunks):
            a, b = bounds[i], bounds[i + 1]
            ta, tb = int(qp_host[a]), int(qp_host[b])
            sub_ptr = (q_ptr[a: b + 1] - ta).contiguous()
            part = self.local_search(sub_ptr, q_term[ta:tb], q_weight[ta:tb], k)  # on the main stream, fresh output tensors
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                outs.append(self._exchange(*part, k, world, slot=i))
                for t in part:
                    t.record_stream(side)
        main.wait_stream(side)
        for o in outs:
            for t in o:
                t.record_stream(main)
        return tuple(torch.cat([o[j] for o in outs]) for j in range(3))

    def _search_graph(self, q_ptr, q_term, q_weight, k: int, world: int):
        """Steady-state submission for a caller that searches the SAME device tensors batch after batch (a serving loop with
        fixed staging buffers; bench.py): the whole step -- srx_search_packed, the RCCL exchange, the packed merge -- is
        captured once per LANE into a HIP graph and replayed.  Two lanes = two HIP streams with buffers and workspace of their
        own, used alternately: the exchange of batch i (lane i & 1) overlaps the scoring of batch i + 1 (the other lane)
        without a single event between them, and a step costs the host one graph launch (measured on one GPU, RCCL group of
        one rank: the eager overlap path needs ~0.25 ms of Python + launches per step -- more than the 0.21 ms the GPU needs).
        The rows of batch i stay valid until lane i & 1 is replayed again (two batches later); :meth:`wait` joins both lanes.
        Returns None (and switches itself off) when the capture is not possible on this stack: the eager path then runs."""
        import torch
        dev = q_ptr.device
        key = (q_ptr.data_ptr(), q_term.data_ptr(), q_weight.data_ptr(), int(q_ptr.shape[0]), int(q_term.shape[0]), k, self.mode, world)
        if getattr(self, "_lanes", None) is None:
            self._lanes, self._lane_step = {}, 0
        lanes = self._lanes.get(key)
        if lanes is None:
            cur = torch.cuda.current_stream(dev)
            lanes = []
            try:
                import torch.distributed as dist
                if getattr(self, "_lane_groups", None) is None:
                    # A communicator of its own per lane: the two lanes' collectives run on different streams with nothing
                    # ordering them against each other, which one RCCL communicator does not allow.  (Collective call: every
                    # rank builds its lanes at the same point of the same first search.)
                    ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(dist.get_world_size()))
                    self._lane_groups = [dist.new_group(ranks=ranks) for _ in range(2)]
                for li in range(2):
                    st = torch.cuda.Stream(device=dev)
                    ws = torch.empty(max(int(self.workspace_bytes(q_ptr.shape[0] - 1, k)), 1 << 16), dtype=torch.uint8, device=dev)
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        for _ in range(2):  # eager warm-up on the lane's stream (RCCL channel setup, lazy allocations) before the capture
                            self._search_packed(q_ptr, q_term, q_weight, k, world, _lane=("lane", li), _workspace=ws, _group=self._lane_groups[li])
                    st.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        out = self._search_packed(q_ptr, q_term, q_weight, k, world, _lane=("lane", li), _workspace=ws, _group=self._lane_groups[li])
                    lanes.append((st, g, out, ws))
            except Exception as e:  # pragma: no cover  (stack without graph-capturable collectives)
                import warnings
                warnings.warn(f"ShardedSearcher: HIP-graph capture of the search step failed ({type(e).__name__}: {e}); using the eager path")
                self.graph = False
                torch.cuda.synchronize(dev)
                return None
            self._lanes[key] = lanes
        st, g, out, _ = lanes[self._lane_step & 1]
        self._lane_step += 1
        with torch.cuda.stream(st):
            g.replay()
        return out

    def _search_packed(self, q_ptr, q_term, q_weight, k: int, world: int, after=None, _lane=None, _workspace=None, _group=None):
        """The exchange of :meth:`_exchange` on packed rows end to end: the local search writes packed rows straight
        into the send buffer, the merge reads the received rows in place and writes packed rows, and the results are
        views of the gathered buffer (doc = rows[:, :k], score = rows[:, k:2k] as f32, count = rows[:, 2k]).

        With ``self.overlap`` (GPU only) the exchange + merge run on a second HIP stream, so that they overlap the
        scoring of the NEXT batch submitted to the main stream (send / receive buffers are double-buffered and guarded
        by events).  The returned tensors are then complete once :meth:`wait` (or a device synchronize) has run."""
        import torch
        import torch.distributed as dist
        nq = q_ptr.shape[0] - 1
        row = 2 * k + 1
        dev = q_ptr.device
        if self._buf is None:
            self._buf = {}
        overlap = bool(getattr(self, "overlap", False)) and q_ptr.is_cuda and _lane is None
        slot = 0 if _lane is None else _lane
        lkw = {} if after is None else {"after": after}
        if _workspace is not None:
            lkw["workspace"] = _workspace
        if overlap:
            self._step = getattr(self, "_step", 0) + 1
            slot = self._step & 1
            cur = torch.cuda.current_stream(dev)
            if getattr(self, "_side", None) is None:
                # Two streams of the searcher's own: the scoring of batch i + 1 and the exchange of batch i only run side by
                # side when NEITHER is the legacy default stream (measured, profiles/r03_exchange_timeline_*: with the search
                # on the default stream every kernel of both streams ran back to back, the "overlap" only added event waits)
                self._score = torch.cuda.Stream(device=dev)
                self._side = torch.cuda.Stream(device=dev)
                self._slot_ev = {}
            main, side = self._score, self._side
            main.wait_stream(cur)  # the caller's query tensors
            ev_prev = self._slot_ev.get(slot)
            if ev_prev is not None:
                main.wait_event(ev_prev)  # the exchange that last used this slot's buffers has finished

        grp = _group if _group is not None else self.group

        def exchange(mine, send, recv):
            if self.mode == "allgather":
                dist.all_gather_into_tensor(recv.view(world * nq, row), mine, group=grp)
                out = torch.empty((nq, row), dtype=torch.int32, device=dev)
                self.merge_packed_out(recv, k, out)
                return out
            blk = recv.shape[1]
            dist.all_to_all_single(recv.view(world * blk, row), send, group=grp)
            merged = torch.empty((blk, row), dtype=torch.int32, device=dev)
            self.merge_packed_out(recv, k, merged)
            allrows = torch.empty((world * blk, row), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(allrows, merged, group=grp)
            return allrows[:nq]

        if self.mode == "allgather":
            key = ("pag", world, nq, k, dev, slot)
            bufs = self._buf.get(key)
            if bufs is None:
                bufs = self._buf[key] = (torch.empty((nq, row), dtype=torch.int32, device=dev),
                                         torch.empty((world, nq, row), dtype=torch.int32, device=dev))
            mine, recv = bufs
            send = mine
            with (torch.cuda.stream(main) if overlap else contextlib.nullcontext()):
                self.local_search_packed(q_ptr, q_term, q_weight, k, mine, **lkw)
        else:
            blk = (nq + world - 1) // world
            key = ("pa2a", world, nq, k, dev, slot)
            bufs = self._buf.get(key)
            if bufs is None:
                bufs = self._buf[key] = (torch.zeros((world * blk, row), dtype=torch.int32, device=dev),  # send; rows >= nq stay empty
                                         torch.empty((world, blk, row), dtype=torch.int32, device=dev))   # the lists of my query block
            send, recv = bufs
            mine = send[:nq]
            with (torch.cuda.stream(main) if overlap else contextlib.nullcontext()):
                self.local_search_packed(q_ptr, q_term, q_weight, k, mine, **lkw)
        if not overlap:
            out = exchange(mine, send, recv)
        else:
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                out = exchange(mine, send, recv)
                done = torch.cuda.Event()
                done.record(side)
            self._slot_ev[slot] = done
            out.record_stream(cur)  # allocated on the side stream, consumed by the caller on its own stream after wait()
            for t in (q_ptr, q_term, q_weight):
                t.record_stream(main)
        return out[:, :k], out[:, k:2 * k].view(torch.float32), out[:, 2 * k]

    def wait(self) -> None:
        """Make the current stream wait for every exchange submitted with ``overlap`` / through the graph lanes (no-op otherwise)."""
        import torch
        side = getattr(self, "_side", None)
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)
        for lanes in (getattr(self, "_lanes", None) or {}).values():
            for st, *_ in lanes:
                torch.cuda.current_stream(st.device).wait_stream(st)

    def close(self) -> None:
        """Join the lanes and drop their captured graphs (they hold RCCL work: destroy them BEFORE the process group)."""
        import torch
        lanes = getattr(self, "_lanes", None)
        if lanes:
            for ls in lanes.values():
                for st, *_ in ls:
                    st.synchronize()
            self._lanes = {}
        if getattr(self, "_side", None) is not None:
            self._side.synchronize()
        self._buf = None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        for g in getattr(self, "_lane_groups", None) or []:  # the lanes' own communicators
            try:
                import torch.distributed as dist
                dist.destroy_process_group(g)
            except Exception:  # pragma: no cover
                pass
        self._lane_groups = None

    def _exchange(self, doc, score, count, k: int, world: int, slot: int):
        """Per-shard top-k -> global top-k.

        mode "a2a" (default): the queries are cut into `world` contiguous blocks; ONE all-to-all sends every rank the
        per-shard lists of ITS block ([world, nq/world, 2k+1] per rank: 1/world of the all-gather traffic and of the
        merge work), it merges them, and ONE all-gather of the merged rows ([nq/world, 2k+1] per rank) replicates the
        final result.  mode "allgather": one all-gather of everything, every rank merges every query."""
        import torch
        import torch.distributed as dist
        nq = count.shape[0]
        row = 2 * k + 1
        mine = self.pack(doc, score, count)  # [nq, 2k+1] i32: nq*(8k+4) bytes per rank
        if self._buf is None:
            self._buf = {}
        mode = self.mode
        if mode == "allgather":
            key = ("ag", world, nq, k, mine.device, slot)
            g = self._buf.get(key)
            if g is None:
                g = self._buf[key] = torch.empty((world, nq, row), dtype=torch.int32, device=mine.device)
            # RCCL over xGMI on the GPU build; output = concatenation along dim 0 (the form every backend accepts)
            dist.all_gather_into_tensor(g.view(world * nq, row), mine, group=self.group)
            return self.merge(g, k)
        # ---- all-to-all: rank j merges query block j ----
        blk = (nq + world - 1) // world
        key = ("a2a", world, nq, k, mine.device, slot)
        bufs = self._buf.get(key)
        if bufs is None:
            bufs = self._buf[key] = (torch.zeros((world * blk, row), dtype=torch.int32, device=mine.device),   # send (padded)
                                     torch.empty((world, blk, row), dtype=torch.int32, device=mine.device),    # lists of my block
                                     torch.empty((world * blk, row), dtype=torch.int32, device=mine.device))   # merged rows, all blocks
        send, recv, allrows = bufs
        send[:nq] = mine  # rows nq .. world*blk-1 stay zero = empty lists (count 0)
        dist.all_to_all_single(recv.view(world * blk, row), send, group=self.group)
        mdoc, mscore, mcount = self.merge(recv, k)  # my block: [blk, k]
        dist.all_gather_into_tensor(allrows, self.pack(mdoc, mscore, mcount), group=self.group)
        out = allrows[:nq]
        return (out[:, :k].contiguous(), out[:, k:2 * k].contiguous().view(torch.float32), out[:, 2 * k].contiguous())
