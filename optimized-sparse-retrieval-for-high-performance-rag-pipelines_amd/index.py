"""Index construction: the host half mirrors ``RetrievalService.build_bm25_index``
(/root/reference/rag_system/core/retrieval.py:129-201) bit for bit; the device half turns the doc-major CSR
into the term-major inverted index + tile skip table that libsparse_rx.so searches (include/sparse_rx.h).
"""
from __future__ import annotations

import ctypes
import re
from collections import Counter
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi

BLOCK_PAD = 256  # SRX_BLOCK_PAD in include/sparse_rx.h: all-sentinel blocks behind the last run
_TOKEN_RE = re.compile(r"\b\w+\b")


def tokenize(text: str) -> List[str]:
    """``re.findall(r'\\b\\w+\\b', text.lower())`` -- retrieval.py:148 / :236 (Unicode ``\\w``, str.lower())."""
    return _TOKEN_RE.findall(text.lower())


@dataclass
class HostIndex:
    """The state ``build_bm25_index`` leaves on the reference object (retrieval.py:110-117)."""
    indptr: np.ndarray        # int32/int64 [n_docs+1]
    indices: np.ndarray       # int32 [nnz], sorted per row
    data: np.ndarray          # float32 [nnz] term counts (or learned weights)
    doc_lengths: np.ndarray   # float32 [n_docs]
    idf: np.ndarray           # float32 [vocab]
    avgdl: float
    vocabulary: Dict[str, int]
    doc_ids: List[str]

    @property
    def n_docs(self) -> int:
        return len(self.indptr) - 1

    @property
    def vocab_size(self) -> int:
        return len(self.idf)

    @property
    def nnz(self) -> int:
        return int(self.indptr[-1])


def bm25_idf(df: np.ndarray, n_docs: int) -> np.ndarray:
    """``np.log((N - df + 0.5) / (df + 0.5)).astype(np.float32)`` -- retrieval.py:187-189 (f64 log, cast)."""
    return np.log((n_docs - df + 0.5) / (df + 0.5)).astype(np.float32)


def tfidf_idf(df: np.ndarray, n_docs: int) -> np.ndarray:
    """``np.log(N / (df + 1)).astype(np.float32)`` -- evaluate_rag_pipeline.py:273-278."""
    return np.log(n_docs / (df + 1)).astype(np.float32)


def build_host_index(corpus: Dict[str, Dict], idf_kind: str = "bm25") -> HostIndex:
    """Tokenise, sorted vocabulary, CSR f32, doc lengths, idf, avgdl -- retrieval.py:129-201, same order of
    operations so every array is bit-equal to the reference's (pinned by tests/golden/text_small.npz)."""
    from scipy.sparse import csr_matrix

    if not corpus:
        raise ValueError("Empty corpus provided")  # retrieval.py:133-134
    doc_ids = list(corpus.keys())  # row order = dict insertion order (:141)
    doc_tokens: List[List[str]] = []
    vocab_set = set()
    for doc_id in doc_ids:
        doc = corpus[doc_id]
        text = doc.get("text", doc.get("content", doc.get("body", "")))  # :145 (title is never indexed)
        if text:
            toks = tokenize(text)
            doc_tokens.append(toks)
            vocab_set.update(toks)
        else:
            doc_tokens.append([])
    vocabulary = {term: idx for idx, term in enumerate(sorted(vocab_set))}  # :155 code-point order
    n_docs, V = len(doc_tokens), len(vocabulary)
    doc_lengths = np.zeros(n_docs, dtype=np.float32)
    rows: List[int] = []
    cols: List[int] = []
    data: List[float] = []
    for i, toks in enumerate(doc_tokens):
        doc_lengths[i] = len(toks)  # :165 token count incl. repeats
        if toks:
            for term, cnt in Counter(toks).items():
                rows.append(i)
                cols.append(vocabulary[term])
                data.append(float(cnt))
    m = csr_matrix((data, (rows, cols)), shape=(n_docs, V), dtype=np.float32)  # :176-180
    m.sort_indices()
    m.eliminate_zeros()
    df = np.bincount(m.indices, minlength=V)  # :187
    idf = bm25_idf(df, n_docs) if idf_kind == "bm25" else tfidf_idf(df, n_docs)
    avgdl = float(np.mean(doc_lengths))  # :190 (np.mean of f32 -> f32-precision value)
    return HostIndex(indptr=m.indptr, indices=m.indices, data=m.data, doc_lengths=doc_lengths, idf=idf, avgdl=avgdl,
                     vocabulary=vocabulary, doc_ids=doc_ids)


def encode_queries(texts: Sequence[str], vocabulary: Dict[str, int], order: str = "term") -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Query texts -> CSR batch (q_ptr i32, q_term i32 unique per query, q_weight f32 = term count).
    Mirrors retrieval.py:236-252: tokenise, Counter, OOV dropped; a query with no in-vocabulary term gets an
    empty row (the reference returns {} for it).

    The engine adds a doc's contributions in the order the query lists its terms, so ``order`` selects which reference
    call site is reproduced bit for bit: ``"term"`` (ascending term id) is the CSR-row order of ``simd_bm25_score``
    (retrieval.py:55-72) and of ``_numpy_bm25_score`` (:302); ``"token"`` (first occurrence in the query text) is the
    ``relevant_terms`` order of the pipeline twin's ``_numpy_score_documents`` (evaluate_rag_pipeline.py:360-370,
    436-479)."""
    if order not in ("term", "token"):
        raise ValueError(f"order must be 'term' or 'token', got {order!r}")
    q_ptr = np.zeros(len(texts) + 1, dtype=np.int32)
    terms: List[np.ndarray] = []
    weights: List[np.ndarray] = []
    for i, text in enumerate(texts):
        cnt = Counter(tokenize(text)) if text else {}
        pairs = [(vocabulary[t], float(c)) for t, c in cnt.items() if t in vocabulary]  # Counter keeps first-occurrence order
        if order == "term":
            pairs.sort()
        q_ptr[i + 1] = q_ptr[i] + len(pairs)
        if pairs:
            terms.append(np.fromiter((p[0] for p in pairs), dtype=np.int32, count=len(pairs)))
            weights.append(np.fromiter((p[1] for p in pairs), dtype=np.float32, count=len(pairs)))
    q_term = np.concatenate(terms) if terms else np.zeros(0, dtype=np.int32)
    q_weight = np.concatenate(weights) if weights else np.zeros(0, dtype=np.float32)
    return q_ptr, q_term, q_weight


def validate_query_batch(q_ptr, q_term, q_weight, vocab: int) -> None:
    """Preconditions of ``srx_search`` (include/sparse_rx.h), checked on the host before anything is launched:
    q_ptr starts at 0 and never decreases, every term id is in [0, vocab), no term occurs twice inside a query
    (the kernels index term_ptr / tile_skip / idf with the ids and assume one posting per (doc, term)).
    Raises ValueError."""
    q_ptr = np.asarray(q_ptr)
    q_term = np.asarray(q_term)
    if q_ptr.ndim != 1 or len(q_ptr) < 1 or int(q_ptr[0]) != 0 or np.any(np.diff(q_ptr.astype(np.int64)) < 0):
        raise ValueError("q_ptr must start at 0 and be non-decreasing")
    n = int(q_ptr[-1])
    if len(q_term) < n or len(np.asarray(q_weight)) < n:
        raise ValueError("q_term / q_weight are shorter than q_ptr[-1]")
    t = q_term[:n]
    if n and (int(t.min()) < 0 or int(t.max()) >= vocab):
        raise ValueError(f"q_term out of range [0, {vocab})")
    if n > 1:
        # fast path (ascending rows, what encode_queries(order="term") emits): strictly increasing inside every row
        inc = t[1:] > t[:-1]
        starts = q_ptr[1:-1]
        starts = starts[(starts > 0) & (starts < n)]
        inc[starts - 1] = True  # the step across a row boundary may go down
        if not inc.all():
            row = np.repeat(np.arange(len(q_ptr) - 1), np.diff(q_ptr))
            o = np.lexsort((t, row))
            if np.any((row[o][1:] == row[o][:-1]) & (t[o][1:] == t[o][:-1])):
                raise ValueError("a query lists the same term twice (merge duplicates into one weight)")


# ---------------------------------------------------------------------------------------------------------
# device index
# ---------------------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _ptr(t) -> int:
    return 0 if t is None or t.numel() == 0 else t.data_ptr()


def _stream_ptr(torch, device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class DeviceIndex:
    """One doc-range shard resident in HBM: term-major postings + tile skip table + idf, plus the
    ``srx_index`` handle.  All tensors are owned here (PyTorch-ROCm is the allocator); the library only
    keeps pointers."""

    def __init__(self, term_ptr, post, tile_skip, idf, n_docs: int, vocab: int, doc_base: int, tile_log2: int, device, *,
                 nnz: int, n_blocks: int, unit_tiles: int, val_type: int, term_bound=None, fine_bound=None, keep_canonical: bool = True):
        torch = _torch()
        self.device = torch.device(device)
        self.term_ptr, self.post, self.tile_skip, self.idf = term_ptr, post, tile_skip, idf
        self.term_bound = term_bound
        self.fine_bound = fine_bound  # [V, len(FINE_KS)]: the same statistic at more ranks K (sharded deployments combine these)
        self.n_docs, self.vocab, self.doc_base, self.tile_log2 = int(n_docs), int(vocab), int(doc_base), int(tile_log2)
        self.n_tiles = (self.n_docs + (1 << tile_log2) - 1) >> tile_log2
        self.nnz, self.n_blocks, self.unit_tiles, self.val_type = int(nnz), int(n_blocks), int(unit_tiles), int(val_type)
        words = 8 if self.val_type == _capi.SRX_VAL_F32 else 6
        if post.dtype != torch.int32 or post.numel() < (self.n_blocks + BLOCK_PAD) * words:
            raise ValueError("post must be int32 and hold n_blocks + SRX_BLOCK_PAD blocks")
        self._h = None
        self.post16 = self._build_compact()
        self._create_handle(term_bound)
        self._ws = None
        self._opts = _capi.SearchOpts()
        if not keep_canonical:
            self.drop_canonical()

    @property
    def value_bytes(self) -> int:
        return 4 if self.val_type == _capi.SRX_VAL_F32 else 2

    UNIT_MAX_DOCS = 49152  # W_UNIT_MAX_DOCS in csrc/srx_common.h

    def _build_compact(self):
        """The compact copy of the blocks the tier-1 kernel streams (16-bit unit-local doc ids: 6 / 4 bytes per posting
        instead of 8 / 6; ``srx_build_compact``).  Derived data: never stored, rebuilt from ``post`` whenever an index is
        created or loaded.  None when a unit covers more docs than 16-bit local ids allow (tier 2 then serves everything)."""
        torch = _torch()
        if (self.unit_tiles << self.tile_log2) > self.UNIT_MAX_DOCS:
            return None
        cw = 6 if self.val_type == _capi.SRX_VAL_F32 else 4
        total = self.n_blocks + BLOCK_PAD
        with torch.cuda.device(self.device):
            out = torch.empty(total * cw, dtype=torch.int32, device=self.device)
            _capi.check(_capi.lib().srx_build_compact(self.device.index or 0, self.val_type, _ptr(self.post), total, self.tile_log2,
                                                      self.unit_tiles, _ptr(out), _stream_ptr(torch, self.device)), "srx_build_compact")
            torch.cuda.current_stream(self.device).synchronize()
        return out

    def drop_canonical(self) -> bool:
        """Keep ONE copy of the postings: free the canonical blocks (8 / 6 bytes per posting) and let the tier-2 kernel read
        the compact copy (6 / 4 bytes) tier 1 streams -- 57 % of the posting memory on fp32 values, and fewer bytes for tier 2
        to move.  Possible when the compact copy exists (units of <= 49 152 docs).  Afterwards the index cannot be saved to a
        shard file (that format stores the canonical blocks) and cannot be searched with another unit than the one it was
        built for (``set_opts(supertile_log2 / unit_tiles)``).  Returns whether the copy was dropped."""
        torch = _torch()
        if self.post16 is None or self.post is None:
            return False
        torch.cuda.synchronize(self.device)
        self.post = None
        self._create_handle(self.term_bound)
        _capi.check(_capi.lib().srx_index_set_opts(self._h, ctypes.byref(self._opts)), "srx_index_set_opts")
        return True

    def _create_handle(self, table) -> None:
        d = _capi.IndexDesc(device=self.device.index or 0, val_type=self.val_type, n_docs=self.n_docs, vocab=self.vocab,
                            nnz=self.nnz, n_blocks=self.n_blocks, doc_base=self.doc_base, tile_log2=self.tile_log2,
                            n_tiles=self.n_tiles, unit_tiles=self.unit_tiles, reserved0=0, term_ptr=_ptr(self.term_ptr),
                            post=_ptr(self.post), tile_skip=_ptr(self.tile_skip), idf=_ptr(self.idf), term_bound=_ptr(table),
                            post16=_ptr(self.post16))
        h = ctypes.c_void_p()
        _capi.check(_capi.lib().srx_index_create(ctypes.byref(d), ctypes.byref(h)), "srx_index_create")
        if self._h:
            _capi.lib().srx_index_destroy(self._h)
        self._h = h

    def set_term_bound(self, table) -> None:
        """Replace the score-bound table (f32[V, 4], K = 1, 10, 100, 1000).  Any table of valid LOWER bounds of the K-th
        largest stored value of each term keeps the search exact; a sharded deployment installs corpus-wide bounds
        (distributed.global_term_bounds), which spares every shard most of its threshold warm-up."""
        torch = _torch()
        if table is not None:
            table = table.to(device=self.device, dtype=torch.float32).contiguous()
            assert tuple(table.shape) == (self.vocab, len(self.BOUND_KS))
        torch.cuda.synchronize(self.device)
        self.term_bound = table
        self._create_handle(table)
        _capi.check(_capi.lib().srx_index_set_opts(self._h, ctypes.byref(self._opts)), "srx_index_set_opts")

    # -- construction ----------------------------------------------------------------------------------
    @classmethod
    def from_csr(cls, indptr, indices, data, idf, *, doc_lengths=None, k1: float = 1.2, b: float = 0.75,
                 avgdl: float = 1.0, mode: str = "bm25", val_dtype: str = "f32", device="cuda:0", doc_base: int = 0,
                 tile_log2: int = 14, score_bounds: bool = True, unit_tiles: int = 0, keep_canonical: bool = True) -> "DeviceIndex":
        """Build from a doc-major CSR (host numpy or device torch arrays).  ``keep_canonical=False``: one resident copy of the
        postings (see :meth:`drop_canonical`).

        mode "bm25": post_val = impact(tf, len) precomputed in fp32 (retrieval.py:58,70-71), idf as given.
        mode "dot" : post_val = data (optionally fp16), contribution data*idf*qw (evaluate_rag_pipeline.py:117).
        """
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.SparseRxUnavailable("no HIP device visible: DeviceIndex needs a GPU (there is no CPU fallback)")
        dev = torch.device(device)
        L = _capi.lib()

        def to_dev(x, dtype):
            if isinstance(x, torch.Tensor):
                return x.to(device=dev, dtype=dtype)
            return torch.as_tensor(np.ascontiguousarray(x), device=dev).to(dtype)

        with torch.cuda.device(dev):
            indptr_d = to_dev(indptr, torch.int64)
            cols = to_dev(indices, torch.int32)
            vals = to_dev(data, torch.float32)
            n_docs = indptr_d.numel() - 1
            counts = indptr_d[1:] - indptr_d[:-1]
            rows = torch.repeat_interleave(torch.arange(n_docs, device=dev, dtype=torch.int32), counts)
            dl = None if doc_lengths is None else to_dev(doc_lengths, torch.float32)
            return cls.from_coo(rows, cols, vals, to_dev(idf, torch.float32), n_docs, doc_lengths=dl, k1=k1, b=b,
                                avgdl=avgdl, mode=mode, val_dtype=val_dtype, device=dev, doc_base=doc_base,
                                tile_log2=tile_log2, score_bounds=score_bounds, unit_tiles=unit_tiles, keep_canonical=keep_canonical)

    @classmethod
    def from_coo(cls, rows, cols, vals, idf, n_docs: int, *, doc_lengths=None, k1: float = 1.2, b: float = 0.75,
                 avgdl: float = 1.0, mode: str = "bm25", val_dtype: str = "f32", device="cuda:0", doc_base: int = 0,
                 tile_log2: int = 14, score_bounds: bool = True, unit_tiles: int = 0, keep_canonical: bool = True) -> "DeviceIndex":
        """Build from device COO triples sorted by (row, col) -- i.e. the CSR's nnz order with explicit rows
        (rows i32 shard-local, cols i32, vals f32).  CSR -> CSC is one stable sort by term, which keeps rows
        ascending inside a term; the term-major arrays are then scattered into the blocked layout of
        include/sparse_rx.h (padded runs per unit of ``unit_tiles`` tiles; 0 = srx_auto_unit_tiles)."""
        torch = _torch()
        dev = torch.device(device)
        L = _capi.lib()
        with torch.cuda.device(dev):
            idf_d = idf.to(device=dev, dtype=torch.float32).contiguous()
            V = idf_d.numel()
            nnz = cols.numel()
            cols_sorted = None
            if nnz > 0:
                df = torch.bincount(cols, minlength=V)
                cols_sorted, perm = torch.sort(cols, stable=True)
                post_doc = rows[perm].contiguous()
                tf = vals[perm].contiguous()
                del perm
                # Duplicate (doc, term) entries: SciPy sums them when the reference assembles its CSR
                # (csr_matrix((data, (rows, cols))), retrieval.py:171-175); here they are adjacent after the stable
                # sort (rows ascend inside a term) and are summed the same way, in input order.  The kernels rely on
                # one posting per (doc, term).
                if nnz > 1:
                    dup = (cols_sorted[1:] == cols_sorted[:-1]) & (post_doc[1:] == post_doc[:-1])
                    if bool(dup.any().item()):
                        first = torch.ones(nnz, dtype=torch.bool, device=dev)
                        first[1:] = ~dup
                        starts = torch.nonzero(first).squeeze(1)
                        starts = torch.cat([starts, torch.tensor([nnz], dtype=torch.int64, device=dev)]).contiguous()
                        merged = torch.empty(starts.numel() - 1, dtype=torch.float32, device=dev)
                        _capi.check(L.srx_build_sum_duplicates(dev.index or 0, _ptr(starts), starts.numel() - 1, _ptr(tf), _ptr(merged),
                                                               _stream_ptr(torch, dev)), "srx_build_sum_duplicates")  # left to right, input order
                        torch.cuda.synchronize(dev)
                        tf = merged
                        del starts, merged
                        cols_sorted, post_doc = cols_sorted[first].contiguous(), post_doc[first].contiguous()
                        nnz = cols_sorted.numel()
                        df = torch.bincount(cols_sorted, minlength=V)
                        del first
                    del dup
            else:
                post_doc = torch.zeros(0, dtype=torch.int32, device=dev)
                tf = torch.zeros(0, dtype=torch.float32, device=dev)
                df = torch.zeros(V, dtype=torch.int64, device=dev)
            term_ptr = torch.zeros(V + 1, dtype=torch.int64, device=dev)
            term_ptr[1:] = torch.cumsum(df, 0)
            stream = _stream_ptr(torch, dev)
            if mode == "bm25":
                if doc_lengths is None:
                    raise ValueError("mode='bm25' needs doc_lengths")
                dl = doc_lengths.to(device=dev, dtype=torch.float32).contiguous()
                post_val = torch.empty(nnz, dtype=torch.float32, device=dev)
                _capi.check(L.srx_build_impacts(dev.index or 0, _ptr(tf), _ptr(post_doc), _ptr(dl), nnz, float(k1), float(b),
                                                float(avgdl), _ptr(post_val), stream), "srx_build_impacts")
                torch.cuda.synchronize(dev)
                del tf
            elif mode == "dot":
                post_val = tf.to(torch.float16) if val_dtype == "f16" else tf
            else:
                raise ValueError(f"unknown mode {mode!r}")
            fine_bound = cls._term_bounds(torch, cols_sorted, post_val, term_ptr, df, V) if score_bounds else None
            term_bound = None if fine_bound is None else fine_bound[:, [cls.FINE_KS.index(K) for K in cls.BOUND_KS]].contiguous()
            n_tiles = (n_docs + (1 << tile_log2) - 1) >> tile_log2
            # ---- unpadded tile skip table, then the blocked layout ----
            skip = torch.empty(V * (n_tiles + 1), dtype=torch.int32, device=dev)
            _capi.check(L.srx_build_tile_skip(dev.index or 0, _ptr(term_ptr), _ptr(post_doc), V, n_tiles, tile_log2,
                                              _ptr(skip), stream), "srx_build_tile_skip")
            if unit_tiles <= 0:
                unit_tiles = _capi.check(L.srx_auto_unit_tiles(n_docs, V, nnz, tile_log2), "srx_auto_unit_tiles")
            n_units = (n_tiles + unit_tiles - 1) // unit_tiles
            edges = torch.arange(0, n_units + 1, device=dev, dtype=torch.int64).mul_(unit_tiles).clamp_(max=n_tiles)
            at_units = skip.view(V, n_tiles + 1)[:, edges]                      # [V, n_units + 1]
            run_len = (at_units[:, 1:] - at_units[:, :-1]).to(torch.int64)      # real postings per (term, unit)
            runpad = torch.zeros(V * n_units + 1, dtype=torch.int64, device=dev)
            runpad[1:] = torch.cumsum(((run_len + 3) & ~3).reshape(-1), 0)
            del at_units, run_len, edges
            n_blocks = int(runpad[-1].item()) // 4
            val_type = _capi.SRX_VAL_F16 if post_val.dtype == torch.float16 else _capi.SRX_VAL_F32
            words = 8 if val_type == _capi.SRX_VAL_F32 else 6
            post = torch.empty((n_blocks + BLOCK_PAD) * words, dtype=torch.int32, device=dev)
            tile_skip = torch.empty(V * (n_tiles + 1), dtype=torch.int32, device=dev)
            term_ptr_pad = torch.empty(V + 1, dtype=torch.int64, device=dev)
            _capi.check(L.srx_build_blocks(dev.index or 0, val_type, _ptr(term_ptr), _ptr(cols_sorted), _ptr(post_doc), _ptr(post_val),
                                           _ptr(skip), _ptr(runpad), V, nnz, n_tiles, tile_log2, unit_tiles, _ptr(post),
                                           _ptr(tile_skip), _ptr(term_ptr_pad), n_blocks, stream), "srx_build_blocks")
            torch.cuda.synchronize(dev)
            del cols_sorted, skip, runpad, post_doc, post_val
        return cls(term_ptr_pad, post, tile_skip, idf_d, n_docs, V, doc_base, tile_log2, dev, nnz=nnz, n_blocks=n_blocks,
                   unit_tiles=unit_tiles, val_type=val_type, term_bound=term_bound, fine_bound=fine_bound, keep_canonical=keep_canonical)

    BOUND_KS = (1, 10, 100, 1000)  # the ranks K the engine looks up (include/sparse_rx.h: term_bound[vocab*4])
    FINE_KS = (1, 2, 4, 8, 10, 16, 32, 64, 100, 128, 256, 512, 1000, 1024)  # ... and the ranks kept for combining shards

    @staticmethod
    def _term_bounds(torch, cols_sorted, post_val, term_ptr, df, V):
        """out[t, j] = the K_j-th largest stored value of term t for K_j in FINE_KS (0 if it has fewer than K_j positive
        values), or None when some value is negative.  ``srx_build_term_bounds``: one streaming pass over the term-major
        values (post_val f32 / f16, term_ptr the unpadded run starts); exact."""
        nnz = post_val.numel()
        if nnz == 0 or cols_sorted is None:
            return None
        dev = post_val.device
        ks = torch.tensor(DeviceIndex.FINE_KS, dtype=torch.int32, device=dev)
        out = torch.empty((V, len(DeviceIndex.FINE_KS)), dtype=torch.float32, device=dev)
        neg = torch.zeros(1, dtype=torch.int32, device=dev)
        val_type = _capi.SRX_VAL_F16 if post_val.dtype == torch.float16 else _capi.SRX_VAL_F32
        _capi.check(_capi.lib().srx_build_term_bounds(dev.index or 0, val_type, _ptr(term_ptr), _ptr(post_val.contiguous()), V, _ptr(ks),
                                                      len(DeviceIndex.FINE_KS), _ptr(out), _ptr(neg), _stream_ptr(torch, dev)),
                    "srx_build_term_bounds")
        if int(neg.item()) != 0:
            return None
        return out

    @classmethod
    def from_host_index(cls, hi: HostIndex, k1: float = 1.2, b: float = 0.75, **kw) -> "DeviceIndex":
        return cls.from_csr(hi.indptr, hi.indices, hi.data, hi.idf, doc_lengths=hi.doc_lengths, k1=k1, b=b,
                            avgdl=hi.avgdl, **kw)

    # -- native shard file -------------------------------------------------------------
    def save(self, path: str) -> None:
        """Write this shard (blocked postings, skip table, idf, bounds) to a native shard file (shardfile.py)."""
        from . import shardfile
        torch = _torch()
        if self.post is None:
            raise ValueError("this index dropped its canonical blocks (drop_canonical): the shard file stores them -- save before dropping")
        torch.cuda.synchronize(self.device)
        arrays = {n: getattr(self, n).cpu().numpy() for n in ("term_ptr", "post", "tile_skip", "idf")}
        if self.term_bound is not None:
            arrays["term_bound"] = self.term_bound.cpu().numpy().reshape(-1)
        if self.fine_bound is not None:
            arrays["fine_bound"] = self.fine_bound.cpu().numpy().reshape(-1)
        shardfile.write_shard_file(path, arrays, {"n_docs": self.n_docs, "vocab": self.vocab, "nnz": self.nnz,
                                                  "n_blocks": self.n_blocks, "doc_base": self.doc_base,
                                                  "tile_log2": self.tile_log2, "unit_tiles": self.unit_tiles,
                                                  "val_type": self.val_type, "block_pad": BLOCK_PAD})

    @classmethod
    def load(cls, path: str, device="cuda:0", doc_base=None, verify: bool = True, chunk_bytes: int = 1 << 28,
             keep_canonical: bool = True) -> "DeviceIndex":
        """Read a native shard file: memory-map it, validate it (shardfile.read_shard_file) and stream every array to
        the GPU in chunks."""
        from . import shardfile
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.SparseRxUnavailable("no HIP device visible: DeviceIndex needs a GPU (there is no CPU fallback)")
        meta, arr = shardfile.read_shard_file(path, verify=verify)
        if int(meta["block_pad"]) < BLOCK_PAD:
            raise ValueError(f"{path}: written with block_pad {meta['block_pad']}, this build needs {BLOCK_PAD}")
        dev = torch.device(device)

        def up(a):
            t = torch.empty(a.shape[0], dtype=getattr(torch, a.dtype.name), device=dev)
            step = max(1, chunk_bytes // max(a.dtype.itemsize, 1))
            for i in range(0, a.shape[0], step):
                t[i: i + step].copy_(torch.from_numpy(np.ascontiguousarray(a[i: i + step])))
            return t

        with torch.cuda.device(dev):
            tb = up(arr["term_bound"]).view(-1, len(cls.BOUND_KS)) if "term_bound" in arr else None
            fb = up(arr["fine_bound"]).view(-1, len(cls.FINE_KS)) if "fine_bound" in arr else None
            return cls(up(arr["term_ptr"]), up(arr["post"]), up(arr["tile_skip"]), up(arr["idf"]), int(meta["n_docs"]),
                       int(meta["vocab"]), int(meta["doc_base"] if doc_base is None else doc_base), int(meta["tile_log2"]), dev,
                       nnz=int(meta["nnz"]), n_blocks=int(meta["n_blocks"]), unit_tiles=int(meta["unit_tiles"]),
                       val_type=int(meta["val_type"]), term_bound=tb, fine_bound=fb, keep_canonical=keep_canonical)

    # -- search ----------------------------------------------------------------------------------------
    def set_opts(self, supertile_log2: int = 0, target_blocks: int = 0, profile: int = 0, debug: int = 0,
                 unit_tiles: int = 0) -> None:
        """profile = N > 0: every N-th search is bracketed with hipEvents (True = every search)."""
        if self.post is None and (supertile_log2 or unit_tiles):
            raise ValueError("this index keeps no canonical blocks (drop_canonical): it is searched with the unit it was built for")
        self._opts = _capi.SearchOpts(supertile_log2=supertile_log2, target_blocks=target_blocks, profile=int(profile),
                                      reserved=int(debug), unit_tiles=int(unit_tiles))
        _capi.check(_capi.lib().srx_index_set_opts(self._h, ctypes.byref(self._opts)), "srx_index_set_opts")

    def workspace_bytes(self, nq: int, k: int) -> int:
        return _capi.check(_capi.lib().srx_search_workspace_bytes(self._h, nq, k), "srx_search_workspace_bytes")

    def search_device(self, q_ptr, q_term, q_weight, k: int, out=None, after=None):
        """Batched search on device tensors (q_ptr i32[nq+1], q_term i32, q_weight f32).
        Returns (doc i32[nq,k], score f32[nq,k], count i32[nq]) device tensors; asynchronous on the current stream.
        Precondition (NOT checked here, the tensors never leave the device): q_ptr starts at 0 and is non-decreasing,
        0 <= q_term < vocab, no term twice inside a query -- ``validate_queries`` / ``search`` check host batches.
        ``after`` = (doc i32[nq] GLOBAL ids, score f32[nq]) device tensors: ``srx_search_after`` -- only docs ranked
        strictly after that row in (score desc, doc asc) order (the next page of a ranking deeper than max_k)."""
        torch = _torch()
        nq = q_ptr.numel() - 1
        if not (1 <= k <= _capi.limits()["max_k"]):
            raise ValueError(f"top_k must be in [1, {_capi.limits()['max_k']}] for the HIP engine, got {k}")
        with torch.cuda.device(self.device):
            if out is None:
                out = (torch.empty((nq, k), dtype=torch.int32, device=self.device),
                       torch.empty((nq, k), dtype=torch.float32, device=self.device),
                       torch.empty((nq,), dtype=torch.int32, device=self.device))
            need = self.workspace_bytes(nq, k)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
            if after is None:
                rc = _capi.lib().srx_search(self._h, _ptr(q_ptr), _ptr(q_term), _ptr(q_weight), nq, k, _ptr(out[0]), _ptr(out[1]),
                                            _ptr(out[2]), _ptr(self._ws), self._ws.numel(), _stream_ptr(torch, self.device))
            else:
                a_doc, a_score = after
                assert a_doc.dtype == torch.int32 and a_score.dtype == torch.float32 and a_doc.numel() == nq == a_score.numel()
                rc = _capi.lib().srx_search_after(self._h, _ptr(q_ptr), _ptr(q_term), _ptr(q_weight), nq, k, a_doc.data_ptr(),
                                                  a_score.data_ptr(), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(self._ws),
                                                  self._ws.numel(), _stream_ptr(torch, self.device))
            _capi.check(rc, "srx_search")
        return out

    def search_packed_device(self, q_ptr, q_term, q_weight, k: int, out=None, stream=None, workspace=None, after=None):
        """``srx_search_packed``: the same search, each query's result written as one row
        [k doc ids][k score bit patterns][count] of ``out`` (i32[nq, 2k+1], contiguous) -- the exchange format of the
        sharded search, so no packing kernel runs.  Returns ``out``.  ``stream`` (a torch stream, default: the current
        one) and ``workspace`` (a uint8 tensor of ``workspace_bytes`` bytes, default: the index's own) let several
        searches of one index be in flight on different streams."""
        torch = _torch()
        nq = q_ptr.numel() - 1
        if not (1 <= k <= _capi.limits()["max_k"]):
            raise ValueError(f"top_k must be in [1, {_capi.limits()['max_k']}] for the HIP engine, got {k}")
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((nq, 2 * k + 1), dtype=torch.int32, device=self.device)
            assert out.dtype == torch.int32 and tuple(out.shape) == (nq, 2 * k + 1) and out.is_contiguous()
            if workspace is None:
                need = self.workspace_bytes(nq, k)
                if self._ws is None or self._ws.numel() < need:
                    self._ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
                workspace = self._ws
            sp = stream.cuda_stream if stream is not None else _stream_ptr(torch, self.device)
            if after is None:
                rc = _capi.lib().srx_search_packed(self._h, _ptr(q_ptr), _ptr(q_term), _ptr(q_weight), nq, k, _ptr(out),
                                                   _ptr(workspace), workspace.numel(), sp)
            else:
                a_doc, a_score = after
                assert a_doc.dtype == torch.int32 and a_score.dtype == torch.float32 and a_doc.numel() == nq == a_score.numel()
                rc = _capi.lib().srx_search_after_packed(self._h, _ptr(q_ptr), _ptr(q_term), _ptr(q_weight), nq, k, a_doc.data_ptr(),
                                                         a_score.data_ptr(), _ptr(out), _ptr(workspace), workspace.numel(), sp)
            _capi.check(rc, "srx_search_packed")
        return out

    def validate_queries(self, q_ptr: np.ndarray, q_term: np.ndarray, q_weight: np.ndarray) -> None:
        validate_query_batch(q_ptr, q_term, q_weight, self.vocab)

    def search(self, q_ptr: np.ndarray, q_term: np.ndarray, q_weight: np.ndarray, k: int):
        """Host arrays in, host arrays out (doc, score, count).  The batch is validated first (``validate_queries``);
        ``search_device`` trusts its device tensors.  Any k >= 1: a ranking deeper than the engine's list capacity
        (``max_k`` = 1024) is paged with ``srx_search_after`` (:func:`deep_search`)."""
        torch = _torch()
        nq = len(q_ptr) - 1
        self.validate_queries(q_ptr, q_term, q_weight)
        if k < 1:
            raise ValueError(f"top_k must be >= 1, got {k}")
        if nq == 0:
            return (np.zeros((0, k), np.int32), np.zeros((0, k), np.float32), np.zeros(0, np.int32))
        dev = self.device
        qp = torch.as_tensor(np.ascontiguousarray(q_ptr, dtype=np.int32), device=dev)
        qt = torch.as_tensor(np.ascontiguousarray(q_term, dtype=np.int32), device=dev)
        qw = torch.as_tensor(np.ascontiguousarray(q_weight, dtype=np.float32), device=dev)
        d, s, c = deep_search(self.search_device, qp, qt, qw, k, _capi.limits()["max_k"])
        torch.cuda.synchronize(dev)
        return d.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()

    def profile_read(self):
        """Average kernel durations (ms) over the profiled searches since the last read."""
        ms = (ctypes.c_float * 4)()
        n = _capi.lib().srx_profile_read(self._h, ms)
        if n < 0:  # nothing was profiled since the last read
            return {"wave_ms": 0.0, "block_ms": 0.0, "merge_ms": 0.0, "total_ms": 0.0, "calls": 0}
        return {"wave_ms": ms[0], "block_ms": ms[1], "merge_ms": ms[2], "total_ms": ms[3], "calls": n}

    def device_bytes(self) -> int:
        """Everything this shard keeps resident: both copies of the postings (canonical 8 / 6 bytes per posting + the
        compact tier-1 copy 6 / 4), the skip table, idf and the bound tables."""
        ts = [self.term_ptr, self.post, self.post16, self.tile_skip, self.idf, self.term_bound, self.fine_bound]  # post is None after drop_canonical
        return sum(t.numel() * t.element_size() for t in ts if t is not None)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib().srx_index_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class HostBatchPipeline:
    """Host batches in, host rows out, with the PCIe traffic off the critical path (the metric's batch
    wall time includes the H2D of the query batch and the D2H of the nq x k results).

    ``depth`` slots (3 = triple buffering), each with pinned host staging for the query CSR (q_ptr | q_term | q_weight
    packed into one int32 block), a device block of packed result rows [k doc ids][k score bits][count] written
    directly by ``srx_search_packed`` and a pinned host block for them.  The searches run back to back on the caller's
    current stream; each batch's rows go to the host with ONE ``hipMemcpyAsync`` (``srx_memcpy_async``) on a copy stream
    ordered behind its search by an event, so batch i+1 is scored while batch i's rows travel.  By default the kernels
    read the query block straight from the pinned (device-mapped) staging buffer (``zero_copy_queries``): a few hundred
    KB touched once, no copy call at all.  (Measured on this stack, tools/copy_issue_probe.py: a stream-guarded
    ``Tensor.copy_`` costs ~1 ms of host time per call, launching the search on a side stream ~0.8 ms, a C-ABI async
    copy ~0.1 ms, an event record a few microseconds.)  ``submit`` returns a ticket at once; ``result(ticket)`` waits
    for that batch's copy and returns NumPy views of the pinned rows (valid until the slot is reused ``depth`` submits
    later).  ``zero_copy_results`` (default): the result rows are written by the kernels directly into the pinned host
    block, no D2H copy call either; ``False`` keeps the device block + one ``hipMemcpyAsync`` on the copy stream."""

    def __init__(self, index: "DeviceIndex", max_queries: int, max_terms: int, k: int, depth: int = 3, validate: bool = True,
                 zero_copy_queries: bool = True, zero_copy_results: bool = True, multi_stream: bool = False):
        torch = _torch()
        if not (1 <= k <= _capi.limits()["max_k"]):
            raise ValueError(f"top_k must be in [1, {_capi.limits()['max_k']}] for the HIP engine, got {k}")
        self.index, self.k, self.depth, self.validate = index, int(k), int(depth), validate
        self.zero_copy = bool(zero_copy_queries)
        self.zero_copy_out = bool(zero_copy_results)
        self.max_queries, self.max_terms = int(max_queries), int(max_terms)
        # multi_stream: every slot searches on a HIP stream of its own with a workspace of its own, so that `depth` batches
        # are in flight on the GPU at once -- what small batches need (a 1 k-query batch fills a fraction of the chip for
        # ~0.05 ms: back to back on one stream the launches, not the kernels, set the rate).  Large batches gain nothing.
        self.multi_stream = bool(multi_stream)
        dev = index.device
        qwords = self.max_queries + 1 + 2 * self.max_terms
        row = 2 * self.k + 1
        self.slots = []
        with torch.cuda.device(dev):
            self.s_copy = torch.cuda.Stream(device=dev)
            for _ in range(self.depth):
                self.slots.append({
                    "h_q": torch.empty(qwords, dtype=torch.int32).pin_memory(), "d_q": torch.empty(qwords, dtype=torch.int32, device=dev),
                    "d_out": torch.empty((self.max_queries, row), dtype=torch.int32, device=dev),
                    "h_out": torch.empty((self.max_queries, row), dtype=torch.int32).pin_memory(),
                    "ev_done": torch.cuda.Event(), "ev_out": torch.cuda.Event(), "busy": False, "nq": 0})
                self.slots[-1]["h_q_np"] = self.slots[-1]["h_q"].numpy()
                if self.multi_stream:
                    self.slots[-1]["stream"] = torch.cuda.Stream(device=dev)
                    self.slots[-1]["ws"] = torch.empty(max(index.workspace_bytes(self.max_queries, self.k), 1 << 16), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
        self._n = 0
        self.host_times = [0.0, 0.0, 0.0, 0.0]  # submit(): validate, staging, search call, events + D2H call (seconds, cumulative)

    def submit(self, q_ptr: np.ndarray, q_term: np.ndarray, q_weight: np.ndarray) -> int:
        import time as _time
        torch = _torch()
        L = _capi.lib()
        T = self.host_times  # seconds spent in: validate, staging, search call, events + D2H call
        t0 = _time.perf_counter()
        nq, nt = len(q_ptr) - 1, int(q_ptr[-1])
        if nq > self.max_queries or nt > self.max_terms:
            raise ValueError("batch larger than the pipeline was sized for")
        if self.validate:
            validate_query_batch(q_ptr, q_term, q_weight, self.index.vocab)
        t1 = _time.perf_counter()
        ticket = self._n
        s = self.slots[ticket % self.depth]
        if s["busy"]:
            raise RuntimeError("slot still holds an unread result: call result() for the ticket submitted `depth` batches ago")
        hq = s["h_q_np"]
        hq[: nq + 1] = q_ptr
        hq[nq + 1: nq + 1 + nt] = q_term[:nt]
        hq[nq + 1 + nt: nq + 1 + 2 * nt].view(np.float32)[:] = q_weight[:nt]
        n_words = nq + 1 + 2 * nt
        t2 = _time.perf_counter()
        main = s["stream"] if self.multi_stream else torch.cuda.current_stream(self.index.device)
        if self.zero_copy:
            dq = s["h_q"]  # pinned host memory is mapped into the device's address space: the kernels read it in place
        else:
            _capi.check(L.srx_memcpy_async(s["d_q"].data_ptr(), s["h_q"].data_ptr(), 4 * n_words, main.cuda_stream), "srx_memcpy_async")
            dq = s["d_q"]
        out = s["h_out"] if self.zero_copy_out else s["d_out"]
        if nq and self.multi_stream:
            need = self.index.workspace_bytes(nq, self.k)  # not monotone in nq (fewer queries are cut into more splits)
            if s["ws"].numel() < need:
                s["stream"].synchronize()
                s["ws"] = torch.empty(need, dtype=torch.uint8, device=self.index.device)
        if nq:
            self.index.search_packed_device(dq[: nq + 1], dq[nq + 1: nq + 1 + nt], dq[nq + 1 + nt: n_words].view(torch.float32),
                                            self.k, out=out[:nq], stream=main if self.multi_stream else None,
                                            workspace=s["ws"] if self.multi_stream else None)
        t3 = _time.perf_counter()
        if self.zero_copy_out:
            # the kernels wrote the rows straight into the pinned (device-mapped) host block: posted writes over PCIe while
            # the batch is scored, no copy call at all (hipMemcpyAsync costs 0.3 ms of HOST time per call on this stack)
            s["ev_out"].record(main)
        else:
            s["ev_done"].record(main)
            self.s_copy.wait_event(s["ev_done"])
            if nq:
                _capi.check(L.srx_memcpy_async(s["h_out"].data_ptr(), s["d_out"].data_ptr(), 4 * nq * (2 * self.k + 1),
                                               self.s_copy.cuda_stream), "srx_memcpy_async")
            s["ev_out"].record(self.s_copy)
        t4 = _time.perf_counter()
        T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2; T[3] += t4 - t3
        s["busy"], s["nq"] = True, nq
        self._n += 1
        return ticket

    def result(self, ticket: int):
        """(doc i32[nq,k], score f32[nq,k], count i32[nq]) as NumPy views of the slot's pinned rows."""
        s = self.slots[ticket % self.depth]
        if not s["busy"] or ticket < self._n - self.depth:
            raise RuntimeError("ticket already consumed or overwritten")
        s["ev_out"].synchronize()
        s["busy"] = False
        rows = s["h_out"].numpy()[: s["nq"]]
        k = self.k
        return rows[:, :k], rows[:, k:2 * k].view(np.float32), rows[:, 2 * k]

    def search(self, q_ptr, q_term, q_weight):
        """One batch, synchronously (copies of the result views)."""
        d, sc, c = self.result(self.submit(q_ptr, q_term, q_weight))
        return d.copy(), sc.copy(), c.copy()

    def close(self) -> None:
        _torch().cuda.synchronize(self.index.device)
        self.slots = []


def deep_search(search_fn, q_ptr, q_term, q_weight, k: int, page: int):
    """Top-k for ANY k on top of a search limited to ``page`` rows per call: ``search_fn(q_ptr, q_term, q_weight, kk,
    after=None | (doc i32[nq], score f32[nq])) -> (doc [nq, kk], score [nq, kk], count [nq])`` torch tensors.  Page p + 1
    asks for the rows ranked strictly after page p's last row (``srx_search_after``); a query whose page came back short
    is exhausted (its bound becomes score 0: nothing ranks after it).  This is the reference's unbounded ``top_k`` --
    ``argpartition`` for any k, the full ``argsort`` when k >= n_docs (retrieval.py:272-284) -- without a list of that
    size inside the kernels.  Works unchanged on a sharded search (the bound is a GLOBAL (score, doc) row)."""
    torch = _torch()
    if k <= page:
        return search_fn(q_ptr, q_term, q_weight, k)
    nq = q_ptr.numel() - 1
    dev = q_ptr.device
    out_d = torch.full((nq, k), -1, dtype=torch.int32, device=dev)
    out_s = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    out_c = torch.zeros((nq,), dtype=torch.int32, device=dev)
    after = None
    got = 0
    while got < k:
        kk = min(page, k - got)
        d, s, c = search_fn(q_ptr, q_term, q_weight, kk) if after is None else search_fn(q_ptr, q_term, q_weight, kk, after=after)
        out_d[:, got: got + kk] = d
        out_s[:, got: got + kk] = s
        out_c += c.to(torch.int32)
        got += kk
        if got >= k or int((c == kk).sum().item()) == 0:  # every query exhausted: one sync per page, pages are rare
            break
        last = (c.long() - 1).clamp(min=0).unsqueeze(1)
        full = c == kk
        a_doc = torch.where(full, d.gather(1, last).squeeze(1), torch.zeros_like(c, dtype=torch.int32)).to(torch.int32).contiguous()
        a_score = torch.where(full, s.gather(1, last).squeeze(1), torch.zeros_like(c, dtype=torch.float32)).contiguous()
        after = (a_doc, a_score)
    return out_d, out_s, out_c


def merge_topk_device(in_doc, in_score, in_count, k: int, gathered: bool = False):
    """``srx_merge_topk`` on device tensors.  gathered=False: in_doc/in_score [nq, n_lists, k], in_count
    [nq, n_lists]; gathered=True: [n_lists, nq, k] / [n_lists, nq] (the all-gather layout)."""
    torch = _torch()
    if gathered:
        n_lists, nq = in_count.shape
    else:
        nq, n_lists = in_count.shape
    dev = in_doc.device
    L = _capi.lib()
    with torch.cuda.device(dev):
        out = (torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
               torch.empty((nq,), dtype=torch.int32, device=dev))
        need = _capi.check(L.srx_merge_workspace_bytes(nq, n_lists, k), "srx_merge_workspace_bytes")
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        in_doc, in_score, in_count = in_doc.contiguous(), in_score.contiguous(), in_count.contiguous()
        rc = L.srx_merge_topk(dev.index or 0, _ptr(in_doc), _ptr(in_score), _ptr(in_count), nq, n_lists, k, int(gathered),
                              _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(ws), ws.numel(), _stream_ptr(torch, dev))
        _capi.check(rc, "srx_merge_topk")
    return out


def combine_term_bounds(fine_tables, world: int = None):
    """Corpus-wide score bounds from the shards' tables.  ``fine_tables``: tensor [W, V, len(FINE_KS)] (or a list of
    [V, F] tensors), shard r's K-th largest stored value per term for K in DeviceIndex.FINE_KS (0 = fewer than K
    postings).  Returns f32[V, 4] for K in BOUND_KS, each entry a valid LOWER bound of the K-th largest value of the
    term over the whole corpus (docs are disjoint across shards and unique inside a term):

      * max over shards of the shard's own K-th largest (K postings >= it exist in that shard alone), and
      * min over shards of the K'-th largest with K' = the smallest kept rank >= ceil(K / W): every shard then holds
        K' postings >= that value, W * K' >= K in total.

    On statistically similar shards the second is close to the true corpus-wide K-th largest value."""
    torch = _torch()
    t = torch.stack(list(fine_tables)) if isinstance(fine_tables, (list, tuple)) else fine_tables
    W = t.shape[0] if world is None else world
    assert t.shape[0] == W and t.shape[2] == len(DeviceIndex.FINE_KS)
    cols = []
    for K in DeviceIndex.BOUND_KS:
        own = t[:, :, DeviceIndex.FINE_KS.index(K)].max(dim=0).values
        need = -(-K // W)
        Kp = next(k for k in DeviceIndex.FINE_KS if k >= need)
        shared = t[:, :, DeviceIndex.FINE_KS.index(Kp)].min(dim=0).values  # 0 as soon as one shard has fewer than K' postings
        cols.append(torch.maximum(own, shared))
    return torch.stack(cols, dim=1).contiguous()


def pack_results(doc, score, count):
    """(doc i32[nq,k], score f32[nq,k], count i32[nq]) -> one i32 tensor [nq, 2k+1] (row = docs, score bits, count)."""
    torch = _torch()
    nq, k = doc.shape
    out = torch.empty((nq, 2 * k + 1), dtype=torch.int32, device=doc.device)
    out[:, :k] = doc
    out[:, k:2 * k] = score.view(torch.int32)
    out[:, 2 * k] = count
    return out


def merge_topk_packed_out_device(packed, k: int, out=None):
    """``srx_merge_topk_packed_out``: gathered packed rows [n_lists, nq, 2k+1] -> packed rows ``out`` [nq, 2k+1]."""
    torch = _torch()
    n_lists, nq, row = packed.shape
    assert row == 2 * k + 1 and packed.dtype == torch.int32 and packed.is_contiguous()
    dev = packed.device
    L = _capi.lib()
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty((nq, row), dtype=torch.int32, device=dev)
        assert out.dtype == torch.int32 and tuple(out.shape) == (nq, row) and out.is_contiguous()
        need = _capi.check(L.srx_merge_workspace_bytes(nq, n_lists, k), "srx_merge_workspace_bytes")
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        rc = L.srx_merge_topk_packed_out(dev.index or 0, _ptr(packed), nq, n_lists, k, _ptr(out), _ptr(ws), ws.numel(),
                                         _stream_ptr(torch, dev))
        _capi.check(rc, "srx_merge_topk_packed_out")
    return out


def merge_topk_packed_device(packed, k: int):
    """``srx_merge_topk_packed`` on a device tensor [n_lists, nq, 2k+1] (the all-gather of pack_results rows)."""
    torch = _torch()
    n_lists, nq, row = packed.shape
    assert row == 2 * k + 1 and packed.dtype == torch.int32
    dev = packed.device
    L = _capi.lib()
    with torch.cuda.device(dev):
        out = (torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
               torch.empty((nq,), dtype=torch.int32, device=dev))
        need = _capi.check(L.srx_merge_workspace_bytes(nq, n_lists, k), "srx_merge_workspace_bytes")
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        packed = packed.contiguous()
        rc = L.srx_merge_topk_packed(dev.index or 0, _ptr(packed), nq, n_lists, k, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                     _ptr(ws), ws.numel(), _stream_ptr(torch, dev))
        _capi.check(rc, "srx_merge_topk_packed")
    return out
