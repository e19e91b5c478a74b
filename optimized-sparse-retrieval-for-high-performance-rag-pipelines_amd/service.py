"""Drop-in mirror of the reference's ``RetrievalService`` BM25 API
(/root/reference/rag_system/core/retrieval.py:95-506) running on the MI355X HIP engine.

Same names, argument meaning, result shape and error behaviour for the hot path:
``build_bm25_index(corpus)`` (:129), ``search_bm25(queries, top_k)`` (:203), ``get_stats()`` (:471),
``clear_cache()`` (:464), ``close()`` / context manager (:495-506), and the dense ``search_by_vector`` (:402-436).  The
document store (MemoryIndex) is outside this build's scope.

``top_k``: any value, like the reference (``k >= n_docs`` ranks the whole corpus, :281-284): rankings deeper than the
engine's 1024-row lists are paged with ``srx_search_after``; ``top_k <= 0`` returns ``{}`` for every query like the
reference's ``argpartition``.

What changes underneath: ``search_bm25`` scores the WHOLE query dict as one batch through
``srx_search`` (libsparse_rx.so) instead of looping ``simd_bm25_score`` + ``fast_topk_selection`` per query
(:210-229, :256-284).  There is no CPU fallback: without the HIP library / a GPU the calls raise.

Multi-GPU: when the process is a rank of an initialised ``torch.distributed`` group of more than one rank (one process
per GPU, backend "nccl" = RCCL), the SAME two calls shard the index by doc-id range: every rank passes the same corpus /
query dicts, ``build_bm25_index`` builds the rank's doc range with corpus-wide vocabulary, idf and avgdl
(distributed.build_sharded_host_index), ``search_bm25`` scores the batch on every shard, exchanges the per-shard top-k
over RCCL and merges (distributed.ShardedSearcher); every rank returns the dicts the single-GPU service returns
(backend.SparseBackend holds that wiring for this class and the registry mirrors).
"""
from __future__ import annotations

import logging
import os
import threading
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from .backend import SparseBackend
from .index import DeviceIndex, HostIndex, encode_queries

logger = logging.getLogger(__name__)

NUMBA_AVAILABLE = False  # kept for get_stats() key compatibility (retrieval.py:476)


class RetrievalService:
    """BM25 retrieval with the reference's API; scoring + top-k run on the GPU."""

    def __init__(self, index_path=None, embedding_path=None, num_workers: int = 4, cache_size: int = 1000, *,
                 device: Optional[str] = None, tile_log2: int = 14, group=None, sharded: Optional[bool] = None,
                 shard_searcher_factory=None, one_copy: bool = True, **compat_kwargs):
        # index_path / embedding_path / num_workers are accepted for signature compatibility (retrieval.py:98-102);
        # README-only kwargs (use_simd, batch_size, monitor, ...) are swallowed.
        self.index_path = index_path
        self.embedding_path = embedding_path
        self.num_workers = num_workers
        self.cache_size = cache_size
        self.logger = logger
        # group: the torch.distributed group to shard over (None = the default group, when one is initialised);
        # shard_searcher_factory: test hook of backend.SparseBackend (the product never sets it)
        # sharded: None = automatic (a group of more than one rank), True = the sharded path even in a group of one rank
        # one_copy: only the compact copy of the postings stays in HBM (backend.SparseBackend)
        self._be = SparseBackend(device, tile_log2, group=group, searcher_factory=shard_searcher_factory, sharded=sharded, one_copy=one_copy)
        self.device = self._be.device
        self.tile_log2 = tile_log2
        self.k1: float = 1.2   # retrieval.py:116
        self.b: float = 0.75   # retrieval.py:117
        self._built_k1b: Optional[Tuple[float, float]] = None
        self.query_cache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        self.cache_lock = threading.RLock()
        self.build_time = 0.0

    # -- reference attribute names (read-only views) -----------------------------------------------------
    @property
    def host(self) -> Optional[HostIndex]:
        return self._be.host  # sharded: this rank's rows, corpus-wide vocabulary / idf / avgdl / doc ids

    @property
    def dev(self) -> Optional[DeviceIndex]:
        return self._be.dev

    @property
    def vocabulary(self) -> Dict[str, int]:
        return self.host.vocabulary if self.host else {}

    @property
    def doc_ids(self) -> List[str]:
        return self.host.doc_ids if self.host else []

    @property
    def idf_weights(self):
        return self.host.idf if self.host else None

    @property
    def doc_lengths(self):
        return self._be.doc_lengths  # the whole corpus' lengths (sharded: all-gathered at build)

    @property
    def avgdl(self) -> float:
        return self.host.avgdl if self.host else 0.0

    @property
    def corpus_tf(self):
        if self.host is None:
            return None
        from scipy.sparse import csr_matrix
        h = self.host
        return csr_matrix((h.data, h.indices, h.indptr), shape=(h.n_docs, h.vocab_size))

    # -- build ------------------------------------------------------------------------------------------
    def build_bm25_index(self, corpus: Dict[str, Dict]) -> None:
        """retrieval.py:129-201 on the host (bit-equal arrays), then the device inverted index."""
        t0 = time.perf_counter()
        self._be.build(corpus, idf_kind="bm25")
        self._upload()
        self.build_time = time.perf_counter() - t0
        self.logger.info("BM25 index built in %.2fs (%d docs, %d terms, %d postings)", self.build_time,
                         self.host.n_docs, self.host.vocab_size, self.host.nnz)

    def _upload(self) -> None:
        self._be.upload("bm25", self.k1, self.b)
        self._built_k1b = (self.k1, self.b)
        with self.cache_lock:
            self.query_cache.clear()

    # -- search -----------------------------------------------------------------------------------------
    def search_bm25(self, queries: Dict[str, str], top_k: int = 10) -> Dict[str, Dict[str, float]]:
        """retrieval.py:203-231 semantics; one batched GPU call for all uncached queries."""
        if self.host is None:
            raise ValueError("BM25 index not built. Call build_bm25_index() first.")  # :205-206
        # k >= n_docs -> everything, ranked (:281-284); any depth: past its 1024-row lists the engine pages with
        # srx_search_after (index.deep_search)
        k_eff = min(int(top_k), self._be.n_docs_total)
        if k_eff <= 0:  # the reference's argpartition(-scores, 0)[:0] keeps nothing (:276-279)
            return {qid: {} for qid in queries}
        if (self.k1, self.b) != self._built_k1b:
            self._upload()  # k1 / b are plain attributes on the reference (:116-117): impacts depend on them
        results: Dict[str, Dict[str, float]] = {}
        pending: Dict[str, List[str]] = {}  # cache_key -> qids waiting for it
        texts: List[str] = []
        keys: List[str] = []
        for qid, text in queries.items():
            if not text or not text.strip():  # :211-213
                results[qid] = {}
                continue
            key = f"{text.strip()}:{top_k}"  # :216
            with self.cache_lock:
                hit = self.query_cache.get(key)
            if hit is not None:
                results[qid] = self._to_dict(*hit)
                continue
            results[qid] = {}  # keeps the caller's qid order; filled below
            if key not in pending:
                pending[key] = []
                texts.append(text)
                keys.append(key)
            pending[key].append(qid)
        if texts:
            q_ptr, q_term, q_weight = encode_queries(texts, self.host.vocabulary)
            docs, scores, counts = self._be.search_arrays(q_ptr, q_term, q_weight, k_eff)
            for i, key in enumerate(keys):
                if q_ptr[i + 1] == q_ptr[i]:  # no token / no in-vocabulary term -> {} and nothing cached (:237-238, :251-252)
                    continue
                c = int(counts[i])
                entry = (docs[i, :c].astype(np.int64), scores[i, :c].copy())
                with self.cache_lock:
                    if len(self.query_cache) < 1000:  # :288
                        self.query_cache[key] = entry
                d = self._to_dict(*entry)
                for qid in pending[key]:
                    results[qid] = dict(d)
        return results

    def _to_dict(self, idx: np.ndarray, sc: np.ndarray) -> Dict[str, float]:
        ids = self.host.doc_ids
        return {ids[int(i)]: float(s) for i, s in zip(idx, sc) if s > 0}  # :292-296

    # -- misc -------------------------------------------------------------------------------------------
    # -- dense side (retrieval.py:320-339, 402-436) ----------------------------------------------------------
    def set_embeddings(self, embeddings) -> None:
        """Make a [n_docs, dim] float32 matrix (row i = doc_ids[i]) the ``embedding_index`` and put it on the GPU.  The
        reference memory-maps ``embedding_path`` with the row count of its document store (``_load_embeddings``,
        retrieval.py:320-339); the store is out of scope here, so the rows come from the caller or, in
        :meth:`search_by_vector`, from ``embedding_path`` once ``build_bm25_index`` has fixed ``doc_ids``."""
        from .dense import DenseF32Index
        e = np.asarray(embeddings, dtype=np.float32)
        if e.ndim != 2 or (self.doc_ids and e.shape[0] != len(self.doc_ids)):
            raise ValueError("embeddings must be [n_docs, dim] with one row per document")
        self.embedding_index = e
        self._dense = DenseF32Index(e, device=self.device)  # streams a memory map to the device in <= 64 MB chunks

    def search_by_vector(self, query_vector: np.ndarray, k: int = 10, min_score: float = 0.0) -> List[Dict]:
        """retrieval.py:402-436: ``np.dot(embedding_index, query_vector)`` + top-k on the GPU (``srx_dense_search_f32``),
        the ``min_score`` cut and the ``[{"doc_id", "score"}]`` result as in the reference.

        The engine ranks positive values only, so the unshifted search runs first: when it fills all k rows they are all
        > 0 >= ``min_score`` and ARE the reference's top-k -- the common case, exact, no host work.  Only when fewer than k
        docs score above 0 and ``min_score <= 0`` (the default 0.0 admits rows with a score of exactly 0, negative
        thresholds admit negative scores, :425-427) a second pass shifts the scores by a bound on the largest |score| so
        that every doc is rankable; its k rows are re-scored on the host with ``np.dot`` (fp32, the reference's own
        expression) and re-ranked, so the shift never shows in the result.  The shift coarsens that pass's ranking to
        ulp(offset): docs whose true scores differ by less can swap places at the k-th boundary (scores <= 0 only)."""
        if getattr(self, "_dense", None) is None and self.embedding_path and os.path.exists(str(self.embedding_path)) and self.doc_ids:
            n = len(self.doc_ids)
            dim = os.path.getsize(str(self.embedding_path)) // (n * 4)  # retrieval.py:324-328
            self.set_embeddings(np.memmap(str(self.embedding_path), dtype="float32", mode="r", shape=(n, dim)))
        if getattr(self, "_dense", None) is None:
            raise ValueError("No embedding index available")
        q = np.asarray(query_vector, dtype=np.float32)
        kk = max(1, min(int(k), self._dense.n_docs))
        d, s, n = self._dense.search(q, kk)
        if min_score > 0 or int(n[0]) == kk:
            idx, sc = d[0, : int(n[0])].astype(np.int64), s[0, : int(n[0])]
        else:
            # |score| <= |q|_2 * max row norm (Cauchy-Schwarz): the smallest shift that makes every doc rankable keeps
            # the most fp32 resolution for the ranking (the shifted scores are only used to pick the k rows)
            bound = float(np.linalg.norm(q.astype(np.float64))) * self._dense.max_row_norm()
            offset = bound * 1.001 + 1e-30
            d, s, n = self._dense.search(q, kk, score_offset=offset)
            idx = d[0, : int(n[0])].astype(np.int64)
            sc = np.dot(self.embedding_index[idx], q).astype(np.float32)  # the reference's fp32 expression on the k rows
            order = np.argsort(-sc, kind="stable")
            idx, sc = idx[order], sc[order]
        results = []
        for i, score in zip(idx, sc):
            if score < min_score:
                break
            if self.doc_ids and i < len(self.doc_ids):
                results.append({"doc_id": self.doc_ids[int(i)], "score": float(score)})
            elif not self.doc_ids:
                results.append({"doc_id": str(int(i)), "score": float(score)})
        return results

    def clear_cache(self) -> None:
        with self.cache_lock:
            self.query_cache.clear()

    def get_stats(self) -> Dict[str, object]:
        """Same keys as retrieval.py:471-493, plus backend facts."""
        stats: Dict[str, object] = {"cache_size": 0, "query_cache_size": len(self.query_cache),
                                    "numba_available": NUMBA_AVAILABLE, "backend": "hip-gfx950"}
        stats["n_gpus"] = self._be.world()  # doc-range shards (one process per GPU); 1 = the whole index on one card
        if self.host is not None:
            h = self.host  # sharded: this rank's rows -- density and memory are the shard's, num_docs the corpus'
            density = h.nnz / max(1, h.n_docs * h.vocab_size)
            memory_mb = (h.data.nbytes + h.indices.nbytes + h.indptr.nbytes) / (1024 * 1024)
            stats.update({"num_docs": self._be.n_docs_total, "vocab_size": h.vocab_size, "matrix_density": density,
                          "bm25_memory_mb": memory_mb, "avgdl": h.avgdl})
            if self._be.sharded():
                stats["shard_docs"], stats["shard_doc_base"] = h.n_docs, self._be.doc_base
        if self.dev is not None:
            stats["device_index_mb"] = self.dev.device_bytes() / (1024 * 1024)
        return stats

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def close(self) -> None:
        self._be.close()
        self.clear_cache()
