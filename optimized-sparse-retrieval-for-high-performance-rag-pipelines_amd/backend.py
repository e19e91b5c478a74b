"""The engine behind the three API mirrors (``RetrievalService``, ``OptimizedBM25Retriever``, ``OptimizedRetriever``):
host index -> device index -> batched search, on one GPU or doc-range sharded over the ranks of a ``torch.distributed``
group (one process per GPU; backend "nccl" = RCCL over xGMI).

The reference has one process and one index (rag_system/core/retrieval.py:129-231); its API is kept as it is.  When
the calling process is a rank of an initialised group of W > 1 ranks, every rank makes the SAME calls with the SAME
corpus / query dicts (SPMD) and

  * ``build``  tokenises and counts the rank's doc range [r n / W, (r + 1) n / W) only, with corpus-wide vocabulary, idf
    and avgdl (distributed.build_sharded_host_index: bit-equal to the single-process arrays), uploads it with
    ``doc_base`` = the range's first row and installs corpus-wide score bounds (distributed.global_term_bounds);
  * ``search_arrays``  scores the batch on the shard, exchanges the packed per-shard top-k over RCCL and merges them
    exactly (distributed.ShardedSearcher) -- every rank gets the rows the single-GPU index returns.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import _capi
from .index import DeviceIndex, HostIndex, build_host_index, deep_search, validate_query_batch


class SparseBackend:
    def __init__(self, device: Optional[str], tile_log2: int, group=None, searcher_factory=None, sharded: Optional[bool] = None,
                 one_copy: bool = True):
        # searcher_factory(host_index, doc_base, mode, k1, b, group) -> ShardedSearcher on CPU tensors: TEST hook that puts
        # another scorer behind the sharding protocol (the gloo tests inject the CPU oracle).  The product never sets it:
        # without it every search runs on the HIP engine or raises.
        # sharded: None = shard iff the process is a rank of a group of more than one rank; True = take the sharded path even in
        # a group of ONE rank (the whole exchange runs -- RCCL collectives, packed merge -- on one GPU: rehearsals and tests)
        # one_copy (default): only the compact copy of the postings stays resident (DeviceIndex.drop_canonical: -56 % index memory
        # on fp32 values, and the tier-2 kernel moves fewer bytes: C4 -2.5 %, C5 -4 % per batch); False keeps the canonical
        # blocks too, which a search with another unit than the built one (set_opts) or a shard-file save needs
        self.one_copy = bool(one_copy)
        self.group = group
        self._searcher_factory = searcher_factory
        self._force_sharded = bool(sharded)
        if device is None:  # one process per GPU: the launcher's LOCAL_RANK picks the card
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}" if self.world() > 1 else "cuda:0"
        self.device = device
        self.tile_log2 = tile_log2
        self.host: Optional[HostIndex] = None
        self.dev: Optional[DeviceIndex] = None
        self.searcher = None
        self.doc_base = 0
        self.n_docs_total = 0
        self._doc_lengths_all = None

    def world(self) -> int:
        try:
            import torch.distributed as dist
        except ImportError:  # pragma: no cover
            return 1
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def sharded(self) -> bool:
        if self._force_sharded:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise ValueError("sharded=True needs an initialised torch.distributed process group")
            return True
        return self.world() > 1

    @property
    def doc_lengths(self):
        if self._doc_lengths_all is not None:
            return self._doc_lengths_all
        return self.host.doc_lengths if self.host is not None else None

    # -- build ------------------------------------------------------------------------------------------
    def build(self, corpus, idf_kind: str = "bm25") -> HostIndex:
        if self.sharded():
            from .distributed import build_sharded_host_index
            self.host, self.doc_base, self.n_docs_total, self._doc_lengths_all = build_sharded_host_index(corpus, idf_kind, self.group)
        else:
            self.set_host(build_host_index(corpus, idf_kind=idf_kind))
        return self.host

    def set_host(self, host: HostIndex) -> None:
        """A complete host index (built here or read from the .npz cache): the one-shard case."""
        if self.sharded():
            raise ValueError("a pre-built whole-corpus index cannot be adopted by a sharded group: build it from the corpus")
        self.host, self.doc_base, self.n_docs_total, self._doc_lengths_all = host, 0, host.n_docs, None

    def upload(self, mode: str, k1: float, b: float) -> None:
        self.close()
        h = self.host
        if self.sharded() and self._searcher_factory is not None:
            self.searcher = self._searcher_factory(h, self.doc_base, mode, k1, b, self.group)
            return
        if mode == "bm25":
            self.dev = DeviceIndex.from_host_index(h, k1=k1, b=b, device=self.device, tile_log2=self.tile_log2, doc_base=self.doc_base,
                                                   keep_canonical=not self.one_copy)
        else:
            self.dev = DeviceIndex.from_csr(h.indptr, h.indices, h.data, h.idf, mode="dot", device=self.device,
                                            tile_log2=self.tile_log2, doc_base=self.doc_base, keep_canonical=not self.one_copy)
        if self.sharded():
            from .distributed import ShardedSearcher, global_term_bounds
            global_term_bounds(self.dev, self.group)  # corpus-wide thresholds; the search stays exact
            self.searcher = ShardedSearcher.for_device_index(self.dev, self.group)
            self.searcher.force_exchange = self._force_sharded

    # -- search -----------------------------------------------------------------------------------------
    def search_arrays(self, q_ptr, q_term, q_weight, k: int):
        """Host CSR batch -> host rows (doc i32[nq, k] GLOBAL row ids, score f32[nq, k], count i32[nq]); any k >= 1."""
        if self.searcher is None:
            return self.dev.search(q_ptr, q_term, q_weight, k)
        import torch
        validate_query_batch(q_ptr, q_term, q_weight, self.host.vocab_size)
        nq = len(q_ptr) - 1
        if nq == 0:
            return np.zeros((0, k), np.int32), np.zeros((0, k), np.float32), np.zeros(0, np.int32)
        dev = self.dev.device if self.dev is not None else torch.device("cpu")
        qp = torch.as_tensor(np.ascontiguousarray(q_ptr, dtype=np.int32), device=dev)
        qt = torch.as_tensor(np.ascontiguousarray(q_term, dtype=np.int32), device=dev)
        qw = torch.as_tensor(np.ascontiguousarray(q_weight, dtype=np.float32), device=dev)
        page = _capi.limits()["max_k"] if self.dev is not None else 1024
        d, s, c = deep_search(self.searcher.search, qp, qt, qw, k, page)
        if dev.type == "cuda":
            self.searcher.wait()
            torch.cuda.synchronize(dev)
        return d.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()

    def close(self) -> None:
        if self.dev is not None:
            self.dev.close()
            self.dev = None
        self.searcher = None
