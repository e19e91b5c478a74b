"""Registry / pipeline adapters and the .npz index cache (row f2).

Mirrors, on the HIP engine, the two other call sites of the reference's hot path:
  * ``OptimizedBM25Retriever`` + ``RetrieverRegistry``  -- /root/reference/rag_system/core/retriever_registry.py:120-356, 562-599
  * ``OptimizedRetriever``                              -- /root/reference/rag_system/pipeline/evaluate_rag_pipeline.py:162-479
    (``bm25*`` types score with ``simd_bm25_score``, every other type with ``simd_tfidf_score`` and
    idf = log(N/(df+1)), :257-278, :378-399; index cache ``.rag_cache/{method}_index_{hash}.npz``, :189-200, :280-312)
so that the YAML experiments and ``benchmark_efficiency`` (objects with ``build_index_from_corpus`` + ``search``) run
unmodified.  The registry's dense types (dpr / contriever / splade) go to the ``QuantizedEmbeddingRetriever`` mirror.

``top_k``: any value, like the reference (deep rankings are paged with ``srx_search_after``); ``top_k <= 0`` gives ``{}``.
"""
from __future__ import annotations

import hashlib
import threading
import time
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .backend import SparseBackend
from .index import DeviceIndex, HostIndex, encode_queries

_BM25_TYPES = ("bm25", "bm25_retriever", "bm25_custom")


class _SparseRetrieverBase:
    """Shared batched search: cache semantics of the reference call sites, one srx_search per call."""

    mode = "bm25"
    strip_cache_key = True
    term_order = "term"  # accumulation order of a doc's contributions (index.encode_queries)

    def __init__(self, k1: float, b: float, device: Optional[str], tile_log2: int, use_cache: bool = True, group=None,
                 shard_searcher_factory=None, sharded: Optional[bool] = None, one_copy: bool = True):
        self.k1, self.b = k1, b
        # one GPU, or -- inside an initialised torch.distributed group -- doc-range shards (backend.SparseBackend)
        self._be = SparseBackend(device, tile_log2, group=group, searcher_factory=shard_searcher_factory, sharded=sharded, one_copy=one_copy)
        self.device, self.tile_log2 = self._be.device, tile_log2
        self.query_cache: Optional[Dict[str, Tuple[np.ndarray, np.ndarray]]] = {} if use_cache else None
        self.cache_lock = threading.RLock()

    @property
    def host(self) -> Optional[HostIndex]:
        return self._be.host

    @property
    def dev(self) -> Optional[DeviceIndex]:
        return self._be.dev

    # reference attribute names
    @property
    def vocabulary(self):
        return self.host.vocabulary if self.host else {}

    @property
    def doc_ids(self):
        return self.host.doc_ids if self.host else []

    @property
    def corpus_tf(self):
        if self.host is None:
            return None
        from scipy.sparse import csr_matrix
        h = self.host
        return csr_matrix((h.data, h.indices, h.indptr), shape=(h.n_docs, h.vocab_size))

    def _upload(self):
        self._be.upload(self.mode, self.k1, self.b)

    def search(self, queries: Dict[str, str], top_k: int = 10) -> Dict[str, Dict[str, float]]:
        if self.host is None:
            raise ValueError("Index not built. Call build_index_from_corpus() first.")
        results: Dict[str, Dict[str, float]] = {}
        pending: Dict[str, List[str]] = {}
        texts, keys = [], []
        for qid, text in queries.items():
            if not text:  # retriever_registry.py:237-239
                results[qid] = {}
                continue
            key = f"{text.strip() if self.strip_cache_key else text}:{top_k}"
            hit = None
            if self.query_cache is not None:
                with self.cache_lock:
                    hit = self.query_cache.get(key)
            if hit is not None:
                results[qid] = self._to_dict(*hit)
                continue
            results[qid] = {}
            if key not in pending:
                pending[key] = []
                texts.append(text)
                keys.append(key)
            pending[key].append(qid)
        if texts:
            q_ptr, q_term, q_w = encode_queries(texts, self.host.vocabulary, order=self.term_order)
            k_eff = min(int(top_k), self._be.n_docs_total)  # any depth: DeviceIndex.search pages past the engine's 1024-row lists
            if k_eff <= 0:  # argpartition(...)[:0] keeps nothing (retriever_registry.py:304-312): {} per query, like the reference
                return results
            docs, scores, counts = self._be.search_arrays(q_ptr, q_term, q_w, k_eff)
            for i, key in enumerate(keys):
                if q_ptr[i + 1] == q_ptr[i]:
                    continue
                c = int(counts[i])
                entry = (docs[i, :c].astype(np.int64), scores[i, :c].copy())
                if self.query_cache is not None:
                    with self.cache_lock:
                        if len(self.query_cache) < 1000:
                            self.query_cache[key] = entry
                d = self._to_dict(*entry)
                for qid in pending[key]:
                    results[qid] = dict(d)
        return results

    def _to_dict(self, idx, sc):
        ids = self.host.doc_ids
        return {ids[int(i)]: float(s) for i, s in zip(idx, sc) if s > 0}

    def close(self):
        self._be.close()


class OptimizedBM25Retriever(_SparseRetrieverBase):
    """retriever_registry.py:120-356 (``method='tfidf'`` is BM25 with k1=1000, b=0 there, :593-595)."""

    def __init__(self, method: str = "bm25", model: str = None, k1: float = 1.2, b: float = 0.75, device: Optional[str] = None,
                 tile_log2: int = 14, **kwargs):
        super().__init__(k1, b, device, tile_log2, use_cache=kwargs.get("cache_queries", True), group=kwargs.get("group"),
                         shard_searcher_factory=kwargs.get("shard_searcher_factory"), sharded=kwargs.get("sharded"),
                         one_copy=kwargs.get("one_copy", True))
        self.method = method.lower()
        self.model_name = model
        self.use_simd = kwargs.get("use_simd", True)  # accepted, meaningless here

    def build_index_from_corpus(self, corpus: Dict[str, Dict]) -> None:
        if not corpus:
            raise ValueError("Empty corpus provided")  # retriever_registry.py:155-156
        self._be.build(corpus, idf_kind="bm25")
        self._upload()


class OptimizedRetriever(_SparseRetrieverBase):
    """evaluate_rag_pipeline.py:162-479: config dict + hardware dict; non-BM25 types use the tf-idf dot product.

    ``accumulation``: the order in which a doc's per-term contributions are added.  The reference has two answers:
    its NumPy fallback ``_numpy_score_documents`` (:436-479, what runs wherever numba is absent -- and what the committed
    fixtures tests/golden/pipeline_small.* were produced with) walks ``relevant_terms`` in QUERY-TOKEN order; its Numba
    kernels (:57-121) walk the CSR row, i.e. ascending term id.  The two differ in the last fp32 bit on about a quarter
    of the fixture queries.  Default ``"token"`` reproduces the runnable reference bit for bit; ``"term"`` gives the
    Numba / ``RetrievalService`` order."""

    strip_cache_key = False  # its cache key is f"{query_text}:{top_k}" (:340)

    def __init__(self, config: Dict[str, Any], hardware_info: Optional[Dict[str, Any]] = None, device: Optional[str] = None,
                 tile_log2: int = 14, cache_dir: str = ".rag_cache", accumulation: str = "token", group=None,
                 shard_searcher_factory=None, sharded: Optional[bool] = None, one_copy: bool = True):
        if accumulation not in ("token", "term"):
            raise ValueError("accumulation must be 'token' or 'term'")
        self.term_order = accumulation
        params = config.get("params", {}) or {}
        hardware_info = hardware_info or {"memory_gb": 8, "cores": 4}
        super().__init__(params.get("k1", 1.2), params.get("b", 0.75), device, tile_log2,
                         use_cache=hardware_info.get("memory_gb", 8) > 4, group=group, shard_searcher_factory=shard_searcher_factory,
                         sharded=sharded, one_copy=one_copy)
        self.config, self.hardware = config, hardware_info
        self.method = config.get("type", "bm25").lower()
        self.mode = "bm25" if self.method in ("bm25", "bm25_custom") else "dot"  # :258-261, :378-399
        self.use_cache = hardware_info.get("memory_gb", 8) > 4
        self.cache_dir = Path(cache_dir)

    def build_index_from_corpus(self, corpus: Dict[str, Dict]) -> None:
        corpus_hash = hashlib.md5(str(sorted(corpus.keys())[:1000]).encode()).hexdigest()[:8]  # :189
        cache_file = self.cache_dir / f"{self.method}_index_{corpus_hash}.npz"
        sharded = self._be.sharded()  # the .npz cache holds a whole-corpus index: shards are always built from the corpus
        if self.use_cache and cache_file.exists() and not sharded:
            self._be.set_host(load_index_npz(cache_file))
        else:
            self._be.build(corpus, idf_kind="bm25" if self.mode == "bm25" else "tfidf")
            if self.use_cache and not sharded:
                self.cache_dir.mkdir(exist_ok=True)
                save_index_npz(cache_file, self.host)
        self._upload()


def save_index_npz(path, h: HostIndex) -> None:
    """The reference's cache schema (evaluate_rag_pipeline.py:280-296): same keys and dtypes."""
    vocab_sorted = sorted(h.vocabulary, key=h.vocabulary.get)
    np.savez_compressed(path, tf_data=h.data, tf_indices=h.indices, tf_indptr=h.indptr,
                        tf_shape=np.array((h.n_docs, h.vocab_size), dtype=np.int64), doc_lengths=h.doc_lengths, idf=h.idf,
                        vocabulary=np.array(vocab_sorted), doc_ids=np.array(h.doc_ids), avgdl=np.float32(h.avgdl))


def load_index_npz(path) -> HostIndex:
    """Reads the schema above with ``allow_pickle=False`` (string arrays are plain ``<U`` arrays)."""
    z = np.load(path, allow_pickle=False)
    return HostIndex(indptr=z["tf_indptr"], indices=z["tf_indices"], data=z["tf_data"], doc_lengths=z["doc_lengths"],
                     idf=z["idf"], avgdl=float(z["avgdl"]), vocabulary={str(t): i for i, t in enumerate(z["vocabulary"])},
                     doc_ids=[str(d) for d in z["doc_ids"]])


class QuantizedEmbeddingRetriever:
    """Mirror of the reference's dense retriever (retriever_registry.py:358-559) on the HIP engine: the same simulated
    embeddings (clustered corpus vectors from ``np.random.seed(42)``, query vectors seeded by ``hash(query_text)``), the
    symmetric INT8 / asymmetric uint8 quantization and result dicts, with ``quantized_dot_product_batch`` + top-k replaced by
    ``srx_dense_search_i8`` (asymmetric: ``srx_dense_search_u8``; ``use_quantization=False``: ``np.dot`` + top-k replaced by
    ``srx_dense_search_f32``).  All
    queries of a ``search`` call go to the GPU as one batch."""

    def __init__(self, method: str, model: str, embedding_dim: int = 768, device: str = "cuda:0", **kwargs):
        self.method = method.lower()
        self.model_name = model
        self.embedding_dim = embedding_dim
        self.use_quantization = kwargs.get("use_quantization", True)
        self.quantization_method = kwargs.get("quantization_method", "symmetric")
        self.device = device
        self.corpus_embeddings_int8: Optional[np.ndarray] = None
        self.corpus_scales: Optional[np.ndarray] = None
        self.corpus_embeddings_fp32: Optional[np.ndarray] = None
        self.doc_ids: List[str] = []
        self._index = None

    def synthetic_embeddings(self, num_docs: int) -> np.ndarray:
        """retriever_registry.py:409-433: cluster centres + 0.1 noise from the legacy NumPy stream seeded with 42, rows
        normalised; drawn in the reference's order (centres, assignments, then one noise row per document)."""
        np.random.seed(42)
        num_clusters = min(50, num_docs // 10)
        centers = np.random.randn(num_clusters, self.embedding_dim).astype(np.float32)
        assign = np.random.randint(0, num_clusters, num_docs)
        noise = np.random.randn(num_docs, self.embedding_dim) * 0.1       # row i = the i-th randn(dim) call of the reference
        emb = (centers[assign] + noise).astype(np.float32)                 # f32 + f64 -> f64, stored as f32 (:426)
        norms = np.linalg.norm(emb, axis=1, keepdims=True)
        return emb / np.maximum(norms, 1e-8)

    def query_embedding_from_seed(self, seed: int) -> np.ndarray:
        """retriever_registry.py:526-536 after the hash: randn(dim) from the legacy stream, as f32, normalised."""
        np.random.seed(seed)
        e = np.random.randn(self.embedding_dim).astype(np.float32)
        return e / np.linalg.norm(e)

    def _generate_query_embedding(self, query_text: str) -> np.ndarray:
        return self.query_embedding_from_seed(hash(query_text) % (2 ** 31))  # process-dependent, like the reference's

    def build_index_from_corpus(self, corpus: Dict[str, Dict]) -> None:
        from .dense import DenseF32Index, DenseInt8Index, DenseUint8Index, quantize_asymmetric, quantize_symmetric
        self.doc_ids = list(corpus.keys())
        emb = self.synthetic_embeddings(len(corpus))
        if self.use_quantization and self.quantization_method == "symmetric":
            self.corpus_embeddings_int8, self.corpus_scales = quantize_symmetric(emb)
            self._index = DenseInt8Index(self.corpus_embeddings_int8, self.corpus_scales, device=self.device)
        elif self.use_quantization:  # any other value is the asymmetric scheme, like the reference's else branch (:449)
            self.corpus_embeddings_int8, self.corpus_scales = quantize_asymmetric(emb)
            self._index = DenseUint8Index(self.corpus_embeddings_int8, self.corpus_scales, device=self.device)
        else:
            self.corpus_embeddings_fp32 = emb
            self._index = DenseF32Index(emb, device=self.device)

    def search(self, queries: Dict[str, str], top_k: int = 10) -> Dict[str, Dict[str, float]]:
        from .dense import quantize_query_asymmetric, quantize_query_symmetric
        if self._index is None:
            raise ValueError("Index not built. Call build_index_from_corpus() first.")
        results: Dict[str, Dict[str, float]] = {qid: {} for qid in queries}
        live = [(qid, text) for qid, text in queries.items() if text]
        if not live:
            return results
        embs = [self._generate_query_embedding(text) for _, text in live]
        k = max(1, min(int(top_k), len(self.doc_ids)))
        if self.use_quantization and self.quantization_method == "symmetric":
            qq = [quantize_query_symmetric(e) for e in embs]
            d, s, n = self._index.search(np.stack([a for a, _ in qq]), np.array([b for _, b in qq], dtype=np.float32), k)
        elif self.use_quantization:
            qq = [quantize_query_asymmetric(e) for e in embs]
            d, s, n = self._index.search(np.stack([a for a, _ in qq]), np.stack([b for _, b in qq]), k)
        else:
            d, s, n = self._index.search(np.stack(embs), k)
        for i, (qid, _) in enumerate(live):
            results[qid] = {self.doc_ids[int(d[i, j])]: float(s[i, j]) for j in range(int(n[i]))}  # score > 0 only (:515-519)
        return results


class RetrieverRegistry:
    """retriever_registry.py:562-599."""

    _retrievers: Dict[str, Any] = {}

    @classmethod
    def register(cls, name: str, retriever_class) -> None:
        cls._retrievers[name] = retriever_class

    @classmethod
    def create(cls, config):
        if isinstance(config, str):
            method, model, params = config, None, {}
        else:
            method = config.get("type", config.get("name"))
            model = config.get("model")
            params = config.get("params", {}) or {}
        if not method:
            raise ValueError("Retriever name/type not specified")
        m = method.lower()
        if m in _BM25_TYPES:
            return OptimizedBM25Retriever(method=method, model=model, **params)
        if m == "tfidf":
            return OptimizedBM25Retriever(method="tfidf", model=model, k1=1000, b=0, **params)  # :593-595
        if m in ("dpr", "contriever", "splade"):  # :588-592
            p2 = dict(params)
            embedding_dim = p2.pop("embedding_dim", 768)
            return QuantizedEmbeddingRetriever(method=method, model=model or f"quantized_{method}", embedding_dim=embedding_dim, **p2)
        if method in cls._retrievers:
            return cls._retrievers[method](**params)
        raise ValueError(f"Unknown retriever method: {method}")

    @classmethod
    def list_available(cls):
        return {"optimized_sparse": ["bm25", "bm25_custom", "tfidf"], "quantized_dense": ["dpr", "contriever", "splade"],
                "registered_custom": list(cls._retrievers.keys())}
