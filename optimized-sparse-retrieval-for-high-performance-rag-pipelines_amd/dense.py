"""Dense INT8 side of the same service: the reference's ``QuantizedEmbeddingRetriever`` hot path,
``quantized_dot_product_batch`` + top-k (rag_system/core/retriever_registry.py:90-117, 435-463, 465-524), on the HIP
engine (``srx_dense_search_i8``: one MFMA int8 GEMM, fp64 scaling like the reference's NumPy scalars, exact top-k), and
its asymmetric uint8 scheme (:449-462, 550-559; ``srx_dense_search_u8``).

Embedding *generation* stays outside (the reference simulates it from ``hash(text)``); this module starts from the
embeddings, like the kernel-level functions of the reference do."""
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _capi
from .index import _ptr, _stream_ptr, _torch

DIMS = (32, 64, 96, 128, 192, 256, 384, 512, 768, 1024)  # row lengths the kernel is instantiated for


def quantize_symmetric(embeddings: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """retriever_registry.py:437-447: per-row scale = max |x| (>= 1e-8), int8 = round(x / scale * 127)."""
    e = np.asarray(embeddings)
    scales = np.maximum(np.max(np.abs(e), axis=1, keepdims=True), 1e-8)
    q = np.round(e / scales * 127.0).astype(np.int8)
    return q, scales.flatten().astype(np.float32)


def quantize_query_symmetric(query_embedding: np.ndarray) -> Tuple[np.ndarray, np.float32]:
    """retriever_registry.py:482-485: int8 = round(x / max|x| * 127), query scale = max|x| / 127 (as f32)."""
    x = np.asarray(query_embedding)
    s = np.max(np.abs(x))
    return np.round(x / s * 127.0).astype(np.int8), np.array([s / 127.0], dtype=np.float32)[0]


def quantize_asymmetric(embeddings: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """retriever_registry.py:449-462: per-row min / max, scale = (max - min) / 255 (>= 1e-8), u8 = round((x - min) / scale);
    the table is all scales followed by all mins (f32[2 n]), exactly as the reference stores it."""
    e = np.asarray(embeddings)
    min_vals = np.min(e, axis=1, keepdims=True)
    max_vals = np.max(e, axis=1, keepdims=True)
    scales = np.maximum((max_vals - min_vals) / 255.0, 1e-8)
    q = np.round((e - min_vals) / scales).astype(np.uint8)
    return q, np.concatenate([scales.flatten(), min_vals.flatten()]).astype(np.float32)


def quantize_query_asymmetric(query_embedding: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """retriever_registry.py:486-491: u8 = round((x - min) / ((max - min) / 255)), query_scales = f32[scale, min]."""
    x = np.asarray(query_embedding)
    qmin, qmax = np.min(x), np.max(x)
    scale = (qmax - qmin) / 255.0
    return np.round((x - qmin) / scale).astype(np.uint8), np.array([scale, qmin], dtype=np.float32)


def dequantize_query_asymmetric(query_uint8: np.ndarray, query_scales: np.ndarray) -> np.ndarray:
    """retriever_registry.py:555: query_fp32 = u8.astype(f32) * query_scale + query_min (f32 scalars)."""
    query_scale, query_min = query_scales
    return query_uint8.astype(np.float32) * query_scale + query_min


def _pad_dim(dim: int) -> int:
    for d in DIMS:
        if d >= dim:
            return d
    raise ValueError(f"embedding dim {dim} > {DIMS[-1]} is not supported by the INT8 engine")


class DenseInt8Index:
    """INT8 corpus resident in HBM: ``corpus_int8`` i8[n_docs, dim] (rows zero-padded to a supported length) and
    ``corpus_scales`` f32[n_docs] -- the state ``QuantizedEmbeddingRetriever.build_index_from_corpus`` keeps
    (retriever_registry.py:389-392).  By default the matrix is kept in MFMA-fragment order only (``srx_dense_pack_i8``:
    same bytes; a wave's B-fragment loads are contiguous); ``packed=False`` keeps the row-major matrix and searches that."""

    def __init__(self, corpus_int8, corpus_scales, device="cuda:0", doc_base: int = 0, packed: bool = True):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.SparseRxUnavailable("no HIP device visible: DenseInt8Index needs a GPU (there is no CPU fallback)")
        _capi.lib()
        self.device = torch.device(device)
        c = corpus_int8 if isinstance(corpus_int8, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(corpus_int8, dtype=np.int8))
        assert c.dtype == torch.int8 and c.dim() == 2
        self.n_docs, self.dim = int(c.shape[0]), int(c.shape[1])
        self.dim_pad = _pad_dim(self.dim)
        with torch.cuda.device(self.device):
            rows = torch.zeros((self.n_docs, self.dim_pad), dtype=torch.int8, device=self.device)
            rows[:, : self.dim] = c.to(self.device)
            self.packed = bool(packed)
            if self.packed:
                L = _capi.lib()
                nbytes = _capi.check(L.srx_dense_packed_bytes(self.n_docs, self.dim_pad), "srx_dense_packed_bytes")
                self.corpus = torch.empty(nbytes, dtype=torch.int8, device=self.device)
                _capi.check(L.srx_dense_pack_i8(self.device.index or 0, _ptr(rows), self.n_docs, self.dim_pad, _ptr(self.corpus),
                                                _stream_ptr(torch, self.device)), "srx_dense_pack_i8")
                torch.cuda.synchronize(self.device)
                del rows
            else:
                self.corpus = rows
            s = corpus_scales if isinstance(corpus_scales, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(corpus_scales, dtype=np.float32))
            self.scales = s.to(device=self.device, dtype=torch.float32).contiguous()
        assert self.scales.numel() == self.n_docs
        self.doc_base = int(doc_base)
        self._ws = None

    def search_device(self, queries_int8, query_scales, k: int):
        """queries i8[nq, dim] + f32[nq] on the device -> (doc i32[nq,k], score f32[nq,k], count i32[nq]); asynchronous."""
        torch = _torch()
        if not (1 <= k <= _capi.limits()["max_k"]):
            raise ValueError(f"top_k must be in [1, {_capi.limits()['max_k']}] for the HIP engine, got {k}")
        nq = int(queries_int8.shape[0])
        L = _capi.lib()
        with torch.cuda.device(self.device):
            q = torch.zeros((nq, self.dim_pad), dtype=torch.int8, device=self.device)
            q[:, : self.dim] = queries_int8
            qs = query_scales.to(device=self.device, dtype=torch.float32).contiguous()
            out = (torch.empty((nq, k), dtype=torch.int32, device=self.device), torch.empty((nq, k), dtype=torch.float32, device=self.device),
                   torch.empty((nq,), dtype=torch.int32, device=self.device))
            if nq == 0:
                return out
            need = _capi.check(L.srx_dense_workspace_bytes(nq, self.n_docs, k), "srx_dense_workspace_bytes")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            fn = L.srx_dense_search_i8_packed if self.packed else L.srx_dense_search_i8
            rc = fn(self.device.index or 0, _ptr(self.corpus), _ptr(self.scales), self.n_docs, self.dim_pad,
                    _ptr(q), _ptr(qs), nq, k, self.doc_base, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                    _ptr(self._ws), self._ws.numel(), _stream_ptr(torch, self.device))
            _capi.check(rc, "srx_dense_search_i8")
        return out

    def search(self, queries_int8: np.ndarray, query_scales: np.ndarray, k: int):
        """Host arrays in, host arrays out."""
        torch = _torch()
        d, s, n = self.search_device(torch.as_tensor(np.ascontiguousarray(queries_int8, dtype=np.int8), device=self.device),
                                     torch.as_tensor(np.ascontiguousarray(query_scales, dtype=np.float32), device=self.device), k)
        torch.cuda.synchronize(self.device)
        return d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()


class DenseUint8Index:
    """Asymmetric-scheme corpus resident in HBM: ``corpus_uint8`` u8[n_docs, dim] and the reference's ``corpus_scales``
    table f32[2 n_docs] unchanged (retriever_registry.py:449-462); ``search`` replaces the de-quantize + ``np.dot`` loop of
    ``_numpy_quantized_similarity`` (:550-559) + the top-k for a batch of de-quantized query vectors
    (``srx_dense_search_u8``, which indexes the table the way the reference's reader does)."""

    def __init__(self, corpus_uint8, corpus_scales, device="cuda:0", doc_base: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.SparseRxUnavailable("no HIP device visible: DenseUint8Index needs a GPU (there is no CPU fallback)")
        _capi.lib()
        self.device = torch.device(device)
        c = torch.as_tensor(np.ascontiguousarray(corpus_uint8, dtype=np.uint8))
        assert c.dim() == 2
        self.n_docs, self.dim = int(c.shape[0]), int(c.shape[1])
        self.dim_pad = (self.dim + 63) // 64 * 64
        if self.dim_pad > 1024:
            raise ValueError(f"embedding dim {self.dim} > 1024 is not supported by the uint8 engine")
        s = np.ascontiguousarray(corpus_scales, dtype=np.float32).reshape(-1)
        if s.size != 2 * self.n_docs:
            raise ValueError("corpus_scales must hold 2 * n_docs floats (retriever_registry.py:459)")
        with torch.cuda.device(self.device):
            self.corpus = torch.zeros((self.n_docs, self.dim_pad), dtype=torch.uint8, device=self.device)
            self.corpus[:, : self.dim] = c.to(self.device)
            self.scales = torch.as_tensor(s).to(self.device)
        self.doc_base = int(doc_base)
        self._ws = None

    def search_device(self, queries_f32, k: int):
        torch = _torch()
        if not (1 <= k <= _capi.limits()["max_k"]):
            raise ValueError(f"top_k must be in [1, {_capi.limits()['max_k']}] for the HIP engine, got {k}")
        nq = int(queries_f32.shape[0])
        L = _capi.lib()
        with torch.cuda.device(self.device):
            q = torch.zeros((nq, self.dim_pad), dtype=torch.float32, device=self.device)
            q[:, : self.dim] = queries_f32
            out = (torch.empty((nq, k), dtype=torch.int32, device=self.device), torch.empty((nq, k), dtype=torch.float32, device=self.device),
                   torch.empty((nq,), dtype=torch.int32, device=self.device))
            if nq == 0:
                return out
            need = _capi.check(L.srx_dense_f32_workspace_bytes(nq, self.n_docs, k), "srx_dense_f32_workspace_bytes")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            rc = L.srx_dense_search_u8(self.device.index or 0, _ptr(self.corpus), _ptr(self.scales), self.n_docs, self.dim_pad, _ptr(q),
                                       nq, k, self.doc_base, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(self._ws),
                                       self._ws.numel(), _stream_ptr(torch, self.device))
            _capi.check(rc, "srx_dense_search_u8")
        return out

    def search(self, queries_uint8: np.ndarray, query_scales: np.ndarray, k: int):
        """queries u8[nq, dim] + f32[nq, 2] (scale, min) as the reference's search builds them (:486-491); host arrays out."""
        torch = _torch()
        qf = np.stack([dequantize_query_asymmetric(q, s) for q, s in zip(np.asarray(queries_uint8), np.asarray(query_scales))])
        d, s, n = self.search_device(torch.as_tensor(np.ascontiguousarray(qf, dtype=np.float32), device=self.device), k)
        torch.cuda.synchronize(self.device)
        return d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()


class QuantizedEmbeddingIndex:
    """The search half of the reference's ``QuantizedEmbeddingRetriever`` (symmetric INT8) over given embeddings:
    ``build(doc_ids, embeddings)`` quantizes like :435-447, ``search(query_embeddings, top_k)`` quantizes each query like
    :482-485 and returns ``{doc_id: score}`` ranked, ``score > 0`` only (:515-519), for every query at once."""

    def __init__(self, device="cuda:0"):
        self.device = device
        self.doc_ids: List[str] = []
        self.index = None

    def build(self, doc_ids: Sequence[str], embeddings: np.ndarray) -> None:
        if len(doc_ids) == 0:
            raise ValueError("Empty corpus provided")
        q, scales = quantize_symmetric(embeddings)
        self.doc_ids = list(doc_ids)
        self.index = DenseInt8Index(q, scales, device=self.device)

    def search(self, query_embeddings: Dict[str, np.ndarray], top_k: int = 10) -> Dict[str, Dict[str, float]]:
        if self.index is None:
            raise ValueError("Index not built. Call build_index_from_corpus() first.")
        qids = list(query_embeddings)
        if not qids:
            return {}
        qq = [quantize_query_symmetric(query_embeddings[q]) for q in qids]
        k = min(top_k, len(self.doc_ids))
        d, s, n = self.index.search(np.stack([a for a, _ in qq]), np.array([b for _, b in qq], dtype=np.float32), k)
        return {qid: {self.doc_ids[int(d[i, j])]: float(s[i, j]) for j in range(int(n[i]))} for i, qid in enumerate(qids)}


class DenseF32Index:
    """f32 embedding matrix resident in HBM: the ``embedding_index`` of ``RetrievalService`` (retrieval.py:329-335);
    ``search`` replaces ``np.dot(self.embedding_index, query_vector)`` + top-k of ``search_by_vector`` (:411-423) for a
    batch of query vectors (``srx_dense_search_f32``)."""

    def __init__(self, embeddings, device="cuda:0", doc_base: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.SparseRxUnavailable("no HIP device visible: DenseF32Index needs a GPU (there is no CPU fallback)")
        _capi.lib()
        self.device = torch.device(device)
        e = embeddings
        assert len(e.shape) == 2 and (not isinstance(e, torch.Tensor) or e.dtype == torch.float32)
        self.n_docs, self.dim = int(e.shape[0]), int(e.shape[1])
        self.dim_pad = (self.dim + 63) // 64 * 64
        if self.dim_pad > 1024:
            raise ValueError(f"embedding dim {self.dim} > 1024 is not supported by the f32 engine")
        with torch.cuda.device(self.device):
            self.emb = torch.zeros((self.n_docs, self.dim_pad), dtype=torch.float32, device=self.device)
            if isinstance(e, torch.Tensor):
                self.emb[:, : self.dim] = e.to(self.device)
            else:
                # host array, possibly a read-only memory map of embedding_path (retrieval.py:329-335): streamed to the device
                # in chunks of <= 64 MB (np.array copies a chunk: a whole-file host copy is never made, and the map itself is
                # never wrapped in a tensor)
                rows = max(1, (64 << 20) // max(1, 4 * self.dim))
                for lo in range(0, self.n_docs, rows):
                    chunk = np.array(e[lo: lo + rows], dtype=np.float32)
                    self.emb[lo: lo + chunk.shape[0], : self.dim] = torch.from_numpy(chunk).to(self.device)
        self.doc_base = int(doc_base)
        self._ws = None
        self._max_norm = None

    def max_row_norm(self) -> float:
        """Largest Euclidean row norm (fp64 accumulation), computed once on the device from the resident matrix in row
        chunks of <= 64 MB -- never a host-side temporary of the (possibly memory-mapped) matrix."""
        if self._max_norm is None:
            torch = _torch()
            rows = max(1, (64 << 20) // (4 * self.dim_pad))
            best = 0.0
            with torch.cuda.device(self.device):
                for lo in range(0, self.n_docs, rows):
                    c = self.emb[lo: lo + rows].double()
                    best = max(best, float((c * c).sum(dim=1).max().item()))
            self._max_norm = best ** 0.5
        return self._max_norm

    def search_device(self, queries, k: int, score_offset: float = 0.0):
        """Top-k of (score + score_offset) > 0 per query (include/sparse_rx.h); the returned scores carry the offset."""
        torch = _torch()
        if not (1 <= k <= _capi.limits()["max_k"]):
            raise ValueError(f"top_k must be in [1, {_capi.limits()['max_k']}] for the HIP engine, got {k}")
        nq = int(queries.shape[0])
        L = _capi.lib()
        with torch.cuda.device(self.device):
            q = torch.zeros((nq, self.dim_pad), dtype=torch.float32, device=self.device)
            q[:, : self.dim] = queries
            out = (torch.empty((nq, k), dtype=torch.int32, device=self.device), torch.empty((nq, k), dtype=torch.float32, device=self.device),
                   torch.empty((nq,), dtype=torch.int32, device=self.device))
            if nq == 0:
                return out
            need = _capi.check(L.srx_dense_f32_workspace_bytes(nq, self.n_docs, k), "srx_dense_f32_workspace_bytes")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            rc = L.srx_dense_search_f32(self.device.index or 0, _ptr(self.emb), self.n_docs, self.dim_pad, _ptr(q), nq, k, self.doc_base,
                                        _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(self._ws), self._ws.numel(),
                                        _stream_ptr(torch, self.device), float(score_offset))
            _capi.check(rc, "srx_dense_search_f32")
        return out

    def search(self, queries: np.ndarray, k: int, score_offset: float = 0.0):
        torch = _torch()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        d, s, n = self.search_device(torch.as_tensor(q, device=self.device), k, score_offset)
        torch.cuda.synchronize(self.device)
        return d.cpu().numpy(), s.cpu().numpy(), n.cpu().numpy()
