"""sparse-rx: MI355X-native batched BM25 / learned-sparse scoring + top-k behind the reference's
``RetrievalService.build_bm25_index()`` / ``search_bm25()`` API.  Import as ``sparse_rx``."""
from . import _capi
from ._capi import SparseRxError, SparseRxUnavailable, build_library
from .index import DeviceIndex, HostBatchPipeline, HostIndex, build_host_index, encode_queries, merge_topk_device, tokenize
from .service import RetrievalService
from .registry import OptimizedBM25Retriever, OptimizedRetriever, QuantizedEmbeddingRetriever, RetrieverRegistry, load_index_npz, save_index_npz
from .dense import (DenseF32Index, DenseInt8Index, DenseUint8Index, QuantizedEmbeddingIndex, quantize_asymmetric,
                    quantize_query_asymmetric, quantize_query_symmetric, quantize_symmetric)
from .distributed import (ShardedSearcher, shard_range, global_df, global_avgdl, global_term_bounds, bm25_idf_from_df,
                          build_sharded_host_index)
from .backend import SparseBackend

__all__ = ["RetrievalService", "DeviceIndex", "HostBatchPipeline", "HostIndex", "build_host_index", "encode_queries", "merge_topk_device",
           "tokenize", "build_library", "SparseRxError", "SparseRxUnavailable", "_capi", "ShardedSearcher", "shard_range",
           "global_df", "global_avgdl", "global_term_bounds", "bm25_idf_from_df", "build_sharded_host_index", "SparseBackend", "OptimizedBM25Retriever", "OptimizedRetriever", "QuantizedEmbeddingRetriever", "RetrieverRegistry",
           "load_index_npz", "save_index_npz", "DenseF32Index", "DenseInt8Index", "DenseUint8Index", "QuantizedEmbeddingIndex", "quantize_symmetric",
           "quantize_asymmetric", "quantize_query_asymmetric",
           "quantize_query_symmetric"]
