/*
 * sparse_rx.h -- C ABI of libsparse_rx.so, the MI355X (gfx950) scoring + top-k engine.
 *
 * The reference (100 % Python) has no FFI; its hot path is two Numba kernels called from
 * RetrievalService._score_bm25_query.  This header is the boundary a maintainer binds with ctypes
 *  to replace exactly those call sites.  Paths are relative to /root/reference:
 *
 *   srx_search            replaces  simd_bm25_score(...)        rag_system/core/retrieval.py:256-266
 *                                   + fast_topk_selection(...)  rag_system/core/retrieval.py:273 (NumPy twin 276-284)
 *                                   + the score>0 filter        rag_system/core/retrieval.py:292-296
 *                         (twins: rag_system/core/retriever_registry.py:287-297,304;
 *                                 rag_system/pipeline/evaluate_rag_pipeline.py:380-399,406 incl. simd_tfidf_score)
 *   srx_build_impacts     evaluates the per-posting BM25 term  retrieval.py:58,70-71  once at index build
 *   srx_build_tile_skip,  device index construction (no reference counterpart: the reference scans the
 *   srx_build_blocks      whole doc-major CSR per query, retrieval.py:55-72)
 *   srx_merge_topk        the (score desc, doc asc) merge of per-shard / per-split top-k lists
 *                         (multi-GPU: after the RCCL all-gather; no reference counterpart)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.  Every array pointer is a DEVICE
 *     pointer (e.g. torch.Tensor.data_ptr()) unless the parameter name starts with h_.
 *   - The caller owns every buffer (index arrays, queries, outputs, workspace).  The library owns only
 *     the small srx_index handle.  srx_search allocates nothing and is asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).
 *   - Return value 0 = OK, negative = error (srx_status); srx_last_error() returns a thread-local
 *     message.  The Python shim maps them to ValueError / RuntimeError.
 *   - Not thread-safe per handle; re-entrant across handles / streams.
 */
#ifndef SPARSE_RX_H
#define SPARSE_RX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRX_VERSION 301 /* 0.3.1: + srx_search_after (ranking of any depth), srx_build_term_bounds, srx_build_sum_duplicates, packed dense corpus */

typedef enum {
    SRX_OK = 0,
    SRX_ERR_INVALID = -1, /* bad argument */
    SRX_ERR_HIP = -2,     /* HIP runtime error */
    SRX_ERR_NOMEM = -3,   /* workspace too small */
    SRX_ERR_NODEVICE = -4 /* no usable GPU */
} srx_status;

typedef enum {
    SRX_VAL_F32 = 0, /* post_val is float32 (BM25 impacts, or raw weights in dot mode) */
    SRX_VAL_F16 = 1  /* post_val is IEEE half (learned-sparse "dot" mode, SPLADE-style) */
} srx_val_type;

/* Limits of this build (srx_limits() returns them at run time). */
#define SRX_MAX_K 1024         /* largest top-k */
#define SRX_MAX_TILE_LOG2 14   /* skip-table granularity G = 2^tile_log2 docs, G <= 16384 */
#define SRX_BLOCK_PAD 256       /* sentinel blocks behind the last run: post must hold n_blocks + SRX_BLOCK_PAD blocks; block j of them
                                 * carries doc -1 - 32 (j mod 64): lane l of a tier-1 wave sends the idle load of step s to block
                                 * l + s * (lanes per term) <= 63 + 3 * 64, same immediate offset as the real load */

/*
 * Device-resident inverted index of one doc-range shard: term-major postings in BLOCKS + a tile skip table.
 *
 * The postings of a term (ascending shard-local doc id) are cut into one run per UNIT of unit_tiles * G consecutive
 * docs (unit_tiles * G <= 49152 for the tier-1 kernel; srx_auto_unit_tiles picks it).  Every run is padded to a
 * multiple of 4 postings with sentinels (doc -1 - 32 * (term % 64), value 0; trailing sentinel block j holds docs
 * -1 - 32 j) and stored as blocks of 4 postings, docs and values side by side:
 *     SRX_VAL_F32: [d0 d1 d2 d3 | v0 v1 v2 v3]   8 x 32-bit words
 *     SRX_VAL_F16: [d0 d1 d2 d3 | h0 h1 h2 h3]   6 words (4 docs, 4 halves)
 * Negative doc ids mark sentinels.  A posting whose stored value is exactly 0 is ignored by every kernel (it would add +-0 and can never be a result:
 * only scores > 0 are returned, retrieval.py:295).  "Padded position" p = block p >> 2, slot p & 3, counted over the
 * whole array.  srx_build_blocks produces all of this from plain term-major arrays.
 *
 *   term_ptr[t]                    padded position of term t's first posting (a multiple of 4); term_ptr[vocab] = end
 *   post                           the blocks; n_blocks of them, followed by SRX_BLOCK_PAD all-sentinel blocks
 *   stored value                   BM25: impact = (tf*(k1+1)) / (tf + k1*(1-b+b*len/avgdl)) precomputed in fp32
 *                                  (retrieval.py:58,70-71); dot mode: the stored weight (tf)
 *   tile_skip[t*(n_tiles+1) + j]   padded postings of term t with doc < j*G, relative to term_ptr[t] (so the run of
 *                                  term t inside docs [a*G, b*G) is [term_ptr[t]+skip[a], term_ptr[t]+skip[b]) );
 *                                  a multiple of 4 wherever j is a multiple of unit_tiles, and for j = n_tiles
 *   idf[t]                         per-term weight (retrieval.py:189; evaluate_rag_pipeline.py:273-278)
 * Per-posting contribution at query time: (value * idf[t]) * q_weight, fp32, summed per doc in the order the
 * query lists its terms.  Ascending term id is the CSR-row order of retrieval.py:72 / evaluate_rag_pipeline.py:117
 * (bit-identical to them); query-token order reproduces the pipeline twin's NumPy fallback
 * (evaluate_rag_pipeline.py:436-479).
 */
typedef struct {
    int32_t device;    /* HIP device ordinal the pointers live on */
    int32_t val_type;  /* srx_val_type */
    int64_t n_docs;    /* shard-local rows */
    int64_t vocab;
    int64_t nnz;       /* real postings (sentinels not counted) */
    int64_t n_blocks;  /* blocks of 4 padded postings (the trailing sentinel blocks not counted) */
    int64_t doc_base;  /* global row id of local row 0 (added to the ids srx_search returns) */
    int32_t tile_log2; /* G = 1 << tile_log2, <= SRX_MAX_TILE_LOG2 */
    int32_t n_tiles;   /* ceil(n_docs / G) */
    int32_t unit_tiles; /* tiles per unit the runs are padded for (1..64) */
    int32_t reserved0;
    const int64_t *term_ptr;  /* [vocab+1] */
    const int32_t *post;      /* [(n_blocks + SRX_BLOCK_PAD) * words per block]; may be NULL when post16 is given: the tier-2
                                 kernel then reads the compact copy too (one resident copy of the postings; search options
                                 that change the unit are refused) */
    const int32_t *tile_skip; /* [vocab*(n_tiles+1)] */
    const float *idf;         /* [vocab] */
    const float *term_bound;  /* optional [vocab*4], may be NULL: the K-th largest stored value of each term in this shard for
                                 K = 1, 10, 100, 1000 (0 where the term has fewer than K postings).  Requires all
                                 values >= 0.  Gives every query an exact lower bound on its k-th best score (a doc's
                                 score is at least any single contribution when all query idf are >= 0), which the
                                 kernels use as the initial top-k threshold. */
    const int32_t *post16;    /* optional, may be NULL: the compact copy of `post` the tier-1 kernel streams (srx_build_compact:
                                 [(n_blocks + SRX_BLOCK_PAD) * 6 words (f32 values) / 4 words (f16)], 16-bit unit-local doc ids:
                                 6 / 4 bytes per posting instead of 8 / 6).  Without it every query is served by the tier-2
                                 kernel from `post` (exact, slower on short queries). */
} srx_index_desc;

typedef struct srx_index srx_index;

/* Search-time tuning knobs; zero-initialise for defaults. */
typedef struct {
    int32_t supertile_log2; /* docs per unit = 2^supertile_log2 (>= tile_log2); 0 = the index's unit.  Any unit other than
                             * the one the index was padded for is served by the tier-2 kernel alone (exact, slower) */
    int32_t target_blocks;  /* wave-sized work items to aim for (splits of a query / unsplit rounds + split tail); 0 = auto (3072) */
    int32_t profile;        /* N > 0: bracket the kernels of every N-th search with hipEvents (read with srx_profile_read); an
                             * event record between two kernels costs the stream a few microseconds, so sampling keeps the
                             * timed steps close to unprofiled ones */
    int32_t reserved;       /* debug bits.  Exact results: 8 = every query through the tier-2 (block) kernel, 16 = ignore
                             * term_bound, 128 = no flat-tile path in tier 2, 256 = block merge kernel only, 2048 = no wave-level
                             * dense tiles, 4096 = their masked form even on one-tile units, 8192 = their general selection.  Timing
                             * experiments with WRONG results (bench ablations): 1 = no multi-term doc resolution, 2 = no
                             * candidate screening, 4 = loads only, 32 = no final ranking, 512 = multi-term docs located but
                             * not summed, 1024 = ... summed but not appended. */
    int32_t unit_tiles;     /* docs per unit = unit_tiles * 2^tile_log2 (1..64, need not be a power of two); 0 = the index's.
                               Takes precedence over supertile_log2. */
} srx_search_opts;

int srx_version(void);
const char *srx_last_error(void);
/* out[0]=max k, out[1]=max tile_log2, out[2]=hash capacity (postings per unit), out[3]=threads per workgroup */
int srx_limits(int32_t *h_out4);
/* Number of visible HIP devices, or a negative srx_status. */
int srx_device_count(void);

int srx_index_create(const srx_index_desc *h_desc, srx_index **h_out);
void srx_index_destroy(srx_index *ix);
int srx_index_set_opts(srx_index *ix, const srx_search_opts *h_opts);

/* Bytes of device workspace srx_search needs for a batch of nq queries at top-k k. */
int64_t srx_search_workspace_bytes(const srx_index *ix, int32_t nq, int32_t k);

/*
 * Batched scoring + top-k.  Queries are a CSR batch: query q has terms q_term[q_ptr[q] .. q_ptr[q+1])
 * (unique inside a query, in-vocabulary -- OOV terms dropped on the host as retrieval.py:245-249 does; their
 * order is the order in which a doc's contributions are added, see srx_index_desc) with weights q_weight
 * (the term count as float, retrieval.py:248).
 * PRECONDITIONS, not checked on the device: q_ptr[0] == 0 and q_ptr non-decreasing; 0 <= q_term < vocab (the
 * kernels index term_ptr / tile_skip / idf / term_bound with it); no term twice in one query.  The host-array entry
 * point of the Python mirror (DeviceIndex.search) validates them before launching.
 * Outputs, row q: out_doc[q*k + r] = doc_base + local row of rank r, out_score[q*k + r] its fp32 score,
 * r < out_count[q]; rank order = (score descending, doc ascending); only score > 0 (retrieval.py:295);
 * the rest of the row is padded with doc -1 / score 0.
 */
int srx_search(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
               int32_t k, int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
               int64_t workspace_bytes, void *stream);

/*
 * Merge n_lists candidate lists per query into one ranked top-k list (same order and padding as
 * srx_search).  gathered == 0: in_doc/in_score are [nq][n_lists][k], in_count [nq][n_lists];
 * gathered == 1: [n_lists][nq][k] and [n_lists][nq] -- the layout an all-gather of per-rank srx_search
 * outputs produces.  Doc ids are taken as they are (global).  workspace may be NULL when
 * n_lists*k <= 4096, otherwise srx_merge_workspace_bytes() bytes.
 */
int64_t srx_merge_workspace_bytes(int32_t nq, int32_t n_lists, int32_t k);
int srx_merge_topk(int32_t device, const int32_t *in_doc, const float *in_score, const int32_t *in_count,
                   int32_t nq, int32_t n_lists, int32_t k, int32_t gathered, int32_t *out_doc, float *out_score,
                   int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Same merge over PACKED rows, the single-buffer form of the multi-GPU exchange: packed is
 * [n_lists][nq][2k+1] int32 with row = k doc ids, k fp32 score bit patterns, 1 count -- each rank packs its
 * srx_search outputs into [nq][2k+1], ONE all-gather produces this buffer, and the merge reads it in place.
 */
int srx_merge_topk_packed(int32_t device, const int32_t *packed, int32_t nq, int32_t n_lists, int32_t k,
                          int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                          int64_t workspace_bytes, void *stream);

/* The same rows as outputs, so that no packing / unpacking kernels are needed around the exchange:
 * srx_search_packed writes each query's result as one row [k doc ids][k score bits][count] of out_packed[nq][2k+1]
 * (same results as srx_search: retrieval.py:256-284), and srx_merge_topk_packed_out merges gathered packed rows
 * [n_lists][nq][2k+1] into packed rows [nq][2k+1]. */
int srx_search_packed(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                      int32_t k, int32_t *out_packed, void *workspace, int64_t workspace_bytes, void *stream);
int srx_merge_topk_packed_out(int32_t device, const int32_t *packed, int32_t nq, int32_t n_lists, int32_t k,
                              int32_t *out_packed, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * "Search after": the same search restricted to the docs ranked strictly AFTER a given row in the result order
 * (score descending, doc ascending): query q only collects docs with score < after_score[q], or score == after_score[q]
 * and global doc id > after_doc[q].  Feeding a pass the last row of the previous one pages through a ranking of any
 * depth, which is how the host side serves top_k > SRX_MAX_K and the reference's "k >= n_docs: rank everything" branch
 * (rag_system/core/retrieval.py:272-284; twin retriever_registry.py:304).  after_score[q] <= 0 returns nothing for q.
 * Served by the tier-2 kernel (the tier-1 kernel and the term_bound thresholds assume an unrestricted top-k).
 */
int srx_search_after(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                     int32_t k, const int32_t *after_doc, const float *after_score, int32_t *out_doc, float *out_score,
                     int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream);
int srx_search_after_packed(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                            int32_t k, const int32_t *after_doc, const float *after_score, int32_t *out_packed,
                            void *workspace, int64_t workspace_bytes, void *stream);

/* ---- device-side index construction helpers ------------------------------------------------------ */

/* impact[p] = (tf[p]*(k1+1)) / (tf[p] + k1*(1-b + b*doc_len[post_doc[p]]/avgdl)), fp32, the reference's
 * operation order (retrieval.py:58,70-71): f32(k1)*(f32(1-b) + (f32(b)*len)/f32(avgdl)). */
int srx_build_impacts(int32_t device, const float *tf, const int32_t *post_doc, const float *doc_len, int64_t nnz,
                      double k1, double b, double avgdl, float *out_impact, void *stream);

/* Plain term-major arrays (term_ptr i64[vocab+1] offsets into post_doc, ascending doc inside a term):
 * out_skip[t*(n_tiles+1)+j] = #postings of term t with doc < j<<tile_log2  (lower_bound per (t, j)) -- the UNPADDED
 * skip table, an input of srx_build_blocks. */
int srx_build_tile_skip(int32_t device, const int64_t *term_ptr, const int32_t *post_doc, int64_t vocab,
                        int32_t n_tiles, int32_t tile_log2, int32_t *out_skip, void *stream);

/* Tiles per unit for a corpus of this density: the largest unit whose average per-term run still fits the tier-1
 * kernel's registers with a 5-sigma margin and whose docs fit the 16-bit local ids of the compact copy (<= 49152).  Host-only,
 * returns the value. */
int32_t srx_auto_unit_tiles(int64_t n_docs, int64_t vocab, int64_t nnz, int32_t tile_log2);

/* Plain term-major arrays -> the blocked layout of srx_index_desc.  Inputs: term_ptr i64[vocab+1], post_term i32[nnz]
 * (the term of every posting), post_doc i32[nnz], post_val f32|f16[nnz] (val_type), skip = srx_build_tile_skip's
 * table, runpad i64[vocab*n_units+1] = exclusive prefix sum over (term, unit) of the run lengths rounded up to a
 * multiple of 4 (n_units = ceil(n_tiles / unit_tiles); run length of (t, u) = skip[t][min((u+1)*unit_tiles, n_tiles)] -
 * skip[t][u*unit_tiles]), n_blocks = runpad[last] / 4.  Outputs: out_post [(n_blocks + SRX_BLOCK_PAD) * words],
 * out_skip i32[vocab*(n_tiles+1)] (padded positions), out_term_ptr i64[vocab+1]. */
int srx_build_blocks(int32_t device, int32_t val_type, const int64_t *term_ptr, const int32_t *post_term,
                     const int32_t *post_doc, const void *post_val, const int32_t *skip, const int64_t *runpad,
                     int64_t vocab, int64_t nnz, int32_t n_tiles, int32_t tile_log2, int32_t unit_tiles, int32_t *out_post,
                     int32_t *out_skip, int64_t *out_term_ptr, int64_t n_blocks, void *stream);

/* Duplicate (doc, term) entries of a COO input (adjacent after the stable sort by term): out_sum[g] = val[first[g]] + ... +
 * val[first[g + 1] - 1], added left to right -- what SciPy does when the reference assembles its CSR from triples
 * (csr_matrix((data, (rows, cols))), rag_system/core/retrieval.py:171-175).  first i64[n_groups + 1] ascending. */
int srx_build_sum_duplicates(int32_t device, const int64_t *first, int64_t n_groups, const float *val, float *out_sum, void *stream);

/* Per-term score bounds (srx_index_desc.term_bound and the finer table the sharded build combines): out_bound[t * nk + j] =
 * the ks[j]-th largest value of term t's run post_val[term_ptr[t] .. term_ptr[t + 1]) of the term-major value array (before
 * blocking; f32 or f16 by val_type), 0 where the term has fewer than ks[j] positive values.  ks i32[nk] on the device,
 * 1 <= ks[j] <= 1024, nk <= 64.  *neg_flag (device i32) is set to 1 when a value is negative: the bounds must then not be
 * used.  One streaming pass (the selection machinery of the search kernels); replaces no reference code: the reference has
 * no score bounds.  Reads ks back to the host first (a stream synchronisation: this is a build step, not a search step). */
int srx_build_term_bounds(int32_t device, int32_t val_type, const int64_t *term_ptr, const void *post_val, int64_t vocab,
                          const int32_t *ks, int32_t nk, float *out_bound, int32_t *neg_flag, void *stream);

/* Compact copy of the blocks for the tier-1 kernel (replaces nothing in the reference: a storage choice of this engine).
 * Block b of `post` (n_blocks_total = n_blocks + SRX_BLOCK_PAD of them) becomes
 *     SRX_VAL_F32: [l0 | l1 << 16][l2 | l3 << 16][v0 v1 v2 v3]   6 words
 *     SRX_VAL_F16: [l0 | l1 << 16][l2 | l3 << 16][h0 h1][h2 h3]  4 words
 * with l = doc - first doc of the doc's unit (units of unit_tiles << tile_log2 <= 49152 docs); a sentinel (doc -1 - 32 x)
 * becomes 49152 + 32 (x mod 64).  Derived data: not stored in shard files, rebuilt after a load. */
int srx_build_compact(int32_t device, int32_t val_type, const int32_t *post, int64_t n_blocks_total, int32_t tile_log2,
                      int32_t unit_tiles, int32_t *out_post16, void *stream);

/* hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream): the copy of a batch's result rows to pinned host memory (or
 * of a query batch to the device) on a copy stream of the caller's, without going through a framework's stream guard
 * (the host-side mirror measured ~1 ms per copy call that way: more than a whole C2 search). */
int srx_memcpy_async(void *dst, const void *src, int64_t bytes, void *stream);

/* ---- profiling (bench.py roofline leg) ------------------------------------------------------------ */
/* Dense INT8 side of the same service.  Replaces quantized_dot_product_batch
 * (rag_system/core/retriever_registry.py:90-117; NumPy twin :538-548) + the top-k that follows it (:505-519):
 *   score[q][d] = f32( f64(sum_i queries[q][i] * corpus[d][i]) * query_scale[q] * corpus_scale[d] )   (int8 x int8 -> int32,
 *   scaled in fp64 like the reference's NumPy scalars), results = the k largest scores > 0 per query, ranked
 *   (score desc, doc asc), ids = doc_base + row, rows padded with -1 / 0, counts in out_count.
 * corpus i8[n_docs][dim], queries i8[nq][dim] row-major, 16-byte aligned, dim in {32,64,96,128,192,256,384,512,768,1024}
 * (pad rows with zeros otherwise); all pointers are device pointers; asynchronous on `stream`. */
int64_t srx_dense_workspace_bytes(int32_t nq, int64_t n_docs, int32_t k);
int srx_dense_search_i8(int32_t device, const int8_t *corpus, const float *corpus_scale, int64_t n_docs, int32_t dim,
                        const int8_t *queries, const float *query_scale, int32_t nq, int32_t k, int64_t doc_base,
                        int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace, int64_t workspace_bytes,
                        void *stream);
/* The same search on a corpus kept in MFMA-fragment order (what the retriever keeps resident instead of the row-major
 * matrix the reference holds, retriever_registry.py:389-392: a storage choice of this engine, same bytes): for tile
 * T = 32 rows, k-step s (32 columns) and lane l the 16 bytes rows[32 T + (l & 31)][32 s + 16 (l >> 5) ..] sit at
 * packed + ((T * dim / 32 + s) * 64 + l) * 16; rows past n_rows are zeros.  A wave's B-fragment load is then one contiguous
 * KiB instead of a load that touches 32 rows.  srx_dense_packed_bytes = the size of that buffer; srx_dense_pack_i8
 * builds it from the row-major matrix (16-byte aligned device pointers).  Results are identical to srx_dense_search_i8. */
int64_t srx_dense_packed_bytes(int64_t n_rows, int32_t dim);
int srx_dense_pack_i8(int32_t device, const int8_t *rows, int64_t n_rows, int32_t dim, void *out_packed, void *stream);
int srx_dense_search_i8_packed(int32_t device, const void *corpus_packed, const float *corpus_scale, int64_t n_docs, int32_t dim,
                               const int8_t *queries, const float *query_scale, int32_t nq, int32_t k, int64_t doc_base,
                               int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                               int64_t workspace_bytes, void *stream);

/* Dense f32 side: replaces np.dot(embedding_index, query_vector) + the top-k after it in
 * RetrievalService.search_by_vector (rag_system/core/retrieval.py:411-423).  emb f32[n_docs][dim], queries f32[nq][dim],
 * dim a multiple of 64, <= 1024 (pad with zeros).  Results: the k largest values of (score + score_offset) > 0 per query,
 * ranked (desc, doc asc), returned WITH the offset.  score_offset = 0 gives the positive scores themselves; an offset
 * above the largest |score| makes every doc rankable, which is how the host side serves search_by_vector's
 * min_score <= 0 (the reference returns zero / negative scores down to min_score, retrieval.py:425-436) -- it then
 * re-evaluates the k returned rows' scores itself.  The summation order of the reference's BLAS matvec is unspecified:
 * scores agree to ~1e-6 relative. */
int64_t srx_dense_f32_workspace_bytes(int32_t nq, int64_t n_docs, int32_t k);
int srx_dense_search_f32(int32_t device, const float *emb, int64_t n_docs, int32_t dim, const float *queries, int32_t nq,
                         int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score, int32_t *out_count,
                         void *workspace, int64_t workspace_bytes, void *stream, float score_offset);

/* Asymmetric (uint8) scheme of the same retriever: replaces the de-quantize + np.dot loop of
 * QuantizedEmbeddingRetriever._numpy_quantized_similarity (rag_system/core/retriever_registry.py:550-559) + the top-k
 * after it (:505-519).  corpus u8[n_docs][dim] (as written by _quantize_embeddings :449-462), corpus_scales f32[2 n_docs] =
 * the reference's scale table UNCHANGED, read the way its reader reads it: doc d takes scale = corpus_scales[2 d] and
 * min = corpus_scales[2 d + 1] (:552-553; the writer :459 stores all scales, then all mins -- results follow the
 * reader, like the reference's).  score[q][d] = sum_i (corpus[d][i] * scale + min) * queries[q][i] in fp32, queries =
 * the de-quantized fp32 query vectors (u8 * query_scale + query_min, :555), dim a multiple of 64, <= 1024 (pad corpus and
 * queries with zeros).  Results: the k largest scores > 0, ranked (desc, doc asc).  Workspace:
 * srx_dense_f32_workspace_bytes.  fp32 summation order differs from the reference's BLAS dot: ~1e-6 relative. */
int srx_dense_search_u8(int32_t device, const uint8_t *corpus, const float *corpus_scales, int64_t n_docs, int32_t dim,
                        const float *queries, int32_t nq, int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score,
                        int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream);

/* Average over the profiled srx_search calls since the last read (at most the latest 256): h_ms[0] = tier-1
 * wave kernel, h_ms[1] = tier-2 block kernel, h_ms[2] = merge kernel, h_ms[3] = whole call (milliseconds,
 * hipEventElapsedTime between events recorded on the search stream around each kernel).  Synchronises the
 * events, resets the window and returns the number of calls averaged (or a negative srx_status). */
int srx_profile_read(srx_index *ix, float *h_ms4);

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_RX_H */
