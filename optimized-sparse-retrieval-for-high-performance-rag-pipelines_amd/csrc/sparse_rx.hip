// sparse_rx.hip -- tier-2 scoring kernel, merge kernels, index-build kernels and the C ABI of libsparse_rx.so
// (hand-written CDNA4 / gfx950).  The tier-1 kernel lives in wave_kernel.hip, the dense side in dense.hip, shared
// primitives in srx_common.h.
//
// Hot path replaced (paths relative to /root/reference):
//   simd_bm25_score      rag_system/core/retrieval.py:41-76      (doc-major full CSR scan per query)
//   simd_tfidf_score     rag_system/pipeline/evaluate_rag_pipeline.py:95-121
//   fast_topk_selection  rag_system/core/retrieval.py:79-92      (+ score>0 filter :292-296)
//
// Design: the index is term-major with a tile skip table; a term's postings are stored as blocks of 4
// (docs and values side by side), one padded run per unit of <= 49152 docs.  A query's doc range is cut into those
// units.  Two tiers score them, a merge kernel ranks:
//   tier 1  srx_wave_kernel   (wave_kernel.hip) ONE WAVEFRONT per (query, split), no barriers; flags what it cannot
//           serve (long runs, many multi-term docs, > 64 terms, k > 128) for tier 2.
//   tier 2  srx_score_kernel  a persistent grid of 256-thread workgroups drains the worklist of flagged (query,
//           split) blocks: block-level LDS hash units of up to 4096 postings, a greedy tile packer, and dense fp32
//           accumulators acc[G] in LDS for tiles whose postings exceed that (barrier between terms keeps the
//           summation order).  Handles everything.
//   merge   srx_merge_kernel / srx_merge_wave_kernel  exact top-k over the per-split / per-tier / per-shard lists +
//           bitonic rank by (score desc, doc asc).  Queries that tier 1 finished on its own are skipped.
// No MFMA (sparse gather/reduce, HBM-bound), no float atomics (LDS ds_add_f32 serialises at ~192 cycles per
// wave-instruction on gfx950, and sums must be deterministic).

#include "srx_common.h"

thread_local char srx_g_err[512] = "";

#ifdef SRX_STAMP2
__device__ unsigned long long g_stamp2[16];
#define T2_T0() unsigned long long t2_prev = __builtin_amdgcn_s_memtime()
#define T2(i)                                                                    \
    do {                                                                         \
        if (threadIdx.x == 0) {                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();           \
            atomicAdd(&g_stamp2[i], t_ - t2_prev);                                \
            t2_prev = t_;                                                        \
        }                                                                        \
    } while (0)
#define T2C(i) do { if (threadIdx.x == 0) atomicAdd(&g_stamp2[i], 1ull); } while (0)
#define T2L_T0() unsigned long long t2l_prev = __builtin_amdgcn_s_memtime()
#define T2L(i)                                                                   \
    do {                                                                         \
        if (threadIdx.x == 0) {                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();           \
            atomicAdd(&g_stamp2[i], t_ - t2l_prev);                               \
            t2l_prev = t_;                                                       \
        }                                                                        \
    } while (0)
#else
#define T2L_T0() do { } while (0)
#define T2L(i) do { } while (0)
#define T2_T0() do { } while (0)
#define T2(i) do { } while (0)
#define T2C(i) do { } while (0)
#endif

namespace {

// Scoring kernel: one workgroup per (query, split of the doc range).
// ------------------------------------------------------------------------------------------------
struct ScoreShared {
    unsigned tbl[TBL_WORDS];  // hash keys [0,SLOTS) + vals [SLOTS,2*SLOTS)  |  dense acc[G]  |  radix hist
    TopkShared tk;
    int64_t m_start[MAXT];  // first posting of term i inside the current unit
    int m_len[MAXT];        // postings of term i inside the current unit
    float m_idf[MAXT];
    float m_qw[MAXT];
    unsigned short st_term[MAX_STEPS];  // step table of a hash unit: (term, first posting of the 256-chunk)
    int st_off[MAX_STEPS];
    int ptile[MAX_TPS + 1];  // overflow packer: postings per tile / group boundaries
    int grp[MAX_TPS + 1];
    int n_grp;
    unsigned ub_bits;  // srx_search_after: only candidates ranked strictly AFTER (ub_bits, ub_doc) in (score desc, doc asc)
    int ub_doc;        // order are collected; ub_bits = 0xFFFFFFFF: no bound (every score's bit pattern is below it)
#ifdef SRX_T2_PAD_WORDS
    int occupancy_pad[SRX_T2_PAD_WORDS];  // dev experiment: more LDS per workgroup = fewer workgroups per CU
#endif
};

// srx_search_after's exclusive upper bound on (score bits, shard-local doc): true when the candidate ranks after it
// AFTER = false is the instance plain searches run: the test (two LDS reads + compares wherever a candidate is formed)
// measured 5.5 % of a C4 batch and 4 % of a C5 batch (profiles/r03_ab_tier2_after_bound.log).
template <bool AFTER>
__device__ __forceinline__ bool after_bound(const ScoreShared &S, unsigned b, int doc) {
    if constexpr (!AFTER) return true;
    return b < S.ub_bits || (b == S.ub_bits && doc > S.ub_doc);
}

// Rank the block's final list (tk.count <= k entries, unordered) by (score desc, doc asc) -- bitonic sort of 64-bit keys
// score bits : 0x7FFFFFFF - doc in `sortkey` (KMAX words of LDS) -- and write the padded result row.
__device__ void block_rank_emit(const TopkShared &tk, unsigned long long *sortkey, int k, int64_t doc_base,
                                int32_t *__restrict__ row_doc, float *__restrict__ row_score, int32_t *__restrict__ row_count) {
    const int tid = threadIdx.x;
    const unsigned cnt = tk.count;
    unsigned n = 1;
    while (n < cnt) n <<= 1;
    for (unsigned i = tid; i < n; i += THREADS)
        sortkey[i] = i < cnt ? (((unsigned long long)tk.bits[i] << 32) | (0x7FFFFFFFu - (unsigned)tk.doc[i])) : 0ull;
    __syncthreads();
    for (unsigned size = 2; size <= n; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned i = tid; i < (n >> 1); i += THREADS) {
                const unsigned pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = sortkey[pos], b = sortkey[pos + stride];
                const bool desc = (pos & size) == 0;
                if (desc ? (a < b) : (a > b)) {
                    sortkey[pos] = b;
                    sortkey[pos + stride] = a;
                }
            }
            __syncthreads();
        }
    }
    for (unsigned i = tid; i < (unsigned)k; i += THREADS) {
        if (i < cnt) {
            const unsigned long long x = sortkey[i];
            row_doc[i] = (int32_t)(doc_base + (int64_t)(0x7FFFFFFFu - (unsigned)(x & 0xFFFFFFFFull)));
            row_score[i] = __uint_as_float((unsigned)(x >> 32));
        } else {
            row_doc[i] = -1;
            row_score[i] = 0.0f;
        }
    }
    if (tid == 0) *row_count = (int)cnt;
}

// Where the tier-2 kernel writes FINAL rows (queries that are one work item: nothing is left for the merge kernel) and the
// worklist length it reports back to the host (pinned word, read without synchronisation by the next call: a hint only).
struct Tier2Final {
    int32_t *out_doc;
    float *out_score;
    int32_t *out_count;
    int64_t ors, ocs;
    int *hint;
};

// Hash-accumulate the unit described by m_start/m_len (P <= HASH_CAP postings) and fold its positive
// scores into the running top-k.  nt = terms in this pass.
// CP: the index dropped its canonical blocks -- postings come from the compact copy (16-bit local ids + ubase = the unit's
// first doc); every posting of a call then lies in ONE build unit (the host refuses unit overrides on such an index).
template <typename VT, bool AFTER, bool CP>
__device__ void hash_unit(ScoreShared &S, const IndexView &ix, int nt, int my_len, int k, int ubase, int dbg = 0) {
    const int tid = threadIdx.x;
    int *keys = reinterpret_cast<int *>(S.tbl);
    float *vals = reinterpret_cast<float *>(S.tbl + SLOTS);
    const int32_t *post = CP ? ix.post16 : ix.post;

    // step table: term i contributes ceil(len_i / 256) steps
    const unsigned my_chunks = (tid < nt) ? (unsigned)((my_len + THREADS - 1) / THREADS) : 0u;
    unsigned n_steps;
    const unsigned first = block_excl_scan(my_chunks, S.tk.red, &n_steps);
    for (unsigned c = 0; c < my_chunks; ++c) {
        S.st_term[first + c] = (unsigned short)tid;
        S.st_off[first + c] = (int)(c * THREADS);
    }
    __syncthreads();

    for (unsigned s0 = 0; s0 < n_steps; s0 += PREFETCH) {
        int d[PREFETCH];
        float v[PREFETCH];
#pragma unroll
        for (int r = 0; r < PREFETCH; ++r) {
            const unsigned s = s0 + r;
            d[r] = -1;
            v[r] = 0.f;
            if (s < n_steps) {
                const int i = S.st_term[s];
                const int p = S.st_off[s] + tid;
                if (p < S.m_len[i]) {
                    const int64_t g = S.m_start[i] + p;
                    d[r] = CP ? post16_doc_at<VT>(post, g, ubase) : post_doc_at<VT>(post, g);  // sentinels (run padding) read as negative docs: skipped below
                    v[r] = CP ? post16_val_at(post, g, VT()) : post_val_at(post, g, VT());
                }
            }
        }
#pragma unroll
        for (int r = 0; r < PREFETCH; ++r) {
            const unsigned s = s0 + r;
            if (s < n_steps) {
                const int i = S.st_term[s];
                if (s > 0 && S.st_term[s - 1] != i) __syncthreads();  // next term: order adds per doc
                if (d[r] >= 0 && !(dbg & 2)) {
                    const float c = (v[r] * S.m_idf[i]) * S.m_qw[i];
                    unsigned h = ((unsigned)d[r] * 0x9E3779B1u) >> (32 - 13);
                    for (;;) {
                        const int old = atomicCAS(&keys[h], EMPTY_KEY, d[r]);
                        if (old == EMPTY_KEY) {
                            vals[h] = 0.0f + c;
                            break;
                        }
                        if (old == d[r]) {
                            vals[h] = vals[h] + c;
                            break;
                        }
                        h = (h + 1) & (SLOTS - 1);
                    }
                }
            }
        }
    }
    __syncthreads();
    // read the table into registers (4 consecutive slots per access), clear the keys behind us
    unsigned ubits[NPT_HASH];
    int udoc[NPT_HASH];
    const unsigned tau = S.tk.tau;
#pragma unroll
    for (int j = 0; j < NPT_HASH / 4; ++j) {
        const int q4 = j * THREADS + tid;
        const int4 kk = reinterpret_cast<const int4 *>(keys)[q4];
        const float4 vv = reinterpret_cast<const float4 *>(vals)[q4];
        reinterpret_cast<int4 *>(keys)[q4] = make_int4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
        const int ks[4] = {kk.x, kk.y, kk.z, kk.w};
        const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned b = __float_as_uint(vs[c]);
            const bool ok = ks[c] != EMPTY_KEY && vs[c] > 0.0f && b >= tau && after_bound<AFTER>(S, b, ks[c]);
            ubits[j * 4 + c] = ok ? b : 0u;
            udoc[j * 4 + c] = ks[c];
        }
    }
    __syncthreads();  // table is free from here: vals region doubles as the radix histogram
    if (!(dbg & 1)) topk_fold<NPT_HASH, true>(ubits, udoc, k, S.tk, S.tbl + SLOTS);
}

// Dense-accumulate one tile of G docs [tile_base, tile_base + G) described by m_start/m_len.
// first_pass: zero the accumulators; last_pass: select.  (Queries with > MAXT terms take several passes.)
template <typename VT, bool CP>
__device__ void dense_tile_accumulate(ScoreShared &S, const IndexView &ix, int nt, int tile_base, bool first_pass, int ubase) {
    const int tid = threadIdx.x;
    float *acc = reinterpret_cast<float *>(S.tbl);
    const int G = 1 << ix.tile_log2;
    const int32_t *post = CP ? ix.post16 : ix.post;
    constexpr int BW = CP ? CompactWords<VT>::value : BlockWords<VT>::value;
    if (first_pass) {
        for (int i = tid; i < G / 4; i += THREADS) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    // Batches of 2048 postings (two stripes of whole blocks = 8 postings per thread) are enumerated term-major; a ring
    // of SRX_DENSE_DEPTH batches is in flight, across term boundaries too, so a term's load latency hides behind the
    // previous terms' work (one batch ahead left the dense tiles latency-bound).  A barrier separates consecutive
    // batches of different terms (the next term may touch the same doc).  A tile's run [start, start + len) starts at
    // an arbitrary padded position: the batches cover the blocks from start & ~3 on, postings outside the run and
    // sentinels (doc -1) are blanked.
#ifndef SRX_DENSE_NB
#define SRX_DENSE_NB 8
#endif
    constexpr int NB = SRX_DENSE_NB;      // postings per thread per batch (whole blocks of 4)
    constexpr int BATCH = THREADS * NB;
    auto next_term = [&](int i) {  // first term index >= i with postings in this tile (uniform), nt if none
        while (i < nt && S.m_len[i] == 0) ++i;
        return i;
    };
    auto span_of = [&](int i) { return (int)(S.m_start[i] & 3) + S.m_len[i]; };  // postings from the first block's start
    auto load_batch = [&](int i_, int o, int (&d)[NB], float (&v)[NB]) {
        const bool valid = i_ < nt;        // past the last batch: the loads are still issued (a constant number in flight ->
        const int i = valid ? i_ : 0;      // counted vmcnt waits), everything masked
        const int64_t start = S.m_start[i];
        const int head = (int)(start & 3);
        const int span = valid ? head + S.m_len[i] : 0;
        const int64_t b0 = start >> 2;
#pragma unroll
        for (int h = 0; h < NB / 4; ++h) {
            const int p = o + h * (THREADS * 4) + tid * 4;  // my block of this stripe, in postings from the first block
            const int64_t blk = (p < span) ? b0 + (p >> 2) : b0;  // idle threads re-read the run's first block (always valid)
            int dd[4];
            float vv[4];
            if constexpr (CP)
                load_block16(post + blk * BW, VT(), dd, vv);  // unit-local ids
            else
                load_block(post + blk * BW, VT(), dd, vv);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // CP: d = the accumulator index inside the tile (local id - the tile's offset in its unit); a sentinel's
                // (local id >= 49152) lies outside [0, G) for every tile of a unit and is skipped by add_batch
                const int dc = CP ? dd[c] - (tile_base - ubase) : dd[c];
                d[4 * h + c] = (p + c >= head && p + c < span) ? dc : -1;
                v[4 * h + c] = vv[c];
            }
        }
    };
    auto add_batch = [&](int i, const int (&d)[NB], const float (&v)[NB]) {
        const float idf = S.m_idf[i], qw = S.m_qw[i];
        // The postings of one batch belong to one term, so their docs are distinct: read all accumulators, then write
        // them all (written as one loop of read-modify-writes the compiler has to assume the addresses may alias and
        // serialises NB LDS round trips per batch -- the dense tiles' main stall before).
        // Branch-free: a masked posting (doc < 0) reads and writes a private dummy word instead of an accumulator (as
        // per-posting branches the compiler emitted one exec-masked block and one LDS wait per posting).
        float *const dummy = reinterpret_cast<float *>(S.st_off) + (tid & 63);  // the hash path's step table is idle here
        float *slot[NB];
        float acc_r[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
            slot[r] = CP ? (((unsigned)d[r] < (unsigned)G) ? acc + d[r] : dummy) : ((d[r] >= 0) ? acc + (d[r] - tile_base) : dummy);
#pragma unroll
        for (int r = 0; r < NB; ++r) acc_r[r] = *slot[r];
#pragma unroll
        for (int r = 0; r < NB; ++r) *slot[r] = acc_r[r] + (v[r] * idf) * qw;
    };
    // K batches in flight: register set j holds batch n with n % K == j; after batch n has been accumulated its set is
    // refilled with batch n + K.  (One batch ahead left a many-term tile -- 50 terms of < 1 batch each -- paying one full
    // memory round trip per term: profiles/r02_c4_*.)
#ifndef SRX_DENSE_DEPTH
#define SRX_DENSE_DEPTH 4
#endif
    constexpr int K = SRX_DENSE_DEPTH;
    int qi[K], qo[K];  // term / offset of the batch in set j (qi == nt: none)
    int dq[K][NB];
    float vq[K][NB];
    int ni = next_term(0), no = 0;  // the next batch to load
    auto advance = [&]() {          // (ni, no) -> its successor in term-major order
        no += BATCH;
        if (no >= span_of(ni)) {
            ni = next_term(ni + 1);
            no = 0;
        }
    };
    if (ni >= nt) return;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        qi[j] = ni;
        qo[j] = no;
        load_batch(ni, no, dq[j], vq[j]);
        if (ni < nt) advance();
    }
    while (qi[0] < nt) {  // set 0 holds the oldest batch at the top of the loop
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (qi[j] < nt) add_batch(qi[j], dq[j], vq[j]);  // uniform
            const int nxt = qi[(j + 1) % K];  // term of the batch that is accumulated next
            if (qi[j] < nt && nxt < nt && nxt != qi[j]) __syncthreads();  // next term: order the adds per doc
            qi[j] = ni;
            qo[j] = no;
            load_batch(ni, no, dq[j], vq[j]);
            if (ni < nt) advance();
        }
    }
    __syncthreads();
}

// One tile (<= 2^14 docs) holding P <= FLAT_CAP postings of MANY terms (learned-sparse queries: 50 terms with ~80
// postings each), all terms at once instead of term by term with a barrier and a memory round trip per term:
//   1. every posting sets its doc's bit in an LDS bitmap; a bit found set marks the doc in a second bitmap (multi-term);
//   2. second pass (postings come from L1/L2 now): a posting of a single-term doc is the doc's whole score (0 + c) and
//      becomes a candidate directly; postings of multi-term docs (a few %) go to an LDS list;
//   3. the list is grouped by doc (hash claim + count + scan + scatter) and each doc's contributions are added in
//      ascending term order by one thread -- the reference's accumulation order, exactly;
//   4. singles and multis are folded into the block's running top-k.
// Returns false (nothing folded, LDS scratch only) when more than FLAT_MCAP postings belong to multi-term docs: the
// caller then uses the dense accumulators.
constexpr int FLAT_CAP = 8192;                 // postings per flat tile (32 per thread)
constexpr int FLAT_NPT = FLAT_CAP / THREADS;
constexpr int FLAT_MCAP = 2048;                // multi-term postings per flat tile
constexpr int FLAT_SLOTS = 2048;               // doc hash slots of the grouping step (>= 2 x docs: a multi doc has >= 2 postings)
constexpr int FLAT_MPT = FLAT_MCAP / THREADS;  // 8
constexpr int FLAT_MIN_TERMS = 12;             // below this the term-by-term paths are at least as good

template <typename VT, bool AFTER, bool CP>
__device__ bool flat_tile(ScoreShared &S, const IndexView &ix, int nt, int my_len, int tile_base, int k) {  // one-tile units: the tile IS the unit
    const int tid = threadIdx.x;
    unsigned *bm1 = S.tbl, *bm2 = S.tbl + 512;
    int *pre = reinterpret_cast<int *>(S.tbl + 1024);  // [nt + 1] exclusive prefix of m_len: flat posting index -> term
    unsigned *mcount = S.tbl + 1024 + MAXT + 1;        // multi-term postings collected
    int *mk_key = reinterpret_cast<int *>(S.tbl + 2048);
    float *mk_c = reinterpret_cast<float *>(S.tbl + 4096);
    int *so_key = reinterpret_cast<int *>(S.tbl + 6144);
    float *so_c = reinterpret_cast<float *>(S.tbl + 8192);
    int *hk = reinterpret_cast<int *>(S.tbl + 10240);
    int *hcnt = reinterpret_cast<int *>(S.tbl + 12288);
    int *hoff = reinterpret_cast<int *>(S.tbl + 14336);
    const int32_t *post = CP ? ix.post16 : ix.post;

    unsigned P;
    const unsigned first = block_excl_scan(tid < nt ? (unsigned)my_len : 0u, S.tk.red, &P);
    if (tid < nt) pre[tid] = (int)first;
    if (tid == 0) {
        pre[nt] = (int)P;
        *mcount = 0;
    }
    reinterpret_cast<uint4 *>(S.tbl)[tid] = make_uint4(0u, 0u, 0u, 0u);  // both bitmaps: 1024 words
    for (int i = tid; i < FLAT_SLOTS; i += THREADS) {
        hk[i] = EMPTY_KEY;
        hcnt[i] = 0;
    }
    __syncthreads();
    // ---- 1. mark ----
    {
        int i = 0;
        for (int f = tid; f < (int)P; f += THREADS) {
            while (f >= pre[i + 1]) ++i;
            const int da = CP ? post16_doc_at<VT>(post, S.m_start[i] + (f - pre[i]), tile_base) : post_doc_at<VT>(post, S.m_start[i] + (f - pre[i]));
            if (da >= 0) {  // not a sentinel
                const int d = da - tile_base;
                const unsigned bit = 1u << (d & 31);
                if (atomicOr(&bm1[d >> 5], bit) & bit) atomicOr(&bm2[d >> 5], bit);
            }
        }
    }
    __syncthreads();
    // ---- 2. classify: singles to registers, multi postings to the list ----
    unsigned ubits[FLAT_NPT];
    int udoc[FLAT_NPT];
    const unsigned tau = S.tk.tau;
    {
        int i = 0;
#pragma unroll
        for (int n = 0; n < FLAT_NPT; ++n) {
            const int f = n * THREADS + tid;
            ubits[n] = 0u;
            udoc[n] = 0;
            if (f < (int)P) {
                while (f >= pre[i + 1]) ++i;
                const int64_t g = S.m_start[i] + (f - pre[i]);
                const int da = CP ? post16_doc_at<VT>(post, g, tile_base) : post_doc_at<VT>(post, g);
                const int d = da - tile_base;
                const float c = ((CP ? post16_val_at(post, g, VT()) : post_val_at(post, g, VT())) * S.m_idf[i]) * S.m_qw[i];
                if (da < 0) {  // sentinel: nothing
                } else if ((bm2[d >> 5] >> (d & 31)) & 1u) {
                    const unsigned e = atomicAdd(mcount, 1u);
                    if (e < (unsigned)FLAT_MCAP) {
                        mk_key[e] = (d << 8) | i;
                        mk_c[e] = c;
                    }
                } else {
                    const float sc = 0.0f + c;
                    const unsigned b = __float_as_uint(sc);
                    if (sc > 0.0f && b >= tau && after_bound<AFTER>(S, b, tile_base + d)) {
                        ubits[n] = b;
                        udoc[n] = tile_base + d;
                    }
                }
            }
        }
    }
    __syncthreads();
    const unsigned M = *mcount;
    if (M > (unsigned)FLAT_MCAP) return false;  // uniform
    // ---- 3. group the multi postings by doc ----
    int slot[FLAT_MPT];
#pragma unroll
    for (int j = 0; j < FLAT_MPT; ++j) {
        const unsigned e = j * THREADS + tid;
        slot[j] = -1;
        if (e < M) {
            const int d = mk_key[e] >> 8;
            unsigned h = ((unsigned)d * 0x9E3779B1u) >> (32 - 11);
            for (;;) {
                const int old = atomicCAS(&hk[h], EMPTY_KEY, d);
                if (old == EMPTY_KEY || old == d) break;
                h = (h + 1) & (FLAT_SLOTS - 1);
            }
            slot[j] = (int)h;
            atomicAdd(&hcnt[h], 1);
        }
    }
    __syncthreads();
    {
        constexpr int SPT = FLAT_SLOTS / THREADS;  // 8 consecutive slots per thread
        int c8[SPT];
        unsigned mine = 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            c8[j] = hcnt[tid * SPT + j];
            mine += (unsigned)c8[j];
        }
        unsigned tot;
        unsigned run = block_excl_scan(mine, S.tk.red, &tot);
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            hoff[tid * SPT + j] = (int)run;
            run += (unsigned)c8[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FLAT_MPT; ++j) {
        if (slot[j] >= 0) {
            const unsigned e = j * THREADS + tid;
            const int pos = hoff[slot[j]] + atomicSub(&hcnt[slot[j]], 1) - 1;
            so_key[pos] = mk_key[e];
            so_c[pos] = mk_c[e];
        }
    }
    __syncthreads();
    // one thread per doc slot: contributions in ascending term order
    unsigned mbits[FLAT_MPT];
    int mdoc[FLAT_MPT];
#pragma unroll
    for (int j = 0; j < FLAT_MPT; ++j) {
        const int sl = tid * FLAT_MPT + j;
        const int a = hoff[sl];
        const int b = (sl + 1 < FLAT_SLOTS) ? hoff[sl + 1] : (int)M;
        mbits[j] = 0u;
        mdoc[j] = 0;
        if (b > a) {
            float sum = 0.0f;
            int last = -1;
            for (int n = a; n < b; ++n) {  // selection by term: b - a is 2 or 3 almost always
                int best = 0x7FFFFFFF, bi = a;
                for (int m = a; m < b; ++m) {
                    const int t = so_key[m] & 0xFF;
                    if (t > last && t < best) {
                        best = t;
                        bi = m;
                    }
                }
                sum = sum + so_c[bi];
                last = best;
            }
            const unsigned bb = __float_as_uint(sum);
            if (sum > 0.0f && bb >= tau && after_bound<AFTER>(S, bb, tile_base + (so_key[a] >> 8))) {
                mbits[j] = bb;
                mdoc[j] = tile_base + (so_key[a] >> 8);
            }
        }
    }
    __syncthreads();  // the scratch is free from here: it doubles as the radix histogram of the folds
    topk_fold<FLAT_NPT, true>(ubits, udoc, k, S.tk, S.tbl);
    topk_fold<FLAT_MPT, true>(mbits, mdoc, k, S.tk, S.tbl);
    return true;
}

// Dense accumulation of ONE tile by ONE wavefront (tiles of <= 4096 docs: four waves' accumulators fit the 64 KiB table,
// so a workgroup takes four consecutive tiles at a time).  No block barrier anywhere: the wave streams the tile's runs
// term by term in the query's term order and one wave's LDS instructions execute in order, which is all the per-doc
// summation order needs.  Lane i < nt carries term i's run in this tile (wstart / wlen) and its weights (my_idf /
// my_qw); K blocks per lane are in flight across term boundaries.  The block kernel's term-by-term form costs a barrier
// and a memory round trip per term: on 50-term learned-sparse queries (C4) that was 85 % of its time.
// ALIGNED (the index has one-tile units: every run of a tile starts on a block boundary and ends in sentinels): only the
// sentinel test is left of the masks, idle lanes read their own all-sentinel block.
// (Measured and dropped: adding with the LDS float atomic ds_add_f32 instead of read / add / write.  It is bit-identical
// to v_add_f32 and ordered -- tools/lds_fadd_probe.hip -- and needs half the instructions, but the LDS executes it at
// about one lane every 7 cycles: C4 went from 47 ms to 166 ms per batch.)
// (Measured and dropped: reading the compact copy of srx_common.h here on one-tile units -- a local id IS the accumulator
// index, one 16-byte load per fp16 block instead of 16 + 8.  C4: 21.7 -> 20.8 ms per batch for 33 % fewer bytes: the path
// is bound by its LDS round trips, not by HBM, so the second copy's traffic saving buys little.)
template <typename VT, bool ALIGNED, bool CP>
__device__ void wave_dense_accumulate(ScoreShared &S, const IndexView &ix, int nt, int64_t tile_base, bool has_tile, int64_t wstart,
                                      int wlen, float my_idf, float my_qw) {
    constexpr int BW = CP ? CompactWords<VT>::value : BlockWords<VT>::value;
    const int ubase = (int)((((tile_base >> ix.tile_log2) / ix.unit_tiles) * ix.unit_tiles) << ix.tile_log2);  // first doc of the tile's build unit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = 1 << ix.tile_log2;
    float *acc = reinterpret_cast<float *>(S.tbl) + wave * G;
    for (int i = lane; i < G / 4; i += 64) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!has_tile) return;  // uniform per wave
    const int32_t *post = CP ? ix.post16 : ix.post;
    const int64_t idle_blk = ix.zero_block + lane;  // my all-sentinel block
    // iterator over (term, step): 64 blocks per step
    int it = 0, istep = 0;       // next (term, step) to load
    int64_t cs = 0;              // its run start / length (uniform)
    int cl = 0, cnb = 0;
    auto seek = [&]() {          // make (it, istep) point at an existing step, or it = nt
        for (;;) {
            if (it >= nt) return;
            if (istep == 0) {
                cs = ((int64_t)__builtin_amdgcn_readlane((int)(wstart >> 32), it) << 32) |
                     (unsigned)__builtin_amdgcn_readlane((int)(wstart & 0xFFFFFFFFll), it);
                cl = __builtin_amdgcn_readlane(wlen, it);
                cnb = cl > 0 ? (int)(((cs & 3) + cl + 3) >> 2) : 0;
            }
            if (istep * 64 < cnb) return;
            ++it;
            istep = 0;
        }
    };
    struct Blk {
        int d[4];
        float v[4];
        float idf, qw;  // uniform
    };
    // Loads the block of (it, istep) for this lane and advances the iterator.  ALWAYS issues its loads (past the end: an
    // all-sentinel block), so that the number of loads in flight is a compile-time constant and the waits before the adds
    // are counted vmcnt waits, not vmcnt(0).
    auto load = [&](Blk &b) {
        const bool valid = it < nt;
        const int t = valid ? it : 0;
        b.idf = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_idf), t));
        b.qw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_qw), t));
        const int bi = istep * 64 + lane;
        const bool ok = valid && bi < cnb;
        int dd[4];
        float vv[4];
        if constexpr (CP)
            load_block16(post + (ok ? (cs >> 2) + bi : idle_blk) * BW, VT(), dd, vv);  // unit-local ids; a sentinel's (>= 49152) is no
        else                                                                           // accumulator index of any tile
            load_block(post + (ok ? (cs >> 2) + bi : idle_blk) * BW, VT(), dd, vv);
        const int head = (int)(cs & 3), span = head + cl;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = bi * 4 + c;
            // CP: b.d = the accumulator index inside the tile (local id - the tile's offset in its unit: 0 on one-tile units);
            // anything outside [0, G) is skipped by add()
            const int dc = CP ? dd[c] - (int)(tile_base - ubase) : dd[c];
            b.d[c] = (ALIGNED || (ok && p >= head && p < span)) ? dc : -1;
            b.v[c] = vv[c];
        }
        if (valid) ++istep;
    };
    // branch-free: a masked posting / sentinel (doc < 0) goes to a private dummy word; one term's docs are distinct
    float *const dummy = reinterpret_cast<float *>(S.st_off) + lane;
    float *const acc0 = acc - (int)tile_base;
    auto add = [&](const Blk &b) {
        float *slot[4];
        float a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            slot[c] = CP ? (((unsigned)b.d[c] < (unsigned)G) ? acc + b.d[c] : dummy) : ((b.d[c] >= 0) ? acc0 + b.d[c] : dummy);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = *slot[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) *slot[c] = a[c] + (b.v[c] * b.idf) * b.qw;
    };
#ifndef SRX_WDENSE_DEPTH
#define SRX_WDENSE_DEPTH 4
#endif
    constexpr int K = SRX_WDENSE_DEPTH;  // blocks in flight per lane
    Blk q[K];
    bool live[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        seek();
        live[j] = it < nt;
        load(q[j]);
    }
    while (live[0]) {  // set 0 always holds the oldest block at the top of the loop
#pragma unroll
        for (int j = 0; j < K; ++j) {
            add(q[j]);  // a dead set holds sentinels / masked postings only: nothing is added
            seek();
            live[j] = it < nt;
            load(q[j]);
        }
    }
}

// Exact k-th largest over n_items keys that STAY IN LDS (keyfn(i) re-reads them in every pass; key 0 = none, keys in
// [1, 2^31)): MSD radix select with 8-bit digits, one histogram bin per thread (hist = 256 words).  The block-level
// sibling of wave_radix_kth: no per-thread key arrays, so nothing spills (the register-array form radix_kth<N> cost the
// dense tiles ~500 bytes of scratch per lane and as many HBM bytes as the postings themselves: profiles/r02_c5_*).
// Requires 1 <= k <= #candidates; mx / mn = max / min candidate key.  Returns T; n_gt = #keys > T, n_eq = #keys == T.
template <typename KeyFn>
__device__ unsigned block_radix_kth_lds(KeyFn keyfn, unsigned n_items, unsigned k, unsigned mx, unsigned mn, unsigned n_cand,
                                        unsigned *hist, unsigned *red, unsigned *n_gt, unsigned *n_eq) {
    if (mx == mn) {
        *n_gt = 0;
        *n_eq = n_cand;
        return mx;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = 31 - __clz(mx ^ mn);
    unsigned prefix = mx & ~((2u << hb) - 1u);
    int shift = hb + 1;
    unsigned krem = k, gt = 0, eq = 0;
    while (shift > 0) {
        const int w = shift < 8 ? shift : 8;
        shift -= w;
        const int hi_shift = shift + w;
        hist[tid] = 0;
        __syncthreads();
        for (unsigned i = tid; i < n_items; i += THREADS) {
            const unsigned x = keyfn(i);
            if (x != 0 && ((x ^ prefix) >> hi_shift) == 0) atomicAdd(&hist[(x >> shift) & ((1u << w) - 1u)], 1u);
        }
        __syncthreads();
        const unsigned sb = hist[tid];
        unsigned suf = sb;  // inclusive suffix sum over threads >= tid
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_down(suf, o);
            if (lane + o < 64) suf += v;
        }
        if (lane == 0) red[wave] = suf;
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww)
            if (ww > wave) suf += red[ww];
        const unsigned above = suf - sb;
        if (above < krem && krem <= suf) {
            red[8] = (unsigned)tid;
            red[9] = above;
            red[10] = sb;
        }
        __syncthreads();
        const unsigned d = red[8], ab = red[9];
        eq = red[10];
        krem -= ab;
        gt += ab;
        prefix |= d << shift;
        __syncthreads();
    }
    *n_gt = gt;
    *n_eq = eq;
    return prefix;
}

// Running list + overflow area -> the k best, tau = the k-th best.  n_total = entries appended so far: positions
// [0, KMAX) live in tk.bits / tk.doc, [KMAX, KMAX + OVF_CAP) in ovf_bits / ovf_doc.  Requires k <= n_total <=
// KMAX + OVF_CAP and k <= KMAX.  Every entry is a real candidate (key >= 1).  Same tie rule as everywhere: the smaller
// doc id wins.  Touches ~1.4 k entries instead of the tile's 16 k accumulators (dense_tile_select's general path).
constexpr int OVF_CAP = 384;  // (sizeof m_start + sizeof m_len) / 8: those tables are idle on the wave-level dense path
__device__ void list_compact_select(ScoreShared &S, int k, unsigned n_total, unsigned *ovf_bits, int *ovf_doc) {
    const int tid = threadIdx.x;
    unsigned *hist = reinterpret_cast<unsigned *>(S.st_off);
    auto key1 = [&](unsigned i) -> unsigned { return i < (unsigned)KMAX ? S.tk.bits[i] : ovf_bits[i - KMAX]; };
    auto doc_of = [&](unsigned i) -> int { return i < (unsigned)KMAX ? S.tk.doc[i] : ovf_doc[i - KMAX]; };
    constexpr int IPT = (KMAX + OVF_CAP + THREADS - 1) / THREADS;  // entries per thread
    unsigned ek[IPT];
    int ed[IPT];
    unsigned mx = 0, mn = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const unsigned i = tid + j * THREADS;
        ek[j] = i < n_total ? key1(i) : 0u;
        ed[j] = i < n_total ? doc_of(i) : 0;
        if (ek[j] != 0u) {
            mx = max(mx, ek[j]);
            mn = min(mn, ek[j]);
        }
    }
    const SumMaxMin r1 = block_sum_max_min(0u, mx, mn, S.tk.red);
    unsigned n_gt, n_eq;
    const unsigned T = block_radix_kth_lds(key1, n_total, (unsigned)k, r1.mx, r1.mn, n_total, hist, S.tk.red, &n_gt, &n_eq);
    const unsigned need = (unsigned)k - n_gt;  // ties to accept, 1 <= need <= n_eq
    unsigned T2 = 0;                            // accept ties with 0x7FFFFFFF - doc >= T2 (smaller docs first)
    if (n_eq > need) {
        auto key2 = [&](unsigned i) -> unsigned { return key1(i) == T ? 0x7FFFFFFFu - (unsigned)doc_of(i) : 0u; };
        unsigned mx2 = 0, mn2 = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < IPT; ++j)
            if (ek[j] == T && ek[j] != 0u) {
                const unsigned x = 0x7FFFFFFFu - (unsigned)ed[j];
                mx2 = max(mx2, x);
                mn2 = min(mn2, x);
            }
        const SumMaxMin r2 = block_sum_max_min(0u, mx2, mn2, S.tk.red);
        unsigned g2, e2;
        T2 = block_radix_kth_lds(key2, n_total, need, r2.mx, r2.mn, n_eq, hist, S.tk.red, &g2, &e2);
    }
    __syncthreads();  // every read of the old entries is done (the registers hold them)
    if (tid == 0) {
        S.tk.count = 0;
        S.tk.tau = T;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j)
        if (ek[j] != 0u && (ek[j] > T || (ek[j] == T && (0x7FFFFFFFu - (unsigned)ed[j]) >= T2))) {
            const unsigned p = atomicAdd(&S.tk.count, 1u);
            S.tk.bits[p] = ek[j];
            S.tk.doc[p] = ed[j];
        }
    __syncthreads();
}

// Fold the positive accumulators of a dense tile into the block's running top-k.  The accumulators stay in LDS: a
// counting pass, then either an append pass (the common case: the lazy list has room) or an exact selection over
// (list U tile candidates) whose keys are re-read from LDS.
// n_old_in >= 0 (the wave-level dense path; the caller read tk.count BEFORE its last barrier, and m_start / m_len are idle):
// one scan appends the candidates to the list and, past its capacity, to an overflow area; a selection then only touches
// those ~1.4 k entries (list_compact_select), and only when the area is full.  The caller shrinks the list back into
// tk (dense_list_flush) before anything else reads it.
template <bool AFTER>
__device__ void dense_tile_select(ScoreShared &S, const IndexView &ix, int tile_base, int k, int span_tiles = 1, int n_old_in = -1,
                                  int ovf_cap = 0) {
    const int tid = threadIdx.x;
    const float *acc = reinterpret_cast<const float *>(S.tbl);
    const int G = span_tiles << ix.tile_log2;  // accumulators in LDS: span_tiles consecutive tiles
    T2L_T0();
    if (n_old_in >= 0) {
        unsigned *ovf_bits = reinterpret_cast<unsigned *>(S.m_start);
        int *ovf_doc = reinterpret_cast<int *>(ovf_bits + OVF_CAP);
        static_assert(sizeof(S.m_start) + sizeof(S.m_len) >= OVF_CAP * 8, "overflow area");
        const int lane = tid & 63;
        unsigned n_old = (unsigned)n_old_in;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const unsigned tau_now = S.tk.tau;
            // accumulators of docs past n_docs were zeroed and never touched: no bound check.  G / 4 is a multiple of
            // THREADS (whole waves run every iteration); four float4 per thread are read before anything is tested
            auto append4 = [&](int i, const float4 a4) {  // whole waves only
                const float a[4] = {a4.x, a4.y, a4.z, a4.w};
                bool ok[4];
                unsigned long long m[4];
                unsigned tot = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ok[c] = a[c] > 0.0f && __float_as_uint(a[c]) >= tau_now && after_bound<AFTER>(S, __float_as_uint(a[c]), tile_base + 4 * i + c);
                    m[c] = __ballot(ok[c]);
                    tot += (unsigned)__popcll(m[c]);
                }
                if (tot == 0u) return;  // uniform
                unsigned base = 0;      // one atomic per wave; a candidate's slot = its rank among the wave's candidates
                if (lane == 0) base = atomicAdd(&S.tk.count, tot);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (ok[c]) {
                        const unsigned p = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[c], 0u));
                        if (p < (unsigned)KMAX) {
                            S.tk.bits[p] = __float_as_uint(a[c]);
                            S.tk.doc[p] = tile_base + 4 * i + c;
                        } else if (p < (unsigned)(KMAX + ovf_cap)) {
                            ovf_bits[p - KMAX] = __float_as_uint(a[c]);
                            ovf_doc[p - KMAX] = tile_base + 4 * i + c;
                        }
                    }
                    base += (unsigned)__popcll(m[c]);
                }
            };
            // signed-int order of the bit patterns = float order for x > 0, negatives sort below: a conservative screen
            const int tau_i = (int)max(tau_now, 1u);
            const float4 *acc4 = reinterpret_cast<const float4 *>(acc);
            auto imax4 = [](const float4 r) {
                return max(max(__float_as_int(r.x), __float_as_int(r.y)), max(__float_as_int(r.z), __float_as_int(r.w)));
            };
            const int n4 = G / 4, n4r = (n4 + 63) & ~63;  // whole waves run every iteration (tiny tiles: n4 < 64)
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            int i = tid;
            for (; i + 3 * THREADS < n4; i += 4 * THREADS) {
                const float4 r0 = acc4[i], r1 = acc4[i + THREADS], r2 = acc4[i + 2 * THREADS], r3 = acc4[i + 3 * THREADS];
                const int mm = max(max(imax4(r0), imax4(r1)), max(imax4(r2), imax4(r3)));
                // no accumulator of these 16 x 64 can enter: the common case once tau has risen
                if (__ballot(mm >= tau_i) == 0ull) continue;
                append4(i, r0);
                append4(i + THREADS, r1);
                append4(i + 2 * THREADS, r2);
                append4(i + 3 * THREADS, r3);
            }
            for (; i < n4r; i += THREADS) append4(i, i < n4 ? acc4[i] : zero4);
            __syncthreads();
            const unsigned n_total = S.tk.count;
            T2L(6);
            if (n_total <= (unsigned)(KMAX + ovf_cap)) return;  // uniform.  The list stays lazy: no selection until it is full
            // The area is full: drop this scan's appends, shrink what was there before to the k best (tau rises) and scan
            // again.  Still too many (a query's first tiles), or nothing to shrink: the general path below.
            T2C(14);
            __syncthreads();
            if (tid == 0) S.tk.count = n_old;
            __syncthreads();
            if (n_old <= (unsigned)k) break;
            T2C(13);
            list_compact_select(S, k, n_old, ovf_bits, ovf_doc);
            T2L(7);
            n_old = (unsigned)k;
        }
    }
    const unsigned tau = S.tk.tau;
    const unsigned n_old = S.tk.count;  // read BEFORE the barriers below
    const int n_valid = (int)min((int64_t)G, ix.n_docs - (int64_t)tile_base);  // docs of this tile that exist
    auto cand_key = [&](int o) -> unsigned {  // key of accumulator o: its score bits when it can enter the list, else 0
        const float x = acc[o];
        const unsigned b = __float_as_uint(x);
        return (x > 0.0f && b >= tau && after_bound<AFTER>(S, b, tile_base + o)) ? b : 0u;
    };
    unsigned mine = 0, lmx = 0, lmn = 0xFFFFFFFFu;
    for (int o = tid; o < n_valid; o += THREADS) {
        const unsigned x = cand_key(o);
        if (x != 0u) {
            ++mine;
            lmx = max(lmx, x);
            lmn = min(lmn, x);
        }
    }
    const SumMaxMin r = block_sum_max_min(mine, lmx, lmn, S.tk.red);
    const unsigned n_new = r.sum;
    if (n_new == 0 && n_old <= (unsigned)k) return;  // uniform
    if (n_old + n_new <= (unsigned)KMAX) {  // room in the lazy list: append
        for (int o = tid; o < n_valid; o += THREADS) {
            const unsigned x = cand_key(o);
            if (x != 0u) {
                const unsigned p = atomicAdd(&S.tk.count, 1u);
                S.tk.bits[p] = x;
                S.tk.doc[p] = tile_base + o;
            }
        }
        __syncthreads();
        return;
    }
    // ---- selection over (list U candidates) ----
    unsigned *hist = reinterpret_cast<unsigned *>(S.st_off);  // 256 words: the hash path's step table is idle here
    unsigned omx = 0, omn = 0xFFFFFFFFu;
    for (unsigned i = tid; i < n_old; i += THREADS) {
        omx = max(omx, S.tk.bits[i]);
        omn = min(omn, S.tk.bits[i]);
    }
    const SumMaxMin r1 = block_sum_max_min(0u, max(omx, r.mx), min(omn, r.mn), S.tk.red);
    const unsigned n_items = n_old + (unsigned)n_valid;
    auto key1 = [&](unsigned i) -> unsigned { return i < n_old ? S.tk.bits[i] : cand_key((int)(i - n_old)); };
    auto doc_of = [&](unsigned i) -> int { return i < n_old ? S.tk.doc[i] : tile_base + (int)(i - n_old); };
    unsigned n_gt, n_eq;
    const unsigned T = block_radix_kth_lds(key1, n_items, (unsigned)k, r1.mx, r1.mn, n_old + n_new, hist, S.tk.red, &n_gt, &n_eq);
    const unsigned need = (unsigned)k - n_gt;  // ties to accept, 1 <= need <= n_eq
    unsigned T2 = 0;                            // accept ties with 0x7FFFFFFF - doc >= T2 (smaller docs first)
    if (n_eq > need) {
        auto key2 = [&](unsigned i) -> unsigned { return key1(i) == T ? 0x7FFFFFFFu - (unsigned)doc_of(i) : 0u; };
        unsigned mx2 = 0, mn2 = 0xFFFFFFFFu;
        for (unsigned i = tid; i < n_items; i += THREADS) {
            const unsigned x = key2(i);
            if (x != 0u) {
                mx2 = max(mx2, x);
                mn2 = min(mn2, x);
            }
        }
        const SumMaxMin r2 = block_sum_max_min(0u, mx2, mn2, S.tk.red);
        unsigned g2, e2;
        T2 = block_radix_kth_lds(key2, n_items, need, r2.mx, r2.mn, n_eq, hist, S.tk.red, &g2, &e2);
    }
    // rebuild the list: the old entries first go to registers (KPT per thread), then everything that survives is appended
    unsigned okey[KPT];
    int odoc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const unsigned i = tid + j * THREADS;
        okey[j] = i < n_old ? S.tk.bits[i] : 0u;
        odoc[j] = i < n_old ? S.tk.doc[i] : 0;
    }
    __syncthreads();
    if (tid == 0) {
        S.tk.count = 0;
        S.tk.tau = T;
    }
    __syncthreads();
    auto keep = [&](unsigned x, int d) { return x != 0u && (x > T || (x == T && (0x7FFFFFFFu - (unsigned)d) >= T2)); };
#pragma unroll
    for (int j = 0; j < KPT; ++j)
        if (keep(okey[j], odoc[j])) {
            const unsigned p = atomicAdd(&S.tk.count, 1u);
            S.tk.bits[p] = okey[j];
            S.tk.doc[p] = odoc[j];
        }
    // candidates below the OLD tau were already excluded by cand_key (tau captured above); T >= that tau
    for (int o = tid; o < n_valid; o += THREADS) {
        const unsigned x = cand_key(o);
        if (keep(x, tile_base + o)) {
            const unsigned p = atomicAdd(&S.tk.count, 1u);
            S.tk.bits[p] = x;
            S.tk.doc[p] = tile_base + o;
        }
    }
    __syncthreads();
}

#ifndef SRX_DENSE_MIN
#define SRX_DENSE_MIN 4096
#endif
constexpr int DENSE_MIN = SRX_DENSE_MIN < HASH_CAP ? SRX_DENSE_MIN : HASH_CAP;  // a tile with more postings than this is accumulated densely

template <typename VT, bool AFTER, bool CP>
__device__ void score_block(ScoreShared &S, int bid, const IndexView &ix, const int32_t *__restrict__ q_ptr,
                            const int32_t *__restrict__ q_term, const float *__restrict__ q_weight, int nq, int k,
                            int n_splits, int n_whole, int tpu, int n_super, int dbg, const unsigned *__restrict__ ovf,
                            int ovf_words, int lists_per_q, int32_t *__restrict__ cand_doc,
                            float *__restrict__ cand_score, int32_t *__restrict__ cand_count,
                            const int32_t *__restrict__ after_doc, const float *__restrict__ after_score, int64_t doc_base,
                            const Tier2Final &fin) {
    const int tid = threadIdx.x;
    T2_T0();
    int q, split, nsq;
    decode_item(bid, n_whole, n_splits, q, split, nsq);
    if (q >= nq) return;
    const int64_t list = (int64_t)q * lists_per_q + n_splits + split;  // tier-2 lists follow the tier-1 lists
    const int t0 = q_ptr[q];
    const int nt_all = q_ptr[q + 1] - t0;
    // this split's supertiles [su_lo, su_hi)
    const int su_lo = (int)(((int64_t)n_super * split) / nsq);
    const int su_hi = (int)(((int64_t)n_super * (split + 1)) / nsq);
    // Tier 2 takes the whole query when tier 1 cannot serve it, otherwise only the units tier 1 flagged.
    const bool all_units = tier1_cannot_serve(ix, nt_all, k, tpu, dbg);
    const unsigned *my_ovf = ovf + (int64_t)bid * ovf_words;  // the flags tier 1's item of the same (query, split) left
    bool any = all_units && nt_all > 0;
    if (!all_units && nt_all > 0)
        for (int wd = su_lo >> 5; wd <= (su_hi - 1) >> 5 && su_lo < su_hi; ++wd) any = any || (my_ovf[wd] != 0u);
    if (!any) {  // uniform
        if (tid == 0) cand_count[list] = 0;
        return;
    }
    const int tps = tpu;  // tiles per unit
    const int row = ix.n_tiles + 1;
    int *keys = reinterpret_cast<int *>(S.tbl);

    // initial threshold from the index's per-term score bounds (see srx_wave_kernel): exact lower bound on the
    // k-th best score when every query idf is >= 0
    unsigned tau0 = 0;
    bool nonfinite = false;
    {
        const int col = bound_column(k);
        unsigned t0b = 0, negf = 0;
        for (int i = tid; i < nt_all; i += THREADS) {
            const int term = q_term[t0 + i];
            const float idf = ix.idf[term], qw = q_weight[t0 + i];
            if (!(fabsf(idf) <= 3.0e38f) || !(fabsf(qw) <= 3.0e38f)) negf |= 0x10000u;  // inf / nan weight
            if (idf < 0.0f || qw < 0.0f) {
                negf |= 1u;
            } else if (ix.term_bound != nullptr && col >= 0 && idf > 0.0f && qw > 0.0f) {
                const float b = 0.0f + (ix.term_bound[(int64_t)term * 4 + col] * idf) * qw;
                t0b = max(t0b, __float_as_uint(b > 0.0f ? b : 0.0f));
            }
        }
        const SumMaxMin r = block_sum_max_min(negf, t0b, 0u, S.tk.red);  // sum: low half = #negative, high half = #non-finite
        tau0 = (r.sum || after_score != nullptr) ? 0u : r.mx;  // the bounds speak of the k best of ALL docs, not of those after a row
        nonfinite = r.sum >= 0x10000u;
    }
    if (tid == 0) {
        S.tk.count = 0;
        S.tk.tau = tau0;
        S.ub_bits = 0xFFFFFFFFu;
        S.ub_doc = 0;
        if (after_score != nullptr) {  // rows come back as GLOBAL ids: the bound is compared in shard-local ids
            const int64_t d = (int64_t)after_doc[q] - doc_base;
            S.ub_bits = __float_as_uint(fmaxf(after_score[q], 0.0f));
            S.ub_doc = d < -1 ? -1 : d > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)d;
        }
    }
    for (int i = tid; i < SLOTS; i += THREADS) keys[i] = EMPTY_KEY;
    __syncthreads();

    const int n_pass = (nt_all + MAXT - 1) / MAXT;  // 1 unless the query has > 256 distinct terms

    if (n_pass == 1) {
        // ---- thread i owns term i ----
        const int nt = nt_all;
        int64_t base = 0;
        const int32_t *skip_row = ix.tile_skip;
        if (tid < nt) {
            const int term = q_term[t0 + tid];
            base = ix.term_ptr[term];
            skip_row = ix.tile_skip + (int64_t)term * row;
            S.m_idf[tid] = ix.idf[term];
            S.m_qw[tid] = q_weight[t0 + tid];
        }
        // wave-level dense path (tiles of <= 4096 docs, <= 64 terms): lane i of EVERY wave carries term i
        const bool wave_dense = (4 << ix.tile_log2) <= TBL_WORDS && nt <= 64 && !(dbg & 2048);
        int64_t wbase = 0;
        const int32_t *wskip = ix.tile_skip;
        float w_idf = 0.f, w_qw = 0.f;
        if (wave_dense && (tid & 63) < nt) {
            const int term = q_term[t0 + (tid & 63)];
            wbase = ix.term_ptr[term];
            wskip = ix.tile_skip + (int64_t)term * row;
            w_idf = ix.idf[term];
            w_qw = q_weight[t0 + (tid & 63)];
        }
        // one-tile units + finite weights: the unmasked form (see wave_dense_accumulate)
        const bool wd_aligned = ix.unit_tiles == 1 && !nonfinite && !(dbg & 4096);
        auto dense_quads = [&](int ja, int jb) {  // tiles [ja, jb): four at a time, one per wave, no block barriers inside
            // my term's run boundaries of the NEXT group's tile are loaded while this group is accumulated (a dependent
            // load at the top of every group exposed one memory round trip per four tiles); clamped index, no branch
            const int jlast = ix.n_tiles - 1;
            int a_n = gload_i32(wskip + min(ja + (tid >> 6), jlast)), b_n = gload_i32(wskip + min(ja + (tid >> 6), jlast) + 1);
            for (int j0 = ja; j0 < jb; j0 += WAVES) {
                const int j = j0 + (tid >> 6);
                const bool has_tile = j < jb;
                const bool mine = has_tile && (tid & 63) < nt;
                const int a = mine ? a_n : 0, b = mine ? b_n : 0;
                a_n = gload_i32(wskip + min(j + WAVES, jlast));
                b_n = gload_i32(wskip + min(j + WAVES, jlast) + 1);
                T2(0);
                if (wd_aligned)
                    wave_dense_accumulate<VT, true, CP>(S, ix, nt, (int64_t)j << ix.tile_log2, has_tile, wbase + a, b - a, w_idf, w_qw);
                else
                    wave_dense_accumulate<VT, false, CP>(S, ix, nt, (int64_t)j << ix.tile_log2, has_tile, wbase + a, b - a, w_idf, w_qw);
                const int n_old = (int)S.tk.count;  // stable here: nothing appends before the barrier
                __syncthreads();
                T2(3); T2C(11);
                if (!(dbg & 16384)) dense_tile_select<AFTER>(S, ix, j0 << ix.tile_log2, k, WAVES, (dbg & 8192) ? -1 : n_old, OVF_CAP);
                T2(4);
            }
            if (S.tk.count > (unsigned)KMAX)  // uniform (stable since the last barrier): the overflow area goes back to its owners
                list_compact_select(S, k, S.tk.count, reinterpret_cast<unsigned *>(S.m_start), reinterpret_cast<int *>(S.m_start) + OVF_CAP);
            for (int i = tid; i < SLOTS; i += THREADS) keys[i] = EMPTY_KEY;  // back to hash mode
            __syncthreads();
        };
        if (all_units && wave_dense) {
            // tier 2 has the whole query (k or the term count rules tier 1 out): units mean nothing here, the split's tile
            // range goes through the wave-level dense path in full groups of four tiles
            dense_quads(su_lo * tps, min(su_hi * tps, ix.n_tiles));
        } else
        for (int su = su_lo; su < su_hi; ++su) {
            if (!all_units && !((my_ovf[su >> 5] >> (su & 31)) & 1u)) continue;  // uniform
            int lo = 0, hi = 0;
            if (tid < nt) {
                lo = gload_i32(skip_row + min(su * tps, ix.n_tiles));
                hi = gload_i32(skip_row + min((su + 1) * tps, ix.n_tiles));
            }
            const int my_len = hi - lo;
            const unsigned P = block_sum((unsigned)my_len, S.tk.red);
            // many-term queries on a one-tile unit: all terms at once (flat_tile) instead of term by term
            const bool flat_ok = tps == 1 && nt >= FLAT_MIN_TERMS && P > 0 && P <= (unsigned)FLAT_CAP && !(dbg & 128);
            bool served = false;
            if (flat_ok) {
                if (tid < nt) {
                    S.m_start[tid] = base + lo;
                    S.m_len[tid] = my_len;
                }
                __syncthreads();
                T2(0);
                served = flat_tile<VT, AFTER, CP>(S, ix, nt, my_len, su << ix.tile_log2, k);
                T2(1); T2C(9);
                for (int i = tid; i < SLOTS; i += THREADS) keys[i] = EMPTY_KEY;  // back to hash mode
                __syncthreads();
            }
            if (served) {
            } else if (P > 0 && P <= (unsigned)DENSE_MIN) {
                if (tid < nt) {
                    S.m_start[tid] = base + lo;
                    S.m_len[tid] = my_len;
                }
                __syncthreads();
                T2(0);
                hash_unit<VT, AFTER, CP>(S, ix, nt, my_len, k, (su * tps) << ix.tile_log2, dbg);
                T2(2); T2C(10);
            } else if (P > 0 && wave_dense) {
                dense_quads(su * tps, min(su * tps + tps, ix.n_tiles));
            } else if (P > 0) {
                // ---- overflow: pack this supertile's tiles greedily into units of <= HASH_CAP postings;
                //      a single tile above that is accumulated densely ----
                const int ja = su * tps;
                const int jb = min(ja + tps, ix.n_tiles);
                const int nt_tiles = jb - ja;
                for (int j = tid; j <= nt_tiles; j += THREADS) S.ptile[j] = 0;
                __syncthreads();
                if (tid < nt) {
                    int prev = skip_row[ja];
                    for (int j = 0; j < nt_tiles; ++j) {
                        const int cur = skip_row[ja + j + 1];
                        if (cur != prev) atomicAdd(&S.ptile[j], cur - prev);
                        prev = cur;
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    int ng = 0, acc_p = 0;
                    S.grp[0] = 0;
                    for (int j = 0; j < nt_tiles; ++j) {
                        const int pj = S.ptile[j];
                        if (acc_p > 0 && acc_p + pj > DENSE_MIN) {
                            S.grp[++ng] = j;
                            acc_p = 0;
                        }
                        acc_p += pj;
                    }
                    S.grp[++ng] = nt_tiles;
                    S.n_grp = ng;
                }
                __syncthreads();
                const int ng = S.n_grp;
                for (int g = 0; g < ng; ++g) {
                    const int ga = ja + S.grp[g], gb = ja + S.grp[g + 1];
                    int glo = 0, ghi = 0;
                    if (tid < nt) {
                        glo = skip_row[ga];
                        ghi = skip_row[gb];
                    }
                    const int glen = ghi - glo;
                    const unsigned GP = block_sum((unsigned)glen, S.tk.red);
                    if (GP == 0) continue;
                    if (tid < nt) {
                        S.m_start[tid] = base + glo;
                        S.m_len[tid] = glen;
                    }
                    const int n_old = (int)S.tk.count;  // stable here: nothing appends before the barrier
                    __syncthreads();
                    if (GP <= (unsigned)DENSE_MIN) {
                        T2(0);
                        hash_unit<VT, AFTER, CP>(S, ix, nt, glen, k, (su * tps) << ix.tile_log2, dbg);
                        T2(2); T2C(10);
                    } else {  // one dense tile (gb == ga + 1 by construction)
                        const int tile_base = ga << ix.tile_log2;
                        T2(0);
                        dense_tile_accumulate<VT, CP>(S, ix, nt, tile_base, true, (su * tps) << ix.tile_log2);
                        T2(3); T2C(11);
                        dense_tile_select<AFTER>(S, ix, tile_base, k, 1, (dbg & 8192) ? -1 : n_old, 0);  // m_start / m_len are live: no overflow area
                        T2(4);
                        for (int i = tid; i < SLOTS; i += THREADS) keys[i] = EMPTY_KEY;  // back to hash mode
                        __syncthreads();
                    }
                }
            }
        }
    } else {
        // ---- general path (> MAXT query terms): tile by tile, dense accumulators, term passes in
        //      ascending order so the per-doc summation order is unchanged ----
        const int ja = su_lo * tps;
        const int jb = min(su_hi * tps, ix.n_tiles);
        for (int j = ja; j < jb; ++j) {
            const int tile_base = j << ix.tile_log2;
            const int n_old = (int)S.tk.count;  // stable: the barrier that opens every pass comes before any append
            for (int pass = 0; pass < n_pass; ++pass) {
                const int nt = min(MAXT, nt_all - pass * MAXT);
                __syncthreads();
                if (tid < nt) {
                    const int term = q_term[t0 + pass * MAXT + tid];
                    const int32_t *skip_row = ix.tile_skip + (int64_t)term * row;
                    const int a = skip_row[j], b = skip_row[j + 1];
                    S.m_start[tid] = ix.term_ptr[term] + a;
                    S.m_len[tid] = b - a;
                    S.m_idf[tid] = ix.idf[term];
                    S.m_qw[tid] = q_weight[t0 + pass * MAXT + tid];
                }
                __syncthreads();
                dense_tile_accumulate<VT, CP>(S, ix, nt, tile_base, pass == 0, ((j / ix.unit_tiles) * ix.unit_tiles) << ix.tile_log2);
            }
            dense_tile_select<AFTER>(S, ix, tile_base, k, 1, (dbg & 8192) ? -1 : n_old, 0);
        }
    }

    // ---- emit this split's list (unordered; the merge kernel ranks) ----
    __syncthreads();
    T2(0);
    topk_shrink(k, S.tk, S.tbl);
    T2(5); T2C(12);
    if (nsq == 1 && fin.out_doc != nullptr) {
        // An unsplit query is ONE work item: this block holds everything tier 1 did not score.  Fold tier 1's list of the
        // same query in (it was complete before this kernel started; its docs come from other units), rank, and write the
        // final row here -- the merge kernel only ever sees split queries.
        const int64_t l1 = (int64_t)q * lists_per_q;
        const int c1 = min(max(cand_count[l1], 0), k);
        const unsigned tau = S.tk.tau;  // k entries >= tau are in the list once a selection has run: nothing below can enter
        unsigned ub[KPT];
        int ud[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = tid + j * THREADS;
            const unsigned b = i < c1 ? __float_as_uint(cand_score[l1 * k + i]) : 0u;
            ub[j] = (b >= tau && b != 0u) ? b : 0u;
            ud[j] = i < c1 ? cand_doc[l1 * k + i] : 0;
        }
        __syncthreads();
        topk_fold<KPT, false>(ub, ud, k, S.tk, S.tbl);
        __syncthreads();
        block_rank_emit(S.tk, reinterpret_cast<unsigned long long *>(S.tbl), k, doc_base, fin.out_doc + (int64_t)q * fin.ors,
                        fin.out_score + (int64_t)q * fin.ors, fin.out_count + (int64_t)q * fin.ocs);
        if (tid == 0) cand_count[l1] = -1;  // final (as when tier 1 finishes a query on its own)
        return;
    }
    const unsigned cnt = S.tk.count;
    const int64_t o = list * k;
    for (unsigned i = tid; i < cnt; i += THREADS) {
        cand_doc[o + i] = S.tk.doc[i];
        cand_score[o + i] = __uint_as_float(S.tk.bits[i]);
    }
    if (tid == 0) cand_count[list] = (int)cnt;
}

// Tier-2 kernel: a fixed grid of workgroups drains the worklist of (query, split) blocks that tier 1 could not
// finish (flagged units, > 64 terms, k > 128).  work[0] = number of entries, work[1..] = block ids.
template <typename VT, bool AFTER, bool CP>
__global__ __launch_bounds__(THREADS, 2) void srx_score_kernel(IndexView ix, const int32_t *__restrict__ q_ptr,
                                                               const int32_t *__restrict__ q_term,
                                                               const float *__restrict__ q_weight, int nq, int k,
                                                               int n_splits, int n_whole, int tpu, int n_super, int dbg,
                                                               const unsigned *__restrict__ ovf, int ovf_words,
                                                               int lists_per_q, const int *__restrict__ work,
                                                               int32_t *__restrict__ cand_doc,
                                                               float *__restrict__ cand_score,
                                                               int32_t *__restrict__ cand_count,
                                                               const int32_t *__restrict__ after_doc,
                                                               const float *__restrict__ after_score, int64_t doc_base,
                                                               const Tier2Final fin) {
    __shared__ ScoreShared S;
    const int n_work = work[0];
    if (fin.hint != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *fin.hint = n_work;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        __syncthreads();  // the previous block's LDS state is dead
        score_block<VT, AFTER, CP>(S, work[1 + w], ix, q_ptr, q_term, q_weight, nq, k, n_splits, n_whole, tpu, n_super, dbg, ovf,
                        ovf_words, lists_per_q, cand_doc, cand_score, cand_count, after_doc, after_score, doc_base, fin);
    }
}

// ------------------------------------------------------------------------------------------------
// Merge kernel: one workgroup per (query, group of lists).  Selects the top-k of up to
// MERGE_NPT*256 candidates; if `final`, ranks them (bitonic sort on (score desc, doc asc)), adds
// doc_base and pads the row.
// ------------------------------------------------------------------------------------------------

// Wave-level final merge for the common small case (n_lists * k <= 1024 candidates per query, k <= 128: the splits /
// tiers of one shard, or 8 shards' top-100): one wavefront per query, no barrier.  The candidates are compacted into
// an LDS list, the exact list selection of tier 1 shrinks it to k and the wave ranks and writes the row.
constexpr int MW_CAP = 1024;
struct MergeWaveShared {
    static constexpr bool HIST_ALIASES_ZEROED_LDS = false;
    unsigned lbits[MW_CAP];
    int ldoc[MW_CAP];
    unsigned hist[256];
    unsigned long long sortkey[128];
};

__global__ __launch_bounds__(THREADS) void srx_merge_wave_kernel(const int32_t *__restrict__ in_doc,
                                                                 const float *__restrict__ in_score,
                                                                 const int32_t *__restrict__ in_count, int nq, int n_lists,
                                                                 int k, int gathered, int64_t row_stride,
                                                                 int64_t cnt_stride, int64_t doc_base,
                                                                 int32_t *__restrict__ out_doc,
                                                                 float *__restrict__ out_score,
                                                                 int32_t *__restrict__ out_count, int64_t out_row_stride,
                                                                 int64_t out_cnt_stride, const int *__restrict__ gate, int q0) {
    __shared__ MergeWaveShared MW[WAVES];
    const int lane = threadIdx.x & 63;
    const int q = q0 + blockIdx.x * WAVES + (threadIdx.x >> 6);  // q0: first query this launch covers (the split ones of a search)
    if (q >= nq) return;
    if (gate != nullptr && *gate == 0) return;  // optional device-side switch (dense fallback pass)
    if (!gathered && in_count[(int64_t)q * n_lists * cnt_stride] < 0) return;  // tier 1 already wrote this query's final row
    MergeWaveShared &S = MW[threadIdx.x >> 6];
    // list lengths first (one round trip), then every candidate slot of the query in one batch of loads (a second
    // round trip), then a ballot compaction of the positive scores into the LDS list
    for (int l = lane; l < n_lists; l += 64) {
        // layout 0: [nq][n_lists][k] (+ counts [nq][n_lists]); gathered: [n_lists][nq][k] (+ [n_lists][nq])
        const int64_t li = gathered ? ((int64_t)l * nq + q) : ((int64_t)q * n_lists + l);
        S.hist[l] = (unsigned)max(0, min(in_count[li * cnt_stride], k));  // n_lists <= 256 (host check); hist is free until the selection
    }
    wsync();
    constexpr int MW_NPL = MW_CAP / 64;  // candidate slots per lane
    float sc[MW_NPL];
    int dd[MW_NPL];
    const int span = n_lists * k;
#pragma unroll
    for (int j = 0; j < MW_NPL; ++j) {
        const int c = j * 64 + lane;
        sc[j] = 0.0f;
        dd[j] = 0;
        if (c < span) {
            const int l = c / k, r = c - l * k;
            if (r < (int)S.hist[l]) {
                const int64_t li = gathered ? ((int64_t)l * nq + q) : ((int64_t)q * n_lists + l);
                const int64_t a = li * row_stride + r;  // row_stride = k for plain lists, 2k+1 for packed rows
                sc[j] = in_score[a];
                dd[j] = in_doc[a];
            }
        }
    }
    unsigned count = 0;  // wave-uniform
#pragma unroll
    for (int j = 0; j < MW_NPL; ++j) {
        const bool ok = sc[j] > 0.0f;
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const unsigned p = count + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            S.lbits[p] = __float_as_uint(sc[j]);
            S.ldoc[p] = dd[j];
        }
        count += (unsigned)__popcll(m);
    }
    wsync();
    if (count > (unsigned)k) {
        wave_list_select(S, count, k);
        count = (unsigned)k;
    }
    wave_rank_emit(S, S.sortkey, count, k, doc_base, out_doc + (int64_t)q * out_row_stride, out_score + (int64_t)q * out_row_stride);
    if (lane == 0) out_count[(int64_t)q * out_cnt_stride] = (int)count;
}

__global__ __launch_bounds__(THREADS) void srx_merge_kernel(const int32_t *__restrict__ in_doc,
                                                            const float *__restrict__ in_score,
                                                            const int32_t *__restrict__ in_count, int nq, int n_lists,
                                                            int k, int lists_per_group, int n_groups, int final_pass,
                                                            int gathered, int64_t row_stride, int64_t cnt_stride,
                                                            int64_t doc_base, int32_t *__restrict__ out_doc,
                                                            float *__restrict__ out_score,
                                                            int32_t *__restrict__ out_count, int64_t out_row_stride,
                                                            int64_t out_cnt_stride, const int *__restrict__ gate, int q0) {
    __shared__ MergeShared M;
    const int tid = threadIdx.x;
    const int q = q0 + blockIdx.x / n_groups;
    const int g = blockIdx.x - (q - q0) * n_groups;
    if (q >= nq) return;
    if (gate != nullptr && *gate == 0) return;  // optional device-side switch (dense fallback pass)
    if (!gathered && in_count[(int64_t)q * n_lists * cnt_stride] < 0) return;  // tier 1 already wrote this query's final row
    const int l0 = g * lists_per_group;
    const int l1 = min(l0 + lists_per_group, n_lists);
    if (tid == 0) {
        M.tk.count = 0;
        M.tk.tau = 0;
    }
    __syncthreads();
    // candidates: flat index c -> (list, rank); lists are dense-packed logically as (l - l0)*k + r
    unsigned ubits[MERGE_NPT];
    int udoc[MERGE_NPT];
    const int span = (l1 - l0) * k;
#pragma unroll
    for (int n = 0; n < MERGE_NPT; ++n) {
        const int c = n * THREADS + tid;
        ubits[n] = 0;
        udoc[n] = 0;
        if (c < span) {
            const int l = l0 + c / k, r = c - (c / k) * k;
            // layout 0: [nq][n_lists][k] (+ counts [nq][n_lists]); gathered: [n_lists][nq][k] (+ [n_lists][nq])
            const int64_t li = gathered ? ((int64_t)l * nq + q) : ((int64_t)q * n_lists + l);
            const int cnt = in_count[li * cnt_stride];
            if (r < cnt) {
                const int64_t a = li * row_stride + r;  // row_stride = k for plain lists, 2k+1 for packed rows
                const float s = in_score[a];
                if (s > 0.0f) {
                    ubits[n] = __float_as_uint(s);
                    udoc[n] = in_doc[a];
                }
            }
        }
    }
    topk_fold<MERGE_NPT, false>(ubits, udoc, k, M.tk, M.hist);
    const unsigned cnt = M.tk.count;
    if (!final_pass) {
        const int64_t o = ((int64_t)q * n_groups + g) * k;
        for (unsigned i = tid; i < cnt; i += THREADS) {
            out_doc[o + i] = M.tk.doc[i];
            out_score[o + i] = __uint_as_float(M.tk.bits[i]);
        }
        if (tid == 0) out_count[(int64_t)q * n_groups + g] = (int)cnt;
        return;
    }
    // rank: bitonic sort, descending on key64 = score bits : (0x7FFFFFFF - doc); final rows may live in a strided (packed) buffer
    block_rank_emit(M.tk, M.sortkey, k, doc_base, out_doc + (int64_t)q * out_row_stride, out_score + (int64_t)q * out_row_stride,
                    out_count + (int64_t)q * out_cnt_stride);
}

// ------------------------------------------------------------------------------------------------
// index-build kernels
// ------------------------------------------------------------------------------------------------
__global__ void srx_impact_kernel(const float *__restrict__ tf, const int32_t *__restrict__ post_doc,
                                  const float *__restrict__ doc_len, int64_t nnz, float k1f, float bf, float omb,
                                  float k1p1, float avf, float *__restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const float t = tf[p];
        const float len = doc_len[post_doc[p]];
        const float norm = k1f * (omb + (bf * len) / avf);  // retrieval.py:58
        out[p] = (t * k1p1) / (t + norm);                   // retrieval.py:70-72
    }
}

__global__ void srx_tile_skip_kernel(const int64_t *__restrict__ term_ptr, const int32_t *__restrict__ post_doc,
                                     int64_t vocab, int n_tiles, int tile_log2, int32_t *__restrict__ out) {
    const int64_t total = vocab * (int64_t)(n_tiles + 1);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = e / (n_tiles + 1);
        const int j = (int)(e - t * (n_tiles + 1));
        const int64_t b = term_ptr[t], en = term_ptr[t + 1];
        const int64_t target = (int64_t)j << tile_log2;
        int64_t lo = b, hi = en;  // lower_bound(post_doc[b..en), target)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)post_doc[mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        out[e] = (int32_t)(lo - b);
    }
}

// ---- layout v2: scatter the term-major postings into padded runs of blocks (srx_common.h, IndexView) ----
// skip   = the UNPADDED tile skip table (srx_tile_skip_kernel on the plain CSC arrays)
// runpad = [vocab * n_units + 1] exclusive prefix of the PADDED run lengths (padded position of every run's start)
// One thread per posting: its run, its rank inside the run, its slot in the blocks; the thread of a run's last posting
// also writes the run's sentinels (doc -1, value 0).
template <typename VT>
__global__ void srx_blocks_scatter_kernel(const int64_t *__restrict__ term_ptr, const int32_t *__restrict__ post_term,
                                          const int32_t *__restrict__ post_doc, const VT *__restrict__ post_val,
                                          const int32_t *__restrict__ skip, const int64_t *__restrict__ runpad, int64_t nnz,
                                          int n_tiles, int tile_log2, int unit_tiles, int n_units,
                                          int32_t *__restrict__ out_post) {
    constexpr int BW = BlockWords<VT>::value;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const int t = post_term[p];
        const int doc = post_doc[p];
        const int u = (doc >> tile_log2) / unit_tiles;
        const int32_t *row = skip + (int64_t)t * (n_tiles + 1);
        const int ja = u * unit_tiles, jb = min(ja + unit_tiles, n_tiles);
        const int64_t i = p - term_ptr[t];               // rank inside the term
        const int64_t kr = i - row[ja];                   // rank inside the run
        int64_t dst = runpad[(int64_t)t * n_units + u] + kr;
        auto put = [&](int64_t q, int d, VT v) {
            int32_t *blk = out_post + (q >> 2) * BW;
            blk[q & 3] = d;
            reinterpret_cast<VT *>(blk + 4)[q & 3] = v;
        };
        put(dst, doc, post_val[p]);
        if (i + 1 == row[jb]) {                           // last posting of its run: pad to a multiple of 4
            // sentinel doc ids -1 - 32 * (t % 64): value 0 makes them no-ops; different terms' sentinels fall into
            // different words of the tier-1 bitmap (an LDS atomic of several lanes on ONE address serialises)
            for (++dst; (dst & 3) != 0; ++dst) put(dst, -1 - 32 * (t & 63), VT(0.0f));
        }
    }
}

// padded tile skip table + padded term offsets from the unpadded table and the run prefix
__global__ void srx_blocks_skip_kernel(const int32_t *__restrict__ skip, const int64_t *__restrict__ runpad, int64_t vocab,
                                       int n_tiles, int unit_tiles, int n_units, int32_t *__restrict__ out_skip,
                                       int64_t *__restrict__ out_term_ptr) {
    const int64_t total = vocab * (int64_t)(n_tiles + 1);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = e / (n_tiles + 1);
        const int j = (int)(e - t * (n_tiles + 1));
        const int64_t r0 = runpad[t * n_units];
        if (j == n_tiles) {
            out_skip[e] = (int32_t)(runpad[(t + 1) * n_units] - r0);  // all of the term, padding included
        } else {
            const int u = j / unit_tiles;
            out_skip[e] = (int32_t)(runpad[t * n_units + u] - r0) + (skip[e] - skip[t * (n_tiles + 1) + u * unit_tiles]);
        }
        if (j == 0) out_term_ptr[t] = r0;
        if (e == total - 1) out_term_ptr[vocab] = runpad[vocab * n_units];
    }
}

template <typename VT>
__global__ void srx_blocks_sentinel_kernel(int32_t *__restrict__ out_post, int64_t first_block, int n) {
    constexpr int BW = BlockWords<VT>::value;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // posting slot
    if (i < 4 * n) {
        int32_t *blk = out_post + (first_block + (i >> 2)) * BW;
        blk[i & 3] = -1 - 32 * ((i >> 2) & 63);  // block j: doc -1 - 32 (j mod 64) -> bitmap word 2047 - j mod 64: the idle loads of one
                                                // step (blocks lane + const) hit a word of their own per lane
        reinterpret_cast<VT *>(blk + 4)[i & 3] = VT(0.0f);
    }
}

// Compact copy for tier 1 (srx_common.h, CompactWords): one thread per block, 16-bit unit-local doc ids.
template <typename VT>
__global__ void srx_compact_blocks_kernel(const int32_t *__restrict__ post, int64_t n_blocks_total, int unit_docs,
                                          int32_t *__restrict__ out) {
    constexpr int BW = BlockWords<VT>::value, CW = CompactWords<VT>::value;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks_total) return;
    const int32_t *src = post + b * BW;
    int32_t *dst = out + b * CW;
    unsigned l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int d = src[c];
        l[c] = d >= 0 ? (unsigned)(d % unit_docs) : (unsigned)W_SENT_BASE + 32u * (unsigned)(((-1 - d) >> 5) & 63);
    }
    dst[0] = (int32_t)(l[0] | (l[1] << 16));
    dst[1] = (int32_t)(l[2] | (l[3] << 16));
#pragma unroll
    for (int c = 4; c < BW; ++c) dst[c - 2] = src[c];  // the values, unchanged
}
}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
constexpr int PROF_SLOTS = 256;
constexpr int PROF_EVENTS = 4;  // start, after tier 1, after tier 2, after merge
struct srx_index {
    srx_index_desc d;
    srx_search_opts opts;
    hipEvent_t *ev;   // PROF_SLOTS x PROF_EVENTS events, created lazily
    int ev_n;         // profiled calls recorded since the last srx_profile_read (<= PROF_SLOTS, then it wraps)
    int64_t ev_calls;
    int64_t n_calls;  // searches issued (profile = N samples every N-th of them)
    int *h_hint;      // pinned, device-mapped word: the tier-2 worklist length of a recent search (sizes the next tier-2 grid)
    int *d_hint;      // its device address
};

SRX_API int srx_version(void) { return SRX_VERSION; }
SRX_API const char *srx_last_error(void) { return g_err; }

SRX_API int srx_limits(int32_t *h_out4) {
    if (!h_out4) return fail(SRX_ERR_INVALID, "srx_limits: null output%s");
    h_out4[0] = KMAX;
    h_out4[1] = SRX_MAX_TILE_LOG2;
    h_out4[2] = HASH_CAP;
    h_out4[3] = THREADS;
    return SRX_OK;
}

SRX_API int srx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(SRX_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

SRX_API int srx_index_create(const srx_index_desc *d, srx_index **out) {
    if (!d || !out) return fail(SRX_ERR_INVALID, "srx_index_create: null argument%s");
    if (d->n_docs <= 0 || d->n_docs >= 0x7FFFFFFFll) return fail(SRX_ERR_INVALID, "srx_index_create: n_docs out of range%s");
    if (d->doc_base < 0 || d->doc_base + d->n_docs >= 0x7FFFFFFFll)
        return fail(SRX_ERR_INVALID, "srx_index_create: doc_base + n_docs must fit int32%s");
    if (d->vocab <= 0 || d->nnz < 0) return fail(SRX_ERR_INVALID, "srx_index_create: bad vocab / nnz%s");
    if (d->tile_log2 < 6 || d->tile_log2 > SRX_MAX_TILE_LOG2)
        return fail(SRX_ERR_INVALID, "srx_index_create: tile_log2 must be in [6, 14]%s");
    const int64_t nt = (d->n_docs + (1ll << d->tile_log2) - 1) >> d->tile_log2;
    if (d->n_tiles != nt) return fail(SRX_ERR_INVALID, "srx_index_create: n_tiles != ceil(n_docs / 2^tile_log2)%s");
    if (d->val_type != SRX_VAL_F32 && d->val_type != SRX_VAL_F16) return fail(SRX_ERR_INVALID, "srx_index_create: bad val_type%s");
    if (d->unit_tiles < 1 || d->unit_tiles > MAX_TPS) return fail(SRX_ERR_INVALID, "srx_index_create: unit_tiles must be in [1, 64]%s");
    if (d->n_blocks < 0 || d->n_blocks * 4 < d->nnz) return fail(SRX_ERR_INVALID, "srx_index_create: n_blocks does not cover nnz%s");
    if (!d->term_ptr || !d->tile_skip || !d->idf || (!d->post && !d->post16))
        return fail(SRX_ERR_INVALID, "srx_index_create: null index array%s");
    if (!d->post && ((int64_t)d->unit_tiles << d->tile_log2) > W_UNIT_MAX_DOCS)
        return fail(SRX_ERR_INVALID, "srx_index_create: an index without canonical blocks needs units of <= 49152 docs (the compact copy)%s");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) return fail(SRX_ERR_NODEVICE, "srx_index_create: device ordinal not visible%s");
    srx_index *ix = new (std::nothrow) srx_index();
    if (!ix) return fail(SRX_ERR_NOMEM, "srx_index_create: host allocation failed%s");
    ix->d = *d;
    memset(&ix->opts, 0, sizeof(ix->opts));
    ix->ev = nullptr;
    ix->ev_n = 0;
    ix->ev_calls = 0;
    ix->n_calls = 0;
    // 64 bytes of pinned host memory, created once with the handle (srx_search itself allocates nothing).  Not fatal when
    // it cannot be had: the tier-2 grid then always has its full size.
    ix->h_hint = nullptr;
    ix->d_hint = nullptr;
    if (hipSetDevice(d->device) == hipSuccess && hipHostMalloc((void **)&ix->h_hint, 64, hipHostMallocMapped) == hipSuccess) {
        ix->h_hint[0] = -1;  // unknown
        if (hipHostGetDevicePointer((void **)&ix->d_hint, ix->h_hint, 0) != hipSuccess) {
            (void)hipHostFree(ix->h_hint);
            ix->h_hint = nullptr;
            ix->d_hint = nullptr;
        }
    } else {
        ix->h_hint = nullptr;
        (void)hipGetLastError();
    }
    *out = ix;
    return SRX_OK;
}

SRX_API void srx_index_destroy(srx_index *ix) {
    if (!ix) return;
    if (ix->ev) {
        for (int i = 0; i < PROF_EVENTS * PROF_SLOTS; ++i) (void)hipEventDestroy(ix->ev[i]);
        delete[] ix->ev;
    }
    if (ix->h_hint) (void)hipHostFree(ix->h_hint);
    delete ix;
}

SRX_API int srx_index_set_opts(srx_index *ix, const srx_search_opts *o) {
    if (!ix || !o) return fail(SRX_ERR_INVALID, "srx_index_set_opts: null argument%s");
    if (o->supertile_log2 != 0 && (o->supertile_log2 < ix->d.tile_log2 || o->supertile_log2 > ix->d.tile_log2 + 6))
        return fail(SRX_ERR_INVALID, "srx_index_set_opts: supertile_log2 must be in [tile_log2, tile_log2+6]%s");
    if (o->unit_tiles < 0 || o->unit_tiles > MAX_TPS) return fail(SRX_ERR_INVALID, "srx_index_set_opts: unit_tiles must be in [0, 64]%s");
    if (o->target_blocks < 0) return fail(SRX_ERR_INVALID, "srx_index_set_opts: target_blocks < 0%s");
    ix->opts = *o;
    return SRX_OK;
}

namespace {
struct Plan {
    int tpu, n_super, n_splits, n_whole, ovf_words, lists_per_q;  // tpu = tiles per unit; queries < n_whole are not split
};

// Supertile (unit) = the doc range one tier-1 unit covers (<= W_UNIT_MAX_DOCS docs, the wave bitmap): the unit the
// index's runs were padded for at build time (srx_auto_unit_tiles), unless the options override it (then only tier 2
// can serve the queries).
Plan make_plan(const srx_index *ix, int nq, int k) {
    Plan p;
    const srx_index_desc &d = ix->d;
    int tpu;
    if (ix->opts.unit_tiles > 0) {
        tpu = ix->opts.unit_tiles;  // differs from the index's unit: tier 2 serves everything (tier 1 needs the padded runs)
    } else if (ix->opts.supertile_log2 != 0) {
        tpu = 1 << (ix->opts.supertile_log2 - d.tile_log2);
    } else {
        tpu = d.unit_tiles;  // the unit the index was built (padded) for: srx_auto_unit_tiles at build time
    }
    if (tpu < 1) tpu = 1;
    if (tpu > MAX_TPS) tpu = MAX_TPS;
    p.tpu = tpu;
    p.n_super = (int)((d.n_tiles + tpu - 1) / tpu);
    const int target = ix->opts.target_blocks > 0 ? ix->opts.target_blocks : 3072;  // wave-sized workgroups: 256 CUs x 12 resident waves = one full round (C2, 1 k queries: 3 splits 0.088 ms, 4 splits 0.094 ms)
    int ns = target / (nq > 0 ? nq : 1);
    int n_whole = 0;
    if (nq > target) {
        // More queries than resident waves: whole rounds of unsplit queries, and the last partial round cut finer (its
        // queries in up to 4 doc-range splits), so that the kernel does not end on a few long waves (a 10 k-query batch
        // is 3.26 rounds of 3072: measured 3 % slower per query than 9216 or 12288).
        n_whole = nq / target * target;
        const int tail = nq - n_whole;
        ns = tail > 0 ? target / tail : 1;
        if (ns > 4) ns = 4;
        if (ns < 2) ns = 2;
        if (tail == 0) ns = 1;
    }
    if (ns < 1) ns = 1;
    if (ns > p.n_super) ns = p.n_super;
    const int cap = (MERGE_NPT * THREADS) / (2 * (k > 0 ? k : 1));  // merge takes <= 4096 candidates: 2 tiers x splits x k
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    if (ns == 1) n_whole = 0;
    p.n_splits = ns;
    p.n_whole = n_whole;
    p.ovf_words = (p.n_super + 31) / 32;
    p.lists_per_q = 2 * ns;  // [0, ns): tier 1, [ns, 2 ns): tier 2
    return p;
}
}  // namespace

SRX_API int64_t srx_search_workspace_bytes(const srx_index *ix, int32_t nq, int32_t k) {
    if (!ix || nq < 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_search_workspace_bytes: bad argument%s");
    const Plan p = make_plan(ix, nq, k);
    const int64_t lists = (int64_t)nq * p.lists_per_q;
    const int64_t items = (int64_t)p.n_whole + (int64_t)(nq - p.n_whole) * p.n_splits;
    return lists * k * 8 + lists * 4 + items * p.ovf_words * 4 + 4 * (1 + (int64_t)(nq - p.n_whole)) + 4 * items + 256;
}

namespace {
int search_impl(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                int32_t k, int32_t *out_doc, float *out_score, int32_t *out_count, int64_t ors, int64_t ocs,
                void *workspace, int64_t workspace_bytes, void *stream_v, const int32_t *after_doc = nullptr,
                const float *after_score = nullptr) {
    if (!ix) return fail(SRX_ERR_INVALID, "srx_search: null index%s");
    if ((after_doc == nullptr) != (after_score == nullptr)) return fail(SRX_ERR_INVALID, "srx_search_after: after_doc and after_score go together%s");
    if (nq < 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_search: need nq >= 0 and 1 <= k <= 1024%s");
    if (nq == 0) return SRX_OK;
    if (!q_ptr || !out_doc || !out_score || !out_count) return fail(SRX_ERR_INVALID, "srx_search: null query / output pointer%s");
    const int64_t need = srx_search_workspace_bytes(ix, nq, k);
    if (!workspace || workspace_bytes < need) return fail(SRX_ERR_NOMEM, "srx_search: workspace too small%s");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ix->d.device));
    const Plan p = make_plan(ix, nq, k);
    if (ix->d.post == nullptr && p.tpu != ix->d.unit_tiles)
        return fail(SRX_ERR_INVALID, "srx_search: this index keeps no canonical blocks: a unit other than the one it was built for cannot be served%s");
    const int64_t lists = (int64_t)nq * p.lists_per_q;
    const int64_t blocks = (int64_t)p.n_whole + (int64_t)(nq - p.n_whole) * p.n_splits;
    if (lists > 0x7FFFFFFFll) return fail(SRX_ERR_INVALID, "srx_search: nq * splits overflows the grid%s");
    int32_t *cand_doc = (int32_t *)workspace;
    float *cand_score = (float *)(cand_doc + lists * k);
    int32_t *cand_count = (int32_t *)(cand_score + lists * k);
    unsigned *ovf = (unsigned *)(cand_count + lists);      // [blocks][ovf_words]
    unsigned *done = ovf + blocks * p.ovf_words;            // the zeroed region: arrival counters of the split queries [nq - n_whole] ...
    int *work = (int *)(done + (nq - p.n_whole));           // ... and work[0] = the worklist length; its entries work[1 .. blocks] follow

    IndexView v;
    v.term_ptr = ix->d.term_ptr;
    v.post = ix->d.post;
    v.post16 = ix->d.post16;
    v.zero_block = ix->d.n_blocks;
    v.unit_tiles = ix->d.unit_tiles;
    v.tile_skip = ix->d.tile_skip;
    v.idf = ix->d.idf;
    v.term_bound = ((ix->opts.reserved & 16) || after_score) ? nullptr : ix->d.term_bound;  // debug bit 16: ignore the score bounds
    v.n_docs = ix->d.n_docs;
    v.vocab = ix->d.vocab;
    v.tile_log2 = ix->d.tile_log2;
    v.n_tiles = ix->d.n_tiles;
    const int dbg = ix->opts.reserved | (after_score ? 8 : 0);  // search-after: the tier-2 kernel applies the bound, it takes every query

    const bool prof = ix->opts.profile > 0 && (ix->n_calls++ % ix->opts.profile) == 0;  // profile = N: every N-th call is bracketed
    hipEvent_t *ev = nullptr;
    if (prof) {
        if (!ix->ev) {
            ix->ev = new (std::nothrow) hipEvent_t[PROF_EVENTS * PROF_SLOTS];
            if (!ix->ev) return fail(SRX_ERR_NOMEM, "srx_search: host allocation failed%s");
            for (int i = 0; i < PROF_EVENTS * PROF_SLOTS; ++i) HIP_TRY(hipEventCreate(&ix->ev[i]));
        }
        ev = ix->ev + PROF_EVENTS * (int)(ix->ev_calls % PROF_SLOTS);
    }
    // one small memset: the worklist length and the split queries' arrival counters (a few KB: one fill launch); every other
    // slot of the workspace is initialised by the tier-1 work item that owns it
    HIP_TRY(hipMemsetAsync(done, 0, (size_t)(1 + (nq - p.n_whole)) * 4, stream));
    if (prof) HIP_TRY(hipEventRecord(ev[0], stream));
    // tier 1: one wavefront per (query, split) (wave_kernel.hip)
    {
        srx_wave_launch wl;
        wl.ix = v;
        wl.q_ptr = q_ptr; wl.q_term = q_term; wl.q_weight = q_weight;
        wl.nq = nq; wl.k = k; wl.n_splits = p.n_splits; wl.n_whole = p.n_whole; wl.n_super = p.n_super;
        wl.dbg = dbg | (p.tpu != ix->d.unit_tiles ? 8 : 0);  // another unit than the padded one: everything to tier 2
        wl.ovf = ovf; wl.ovf_words = p.ovf_words; wl.lists_per_q = p.lists_per_q; wl.work = work; wl.done = done;
        wl.cand_doc = cand_doc; wl.cand_score = cand_score; wl.cand_count = cand_count;
        wl.doc_base = ix->d.doc_base; wl.out_doc = out_doc; wl.out_score = out_score; wl.out_count = out_count;
        wl.out_row_stride = ors; wl.out_cnt_stride = ocs;
        const int rc = srx_launch_wave_kernel(wl, ix->d.val_type, blocks, stream);
        if (rc != SRX_OK) return rc;
    }
    if (prof) HIP_TRY(hipEventRecord(ev[1], stream));
    // tier 2: flagged units, long queries, k > 128 -- a persistent grid drains the worklist tier 1 filled.  Any grid size
    // is correct; when a recent search of this index left the worklist empty (the hint word the kernel writes to pinned host
    // memory, read here without synchronisation) a small grid spares an otherwise idle launch most of its dispatch time.
    const int dbg2 = dbg | (p.tpu != ix->d.unit_tiles ? 8 : 0);
    const bool t2_everything = (dbg2 & 8) != 0 || k > W1_KMAX || ix->d.post16 == nullptr;  // tier 1 serves nothing: full grid
    const int hint = ix->h_hint ? *(volatile int *)ix->h_hint : -1;
    const int64_t t2_full = blocks < 1024 ? blocks : 1024;
    const unsigned t2_grid = (unsigned)((hint == 0 && !t2_everything && t2_full > 128) ? 128 : t2_full);
    Tier2Final fin;
    fin.out_doc = out_doc; fin.out_score = out_score; fin.out_count = out_count; fin.ors = ors; fin.ocs = ocs; fin.hint = ix->d_hint;
#define SRX_LAUNCH_T2(VT, AFTER, CP)                                                                                                \
    hipLaunchKernelGGL((srx_score_kernel<VT, AFTER, CP>), dim3(t2_grid), dim3(THREADS), 0, stream, v, q_ptr, q_term, q_weight, nq, k, \
                       p.n_splits, p.n_whole, p.tpu, p.n_super, dbg2, ovf, p.ovf_words, p.lists_per_q, work, cand_doc, cand_score,   \
                       cand_count, after_doc, after_score, ix->d.doc_base, fin)
    const bool cp = ix->d.post == nullptr;  // no canonical blocks: tier 2 reads the compact copy too
    if (ix->d.val_type == SRX_VAL_F32) {
        if (after_score) { if (cp) SRX_LAUNCH_T2(float, true, true); else SRX_LAUNCH_T2(float, true, false); }
        else { if (cp) SRX_LAUNCH_T2(float, false, true); else SRX_LAUNCH_T2(float, false, false); }
    } else {
        if (after_score) { if (cp) SRX_LAUNCH_T2(__half, true, true); else SRX_LAUNCH_T2(__half, true, false); }
        else { if (cp) SRX_LAUNCH_T2(__half, false, true); else SRX_LAUNCH_T2(__half, false, false); }
    }
#undef SRX_LAUNCH_T2
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(ev[2], stream));
    // merge: only the SPLIT queries [n_whole, nq) have lists to merge (an unsplit query's final row was written by tier 1
    // or, when it had work for tier 2, by tier 2)
    const int nq_m = nq - p.n_whole;
    if (nq_m > 0) {
        if (k <= W_KMAX && (int64_t)p.lists_per_q * k <= MW_CAP && p.lists_per_q <= 256 && !(dbg & 256))
            hipLaunchKernelGGL(srx_merge_wave_kernel, dim3((unsigned)((nq_m + WAVES - 1) / WAVES)), dim3(THREADS), 0, stream, cand_doc,
                               cand_score, cand_count, nq, p.lists_per_q, k, 0, (int64_t)k, (int64_t)1, ix->d.doc_base, out_doc,
                               out_score, out_count, ors, ocs, (const int *)nullptr, p.n_whole);
        else
            hipLaunchKernelGGL(srx_merge_kernel, dim3((unsigned)nq_m), dim3(THREADS), 0, stream, cand_doc, cand_score, cand_count, nq,
                               p.lists_per_q, k, p.lists_per_q, 1, 1, 0, (int64_t)k, (int64_t)1, ix->d.doc_base, out_doc, out_score,
                               out_count, ors, ocs, (const int *)nullptr, p.n_whole);
        HIP_TRY(hipGetLastError());
    }
    if (prof) {
        HIP_TRY(hipEventRecord(ev[3], stream));
        ++ix->ev_calls;
        if (ix->ev_n < PROF_SLOTS) ++ix->ev_n;
    }
    return SRX_OK;
}
}  // namespace

SRX_API int srx_search(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                       int32_t k, int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                       int64_t workspace_bytes, void *stream_v) {
    return search_impl(ix, q_ptr, q_term, q_weight, nq, k, out_doc, out_score, out_count, (int64_t)k, (int64_t)1, workspace,
                       workspace_bytes, stream_v);
}

SRX_API int srx_search_packed(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight,
                              int32_t nq, int32_t k, int32_t *out_packed, void *workspace, int64_t workspace_bytes,
                              void *stream_v) {
    if (!out_packed || k <= 0) return fail(SRX_ERR_INVALID, "srx_search_packed: bad argument%s");
    const int64_t row = 2 * (int64_t)k + 1;  // [k doc ids][k score bit patterns][count]
    return search_impl(ix, q_ptr, q_term, q_weight, nq, k, out_packed, reinterpret_cast<float *>(out_packed + k),
                       out_packed + 2 * k, row, row, workspace, workspace_bytes, stream_v);
}

SRX_API int srx_search_after(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                             int32_t k, const int32_t *after_doc, const float *after_score, int32_t *out_doc, float *out_score,
                             int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (!after_doc || !after_score) return fail(SRX_ERR_INVALID, "srx_search_after: null bound arrays%s");
    return search_impl(ix, q_ptr, q_term, q_weight, nq, k, out_doc, out_score, out_count, (int64_t)k, (int64_t)1, workspace,
                       workspace_bytes, stream_v, after_doc, after_score);
}

SRX_API int srx_search_after_packed(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight,
                                    int32_t nq, int32_t k, const int32_t *after_doc, const float *after_score,
                                    int32_t *out_packed, void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (!out_packed || k <= 0 || !after_doc || !after_score) return fail(SRX_ERR_INVALID, "srx_search_after_packed: bad argument%s");
    const int64_t row = 2 * (int64_t)k + 1;
    return search_impl(ix, q_ptr, q_term, q_weight, nq, k, out_packed, reinterpret_cast<float *>(out_packed + k),
                       out_packed + 2 * k, row, row, workspace, workspace_bytes, stream_v, after_doc, after_score);
}

SRX_API int srx_profile_read(srx_index *ix, float *h_ms4) {
    if (!ix || !h_ms4) return fail(SRX_ERR_INVALID, "srx_profile_read: null argument%s");
    if (ix->ev_n == 0 || !ix->ev) return fail(SRX_ERR_INVALID, "srx_profile_read: no profiled srx_search has run%s");
    double acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < ix->ev_n; ++i) {
        const int slot = (int)((ix->ev_calls - 1 - i) % PROF_SLOTS);
        hipEvent_t *ev = ix->ev + PROF_EVENTS * slot;
        float a = 0, b = 0, c = 0, d = 0;
        HIP_TRY(hipEventSynchronize(ev[3]));
        HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
        HIP_TRY(hipEventElapsedTime(&c, ev[2], ev[3]));
        HIP_TRY(hipEventElapsedTime(&d, ev[0], ev[3]));
        acc[0] += a;
        acc[1] += b;
        acc[2] += c;
        acc[3] += d;
    }
    for (int j = 0; j < 4; ++j) h_ms4[j] = (float)(acc[j] / ix->ev_n);
    const int n = ix->ev_n;
    ix->ev_n = 0;
    return n;
}

SRX_API int64_t srx_merge_workspace_bytes(int32_t nq, int32_t n_lists, int32_t k) {
    if (nq < 0 || n_lists <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_merge_workspace_bytes: bad argument%s");
    const int fan = (MERGE_NPT * THREADS) / k;
    if (n_lists <= fan) return 0;
    // two ping-pong buffers sized for the first reduction level
    const int64_t g = (n_lists + fan - 1) / fan;
    return 2 * ((int64_t)nq * g * k * 8 + (int64_t)nq * g * 4 + 256);
}

int srx_merge_impl(int32_t device, const int32_t *in_doc, const float *in_score, const int32_t *in_count, int32_t nq,
                   int32_t n_lists, int32_t k, int lay, int64_t row_stride, int64_t cnt_stride, int32_t *out_doc,
                   float *out_score, int32_t *out_count, int64_t ors, int64_t ocs, void *workspace, int64_t workspace_bytes,
                   void *stream_v, const int *gate) {
    if (nq < 0 || n_lists <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_merge_topk: bad argument%s");
    if (nq == 0) return SRX_OK;
    if (!in_doc || !in_score || !in_count || !out_doc || !out_score || !out_count)
        return fail(SRX_ERR_INVALID, "srx_merge_topk: null pointer%s");
    const int64_t need = srx_merge_workspace_bytes(nq, n_lists, k);
    if (need > 0 && (!workspace || workspace_bytes < need)) return fail(SRX_ERR_NOMEM, "srx_merge_topk: workspace too small%s");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_v;
    const int fan = (MERGE_NPT * THREADS) / k;
    const int32_t *cur_doc = in_doc;
    const float *cur_score = in_score;
    const int32_t *cur_count = in_count;
    int lists = n_lists;
    int level = 0;
    const int64_t half = need / 2;
    while (lists > fan) {  // tree levels: groups of `fan` lists -> one unordered list each (plain layout)
        const int groups = (lists + fan - 1) / fan;
        char *buf = (char *)workspace + (level & 1) * half;
        int32_t *od = (int32_t *)buf;
        float *os = (float *)(od + (int64_t)nq * groups * k);
        int32_t *oc = (int32_t *)(os + (int64_t)nq * groups * k);
        hipLaunchKernelGGL(srx_merge_kernel, dim3((unsigned)((int64_t)nq * groups)), dim3(THREADS), 0, stream, cur_doc,
                           cur_score, cur_count, nq, lists, k, fan, groups, 0, lay, row_stride, cnt_stride, (int64_t)0, od, os,
                           oc, (int64_t)k, (int64_t)1, gate, 0);
        HIP_TRY(hipGetLastError());
        lay = 0;
        row_stride = k;
        cnt_stride = 1;
        cur_doc = od;
        cur_score = os;
        cur_count = oc;
        lists = groups;
        ++level;
    }
    if (k <= W_KMAX && (int64_t)lists * k <= MW_CAP && lists <= 256)
        hipLaunchKernelGGL(srx_merge_wave_kernel, dim3((unsigned)((nq + WAVES - 1) / WAVES)), dim3(THREADS), 0, stream, cur_doc,
                           cur_score, cur_count, nq, lists, k, lay, row_stride, cnt_stride, (int64_t)0, out_doc, out_score,
                           out_count, ors, ocs, gate, 0);
    else
        hipLaunchKernelGGL(srx_merge_kernel, dim3((unsigned)nq), dim3(THREADS), 0, stream, cur_doc, cur_score, cur_count, nq,
                           lists, k, lists, 1, 1, lay, row_stride, cnt_stride, (int64_t)0, out_doc, out_score, out_count, ors, ocs, gate, 0);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

SRX_API int srx_merge_topk(int32_t device, const int32_t *in_doc, const float *in_score, const int32_t *in_count,
                           int32_t nq, int32_t n_lists, int32_t k, int32_t gathered, int32_t *out_doc, float *out_score,
                           int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream_v) {
    return srx_merge_impl(device, in_doc, in_score, in_count, nq, n_lists, k, gathered ? 1 : 0, (int64_t)k, (int64_t)1, out_doc,
                      out_score, out_count, (int64_t)k, (int64_t)1, workspace, workspace_bytes, stream_v, nullptr);
}

SRX_API int srx_merge_topk_packed(int32_t device, const int32_t *packed, int32_t nq, int32_t n_lists, int32_t k,
                                  int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                                  int64_t workspace_bytes, void *stream_v) {
    if (!packed || k <= 0) return fail(SRX_ERR_INVALID, "srx_merge_topk_packed: bad argument%s");
    const int64_t row = 2 * (int64_t)k + 1;  // [k doc ids][k score bit patterns][count]
    return srx_merge_impl(device, packed, reinterpret_cast<const float *>(packed + k), packed + 2 * k, nq, n_lists, k, 1, row, row,
                      out_doc, out_score, out_count, (int64_t)k, (int64_t)1, workspace, workspace_bytes, stream_v, nullptr);
}

SRX_API int srx_merge_topk_packed_out(int32_t device, const int32_t *packed, int32_t nq, int32_t n_lists, int32_t k,
                                      int32_t *out_packed, void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (!packed || !out_packed || k <= 0) return fail(SRX_ERR_INVALID, "srx_merge_topk_packed_out: bad argument%s");
    const int64_t row = 2 * (int64_t)k + 1;
    return srx_merge_impl(device, packed, reinterpret_cast<const float *>(packed + k), packed + 2 * k, nq, n_lists, k, 1, row, row,
                      out_packed, reinterpret_cast<float *>(out_packed + k), out_packed + 2 * k, row, row, workspace,
                      workspace_bytes, stream_v, nullptr);
}

SRX_API int srx_build_impacts(int32_t device, const float *tf, const int32_t *post_doc, const float *doc_len,
                              int64_t nnz, double k1, double b, double avgdl, float *out_impact, void *stream_v) {
    if (nnz < 0 || (nnz > 0 && (!tf || !post_doc || !doc_len || !out_impact)))
        return fail(SRX_ERR_INVALID, "srx_build_impacts: bad argument%s");
    if (nnz == 0) return SRX_OK;
    HIP_TRY(hipSetDevice(device));
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(srx_impact_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_v, tf, post_doc, doc_len,
                       nnz, (float)k1, (float)b, (float)(1.0 - b), (float)(k1 + 1.0), (float)avgdl, out_impact);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------------
// Term bounds: out[t * nk + j] = the ks[j]-th largest stored value of term t (0 when it has fewer than ks[j] positive
// values).  One workgroup per term (grid-stride): the term's run of the term-major value array is folded into the
// exact lazy top-k list of the search kernels (k = the largest rank asked for, <= 1 024), then every kept value counts
// the kept values above it and equal to it and writes the ranks it covers.  One streaming pass over the values; the
// first form was a 64-bit sort of (term, value) keys of ALL postings (16+ bytes of temporary memory per posting, the
// build's peak).  A negative value raises *neg_flag (the bounds are only valid for non-negative values).
namespace {
constexpr int TB_NPT = 16;
template <typename VT>
__global__ __launch_bounds__(THREADS) void srx_term_bounds_kernel(const int64_t *__restrict__ term_ptr, const VT *__restrict__ val,
                                                                  int64_t vocab, const int32_t *__restrict__ ks, int nk, int kmax,
                                                                  float *__restrict__ out, int *__restrict__ neg_flag) {
    __shared__ MergeShared M;
    const int tid = threadIdx.x;
    for (int64_t t = blockIdx.x; t < vocab; t += gridDim.x) {
        const int64_t lo = term_ptr[t], hi = term_ptr[t + 1];
        if (tid < nk) out[t * nk + tid] = 0.0f;
        if (tid == 0) {
            M.tk.count = 0;
            M.tk.tau = 0;
        }
        __syncthreads();
        bool neg = false;
        for (int64_t c0 = lo; c0 < hi; c0 += (int64_t)THREADS * TB_NPT) {
            unsigned ubits[TB_NPT];
            int udoc[TB_NPT];
            const unsigned tau = M.tk.tau;
#pragma unroll
            for (int n = 0; n < TB_NPT; ++n) {
                const int64_t c = c0 + (int64_t)n * THREADS + tid;
                float x = 0.0f;
                if (c < hi) x = (float)val[c];
                neg |= x < 0.0f;
                const unsigned b = __float_as_uint(x);
                ubits[n] = (x > 0.0f && b >= tau) ? b : 0u;
                udoc[n] = (int)(c - lo);
            }
            topk_fold<TB_NPT, true>(ubits, udoc, kmax, M.tk, M.hist);
        }
        if (neg) *neg_flag = 1;
        __syncthreads();
        topk_shrink(kmax, M.tk, M.hist);
        __syncthreads();
        const unsigned cnt = M.tk.count;
        for (unsigned i = tid; i < cnt; i += THREADS) {
            const unsigned xi = M.tk.bits[i];
            int gt = 0, eq = 0;
            for (unsigned j = 0; j < cnt; ++j) {  // every thread reads the same word: an LDS broadcast
                const unsigned xj = M.tk.bits[j];
                gt += xj > xi;
                eq += xj == xi;
            }
            for (int j = 0; j < nk; ++j) {
                const int K = ks[j];
                if (gt < K && K <= gt + eq) out[t * nk + j] = __uint_as_float(xi);  // ties write the same value
            }
        }
        __syncthreads();  // the list is re-initialised by the next term
    }
}
}  // namespace

SRX_API int srx_build_term_bounds(int32_t device, int32_t val_type, const int64_t *term_ptr, const void *post_val, int64_t vocab,
                                  const int32_t *ks, int32_t nk, float *out_bound, int32_t *neg_flag, void *stream_v) {
    if (!term_ptr || !post_val || !ks || !out_bound || !neg_flag || vocab <= 0 || nk <= 0 || nk > 64)
        return fail(SRX_ERR_INVALID, "srx_build_term_bounds: bad argument (1 <= nk <= 64)%s");
    if (val_type != SRX_VAL_F32 && val_type != SRX_VAL_F16) return fail(SRX_ERR_INVALID, "srx_build_term_bounds: unknown val_type%s");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_v;
    int32_t hks[64];
    HIP_TRY(hipMemcpyAsync(hks, ks, sizeof(int32_t) * nk, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    int kmax = 0;
    for (int j = 0; j < nk; ++j) {
        if (hks[j] < 1 || hks[j] > KMAX) return fail(SRX_ERR_INVALID, "srx_build_term_bounds: ranks must be in 1 .. 1024%s");
        if (hks[j] > kmax) kmax = hks[j];
    }
    HIP_TRY(hipMemsetAsync(neg_flag, 0, sizeof(int32_t), stream));
    const unsigned blocks = (unsigned)(vocab < 256 * 16 ? vocab : 256 * 16);
    if (val_type == SRX_VAL_F32)
        hipLaunchKernelGGL(srx_term_bounds_kernel<float>, dim3(blocks), dim3(THREADS), 0, stream, term_ptr, (const float *)post_val, vocab,
                           ks, (int)nk, kmax, out_bound, (int *)neg_flag);
    else
        hipLaunchKernelGGL(srx_term_bounds_kernel<__half>, dim3(blocks), dim3(THREADS), 0, stream, term_ptr, (const __half *)post_val,
                           vocab, ks, (int)nk, kmax, out_bound, (int *)neg_flag);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

// Duplicate (doc, term) entries of a COO input, adjacent after the stable term sort: group g = entries
// [first[g], first[g + 1]) is summed left to right in input order, like SciPy sums duplicates when the reference assembles
// its CSR (csr_matrix((data, (rows, cols))), retrieval.py:171-175).  One thread per group (groups are almost all of length 1).
namespace {
__global__ __launch_bounds__(256) void srx_sum_groups_kernel(const int64_t *__restrict__ first, int64_t n_groups,
                                                             const float *__restrict__ val, float *__restrict__ out) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * 256) {
        const int64_t a = first[g], b = first[g + 1];
        float s = val[a];
        for (int64_t i = a + 1; i < b; ++i) s = s + val[i];
        out[g] = s;
    }
}
}  // namespace

SRX_API int srx_build_sum_duplicates(int32_t device, const int64_t *first, int64_t n_groups, const float *val, float *out_sum,
                                     void *stream_v) {
    if (n_groups < 0 || (n_groups > 0 && (!first || !val || !out_sum))) return fail(SRX_ERR_INVALID, "srx_build_sum_duplicates: bad argument%s");
    if (n_groups == 0) return SRX_OK;
    HIP_TRY(hipSetDevice(device));
    int64_t blocks = (n_groups + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(srx_sum_groups_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_v, first, n_groups, val, out_sum);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

SRX_API int srx_build_tile_skip(int32_t device, const int64_t *term_ptr, const int32_t *post_doc, int64_t vocab,
                                int32_t n_tiles, int32_t tile_log2, int32_t *out_skip, void *stream_v) {
    if (!term_ptr || !out_skip || vocab <= 0 || n_tiles <= 0 || tile_log2 < 0 || tile_log2 > 30)
        return fail(SRX_ERR_INVALID, "srx_build_tile_skip: bad argument%s");
    HIP_TRY(hipSetDevice(device));
    const int64_t total = vocab * (int64_t)(n_tiles + 1);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(srx_tile_skip_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_v, term_ptr, post_doc,
                       vocab, n_tiles, tile_log2, out_skip);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

// ================================================================================================
// Dense INT8 side of the same service: quantized_dot_product_batch
// (rag_system/core/retriever_registry.py:90-117; NumPy twin :538-548) + the same top-k (:505-519).

SRX_API int srx_memcpy_async(void *dst, const void *src, int64_t bytes, void *stream_v) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(SRX_ERR_INVALID, "srx_memcpy_async: bad argument%s");
    if (bytes == 0) return SRX_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, (hipStream_t)stream_v));
    return SRX_OK;
}

SRX_API int32_t srx_auto_unit_tiles(int64_t n_docs, int64_t vocab, int64_t nnz, int32_t tile_log2) {
    if (n_docs <= 0 || vocab <= 0 || nnz < 0 || tile_log2 < 6 || tile_log2 > SRX_MAX_TILE_LOG2)
        return fail(SRX_ERR_INVALID, "srx_auto_unit_tiles: bad argument%s");
    // The largest unit (in tiles) for which the run of an average term inside a unit overflows the registers of its lane
    // group (8 lanes x W_R postings in the reference case of an 8-term query) with negligible probability (mean +
    // 5 sigma, Poisson), and whose docs fit the tier-1 bitmap / the compact copy's local ids (49152).
    const int max_tpu_bitmap = W_UNIT_MAX_DOCS >> tile_log2;  // 16-bit unit-local ids below the sentinels' range (>= 3 tiles of 16384)
    const double per_doc_per_term = (double)nnz / ((double)n_docs * (double)vocab);
    auto fits = [&](int t) {
        const double mean = per_doc_per_term * (double)t * (double)(1ll << tile_log2);
        return mean + 5.0 * sqrt(mean) <= 8.0 * W_R;
    };
    int tpu = 1;
    while (tpu < MAX_TPS && tpu < max_tpu_bitmap && fits(tpu + 1)) ++tpu;
    return tpu;
}

SRX_API int srx_build_compact(int32_t device, int32_t val_type, const int32_t *post, int64_t n_blocks_total, int32_t tile_log2,
                              int32_t unit_tiles, int32_t *out_post16, void *stream_v) {
    if (!post || !out_post16 || n_blocks_total <= 0) return fail(SRX_ERR_INVALID, "srx_build_compact: bad argument%s");
    if (val_type != SRX_VAL_F32 && val_type != SRX_VAL_F16) return fail(SRX_ERR_INVALID, "srx_build_compact: bad val_type%s");
    if (tile_log2 < 6 || tile_log2 > SRX_MAX_TILE_LOG2 || unit_tiles < 1 || ((int64_t)unit_tiles << tile_log2) > W_UNIT_MAX_DOCS)
        return fail(SRX_ERR_INVALID, "srx_build_compact: a unit must cover at most 49152 docs (the tier-1 bitmap; the sentinels' local ids lie above)%s");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_v;
    const unsigned grid = (unsigned)((n_blocks_total + 255) / 256);
    const int unit_docs = unit_tiles << tile_log2;
    if (val_type == SRX_VAL_F32)
        hipLaunchKernelGGL(srx_compact_blocks_kernel<float>, dim3(grid), dim3(256), 0, stream, post, n_blocks_total, unit_docs, out_post16);
    else
        hipLaunchKernelGGL(srx_compact_blocks_kernel<__half>, dim3(grid), dim3(256), 0, stream, post, n_blocks_total, unit_docs, out_post16);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

SRX_API int srx_build_blocks(int32_t device, int32_t val_type, const int64_t *term_ptr, const int32_t *post_term,
                             const int32_t *post_doc, const void *post_val, const int32_t *skip, const int64_t *runpad,
                             int64_t vocab, int64_t nnz, int32_t n_tiles, int32_t tile_log2, int32_t unit_tiles,
                             int32_t *out_post, int32_t *out_skip, int64_t *out_term_ptr, int64_t n_blocks, void *stream_v) {
    if (!term_ptr || !skip || !runpad || !out_post || !out_skip || !out_term_ptr || vocab <= 0 || nnz < 0 || n_tiles <= 0 ||
        unit_tiles < 1 || n_blocks < 0 || (nnz > 0 && (!post_term || !post_doc || !post_val)))
        return fail(SRX_ERR_INVALID, "srx_build_blocks: bad argument%s");
    if (val_type != SRX_VAL_F32 && val_type != SRX_VAL_F16) return fail(SRX_ERR_INVALID, "srx_build_blocks: bad val_type%s");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_v;
    const int n_units = (n_tiles + unit_tiles - 1) / unit_tiles;
    if (nnz > 0) {
        int64_t blocks = (nnz + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        if (val_type == SRX_VAL_F32)
            hipLaunchKernelGGL(srx_blocks_scatter_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, term_ptr, post_term,
                               post_doc, (const float *)post_val, skip, runpad, nnz, n_tiles, tile_log2, unit_tiles, n_units, out_post);
        else
            hipLaunchKernelGGL(srx_blocks_scatter_kernel<__half>, dim3((unsigned)blocks), dim3(256), 0, stream, term_ptr, post_term,
                               post_doc, (const __half *)post_val, skip, runpad, nnz, n_tiles, tile_log2, unit_tiles, n_units, out_post);
        HIP_TRY(hipGetLastError());
    }
    {
        const int64_t total = vocab * (int64_t)(n_tiles + 1);
        int64_t blocks = (total + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(srx_blocks_skip_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, skip, runpad, vocab, n_tiles, unit_tiles,
                           n_units, out_skip, out_term_ptr);
        HIP_TRY(hipGetLastError());
    }
    // the SRX_BLOCK_PAD sentinel blocks behind the last run: lane j of a tier-1 wave redirects its idle loads to block n_blocks + j
    if (val_type == SRX_VAL_F32)
        hipLaunchKernelGGL(srx_blocks_sentinel_kernel<float>, dim3(SRX_BLOCK_PAD * 4 / 64), dim3(64), 0, stream, out_post, n_blocks, SRX_BLOCK_PAD);
    else
        hipLaunchKernelGGL(srx_blocks_sentinel_kernel<__half>, dim3(SRX_BLOCK_PAD * 4 / 64), dim3(64), 0, stream, out_post, n_blocks, SRX_BLOCK_PAD);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

#ifdef SRX_STAMP2
SRX_API int srx_debug_read_stamps2(unsigned long long *h_out16) {
    HIP_TRY(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_stamp2), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp2), z, sizeof(z)));
    return SRX_OK;
}
#endif
