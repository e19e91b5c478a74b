// sparse_rx.hip -- hand-written CDNA4 (gfx950) kernels + the C ABI of libsparse_rx.so.
//
// Hot path replaced (paths relative to /root/reference):
//   simd_bm25_score      rag_system/core/retrieval.py:41-76      (doc-major full CSR scan per query)
//   simd_tfidf_score     rag_system/pipeline/evaluate_rag_pipeline.py:95-121
//   fast_topk_selection  rag_system/core/retrieval.py:79-92      (+ score>0 filter :292-296)
//
// Design: the index is term-major (CSC) with a tile skip table.  A query's doc range is cut
// into *units* of a few tiles.  Two tiers score them, a merge kernel ranks:
//   tier 1  srx_wave_kernel   ONE WAVEFRONT per (query, split), no barriers.  Each query term owns a group of lanes;
//           a lane streams 4 consecutive postings per dwordx4 load, the next unit's loads are in flight while this
//           unit is scored from registers.  A wave-private LDS bitmap (1 bit per doc of the unit, ds_or_rtn) tells
//           which docs are matched by more than one term; every other posting is a single-term doc whose score is
//           its own contribution.  Postings of multi-term docs are parked in LDS and summed in ascending term id
//           (one wave's DS instructions execute in order), bit-identical to the reference's CSR row order.  A lazy
//           top-k list lives in LDS; an exact wave-level radix select shrinks it when it fills; the initial
//           threshold comes from per-term score bounds stored in the index.  Units that do not fit (long runs,
//           many duplicates), queries with > 64 terms and k > 128 are flagged for tier 2.
//   tier 2  srx_score_kernel  a persistent grid of 256-thread workgroups drains the worklist of flagged (query,
//           split) blocks: block-level LDS hash units of up to 4096 postings, a greedy tile packer, and dense fp32
//           accumulators acc[G] in LDS for tiles whose postings exceed that (barrier between terms keeps the
//           summation order).  Handles everything.
//   merge   srx_merge_kernel  exact top-k over the per-split / per-tier / per-shard lists + bitonic rank by
//           (score desc, doc asc).  Queries that tier 1 finished on its own are ranked there and skipped here.
// No MFMA (sparse gather/reduce, HBM-bound), no float atomics (LDS ds_add_f32 serialises at ~192 cycles per
// wave-instruction on gfx950, and sums must be deterministic).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (no fused multiply-add: the reference's
// arithmetic is separate fp32 multiply / add / IEEE divide).

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <new>

#include "sparse_rx.h"

#define SRX_API extern "C" __attribute__((visibility("default")))

#ifdef SRX_STAMP
__device__ unsigned long long g_stamp[16];
#define STAMP(i)                                                                         \
    do {                                                                                 \
        unsigned long long t_;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                               \
        st_acc[i] += t_ - st_prev;                                                       \
        st_prev = t_;                                                                    \
    } while (0)
#else
#define STAMP(i) \
    do {         \
    } while (0)
#endif

namespace {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int TBL_WORDS = 16384;            // 64 KiB LDS: hash table (keys+vals) or dense accumulators
constexpr int SLOTS = TBL_WORDS / 2;        // 8192 hash slots
constexpr int HASH_CAP = 4096;              // max postings accumulated by one hash unit (load <= 0.5)
constexpr int NPT_HASH = SLOTS / THREADS;   // 32 table slots per thread
constexpr int MAX_G = TBL_WORDS;            // dense tile <= 16384 docs
constexpr int NPT_DENSE = MAX_G / THREADS;  // 64
constexpr int KMAX = SRX_MAX_K;             // 1024
constexpr int KPT = KMAX / THREADS;         // 4 running-list entries per thread
constexpr int MAXT = 256;                   // query terms handled per pass (one per thread)
constexpr int MAX_STEPS = MAXT + HASH_CAP / THREADS + 16;  // step table entries of a hash unit
constexpr int PREFETCH = 8;                 // posting loads in flight per thread
constexpr int RADIX_BITS = 11;
constexpr int RADIX_BINS = 1 << RADIX_BITS;  // 2048-bin histogram (aliases the table region)
constexpr int MAX_TPS = 64;                 // tiles per supertile handled by the overflow packer
constexpr int MERGE_NPT = 16;               // merge kernel: candidates per thread (4096 per workgroup)
constexpr int EMPTY_KEY = -1;
// tier 1 (one wavefront per (query, split))
constexpr int W_UNIT_LOG2 = 16;             // a unit covers <= 65536 docs: 1 bit per doc in a wave-private LDS bitmap
constexpr int W_BM_WORDS = 1 << (W_UNIT_LOG2 - 5);  // 2048 words = 8 KiB
#ifndef SRX_W_R
#define SRX_W_R 16
#endif
constexpr int W_R = SRX_W_R;                // postings per lane per unit held in registers (8, 12 or 16)
constexpr int W_WAVES_PER_EU = W_R <= 8 ? 4 : 3;  // what the register budget of that choice allows
constexpr int W_CAP = W_R * 64;             // hence <= 1024 postings per tier-1 unit
constexpr int W_DUPCAP = 48;                // dup postings per unit resolved in tier 1 (more: the unit is dense -> tier 2)
constexpr int W_LCAP = 384;                 // lazy top-k list capacity (entries; a multiple of 64)
constexpr int W_KMAX = 128;                 // largest k served by tier 1
constexpr int W_MAXT = 64;                  // query terms (each owns 64 / 2^ceil(log2 nt) lanes)

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return fail(SRX_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// column of term_bound that is valid for top-k k: the smallest K in {1, 10, 100, 1000} with K >= k (-1: none)
__host__ __device__ inline int bound_column(int k) { return k <= 1 ? 0 : k <= 10 ? 1 : k <= 100 ? 2 : k <= 1000 ? 3 : -1; }

struct IndexView {
    const int64_t *term_ptr;
    const int32_t *post_doc;
    const void *post_val;
    const int32_t *tile_skip;
    const float *idf;
    const float *term_bound;  // optional [vocab*4]: K-th largest post_val per term for K = 1, 10, 100, 1000
    int64_t n_docs;
    int64_t vocab;
    int32_t tile_log2;
    int32_t n_tiles;
};

// ------------------------------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned wave_max(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_min(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}

// All three return the block-wide value to every thread.  `red` = 3*WAVES words of LDS.  Ends with a
// barrier, so `red` may be reused immediately.
__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned *red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}
struct SumMaxMin {
    unsigned sum, mx, mn;
};
__device__ __forceinline__ SumMaxMin block_sum_max_min(unsigned s, unsigned mx, unsigned mn, unsigned *red) {
    s = wave_sum(s);
    mx = wave_max(mx);
    mn = wave_min(mn);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s;
        red[WAVES + (threadIdx.x >> 6)] = mx;
        red[2 * WAVES + (threadIdx.x >> 6)] = mn;
    }
    __syncthreads();
    SumMaxMin r;
    r.sum = red[0] + red[1] + red[2] + red[3];
    r.mx = max(max(red[WAVES + 0], red[WAVES + 1]), max(red[WAVES + 2], red[WAVES + 3]));
    r.mn = min(min(red[2 * WAVES + 0], red[2 * WAVES + 1]), min(red[2 * WAVES + 2], red[2 * WAVES + 3]));
    __syncthreads();
    return r;
}

// Exclusive prefix sum over the block (thread order); total returned through *total.
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *red, unsigned *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned w = __shfl_up(inc, o);
        if (lane >= o) inc += w;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        unsigned x = red[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// Exact k-th largest of the block's keys (radix select, MSD, 11-bit digits, LDS histogram).
// key == 0 means "not a candidate"; all candidate keys are in [1, 2^31).  Requires
// 1 <= k <= #candidates.  mx / mn = max / min over candidate keys.  Returns T = the k-th largest
// key; n_gt = #keys > T (< k), n_eq = #keys == T (>= k - n_gt).
// hist: RADIX_BINS words, red: >= 16 words of LDS.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ unsigned radix_kth(const unsigned (&key)[N], unsigned k, unsigned mx, unsigned mn, unsigned n_cand,
                              unsigned *hist, unsigned *red, unsigned *n_gt, unsigned *n_eq) {
    if (mx == mn) {
        *n_gt = 0;
        *n_eq = n_cand;
        return mx;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = 31 - __clz(mx ^ mn);  // highest bit in which candidates differ (<= 30)
    unsigned prefix = mx & ~((2u << hb) - 1u);
    int shift = hb + 1;
    unsigned krem = k, gt = 0, eq = 0;
    while (shift > 0) {
        const int w = shift < RADIX_BITS ? shift : RADIX_BITS;
        shift -= w;
        const int hi_shift = shift + w;  // <= 31
        for (int i = tid; i < RADIX_BINS; i += THREADS) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const unsigned x = key[n];
            if (x != 0 && ((x ^ prefix) >> hi_shift) == 0) atomicAdd(&hist[(x >> shift) & ((1u << w) - 1u)], 1u);
        }
        __syncthreads();
        // thread t owns bins [8t, 8t+8); find the bin holding the krem-th largest
        const uint4 a = reinterpret_cast<const uint4 *>(hist)[2 * tid];
        const uint4 b = reinterpret_cast<const uint4 *>(hist)[2 * tid + 1];
        const unsigned h[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const unsigned s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
        unsigned suf = s;  // inclusive suffix sum over threads >= tid
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned v = __shfl_down(suf, o);
            if (lane + o < 64) suf += v;
        }
        if (lane == 0) red[wave] = suf;
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww)
            if (ww > wave) suf += red[ww];
        const unsigned above = suf - s;
        if (above < krem && krem <= suf) {
            unsigned run = above;
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                if (run + h[i] >= krem) {
                    red[8] = (unsigned)(8 * tid + i);
                    red[9] = run;
                    red[10] = h[i];
                    break;
                }
                run += h[i];
            }
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

// ------------------------------------------------------------------------------------------------
// Running top-k list of a workgroup, kept in LDS (unordered).  `tau` = key of the k-th best once a
// selection has run (0 before): a later candidate with key < tau cannot enter.
// Total order: larger score first, then smaller doc ("key2" = 0x7FFFFFFF - doc, larger first).
// ------------------------------------------------------------------------------------------------
struct TopkShared {
    unsigned bits[KMAX];
    int doc[KMAX];
    unsigned count;
    unsigned tau;
    unsigned red[16];
};

// Fold the candidates of one unit (register arrays ubits/udoc, ubits == 0 -> none) into the list.
// Candidates must already satisfy ubits >= tau.  hist = RADIX_BINS words of free LDS.
// LAZY: candidates are appended while the list has room (capacity KMAX) and the selection only runs when it
// would overflow; the caller finishes with topk_shrink().  !LAZY: the list never exceeds k.
template <int N, bool LAZY>
__device__ void topk_fold(unsigned (&ubits)[N], const int (&udoc)[N], int k, TopkShared &tk, unsigned *hist) {
    const int tid = threadIdx.x;
    const unsigned n_old = tk.count;  // read BEFORE the barriers below: later appends must not be seen by slow threads
    unsigned mine = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) mine += (ubits[n] != 0);
    const unsigned n_new = block_sum(mine, tk.red);
    if (n_new == 0 && n_old <= (unsigned)k) return;
    if (n_old + n_new <= (unsigned)(LAZY ? KMAX : k)) {
#pragma unroll
        for (int n = 0; n < N; ++n)
            if (ubits[n] != 0) {
                const unsigned p = atomicAdd(&tk.count, 1u);
                tk.bits[p] = ubits[n];
                tk.doc[p] = udoc[n];
            }
        __syncthreads();
        return;
    }
    // selection over (list U candidates)
    unsigned key[N + KPT];
    int doc[N + KPT];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        key[n] = ubits[n];
        doc[n] = udoc[n];
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const unsigned i = tid + j * THREADS;
        const bool ok = i < n_old;
        key[N + j] = ok ? tk.bits[i] : 0u;
        doc[N + j] = ok ? tk.doc[i] : 0;
    }
    unsigned lmx = 0, lmn = 0xFFFFFFFFu;
#pragma unroll
    for (int n = 0; n < N + KPT; ++n)
        if (key[n] != 0) {
            lmx = max(lmx, key[n]);
            lmn = min(lmn, key[n]);
        }
    const SumMaxMin r = block_sum_max_min(0u, lmx, lmn, tk.red);  // also orders the list reads above
    unsigned n_gt, n_eq;
    const unsigned T = radix_kth<N + KPT>(key, (unsigned)k, r.mx, r.mn, n_old + n_new, hist, tk.red, &n_gt, &n_eq);
    const unsigned need = (unsigned)k - n_gt;  // ties to accept, 1 <= need <= n_eq
    unsigned T2 = 0;                            // accept ties with key2 >= T2
    if (n_eq > need) {
        unsigned key2[N + KPT];
        unsigned mx2 = 0, mn2 = 0xFFFFFFFFu;
#pragma unroll
        for (int n = 0; n < N + KPT; ++n) {
            key2[n] = (key[n] == T) ? (0x7FFFFFFFu - (unsigned)doc[n]) : 0u;
            if (key2[n] != 0) {
                mx2 = max(mx2, key2[n]);
                mn2 = min(mn2, key2[n]);
            }
        }
        const SumMaxMin r2 = block_sum_max_min(0u, mx2, mn2, tk.red);
        unsigned g2, e2;
        T2 = radix_kth<N + KPT>(key2, need, r2.mx, r2.mn, n_eq, hist, tk.red, &g2, &e2);
    }
    if (tid == 0) {
        tk.count = 0;
        tk.tau = T;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N + KPT; ++n) {
        const unsigned x = key[n];
        const bool take = (x > T) || (x == T && (0x7FFFFFFFu - (unsigned)doc[n]) >= T2);
        if (x != 0 && take) {
            const unsigned p = atomicAdd(&tk.count, 1u);
            tk.bits[p] = x;
            tk.doc[p] = doc[n];
        }
    }
    __syncthreads();
}

// Shrink a lazily grown list to its top k (no-op when it already fits).
__device__ void topk_shrink(int k, TopkShared &tk, unsigned *hist) {
    if (tk.count <= (unsigned)k) return;  // uniform: count was last written before a barrier
    unsigned none_b[1] = {0u};
    const int none_d[1] = {0};
    __syncthreads();
    topk_fold<1, false>(none_b, none_d, k, tk, hist);
}

// 4 consecutive postings with one instruction (global_load_dwordx4 / dwordx2); only 4-byte alignment is guaranteed
struct __attribute__((packed, aligned(4))) PackI4 {
    int x, y, z, w;
};
struct __attribute__((packed, aligned(4))) PackF4 {
    float x, y, z, w;
};
struct __attribute__((packed, aligned(2))) PackH4 {
    __half x, y, z, w;
};
__device__ __forceinline__ void load4(const int32_t *p, int elem_off, int &a, int &b, int &c, int &d) {
    const PackI4 t = *reinterpret_cast<const PackI4 *>(p + elem_off);
    a = t.x; b = t.y; c = t.z; d = t.w;
}
__device__ __forceinline__ void load4(const float *p, int elem_off, float &a, float &b, float &c, float &d) {
    const PackF4 t = *reinterpret_cast<const PackF4 *>(p + elem_off);
    a = t.x; b = t.y; c = t.z; d = t.w;
}
__device__ __forceinline__ void load4(const __half *p, int elem_off, float &a, float &b, float &c, float &d) {
    const PackH4 t = *reinterpret_cast<const PackH4 *>(p + elem_off);
    a = __half2float(t.x); b = __half2float(t.y); c = __half2float(t.z); d = __half2float(t.w);
}

__device__ __forceinline__ float load_val(const float *p, int64_t i) { return p[i]; }
__device__ __forceinline__ float load_val(const __half *p, int64_t i) { return __half2float(p[i]); }

// ------------------------------------------------------------------------------------------------
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
};

// Hash-accumulate the unit described by m_start/m_len (P <= HASH_CAP postings) and fold its positive
// scores into the running top-k.  nt = terms in this pass.
template <typename VT>
__device__ void hash_unit(ScoreShared &S, const IndexView &ix, int nt, int my_len, int k, int dbg = 0) {
    const int tid = threadIdx.x;
    int *keys = reinterpret_cast<int *>(S.tbl);
    float *vals = reinterpret_cast<float *>(S.tbl + SLOTS);
    const int32_t *post_doc = ix.post_doc;
    const VT *post_val = reinterpret_cast<const VT *>(ix.post_val);

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
                    d[r] = post_doc[g];
                    v[r] = load_val(post_val, g);
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
            const bool ok = ks[c] != EMPTY_KEY && vs[c] > 0.0f && b >= tau;
            ubits[j * 4 + c] = ok ? b : 0u;
            udoc[j * 4 + c] = ks[c];
        }
    }
    __syncthreads();  // table is free from here: vals region doubles as the radix histogram
    if (!(dbg & 1)) topk_fold<NPT_HASH, true>(ubits, udoc, k, S.tk, S.tbl + SLOTS);
}

// Dense-accumulate one tile of G docs [tile_base, tile_base + G) described by m_start/m_len.
// first_pass: zero the accumulators; last_pass: select.  (Queries with > MAXT terms take several passes.)
template <typename VT>
__device__ void dense_tile_accumulate(ScoreShared &S, const IndexView &ix, int nt, int tile_base, bool first_pass) {
    const int tid = threadIdx.x;
    float *acc = reinterpret_cast<float *>(S.tbl);
    const int G = 1 << ix.tile_log2;
    const int32_t *post_doc = ix.post_doc;
    const VT *post_val = reinterpret_cast<const VT *>(ix.post_val);
    if (first_pass) {
        for (int i = tid; i < G / 4; i += THREADS) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    // Batches of 2048 postings (two dwordx4 stripes = 8 per thread) are enumerated term-major; the loads of batch n+1
    // are issued before batch n is accumulated, across term boundaries too, so 2 x 16 KB per workgroup stay in flight
    // and a term's load latency hides behind the previous term's work.  A barrier separates consecutive batches of
    // different terms (the next term may touch the same doc).
    constexpr int NB = 8;                 // postings per thread per batch
    constexpr int BATCH = THREADS * NB;   // 2048
    auto next_term = [&](int i) {  // first term index >= i with postings in this tile (uniform), nt if none
        while (i < nt && S.m_len[i] == 0) ++i;
        return i;
    };
    auto load_batch = [&](int i, int o, int (&d)[NB], float (&v)[NB]) {
        const int len = S.m_len[i];
        const int64_t start = S.m_start[i];
#pragma unroll
        for (int h = 0; h < NB / 4; ++h) {
            const int p = o + h * (THREADS * 4) + tid * 4;  // my 4 consecutive postings of this stripe
            const int64_t g = (p < len) ? start + p : start;  // idle threads re-read the run's head (always valid)
            load4(post_doc + g, 0, d[4 * h], d[4 * h + 1], d[4 * h + 2], d[4 * h + 3]);
            load4(post_val + g, 0, v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (p + c >= len) d[4 * h + c] = -1;
        }
    };
    auto add_batch = [&](int i, const int (&d)[NB], const float (&v)[NB]) {
        const float idf = S.m_idf[i], qw = S.m_qw[i];
#pragma unroll
        for (int r = 0; r < NB; ++r)
            if (d[r] >= 0) {
                const int o = d[r] - tile_base;
                acc[o] = acc[o] + (v[r] * idf) * qw;  // docs unique within a term: no conflict
            }
    };
    int ci = next_term(0), co = 0;
    if (ci >= nt) return;
    int dA[NB], dB[NB];
    float vA[NB], vB[NB];
    load_batch(ci, co, dA, vA);
    for (;;) {
        // successor batch
        int ni = ci, no = co + BATCH;
        if (no >= S.m_len[ci]) {
            ni = next_term(ci + 1);
            no = 0;
        }
        const bool more = ni < nt;
        if (more) load_batch(ni, no, dB, vB);
        add_batch(ci, dA, vA);
        if (!more) break;
        if (ni != ci) __syncthreads();
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            dA[r] = dB[r];
            vA[r] = vB[r];
        }
        ci = ni;
        co = no;
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

template <typename VT>
__device__ bool flat_tile(ScoreShared &S, const IndexView &ix, int nt, int my_len, int tile_base, int k) {
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
    const int32_t *post_doc = ix.post_doc;
    const VT *post_val = reinterpret_cast<const VT *>(ix.post_val);

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
            const int d = post_doc[S.m_start[i] + (f - pre[i])] - tile_base;
            const unsigned bit = 1u << (d & 31);
            if (atomicOr(&bm1[d >> 5], bit) & bit) atomicOr(&bm2[d >> 5], bit);
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
                const int d = post_doc[g] - tile_base;
                const float c = (load_val(post_val, g) * S.m_idf[i]) * S.m_qw[i];
                if ((bm2[d >> 5] >> (d & 31)) & 1u) {
                    const unsigned e = atomicAdd(mcount, 1u);
                    if (e < (unsigned)FLAT_MCAP) {
                        mk_key[e] = (d << 8) | i;
                        mk_c[e] = c;
                    }
                } else {
                    const float sc = 0.0f + c;
                    const unsigned b = __float_as_uint(sc);
                    if (sc > 0.0f && b >= tau) {
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
            if (sum > 0.0f && bb >= tau) {
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

__device__ void dense_tile_select(ScoreShared &S, const IndexView &ix, int tile_base, int k) {
    const int tid = threadIdx.x;
    const float *acc = reinterpret_cast<const float *>(S.tbl);
    const int G = 1 << ix.tile_log2;
    unsigned ubits[NPT_DENSE];
    int udoc[NPT_DENSE];
    const unsigned tau = S.tk.tau;
#pragma unroll
    for (int n = 0; n < NPT_DENSE; ++n) {
        const int o = n * THREADS + tid;
        float x = 0.f;
        if (o < G) x = acc[o];
        const unsigned b = __float_as_uint(x);
        const bool ok = x > 0.0f && b >= tau && (int64_t)tile_base + o < ix.n_docs;
        ubits[n] = ok ? b : 0u;
        udoc[n] = tile_base + o;
    }
    __syncthreads();  // accumulators are in registers; their LDS doubles as the radix histogram
    topk_fold<NPT_DENSE, true>(ubits, udoc, k, S.tk, S.tbl);
}

// Work item -> (query, split, splits of that query).  The first n_whole queries are one item each; the others are cut
// into n_splits doc-range splits (n_whole = 0: every query is split the same way).  With more queries than resident
// waves, the last partial round of a batch is cut finer, so that the kernel's tail is made of short items.
__device__ __forceinline__ void decode_item(int item, int n_whole, int n_splits, int &q, int &split, int &nsq) {
    if (item < n_whole) {
        q = item;
        split = 0;
        nsq = 1;
    } else {
        const int j = item - n_whole;
        q = n_whole + j / n_splits;
        split = j - (j / n_splits) * n_splits;
        nsq = n_splits;
    }
}

template <typename VT>
__device__ void score_block(ScoreShared &S, int bid, const IndexView &ix, const int32_t *__restrict__ q_ptr,
                            const int32_t *__restrict__ q_term, const float *__restrict__ q_weight, int nq, int k,
                            int n_splits, int n_whole, int tpu, int n_super, int dbg, const unsigned *__restrict__ ovf,
                            int ovf_words, int lists_per_q, int32_t *__restrict__ cand_doc,
                            float *__restrict__ cand_score, int32_t *__restrict__ cand_count) {
    const int tid = threadIdx.x;
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
    const bool all_units = (nt_all > W_MAXT) || (k > W_KMAX) || ((tpu << ix.tile_log2) > (1 << W_UNIT_LOG2)) || (dbg & 8);
    const unsigned *my_ovf = ovf + (int64_t)q * ovf_words;
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
    {
        const int col = bound_column(k);
        unsigned t0b = 0, negf = 0;
        for (int i = tid; i < nt_all; i += THREADS) {
            const int term = q_term[t0 + i];
            const float idf = ix.idf[term], qw = q_weight[t0 + i];
            if (idf < 0.0f || qw < 0.0f) {
                negf = 1;
            } else if (ix.term_bound != nullptr && col >= 0 && idf > 0.0f && qw > 0.0f) {
                const float b = 0.0f + (ix.term_bound[(int64_t)term * 4 + col] * idf) * qw;
                t0b = max(t0b, __float_as_uint(b > 0.0f ? b : 0.0f));
            }
        }
        const SumMaxMin r = block_sum_max_min(negf, t0b, 0u, S.tk.red);
        tau0 = r.sum ? 0u : r.mx;
    }
    if (tid == 0) {
        S.tk.count = 0;
        S.tk.tau = tau0;
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
        for (int su = su_lo; su < su_hi; ++su) {
            if (!all_units && !((my_ovf[su >> 5] >> (su & 31)) & 1u)) continue;  // uniform
            int lo = 0, hi = 0;
            if (tid < nt) {
                lo = skip_row[min(su * tps, ix.n_tiles)];
                hi = skip_row[min((su + 1) * tps, ix.n_tiles)];
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
                served = flat_tile<VT>(S, ix, nt, my_len, su << ix.tile_log2, k);
                for (int i = tid; i < SLOTS; i += THREADS) keys[i] = EMPTY_KEY;  // back to hash mode
                __syncthreads();
            }
            if (served) {
            } else if (P > 0 && P <= (unsigned)HASH_CAP) {
                if (tid < nt) {
                    S.m_start[tid] = base + lo;
                    S.m_len[tid] = my_len;
                }
                __syncthreads();
                hash_unit<VT>(S, ix, nt, my_len, k, dbg);
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
                        if (acc_p > 0 && acc_p + pj > HASH_CAP) {
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
                    __syncthreads();
                    if (GP <= (unsigned)HASH_CAP) {
                        hash_unit<VT>(S, ix, nt, glen, k, dbg);
                    } else {  // one dense tile (gb == ga + 1 by construction)
                        const int tile_base = ga << ix.tile_log2;
                        dense_tile_accumulate<VT>(S, ix, nt, tile_base, true);
                        dense_tile_select(S, ix, tile_base, k);
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
                dense_tile_accumulate<VT>(S, ix, nt, tile_base, pass == 0);
            }
            dense_tile_select(S, ix, tile_base, k);
        }
    }

    // ---- emit this split's list (unordered; the merge kernel ranks) ----
    __syncthreads();
    topk_shrink(k, S.tk, S.tbl);
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
template <typename VT>
__global__ __launch_bounds__(THREADS, 2) void srx_score_kernel(IndexView ix, const int32_t *__restrict__ q_ptr,
                                                               const int32_t *__restrict__ q_term,
                                                               const float *__restrict__ q_weight, int nq, int k,
                                                               int n_splits, int n_whole, int tpu, int n_super, int dbg,
                                                               const unsigned *__restrict__ ovf, int ovf_words,
                                                               int lists_per_q, const int *__restrict__ work,
                                                               int32_t *__restrict__ cand_doc,
                                                               float *__restrict__ cand_score,
                                                               int32_t *__restrict__ cand_count) {
    __shared__ ScoreShared S;
    const int n_work = work[0];
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        __syncthreads();  // the previous block's LDS state is dead
        score_block<VT>(S, work[1 + w], ix, q_ptr, q_term, q_weight, nq, k, n_splits, n_whole, tpu, n_super, dbg, ovf,
                        ovf_words, lists_per_q, cand_doc, cand_score, cand_count);
    }
}

// ================================================================================================
// Tier 1: one wavefront per (query, split).  Wave-synchronous: no s_barrier anywhere; LDS executes one
// wave's DS instructions in order, wsync() only stops the compiler from reordering across the hand-off.
// ================================================================================================
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uniu(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

struct WaveShared {
    unsigned lbits[W_LCAP];        // lazy top-k list (score bits, doc), unordered
    int ldoc[W_LCAP];
    unsigned hist[256];            // radix histogram of the list selection
};

// Exact k-th largest of the keys keyfn(i), i < count (key 0 = none; keys in [1, 2^31)); 8-bit MSD radix with a
// 256-bin LDS histogram, 4 bins per lane.  The keys are re-read from LDS in every pass (a loop, not registers): the
// selection is rare, and a small register footprint here is what keeps the calling kernel's VGPR count low (the
// caller's live values must sit above the callee's registers).  Requires 1 <= k <= #candidates.
template <typename KeyFn>
__device__ __forceinline__ unsigned wave_radix_kth(KeyFn keyfn, unsigned count, unsigned k, unsigned mx, unsigned mn,
                                                   unsigned n_cand, unsigned *hist, unsigned *n_gt, unsigned *n_eq) {
    if (mx == mn) {
        *n_gt = 0;
        *n_eq = n_cand;
        return mx;
    }
    const int lane = threadIdx.x & 63;  // also used by multi-wave blocks (srx_merge_wave_kernel)
    const int hb = 31 - __clz(mx ^ mn);
    unsigned prefix = mx & ~((2u << hb) - 1u);
    int shift = hb + 1;
    unsigned krem = k, gt = 0, eq = 0;
    while (shift > 0) {
        const int w = shift < 8 ? shift : 8;
        shift -= w;
        const int hi_shift = shift + w;
        reinterpret_cast<uint4 *>(hist)[lane] = make_uint4(0u, 0u, 0u, 0u);
        wsync();
        for (unsigned i = lane; i < count; i += 64) {
            const unsigned x = keyfn(i);
            if (x != 0 && ((x ^ prefix) >> hi_shift) == 0) atomicAdd(&hist[(x >> shift) & ((1u << w) - 1u)], 1u);
        }
        wsync();
        const uint4 a = reinterpret_cast<const uint4 *>(hist)[lane];
        const unsigned h[4] = {a.x, a.y, a.z, a.w};
        const unsigned s = (a.x + a.y) + (a.z + a.w);
        unsigned suf = s;  // inclusive suffix sum over lanes >= lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_down(suf, o);
            if (lane + o < 64) suf += v;
        }
        const unsigned above = suf - s;
        const bool own = above < krem && krem <= suf;
        unsigned d = 0, ab = 0, cn = 0;
        if (own) {
            unsigned run = above;
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                if (run + h[i] >= krem) {
                    d = (unsigned)(4 * lane + i);
                    ab = run;
                    cn = h[i];
                    break;
                }
                run += h[i];
            }
        }
        const int owner = __ffsll((unsigned long long)__ballot(own)) - 1;
        d = (unsigned)__shfl((int)d, owner);
        ab = (unsigned)__shfl((int)ab, owner);
        eq = (unsigned)__shfl((int)cn, owner);
        krem -= ab;
        gt += ab;
        prefix |= d << shift;
        wsync();
    }
    *n_gt = gt;
    *n_eq = eq;
    return prefix;
}

// Shrink the wave's list (count > k entries in LDS) to its exact top k; returns tau = key of the k-th.
// Ties at the k-th score keep the smallest doc ids (the order contract).  Works in place on the LDS list.
template <typename SH>
__device__ __noinline__ unsigned wave_list_select(SH &S, unsigned count, int k) {
    const int lane = threadIdx.x & 63;  // also used by multi-wave blocks (srx_merge_wave_kernel)
    unsigned mx = 0, mn = 0xFFFFFFFFu;
    for (unsigned i = lane; i < count; i += 64) {
        const unsigned x = S.lbits[i];
        mx = max(mx, x);
        mn = min(mn, x);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    unsigned n_gt, n_eq;
    const unsigned T = wave_radix_kth([&](unsigned i) -> unsigned { return S.lbits[i]; }, count, (unsigned)k, mx, mn, count,
                                      S.hist, &n_gt, &n_eq);
    const unsigned need = (unsigned)k - n_gt;
    unsigned T2 = 0;
    if (n_eq > need) {  // uniform: more entries tie at T than fit -> the `need` smallest doc ids among them
        auto key2 = [&](unsigned i) -> unsigned { return S.lbits[i] == T ? 0x7FFFFFFFu - (unsigned)S.ldoc[i] : 0u; };
        unsigned mx2 = 0, mn2 = 0xFFFFFFFFu;
        for (unsigned i = lane; i < count; i += 64) {
            const unsigned x = key2(i);
            if (x != 0) {
                mx2 = max(mx2, x);
                mn2 = min(mn2, x);
            }
        }
        mx2 = wave_max(mx2);
        mn2 = wave_min(mn2);
        unsigned g2, e2;
        T2 = wave_radix_kth(key2, count, need, mx2, mn2, n_eq, S.hist, &g2, &e2);
    }
    wsync();
    // Deterministic in-place compaction: 64 entries per step are read before any is written, and an entry only moves
    // down (its new position <= the number of entries read so far).
    unsigned base = 0;  // wave-uniform running count
    for (unsigned i0 = 0; i0 < count; i0 += 64) {
        const unsigned i = i0 + lane;
        const unsigned x = i < count ? S.lbits[i] : 0u;
        const int dd = i < count ? S.ldoc[i] : 0;
        const bool take = x != 0 && ((x > T) || (x == T && (0x7FFFFFFFu - (unsigned)dd) >= T2));
        const unsigned long long m = __ballot(take);
        wsync();
        if (take) {
            const unsigned p = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            S.lbits[p] = x;
            S.ldoc[p] = dd;
        }
        base += (unsigned)__popcll(m);
        wsync();
    }
    return T;
}

struct WaveTopk {
    unsigned count, tau;
};

// Append candidates (one per lane at most) to the wave's lazy list, shrinking it first when it is nearly full.
__device__ __forceinline__ void wave_append(WaveShared &S, WaveTopk &tk, int k, bool cand, unsigned bits, int doc) {
    const int lane = threadIdx.x;
    const unsigned long long m = __ballot(cand);
    if (m != 0ull) {  // uniform
        if (tk.count > (unsigned)(W_LCAP - 64)) {  // make room for up to 64 more entries
            tk.tau = uniu(wave_list_select(S, tk.count, k));
            tk.count = (unsigned)k;
        }
        const bool c2 = cand && bits >= tk.tau;  // tau may just have risen
        const unsigned long long m2 = __ballot(c2);
        if (c2) {
            const unsigned p = tk.count + (unsigned)__popcll(m2 & ((1ull << lane) - 1ull));
            S.lbits[p] = bits;
            S.ldoc[p] = doc;
        }
        tk.count += (unsigned)__popcll(m2);
    }
}

// Rank a wave's final list (count <= k <= 128 entries in S.lbits / S.ldoc) and write the padded result row:
// wave-level bitonic sort of 128 keys (score bits : ~doc, descending) in the LDS scratch K, two keys per lane,
// no barrier.
template <typename SH>
__device__ __forceinline__ void wave_rank_emit(SH &S, unsigned long long *K, unsigned count, int k, int64_t doc_base,
                                               int32_t *__restrict__ row_doc, float *__restrict__ row_score) {
    const int lane = threadIdx.x & 63;
    wsync();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned i = lane + 64 * j;
        K[i] = i < count ? (((unsigned long long)S.lbits[i] << 32) | (0x7FFFFFFFu - (unsigned)S.ldoc[i])) : 0ull;
    }
    wsync();
    for (unsigned size = 2; size <= 128; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned pos = 2 * lane - (lane & (stride - 1));
            const unsigned long long a = K[pos], b = K[pos + stride];
            const bool desc = (pos & size) == 0;
            if (desc ? (a < b) : (a > b)) {
                K[pos] = b;
                K[pos + stride] = a;
            }
            wsync();
        }
    }
    for (unsigned i = lane; i < (unsigned)k; i += 64) {
        if (i < count) {
            const unsigned long long x = K[i];
            row_doc[i] = (int32_t)(doc_base + (int64_t)(0x7FFFFFFFu - (unsigned)(x & 0xFFFFFFFFull)));
            row_score[i] = __uint_as_float((unsigned)(x >> 32));
        } else {
            row_doc[i] = -1;
            row_score[i] = 0.0f;
        }
    }
}

template <int L>
struct IntC {
    static constexpr int value = L;
};

template <typename VT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W_WAVES_PER_EU))) void srx_wave_kernel(IndexView ix, const int32_t *__restrict__ q_ptr,
                                                      const int32_t *__restrict__ q_term,
                                                      const float *__restrict__ q_weight, int nq, int k, int n_splits,
                                                      int n_whole, int tpu, int n_super, int dbg,
                                                      unsigned *__restrict__ ovf, int ovf_words, int lists_per_q,
                                                      int *__restrict__ work, int32_t *__restrict__ cand_doc,
                                                      float *__restrict__ cand_score,
                                                      int32_t *__restrict__ cand_count, int64_t doc_base,
                                                      int32_t *__restrict__ out_doc, float *__restrict__ out_score,
                                                      int32_t *__restrict__ out_count, int64_t out_row_stride,
                                                      int64_t out_cnt_stride) {
    __shared__ WaveShared S;
    extern __shared__ __attribute__((aligned(16))) unsigned bm[];  // doc bitmap of the current unit: bm_words words (>= 256)
    const int lane = threadIdx.x;
    int q, split, nsq;
    decode_item((int)blockIdx.x, n_whole, n_splits, q, split, nsq);
    if (q >= nq) return;
    const int64_t list = (int64_t)q * lists_per_q + split;
    const int t0 = q_ptr[q];
    const int nt = q_ptr[q + 1] - t0;
    if (nt == 0 || nt > W_MAXT || k > W_KMAX || (tpu << ix.tile_log2) > (1 << W_UNIT_LOG2) || (dbg & 8)) {  // tier 2 serves it
        if (lane == 0) {
            cand_count[list] = 0;
            if (nt > 0) work[1 + atomicAdd(&work[0], 1)] = (int)blockIdx.x;
        }
        return;
    }
    const int su_lo = (int)(((int64_t)n_super * split) / nsq);
    const int su_hi = (int)(((int64_t)n_super * (split + 1)) / nsq);
    const int row = ix.n_tiles + 1;

    const int bm_words = max(256, ((tpu << ix.tile_log2) + 31) >> 5);  // the launch's dynamic LDS holds bm_words + 64 words
    for (int i = lane; i < (bm_words + 64) / 4; i += 64) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    wsync();
    WaveTopk tk = {0u, 0u};  // wave-uniform lazy top-k list state
    int sink = 0;            // debug only
#ifdef SRX_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
    bool flagged = false;    // wave-uniform: some unit of this block was handed to tier 2
    int lg = 0;
    while ((1 << lg) < nt) ++lg;

    // Query term t owns a group of LPT = 64 / 2^ceil(log2 nt) lanes; lane jl of the group handles postings
    // jl, jl + LPT, jl + 2 LPT, ... of the term's run inside the unit.  Term data stays in registers.  LPT is a
    // compile-time constant of the body (7 instantiations): loads use immediate offsets, no per-step address math.
    auto run = [&](auto lconst) __attribute__((always_inline)) {
        constexpr int LPT_LOG2 = decltype(lconst)::value;
        constexpr int LPT = 1 << LPT_LOG2;
        const int tslot = lane >> LPT_LOG2;  // my term slot (ascending term id)
        const int jl = lane & (LPT - 1);
        const bool has_term = tslot < nt;
        int64_t base = 0;
        const int32_t *skip_row = ix.tile_skip;
        float my_idf = 0.f, my_qw = 0.f;
        if (has_term) {
            const int term = q_term[t0 + tslot];
            base = ix.term_ptr[term];
            skip_row = ix.tile_skip + (int64_t)term * row;
            my_idf = ix.idf[term];
            my_qw = q_weight[t0 + tslot];
        }
        // Initial threshold: with all query idf >= 0 a doc's score is at least any single contribution, so the K-th
        // largest contribution of any one term (K >= k, from the index's term_bound table) is an exact lower bound
        // on this shard's k-th best score.  Candidates below it can be dropped from the very first unit.
        {
            const int col = bound_column(k);
            float bnd = 0.0f;
            if (ix.term_bound != nullptr && col >= 0 && has_term && my_idf > 0.0f && my_qw > 0.0f)
                bnd = 0.0f + (ix.term_bound[(int64_t)q_term[t0 + tslot] * 4 + col] * my_idf) * my_qw;
            const bool neg = has_term && (my_idf < 0.0f || my_qw < 0.0f);
            const unsigned t0bits = wave_max(__float_as_uint(bnd > 0.0f ? bnd : 0.0f));
            tk.tau = (__ballot(neg) != 0ull) ? 0u : uniu(t0bits);
        }
        unsigned tau_seen = 0xFFFFFFFFu;  // uniform: tau the screening threshold vthr was derived from
        float vthr = 0.0f;
        const int32_t *const doc0 = ix.post_doc;
        const VT *const val0 = reinterpret_cast<const VT *>(ix.post_val);

        // unit boundary j of my term: #postings with doc < j * tpu * G
        auto bound = [&](int j) __attribute__((always_inline)) -> int {
            return has_term ? skip_row[min(j * tpu, ix.n_tiles)] : 0;
        };

        // Issue the loads of my term's run [lo, lo + len) of the unit.  A lane loads 4 consecutive postings per step
        // (dwordx4: the group's LPT lanes read 16 * LPT contiguous bytes, whole cache lines): register r = 4 s + i
        // holds posting (s * LPT + jl) * 4 + i.  Always exactly 2 * W_R / 4 loads, no branches (idle lanes read
        // postings 0..3 through a pre-biased dummy pointer, same immediate offset), so that the compiler can wait for
        // THIS unit's data with a counted s_waitcnt vmcnt(N) while the NEXT unit's loads stay in flight.
        auto issue = [&](int lo, int len, int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) {
            const int32_t *dp = doc0 + (base + lo + 4 * jl);
            const VT *vp = val0 + (base + lo + 4 * jl);
            const int rem = len - 4 * jl;  // register r is mine iff pos(r) < rem, pos(r) = (r / 4) * 4 LPT + r % 4
#pragma unroll
            for (int s4 = 0; s4 < W_R / 4; ++s4) {
                const int eo = s4 << (LPT_LOG2 + 2);
                const bool ok = eo < rem;
                load4(ok ? dp : doc0 - eo, eo, d[4 * s4], d[4 * s4 + 1], d[4 * s4 + 2], d[4 * s4 + 3]);
                load4(ok ? vp : val0 - eo, eo, v[4 * s4], v[4 * s4 + 1], v[4 * s4 + 2], v[4 * s4 + 3]);
            }
        };

        // Score one unit from registers.  Idle lanes are made harmless once (private dummy bitmap word with a distinct
        // bit per register, value 0) and d[] becomes the doc offset inside the unit, so every pass runs unpredicated.
        // Pass 1 sets every posting's doc bit (ds_or_rtn).  A bit found already set means an earlier posting matched
        // the same doc: that (rare) lane parks its posting in the pending list, publishes the doc through LDS and
        // blanks its value.  The first posting of such a doc finds itself by comparing against the few published
        // docs and parks too.  Everything still non-blank is a single-term doc whose score is 0 + c, screened with
        // one compare against a conservative per-lane threshold.  Parked postings are summed by wave_resolve_multi in
        // ascending term order (exact).  Slots in the LDS lists come from LDS counters inside the rare exec-masked
        // blocks, so the common path carries no per-lane masks.  false -> the unit goes to tier 2 (nothing emitted).
        auto process = [&](auto nrc, int su, int len, int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) -> bool {
            constexpr int NR = decltype(nrc)::value;
            const int ubase = (su * tpu) << ix.tile_log2;
            const int rem = len - 4 * jl;  // register r holds posting pos(r) = (r / 4) * 4 LPT + r % 4 of my lane's stripe
            const int dummy = (bm_words + lane) << 5;
            if (tk.count > (unsigned)(W_LCAP - 64 - W_DUPCAP)) {  // uniform, rare: room for this unit's multi-term docs
                tk.tau = uniu(wave_list_select(S, tk.count, k));
                tk.count = (unsigned)k;
            }
            unsigned old[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool ok = (((r >> 2) << (LPT_LOG2 + 2)) + (r & 3)) < rem;
                d[r] = ok ? d[r] - ubase : dummy + r;
                v[r] = ok ? v[r] : 0.0f;
                old[r] = atomicOr(&bm[(unsigned)d[r] >> 5], 1u << (d[r] & 31));
            }
            STAMP(2);  // wait for the unit's postings + pass 1
            unsigned dm = 0;  // bit r: posting r found its doc's bit already set (an earlier posting matched the same doc)
#pragma unroll
            for (int r = 0; r < NR; ++r) dm |= ((old[r] >> (d[r] & 31)) & 1u) << r;
            bool dense = false;
            if (__ballot(dm != 0) != 0ull) {  // uniform: some doc of this unit is matched by several terms (~2 units in 3)
                unsigned ndup = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r) ndup += (unsigned)__popcll(__ballot(((dm >> r) & 1u) != 0u));
                if (ndup > (unsigned)W_DUPCAP) {
                    dense = true;  // too many multi-term docs for this path: tier 2 takes the unit (nothing was emitted)
                } else {
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        // postings that found their doc's bit set, still unresolved (v != 0)
                        unsigned long long m = __ballot(((dm >> r) & 1u) != 0u && v[r] != 0.0f);
                        while (m != 0ull) {  // uniform loop, about one doc per unit on sparse queries
                            const int src = __ffsll((long long)m) - 1;
                            const int dd = __builtin_amdgcn_readlane(d[r], src);  // the doc (offset in the unit), wave-uniform
                            // A doc occurs at most once per term, hence at most once per lane: pick up my posting of it
                            // (if any) and blank it, so that the single-term screening below never sees it.
                            float myv = 0.0f;
#pragma unroll
                            for (int r2 = 0; r2 < NR; ++r2) {
                                const bool hit = d[r2] == dd;
                                myv = hit ? v[r2] : myv;
                                v[r2] = hit ? 0.0f : v[r2];
                            }
                            const float myc = 0.0f + (myv * my_idf) * my_qw;
                            // exact score: contributions in ascending term id = ascending lane (term slots own lane groups)
                            unsigned long long mm = __ballot(myv != 0.0f);
                            float sum = 0.0f;
                            while (mm != 0ull) {
                                const int l2 = __ffsll((long long)mm) - 1;
                                sum = sum + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(myc), l2));
                                mm &= mm - 1ull;
                            }
                            const unsigned b = __float_as_uint(sum);
                            if (sum > 0.0f && b >= tk.tau) {  // uniform; room for W_DUPCAP entries was made above
                                if (lane == 0) {
                                    S.lbits[tk.count] = b;
                                    S.ldoc[tk.count] = dd + ubase;
                                }
                                ++tk.count;
                            }
                            m = __ballot(((dm >> r) & 1u) != 0u && v[r] != 0.0f);
                        }
                    }
                }
            }
            STAMP(3);  // duplicate resolution
#pragma unroll
            for (int r = 0; r < NR; ++r) bm[(unsigned)d[r] >> 5] = 0u;  // restore the bitmap
            STAMP(4);  // restore
            if (dense) return false;
            if (dbg & 2) return true;
            STAMP(5);
            if (!(dbg & 1)) {
                // Single-term docs.  Almost no posting can beat tau once the list has warmed up, so a conservative
                // per-lane threshold on the stored value (vthr <= the smallest v whose contribution could reach tau,
                // and > 0 so that blanked registers never pass) screens them with one compare; the exact fp32 test
                // runs only for survivors.
                if (tk.tau != tau_seen) {  // uniform, rare
                    tau_seen = tk.tau;
                    const float tau_f = __uint_as_float(max(tau_seen, 1u));
                    vthr = (my_idf > 0.0f && my_qw > 0.0f) ? fmaxf(((tau_f / my_qw) / my_idf) * 0.99999f, __uint_as_float(1u))
                                                         : __builtin_inff();
                }
                // one test per unit in the steady state: the lane's largest value (v_max3 tree) against its threshold
                float vmax = v[0];
#pragma unroll
                for (int r = 1; r + 1 < NR; r += 2) vmax = fmaxf(fmaxf(vmax, v[r]), v[r + 1]);
                if constexpr (NR % 2 == 0) vmax = fmaxf(vmax, v[NR - 1]);
                if (__ballot(vmax >= vthr) != 0ull && !(dbg & 64)) {  // uniform, rare after warm-up (debug 64: timing experiment, screening only)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const bool pass = v[r] >= vthr;
                        if (__ballot(pass) != 0ull) {
                            const float c = 0.0f + (v[r] * my_idf) * my_qw;
                            const unsigned b = __float_as_uint(c);
                            wave_append(S, tk, k, pass && c > 0.0f && b >= tk.tau, b, d[r] + ubase);
                        }
                    }
                }
            }
            STAMP(6);  // candidate screening + appends (+ selects)
            return true;
        };

        auto flag_tier2 = [&](int su) __attribute__((always_inline)) {
            if (lane == 0) atomicOr(&ovf[(int64_t)q * ovf_words + (su >> 5)], 1u << (su & 31));
            flagged = true;
        };

        // ---- software pipeline over units, unrolled by two (register sets A / B alternate): issue the loads of
        //      unit u+1, then score unit u from registers ----
        int dA[W_R], dB[W_R];
        float vA[W_R], vB[W_R];
        int b0 = bound(su_lo), b1 = bound(su_lo + 1), b2 = bound(su_lo + 2);  // b_j = boundary j; unit u = [b_u, b_{u+1})
        int lenA = (su_lo < su_hi) ? b1 - b0 : 0, lenB = 0;
        issue(b0, (__ballot(lenA > W_R * LPT) == 0ull) ? lenA : 0, dA, vA);
        // one stage: unit su is in (lenc, d, v); unit su+1 goes to (lenn, dn, vn)
        auto stage = [&](int su, int lenc, int (&d)[W_R], float (&v)[W_R], int &lenn, int (&dn)[W_R],
                         float (&vn)[W_R]) __attribute__((always_inline)) {
            const int b3 = bound(su + 3);  // boundary needed two units from now (clamped to the row end)
            lenn = (su + 1 < su_hi) ? b2 - b1 : 0;
            const bool fitn = __ballot(lenn > W_R * LPT) == 0ull;  // uniform: every term's run fits W_R steps
            STAMP(0);  // loop overhead / previous tail
            issue(b1, fitn ? lenn : 0, dn, vn);
            STAMP(1);  // issue
            if (dbg & 4) {  // timing experiment: loads only (results are wrong)
#pragma unroll
                for (int r = 0; r < W_R; ++r) sink += d[r] ^ (int)__float_as_uint(v[r]);
            } else if (__ballot(lenc > W_R * LPT) != 0ull) {
                flag_tier2(su);
            } else if (__ballot(lenc > 0) != 0ull) {
                bool fine;
                bool done = false;
                if constexpr (W_R > 12) {
                    if (__ballot(lenc - 4 * jl > 12 * LPT) != 0ull) {  // uniform: the fourth load step holds postings
                        fine = process(IntC<16>{}, su, lenc, d, v);
                        done = true;
                    }
                }
                if constexpr (W_R > 8) {
                    if (!done && __ballot(lenc - 4 * jl > 8 * LPT) != 0ull) {  // uniform: the third load step holds postings
                        fine = process(IntC<12>{}, su, lenc, d, v);
                        done = true;
                    }
                }
                if (!done) {
                    if (__ballot(lenc - 4 * jl > 4 * LPT) != 0ull)
                        fine = process(IntC<8>{}, su, lenc, d, v);
                    else
                        fine = process(IntC<4>{}, su, lenc, d, v);
                }
                if (!fine) flag_tier2(su);
            }
            b1 = b2;
            b2 = b3;
        };
        for (int su = su_lo; su < su_hi; su += 2) {
            stage(su, lenA, dA, vA, lenB, dB, vB);
            if (su + 1 < su_hi) stage(su + 1, lenB, dB, vB, lenA, dA, vA);
        }
    };
    switch (6 - lg) {
        case 0: run(IntC<0>{}); break;
        case 1: run(IntC<1>{}); break;
        case 2: run(IntC<2>{}); break;
        case 3: run(IntC<3>{}); break;
        case 4: run(IntC<4>{}); break;
        case 5: run(IntC<5>{}); break;
        default: run(IntC<6>{}); break;
    }
    if ((dbg & 4) && sink == 0x7F123457) cand_count[list] = sink;  // keeps the loads of the timing experiment alive
    unsigned count = tk.count;
    if (dbg & 32) count = 0;  // timing experiment: no final selection / ranking
    if (count > (unsigned)k) {
        wave_list_select(S, count, k);
        count = (unsigned)k;
    }
#ifdef SRX_STAMP
    STAMP(7);  // epilogue (final resolve / select)
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamp[i], st_acc[i]);
        atomicAdd(&g_stamp[8], 1ull);
    }
#endif
    if (nsq == 1 && !flagged && out_doc != nullptr) {
        // This wave holds the query's complete top-k (one split, nothing handed to tier 2): rank it here and write
        // the final row, so the merge kernel can skip the query.
        wave_rank_emit(S, reinterpret_cast<unsigned long long *>(bm), count, k, doc_base, out_doc + (int64_t)q * out_row_stride,
                       out_score + (int64_t)q * out_row_stride);
        if (lane == 0) {
            out_count[(int64_t)q * out_cnt_stride] = (int)count;
            cand_count[list] = -1;  // tells the merge kernel this query is final
        }
        return;
    }
    const int64_t o = list * k;
    for (unsigned i = lane; i < count; i += 64) {
        cand_doc[o + i] = S.ldoc[i];
        cand_score[o + i] = __uint_as_float(S.lbits[i]);
    }
    if (lane == 0) {
        cand_count[list] = (int)count;
        if (flagged) work[1 + atomicAdd(&work[0], 1)] = (int)blockIdx.x;
    }
}

// ------------------------------------------------------------------------------------------------
// Merge kernel: one workgroup per (query, group of lists).  Selects the top-k of up to
// MERGE_NPT*256 candidates; if `final`, ranks them (bitonic sort on (score desc, doc asc)), adds
// doc_base and pads the row.
// ------------------------------------------------------------------------------------------------
struct MergeShared {
    TopkShared tk;
    unsigned hist[RADIX_BINS];
    unsigned long long sortkey[KMAX];
    int lstart[64];
};

// Wave-level final merge for the common small case (n_lists * k <= 1024 candidates per query, k <= 128: the splits /
// tiers of one shard, or 8 shards' top-100): one wavefront per query, no barrier.  The candidates are compacted into
// an LDS list, the exact list selection of tier 1 shrinks it to k and the wave ranks and writes the row.
constexpr int MW_CAP = 1024;
struct MergeWaveShared {
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
                                                                 int64_t out_cnt_stride, const int *__restrict__ gate) {
    __shared__ MergeWaveShared MW[WAVES];
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * WAVES + (threadIdx.x >> 6);
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
                                                            int64_t out_cnt_stride, const int *__restrict__ gate) {
    __shared__ MergeShared M;
    const int tid = threadIdx.x;
    const int q = blockIdx.x / n_groups;
    const int g = blockIdx.x - q * n_groups;
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
    // rank: bitonic sort, descending on key64 = score bits : (0x7FFFFFFF - doc)
    unsigned n = 1;
    while (n < cnt) n <<= 1;
    for (unsigned i = tid; i < n; i += THREADS)
        M.sortkey[i] = i < cnt ? (((unsigned long long)M.tk.bits[i] << 32) | (0x7FFFFFFFu - (unsigned)M.tk.doc[i])) : 0ull;
    __syncthreads();
    for (unsigned size = 2; size <= n; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned i = tid; i < (n >> 1); i += THREADS) {
                const unsigned pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = M.sortkey[pos], b = M.sortkey[pos + stride];
                const bool desc = (pos & size) == 0;
                if (desc ? (a < b) : (a > b)) {
                    M.sortkey[pos] = b;
                    M.sortkey[pos + stride] = a;
                }
            }
            __syncthreads();
        }
    }
    const int64_t o = (int64_t)q * out_row_stride;  // final rows may live in a strided (packed) buffer
    for (unsigned i = tid; i < (unsigned)k; i += THREADS) {
        if (i < cnt) {
            const unsigned long long x = M.sortkey[i];
            out_doc[o + i] = (int32_t)(doc_base + (int64_t)(0x7FFFFFFFu - (unsigned)(x & 0xFFFFFFFFull)));
            out_score[o + i] = __uint_as_float((unsigned)(x >> 32));
        } else {
            out_doc[o + i] = -1;
            out_score[o + i] = 0.0f;
        }
    }
    if (tid == 0) out_count[(int64_t)q * out_cnt_stride] = (int)cnt;
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
    if (!d->term_ptr || !d->tile_skip || !d->idf || (d->nnz > 0 && (!d->post_doc || !d->post_val)))
        return fail(SRX_ERR_INVALID, "srx_index_create: null index array%s");
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
    *out = ix;
    return SRX_OK;
}

SRX_API void srx_index_destroy(srx_index *ix) {
    if (!ix) return;
    if (ix->ev) {
        for (int i = 0; i < PROF_EVENTS * PROF_SLOTS; ++i) (void)hipEventDestroy(ix->ev[i]);
        delete[] ix->ev;
    }
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

// Supertile (unit) = the doc range one tier-1 unit covers (<= 2^W_UNIT_LOG2 docs, the wave bitmap).  Auto rule: the
// largest power of two (tile .. tile*64) for which the run of an average term inside a unit fits the registers of
// its lane group in the reference case of an 8-term query (8 lanes x W_R steps), with a 3.5-sigma Poisson margin.
Plan make_plan(const srx_index *ix, int nq, int k) {
    Plan p;
    const srx_index_desc &d = ix->d;
    const int max_tpu_bitmap = (1 << W_UNIT_LOG2) >> d.tile_log2;  // the tier-1 bitmap covers 65536 docs
    int tpu;
    if (ix->opts.unit_tiles > 0) {
        tpu = ix->opts.unit_tiles;
    } else if (ix->opts.supertile_log2 != 0) {
        tpu = 1 << (ix->opts.supertile_log2 - d.tile_log2);
    } else {
        // Auto: the largest unit (in tiles) for which the run of an average term inside a unit overflows the registers
        // of its lane group (8 lanes x W_R postings in the reference case of an 8-term query) with negligible
        // probability (mean + 5 sigma, Poisson).  Shorter units also keep the per-unit duplicate work (quadratic in
        // the unit's postings) small.
        const double per_doc_per_term = (double)d.nnz / ((double)d.n_docs * (double)d.vocab);  // E[postings of a term per doc]
        auto fits = [&](int t) {
            const double mean = per_doc_per_term * (double)t * (double)(1ll << d.tile_log2);
            return mean + 5.0 * sqrt(mean) <= 8.0 * W_R;
        };
        tpu = 1;
        while (tpu < MAX_TPS && tpu < max_tpu_bitmap && fits(tpu + 1)) ++tpu;
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
    return lists * k * 8 + lists * 4 + (int64_t)nq * p.ovf_words * 4 + 4 * (1 + (int64_t)nq * p.n_splits) + 256;
}

namespace {
int search_impl(srx_index *ix, const int32_t *q_ptr, const int32_t *q_term, const float *q_weight, int32_t nq,
                int32_t k, int32_t *out_doc, float *out_score, int32_t *out_count, int64_t ors, int64_t ocs,
                void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (!ix) return fail(SRX_ERR_INVALID, "srx_search: null index%s");
    if (nq < 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_search: need nq >= 0 and 1 <= k <= 1024%s");
    if (nq == 0) return SRX_OK;
    if (!q_ptr || !out_doc || !out_score || !out_count) return fail(SRX_ERR_INVALID, "srx_search: null query / output pointer%s");
    const int64_t need = srx_search_workspace_bytes(ix, nq, k);
    if (!workspace || workspace_bytes < need) return fail(SRX_ERR_NOMEM, "srx_search: workspace too small%s");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ix->d.device));
    const Plan p = make_plan(ix, nq, k);
    const int64_t lists = (int64_t)nq * p.lists_per_q;
    const int64_t blocks = (int64_t)p.n_whole + (int64_t)(nq - p.n_whole) * p.n_splits;
    if (lists > 0x7FFFFFFFll) return fail(SRX_ERR_INVALID, "srx_search: nq * splits overflows the grid%s");
    int32_t *cand_doc = (int32_t *)workspace;
    float *cand_score = (float *)(cand_doc + lists * k);
    int32_t *cand_count = (int32_t *)(cand_score + lists * k);
    unsigned *ovf = (unsigned *)(cand_count + lists);
    int *work = (int *)(ovf + (int64_t)nq * p.ovf_words);  // work[0] = count, then block ids

    IndexView v;
    v.term_ptr = ix->d.term_ptr;
    v.post_doc = ix->d.post_doc;
    v.post_val = ix->d.post_val;
    v.tile_skip = ix->d.tile_skip;
    v.idf = ix->d.idf;
    v.term_bound = (ix->opts.reserved & 16) ? nullptr : ix->d.term_bound;  // debug bit 16: ignore the score bounds
    v.n_docs = ix->d.n_docs;
    v.vocab = ix->d.vocab;
    v.tile_log2 = ix->d.tile_log2;
    v.n_tiles = ix->d.n_tiles;
    const int dbg = ix->opts.reserved;

    const bool prof = ix->opts.profile != 0;
    hipEvent_t *ev = nullptr;
    if (prof) {
        if (!ix->ev) {
            ix->ev = new (std::nothrow) hipEvent_t[PROF_EVENTS * PROF_SLOTS];
            if (!ix->ev) return fail(SRX_ERR_NOMEM, "srx_search: host allocation failed%s");
            for (int i = 0; i < PROF_EVENTS * PROF_SLOTS; ++i) HIP_TRY(hipEventCreate(&ix->ev[i]));
        }
        ev = ix->ev + PROF_EVENTS * (int)(ix->ev_calls % PROF_SLOTS);
    }
    // one memset: list counts (tier-2 lists that never run must read as empty), overflow bitmap, worklist counter
    HIP_TRY(hipMemsetAsync(cand_count, 0, (size_t)(lists + (int64_t)nq * p.ovf_words + 1) * 4, stream));
    if (prof) HIP_TRY(hipEventRecord(ev[0], stream));
    // tier 1: one wavefront per (query, split); dynamic LDS = the unit's doc bitmap (1 bit per doc, >= 1 KiB, 16-B multiple)
    int64_t unit_docs = (int64_t)p.tpu << ix->d.tile_log2;
    if (unit_docs > (1 << W_UNIT_LOG2)) unit_docs = 1 << W_UNIT_LOG2;  // larger units are served by tier 2 anyway
    const unsigned bm_bytes = ((unsigned)(((unit_docs + 31) / 32 < 256 ? 256 : (unit_docs + 31) / 32) * 4 + 15) & ~15u) + 256u;  // + 64 dummy words
    if (ix->d.val_type == SRX_VAL_F32)
        hipLaunchKernelGGL(srx_wave_kernel<float>, dim3((unsigned)blocks), dim3(64), bm_bytes, stream, v, q_ptr, q_term, q_weight,
                           nq, k, p.n_splits, p.n_whole, p.tpu, p.n_super, dbg, ovf, p.ovf_words, p.lists_per_q, work,
                           cand_doc, cand_score, cand_count, ix->d.doc_base, out_doc, out_score, out_count, ors, ocs);
    else
        hipLaunchKernelGGL(srx_wave_kernel<__half>, dim3((unsigned)blocks), dim3(64), bm_bytes, stream, v, q_ptr, q_term, q_weight,
                           nq, k, p.n_splits, p.n_whole, p.tpu, p.n_super, dbg, ovf, p.ovf_words, p.lists_per_q, work,
                           cand_doc, cand_score, cand_count, ix->d.doc_base, out_doc, out_score, out_count, ors, ocs);
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(ev[1], stream));
    // tier 2: flagged units, long queries, k > 128 -- a fixed grid drains the worklist tier 1 filled
    const unsigned t2_grid = (unsigned)(blocks < 1024 ? blocks : 1024);
    if (ix->d.val_type == SRX_VAL_F32)
        hipLaunchKernelGGL(srx_score_kernel<float>, dim3(t2_grid), dim3(THREADS), 0, stream, v, q_ptr, q_term,
                           q_weight, nq, k, p.n_splits, p.n_whole, p.tpu, p.n_super, dbg, ovf, p.ovf_words, p.lists_per_q,
                           work, cand_doc, cand_score, cand_count);
    else
        hipLaunchKernelGGL(srx_score_kernel<__half>, dim3(t2_grid), dim3(THREADS), 0, stream, v, q_ptr, q_term,
                           q_weight, nq, k, p.n_splits, p.n_whole, p.tpu, p.n_super, dbg, ovf, p.ovf_words, p.lists_per_q,
                           work, cand_doc, cand_score, cand_count);
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(ev[2], stream));
    if (k <= W_KMAX && (int64_t)p.lists_per_q * k <= MW_CAP && p.lists_per_q <= 256 && !(dbg & 256))
        hipLaunchKernelGGL(srx_merge_wave_kernel, dim3((unsigned)((nq + WAVES - 1) / WAVES)), dim3(THREADS), 0, stream, cand_doc,
                           cand_score, cand_count, nq, p.lists_per_q, k, 0, (int64_t)k, (int64_t)1, ix->d.doc_base, out_doc,
                           out_score, out_count, ors, ocs, (const int *)nullptr);
    else
        hipLaunchKernelGGL(srx_merge_kernel, dim3((unsigned)nq), dim3(THREADS), 0, stream, cand_doc, cand_score, cand_count, nq,
                           p.lists_per_q, k, p.lists_per_q, 1, 1, 0, (int64_t)k, (int64_t)1, ix->d.doc_base, out_doc, out_score,
                           out_count, ors, ocs, (const int *)nullptr);
    HIP_TRY(hipGetLastError());
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

namespace {
int merge_impl(int32_t device, const int32_t *in_doc, const float *in_score, const int32_t *in_count, int32_t nq,
               int32_t n_lists, int32_t k, int lay, int64_t row_stride, int64_t cnt_stride, int32_t *out_doc,
               float *out_score, int32_t *out_count, int64_t ors, int64_t ocs, void *workspace, int64_t workspace_bytes,
               void *stream_v, const int *gate = nullptr) {
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
                           oc, (int64_t)k, (int64_t)1, gate);
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
                           out_count, ors, ocs, gate);
    else
        hipLaunchKernelGGL(srx_merge_kernel, dim3((unsigned)nq), dim3(THREADS), 0, stream, cur_doc, cur_score, cur_count, nq,
                           lists, k, lists, 1, 1, lay, row_stride, cnt_stride, (int64_t)0, out_doc, out_score, out_count, ors, ocs, gate);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}
}  // namespace

SRX_API int srx_merge_topk(int32_t device, const int32_t *in_doc, const float *in_score, const int32_t *in_count,
                           int32_t nq, int32_t n_lists, int32_t k, int32_t gathered, int32_t *out_doc, float *out_score,
                           int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream_v) {
    return merge_impl(device, in_doc, in_score, in_count, nq, n_lists, k, gathered ? 1 : 0, (int64_t)k, (int64_t)1, out_doc,
                      out_score, out_count, (int64_t)k, (int64_t)1, workspace, workspace_bytes, stream_v);
}

SRX_API int srx_merge_topk_packed(int32_t device, const int32_t *packed, int32_t nq, int32_t n_lists, int32_t k,
                                  int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                                  int64_t workspace_bytes, void *stream_v) {
    if (!packed || k <= 0) return fail(SRX_ERR_INVALID, "srx_merge_topk_packed: bad argument%s");
    const int64_t row = 2 * (int64_t)k + 1;  // [k doc ids][k score bit patterns][count]
    return merge_impl(device, packed, reinterpret_cast<const float *>(packed + k), packed + 2 * k, nq, n_lists, k, 1, row, row,
                      out_doc, out_score, out_count, (int64_t)k, (int64_t)1, workspace, workspace_bytes, stream_v);
}

SRX_API int srx_merge_topk_packed_out(int32_t device, const int32_t *packed, int32_t nq, int32_t n_lists, int32_t k,
                                      int32_t *out_packed, void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (!packed || !out_packed || k <= 0) return fail(SRX_ERR_INVALID, "srx_merge_topk_packed_out: bad argument%s");
    const int64_t row = 2 * (int64_t)k + 1;
    return merge_impl(device, packed, reinterpret_cast<const float *>(packed + k), packed + 2 * k, nq, n_lists, k, 1, row, row,
                      out_packed, reinterpret_cast<float *>(out_packed + k), out_packed + 2 * k, row, row, workspace,
                      workspace_bytes, stream_v);
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
//   score[q][d] = f32( f64(sum_i query_i8[q][i] * corpus_i8[d][i]) * query_scale[q] * corpus_scale[d] )
// The integer dot products are one MFMA GEMM (v_mfma_i32_32x32x32_i8, exact); the two scalings are done in fp64 like
// the reference's NumPy scalars (int32 * float32 -> float64), so the stored fp32 score is the reference's bit for bit.
// First form: the scaled scores go through HBM once (a [query batch][n_docs] fp32 matrix in the workspace) and the
// block top-k machinery of the sparse path ranks each row; only scores > 0 are results (retriever_registry.py:519).
// ================================================================================================
namespace {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int DENSE_CNT_STRIDE = 32;  // ints between two queries' candidate counters: one 128-byte line each (all waves
                                      // add to these: counters sharing a line serialise in one L2 channel)

// Query batch in MFMA-fragment order: apack[(tile * KS + s) * 64 + lane] = the 16 bytes lane `lane` feeds into k-step s
// of query tile `tile` (row tile * 32 + (lane & 31), columns 32 s + 16 (lane >> 5) ..).  A wave's A load is then one
// contiguous 1 KiB block instead of 32 scattered 32-byte segments (the request rate of the texture path was the limit).
__global__ __launch_bounds__(THREADS) void srx_dense_pack_queries_kernel(const int8_t *__restrict__ queries, int nq, int dim,
                                                                         v4i *__restrict__ apack) {
    const int ks = dim / 32;
    const int64_t n = (int64_t)((nq + 31) / 32) * ks * 64;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const int lane = (int)(i & 63);
        const int64_t ts = i >> 6;
        const int s = (int)(ts % ks);
        const int q = (int)(ts / ks) * 32 + (lane & 31);
        v4i x = {0, 0, 0, 0};
        if (q < nq) x = *reinterpret_cast<const v4i *>(queries + (int64_t)q * dim + s * 32 + 16 * (lane >> 5));
        apack[i] = x;
    }
}

// One wave = 32 docs x (all queries, 32 at a time); a workgroup = 4 waves = 128 consecutive docs.  The wave keeps its
// docs' B fragments in registers for the whole query loop (KS k-steps of 32: lane l holds corpus[d0 + (l & 31)]
// [32 s + 16 (l >> 5) .. + 15], one 16-byte load); the queries' A fragments (the same map on the query rows) stream
// from L2.  D[row = query][col = doc]: lane l holds doc l & 31, rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
template <int KS>
__global__ __launch_bounds__(THREADS) void srx_dense_i8_scores_kernel(const int8_t *__restrict__ corpus,
                                                                       const float *__restrict__ corpus_scale,
                                                                       int64_t n_docs, const v4i *__restrict__ apack,
                                                                       const float *__restrict__ query_scale, int nq,
                                                                       float *__restrict__ scores, int64_t ld,
                                                                       const int *__restrict__ gate) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t d0 = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * 32;
    if (d0 >= n_docs) return;
    if (gate != nullptr && *gate == 0) return;  // fallback pass: only runs when some query's candidate buffer overflowed
    constexpr int DIM = KS * 32;
    const int64_t d = d0 + r;
    const bool dok = d < n_docs;
    v4i B[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        B[s] = (v4i){0, 0, 0, 0};
        if (dok) B[s] = *reinterpret_cast<const v4i *>(corpus + d * DIM + s * 32 + 16 * h);
    }
    const double ds = dok ? (double)corpus_scale[d] : 0.0;
    for (int q0 = 0; q0 < nq; q0 += 32) {
        const int qa = q0 + r;
        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i A = apack[((int64_t)(q0 >> 5) * KS + s) * 64 + lane];
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[s], acc, 0, 0, 0);
        }
        // the tile's 32 query scales: one coalesced load, then a lane permute per accumulator row (a global load per
        // row would put 16 dependent L1 round trips behind every tile)
        const float qs_mine = qa < nq ? query_scale[qa] : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float qsr = __shfl(qs_mine, row);
            if (dok && q0 + row < nq) scores[(int64_t)(q0 + row) * ld + d] = (float)(((double)acc[reg] * (double)qsr) * ds);
        }
    }
}

// The same GEMM with the top-k filter fused in: instead of writing the score, a lane keeps it only if it can still
// reach the query's top k (score > 0 and >= tau[q], a valid lower bound of the k-th best score taken from a sample of
// the corpus) and appends (doc, score) to the query's candidate buffer (one atomicAdd per query row and lane half).
// A full buffer raises the query's overflow flag (the caller then re-ranks that query through the score matrix).
template <int KS>
__global__ __launch_bounds__(THREADS) void srx_dense_i8_filter_kernel(const int8_t *__restrict__ corpus,
                                                                       const float *__restrict__ corpus_scale,
                                                                       int64_t n_docs, const v4i *__restrict__ apack,
                                                                       const float *__restrict__ query_scale, int nq,
                                                                       const unsigned *__restrict__ tau, int cap,
                                                                       int64_t doc_base, int32_t *__restrict__ buf_doc,
                                                                       float *__restrict__ buf_score,
                                                                       int *__restrict__ buf_cnt, int *__restrict__ ovf,
                                                                       int *__restrict__ any_ovf) {
    // DT doc tiles of 32 per wave: with two, every A fragment (query tile) read from L2 feeds two MFMAs; the B
    // fragments of both tiles must fit the register file (KS <= 12, i.e. rows up to 384 bytes).
    constexpr int DT = KS <= 12 ? 2 : 1;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t d0 = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * (32 * DT);
    if (d0 >= n_docs) return;
    constexpr int DIM = KS * 32;
    v4i B[DT][KS];
    double ds[DT];
    bool dok[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int64_t d = d0 + 32 * t + r;
        dok[t] = d < n_docs;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            B[t][s] = (v4i){0, 0, 0, 0};
            if (dok[t]) B[t][s] = *reinterpret_cast<const v4i *>(corpus + d * DIM + s * 32 + 16 * h);
        }
        ds[t] = dok[t] ? (double)corpus_scale[d] : 0.0;
    }
    for (int q0 = 0; q0 < nq; q0 += 32) {
        const int qa = q0 + r;
        v16i acc[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i A = apack[((int64_t)(q0 >> 5) * KS + s) * 64 + lane];
#pragma unroll
            for (int t = 0; t < DT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[t][s], acc[t], 0, 0, 0);
        }
        // the tile's 32 query scales and thresholds: one coalesced load each, then a lane permute per accumulator row
        const float qs_mine = qa < nq ? query_scale[qa] : 0.0f;
        const unsigned tau_mine = qa < nq ? tau[qa] : 0xFFFFFFFFu;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            // Survivors of one 32 x 32 tile.  Three phases so that the (returning) atomics of all 16 accumulator
            // registers are in flight together -- one global round trip per tile instead of one per register with
            // survivors: scores + pass bits; one atomicAdd per (query row, lane half) with survivors; broadcast the
            // bases and store.
            const int64_t d = d0 + 32 * t + r;
            float scv[16];
            unsigned passbits = 0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;  // differs between the two lane halves
                const float qsr = __shfl(qs_mine, row);
                const unsigned taur = (unsigned)__shfl((int)tau_mine, row);
                scv[reg] = dok[t] ? (float)(((double)acc[t][reg] * (double)qsr) * ds[t]) : 0.0f;
                if (q0 + row < nq && scv[reg] > 0.0f && __float_as_uint(scv[reg]) >= taur) passbits |= 1u << reg;
            }
            if (__ballot(passbits != 0u) != 0ull) {  // uniform; about 6 survivors per tile at the design point
                int basev[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const bool pass = (passbits >> reg) & 1u;
                    const unsigned long long m = __ballot(pass);
                    const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)(m & 0xFFFFFFFFull);  // my half's survivors: one query row
                    basev[reg] = 0;
                    if (pass && (mh & ((1u << r) - 1u)) == 0u)  // first survivor of the row reserves room for all of them
                        basev[reg] = atomicAdd(&buf_cnt[(q0 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * DENSE_CNT_STRIDE], __popc(mh));
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const bool pass = (passbits >> reg) & 1u;
                    const unsigned long long m = __ballot(pass);
                    if (m != 0ull) {  // uniform
                        const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)(m & 0xFFFFFFFFull);
                        const int base = __shfl(basev[reg], h * 32 + (mh ? __ffs((int)mh) - 1 : 0));
                        if (pass) {
                            const int q = q0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                            const int p = base + __popc(mh & ((1u << r) - 1u));
                            if (p < cap) {
                                buf_doc[(int64_t)q * cap + p] = (int32_t)(doc_base + d);
                                buf_score[(int64_t)q * cap + p] = scv[reg];
                            } else {
                                ovf[q] = 1;
                                *any_ovf = 1;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Row top-k: one workgroup per (query, split of the doc range) folds its slice of the score row into an exact lazy
// top-k list (topk_fold, the sparse path's machinery); the merge kernels rank the splits' lists.
constexpr int DENSE_NPT = 16;
// mode 0: rank scores[q][lo..hi) (ids = doc_base + column).  mode 1: rank the query's candidate buffer (buf_doc /
// scores hold (doc, score) pairs, buf_cnt[q] of them).  only_flag: +1 = only queries with ovf[q] != 0, -1 = only queries
// with ovf[q] == 0, 0 = all; a skipped query writes count -1 for its first list (the merge kernels then leave its
// output row alone).  tau_out (optional): the k-th best score's bits when the list holds k entries, else 0.
__global__ __launch_bounds__(THREADS) void srx_dense_topk_kernel(const float *__restrict__ scores, int64_t ld, int64_t n_docs,
                                                                 int nq, int k, int n_splits, int64_t doc_base, int mode,
                                                                 const int32_t *__restrict__ buf_doc,
                                                                 const int *__restrict__ buf_cnt, int cap,
                                                                 const int *__restrict__ ovf, int only_flag,
                                                                 const int *__restrict__ gate,
                                                                 int32_t *__restrict__ cand_doc,
                                                                 float *__restrict__ cand_score,
                                                                 int32_t *__restrict__ cand_count,
                                                                 unsigned *__restrict__ tau_out) {
    __shared__ MergeShared M;
    const int tid = threadIdx.x;
    const int q = blockIdx.x / n_splits, split = blockIdx.x - q * n_splits;
    if (q >= nq) return;
    if (gate != nullptr && *gate == 0) return;
    if (only_flag != 0 && ((ovf[q] != 0) != (only_flag > 0))) {
        if (tid == 0) cand_count[blockIdx.x] = split == 0 ? -1 : 0;
        return;
    }
    int64_t total = n_docs;
    if (mode == 1) total = min(buf_cnt[q * DENSE_CNT_STRIDE], cap);
    const int64_t lo = total * split / n_splits, hi = total * (split + 1) / n_splits;
    if (tid == 0) {
        M.tk.count = 0;
        M.tk.tau = 0;
    }
    __syncthreads();
    const float *row = scores + (int64_t)q * ld;
    const int32_t *drow = mode == 1 ? buf_doc + (int64_t)q * ld : nullptr;
    for (int64_t c0 = lo; c0 < hi; c0 += (int64_t)THREADS * DENSE_NPT) {
        unsigned ubits[DENSE_NPT];
        int udoc[DENSE_NPT];
        const unsigned tau = M.tk.tau;
#pragma unroll
        for (int n = 0; n < DENSE_NPT; ++n) {
            const int64_t c = c0 + (int64_t)n * THREADS + tid;
            float x = 0.0f;
            int dd = 0;
            if (c < hi) {
                x = row[c];
                dd = mode == 1 ? drow[c] : (int)(doc_base + c);
            }
            const unsigned b = __float_as_uint(x);
            ubits[n] = (x > 0.0f && b >= tau) ? b : 0u;
            udoc[n] = dd;
        }
        topk_fold<DENSE_NPT, true>(ubits, udoc, k, M.tk, M.hist);
    }
    __syncthreads();
    topk_shrink(k, M.tk, M.hist);
    const unsigned cnt = M.tk.count;
    const int64_t o = (int64_t)blockIdx.x * k;
    unsigned mn = 0xFFFFFFFFu;
    for (unsigned i = tid; i < cnt; i += THREADS) {
        cand_doc[o + i] = M.tk.doc[i];
        cand_score[o + i] = __uint_as_float(M.tk.bits[i]);
        mn = min(mn, M.tk.bits[i]);
    }
    if (tid == 0) cand_count[blockIdx.x] = (int)cnt;
    if (tau_out != nullptr) {  // n_splits == 1 here
        const SumMaxMin rr = block_sum_max_min(0u, 0u, mn, M.tk.red);
        if (tid == 0) tau_out[q] = cnt >= (unsigned)k ? rr.mn : 0u;
    }
}

// Queries per pass: as many as keep the fallback's score matrix (queries x n_docs x 4 B) within 4 GiB, 32 .. 1024.
// More queries per pass = the docs' B fragments are loaded once for more query tiles.
int dense_qb(int64_t n_docs) {
    int64_t q = (4ll << 30) / (((n_docs + 63) / 64 * 64) * 4);
    q = q / 32 * 32;
    if (q < 32) q = 32;
    if (q > 1024) q = 1024;
    return (int)q;
}
constexpr int DENSE_CAP = 65536;  // candidate buffer entries per query of the filtered path
int dense_splits(int64_t n_docs, int nq, int k) {
    int64_t s = 2048 / (nq > 0 ? nq : 1);  // >= 2048 workgroups when the batch is small
    const int64_t by_docs = n_docs / (THREADS * DENSE_NPT * 4);
    if (s > by_docs) s = by_docs;
    const int64_t cap = (MERGE_NPT * THREADS) / (k > 0 ? k : 1);  // one merge level
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}
// Sample size of the threshold pass: the k-th best score of S docs leaves about k * n_docs / S survivors per query;
// aim at DENSE_CAP / 8.  0 = corpus too small for the filtered path to pay.
int64_t dense_sample(int64_t n_docs, int k) {
    int64_t S = (8 * (int64_t)k * n_docs + DENSE_CAP - 1) / DENSE_CAP;
    if (S < 16384) S = 16384;
    S = (S + 127) / 128 * 128;
    return (S * 4 <= n_docs) ? S : 0;
}
struct DenseWs {
    float *scores;
    int32_t *cand_doc;
    float *cand_score;
    int32_t *cand_count;
    unsigned *tau;
    int *buf_cnt, *ovf, *any_ovf;
    int32_t *buf_doc;
    float *buf_score;
    v4i *apack;
    int64_t bytes;
};
DenseWs dense_ws(void *base, int nq, int64_t n_docs, int k) {
    const int QB = dense_qb(n_docs);
    const int qb = nq < QB ? nq : QB;
    const int64_t ld = (n_docs + 63) / 64 * 64;
    const int ns = dense_splits(n_docs, qb, k);
    const bool filt = dense_sample(n_docs, k) > 0;
    DenseWs w;
    char *p = (char *)base;
    auto take = [&](int64_t bytes) {
        char *r = p;
        p += (bytes + 255) / 256 * 256;
        return r;
    };
    w.scores = (float *)take((int64_t)qb * ld * 4);
    w.cand_doc = (int32_t *)take((int64_t)qb * ns * k * 4);
    w.cand_score = (float *)take((int64_t)qb * ns * k * 4);
    w.cand_count = (int32_t *)take((int64_t)qb * ns * 4);
    w.tau = (unsigned *)take((int64_t)qb * 4);
    w.buf_cnt = (int *)take((int64_t)(qb * DENSE_CNT_STRIDE + qb + 1) * 4);  // counts, overflow flags, any-overflow: one memset
    w.ovf = w.buf_cnt + qb * DENSE_CNT_STRIDE;
    w.any_ovf = w.ovf + qb;
    w.buf_doc = (int32_t *)take(filt ? (int64_t)qb * DENSE_CAP * 4 : 0);
    w.buf_score = (float *)take(filt ? (int64_t)qb * DENSE_CAP * 4 : 0);
    w.apack = (v4i *)take((int64_t)((qb + 31) / 32) * 32 * 1024);  // dim <= 1024 bytes per query row
    w.bytes = (int64_t)(p - (char *)base) + 256;
    return w;
}
}  // namespace

SRX_API int64_t srx_dense_workspace_bytes(int32_t nq, int64_t n_docs, int32_t k) {
    if (nq < 0 || n_docs <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_dense_workspace_bytes: bad argument%s");
    return dense_ws(nullptr, nq, n_docs, k).bytes;
}

SRX_API int srx_dense_search_i8(int32_t device, const int8_t *corpus, const float *corpus_scale, int64_t n_docs, int32_t dim,
                                const int8_t *queries, const float *query_scale, int32_t nq, int32_t k, int64_t doc_base,
                                int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                                int64_t workspace_bytes, void *stream_v) {
    if (nq < 0 || n_docs <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_dense_search_i8: need n_docs > 0, 1 <= k <= 1024%s");
    if (dim <= 0 || dim % 32 != 0 || dim > 1024)
        return fail(SRX_ERR_INVALID, "srx_dense_search_i8: dim must be a multiple of 32, <= 1024 (pad the rows with zeros)%s");
    if (doc_base < 0 || doc_base + n_docs >= 0x7FFFFFFFll) return fail(SRX_ERR_INVALID, "srx_dense_search_i8: doc_base + n_docs must fit int32%s");
    if (nq == 0) return SRX_OK;
    if (!corpus || !corpus_scale || !queries || !query_scale || !out_doc || !out_score || !out_count)
        return fail(SRX_ERR_INVALID, "srx_dense_search_i8: null pointer%s");
    if (((uintptr_t)corpus | (uintptr_t)queries) & 15) return fail(SRX_ERR_INVALID, "srx_dense_search_i8: corpus / queries must be 16-byte aligned%s");
    const int64_t need = srx_dense_workspace_bytes(nq, n_docs, k);
    if (!workspace || workspace_bytes < need) return fail(SRX_ERR_NOMEM, "srx_dense_search_i8: workspace too small%s");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_v;
    const int QB = dense_qb(n_docs);
    const int qbmax = nq < QB ? nq : QB;
    const int64_t ld = (n_docs + 63) / 64 * 64;
    const int ns = dense_splits(n_docs, qbmax, k);
    const int64_t S = dense_sample(n_docs, k);
    const DenseWs w = dense_ws(workspace, nq, n_docs, k);
    auto blocks_for = [](int64_t docs) { return (unsigned)((docs + 32 * WAVES - 1) / (32 * WAVES)); };
    int ks_ok = 1;
    // KERNEL<KS> dispatch on dim / 32
#define SRX_DENSE_DISPATCH(KERNEL, GRID, ...)                                                                         \
    switch (dim / 32) {                                                                                               \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;                \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;                \
        case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;                \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;                \
        case 6: hipLaunchKernelGGL(KERNEL<6>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;                \
        case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;                \
        case 12: hipLaunchKernelGGL(KERNEL<12>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;              \
        case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;              \
        case 24: hipLaunchKernelGGL(KERNEL<24>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;              \
        case 32: hipLaunchKernelGGL(KERNEL<32>, dim3(GRID), dim3(THREADS), 0, stream, __VA_ARGS__); break;              \
        default: ks_ok = 0;                                                                                           \
    }
    for (int q0 = 0; q0 < nq; q0 += QB) {
        const int qb = nq - q0 < QB ? nq - q0 : QB;
        const int8_t *qp = queries + (int64_t)q0 * dim;
        const float *qs = query_scale + q0;
        int32_t *od = out_doc + (int64_t)q0 * k;
        float *os = out_score + (int64_t)q0 * k;
        int32_t *oc = out_count + q0;
        const int *no_gate = nullptr;
        hipLaunchKernelGGL(srx_dense_pack_queries_kernel, dim3(64), dim3(THREADS), 0, stream, qp, qb, (int)dim, w.apack);
        if (S > 0) {
            // ---- filtered path: threshold from a sample, GEMM with the filter fused in, rank the candidate buffers ----
            HIP_TRY(hipMemsetAsync(w.buf_cnt, 0, (size_t)(qbmax * DENSE_CNT_STRIDE + qbmax + 1) * 4, stream));
            SRX_DENSE_DISPATCH(srx_dense_i8_scores_kernel, blocks_for(S), corpus, corpus_scale, S, (const v4i *)w.apack, qs, qb, w.scores, ld, no_gate);
            if (!ks_ok) break;
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)qb), dim3(THREADS), 0, stream, w.scores, ld, S, qb, k, 1, doc_base,
                               0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, 0, no_gate, w.cand_doc,
                               w.cand_score, w.cand_count, w.tau);
            SRX_DENSE_DISPATCH(srx_dense_i8_filter_kernel, (dim / 32 <= 12 ? blocks_for((n_docs + 1) / 2) : blocks_for(n_docs)), corpus, corpus_scale, n_docs, (const v4i *)w.apack, qs, qb, w.tau,
                               DENSE_CAP, doc_base, w.buf_doc, w.buf_score, w.buf_cnt, w.ovf, w.any_ovf);
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)qb), dim3(THREADS), 0, stream, w.buf_score, (int64_t)DENSE_CAP,
                               n_docs, qb, k, 1, doc_base, 1, w.buf_doc, w.buf_cnt, DENSE_CAP, w.ovf, -1, no_gate, w.cand_doc,
                               w.cand_score, w.cand_count, (unsigned *)nullptr);
            HIP_TRY(hipGetLastError());
            int rc = merge_impl(device, w.cand_doc, w.cand_score, w.cand_count, qb, 1, k, 0, (int64_t)k, (int64_t)1, od, os, oc,
                                (int64_t)k, (int64_t)1, nullptr, 0, stream_v);
            if (rc != SRX_OK) return rc;
            // ---- fallback for queries whose buffer overflowed (degenerate score distributions): through the score
            //      matrix; both kernels return at once unless the any-overflow flag is set ----
            SRX_DENSE_DISPATCH(srx_dense_i8_scores_kernel, blocks_for(n_docs), corpus, corpus_scale, n_docs, (const v4i *)w.apack, qs, qb, w.scores, ld,
                               (const int *)w.any_ovf);
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)((int64_t)qb * ns)), dim3(THREADS), 0, stream, w.scores, ld, n_docs,
                               qb, k, ns, doc_base, 0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)w.ovf, 1,
                               (const int *)w.any_ovf, w.cand_doc, w.cand_score, w.cand_count, (unsigned *)nullptr);
            HIP_TRY(hipGetLastError());
            rc = merge_impl(device, w.cand_doc, w.cand_score, w.cand_count, qb, ns, k, 0, (int64_t)k, (int64_t)1, od, os, oc,
                            (int64_t)k, (int64_t)1, nullptr, 0, stream_v, (const int *)w.any_ovf);
            if (rc != SRX_OK) return rc;
        } else {
            SRX_DENSE_DISPATCH(srx_dense_i8_scores_kernel, blocks_for(n_docs), corpus, corpus_scale, n_docs, (const v4i *)w.apack, qs, qb, w.scores, ld, no_gate);
            if (!ks_ok) break;
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)((int64_t)qb * ns)), dim3(THREADS), 0, stream, w.scores, ld, n_docs,
                               qb, k, ns, doc_base, 0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, 0, no_gate,
                               w.cand_doc, w.cand_score, w.cand_count, (unsigned *)nullptr);
            HIP_TRY(hipGetLastError());
            const int rc = merge_impl(device, w.cand_doc, w.cand_score, w.cand_count, qb, ns, k, 0, (int64_t)k, (int64_t)1, od, os, oc,
                                      (int64_t)k, (int64_t)1, nullptr, 0, stream_v);
            if (rc != SRX_OK) return rc;
        }
    }
#undef SRX_DENSE_DISPATCH
    if (!ks_ok) return fail(SRX_ERR_INVALID, "srx_dense_search_i8: dim must be 32, 64, 96, 128, 192, 256, 384, 512, 768 or 1024 (pad the rows with zeros)%s");
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------------
// Dense f32 side: RetrievalService.search_by_vector (rag_system/core/retrieval.py:402-436):
// similarities = np.dot(embedding_index, query_vector), then the same top-k.  A matvec per query is HBM-bound (the
// embedding matrix streams once per pass of up to 4 queries): one wave per doc row, the lane's slices of the queries
// in registers, products summed in ascending column order per lane, then a fixed butterfly across lanes.  The BLAS
// summation order of the reference is unspecified, so parity is to 1e-4 relative (north_star), not bit-exact.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int F32_QP = 4;    // queries per pass
constexpr int F32_MAXS = 16; // dim <= 1024 = 16 slices of 64

__global__ __launch_bounds__(THREADS) void srx_dense_f32_scores_kernel(const float *__restrict__ emb, int64_t n_docs, int dim,
                                                                       const float *__restrict__ queries, int nqp,
                                                                       float *__restrict__ scores, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int ns = dim >> 6;  // slices of 64 columns (dim is a multiple of 64)
    float qv[F32_QP][F32_MAXS];
#pragma unroll
    for (int q = 0; q < F32_QP; ++q)
#pragma unroll
        for (int i = 0; i < F32_MAXS; ++i) qv[q][i] = (q < nqp && i < ns) ? queries[(int64_t)q * dim + lane + 64 * i] : 0.0f;
    const int64_t wave = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * WAVES;
    for (int64_t d = wave; d < n_docs; d += n_waves) {
        const float *row = emb + d * dim;
        float r[F32_MAXS];
#pragma unroll
        for (int i = 0; i < F32_MAXS; ++i) r[i] = i < ns ? row[lane + 64 * i] : 0.0f;
#pragma unroll
        for (int q = 0; q < F32_QP; ++q) {
            if (q < nqp) {  // uniform
                float a = 0.0f;
#pragma unroll
                for (int i = 0; i < F32_MAXS; ++i) a = a + r[i] * qv[q][i];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a = a + __shfl_xor(a, o);
                if (lane == 0) scores[(int64_t)q * ld + d] = a;
            }
        }
    }
}
}  // namespace

SRX_API int64_t srx_dense_f32_workspace_bytes(int32_t nq, int64_t n_docs, int32_t k) {
    if (nq < 0 || n_docs <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_dense_f32_workspace_bytes: bad argument%s");
    const int64_t ld = (n_docs + 63) / 64 * 64;
    const int ns = dense_splits(n_docs, F32_QP, k);
    return (int64_t)F32_QP * ld * 4 + (int64_t)F32_QP * ns * k * 8 + (int64_t)F32_QP * ns * 4 + 1024;
}

SRX_API int srx_dense_search_f32(int32_t device, const float *emb, int64_t n_docs, int32_t dim, const float *queries, int32_t nq,
                                 int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score, int32_t *out_count,
                                 void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (nq < 0 || n_docs <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "srx_dense_search_f32: need n_docs > 0, 1 <= k <= 1024%s");
    if (dim <= 0 || dim % 64 != 0 || dim > 64 * F32_MAXS)
        return fail(SRX_ERR_INVALID, "srx_dense_search_f32: dim must be a multiple of 64, <= 1024 (pad the rows with zeros)%s");
    if (doc_base < 0 || doc_base + n_docs >= 0x7FFFFFFFll) return fail(SRX_ERR_INVALID, "srx_dense_search_f32: doc_base + n_docs must fit int32%s");
    if (nq == 0) return SRX_OK;
    if (!emb || !queries || !out_doc || !out_score || !out_count) return fail(SRX_ERR_INVALID, "srx_dense_search_f32: null pointer%s");
    const int64_t need = srx_dense_f32_workspace_bytes(nq, n_docs, k);
    if (!workspace || workspace_bytes < need) return fail(SRX_ERR_NOMEM, "srx_dense_search_f32: workspace too small%s");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_v;
    const int64_t ld = (n_docs + 63) / 64 * 64;
    const int ns = dense_splits(n_docs, F32_QP, k);
    float *scores = (float *)workspace;
    int32_t *cand_doc = (int32_t *)(scores + (int64_t)F32_QP * ld);
    float *cand_score = (float *)(cand_doc + (int64_t)F32_QP * ns * k);
    int32_t *cand_count = (int32_t *)(cand_score + (int64_t)F32_QP * ns * k);
    int64_t blocks = (n_docs + WAVES - 1) / WAVES;
    if (blocks > 256 * 16) blocks = 256 * 16;
    for (int q0 = 0; q0 < nq; q0 += F32_QP) {
        const int qb = nq - q0 < F32_QP ? nq - q0 : F32_QP;
        hipLaunchKernelGGL(srx_dense_f32_scores_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, stream, emb, n_docs, (int)dim,
                           queries + (int64_t)q0 * dim, qb, scores, ld);
        hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)((int64_t)qb * ns)), dim3(THREADS), 0, stream, scores, ld, n_docs, qb, k,
                           ns, doc_base, 0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, 0,
                           (const int *)nullptr, cand_doc, cand_score, cand_count, (unsigned *)nullptr);
        HIP_TRY(hipGetLastError());
        const int rc = merge_impl(device, cand_doc, cand_score, cand_count, qb, ns, k, 0, (int64_t)k, (int64_t)1,
                                  out_doc + (int64_t)q0 * k, out_score + (int64_t)q0 * k, out_count + q0, (int64_t)k, (int64_t)1,
                                  nullptr, 0, stream_v);
        if (rc != SRX_OK) return rc;
    }
    return SRX_OK;
}

#ifdef SRX_STAMP
// Diagnostic build only: cumulative s_memtime ticks per kernel segment (see STAMP in srx_wave_kernel); resets.
extern "C" __attribute__((visibility("default"))) int srx_debug_read_stamps(unsigned long long *h_out16) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)));
    return 0;
}
#endif
