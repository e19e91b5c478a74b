// srx_common.h -- shared device primitives of libsparse_rx.so (gfx950): constants, the index view, wave / block
// reductions, the exact radix selections and the running top-k lists used by the scoring, merge and dense kernels.
// Everything here has internal linkage (anonymous namespace): each translation unit (sparse_rx.hip, wave_kernel.hip,
// dense.hip) compiles its own copy; only the few host functions declared at the end cross units.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (no fused multiply-add: the reference's arithmetic is
// separate fp32 multiply / add / IEEE divide).
#pragma once
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
__device__ unsigned long long g_stamp[32];
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

extern thread_local char srx_g_err[512];
#define g_err srx_g_err

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
#ifndef SRX_W_WPE
#define SRX_W_WPE 4
#endif
constexpr int W_WAVES_PER_EU = SRX_W_WPE;   // tier-1 waves per SIMD the kernel is compiled for (its register budget).  The kernel needs 94 VGPRs
                                            // and 8.25 KB of LDS per wave, so 19 waves are resident per CU.  Built for 5 (192-entry list, 7.75 KB:
                                            // 20 waves per CU) it measured 1 % slower on C3 and 4 % slower on a 1.25 M-doc shard (more selections)
constexpr int W_UNIT_MAX_DOCS = 49152;      // a unit covers <= 49152 docs (3 tiles of 16384): its local doc ids are the bit positions of
                                            // the wave-private LDS bitmap ...
constexpr int W_SENT_BASE = 49152;          // ... and the sentinels of the compact copy take the 64 bitmap words above them: local id
                                            // W_SENT_BASE + 32 j, j < 64 (a word of their own each)
constexpr int W_BM_WORDS = (W_SENT_BASE + 64 * 32) / 32;  // 1600 words = 6.25 KiB
constexpr unsigned W_BM_ADR_MASK = 0x1FFCu; // byte offset of a bitmap word from a 16-bit id >> 3 (ids stay below 51200 by construction)
#ifndef SRX_W_R
#define SRX_W_R 12
#endif
constexpr int W_R = SRX_W_R;                // postings per lane per unit held in registers (8 or 12)
constexpr int W_CAP = W_R * 64;             // hence <= 768 postings per tier-1 unit
constexpr int W_DUPCAP = 48;                // dup postings per unit resolved in tier 1 (more: the unit is dense -> tier 2)
constexpr int W_LCAP = 384;                 // lazy top-k list capacity of the merge wave kernel's callers (entries; a multiple of 64)
constexpr int W_KMAX = 128;                 // largest k ranked by one wavefront (wave_rank_emit: 2 keys per lane)
#ifndef SRX_W_LCAP
#define SRX_W_LCAP (SRX_W_WPE >= 5 ? 192 : 256)
#endif
constexpr int W1_LCAP = SRX_W_LCAP;         // tier 1's lazy top-k list capacity (entries; a multiple of 64)
constexpr int W1_KMAX = W1_LCAP - W_DUPCAP - 32 < 112 ? W1_LCAP - W_DUPCAP - 32 : 112;  // largest k tier 1 serves: next to k entries the
                                            // list keeps room for a unit's multi-term docs and >= 32 single-term candidates
constexpr int W_MAXT = 64;                  // query terms (each owns 64 / 2^ceil(log2 nt) lanes)

inline int fail(int code, const char *fmt, const char *detail = "") {
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

}  // namespace

// Posting storage (layout v2, "blocked"): the postings of a term are cut into runs, one per UNIT of
// unit_tiles * 2^tile_log2 <= 49152 consecutive docs; every run is padded to a multiple of 4 postings with sentinels
// (negative doc, value 0) and stored as blocks of 4 postings, docs and values of a block side by side:
//     f32 values: [d0 d1 d2 d3 | v0 v1 v2 v3]           8 words = 32 bytes
//     f16 values: [d0 d1 d2 d3 | h0 h1 | h2 h3]          6 words = 24 bytes
// so a lane's two loads of one block are adjacent (a lane group reads one contiguous piece per step: measured
// +9 % on the load side against separate doc / value arrays, tools/stream_microbench.hip) and a posting whose value is
// exactly 0 is a no-op for every kernel (it adds +-0 to a sum and can never be a result: only scores > 0 are), which
// is what lets the tier-1 kernel run without per-posting validity predicates.  Positions below are PADDED posting
// positions (position p lives in block p >> 2, slot p & 3).
template <typename VT>
struct BlockWords {
    static constexpr int value = 8;
};
template <>
struct BlockWords<__half> {
    static constexpr int value = 6;
};

// Compact copy for tier 1 (srx_build_compact): block b of `post` re-encoded with 16-bit UNIT-LOCAL doc ids,
//     f32 values: [l0 | l1 << 16][l2 | l3 << 16][v0 v1 v2 v3]   6 words = 24 bytes  (8 -> 6 bytes per posting)
//     f16 values: [l0 | l1 << 16][l2 | l3 << 16][h0 h1][h2 h3]  4 words = 16 bytes  (6 -> 4 bytes per posting)
// local id = doc - unit_first_doc (units of unit_tiles << tile_log2 <= W_UNIT_MAX_DOCS docs); a sentinel (doc -1 - 32 x)
// becomes W_SENT_BASE + 32 (x mod 64): above every real local id, a bitmap word of its own, value 0 as before.
template <typename VT>
struct CompactWords {
    static constexpr int value = 6;
};
template <>
struct CompactWords<__half> {
    static constexpr int value = 4;
};

struct IndexView {
    const int64_t *term_ptr;   // [vocab+1] padded position of the term's first posting (a multiple of 4)
    const int32_t *post;       // the blocks
    const int32_t *post16;     // the compact copy tier 1 streams (same block indices); nullptr: tier 2 serves every query
    const int32_t *tile_skip;  // [vocab*(n_tiles+1)] padded postings of term t before tile j, relative to term_ptr[t]
                               // (a multiple of 4 wherever j is a multiple of unit_tiles)
    const float *idf;
    const float *term_bound;  // optional [vocab*4]: K-th largest stored value per term for K = 1, 10, 100, 1000
    int64_t n_docs;
    int64_t vocab;
    int64_t zero_block;       // index of the first of SRX_BLOCK_PAD all-sentinel blocks (lane j redirects idle loads to block + j)
    int32_t tile_log2;
    int32_t n_tiles;
    int32_t unit_tiles;       // tiles per unit the runs are padded for
};

namespace {

// Index arrays are device (global) memory.  Pointers that reach a function through the IndexView reference are generic
// to the compiler, and a generic load is a FLAT instruction: it counts against lgkmcnt as well as vmcnt, so every LDS wait
// also drains the posting loads in flight -- which serialised tier 2's prefetch rings behind its LDS round trips.  The
// round trip through address space 1 makes the loads global_load (vmcnt only).
#define SRX_GLOBAL __attribute__((address_space(1)))
typedef int srx_i4u __attribute__((ext_vector_type(4), aligned(4)));  // 4 / 2 words at 4-byte alignment
typedef int srx_i2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ int gload_i32(const int32_t *p) { return *(const SRX_GLOBAL int32_t *)p; }
__device__ __forceinline__ srx_i4u gload_i4(const int32_t *p) { return *(const SRX_GLOBAL srx_i4u *)p; }
__device__ __forceinline__ srx_i2u gload_i2(const int32_t *p) { return *(const SRX_GLOBAL srx_i2u *)p; }

// Loads through the CONSTANT address space: with a wave-uniform address the compiler emits a scalar load (s_load), which
// travels through the scalar data cache and not through the CU's vector memory pipeline.  Only for data nothing in the
// kernel writes (index arrays, the query batch).
#define SRX_CONSTANT __attribute__((address_space(4)))
__device__ __forceinline__ int cload_i32(const int32_t *p) { return *(const SRX_CONSTANT int32_t *)p; }
__device__ __forceinline__ float cload_f32(const float *p) { return *(const SRX_CONSTANT float *)p; }
__device__ __forceinline__ int64_t cload_i64(const int64_t *p) { return *(const SRX_CONSTANT int64_t *)p; }

// one posting by padded position (scalar access: tier 2's hash / flat paths)
template <typename VT>
__device__ __forceinline__ int post_doc_at(const int32_t *post, int64_t p) {
    return gload_i32(post + (p >> 2) * BlockWords<VT>::value + (p & 3));
}
__device__ __forceinline__ float post_val_at(const int32_t *post, int64_t p, float) {
    return __int_as_float(gload_i32(post + (p >> 2) * 8 + 4 + (p & 3)));
}
__device__ __forceinline__ float post_val_at(const int32_t *post, int64_t p, __half) {
    const unsigned w = (unsigned)gload_i32(post + (p >> 2) * 6 + 4 + ((p & 3) >> 1));  // two halves per word
    return __half2float(__ushort_as_half((unsigned short)((p & 1) ? (w >> 16) : (w & 0xFFFFu))));
}

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


struct __attribute__((packed, aligned(4))) PackI4 {
    int x, y, z, w;
};
struct __attribute__((packed, aligned(4))) PackI2 {
    int x, y;
};
// one block: 4 docs + 4 values (as floats).  `blk` points at the block's first word.
__device__ __forceinline__ void load_block(const int32_t *blk, float, int (&d)[4], float (&v)[4]) {
    const srx_i4u a = gload_i4(blk), b = gload_i4(blk + 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    v[0] = __int_as_float(b.x); v[1] = __int_as_float(b.y); v[2] = __int_as_float(b.z); v[3] = __int_as_float(b.w);
}
__device__ __forceinline__ void load_block(const int32_t *blk, __half, int (&d)[4], float (&v)[4]) {
    const srx_i4u a = gload_i4(blk);
    const srx_i2u b = gload_i2(blk + 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    const int bx = b.x, by = b.y;
    const __half2 h0 = *reinterpret_cast<const __half2 *>(&bx), h1 = *reinterpret_cast<const __half2 *>(&by);
    v[0] = __low2float(h0); v[1] = __high2float(h0); v[2] = __low2float(h1); v[3] = __high2float(h1);
}

// The one rule both tiers apply: tier 1 (wave_kernel.hip) serves a query of nt > 0 terms iff this is false; tier 2
// (sparse_rx.hip) then takes the whole query instead of the flagged units only.
__device__ __forceinline__ bool tier1_cannot_serve(const IndexView &ix, int nt, int k, int tpu, int dbg) {
    return nt > W_MAXT || k > W1_KMAX || (tpu << ix.tile_log2) > W_UNIT_MAX_DOCS || ix.post16 == nullptr || (dbg & 8) != 0 ||
           ix.vocab * (int64_t)(ix.n_tiles + 1) >= (1ll << 30);  // tier 1 addresses the skip table with 32-bit byte offsets
}

// one block of the compact copy: 4 unit-local docs + 4 values
__device__ __forceinline__ void load_block16(const int32_t *blk, float, int (&d)[4], float (&v)[4]) {
    const srx_i2u a = gload_i2(blk);
    const srx_i4u b = gload_i4(blk + 2);
    d[0] = (int)((unsigned)a.x & 0xFFFFu); d[1] = (int)((unsigned)a.x >> 16);
    d[2] = (int)((unsigned)a.y & 0xFFFFu); d[3] = (int)((unsigned)a.y >> 16);
    v[0] = __int_as_float(b.x); v[1] = __int_as_float(b.y); v[2] = __int_as_float(b.z); v[3] = __int_as_float(b.w);
}
__device__ __forceinline__ void load_block16(const int32_t *blk, __half, int (&d)[4], float (&v)[4]) {
    const srx_i4u a = gload_i4(blk);
    d[0] = (int)((unsigned)a.x & 0xFFFFu); d[1] = (int)((unsigned)a.x >> 16);
    d[2] = (int)((unsigned)a.y & 0xFFFFu); d[3] = (int)((unsigned)a.y >> 16);
    const int bx = a.z, by = a.w;
    const __half2 h0 = *reinterpret_cast<const __half2 *>(&bx), h1 = *reinterpret_cast<const __half2 *>(&by);
    v[0] = __low2float(h0); v[1] = __high2float(h0); v[2] = __low2float(h1); v[3] = __high2float(h1);
}

// Tier 2 on the compact copy (an index that dropped its canonical blocks): the same accessors with the unit's first doc
// added to the 16-bit local id; a sentinel (local id >= W_SENT_BASE) reads as doc -1 like a canonical one.
template <typename VT>
__device__ __forceinline__ int post16_doc_at(const int32_t *p16, int64_t p, int ubase) {
    const unsigned w = (unsigned)gload_i32(p16 + (p >> 2) * CompactWords<VT>::value + ((p & 3) >> 1));
    const unsigned l = (p & 1) ? (w >> 16) : (w & 0xFFFFu);
    return l >= (unsigned)W_SENT_BASE ? -1 : ubase + (int)l;
}
__device__ __forceinline__ float post16_val_at(const int32_t *p16, int64_t p, float) {
    return __int_as_float(gload_i32(p16 + (p >> 2) * 6 + 2 + (p & 3)));
}
__device__ __forceinline__ float post16_val_at(const int32_t *p16, int64_t p, __half) {
    const unsigned w = (unsigned)gload_i32(p16 + (p >> 2) * 4 + 2 + ((p & 3) >> 1));
    return __half2float(__ushort_as_half((unsigned short)((p & 1) ? (w >> 16) : (w & 0xFFFFu))));
}
// the same block with the two id words left packed (tier 1 keeps them that way in registers)
__device__ __forceinline__ void load_block16p(const int32_t *blk, float, unsigned (&d)[2], float (&v)[4]) {
    const srx_i2u a = gload_i2(blk);
    const srx_i4u b = gload_i4(blk + 2);
    d[0] = (unsigned)a.x; d[1] = (unsigned)a.y;
    v[0] = __int_as_float(b.x); v[1] = __int_as_float(b.y); v[2] = __int_as_float(b.z); v[3] = __int_as_float(b.w);
}
__device__ __forceinline__ void load_block16p(const int32_t *blk, __half, unsigned (&d)[2], float (&v)[4]) {
    const srx_i4u a = gload_i4(blk);
    d[0] = (unsigned)a.x; d[1] = (unsigned)a.y;
    const int bx = a.z, by = a.w;
    const __half2 h0 = *reinterpret_cast<const __half2 *>(&bx), h1 = *reinterpret_cast<const __half2 *>(&by);
    v[0] = __low2float(h0); v[1] = __high2float(h0); v[2] = __low2float(h1); v[3] = __high2float(h1);
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

// ================================================================================================
// Tier 1: one wavefront per (query, split).  Wave-synchronous: no s_barrier anywhere; LDS executes one
// wave's DS instructions in order, wsync() only stops the compiler from reordering across the hand-off.
// ================================================================================================
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// number of set bits of the wave mask m below this lane (v_mbcnt_lo / v_mbcnt_hi: no 64-bit lane-mask constant to keep in
// registers -- the compiler hoisted (1 << lane) - 1 out of the unit loop and spilled it to scratch)
__device__ __forceinline__ unsigned lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uniu(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

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
            const unsigned p = base + lane_rank(m);
            S.lbits[p] = x;
            S.ldoc[p] = dd;
        }
        base += (unsigned)__popcll(m);
        wsync();
    }
    if constexpr (SH::HIST_ALIASES_ZEROED_LDS) {  // the histogram borrowed words that must read as zero again (tier 1's doc bitmap)
        reinterpret_cast<uint4 *>(S.hist)[lane] = make_uint4(0u, 0u, 0u, 0u);
        wsync();
    }
    return T;
}

struct WaveTopk {
    unsigned count, tau;
};

// Append candidates (one per lane at most) to the wave's lazy list, shrinking it first when it is nearly full.
template <typename SH>
__device__ __forceinline__ void wave_append(SH &S, WaveTopk &tk, int k, bool cand, unsigned bits, int doc) {
    const int lane = threadIdx.x;
    const unsigned long long m = __ballot(cand);
    if (m != 0ull) {  // uniform
        if (tk.count > (unsigned)(SH::LCAP - 64)) {  // make room for up to 64 more entries
            tk.tau = uniu(wave_list_select(S, tk.count, k));
            tk.count = (unsigned)k;
        }
        const bool c2 = cand && bits >= tk.tau;  // tau may just have risen
        const unsigned long long m2 = __ballot(c2);
        if (c2) {
            const unsigned p = tk.count + lane_rank(m2);
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

struct MergeShared {
    TopkShared tk;
    unsigned hist[RADIX_BINS];
    unsigned long long sortkey[KMAX];
    int lstart[64];
};

}  // namespace

// ---- host functions shared between the translation units (hidden visibility: not part of the C ABI) ----
struct srx_wave_launch {
    IndexView ix;
    const int32_t *q_ptr, *q_term;
    const float *q_weight;
    int nq, k, n_splits, n_whole, n_super, dbg;
    unsigned *ovf;       // [work items][ovf_words]: units item i left to tier 2 (bit su)
    int ovf_words, lists_per_q;
    int *work;
    unsigned *done;      // [nq - n_whole], zeroed before the launch: arrivals of a split query's items (see the kernel's tail)
    int32_t *cand_doc;
    float *cand_score;
    int32_t *cand_count;
    int64_t doc_base;
    int32_t *out_doc;
    float *out_score;
    int32_t *out_count;
    int64_t out_row_stride, out_cnt_stride;
};
// tier 1 (wave_kernel.hip): one wavefront per (query, split); `blocks` work items
int srx_launch_wave_kernel(const srx_wave_launch &a, int val_type, int64_t blocks, hipStream_t stream);
// exact merge of candidate lists (sparse_rx.hip), also used by the dense side
int srx_merge_impl(int32_t device, const int32_t *in_doc, const float *in_score, const int32_t *in_count, int32_t nq,
                   int32_t n_lists, int32_t k, int lay, int64_t row_stride, int64_t cnt_stride, int32_t *out_doc,
                   float *out_score, int32_t *out_count, int64_t ors, int64_t ocs, void *workspace, int64_t workspace_bytes,
                   void *stream_v, const int *gate = nullptr);
