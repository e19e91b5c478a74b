// dense.hip -- the dense side of the same service: INT8 MFMA GEMM + fused top-k filter, and the f32
// streaming matvec of search_by_vector.  Shares the block top-k machinery and the merge kernels (srx_common.h,
// sparse_rx.hip).

#include "srx_common.h"

// The integer dot products are one MFMA GEMM (v_mfma_i32_32x32x32_i8, exact); the two scalings are done in fp64 like
// the reference's NumPy scalars (int32 * float32 -> float64), so the stored fp32 score is the reference's bit for bit.
// First form: the scaled scores go through HBM once (a [query batch][n_docs] fp32 matrix in the workspace) and the
// block top-k machinery of the sparse path ranks each row; only scores > 0 are results (retriever_registry.py:519).
// ================================================================================================
namespace {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#ifndef SRX_DENSE_NW_LONG
#define SRX_DENSE_NW_LONG 4  // waves per workgroup of the filter kernel on rows longer than 384 bytes (8: measured, no gain)
#endif
#ifndef SRX_DENSE_PF
#define SRX_DENSE_PF 6  // LDS reads of query fragments in flight ahead of the MFMAs
#endif
constexpr int DENSE_CNT_STRIDE = 32;  // ints between two queries' candidate counters: one 128-byte line each (all waves
                                      // add to these: counters sharing a line serialise in one L2 channel)

// Query batch in MFMA-fragment order: apack[(tile * KS + s) * 64 + lane] = the 16 bytes lane `lane` feeds into k-step s
// of query tile `tile` (row tile * 32 + (lane & 31), columns 32 s + 16 (lane >> 5) ..).  A wave's A load is then one
// contiguous 1 KiB block instead of 32 scattered 32-byte segments (the request rate of the texture path was the limit).
// The corpus can be kept in the same order (srx_dense_pack_i8: tile = 32 docs): a wave's B fragments are then KS contiguous
// 1 KiB loads instead of KS loads that each touch 32 rows.
__global__ __launch_bounds__(THREADS) void srx_dense_pack_queries_kernel(const int8_t *__restrict__ queries, int64_t nq, int dim,
                                                                         v4i *__restrict__ apack) {
    const int ks = dim / 32;
    const int64_t n = ((nq + 31) / 32) * ks * 64;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const int lane = (int)(i & 63);
        const int64_t ts = i >> 6;
        const int s = (int)(ts % ks);
        const int64_t q = (ts / ks) * 32 + (lane & 31);
        v4i x = {0, 0, 0, 0};
        if (q < nq) x = *reinterpret_cast<const v4i *>(queries + q * dim + s * 32 + 16 * (lane >> 5));
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
                                                                       const int *__restrict__ gate, int packed) {
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
        if (packed)  // uniform: fragment order, rows past the corpus are zeros there
            B[s] = reinterpret_cast<const v4i *>(corpus)[((d0 >> 5) * KS + s) * 64 + lane];
        else if (dok)
            B[s] = *reinterpret_cast<const v4i *>(corpus + d * DIM + s * 32 + 16 * h);
    }
    const double ds = dok ? (double)corpus_scale[d] : 0.0;
    // gridDim.y splits the query tiles (the threshold sample is a few hundred workgroups of docs only: the split fills the chip)
    const int n_qt = (nq + 31) / 32, qt_per = (n_qt + (int)gridDim.y - 1) / (int)gridDim.y;
    const int q_lo = (int)blockIdx.y * qt_per * 32, q_hi = min(nq, q_lo + qt_per * 32);
    for (int q0 = q_lo; q0 < q_hi; q0 += 32) {
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
// the corpus) and appends (doc, score) to the query's candidate buffer.  A full buffer raises the query's overflow flag
// (the caller then re-ranks that query through the score matrix).
//
// Round 3 (the kernel took 1.8 ms per 1 M x 1 024 batch whatever the row length: it waited for one returning global
// atomic round trip per 32 x 32 tile at two waves per SIMD, and scaled every one of the 10^9 dot products in fp64):
//  * the query tile's A fragments (KS KiB) are staged through LDS ONCE PER WORKGROUP, two or three buffers, by direct
//    global -> LDS loads (every wave used to read them from L2 itself); a table {screen bound, threshold, scale} of all
//    the pass's queries is built in LDS once per workgroup;
//  * an fp32 screen decides which accumulator rows need the exact arithmetic at all: with a = fl32(acc) * ds (one rounding;
//    |acc| < 2^24 is exact in fp32), a score can reach tau only if a >= tau / qs * (1 - 2^-23)(1 - 2^-24) -- the exact chain
//    fl32(fl64(fl64(acc * qs) * ds)) >= tau needs acc * qs * ds >= tau (1 - 2^-24)(1 - 2^-52)^2 -- and the table holds
//    fl32(fl32(tau / qs) * (1 - 2^-20)), which is below that (rows whose quotient is not a normal positive number, and
//    qs <= 0, are not screened).  3 fp32 instructions per accumulator register; the fp64 chain runs for the registers in
//    which some lane passes (one in ten in the second filter round) and decides alone;
//  * survivors go to a per-wave LDS list ((query, doc-in-wave), score: 8 bytes) -- wave ballot + mbcnt, no atomics -- and
//    the list is flushed to the queries' global buffers (one returning atomicAdd per entry, 64 in flight) when half full
//    and at the end: one global round trip per few hundred survivors instead of one per tile.
struct DenseTab {    // per query of the pass (<= 1 024)
    float thr[1024];  // the screen's bound on fl32(acc) * ds; -inf = not screened, +inf = no such query
    unsigned tau[1024];
    float qs[1024];
};
// In-kernel stamps of the filter kernel (diagnostic build -DSRX_DSTAMP, tools/dense_stamp_run.py): s_memtime ticks per phase,
// summed over waves.
#ifdef SRX_DSTAMP
__device__ unsigned long long g_dstamp[16];
#define DSTAMP(i)                                                                        \
    do {                                                                                 \
        unsigned long long t_;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                               \
        dst_acc[i] += t_ - dst_prev;                                                     \
        dst_prev = t_;                                                                   \
    } while (0)
#else
#define DSTAMP(i) \
    do {          \
    } while (0)
#endif
// ds_read_b128 with a compile-time offset, issued where it stands (the scheduler sinks plain LDS reads down to one MFMA before
// their use); the consumer waits with lds_wait<N>, which also ties the value to the wait
template <int OFF>
__device__ __forceinline__ void lds_read128(v4i &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(v4i &x) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N));
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}
// (Measured and dropped, profiles/r03_dense_filter_variants.log: eight waves per workgroup on long rows -- half the L2 -> LDS
// traffic of the query fragments, but one workgroup per CU: its prologue and final flush are dead time -- 3-5 % slower; a
// ping-pong form of that workgroup -- waves 0-3 and 4-7 half a tile apart, one group in its MFMAs while its SIMD partners
// screen the tile before, a barrier per phase -- 12-30 % slower: the screening phase (staging issue + screen + wait) is
// 1.7 x the MFMA phase, and a lock-step pair runs at the pace of the longer one.)
template <int KS>
constexpr int dense_filter_waves() { return KS > 12 ? SRX_DENSE_NW_LONG : 4; }

template <int KS>
__global__ __launch_bounds__(64 * dense_filter_waves<KS>()) __attribute__((amdgpu_waves_per_eu(2))) void srx_dense_i8_filter_kernel(const int8_t *__restrict__ corpus,
                                                                       const float *__restrict__ corpus_scale,
                                                                       int64_t n_docs, const v4i *__restrict__ apack,
                                                                       const float *__restrict__ query_scale, int nq,
                                                                       const unsigned *__restrict__ tau, int cap,
                                                                       int64_t doc_base, int32_t *__restrict__ buf_doc,
                                                                       float *__restrict__ buf_score,
                                                                       int *__restrict__ buf_cnt, int *__restrict__ ovf,
                                                                       int *__restrict__ any_ovf, int packed) {
    // DT doc tiles of 32 per wave: with two, every A fragment (query tile) feeds two MFMAs; the B fragments of both
    // tiles must fit the register file (KS <= 12, i.e. rows up to 384 bytes).
    constexpr int DT = KS <= 12 ? 2 : 1;
    constexpr int NW = dense_filter_waves<KS>();  // waves per workgroup
    // LDS: two free-running workgroups of four waves per CU must fit (80 KiB each); one of eight waves may take more
    constexpr int DENSE_CB = KS <= 12 ? 512 : (KS <= 24 ? 256 : 128);  // per-wave survivor list entries
    constexpr int NBUF = (KS <= 16 || NW == 8) ? 3 : 2;  // query tiles in LDS: the one in use and one or two on their way
    constexpr int PD = NBUF - 1;                         // tiles staged ahead
    constexpr int ROWS = (KS + NW - 1) / NW;             // 1 KiB fragment rows of a tile that one wave stages
    __shared__ v4i ldsA[NBUF][KS * 64];
    __shared__ DenseTab tab;
    __shared__ uint2 cb[NW][DENSE_CB];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // in a scalar register: the staging addresses are then scalar
    const int64_t d0 = ((int64_t)blockIdx.x * NW + wv) * (32 * DT);  // may lie past the corpus: the wave then only helps
    constexpr int DIM = KS * 32;                                      // staging and takes part in the barriers
    const int n_qt = (nq + 31) / 32;
#ifdef SRX_DSTAMP
    unsigned long long dst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dst_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst_prev)::"memory");
#endif
    // global -> LDS without a trip through registers (global_load_lds_dwordx4: lane l of a wave writes 16 bytes at the wave's
    // LDS base + 16 l): wave w copies the 64-fragment rows w, w + NW, ... of the tile -- EXACTLY ROWS loads per wave and tile
    // (a wave with a row too many re-copies the last row of its share: same bytes, same place), so that the wait for a tile can
    // be a counted one: the loads of the tile after it stay in flight across the barrier.
    // (Measured and dropped: staging through registers -- global_load_dwordx4 at the top of a tile, ds_write_b128 at its end.
    // In-kernel stamps put 100-200 cycles on the issue of each staging load either way; the register form was 3 % slower.)
    auto stage_row = [&](int tile, int buf, int s) __attribute__((always_inline)) {
        const v4i *src = apack + ((int64_t)tile * KS + s) * 64;  // scalar base + a 32-bit lane offset
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane),
                                         (__attribute__((address_space(3))) void *)&ldsA[buf][s * 64], 16, 0, 0);
    };
    auto stage_tile = [&](int tile, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int row = 0; row < ROWS; ++row) stage_row(tile, buf, min(wv + row * NW, KS - 1));
    };
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (i < n_qt) stage_tile(i, i);
    // every query's screen bound, threshold and scale, once per workgroup (12 KiB of LDS for 1 024 queries)
    for (int q = threadIdx.x; q < n_qt * 32; q += NW * 64) {
        float thr = __builtin_inff(), qsn = 0.0f;
        unsigned taun = 0xFFFFFFFFu;
        if (q < nq) {
            qsn = query_scale[q];
            taun = tau[q];
            thr = -__builtin_inff();
            if (qsn > 0.0f) {
                const float x = (__uint_as_float(taun) / qsn) * 0.99999905f;  // 1 - 2^-20
                if (x >= 1e-30f && x < 3e38f) thr = x;
            }
        }
        tab.thr[q] = thr;
        tab.qs[q] = qsn;
        tab.tau[q] = taun;
    }
    v4i B[DT][KS];
    float dsf[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int64_t d = d0 + 32 * t + r;
        const bool dok = d < n_docs;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            B[t][s] = (v4i){0, 0, 0, 0};
            if (packed) {  // uniform: fragment order (srx_dense_pack_i8), one contiguous KiB per load; rows past the corpus are zeros
                // (non-temporal: the corpus is read once per pass; the query fragments every workgroup re-reads should stay in L2.
                // 1-3 % on 4 M docs / 1 024-byte rows, nothing elsewhere)
                if (d0 + 32 * t < n_docs) B[t][s] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(corpus) + (((d0 >> 5) + t) * KS + s) * 64 + lane);
            } else if (dok) {
                B[t][s] = *reinterpret_cast<const v4i *>(corpus + d * DIM + s * 32 + 16 * h);
            }
        }
        dsf[t] = dok ? corpus_scale[d] : 0.0f;  // 0: every score of a row past the corpus is 0 or NaN, never > 0
    }
    int cnt = 0;         // entries in my wave's survivor list (wave-uniform)
    bool dirty = false;  // stores / atomics of mine may be in flight (wave-uniform): the next wait for a staged tile drains them
    auto flush = [&]() __attribute__((always_inline)) {
        for (int i = lane; i < cnt; i += 64) {
            const uint2 e = cb[wv][i];
            const int q = (int)(e.x >> 6);
            const int p = atomicAdd(&buf_cnt[q * DENSE_CNT_STRIDE], 1);
            if (p < cap) {
                buf_doc[(int64_t)q * cap + p] = (int32_t)(doc_base + d0 + (int)(e.x & 63u));
                buf_score[(int64_t)q * cap + p] = __uint_as_float(e.y);
            } else {
                ovf[q] = 1;
                *any_ovf = 1;
            }
        }
        cnt = 0;
        dirty = true;
    };
    v16i acc[DT];
    v4i thb[4];
    // The MFMAs of query tile `tile` (buffer tile % NBUF).  The tile's 32 screen bounds (rows 8 g + 4 h + 0..3 are accumulator
    // registers 4 g .. 4 g + 3) and the A fragments are LDS reads issued where they stand, PF of the latter in flight ahead of
    // the MFMA that consumes them (an LDS read returns after ~100+ cycles, an MFMA issues every 32; the compiler's schedule
    // reads one ahead and puts each group's bounds right before their use: four more LDS round trips per tile).  LDS reads
    // return in order, so before step s at most min(PF - 1, KS - 1 - s) younger reads may still be out.
    auto do_mfma = [&](int tile) __attribute__((always_inline)) {
        constexpr int PF = KS < SRX_DENSE_PF ? KS : SRX_DENSE_PF;
        v4i Ab[PF];
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned a_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)&ldsA[tile % NBUF][lane];
        const unsigned t_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)&tab.thr[tile * 32 + 4 * h];
        static_for<0, 4>([&](auto g) { lds_read128<decltype(g)::value * 32>(thb[decltype(g)::value], t_addr); });
        static_for<0, PF>([&](auto i) { lds_read128<decltype(i)::value * 1024>(Ab[decltype(i)::value], a_addr); });
        static_for<0, KS>([&](auto i) {
            constexpr int s = decltype(i)::value;
            lds_wait<(KS - 1 - s < PF - 1 ? KS - 1 - s : PF - 1)>(Ab[s % PF]);
#pragma unroll
            for (int t = 0; t < DT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ab[s % PF], B[t][s], acc[t], 0, 0, 0);
            if constexpr (s + PF < KS) lds_read128<(s + PF) * 1024>(Ab[s % PF], a_addr);
        });
    };
    // Screen + exact arithmetic + survivor list of the tile in acc / thb
    auto do_epilogue = [&](int tile) __attribute__((always_inline)) {
        const int q0 = tile * 32;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const double ds = (double)dsf[t];
            // all 16 screens first: independent instructions, one vector-to-scalar round trip and one branch for the whole tile
            // when nothing passes (a branch per group of four put four such round trips in a row).  "not below" instead of ">=":
            // a NaN (0 * inf, a NaN scale) goes to the exact arithmetic too
            unsigned long long mka[16];  // lane masks in scalar registers: v_cmp writes them, s_or combines them
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float thv[4] = {__int_as_float(thb[g].x), __int_as_float(thb[g].y), __int_as_float(thb[g].z), __int_as_float(thb[g].w)};
#pragma unroll
                for (int j = 0; j < 4; ++j) mka[4 * g + j] = __ballot(!((float)acc[t][4 * g + j] * dsf[t] < thv[j]));
            }
            unsigned long long many = 0ull;
#pragma unroll
            for (int i = 0; i < 16; ++i) many |= mka[i];
            if (many == 0ull) continue;  // uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float thv[4] = {__int_as_float(thb[g].x), __int_as_float(thb[g].y), __int_as_float(thb[g].z), __int_as_float(thb[g].w)};
                const unsigned long long mk[4] = {mka[4 * g], mka[4 * g + 1], mka[4 * g + 2], mka[4 * g + 3]};
                if ((mk[0] | mk[1] | mk[2] | mk[3]) == 0ull) continue;  // uniform: most groups of four end here
                // the rows' thresholds and scales, read once per group (a read inside every exact branch put two LDS round
                // trips into each of them, at two waves per SIMD)
                const uint4 ta = *reinterpret_cast<const uint4 *>(&tab.tau[q0 + 8 * g + 4 * h]);
                const float4 qa = *reinterpret_cast<const float4 *>(&tab.qs[q0 + 8 * g + 4 * h]);
                const unsigned tav[4] = {ta.x, ta.y, ta.z, ta.w};
                const float qsv[4] = {qa.x, qa.y, qa.z, qa.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int reg = 4 * g + j;
                    if (mk[j] != 0ull) {  // uniform
                        const int row = 8 * g + 4 * h + j;
                        const unsigned taur = tav[j];
                        const float sc = (float)(((double)acc[t][reg] * (double)qsv[j]) * ds);
                        const bool pass = !((float)acc[t][reg] * dsf[t] < thv[j]) && q0 + row < nq && sc > 0.0f && __float_as_uint(sc) >= taur;
                        const unsigned long long m = __ballot(pass);
                        if (m != 0ull) {  // uniform
                            const int n = __popcll(m);
                            if (cnt + n <= DENSE_CB) {
                                if (pass) cb[wv][cnt + (int)lane_rank(m)] = make_uint2(((unsigned)(q0 + row) << 6) | (unsigned)(32 * t + r), __float_as_uint(sc));
                                cnt += n;
                            } else {  // list full inside one tile (degenerate score distributions): straight to the buffer
                                dirty = true;
                                if (pass) {
                                    const int q = q0 + row;
                                    const int p = atomicAdd(&buf_cnt[q * DENSE_CNT_STRIDE], 1);
                                    if (p < cap) {
                                        buf_doc[(int64_t)q * cap + p] = (int32_t)(doc_base + d0 + 32 * t + r);
                                        buf_score[(int64_t)q * cap + p] = sc;
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
    };
    // A bare s_barrier: __syncthreads() puts s_waitcnt vmcnt(0) in front of it, which would wait for the staging loads that are
    // meant to stay in flight.  Every wave has waited (counted) for its own share of the tile that the barrier publishes.
    auto phase_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DSTAMP(0);  // prologue: B fragments, table, first tiles staged
    for (int qt = 0; qt < n_qt; ++qt) {
        const bool more = qt + PD < n_qt;                  // uniform
        if (more) stage_tile(qt + PD, (qt + PD) % NBUF);  // that buffer was last read in tile qt - 1: one barrier ago
        DSTAMP(1);  // stage issue
        do_mfma(qt);
        DSTAMP(2);  // MFMA loop (to the issue of the last one)
        do_epilogue(qt);
        DSTAMP(3);  // epilogue (waits for the accumulators first)
        if (cnt >= DENSE_CB / 2) flush();  // uniform
        // my share of tile qt + 1 has landed (with three buffers it was issued a whole tile ago, and the loads of tile qt + 2
        // stay in flight).  Loads return in order; stores / atomics of mine (a flush) may not, so after one everything is drained.
        if (PD > 1 && more && !dirty)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * ROWS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dirty = false;
        DSTAMP(4);  // flush, wait for the staged tile
        phase_barrier();  // every wave's share of tile qt + 1 is in LDS; tile qt's buffer is free
        DSTAMP(5);  // barrier
    }
    if (cnt > 0) flush();
    DSTAMP(6);  // final flush
#ifdef SRX_DSTAMP
    if (lane == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_dstamp[i], dst_acc[i]);
        atomicAdd(&g_dstamp[8], 1ull);
    }
#endif
}

// Row top-k: one workgroup per (query, split of the doc range) folds its slice of the score row into an exact lazy
// top-k list (topk_fold, the sparse path's machinery); the merge kernels rank the splits' lists.
constexpr int DENSE_NPT = 16;
// mode 0: rank scores[q][lo..hi) (ids = doc_base + column).  mode 1: rank the query's candidate buffer (buf_doc /
// scores hold (doc, score) pairs, buf_cnt[q] of them).  only_flag: +1 = only queries with ovf[q] != 0, -1 = only queries
// with ovf[q] == 0, 0 = all; a skipped query writes count -1 for its first list (the merge kernels then leave its
// output row alone).  tau_out (optional): the k-th best score's bits when the list holds k entries, else 0 (tau_keep: the
// larger of that and the value already there).
__global__ __launch_bounds__(THREADS) void srx_dense_topk_kernel(const float *__restrict__ scores, int64_t ld, int64_t n_docs,
                                                                 int nq, int k, int n_splits, int64_t doc_base, int mode,
                                                                 const int32_t *__restrict__ buf_doc,
                                                                 const int *__restrict__ buf_cnt, int cap,
                                                                 const int *__restrict__ ovf, int only_flag,
                                                                 const int *__restrict__ gate,
                                                                 int32_t *__restrict__ cand_doc,
                                                                 float *__restrict__ cand_score,
                                                                 int32_t *__restrict__ cand_count,
                                                                 unsigned *__restrict__ tau_out, int tau_keep) {
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
        if (tid == 0) tau_out[q] = max(tau_keep ? tau_out[q] : 0u, cnt >= (unsigned)k ? rr.mn : 0u);
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
// Sample size of the threshold pass: the k-th best score of S docs leaves about k * n_docs / S survivors per query in one
// filter round (aim: DENSE_CAP / 3), 2 k sqrt(n_docs / S) in two.  The sample goes through the score matrix, so it should
// be small: swept in profiles/r03_dense_filter_variants.log (k = 1 000: 2.51 ms at 8 k n / CAP, 2.19-2.22 ms at 2-4 k n / CAP).
// 0 = corpus too small for the filtered path to pay.
int64_t dense_sample(int64_t n_docs, int k) {
    if (n_docs < 65536) return 0;
    int64_t S = (3 * (int64_t)k * n_docs + DENSE_CAP - 1) / DENSE_CAP;
    // at least ~k / 0.6 % docs (what the second round's tighter threshold makes of it costs little), 4 096 .. 16 384: for small k
    // the sample pass itself is what counts (k = 10: 1.14 -> 1.11 ms at 4 096, sweep in profiles/r03_dense_filter_variants.log)
    const int64_t floor_s = 164ll * k < 4096 ? 4096 : (164ll * k > 16384 ? 16384 : 164ll * k);
    if (S < floor_s) S = floor_s;
#ifdef SRX_DENSE_KNOBS  // dev build: scale the sample (tools/r3_run33.sh)
    if (const char *e = getenv("SRX_DENSE_SAMPLE_MULT")) S = (int64_t)((double)S * atof(e));
    if (S < 2048) S = 2048;
#endif
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

namespace {
int dense_search_i8_impl(int32_t device, const int8_t *corpus, const float *corpus_scale, int64_t n_docs, int32_t dim,
                         const int8_t *queries, const float *query_scale, int32_t nq, int32_t k, int64_t doc_base,
                         int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace, int64_t workspace_bytes,
                         void *stream_v, int packed) {
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
#define SRX_DENSE_DISPATCH(KERNEL, GRID, BLOCK, ...)                                                                        \
    switch (dim / 32) {                                                                                               \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;                \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;                \
        case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;                \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;                \
        case 6: hipLaunchKernelGGL(KERNEL<6>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;                \
        case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;                \
        case 12: hipLaunchKernelGGL(KERNEL<12>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;              \
        case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;              \
        case 24: hipLaunchKernelGGL(KERNEL<24>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;              \
        case 32: hipLaunchKernelGGL(KERNEL<32>, dim3(GRID), dim3(BLOCK), 0, stream, __VA_ARGS__); break;              \
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
        hipLaunchKernelGGL(srx_dense_pack_queries_kernel, dim3(64), dim3(THREADS), 0, stream, qp, (int64_t)qb, (int)dim, w.apack);
        if (S > 0) {
            // ---- filtered path: threshold from a sample, GEMM with the filter fused in, rank the candidate buffers ----
            HIP_TRY(hipMemsetAsync(w.buf_cnt, 0, (size_t)(qbmax * DENSE_CNT_STRIDE + qbmax + 1) * 4, stream));
            SRX_DENSE_DISPATCH(srx_dense_i8_scores_kernel, dim3(blocks_for(S), (unsigned)((qb + 127) / 128)), THREADS, corpus, corpus_scale, S, (const v4i *)w.apack, qs, qb, w.scores, ld, no_gate, packed);
            if (!ks_ok) break;
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)qb), dim3(THREADS), 0, stream, w.scores, ld, S, qb, k, 1, doc_base,
                               0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, 0, no_gate, w.cand_doc,
                               w.cand_score, w.cand_count, w.tau, 0);
            // Two rounds (round 3): the sample's threshold lets about k n / S docs per query through; the first round filters only
            // the docs [0, S1), S1 = sqrt(S n), the k-th best of its survivors (a valid lower bound too, and never below the
            // sample's) is the threshold of the second round over [S1, n): about k (S1 / S + n / S1) survivors per query instead
            // of k n / S -- 4x fewer exact-path rows and buffer appends at 1 M docs.  Both rounds append to the same buffers.
            const int ks = dim / 32;
            const int filter_waves = ks > 12 ? SRX_DENSE_NW_LONG : 4;  // = dense_filter_waves<KS>()
            const int filter_threads = 64 * filter_waves;
            const int64_t docs_per_block = (ks <= 12 ? 64 : 32) * filter_waves;
            int64_t S1 = (int64_t)sqrt((double)S * (double)n_docs);
            const int64_t chip = docs_per_block * (2048 / filter_waves);  // whole rounds of the chip at eight waves per CU
            S1 = S1 >= chip ? (S1 + chip - 1) / chip * chip : (S1 + docs_per_block - 1) / docs_per_block * docs_per_block;
            if (S1 * 2 > n_docs) S1 = n_docs;  // small corpus: one round
            for (int round = 0; round < 2; ++round) {
                const int64_t lo = round == 0 ? 0 : S1, hi = round == 0 ? S1 : n_docs;
                if (lo >= hi) break;
                SRX_DENSE_DISPATCH(srx_dense_i8_filter_kernel, (unsigned)((hi - lo + docs_per_block - 1) / docs_per_block), filter_threads, corpus + lo * dim,
                                   corpus_scale + lo, hi - lo, (const v4i *)w.apack, qs, qb, w.tau, DENSE_CAP, doc_base + lo, w.buf_doc,
                                   w.buf_score, w.buf_cnt, w.ovf, w.any_ovf, packed);
                if (round == 0 && hi < n_docs)
                    hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)qb), dim3(THREADS), 0, stream, w.buf_score, (int64_t)DENSE_CAP,
                                       n_docs, qb, k, 1, doc_base, 1, w.buf_doc, w.buf_cnt, DENSE_CAP, (const int *)nullptr, 0, no_gate,
                                       w.cand_doc, w.cand_score, w.cand_count, w.tau, 1);
            }
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)qb), dim3(THREADS), 0, stream, w.buf_score, (int64_t)DENSE_CAP,
                               n_docs, qb, k, 1, doc_base, 1, w.buf_doc, w.buf_cnt, DENSE_CAP, w.ovf, -1, no_gate, w.cand_doc,
                               w.cand_score, w.cand_count, (unsigned *)nullptr, 0);
            HIP_TRY(hipGetLastError());
            int rc = srx_merge_impl(device, w.cand_doc, w.cand_score, w.cand_count, qb, 1, k, 0, (int64_t)k, (int64_t)1, od, os, oc,
                                (int64_t)k, (int64_t)1, nullptr, 0, stream_v);
            if (rc != SRX_OK) return rc;
            // ---- fallback for queries whose buffer overflowed (degenerate score distributions): through the score
            //      matrix; both kernels return at once unless the any-overflow flag is set ----
            SRX_DENSE_DISPATCH(srx_dense_i8_scores_kernel, blocks_for(n_docs), THREADS, corpus, corpus_scale, n_docs, (const v4i *)w.apack, qs, qb, w.scores, ld,
                               (const int *)w.any_ovf, packed);
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)((int64_t)qb * ns)), dim3(THREADS), 0, stream, w.scores, ld, n_docs,
                               qb, k, ns, doc_base, 0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)w.ovf, 1,
                               (const int *)w.any_ovf, w.cand_doc, w.cand_score, w.cand_count, (unsigned *)nullptr, 0);
            HIP_TRY(hipGetLastError());
            rc = srx_merge_impl(device, w.cand_doc, w.cand_score, w.cand_count, qb, ns, k, 0, (int64_t)k, (int64_t)1, od, os, oc,
                            (int64_t)k, (int64_t)1, nullptr, 0, stream_v, (const int *)w.any_ovf);
            if (rc != SRX_OK) return rc;
        } else {
            SRX_DENSE_DISPATCH(srx_dense_i8_scores_kernel, blocks_for(n_docs), THREADS, corpus, corpus_scale, n_docs, (const v4i *)w.apack, qs, qb, w.scores, ld, no_gate, packed);
            if (!ks_ok) break;
            hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)((int64_t)qb * ns)), dim3(THREADS), 0, stream, w.scores, ld, n_docs,
                               qb, k, ns, doc_base, 0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, 0, no_gate,
                               w.cand_doc, w.cand_score, w.cand_count, (unsigned *)nullptr, 0);
            HIP_TRY(hipGetLastError());
            const int rc = srx_merge_impl(device, w.cand_doc, w.cand_score, w.cand_count, qb, ns, k, 0, (int64_t)k, (int64_t)1, od, os, oc,
                                      (int64_t)k, (int64_t)1, nullptr, 0, stream_v);
            if (rc != SRX_OK) return rc;
        }
    }
#undef SRX_DENSE_DISPATCH
    if (!ks_ok) return fail(SRX_ERR_INVALID, "srx_dense_search_i8: dim must be 32, 64, 96, 128, 192, 256, 384, 512, 768 or 1024 (pad the rows with zeros)%s");
    return SRX_OK;
}
}  // namespace

SRX_API int srx_dense_search_i8(int32_t device, const int8_t *corpus, const float *corpus_scale, int64_t n_docs, int32_t dim,
                                const int8_t *queries, const float *query_scale, int32_t nq, int32_t k, int64_t doc_base,
                                int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                                int64_t workspace_bytes, void *stream_v) {
    return dense_search_i8_impl(device, corpus, corpus_scale, n_docs, dim, queries, query_scale, nq, k, doc_base, out_doc, out_score,
                                out_count, workspace, workspace_bytes, stream_v, 0);
}

SRX_API int srx_dense_search_i8_packed(int32_t device, const void *corpus_packed, const float *corpus_scale, int64_t n_docs, int32_t dim,
                                       const int8_t *queries, const float *query_scale, int32_t nq, int32_t k, int64_t doc_base,
                                       int32_t *out_doc, float *out_score, int32_t *out_count, void *workspace,
                                       int64_t workspace_bytes, void *stream_v) {
    return dense_search_i8_impl(device, (const int8_t *)corpus_packed, corpus_scale, n_docs, dim, queries, query_scale, nq, k, doc_base,
                                out_doc, out_score, out_count, workspace, workspace_bytes, stream_v, 1);
}

SRX_API int64_t srx_dense_packed_bytes(int64_t n_rows, int32_t dim) {
    if (n_rows < 0 || dim <= 0 || dim % 32 != 0 || dim > 1024) return fail(SRX_ERR_INVALID, "srx_dense_packed_bytes: bad argument%s");
    return (n_rows + 31) / 32 * 32 * (int64_t)dim;
}

SRX_API int srx_dense_pack_i8(int32_t device, const int8_t *rows, int64_t n_rows, int32_t dim, void *out_packed, void *stream_v) {
    if (n_rows <= 0 || dim <= 0 || dim % 32 != 0 || dim > 1024 || !rows || !out_packed)
        return fail(SRX_ERR_INVALID, "srx_dense_pack_i8: need rows, n_rows > 0, dim a multiple of 32 <= 1024%s");
    if (((uintptr_t)rows | (uintptr_t)out_packed) & 15) return fail(SRX_ERR_INVALID, "srx_dense_pack_i8: rows / out_packed must be 16-byte aligned%s");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(srx_dense_pack_queries_kernel, dim3(4096), dim3(THREADS), 0, (hipStream_t)stream_v, rows, n_rows, (int)dim,
                       (v4i *)out_packed);
    HIP_TRY(hipGetLastError());
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
                                                                       float *__restrict__ scores, int64_t ld, float score_offset) {
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
                if (lane == 0) scores[(int64_t)q * ld + d] = a + score_offset;  // + 0.0f by default (exact)
            }
        }
    }
}

// The asymmetric (uint8) scheme of the same retriever (rag_system/core/retriever_registry.py:449-462, 550-559): the
// reference de-quantizes both sides in fp32 -- doc_fp32 = u8 * doc_scale + doc_min, the query likewise -- and takes
// np.dot of the two fp32 vectors.  Same streaming matvec as above with the row de-quantized in registers (one fp32
// multiply and one add per element, the reference's two operations; the corpus streams as 1 byte per element).
// scale_min is the reference's corpus_scales table AS ITS READER INDEXES IT: doc d takes [2 d] and [2 d + 1] (:552-553).
__global__ __launch_bounds__(THREADS) void srx_dense_u8_scores_kernel(const uint8_t *__restrict__ corpus,
                                                                      const float *__restrict__ scale_min, int64_t n_docs, int dim,
                                                                      const float *__restrict__ queries, int nqp,
                                                                      float *__restrict__ scores, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int ns = dim >> 6;
    float qv[F32_QP][F32_MAXS];
#pragma unroll
    for (int q = 0; q < F32_QP; ++q)
#pragma unroll
        for (int i = 0; i < F32_MAXS; ++i) qv[q][i] = (q < nqp && i < ns) ? queries[(int64_t)q * dim + lane + 64 * i] : 0.0f;
    const int64_t wave = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * WAVES;
    for (int64_t d = wave; d < n_docs; d += n_waves) {
        const uint8_t *row = corpus + d * dim;
        const float sc = scale_min[2 * d], mn = scale_min[2 * d + 1];
        float r[F32_MAXS];
#pragma unroll
        for (int i = 0; i < F32_MAXS; ++i) r[i] = i < ns ? (float)row[lane + 64 * i] * sc + mn : 0.0f;
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

namespace {
// rows: f32 embeddings (u8_scale_min == nullptr) or uint8 rows de-quantized with u8_scale_min
int dense_rows_search(const char *who, int32_t device, const void *rows, const float *u8_scale_min, int64_t n_docs, int32_t dim,
                      const float *queries, int32_t nq, int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score,
                      int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream_v, float score_offset);
}  // namespace

SRX_API int srx_dense_search_f32(int32_t device, const float *emb, int64_t n_docs, int32_t dim, const float *queries, int32_t nq,
                                 int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score, int32_t *out_count,
                                 void *workspace, int64_t workspace_bytes, void *stream_v, float score_offset) {
    return dense_rows_search("srx_dense_search_f32", device, emb, nullptr, n_docs, dim, queries, nq, k, doc_base, out_doc, out_score,
                             out_count, workspace, workspace_bytes, stream_v, score_offset);
}

SRX_API int srx_dense_search_u8(int32_t device, const uint8_t *corpus, const float *corpus_scales, int64_t n_docs, int32_t dim,
                                const float *queries, int32_t nq, int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score,
                                int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream_v) {
    if (!corpus_scales) return fail(SRX_ERR_INVALID, "srx_dense_search_u8: null pointer%s");
    return dense_rows_search("srx_dense_search_u8", device, corpus, corpus_scales, n_docs, dim, queries, nq, k, doc_base, out_doc,
                             out_score, out_count, workspace, workspace_bytes, stream_v, 0.0f);
}

namespace {
int dense_rows_search(const char *who, int32_t device, const void *rows, const float *u8_scale_min, int64_t n_docs, int32_t dim,
                      const float *queries, int32_t nq, int32_t k, int64_t doc_base, int32_t *out_doc, float *out_score,
                      int32_t *out_count, void *workspace, int64_t workspace_bytes, void *stream_v, float score_offset) {
    const float *emb = (const float *)rows;
    if (nq < 0 || n_docs <= 0 || k <= 0 || k > KMAX) return fail(SRX_ERR_INVALID, "%s: need n_docs > 0, 1 <= k <= 1024", who);
    if (dim <= 0 || dim % 64 != 0 || dim > 64 * F32_MAXS)
        return fail(SRX_ERR_INVALID, "%s: dim must be a multiple of 64, <= 1024 (pad the rows with zeros)", who);
    if (doc_base < 0 || doc_base + n_docs >= 0x7FFFFFFFll) return fail(SRX_ERR_INVALID, "%s: doc_base + n_docs must fit int32", who);
    if (nq == 0) return SRX_OK;
    if (!emb || !queries || !out_doc || !out_score || !out_count) return fail(SRX_ERR_INVALID, "%s: null pointer", who);
    const int64_t need = srx_dense_f32_workspace_bytes(nq, n_docs, k);
    if (!workspace || workspace_bytes < need) return fail(SRX_ERR_NOMEM, "%s: workspace too small", who);
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
        if (u8_scale_min == nullptr)
            hipLaunchKernelGGL(srx_dense_f32_scores_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, stream, emb, n_docs, (int)dim,
                               queries + (int64_t)q0 * dim, qb, scores, ld, score_offset);
        else
            hipLaunchKernelGGL(srx_dense_u8_scores_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, stream, (const uint8_t *)rows,
                               u8_scale_min, n_docs, (int)dim, queries + (int64_t)q0 * dim, qb, scores, ld);
        hipLaunchKernelGGL(srx_dense_topk_kernel, dim3((unsigned)((int64_t)qb * ns)), dim3(THREADS), 0, stream, scores, ld, n_docs, qb, k,
                           ns, doc_base, 0, (const int32_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, 0,
                           (const int *)nullptr, cand_doc, cand_score, cand_count, (unsigned *)nullptr, 0);
        HIP_TRY(hipGetLastError());
        const int rc = srx_merge_impl(device, cand_doc, cand_score, cand_count, qb, ns, k, 0, (int64_t)k, (int64_t)1,
                                  out_doc + (int64_t)q0 * k, out_score + (int64_t)q0 * k, out_count + q0, (int64_t)k, (int64_t)1,
                                  nullptr, 0, stream_v);
        if (rc != SRX_OK) return rc;
    }
    return SRX_OK;
}
}  // namespace

#ifdef SRX_DSTAMP
extern "C" __attribute__((visibility("default"))) int srx_debug_read_dstamps(unsigned long long *h_out16) {
    HIP_TRY(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_dstamp), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dstamp), z, sizeof(z)));
    return SRX_OK;
}
#endif
