// Posting-stream microbenchmark, round 3 (design input for the tier-1 kernel; not product code).
//
// Emulates the memory side of srx_wave_kernel on the C3 workload as it stands now, WITHOUT scoring: one wavefront per
// query, 8 term streams per query (8 lanes each), one run of 33..64 postings (mean 48.5) per (term, unit), UNITS units per
// query (205 = the 10 M-doc index, 26 = a 1.25 M-doc shard), 4 waves per SIMD, 10 KB of LDS per wave.  What varies:
//   format  B24  compact blocks of 24 bytes [4 x u16 ids][4 x f32]: dwordx2 + dwordx4 per lane and step   (the shipped form)
//           B32  canonical blocks of 32 bytes [4 x i32][4 x f32]: 2 dwordx4
//           S24  "split run": a run of n blocks stored as [n x 8 B of ids][n x 16 B of values]: a lane group's dwordx2 loads
//                are contiguous (64 B per step), its dwordx4 loads too (128 B per step); same 6 bytes per posting
//           P48  pairs: 8 postings = [8 x u16 ids (16 B)][8 x f32 (32 B)]: three dwordx4 per lane and step, runs padded to 8
//   depth   register sets: 2 = one unit in flight while one is consumed, 3 = two in flight
//   steps   F = always 3 load steps per unit (idle steps read a sentinel region), A = the third step only when some lane
//           needs it (uniform branch)
// Reports useful bytes / time (6 B per posting for the compact forms, 8 for B32).
// usage: stream_microbench2 [n_queries] [units]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LPT = 8;
constexpr int STEPS = 3;
enum { B24 = 0, B32 = 1, S24 = 2, P48 = 3 };

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ c * 0xC2B2AE35u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}
__device__ __forceinline__ int run_blocks(unsigned q, unsigned t, unsigned u) { return (33 + (int)(hash3(q, t, u) & 31) + 3) >> 2; }  // 9..16

typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
typedef int i2u __attribute__((ext_vector_type(2), aligned(4)));
#define GL __attribute__((address_space(1)))
__device__ __forceinline__ i4u ld4(const int *p) { return *(const GL i4u *)p; }
__device__ __forceinline__ i2u ld2(const int *p) { return *(const GL i2u *)p; }

template <int FMT>
struct Regs {
    int w[FMT == B32 ? 8 * STEPS : (FMT == P48 ? 12 * 2 : 6 * STEPS)];
};

template <int FMT, int DEPTH, bool ADAPT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k(const int *__restrict__ A, uint64_t words, int units, int *sink) {
    __shared__ int lds[2560];  // 10 KB per wave, as the kernel
    const int lane = threadIdx.x, q = blockIdx.x;
    const int t = lane >> 3, jl = lane & 7;
    if (lane == 0) lds[0] = 0;
    constexpr int BW = FMT == B32 ? 8 : 6;  // words per block of 4 postings
    // stream start of my term in BLOCKS, pseudo-random and far apart; the first 4096 words of A are the "sentinel region"
    uint64_t pos = 1024 + ((uint64_t)hash3(q, t, 12345u) * 2654435761ull) % (words / BW - (uint64_t)units * 20 - 2048);
    const int *zb = A + lane * BW;
    int acc = 0;
    auto issue = [&](uint64_t p, int nb, Regs<FMT> &r) __attribute__((always_inline)) -> int {
        const int *base = A + p * BW;
        const int rem = nb - jl;
        const bool third = __ballot(nb > 2 * LPT) != 0ull;
        if (FMT == P48) {  // 8 postings per lane and step: 2 steps cover 16 lanes-blocks... a pair = 2 blocks = 12 words
            const int np = (nb + 1) >> 1;  // pairs
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool ok = s * LPT + jl < np;
                const int *b = (ok ? base + (s * LPT + jl) * 12 : zb + s * LPT * 12);
                const i4u a = ld4(b), v0 = ld4(b + 4), v1 = ld4(b + 8);
                r.w[12 * s + 0] = a.x; r.w[12 * s + 1] = a.y; r.w[12 * s + 2] = a.z; r.w[12 * s + 3] = a.w;
                r.w[12 * s + 4] = v0.x; r.w[12 * s + 5] = v0.y; r.w[12 * s + 6] = v0.z; r.w[12 * s + 7] = v0.w;
                r.w[12 * s + 8] = v1.x; r.w[12 * s + 9] = v1.y; r.w[12 * s + 10] = v1.z; r.w[12 * s + 11] = v1.w;
            }
            return 2;
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (ADAPT && s == 2 && !third) break;  // uniform
            const bool ok = s * LPT < rem;
            if (FMT == B24) {
                const int *b = (ok ? base + jl * 6 : zb) + s * LPT * 6;
                const i2u a = ld2(b);
                const i4u v = ld4(b + 2);
                r.w[6 * s] = a.x; r.w[6 * s + 1] = a.y; r.w[6 * s + 2] = v.x; r.w[6 * s + 3] = v.y; r.w[6 * s + 4] = v.z; r.w[6 * s + 5] = v.w;
            } else if (FMT == B32) {
                const int *b = (ok ? base + jl * 8 : zb) + s * LPT * 8;
                const i4u a = ld4(b), v = ld4(b + 4);
                r.w[8 * s] = a.x; r.w[8 * s + 1] = a.y; r.w[8 * s + 2] = a.z; r.w[8 * s + 3] = a.w;
                r.w[8 * s + 4] = v.x; r.w[8 * s + 5] = v.y; r.w[8 * s + 6] = v.z; r.w[8 * s + 7] = v.w;
            } else {  // S24: ids at base + 2 blk, values at base + 2 nb + 4 blk
                const int *bi = (ok ? base + jl * 2 : zb) + s * LPT * 2;
                const int *bv = (ok ? base + 2 * nb + jl * 4 : zb + 128) + s * LPT * 4;
                const i2u a = ld2(bi);
                const i4u v = ld4(bv);
                r.w[6 * s] = a.x; r.w[6 * s + 1] = a.y; r.w[6 * s + 2] = v.x; r.w[6 * s + 3] = v.y; r.w[6 * s + 4] = v.z; r.w[6 * s + 5] = v.w;
            }
        }
        return (ADAPT && !third) ? 2 : 3;
    };
    auto consume = [&](Regs<FMT> &r, int ns) __attribute__((always_inline)) {
        constexpr int per = FMT == B32 ? 8 : (FMT == P48 ? 12 : 6);
#pragma unroll
        for (int i = 0; i < per * 2; ++i) acc += r.w[i];
        if (FMT != P48 && ns > 2) {  // uniform
#pragma unroll
            for (int i = per * 2; i < per * 3; ++i) acc += r.w[i];
        }
    };
    auto adv = [&](int nb) __attribute__((always_inline)) { return (uint64_t)(FMT == P48 ? ((nb + 1) & ~1) : nb); };
    Regs<FMT> rA, rB, rC;
    if (DEPTH == 2) {
        int nb = run_blocks(q, t, 0);
        int nA = issue(pos, nb, rA), nB = 0;
        pos += adv(nb);
        for (int u = 0; u < units; u += 2) {
            int n1 = run_blocks(q, t, u + 1);
            nB = issue(pos, u + 1 < units ? n1 : 0, rB);
            pos += adv(n1);
            consume(rA, nA);
            int n2 = run_blocks(q, t, u + 2);
            nA = issue(pos, u + 2 < units ? n2 : 0, rA);
            pos += adv(n2);
            consume(rB, nB);
        }
    } else {
        int n0 = run_blocks(q, t, 0);
        int nA = issue(pos, n0, rA); pos += adv(n0);
        int n1 = run_blocks(q, t, 1);
        int nB = issue(pos, n1, rB), nC = 0; pos += adv(n1);
        for (int u = 0; u < units; u += 3) {
            int n2 = run_blocks(q, t, u + 2);
            nC = issue(pos, u + 2 < units ? n2 : 0, rC); pos += adv(n2);
            consume(rA, nA);
            int n3 = run_blocks(q, t, u + 3);
            nA = issue(pos, u + 3 < units ? n3 : 0, rA); pos += adv(n3);
            consume(rB, nB);
            int n4 = run_blocks(q, t, u + 4);
            nB = issue(pos, u + 4 < units ? n4 : 0, rB); pos += adv(n4);
            consume(rC, nC);
        }
    }
    if (acc == 0x7F123457) sink[0] = acc + lds[0];
}

int main(int argc, char **argv) {
    const int nq = argc > 1 ? atoi(argv[1]) : 10000;
    const int units = argc > 2 ? atoi(argv[2]) : 205;
    const uint64_t words = 1ull << 31;  // 8 GiB
    int *A, *sink;
    CHECK(hipMalloc(&A, words * 4));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(A, 1, words * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double postings = (double)nq * 8 * units * 48.5;
    auto run = [&](const char *name, auto kern, double bpp) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(nq), dim3(64), 0, 0, A, words, units, sink);
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        const int R = 10;
        for (int rep = 0; rep < R; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nq), dim3(64), 0, 0, A, words, units, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-22s avg %.3f ms  best %.3f ms  -> %.2f TB/s useful at %.0f B/posting\n", name, sum / R, best,
               postings * bpp / (sum / R * 1e-3) / 1e12, bpp);
    };
    printf("nq %d units %d\n", nq, units);
    run("B24 depth2 fixed", k<B24, 2, false>, 6.0);
    run("B24 depth2 adaptive", k<B24, 2, true>, 6.0);
    run("B24 depth3 fixed", k<B24, 3, false>, 6.0);
    run("B24 depth3 adaptive", k<B24, 3, true>, 6.0);
    run("S24 depth2 fixed", k<S24, 2, false>, 6.0);
    run("S24 depth2 adaptive", k<S24, 2, true>, 6.0);
    run("S24 depth3 fixed", k<S24, 3, false>, 6.0);
    run("S24 depth3 adaptive", k<S24, 3, true>, 6.0);
    run("P48 depth2", k<P48, 2, false>, 6.0);
    run("P48 depth3", k<P48, 3, false>, 6.0);
    run("B32 depth2 fixed", k<B32, 2, false>, 8.0);
    run("B32 depth3 fixed", k<B32, 3, false>, 8.0);
    return 0;
}
