// Posting-stream load ceiling microbenchmark for gfx950 (design input for the tier-1 kernel; not product code).
//
// Emulates the memory access pattern of srx_wave_kernel on the C3 workload WITHOUT any scoring: one wavefront per
// query, 8 term streams per query (a group of 8 lanes each), every stream advances by one "run" of ~65 postings per
// unit, 153 units per query, the next unit's loads are issued before the current unit's data is consumed.
// Variants:
//   0  split arrays (docs / values), runs start at arbitrary 4-byte offsets, 4+4 dwordx4 per lane per unit, idle
//      slots redirected to the head of the array                      -- round-1 layout
//   1  split arrays, runs padded to a multiple of 4 postings and 16-byte aligned
//   2  block-interleaved [4 docs | 4 values] 32-byte blocks, runs padded + 32-byte aligned: a lane's two dwordx4 are
//      adjacent, a lane group reads 256 contiguous bytes per step
//   3  as 2 with 6-byte postings: [4 x u16 docs (8 B) | 4 x f32 (16 B)] = 24-byte blocks (dwordx2 + dwordx4)
//   4  as 2, prefetch distance 2 units (three register sets)
//   5  as 3 with the 24-byte blocks stored in 48-byte pairs [docs A | docs B][values A][values B] (aligned dwordx4)
// Reports useful bytes / time.  usage: stream_microbench [n_queries] [lds_bytes_per_wave]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNITS = 153;
constexpr int NT = 8;        // terms per query
constexpr int LPT = 8;       // lanes per term
constexpr int STEPS = 4;     // dwordx4 steps per unit (16 postings per lane)

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ c * 0xC2B2AE35u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}
// run length of (query, term, unit): 49..80, mean ~64.5 (C3: Poisson mean 65.5)
__device__ __forceinline__ int run_len(unsigned q, unsigned t, unsigned u) { return 49 + (int)(hash3(q, t, u) & 31); }

struct __attribute__((packed, aligned(4))) P4 { int x, y, z, w; };
struct __attribute__((packed, aligned(4))) P2 { int x, y; };

template <int VAR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void k(const int *__restrict__ A, const int *__restrict__ B,
                                                                                     uint64_t words, int *sink) {
    extern __shared__ int lds[];
    const int lane = threadIdx.x, q = blockIdx.x;
    const int t = lane >> 3, jl = lane & 7;
    if (lane == 0) lds[0] = 0;
    // stream start (in postings) of my term: pseudo-random, far apart
    uint64_t pos = ((uint64_t)hash3(q, t, 12345u) * 2654435761ull) % (words - (uint64_t)UNITS * 96 - 64);
    if (VAR >= 1) pos &= ~3ull;
    int acc = 0;
    auto issue = [&](uint64_t p, int len, int (&d)[16], int (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int eo = (s * LPT + jl) * 4;  // posting offset of my 4
            const bool ok = eo < len;
            if (VAR <= 1) {
                const P4 a = *reinterpret_cast<const P4 *>(A + (ok ? p + eo : (uint64_t)eo));
                const P4 b = *reinterpret_cast<const P4 *>(B + (ok ? p + eo : (uint64_t)eo));
                d[4 * s] = a.x; d[4 * s + 1] = a.y; d[4 * s + 2] = a.z; d[4 * s + 3] = a.w;
                v[4 * s] = b.x; v[4 * s + 1] = b.y; v[4 * s + 2] = b.z; v[4 * s + 3] = b.w;
            } else if (VAR == 3) {
                const uint64_t w0 = ok ? ((p + eo) >> 2) * 6 : (uint64_t)(eo >> 2) * 6;  // 24-byte blocks = 6 dwords
                const P2 a = *reinterpret_cast<const P2 *>(A + w0);
                const P4 b = *reinterpret_cast<const P4 *>(A + w0 + 2);
                d[4 * s] = a.x & 0xFFFF; d[4 * s + 1] = a.x >> 16; d[4 * s + 2] = a.y & 0xFFFF; d[4 * s + 3] = a.y >> 16;
                v[4 * s] = b.x; v[4 * s + 1] = b.y; v[4 * s + 2] = b.z; v[4 * s + 3] = b.w;
            } else if (VAR == 5) {
                // 24-byte blocks stored as PAIRS of 48 bytes: [docs A (8 B) | docs B (8 B)][values A (16 B)][values B (16 B)]:
                // every dwordx4 is 16-byte aligned, every dwordx2 8-byte aligned, for any block index
                const uint64_t blk = ok ? ((p + eo) >> 2) : (uint64_t)(eo >> 2);
                const uint64_t w0 = (blk >> 1) * 12 + (blk & 1) * 2;
                const P2 a = *reinterpret_cast<const P2 *>(A + w0);
                const P4 b = *reinterpret_cast<const P4 *>(A + w0 + 4 + (blk & 1) * 2);
                d[4 * s] = a.x & 0xFFFF; d[4 * s + 1] = a.x >> 16; d[4 * s + 2] = a.y & 0xFFFF; d[4 * s + 3] = a.y >> 16;
                v[4 * s] = b.x; v[4 * s + 1] = b.y; v[4 * s + 2] = b.z; v[4 * s + 3] = b.w;
            } else {
                const uint64_t w0 = ok ? ((p + eo) >> 2) * 8 : (uint64_t)(eo >> 2) * 8;  // 32-byte blocks = 8 dwords
                const P4 a = *reinterpret_cast<const P4 *>(A + w0);
                const P4 b = *reinterpret_cast<const P4 *>(A + w0 + 4);
                d[4 * s] = a.x; d[4 * s + 1] = a.y; d[4 * s + 2] = a.z; d[4 * s + 3] = a.w;
                v[4 * s] = b.x; v[4 * s + 1] = b.y; v[4 * s + 2] = b.z; v[4 * s + 3] = b.w;
            }
        }
    };
    auto consume = [&](int (&d)[16], int (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += d[r] ^ v[r];
    };
    auto adv = [&](int len) __attribute__((always_inline)) { return (uint64_t)(VAR >= 1 ? ((len + 3) & ~3) : len); };
    int dA[16], vA[16], dB[16], vB[16], dC[16], vC[16];
    if (VAR != 4) {
        int len = run_len(q, t, 0);
        issue(pos, len, dA, vA);
        pos += adv(len);
        for (int u = 0; u < UNITS; u += 2) {
            int l1 = run_len(q, t, u + 1);
            issue(pos, u + 1 < UNITS ? l1 : 0, dB, vB);
            pos += adv(l1);
            consume(dA, vA);
            int l2 = run_len(q, t, u + 2);
            issue(pos, u + 2 < UNITS ? l2 : 0, dA, vA);
            pos += adv(l2);
            consume(dB, vB);
        }
    } else {
        int l0 = run_len(q, t, 0);
        issue(pos, l0, dA, vA); pos += adv(l0);
        int l1 = run_len(q, t, 1);
        issue(pos, l1, dB, vB); pos += adv(l1);
        for (int u = 0; u < UNITS; u += 3) {
            int l2 = run_len(q, t, u + 2);
            issue(pos, u + 2 < UNITS ? l2 : 0, dC, vC); pos += adv(l2);
            consume(dA, vA);
            int l3 = run_len(q, t, u + 3);
            issue(pos, u + 3 < UNITS ? l3 : 0, dA, vA); pos += adv(l3);
            consume(dB, vB);
            int l4 = run_len(q, t, u + 4);
            issue(pos, u + 4 < UNITS ? l4 : 0, dB, vB); pos += adv(l4);
            consume(dC, vC);
        }
    }
    if (acc == 0x7F123457) sink[0] = acc + lds[0];
}

int main(int argc, char **argv) {
    const int nq = argc > 1 ? atoi(argv[1]) : 10000;
    const int lds = argc > 2 ? atoi(argv[2]) : 12672;  // bytes per wave: what srx_wave_kernel uses (12 waves per CU)
    const uint64_t words = 1ull << 30;  // postings per array: 4 GiB of docs + 4 GiB of values (8 GiB interleaved)
    int *A, *B, *sink;
    CHECK(hipMalloc(&A, words * 8));  // interleaved variants use all of it; split variants A = first half, B = second half
    B = A + words;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(A, 1, words * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // mean useful postings per (query, term, unit) = 64.5
    const double postings = (double)nq * NT * UNITS * 64.5;
    auto run = [&](int var, auto kern, double bytes_per_posting) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(nq), dim3(64), lds, 0, A, B, words, sink);
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        const int R = 10;
        for (int rep = 0; rep < R; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nq), dim3(64), lds, 0, A, B, words, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("variant %d: avg %.3f ms  best %.3f ms  -> %.2f TB/s useful (%.0f B/posting; 8-B equivalent %.2f TB/s)\n", var, sum / R, best,
               postings * bytes_per_posting / (sum / R * 1e-3) / 1e12, bytes_per_posting, postings * 8.0 / (sum / R * 1e-3) / 1e12);
    };
    run(0, k<0>, 8.0);
    run(1, k<1>, 8.0);
    run(2, k<2>, 8.0);
    run(3, k<3>, 6.0);
    run(4, k<4>, 8.0);
    run(5, k<5>, 6.0);
    return 0;
}
