// Dev probe (GPU box): is the LDS float atomic (ds_add_f32) bit-identical to v_add_f32 (round-to-nearest-even, denormals
// kept), and do one wave's LDS atomics apply in program order?  Design input for the tier-2 dense accumulators.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void lds_fadd(float *p, float v) {
    asm volatile("ds_add_f32 %0, %1" : : "v"((unsigned)(uintptr_t)p), "v"(v) : "memory");
}

// out[i] = a[i] (+) b[i] through the LDS atomic; ref[i] through the VALU
__global__ void k_pair(const float *a, const float *b, float *out, float *ref, int n) {
    __shared__ float s[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    s[threadIdx.x] = a[i];
    lds_fadd(&s[threadIdx.x], b[i]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[i] = s[threadIdx.x];
    ref[i] = a[i] + b[i];
}
// chains: slot j of lane l receives m values in program order through the atomic; reference = ordered VALU sum
__global__ void k_chain(const float *v, float *out, float *ref, int m) {
    __shared__ float s[256];
    const int t = threadIdx.x, i = blockIdx.x * 256 + t;
    s[t] = 0.f;
    float r = 0.f;
    for (int j = 0; j < m; ++j) {
        const float x = v[(size_t)i * m + j];
        lds_fadd(&s[t], x);
        r = r + x;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[i] = s[t];
    ref[i] = r;
}

int main() {
    const int n = 1 << 22;
    float *a, *b, *o, *r;
    CHECK(hipMallocManaged(&a, n * 4)); CHECK(hipMallocManaged(&b, n * 4));
    CHECK(hipMallocManaged(&o, n * 4)); CHECK(hipMallocManaged(&r, n * 4));
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    auto mk = [&](int i) -> float {
        uint32_t x = rnd();
        const int cls = i & 7;
        if (cls == 0) x &= 0x807FFFFFu;                        // denormal / zero
        else if (cls == 1) x = (x & 0x007FFFFFu) | 0x00800000u | (x & 0x80000000u);  // smallest normals
        else if (cls == 2) x = (x & 0x807FFFFFu) | ((100u + (x >> 23) % 60u) << 23); // BM25-like magnitudes
        else if (cls == 3) x &= 0x7FFFFFFFu;                   // any non-negative (incl. inf / nan)
        float f; memcpy(&f, &x, 4); return f;
    };
    for (int i = 0; i < n; ++i) { a[i] = mk(i); b[i] = mk(i * 7 + 3); }
    hipLaunchKernelGGL(k_pair, dim3(n / 256), dim3(256), 0, 0, a, b, o, r, n);
    CHECK(hipDeviceSynchronize());
    long bad = 0, badden = 0, nden = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t x, y; memcpy(&x, &o[i], 4); memcpy(&y, &r[i], 4);
        const bool nan = (r[i] != r[i]) && (o[i] != o[i]);
        const bool den = (y & 0x7F800000u) == 0 || (((uint32_t &)a[i]) & 0x7F800000u) == 0 || (((uint32_t &)b[i]) & 0x7F800000u) == 0;
        nden += den;
        if (x != y && !nan) { if (den) ++badden; else ++bad; if (bad + badden <= 8) printf("mismatch a=%a b=%a lds=%a valu=%a\n", a[i], b[i], o[i], r[i]); }
    }
    printf("pair: n=%d mismatches(normal)=%ld mismatches(denormal-involved)=%ld of %ld\n", n, bad, badden, nden);
    const int m = 16, nc = n / m;
    for (int i = 0; i < n; ++i) { uint32_t x = (rnd() & 0x007FFFFFu) | ((110u + rnd() % 30u) << 23); memcpy(&a[i], &x, 4); }
    hipLaunchKernelGGL(k_chain, dim3(nc / 256), dim3(256), 0, 0, a, o, r, m);
    CHECK(hipDeviceSynchronize());
    long badc = 0;
    for (int i = 0; i < nc; ++i) badc += memcmp(&o[i], &r[i], 4) != 0;
    printf("chain: %d chains of %d ordered adds, mismatches=%ld\n", nc, m, badc);
    return (bad || badc) ? 1 : 0;
}
