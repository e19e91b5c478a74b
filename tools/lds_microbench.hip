// LDS primitive cost microbenchmark for gfx950 (design input for the scoring kernel; not product code).
// Each kernel runs ITER x 16 wave-instructions of one LDS op with pseudo-random (or linear) addresses and
// reports shader cycles per wave-instruction, at `waves` 64-thread workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int SLOTS = 2048;
constexpr int ITER = 256;

template <int OP, bool RANDOM>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int *sink) {
    __shared__ int tbl[SLOTS];
    __shared__ float ftbl[SLOTS];
    const int lane = threadIdx.x;
    for (int i = lane; i < SLOTS; i += 64) { tbl[i] = -1; ftbl[i] = 0.f; }
    __syncthreads();
    unsigned x = (blockIdx.x * 64 + lane) * 2654435761u + 12345u;
    int acc = 0;
    float facc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x = x * 1664525u + 1013904223u;
            const unsigned a = RANDOM ? (x >> 21) : ((unsigned)(lane + 64 * r + it) & (SLOTS - 1));
            if (OP == 0) acc += atomicCAS(&tbl[a], -1, (int)(x >> 8));          // ds_cmpst_rtn_b32
            if (OP == 1) atomicAdd(&ftbl[a], 1.0f);                             // ds_add_f32 (no return)
            if (OP == 2) atomicOr((unsigned *)&tbl[a], 1u << (x & 31));         // ds_or_b32 (no return)
            if (OP == 3) acc += tbl[a];                                         // ds_read_b32
            if (OP == 4) tbl[a] = (int)x;                                       // ds_write_b32
            if (OP == 5) { const int o = tbl[a]; if (o == -1) tbl[a] = (int)x; acc += tbl[a]; }  // read / cond write / verify
            if (OP == 6) facc += atomicAdd(&ftbl[a], 1.0f);                     // ds_add_rtn_f32
            if (OP == 7) { ftbl[a] = ftbl[a] + 1.0f; }                          // plain RMW
            if (OP == 8) acc += atomicAdd((unsigned *)&tbl[a], 1u);             // ds_add_rtn_u32
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x7fffffff || facc == 1234.5f) sink[0] = acc;
}

template <int OP, bool RANDOM>
void run(const char *name, int waves_per_cu) {
    int blocks = 256 * waves_per_cu;
    unsigned long long *out;
    int *sink;
    hipMalloc(&out, blocks * 8);
    hipMalloc(&sink, 4);
    hipLaunchKernelGGL((k<OP, RANDOM>), dim3(blocks), dim3(64), 0, 0, out, sink);
    hipLaunchKernelGGL((k<OP, RANDOM>), dim3(blocks), dim3(64), 0, 0, out, sink);
    hipDeviceSynchronize();
    unsigned long long *h = (unsigned long long *)malloc(blocks * 8);
    hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    // s_memtime ticks at 100 MHz on gfx9? report raw ticks per wave-instruction and let the reader scale
    printf("%-34s %-6s waves/CU=%2d  ticks/wave-instr = %8.2f\n", name, RANDOM ? "random" : "linear", waves_per_cu,
           s / blocks / (ITER * 16.0));
    free(h);
    hipFree(out);
    hipFree(sink);
}

int main() {
    for (int w : {1, 4, 8, 16}) {
        run<0, true>("ds_cmpst_rtn_b32", w);
        run<1, true>("ds_add_f32 (no rtn)", w);
        run<2, true>("ds_or_b32 (no rtn)", w);
        run<3, true>("ds_read_b32", w);
        run<4, true>("ds_write_b32", w);
        run<5, true>("read+condwrite+verify", w);
        run<6, true>("ds_add_rtn_f32", w);
        run<7, true>("plain float RMW", w);
        run<8, true>("ds_add_rtn_u32", w);
        run<0, false>("ds_cmpst_rtn_b32", w);
        run<1, false>("ds_add_f32 (no rtn)", w);
        run<3, false>("ds_read_b32", w);
        run<4, false>("ds_write_b32", w);
    }
    return 0;
}
