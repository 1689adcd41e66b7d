// Microbenchmark (round 6): clocks per v_mfma_f32_32x32x2_f32 at one wave per SIMD with 1 / 2 / 4 independent accumulator chains, on one CU and on all 256
// (shader clock under full matrix load), with and without an LDS read consumed on the spot.  hipcc --offload-arch=gfx950 -O3 tools/exp_mfma_chain.hip -o /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, bool LDSR>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters) {
    __shared__ float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = 1.0f / (1 + i);
    __syncthreads();
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x4 fr = {a, a, a, a};
    long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDSR) fr = *reinterpret_cast<const f32x4*>(sm + ((threadIdx.x * 4 + it * 64) & 8188));
#pragma unroll
        for (int j = 0; j < 16 / CH; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(LDSR ? fr[j & 3] : a, b, acc[c], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = (long long)(r1 - r0); }
}

template <int CH, bool LDSR>
void run(const char* name, int grid) {
    float* out; long long* clk;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 16);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<CH, LDSR>), dim3(grid), dim3(256), 0, 0, out, clk, iters); hipDeviceSynchronize(); }
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    double sc = 0, sr = 0;
    for (int i = 0; i < grid; ++i) { sc += h[2 * i]; sr += h[2 * i + 1]; }
    const double n = 16.0 * iters;
    printf("%-28s grid %4d: %.1f s_memtime ticks / MFMA, %.2f ns / MFMA (=> %.1f clocks at 2.4 GHz), s_memtime rate %.1f MHz\n", name, grid, sc / grid / n,
           sr / grid * 10.0 / n, sr / grid * 10.0 / n * 2.4, sc / sr * 100.0);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int grid : {1, 256}) {
        run<1, false>("1 chain", grid);
        run<2, false>("2 chains", grid);
        run<4, false>("4 chains", grid);
        run<1, true>("1 chain + ds_read_b128", grid);
        run<2, true>("2 chains + ds_read_b128", grid);
        run<4, true>("4 chains + ds_read_b128", grid);
    }
    return 0;
}
