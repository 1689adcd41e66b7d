// Micro-benchmarks of the fp32 MFMA issue rate on gfx950 under the instruction mixes of dg_gemm.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// variant 0: pure MFMA, NACC independent accumulators, operands fixed registers
template <int NACC>
__global__ __launch_bounds__(256) void k_pure(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// variant 1: LDS fragment reads (b128) feeding MFMAs, double-buffered in registers, barrier every chunk
template <int TM, int TN, bool BARRIER, bool PREFETCH>
__global__ __launch_bounds__(256) void k_lds(float* out, int chunks, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* f = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * (64 * (TM + TN)) * 32; i += 256) f[i] = seed + (i & 15);
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
    for (int i = 0; i < TM; ++i) { int r = wm * 32 * TM + i * 32 + frow; a_rd[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
    for (int j = 0; j < TN; ++j) { int r = wn * 32 * TN + j * 32 + frow; b_rd[j] = 64 * TM * 128 + r * 128; b_sw[j] = (r >> 1) & 7; }
    const int stage_bytes = 64 * (TM + TN) * 128;
    for (int c = 0; c < chunks; ++c) {
        if (BARRIER) __syncthreads();
        const char* st = smem + (c & 1) * stage_bytes;
        if (PREFETCH) {
            f32x4 a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((fh ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((fh ^ b_sw[j]) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
                    const int chunk = (kk + 1) * 2 + fh;
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((chunk ^ a_sw[i]) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 a[TM], b[TN];
                const int chunk = kk * 2 + fh;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((chunk ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 256 * 4096 * sizeof(float)));
    const int CUS = 256;
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = CUS * wgs_per_cu;
        {
            const int iters = 2000;
            double ms = time_ms([&] { hipLaunchKernelGGL(k_pure<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3);
            double fl = (double)grid * 4 * iters * 8 * 4 * 4096.0;
            printf("pure4acc      wg/cu=%d  %.3f ms  %.1f TF\n", wgs_per_cu, ms, fl / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_pure<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3);
            fl = (double)grid * 4 * iters * 8 * 2 * 4096.0;
            printf("pure2acc      wg/cu=%d  %.3f ms  %.1f TF\n", wgs_per_cu, ms, fl / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_pure<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3);
            fl = (double)grid * 4 * iters * 8 * 1 * 4096.0;
            printf("pure1acc      wg/cu=%d  %.3f ms  %.1f TF\n", wgs_per_cu, ms, fl / ms / 1e9);
        }
        const int chunks = 2000;
#define RUN(TM, TN, BAR, PF, name)                                                                              \
        {                                                                                                       \
            const int lds = 2 * 64 * (TM + TN) * 128;                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<TM, TN, BAR, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            double ms = time_ms([&] { hipLaunchKernelGGL((k_lds<TM, TN, BAR, PF>), dim3(grid), dim3(256), lds, 0, out, chunks, 1.f); }, 3); \
            double fl = (double)grid * 4 * chunks * 16.0 * TM * TN * 4096.0;                                   \
            printf("%-13s wg/cu=%d  %.3f ms  %.1f TF\n", name, wgs_per_cu, ms, fl / ms / 1e9);               \
        }
        RUN(1, 2, true, false, "lds 1x2 bar")
        RUN(1, 2, false, false, "lds 1x2 nobar")
        RUN(1, 2, true, true, "lds 1x2 bar pf")
        RUN(2, 2, true, false, "lds 2x2 bar")
        RUN(2, 2, true, true, "lds 2x2 bar pf")
        RUN(2, 2, false, true, "lds 2x2 nobar pf")
    }
    hipFree(out);
    return 0;
}
