// Micro-benchmark: cost of the global->LDS staging flavour inside the production K-loop structure
// (double-buffered LDS, one barrier per 32-float K chunk, 4 waves as 2x2, fp32 MFMA 32x32x2).
// STAGE 0: no staging (LDS content constant)   1: global_load_lds dwordx4 (LDS-DMA)
//       2: global_load_dwordx4 -> VGPR -> ds_write_b128 after the barrier (register staging, T14 split)
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

template <int TM, int TN, int STAGE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int chunks, int src_rowstride) {
    constexpr int BM = 64 * TM, BN = 64 * TN, SA = BM / 32, SB = BN / 32, ST = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    float* f = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * ST / 4; i += 256) f[i] = 1.0f + (i & 15);
    __syncthreads();
    const float* asrc[SA]; const float* wsrc[SB];
    for (int s = 0; s < SA; ++s) { int r = (s * 4 + wave) * 8 + (lane >> 3); int c = (lane & 7) ^ ((r >> 1) & 7); asrc[s] = src + (long long)(blockIdx.x % 7 * BM + r) * src_rowstride + c * 4; }
    for (int s = 0; s < SB; ++s) { int r = (s * 4 + wave) * 8 + (lane >> 3); int c = (lane & 7) ^ ((r >> 1) & 7); wsrc[s] = src + (long long)(r + 1024) * src_rowstride + c * 4; }
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
    for (int i = 0; i < TM; ++i) { int r = wm * 32 * TM + i * 32 + frow; a_rd[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
    for (int j = 0; j < TN; ++j) { int r = wn * 32 * TN + j * 32 + frow; b_rd[j] = BM * 128 + r * 128; b_sw[j] = (r >> 1) & 7; }
    f32x4 ra[SA], rb[SB];
    int koff = 0;
    auto issue_dma = [&](int stage) {
        char* sA = smem + stage * ST; char* sB = sA + BM * 128;
#pragma unroll
        for (int s = 0; s < SA; ++s) __builtin_amdgcn_global_load_lds(GP(asrc[s] + koff), LP(sA + (s * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int s = 0; s < SB; ++s) __builtin_amdgcn_global_load_lds(GP(wsrc[s] + koff), LP(sB + (s * 4 + wave) * 1024), 16, 0, 0);
        koff = (koff + 32) & 255;
    };
    auto issue_reg = [&]() {
#pragma unroll
        for (int s = 0; s < SA; ++s) ra[s] = *reinterpret_cast<const f32x4*>(asrc[s] + koff);
#pragma unroll
        for (int s = 0; s < SB; ++s) rb[s] = *reinterpret_cast<const f32x4*>(wsrc[s] + koff);
        koff = (koff + 32) & 255;
    };
    auto write_reg = [&](int stage) {
        char* sA = smem + stage * ST; char* sB = sA + BM * 128;
#pragma unroll
        for (int s = 0; s < SA; ++s) *reinterpret_cast<f32x4*>(sA + (s * 4 + wave) * 1024 + lane * 16) = ra[s];
#pragma unroll
        for (int s = 0; s < SB; ++s) *reinterpret_cast<f32x4*>(sB + (s * 4 + wave) * 1024 + lane * 16) = rb[s];
    };
    if (STAGE == 1) issue_dma(0);
    if (STAGE == 2) { issue_reg(); write_reg(0); issue_reg(); }
    for (int c = 0; c < chunks; ++c) {
        if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (STAGE == 1) issue_dma((c + 1) & 1);
        if (STAGE == 2) { write_reg((c + 1) & 1); issue_reg(); }    // regs hold chunk c+1 (loaded during chunk c-1)
        const char* st = smem + (c & 1) * ST;
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
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TM, int TN, int STAGE>
void run(const char* name, const float* src, float* out, int wgs_per_cu) {
    const int lds = 2 * 64 * (TM + TN) * 128, chunks = 1500, grid = 256 * wgs_per_cu;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TM, TN, STAGE>), dim3(grid), dim3(256), lds, 0, src, out, chunks, 4096);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TM, TN, STAGE>), dim3(grid), dim3(256), lds, 0, src, out, chunks, 4096);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-22s wg/cu=%d %8.3f ms %6.1f TF   (%s)\n", name, wgs_per_cu, ms, (double)grid * 4 * chunks * 16.0 * TM * TN * 4096.0 / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&out, 4 << 20);
    for (int w = 1; w <= 3; ++w) {
        run<1, 2, 0>("64x128 none", src, out, w);
        run<1, 2, 1>("64x128 lds-dma", src, out, w);
        run<1, 2, 2>("64x128 reg-staged", src, out, w);
        if (w <= 2) {
            run<2, 2, 0>("128x128 none", src, out, w);
            run<2, 2, 1>("128x128 lds-dma", src, out, w);
            run<2, 2, 2>("128x128 reg-staged", src, out, w);
        }
    }
    return 0;
}
