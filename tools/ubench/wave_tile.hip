// Micro-benchmark: WAVE-PRIVATE tiles -- every wave owns a (32*TM) x (32*TN) output tile, stages its own operands
// by LDS-DMA into its own LDS region, no workgroup barrier at all.  BK = floats per K chunk (16 or 32).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

template <int TM, int TN, int BK, int WPB, bool STAGE>
__global__ __launch_bounds__(64 * WPB) void k(const float* __restrict__ src, float* out, int chunks, int rs) {
    constexpr int ROWS = 32 * (TM + TN), RB = BK * 4, ST = ROWS * RB;   // bytes per stage per wave
    constexpr int CPR = RB / 16;                                        // 16-B chunks per row (4 or 8)
    constexpr int RPI = 64 / CPR;                                       // rows per DMA instruction (16 or 8)
    constexpr int NS = ROWS / RPI;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* smem = smem_all + wave * 2 * ST;
    float* f = reinterpret_cast<float*>(smem);
    for (int i = lane; i < 2 * ST / 4; i += 64) f[i] = 1.0f + (i & 15);
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    auto swz = [](int r) { return CPR == 8 ? (r >> 1) & 7 : (r >> 2) & 3; };
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
    for (int i = 0; i < TM; ++i) { int r = i * 32 + frow; a_rd[i] = r * RB; a_sw[i] = swz(r); }
    for (int j = 0; j < TN; ++j) { int r = 32 * TM + j * 32 + frow; b_rd[j] = r * RB; b_sw[j] = swz(r); }
    const float* gsrc[NS];
    for (int s = 0; s < NS; ++s) {
        int r = s * RPI + lane / CPR;
        int c = (lane % CPR) ^ swz(r);
        gsrc[s] = src + (long long)(r + ((blockIdx.x * WPB + wave) % 5) * 64) * rs + c * 4;
    }
    int koff = 0;
    if (STAGE) {
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(smem + s * 1024), 16, 0, 0);
        koff = BK;
    }
    constexpr int KK = BK / 8;
    for (int c = 0; c < chunks; ++c) {
        if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* st = smem + (c & 1) * ST;
        char* nx = smem + ((c + 1) & 1) * ST;
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((fh ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((fh ^ b_sw[j]) << 4));
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {
                const int chunk = (kk + 1) * 2 + fh;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((chunk ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
            if (STAGE) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (s % KK == kk) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(nx + s * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
        }
        koff = (koff + BK) & 255;
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[(blockIdx.x * WPB + wave) * 64 + lane] = s;
}

template <int TM, int TN, int BK, int WPB, bool STAGE>
void run(const char* name, const float* src, float* out, int waves_per_cu) {
    constexpr int ST = 32 * (TM + TN) * BK * 4;
    const int lds = 2 * ST * WPB;
    const int blocks_per_cu = waves_per_cu / WPB;
    if (blocks_per_cu < 1 || lds * blocks_per_cu > 160 * 1024) return;
    const int chunks = 1500 * 32 / BK, grid = 256 * blocks_per_cu;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, BK, WPB, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TM, TN, BK, WPB, STAGE>), dim3(grid), dim3(64 * WPB), lds, 0, src, out, chunks, 4096);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TM, TN, BK, WPB, STAGE>), dim3(grid), dim3(64 * WPB), lds, 0, src, out, chunks, 4096);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-28s %s waves/cu=%2d %8.3f ms %6.1f TF   (%s)\n", name, STAGE ? "dma " : "none", waves_per_cu, ms,
           (double)grid * WPB * chunks * (BK / 2.0) * TM * TN * 4096.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&out, 16 << 20);
    for (int w : {4, 8, 12, 16}) {
        run<2, 2, 16, 4, false>("wave 64x64 BK16 (4 w/blk)", src, out, w);
        run<2, 2, 16, 4, true>("wave 64x64 BK16 (4 w/blk)", src, out, w);
        run<2, 2, 32, 4, true>("wave 64x64 BK32 (4 w/blk)", src, out, w);
        run<2, 2, 16, 1, true>("wave 64x64 BK16 (1 w/blk)", src, out, w);
        run<2, 1, 16, 4, true>("wave 64x32 BK16 (4 w/blk)", src, out, w);
        run<1, 1, 16, 4, true>("wave 32x32 BK16 (4 w/blk)", src, out, w);
        run<4, 2, 16, 4, true>("wave 128x64 BK16 (4 w/blk)", src, out, w);
    }
    return 0;
}
