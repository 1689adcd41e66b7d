// Micro-benchmark: wave-private tiles, A staged by LDS-DMA into the wave's own LDS region (no barrier), B fragments
// loaded straight into VGPRs from a FRAGMENT-ORDER pack (one coalesced 1 KB run per load).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

// BMODE 0: B through LDS too (reference), 1: B direct packed, 2: B direct packed + A direct (fragment-shaped loads)
template <int TM, int TN, int WPB, int BMODE>
__global__ __launch_bounds__(64 * WPB) void k(const float* __restrict__ src, const float* __restrict__ bpack, float* out,
                                              int chunks, int rs) {
    constexpr int BK = 32, RB = 128;
    constexpr int AROWS = 32 * TM, BROWS = BMODE == 0 ? 32 * TN : 0, ROWS = AROWS + BROWS, ST = ROWS * RB;
    constexpr int NS = ROWS / 8;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* smem = smem_all + wave * 2 * ST;
    float* f = reinterpret_cast<float*>(smem);
    for (int i = lane; i < 2 * ST / 4; i += 64) f[i] = 1.0f + (i & 15);
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    int a_rd[TM], a_sw[TM], b_rd[TN], b_sw[TN];
    for (int i = 0; i < TM; ++i) { int r = i * 32 + frow; a_rd[i] = r * RB; a_sw[i] = (r >> 1) & 7; }
    for (int j = 0; j < TN; ++j) { int r = AROWS + j * 32 + frow; b_rd[j] = r * RB; b_sw[j] = (r >> 1) & 7; }
    const float* gsrc[NS > 0 ? NS : 1];
    for (int s = 0; s < NS; ++s) {
        int r = s * 8 + (lane >> 3);
        int c = (lane & 7) ^ ((r >> 1) & 7);
        gsrc[s] = src + (long long)(r + ((blockIdx.x * WPB + wave) % 5) * 64) * rs + c * 4;
    }
    const int wid = (blockIdx.x * WPB + wave);
    // packed B: [tile j][chunk (8 per 256-float K window)][kk (4)][lane][4]
    const float* bp = bpack + (long long)(wid % 3) * TN * 8 * 4 * 256;
    int koff = 0, kci = 0;
    if (NS > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(smem + s * 1024), 16, 0, 0);
    }
    koff = BK; 
    f32x4 bcur[TN], bnext[TN];
    if (BMODE >= 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bcur[j] = *reinterpret_cast<const f32x4*>(bp + ((j * 8 + 0) * 4 + 0) * 256 + lane * 4);
    }
    for (int c = 0; c < chunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* st = smem + (c & 1) * ST;
        char* nx = smem + ((c + 1) & 1) * ST;
        const int kcn = (kci + 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[TM], b[TN];
            const int chunk = kk * 2 + fh;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((chunk ^ a_sw[i]) << 4));
            if (BMODE == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
            } else {
                // next k-step's B fragments (next chunk's first k-step when kk == 3)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int kc2 = kk < 3 ? kci : kcn, kk2 = (kk + 1) & 3;
                    bnext[j] = *reinterpret_cast<const f32x4*>(bp + ((j * 8 + kc2) * 4 + kk2) * 256 + lane * 4);
                    b[j] = bcur[j];
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if ((s & 3) == kk) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(nx + s * 1024), 16, 0, 0);
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            if (BMODE >= 1) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bcur[j] = bnext[j];
            }
        }
        koff = (koff + BK) & 255;
        kci = kcn;
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[(blockIdx.x * WPB + wave) * 64 + lane] = s;
}

template <int TM, int TN, int WPB, int BMODE>
void run(const char* name, const float* src, const float* bpack, float* out, int waves_per_cu) {
    constexpr int ST = (32 * TM + (BMODE == 0 ? 32 * TN : 0)) * 128;
    const int lds = 2 * ST * WPB;
    const int blocks_per_cu = waves_per_cu / WPB;
    if (blocks_per_cu < 1 || lds * blocks_per_cu > 160 * 1024) return;
    const int chunks = 1500, grid = 256 * blocks_per_cu;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, WPB, BMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TM, TN, WPB, BMODE>), dim3(grid), dim3(64 * WPB), lds, 0, src, bpack, out, chunks, 4096);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TM, TN, WPB, BMODE>), dim3(grid), dim3(64 * WPB), lds, 0, src, bpack, out, chunks, 4096);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-26s %s waves/cu=%2d %8.3f ms %6.1f TF   (%s)\n", name, BMODE == 0 ? "B-lds   " : "B-direct", waves_per_cu, ms,
           (double)grid * WPB * chunks * 16.0 * TM * TN * 4096.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *src, *out, *bpack;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&bpack, 16 << 20); (void)hipMemset(bpack, 0, 16 << 20);
    (void)hipMalloc(&out, 16 << 20);
    for (int w : {4, 8, 12, 16}) {
        run<2, 2, 4, 0>("wave 64x64", src, bpack, out, w);
        run<2, 2, 4, 1>("wave 64x64", src, bpack, out, w);
        run<1, 2, 4, 0>("wave 32x64", src, bpack, out, w);
        run<1, 2, 4, 1>("wave 32x64", src, bpack, out, w);
        run<2, 4, 4, 1>("wave 64x128", src, bpack, out, w);
        run<1, 4, 4, 1>("wave 32x128", src, bpack, out, w);
    }
    return 0;
}
