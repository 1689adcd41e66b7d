// Micro-benchmark round 5: wave-count / wave-arrangement variants of the interleaved LDS-DMA K loop.
//   WAVES_M x WAVES_N waves, each TM x TN MFMA tiles (32x32); BM = 32*TM*WAVES_M, BN = 32*TN*WAVES_N.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

template <int WM, int WN, int TM, int TN, int STAGE>
__global__ __launch_bounds__(64 * WM * WN) void k(const float* __restrict__ src, float* out, int chunks, int rs) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, ST = (BM + BN) * 128;
    constexpr int ROWS = BM + BN;                       // 8 rows per wave-instruction
    constexpr int NS = ROWS / 8 / NW;                   // staging slots per thread per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* f = reinterpret_cast<float*>(smem);
    for (int i = tid; i < (STAGE == 4 ? 3 : 2) * ST / 4; i += NT) f[i] = 1.0f + (i & 15);
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5, wm = wave / WN, wn = wave % WN;
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
    for (int i = 0; i < TM; ++i) { int r = wm * 32 * TM + i * 32 + frow; a_rd[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
    for (int j = 0; j < TN; ++j) { int r = wn * 32 * TN + j * 32 + frow; b_rd[j] = BM * 128 + r * 128; b_sw[j] = (r >> 1) & 7; }
    const float* gsrc[NS];
    int loff[NS];
    for (int s = 0; s < NS; ++s) {
        const int wi = s * NW + wave;                    // wave-instruction index over all staged rows
        int r = wi * 8 + (lane >> 3);
        int c = (lane & 7) ^ ((r >> 1) & 7);
        gsrc[s] = src + (long long)(r + (blockIdx.x % 5) * 64) * rs + c * 4;
        loff[s] = wi * 1024;
    }
    int koff = 0;
    if (STAGE) {
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(smem + loff[s]), 16, 0, 0);
        koff = 32;
    }
    constexpr int NSTG = STAGE == 4 ? 3 : 2;
    if (STAGE == 4) {
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(smem + ST + loff[s]), 16, 0, 0);
        koff = 64;
    }
    int sc = 0;
    for (int c = 0; c < chunks; ++c) {
        if (STAGE == 4) {
            // chunk c landed when at most the NS loads of chunk c+1 are still in flight
            if (NS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (NS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (NS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (NS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (NS == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            if (STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const char* st = smem + (STAGE == 4 ? sc : (c & 1)) * ST;
        char* nx = smem + (STAGE == 4 ? (sc + 2) % 3 : ((c + 1) & 1)) * ST;
        sc = sc == NSTG - 1 ? 0 : sc + 1;
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
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
            if (STAGE) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bool now = STAGE == 1 ? (s & 3) == kk : STAGE == 2 ? kk == 0 : STAGE == 3 ? (s & 1) == kk : (s & 3) == kk;
                    if (now) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(nx + loff[s]), 16, 0, 0);
                }
            }
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
        }
        koff = (koff + 32) & 255;
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * NT + tid] = s;
}

template <int WM, int WN, int TM, int TN, int STAGE>
void run(const char* name, const float* src, float* out, int wgs_per_cu) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int lds = (STAGE == 4 ? 3 : 2) * (BM + BN) * 128, chunks = 1500, grid = 256 * wgs_per_cu;
    if (lds * wgs_per_cu > 160 * 1024 || 64 * WM * WN * wgs_per_cu > 2048) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<WM, WN, TM, TN, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WM, WN, TM, TN, STAGE>), dim3(grid), dim3(64 * WM * WN), lds, 0, src, out, chunks, 4096);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<WM, WN, TM, TN, STAGE>), dim3(grid), dim3(64 * WM * WN), lds, 0, src, out, chunks, 4096);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-34s %s wg/cu=%d %8.3f ms %6.1f TF   (%s)\n", name, STAGE == 0 ? "none " : STAGE == 1 ? "dma-kk" : STAGE == 2 ? "dma-k0" : STAGE == 3 ? "dma-k01" : "dma3st", wgs_per_cu, ms,
           (double)grid * WM * WN * chunks * 16.0 * TM * TN * 4096.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

#define BOTH(WM, WN, TM, TN, name, w) run<WM, WN, TM, TN, 0>(name, src, out, w); run<WM, WN, TM, TN, 1>(name, src, out, w); run<WM, WN, TM, TN, 2>(name, src, out, w); run<WM, WN, TM, TN, 3>(name, src, out, w); run<WM, WN, TM, TN, 4>(name, src, out, w);

int main() {
    float *src, *out;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&out, 16 << 20);
    for (int w = 1; w <= 3; ++w) {
        BOTH(2, 2, 2, 2, "128x128 4w(2x2) 64x64/wave", w)
        BOTH(2, 2, 1, 2, "64x128 4w(2x2) 32x64/wave", w)
        BOTH(2, 2, 2, 1, "128x64 4w(2x2) 64x32/wave", w)
    }
    return 0;
}
