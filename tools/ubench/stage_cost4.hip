// Micro-benchmark round 4: staging work INTERLEAVED with the MFMA k-steps instead of bursting right after
// the barrier (where it delays the first fragment reads of the chunk).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

// V: 1 = LDS-DMA interleaved, 2 = register staged interleaved (ds_write spread), 3 = reg staged, writes after kk=0
template <int TM, int TN, int V, bool PIN>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int chunks, int rs) {
    constexpr int BM = 64 * TM, BN = 64 * TN, ST = (BM + BN) * 128, SA = BM / 32, SB = BN / 32, NS = SA + SB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* f = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * ST / 4; i += 256) f[i] = 1.0f + (i & 15);
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
    for (int i = 0; i < TM; ++i) { int r = wm * 32 * TM + i * 32 + frow; a_rd[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
    for (int j = 0; j < TN; ++j) { int r = wn * 32 * TN + j * 32 + frow; b_rd[j] = BM * 128 + r * 128; b_sw[j] = (r >> 1) & 7; }
    // unified staging slots: slot s < SA -> A rows, else B rows; LDS offset of slot s inside a stage
    const float* gsrc[NS];
    int loff[NS];
    for (int s = 0; s < NS; ++s) {
        const bool isA = s < SA;
        const int ss = isA ? s : s - SA;
        int r = (ss * 4 + wave) * 8 + (lane >> 3);
        int c = (lane & 7) ^ ((r >> 1) & 7);
        gsrc[s] = src + (long long)((isA ? (blockIdx.x % 7) * BM : 1024) + r) * rs + c * 4;
        loff[s] = (isA ? 0 : BM * 128) + (ss * 4 + wave) * 1024;
    }
    f32x4 rg[NS];
    int koff = 0;
    if (V == 1) {
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(smem + loff[s]), 16, 0, 0);
        koff = 32;
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) rg[s] = *reinterpret_cast<const f32x4*>(gsrc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(smem + loff[s] + lane * 16) = rg[s];
        koff = 32;
#pragma unroll
        for (int s = 0; s < NS; ++s) rg[s] = *reinterpret_cast<const f32x4*>(gsrc[s] + koff);
    }
    constexpr int PER = NS / 4;           // staging ops per kk-step
    for (int c = 0; c < chunks; ++c) {
        if (V == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* st = smem + (c & 1) * ST;
        char* nx = smem + ((c + 1) & 1) * ST;
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
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            // first half of this k-step's MFMAs
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            // this k-step's share of the staging work
            if (V == 1) {
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int s = kk * PER + q; __builtin_amdgcn_global_load_lds(GP(gsrc[s] + koff), LP(nx + loff[s]), 16, 0, 0); }
            } else if (V == 2) {
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int s = kk * PER + q; *reinterpret_cast<f32x4*>(nx + loff[s] + lane * 16) = rg[s]; }
            } else if (V == 3 && kk == 0) {
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(nx + loff[s] + lane * 16) = rg[s];
            }
            if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
        }
        koff = (koff + 32) & 255;
        if (V != 1) {
            // global loads of chunk c+2 into the (now written-out) staging registers
#pragma unroll
            for (int s = 0; s < NS; ++s) rg[s] = *reinterpret_cast<const f32x4*>(gsrc[s] + koff);
        }
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int TM, int TN, int V, bool PIN>
void run(const char* name, const float* src, float* out, int wgs_per_cu) {
    const int lds = 2 * 64 * (TM + TN) * 128, chunks = 1500, grid = 256 * wgs_per_cu;
    if (lds * wgs_per_cu > 160 * 1024) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, V, PIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TM, TN, V, PIN>), dim3(grid), dim3(256), lds, 0, src, out, chunks, 4096);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TM, TN, V, PIN>), dim3(grid), dim3(256), lds, 0, src, out, chunks, 4096);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-30s wg/cu=%d %8.3f ms %6.1f TF   (%s)\n", name, wgs_per_cu, ms, (double)grid * 4 * chunks * 16.0 * TM * TN * 4096.0 / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&out, 4 << 20);
    for (int w = 1; w <= 3; ++w) {
        run<1, 2, 1, true>("64x128 dma interleaved pin", src, out, w);
        run<1, 2, 1, false>("64x128 dma interleaved", src, out, w);
        run<1, 2, 2, true>("64x128 reg interleaved pin", src, out, w);
        run<1, 2, 2, false>("64x128 reg interleaved", src, out, w);
        run<1, 2, 3, true>("64x128 reg after-kk0 pin", src, out, w);
        run<2, 2, 1, true>("128x128 dma interleaved pin", src, out, w);
        run<2, 2, 1, false>("128x128 dma interleaved", src, out, w);
        run<2, 2, 2, true>("128x128 reg interleaved pin", src, out, w);
        run<2, 2, 2, false>("128x128 reg interleaved", src, out, w);
        run<2, 2, 3, true>("128x128 reg after-kk0 pin", src, out, w);
    }
    return 0;
}
