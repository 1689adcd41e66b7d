// Micro-benchmark round 2: staging variants inside the production K-loop structure (fp32 MFMA 32x32x2).
//  V_DMA2   : LDS-DMA, 2 stages, vmcnt(0) per chunk (production v1)
//  V_DMA3   : LDS-DMA, 3 stages, counted vmcnt (one chunk stays in flight across the barrier)
//  V_REG    : register staged (global_load_dwordx4 -> ds_write_b128), 2 stages
//  V_SPEC   : wave specialisation: 4 MFMA waves + NL loader waves (LDS-DMA), 2 stages
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

enum { V_NONE = 0, V_DMA2 = 1, V_DMA3 = 2, V_REG = 3, V_SPEC = 4 };

template <int TM, int TN>
struct Frag {
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
    __device__ void init(int wm, int wn, int frow, int BM) {
        for (int i = 0; i < TM; ++i) { int r = wm * 32 * TM + i * 32 + frow; a_rd[i] = r * 128; a_sw[i] = (r >> 1) & 7; }
        for (int j = 0; j < TN; ++j) { int r = wn * 32 * TN + j * 32 + frow; b_rd[j] = BM * 128 + r * 128; b_sw[j] = (r >> 1) & 7; }
    }
};

template <int TM, int TN>
__device__ __forceinline__ void compute_chunk(const char* st, const Frag<TM, TN>& fr, int fh, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        f32x4 a[TM], b[TN];
        const int chunk = kk * 2 + fh;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + fr.a_rd[i] + ((chunk ^ fr.a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + fr.b_rd[j] + ((chunk ^ fr.b_sw[j]) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
}

template <int TM, int TN, int V, int NL>
__global__ __launch_bounds__(256 + 64 * NL) void k(const float* __restrict__ src, float* out, int chunks, int rs, long long* clkout) {
    const long long c0 = clock64(), r0 = wall_clock64();
    constexpr int BM = 64 * TM, BN = 64 * TN, ST = (BM + BN) * 128;
    constexpr int NSTAGE = (V == V_DMA3) ? 3 : 2;
    constexpr int ROWS = BM + BN;                 // staged rows per chunk; 8 rows per wave-instruction
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* f = reinterpret_cast<float*>(smem);
    for (int i = tid; i < NSTAGE * ST / 4; i += 256 + 64 * NL) f[i] = 1.0f + (i & 15);
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    Frag<TM, TN> fr;
    fr.init((wave >> 1) & 1, wave & 1, frow, BM);

    if (V == V_SPEC) {
        constexpr int NLs = NL > 0 ? NL : 1;
        constexpr int PER = ROWS / 8 / NLs;       // wave-instructions per loader wave per chunk
        if (wave >= 4) {
            const int lw = wave - 4;
            const float* p[PER];
            for (int s = 0; s < PER; ++s) { int wi = s * NLs + lw; int r = wi * 8 + (lane >> 3); int c = (lane & 7) ^ ((r >> 1) & 7); p[s] = src + (long long)(blockIdx.x % 7 * 64 + r) * rs + c * 4; }
            int koff = 0;
            for (int c = 0; c < chunks; ++c) {
                char* sA = smem + ((c + 1) & 1) * ST;
                // stage (c+1)&1 is free once every compute wave passed barrier c (= finished chunk c-1)
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int s = 0; s < PER; ++s) __builtin_amdgcn_global_load_lds(GP(p[s] + koff), LP(sA + (s * NLs + lw) * 1024), 16, 0, 0);
                koff = (koff + 32) & 255;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            return;
        }
        (void)c0; (void)r0;
        for (int c = 0; c < chunks; ++c) {
            __builtin_amdgcn_s_barrier();          // loaders finished chunk c (waited vmcnt before arriving)
            compute_chunk<TM, TN>(smem + (c & 1) * ST, fr, fh, acc);
        }
        __builtin_amdgcn_s_barrier();
    } else {
        constexpr int SA = BM / 32, SB = BN / 32;
        const float* asrc[SA]; const float* wsrc[SB];
        for (int s = 0; s < SA; ++s) { int r = (s * 4 + wave) * 8 + (lane >> 3); int c = (lane & 7) ^ ((r >> 1) & 7); asrc[s] = src + (long long)(blockIdx.x % 7 * BM + r) * rs + c * 4; }
        for (int s = 0; s < SB; ++s) { int r = (s * 4 + wave) * 8 + (lane >> 3); int c = (lane & 7) ^ ((r >> 1) & 7); wsrc[s] = src + (long long)(r + 1024) * rs + c * 4; }
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
        if (V == V_DMA2) issue_dma(0);
        if (V == V_DMA3) { issue_dma(0); issue_dma(1); }
        if (V == V_REG) { issue_reg(); write_reg(0); issue_reg(); }
        int stage = 0;
        for (int c = 0; c < chunks; ++c) {
            if (V == V_DMA2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); issue_dma((c + 1) & 1); }
            if (V == V_DMA3) {
                // chunk c landed when at most one chunk (SA+SB loads) is still in flight
                if (SA + SB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_dma(stage == 0 ? 2 : stage - 1);     // stage of chunk c+2 == stage of chunk c-1
            }
            if (V == V_REG) { __syncthreads(); write_reg((c + 1) & 1); issue_reg(); }
            if (V == V_NONE) __syncthreads();
            const char* st = smem + ((V == V_DMA3) ? stage : (c & 1)) * ST;
            compute_chunk<TM, TN>(st, fr, fh, acc);
            if (V == V_DMA3) stage = stage == 2 ? 0 : stage + 1;
        }
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (wave < 4) out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 3 && tid == 0) { clkout[0] = clock64() - c0; clkout[1] = wall_clock64() - r0; }
}

template <int TM, int TN, int V, int NL>
void run(const char* name, const float* src, float* out, int wgs_per_cu) {
    static long long* clk = nullptr; if (!clk) (void)hipMalloc(&clk, 16);
    const int nst = V == V_DMA3 ? 3 : 2;
    const int lds = nst * 64 * (TM + TN) * 128, chunks = 1500, grid = 256 * wgs_per_cu;
    if (lds * wgs_per_cu > 160 * 1024) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, V, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TM, TN, V, NL>), dim3(grid), dim3(256 + 64 * NL), lds, 0, src, out, chunks, 4096, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TM, TN, V, NL>), dim3(grid), dim3(256 + 64 * NL), lds, 0, src, out, chunks, 4096, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-24s wg/cu=%d %8.3f ms %6.1f TF  clk %.3f GHz  (%s)\n", name, wgs_per_cu, ms, (double)grid * 4 * chunks * 16.0 * TM * TN * 4096.0 / ms / 1e9,
           hc[1] ? 0.1 * hc[0] / hc[1] : 0.0, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&out, 4 << 20);
    for (int w = 1; w <= 3; ++w) {
        run<1, 2, V_NONE, 0>("64x128 none", src, out, w);
        run<1, 2, V_DMA2, 0>("64x128 dma2", src, out, w);
        run<1, 2, V_DMA3, 0>("64x128 dma3", src, out, w);
        run<1, 2, V_REG, 0>("64x128 reg", src, out, w);
        run<1, 2, V_SPEC, 1>("64x128 spec 1 loader", src, out, w);
        run<1, 2, V_SPEC, 2>("64x128 spec 2 loaders", src, out, w);
        run<2, 2, V_NONE, 0>("128x128 none", src, out, w);
        run<2, 2, V_DMA2, 0>("128x128 dma2", src, out, w);
        run<2, 2, V_DMA3, 0>("128x128 dma3", src, out, w);
        run<2, 2, V_REG, 0>("128x128 reg", src, out, w);
        run<2, 2, V_SPEC, 1>("128x128 spec 1 loader", src, out, w);
        run<2, 2, V_SPEC, 2>("128x128 spec 2 loaders", src, out, w);
        run<2, 2, V_SPEC, 4>("128x128 spec 4 loaders", src, out, w);
    }
    return 0;
}
