// Micro-benchmark round 3: A operand straight from global memory into MFMA fragments (no LDS for A),
// B (filters) through LDS.  Waves arranged 4 x 1: each wave owns 32*TM rows x all BN columns.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

// BSTAGE: 1 = LDS-DMA, 2 = register staged.  DEPTH: A prefetch depth in kk-steps (1 or 2)
template <int TM, int TN, int BSTAGE, int DEPTH>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int chunks, int rs) {
    constexpr int BN = 32 * TN, ST = BN * 128, SB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* f = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * ST / 4; i += 256) f[i] = 1.0f + (i & 15);
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    int b_rd[TN], b_sw[TN];
    for (int j = 0; j < TN; ++j) { int r = j * 32 + frow; b_rd[j] = r * 128; b_sw[j] = (r >> 1) & 7; }
    const float* arow[TM];
    for (int i = 0; i < TM; ++i) arow[i] = src + (long long)((blockIdx.x % 7) * 128 * TM + wave * 32 * TM + i * 32 + frow) * rs + fh * 4;
    const float* wsrc[SB];
    for (int s = 0; s < SB; ++s) { int r = (s * 4 + wave) * 8 + (lane >> 3); int c = (lane & 7) ^ ((r >> 1) & 7); wsrc[s] = src + (long long)(r + 2048) * rs + c * 4; }
    f32x4 rb[SB];
    int koff = 0;
    auto issue_b = [&](int stage) {
        char* sB = smem + stage * ST;
        if (BSTAGE == 1) {
#pragma unroll
            for (int s = 0; s < SB; ++s) __builtin_amdgcn_global_load_lds(GP(wsrc[s] + koff), LP(sB + (s * 4 + wave) * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < SB; ++s) rb[s] = *reinterpret_cast<const f32x4*>(wsrc[s] + koff);
        }
    };
    auto write_b = [&](int stage) {
        char* sB = smem + stage * ST;
#pragma unroll
        for (int s = 0; s < SB; ++s) *reinterpret_cast<f32x4*>(sB + (s * 4 + wave) * 1024 + lane * 16) = rb[s];
    };
    // A fragment ring: a[d][i] holds kk-step (cur + d)
    f32x4 a[DEPTH + 1][TM];
    int akoff = 0;          // float offset of the next kk-step to fetch
    auto fetch_a = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[slot][i] = *reinterpret_cast<const f32x4*>(arow[i] + akoff);
        akoff = (akoff + 8) & 255;
    };
    if (BSTAGE == 1) issue_b(0); else { issue_b(0); write_b(0); koff = 32; issue_b(1); }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch_a(d);
    for (int c = 0; c < chunks; ++c) {
        if (BSTAGE == 1) {
            // NOTE: vmcnt(0) would also drain the A prefetch; count: DEPTH*TM A loads may stay in flight
            if (DEPTH * TM == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (DEPTH * TM == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            koff = (koff + 32) & 255;
            issue_b((c + 1) & 1);
        } else {
            __syncthreads();
            write_b((c + 1) & 1);
            koff = (koff + 32) & 255;
            issue_b(0);
        }
        const char* st = smem + (c & 1) * ST;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 b[TN];
            const int chunk = kk * 2 + fh;
            fetch_a((kk + DEPTH) % (DEPTH + 1));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk % (DEPTH + 1)][i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int TM, int TN, int BSTAGE, int DEPTH>
void run(const char* name, const float* src, float* out, int wgs_per_cu) {
    const int lds = 2 * 32 * TN * 128, chunks = 1500, grid = 256 * wgs_per_cu;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, BSTAGE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TM, TN, BSTAGE, DEPTH>), dim3(grid), dim3(256), lds, 0, src, out, chunks, 4096);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<TM, TN, BSTAGE, DEPTH>), dim3(grid), dim3(256), lds, 0, src, out, chunks, 4096);
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
        run<1, 4, 1, 1>("128x128 (4x1) Adir d1 Bdma", src, out, w);
        run<1, 4, 1, 2>("128x128 (4x1) Adir d2 Bdma", src, out, w);
        run<1, 4, 2, 2>("128x128 (4x1) Adir d2 Breg", src, out, w);
        run<2, 4, 1, 1>("256x128 (4x1) Adir d1 Bdma", src, out, w);
        run<2, 4, 1, 2>("256x128 (4x1) Adir d2 Bdma", src, out, w);
        run<1, 2, 1, 2>("128x64 (4x1) Adir d2 Bdma", src, out, w);
        run<2, 2, 1, 2>("256x64 (4x1) Adir d2 Bdma", src, out, w);
    }
    return 0;
}
