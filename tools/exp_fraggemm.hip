// Microbenchmark (round 6): does an LDS-free inner loop beat the LDS-DMA-staged one of dg_gemm.hip?
//
// Both operands are stored in MFMA FRAGMENT ORDER -- a (32 rows x 8 k) block is one contiguous KB, lane l holding
// row l % 32, k = 4 * (l / 32) .. + 3 -- so every operand fetch is one fully coalesced buffer_load_dwordx4 per wave straight
// into the registers v_mfma_f32_32x32x2_f32 reads: no LDS, no barrier, no ds_read, waves independent of each other.
// The weights are the FIRST MFMA operand (rows = output channels), the activations the second (columns = (latent row, position)
// pairs): the accumulators then hold 4 consecutive channels of one activation row in 4 consecutive registers, i.e. the output
// is already in the fragment order of the next layer's input and leaves in contiguous 1 KB stores.
//
// Shape mimics Generator.3's forward at 2560 latent rows: M = 125 440 activation rows (A = 64 MB at 128 channels), T taps that
// re-read shifted row blocks of A, N = 64 or 128 output channels.  Prints TFLOP/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_fraggemm.hip -o scratch/exp_fraggemm && scratch/exp_fraggemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_MFMA = 0x8, SG_VMEM = 0x20;   // LLVM SchedGroupMask: 0x20 = VMEM read

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args {
    const float* A;      // activations, fragment order: [MB][KC8][64][4], MB = M / 32, KC8 = Cin / 8
    const float* W;      // weights, fragment order: [NB][T * KC8][64][4], NB = N / 32
    float* Out;          // fragment order of the next layer: [MB][N / 8][64][4]
    int MB, KC8, T, NB;
    int tap_shift;       // tap t reads activation block (mb + t * tap_shift) % MB
};

// one wave = TN activation blocks (32 rows each) x TW channel blocks (32 channels each); ring of D k8-steps in flight
template <int TW, int TN, int D, int OCC>
__global__ __launch_bounds__(256, OCC) void frag_gemm(Args g) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int n_wtiles = g.NB / TW;
    const int mt = wave / n_wtiles, wt = wave % n_wtiles;
    if (mt * TN >= g.MB) return;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, 0x7ffffff0, 0x00020000);
    const int voff = lane * 16;
    const int S = g.T * g.KC8;                 // k8-steps
    f32x16 acc[TW][TN];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 fw[D][TW], fa[D][TN];
    auto load_step = [&](int s, int d) {
        const int sc = s < S ? s : S - 1;
        const int t = sc >> 4, k8 = sc & 15;                 // KC8 == 16 (checked by main)
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const int soff = ((wt * TW + i) * S + sc) * 1024;
            fw[d][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, soff, 0));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int mb = mt * TN + j + t * g.tap_shift;
            if (mb >= g.MB) mb -= g.MB;
            const int soff = (mb * 16 + k8) * 1024;
            fa[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voff, soff, 0));
        }
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) { load_step(d, d); __builtin_amdgcn_sched_barrier(0); }    // keep the prologue loads in ring order: the waits of the loop's back edge are merged with this path's
    for (int s0 = 0; s0 < S; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // refill the stage consumed in the previous step with step s0 + d + D - 1, then the MFMAs of stage d
            load_step(s0 + d + D - 1, (d + D - 1) % D);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TW; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[d][i][e], fa[d][j][e], acc[i][j], 0, 0, 0);
            // pin: loads spread evenly among the MFMAs
            constexpr int NL = TW + TN, NM = 4 * TW * TN;
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM / NL, 0);
            }
            if (NM % NL) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM % NL, 0);
        }
    }
    // epilogue: bias-free ReLU, stores in the next layer's fragment order
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(g.Out, 0, 0x7ffffff0, 0x00020000);
    const int C8 = g.NB * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                const int soff = ((mt * TN + j) * C8 + (wt * TW + i) * 4 + q) * 1024;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, voff, soff, 0);
            }
}

template <int TW, int TN, int D, int OCC>
double run(const char* name, const Args& a, int reps, std::vector<float>* out_host) {
    const int waves = (a.MB / TN) * (a.NB / TW);
    const int grid = (waves + 3) / 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((frag_gemm<TW, TN, D, OCC>), dim3(grid), dim3(256), 0, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((frag_gemm<TW, TN, D, OCC>), dim3(grid), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flop = 2.0 * a.MB * 32.0 * a.NB * 32.0 * a.T * a.KC8 * 8.0;
    const double tf = flop / (us * 1e-6) * 1e-12;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(frag_gemm<TW, TN, D, OCC>)));
    printf("%-28s N %3d  wave tile %3dx%3d  ring %d  regs %3d  grid %5d  %8.1f us  %6.1f TF  %.3f of 157.3\n", name, a.NB * 32, TN * 32, TW * 32, D,
           fa.numRegs, grid, us, tf, tf / 157.3);
    fflush(stdout);
    if (out_host) {
        out_host->resize((size_t)a.MB * 32 * a.NB * 32);
        CK(hipMemcpy(out_host->data(), a.Out, out_host->size() * 4, hipMemcpyDeviceToHost));
    }
    return tf;
}

// control: the same accumulator structure and MFMA count per wave, operands from registers only -- what the matrix pipes deliver on this box
template <int TW, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void mfma_only(float* out, int steps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[TW][TN];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float fw[TW], fa[TN];
#pragma unroll
    for (int i = 0; i < TW; ++i) fw[i] = 1e-3f * (lane + i);
#pragma unroll
    for (int j = 0; j < TN; ++j) fa[j] = 1e-3f * (lane - j);
    for (int s = 0; s < steps; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[i], fa[j], acc[i][j], 0, 0, 0);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int TW, int TN, int OCC>
void run_ctl(const char* name, float* out, int steps, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mfma_only<TW, TN, OCC>), dim3(grid), dim3(256), 0, 0, out, steps);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mfma_only<TW, TN, OCC>), dim3(grid), dim3(256), 0, 0, out, steps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flop = (double)grid * 4 * steps * 4.0 * TW * TN * 4096.0;
    printf("%-28s grid %5d  %8.1f us  %6.1f TF  %.3f of 157.3\n", name, grid, us, flop / (us * 1e-6) * 1e-12, flop / (us * 1e-6) * 1e-12 / 157.3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    // MB = 4096 x k row blocks: every variant below is a whole number of dispatch rounds of equal jobs (no ragged end: the
    // steady-state rate of the loop, what the job lists of dg_plan.cpp approach at large row counts)
    const int MB = 4096 * (argc > 1 ? atoi(argv[1]) : 1), KC8 = 16, T = 9;
    const int hot = argc > 2 ? atoi(argv[2]) : 0;             // 1: every tap re-reads the SAME blocks (tap_shift 0: operands L2-hot)
    const int Cin = KC8 * 8;
    const size_t a_floats = (size_t)MB * 32 * Cin;
    std::vector<float> hA(a_floats);
    srand(1);
    for (auto& v : hA) v = (float)(rand() & 0xffffff) / 16777216.0f - 0.5f;
    float *dA, *dW, *dO;
    CK(hipMalloc(&dA, a_floats * 4));
    CK(hipMemcpy(dA, hA.data(), a_floats * 4, hipMemcpyHostToDevice));
    for (int N : {64, 128}) {
        const int NB = N / 32, S = T * KC8;
        std::vector<float> hW((size_t)NB * S * 256);
        for (auto& v : hW) v = ((float)(rand() & 0xffffff) / 16777216.0f - 0.5f) * 0.1f;
        CK(hipMalloc(&dW, hW.size() * 4));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dO, (size_t)MB * 32 * N * 4));
        Args a{dA, dW, dO, MB, KC8, T, NB, hot ? 0 : 7};
        std::vector<float> out;
        // correctness of the layouts, once: element (row r, channel c) against a float64 dot product
        run<2, 2, 4, 2>("check", a, 1, &out);
        double worst = 0.0;
        for (int probe = 0; probe < 64; ++probe) {
            const int r = (int)(((long long)probe * 7919 * 131) % ((long long)MB * 32)), c = (probe * 37) % N;
            double ref = 0.0;
            for (int t = 0; t < T; ++t) {
                const int mb = (r / 32 + t * a.tap_shift) % MB;
                for (int k = 0; k < Cin; ++k) {
                    const float av = hA[(((size_t)mb * KC8 + k / 8) * 64 + ((k % 8) / 4) * 32 + r % 32) * 4 + k % 4];
                    const int s = t * KC8 + k / 8;
                    const float wv = hW[(((size_t)(c / 32) * S + s) * 64 + ((k % 8) / 4) * 32 + c % 32) * 4 + k % 4];
                    ref += (double)av * wv;
                }
            }
            if (ref < 0) ref = 0;
            const float got = out[(((size_t)(r / 32) * (N / 8) + c / 8) * 64 + ((c % 8) / 4) * 32 + r % 32) * 4 + c % 4];
            worst = std::max(worst, std::fabs(got - ref));
        }
        printf("N %d: max |device - float64| over 64 probes = %.3g\n", N, worst);
        const int reps = 20;
        run<2, 2, 3, 2>("w64x64 ring3 occ2", a, reps, nullptr);
        run<2, 2, 4, 2>("w64x64 ring4 occ2", a, reps, nullptr);
        run<2, 2, 6, 2>("w64x64 ring6 occ2", a, reps, nullptr);
        run<2, 2, 4, 3>("w64x64 ring4 occ3", a, reps, nullptr);
        run<2, 2, 3, 3>("w64x64 ring3 occ3", a, reps, nullptr);
        run<2, 4, 3, 2>("w64ch x128rows ring3 occ2", a, reps, nullptr);
        run<2, 4, 4, 2>("w64ch x128rows ring4 occ2", a, reps, nullptr);
        run<2, 4, 4, 1>("w64ch x128rows ring4 occ1", a, reps, nullptr);
        if (N == 128) {
            run<4, 2, 3, 2>("w128ch x64rows ring3 occ2", a, reps, nullptr);
            run<4, 2, 4, 2>("w128ch x64rows ring4 occ2", a, reps, nullptr);
            run<4, 4, 3, 1>("w128ch x128rows ring3 occ1", a, reps, nullptr);
            run<4, 4, 4, 1>("w128ch x128rows ring4 occ1", a, reps, nullptr);
        }
        CK(hipFree(dW)); CK(hipFree(dO));
    }
    CK(hipMalloc(&dO, 4096));
    run_ctl<2, 2, 2>("control: MFMA only 2x2 occ2", dO, 144, 512, 20);
    run_ctl<2, 4, 2>("control: MFMA only 2x4 occ2", dO, 144, 512, 20);
    run_ctl<2, 4, 1>("control: MFMA only 2x4 occ1", dO, 144, 256, 20);
    run_ctl<2, 4, 2>("control: MFMA only 2x4 occ2 x4 rounds", dO, 144, 2048, 20);
    return 0;
}
