// Shared device helpers of the generator-tail kernels (dg_tail_mnist.hip, dg_tail_celeba.hip), gfx950.
//
//
// A Cout <= 3 transposed conv is GEMV-shaped per output pixel, but per INPUT position it is a dense
// contraction: every position q contributes to its 25 taps x Cout outputs
//     P[q, kappa] = sum_c H[q, c] * F[kappa, c],      kappa = (kh*5 + kw)*Cout + co            (forward)
// a [positions x C] . [C x 25*Cout] GEMM (F's reference layout [5,5,Cout,C] IS [kappa][c], K-contiguous), followed by
// a fixed-order gather of the <= 9 P entries that land on each output pixel (i = 2*oh + kh - 1).  The backward is
//     dH[q, c] = sum_kappa G[q, kappa] * F[kappa, c],  G[q, kappa] = da_out[2*oh+kh-1, 2*ow+kw-1, co]
// a [positions x 25*Cout] . [25*Cout x C] GEMM whose A operand is gathered from a zero-bordered LDS image of da_out.
// Both run on v_mfma_f32_32x32x2_f32 (exact fp32); kappa is zero-padded to 32 (MNIST) / 96 (CelebA) columns
// forward and to an even count backward.  The filter fragments live in registers for the whole workgroup.
//
//   MNIST  (dataset_models.py:66-69): one workgroup per latent row, forward + backward fused, da3 written in
//          place over h3 with the ReluGrad mask.
//   CelebA (dataset_models.py:160-163): 8 bands per latent row; forward bands own 8 output rows (6 input rows
//          incl. halo), backward bands own 4 input rows; da6 is parked in HBM between the two kernels.
#pragma once
#include "dg_kernels.h"
#include "dg_device.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier for LDS hand-offs only: this wave's LDS operations are complete, then s_barrier.  Unlike __syncthreads()
// it does not wait for outstanding global / LDS-DMA traffic (vmcnt), which the M waves keep in flight across steps on purpose.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// ---- forward GEMM of one 32-position tile: P[q0 .. q0+31][0 .. 32*NT) -> LDS --------------------------------
// hrow: base of this latent row's input map [positions][C]; gp = global position index of this lane's row or -1.
template <int C>
__device__ __forceinline__ void tail_fwd_load(const float* __restrict__ hrow, int gp, f32x4 (&a)[C / 8], int lane) {
    const int fh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < C / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gp >= 0) v = *reinterpret_cast<const f32x4*>(hrow + (long long)gp * C + kk * 8 + fh * 4);
        a[kk] = v;
    }
}

template <int C, int NT>
__device__ __forceinline__ void tail_fwd_compute(const f32x4 (&a)[C / 8], int q0, const f32x4 (&w)[NT][C / 8], float* sP,
                                                 int NKP, int col0, int lane) {
    constexpr int KK = C / 8;
    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][e], w[t][kk][e], acc[t], 0, 0, 0);
    // D layout: col = lane&31 (kappa), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (position)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * fh;
            sP[(q0 + r) * NKP + col0 + t * 32 + frow] = acc[t][e];
        }
}

template <int C, int NT, int COUT>
__device__ __forceinline__ void tail_load_fwd_weights(const float* __restrict__ F, f32x4 (&w)[NT][C / 8], int t0, int lane) {
    constexpr int NK = 25 * COUT;
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int kappa = (t0 + t) * 32 + frow;
#pragma unroll
        for (int kk = 0; kk < C / 8; ++kk) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kappa < NK) v = *reinterpret_cast<const f32x4*>(F + kappa * C + kk * 8 + fh * 4);
            w[t][kk] = v;
        }
    }
}

#ifdef DG_MEASURE
// Same fragments from the FRAGMENT-ORDER pack built once at weight-load time (dg_engine.cpp, pack_tail_fragments):
// Wp[((t*C/8 + kk)*64 + lane)*4 + e] -- one fully coalesced 1 KB line run per load instead of 32 partial lines.
template <int C>
__device__ __forceinline__ void tail_load_fwd_weights_packed(const float* __restrict__ Wp, f32x4 (&w)[1][C / 8], int t, int lane) {
#pragma unroll
    for (int kk = 0; kk < C / 8; ++kk)
        w[0][kk] = *reinterpret_cast<const f32x4*>(Wp + (((long long)t * (C / 8) + kk) * 64 + lane) * 4);
}

#endif  // DG_MEASURE

// ---- backward GEMM of one 32-position tile ---------------------------------------------------------------------
// sg: zero-bordered da_out image in LDS, element (row, col, co) at (row*GWP + col)*COUT + co; the tile's position
// q (local) has its tap (kh,kw) at image row 2*ohl + kh, col 2*ow + kw.
template <int C, int COUT, int GWP>
struct BwdWeights {
    static constexpr int NK = 25 * COUT;
    static constexpr int NS = (NK + 1) / 2;      // k-steps (K = 2 per MFMA)
    float w[C / 32][NS];
    __device__ __forceinline__ void load(const float* __restrict__ F, int lane) {
        const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int u = 0; u < C / 32; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int kappa = 2 * s + fh;
                w[u][s] = kappa < NK ? F[kappa * C + u * 32 + frow] : 0.f;
            }
    }
};

template <int C, int COUT, int GWP>
__device__ __forceinline__ void tail_bwd_tile(const float* sg, int gbase, bool valid, const BwdWeights<C, COUT, GWP>& bw,
                                              f32x16 (&acc)[C / 32], int lane) {
    constexpr int NK = 25 * COUT, NS = (NK + 1) / 2;
    const int fh = lane >> 5;
#pragma unroll
    for (int u = 0; u < C / 32; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
    // The gathered A values are read in batches ahead of the MFMAs that consume them: read-then-use per k-step leaves
    // the LDS latency exposed behind every MFMA pair.
    constexpr int BATCH = NS <= 20 ? NS : (NS + 1) / 2;
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += BATCH) {
        float gv[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int s = s0 + q;
            if (s >= NS) { gv[q] = 0.f; continue; }
            // kappa = 2s + fh -> (tap = kappa / COUT, co = kappa % COUT), tap -> (kh, kw)
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int t0 = k0 / COUT, c0 = k0 % COUT, t1 = (k1 < NK ? k1 : k0) / COUT, c1 = (k1 < NK ? k1 : k0) % COUT;
            const int off0 = ((t0 / 5) * GWP + (t0 % 5)) * COUT + c0;
            const int off1 = ((t1 / 5) * GWP + (t1 % 5)) * COUT + c1;
            gv[q] = sg[gbase + (fh ? off1 : off0)];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            asm volatile("" : "+v"(gv[q]));
            if (!valid) gv[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int s = s0 + q;
            if (s >= NS) continue;
#pragma unroll
            for (int u = 0; u < C / 32; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[q], bw.w[u][s], acc[u], 0, 0, 0);
        }
    }
}

// Variants with the filter fragments in LDS instead of registers (the pipelined MNIST kernel keeps two rows of A
// fragments per wave and has no registers left for 58 filter values).
//   forward:  sWf[(kk*64 + lane)*4 + e]  = F[kappa = lane&31][c = 8*kk + 4*(lane>>5) + e]   (one kappa tile)
//   backward: sWb[(s*64 + lane)*U + u]   = F[kappa = 2*s + (lane>>5)][c = 32*u + (lane&31)], U = C/32
template <int C>
__device__ __forceinline__ void tail_fwd_compute_ldsw(const f32x4 (&a)[C / 8], int q0, const float* sWf, float* sP, int NKP, int lane) {
    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < C / 8; ++kk) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][e], w[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * fh;
        sP[(q0 + r) * NKP + frow] = acc[e];
    }
}

template <int C, int COUT, int GWP, int NBATCH = 3>
__device__ __forceinline__ void tail_bwd_tile_ldsw(const float* sg, int gbase, bool valid, const float* sWb, f32x16 (&acc)[C / 32], int lane) {
    constexpr int NK = 25 * COUT, NS = (NK + 1) / 2, U = C / 32;
    const int fh = lane >> 5;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
    // The operand reads are issued in three batches, each waited for once: left to itself hipcc sinks every read to its use
    // (ds_read; s_waitcnt lgkmcnt(0); mfma -- 2 * NS exposed LDS round trips per tile, 2/3 of this phase's time).  Bigger
    // batches need more registers than the pipelined kernel (at its 168-VGPR cap) has: two batches already spill.  (The
    // third-generation kernel's backward waves hold nothing else: one batch, NBATCH = 1.)
    constexpr int BATCH = (NS + NBATCH - 1) / NBATCH;
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += BATCH) {
        float gv[BATCH];
        float w[BATCH][U];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int s = s0 + q < NS ? s0 + q : NS - 1;
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int t0 = k0 / COUT, c0 = k0 % COUT, t1 = (k1 < NK ? k1 : k0) / COUT, c1 = (k1 < NK ? k1 : k0) % COUT;
            const int off0 = ((t0 / 5) * GWP + (t0 % 5)) * COUT + c0;
            const int off1 = ((t1 / 5) * GWP + (t1 % 5)) * COUT + c1;
            gv[q] = sg[gbase + (fh ? off1 : off0)];
#pragma unroll
            for (int u = 0; u < U; ++u) w[q][u] = sWb[(s * 64 + lane) * U + u];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            asm volatile("" : "+v"(gv[q]));
#pragma unroll
            for (int u = 0; u < U; ++u) asm volatile("" : "+v"(w[q][u]));
        }
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            if (s0 + q >= NS) continue;
            const float g = valid ? gv[q] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(g, w[q][u], acc[u], 0, 0, 0);
        }
    }
}

// =================================================================================================================
// MNIST: Generator.5 (C -> 1, 14x14 -> 28x28) + sigmoid + loss + backward, one workgroup per latent row
// =================================================================================================================

}  // namespace dg
