// MFMA formulation of the generator tails (last Deconv2D with Cout = 1 / 3, output nonlinearity, loss, and the
// backward to the last GEMM activation), gfx950.
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
#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// Same fragments from the FRAGMENT-ORDER pack built once at weight-load time (dg_engine.cpp, pack_tail_fragments):
// Wp[((t*C/8 + kk)*64 + lane)*4 + e] -- one fully coalesced 1 KB line run per load instead of 32 partial lines.
template <int C>
__device__ __forceinline__ void tail_load_fwd_weights_packed(const float* __restrict__ Wp, f32x4 (&w)[1][C / 8], int t, int lane) {
#pragma unroll
    for (int kk = 0; kk < C / 8; ++kk)
        w[0][kk] = *reinterpret_cast<const f32x4*>(Wp + (((long long)t * (C / 8) + kk) * 64 + lane) * 4);
}

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
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // kappa = 2s + fh -> (tap = kappa / COUT, co = kappa % COUT), tap -> (kh, kw)
        constexpr int dummy = 0; (void)dummy;
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int t0 = k0 / COUT, c0 = k0 % COUT, t1 = (k1 < NK ? k1 : k0) / COUT, c1 = (k1 < NK ? k1 : k0) % COUT;
        const int off0 = ((t0 / 5) * GWP + (t0 % 5)) * COUT + c0;
        const int off1 = ((t1 / 5) * GWP + (t1 % 5)) * COUT + c1;
        float gv = sg[gbase + (fh ? off1 : off0)];
        if (!valid) gv = 0.f;
#pragma unroll
        for (int u = 0; u < C / 32; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, bw.w[u][s], acc[u], 0, 0, 0);
    }
}

// =================================================================================================================
// MNIST: Generator.5 (C -> 1, 14x14 -> 28x28) + sigmoid + loss + backward, one workgroup per latent row
// =================================================================================================================
constexpr int MN_NKP = 35;       // P row pitch (32 kappa columns + 3 pad: gather reads <= 2-way bank conflicted)
constexpr int MN_GWP = 32;       // da5 image pitch; rows/cols are image index + 1, 31 used
constexpr int MN_GR = 31;

template <int C>
__global__ __launch_bounds__(256) void mnist_tail_mfma_kernel(MnistTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sP = reinterpret_cast<float*>(smem);                  // [224][MN_NKP]
    float* sg = sP + 224 * MN_NKP;                               // [31][32]
    unsigned* smask = reinterpret_cast<unsigned*>(sg + MN_GR * MN_GWP);   // [224][C/32] ReluGrad bits
    float* sred = reinterpret_cast<float*>(smask + 224 * (C / 32));     // [4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const int n = blockIdx.x;
    const int b = n / a.R;
    float* hrow = a.h3 + (long long)n * (196 * C);

    for (int i = tid; i < MN_GR * MN_GWP; i += 256) sg[i] = 0.f;

    // ---- forward GEMM: 7 tiles of 32 positions (196 valid); wave w owns tiles w and w+4, both loaded up front ----
    {
        f32x4 w[1][C / 8];
        tail_load_fwd_weights<C, 1, 1>(a.F5, w, 0, lane);
        const int q0 = wave * 32 + frow, q1 = q0 + 128;
        f32x4 a0[C / 8], a1[C / 8];
        tail_fwd_load<C>(hrow, q0 < 196 ? q0 : -1, a0, lane);
        const bool second = wave < 3;
        if (second) tail_fwd_load<C>(hrow, q1 < 196 ? q1 : -1, a1, lane);
        // mask word layout: word (c >> 5) of a position holds channels 32*(c>>5) .. +31; this lane covers channels
        // 8*kk + 4*fh + e: bit ((c & 31)) of word c >> 5
        auto store_mask = [&](const f32x4 (&av)[C / 8], int q) {
#pragma unroll
            for (int wd = 0; wd < C / 32; ++wd) {
                unsigned m = 0;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m |= (av[wd * 4 + k4][e] > 0.f ? 1u : 0u) << (k4 * 8 + fh * 4 + e);
                m |= __shfl_xor(m, 32, 64);            // the two lane halves hold disjoint bits of the same word
                if (fh == 0) smask[q * (C / 32) + wd] = m;
            }
        };
        store_mask(a0, q0);
        if (a.dbg != 2) tail_fwd_compute<C, 1>(a0, wave * 32, w, sP, MN_NKP, 0, lane);
        if (second) {
            store_mask(a1, q1);
            if (a.dbg != 2) tail_fwd_compute<C, 1>(a1, wave * 32 + 128, w, sP, MN_NKP, 0, lane);
        }
    }
    __syncthreads();

    // ---- gather (taps of matching parity only) + sigmoid + loss + da5 ------------------------------------------------
    const float* xrow = a.x + (long long)b * 784;
    const float bias = a.b5[0];
    const float gscale = 2.0f / 784.0f;
    float sq = 0.f;
    for (int p = tid; p < (a.dbg == 1 ? 0 : 784); p += 256) {
        const int i = p / 28, j = p - i * 28;
        const int kh0 = (i + 1) & 1, kw0 = (j + 1) & 1;
        float s = 0.f;
#pragma unroll
        for (int ah = 0; ah < 3; ++ah) {
            const int kh = kh0 + 2 * ah;
            const int oh = (i + 1 - kh) >> 1;
            if (kh > 4 || oh < 0 || oh >= 14) continue;
#pragma unroll
            for (int aw = 0; aw < 3; ++aw) {
                const int kw = kw0 + 2 * aw;
                const int ow = (j + 1 - kw) >> 1;
                if (kw > 4 || ow < 0 || ow >= 14) continue;
                s += sP[(oh * 14 + ow) * MN_NKP + kh * 5 + kw];
            }
        }
        const float pre = s + bias;
        const float y = 1.0f / (1.0f + expf(-pre));
        const float d = y - xrow[p];
        sq = __builtin_fmaf(d, d, sq);
        sg[(i + 1) * MN_GWP + (j + 1)] = gscale * d * y * (1.0f - y);
        if (a.y) a.y[(long long)n * 784 + p] = y;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
    if (lane == 0) sred[wave] = sq;
    __syncthreads();
    if (tid == 0) a.loss[n] = ((sred[0] + sred[1]) + (sred[2] + sred[3])) * (1.0f / 784.0f);
    if (!a.do_backward) return;

    // ---- backward GEMM + ReluGrad (mask bits from LDS), in place over h3 ---------------------------------------------
    BwdWeights<C, 1, MN_GWP> bw;
    bw.load(a.F5, lane);
    for (int mt = wave; mt < (a.dbg == 3 ? 0 : 7); mt += 4) {
        const int q = mt * 32 + frow;
        const bool valid = q < 196;
        const int qq = valid ? q : 0;
        const int oh = qq / 14, ow = qq - oh * 14;
        f32x16 acc[C / 32];
        tail_bwd_tile<C, 1, MN_GWP>(sg, (2 * oh) * MN_GWP + 2 * ow, valid, bw, acc, lane);
#pragma unroll
        for (int u = 0; u < C / 32; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qr = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const unsigned mw = smask[qr * (C / 32) + u];
                if (qr < 196) hrow[qr * C + u * 32 + frow] = ((mw >> frow) & 1u) ? acc[u][e] : 0.f;
            }
    }
}

void launch_mnist_tail_mfma(const MnistTailArgs& a, hipStream_t s) {
    const int lds = (224 * MN_NKP + MN_GR * MN_GWP + 224 * (a.C / 32) + 4) * 4;
    if (a.C == 64) hipLaunchKernelGGL((mnist_tail_mfma_kernel<64>), dim3(a.n_rows), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mnist_tail_mfma_kernel<128>), dim3(a.n_rows), dim3(256), lds, s, a);
}

// =================================================================================================================
// CelebA: Generator.6 (C -> 3, 32x32 -> 64x64) + tanh + loss ; backward to da5 (no ReluGrad: Generator.5 is linear)
// =================================================================================================================
constexpr int CE_NKP = 99;       // 75 kappa columns padded to 96 (+3: gather reads are <= 2-way bank conflicted)
constexpr int CE_GWP = 68;       // da6 image pitch (cols are image index + 1, 67 used)
constexpr int CE_GROWS = 11;

// One workgroup per (latent row, band of 8 output rows); 6 waves, wave w owns local input row w (4 + 2 halo).
// Measured alternatives (profiles/r01 notes): fragment-shaped global loads of H (slower than the LDS-DMA staging
// below), a persistent variant with register prefetch of the next band (slower: both resident workgroups run in
// lockstep), filters re-read per tile in reference layout (slower than the fragment-order pack).
template <int C>
__global__ __launch_bounds__(384) void celeba_tail_fwd_mfma_kernel(CelebaTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sP = reinterpret_cast<float*>(smem);                  // [192][CE_NKP]: 6 input rows x 32 positions
    float* sred = sP + 192 * CE_NKP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x >> 3, band = blockIdx.x & 7;
    const int b = n / a.R;
    const float* hrow = a.h5 + (long long)n * (1024 * C);
    const int oh_lo = 4 * band - 1;                              // local input row lr <-> oh_lo + lr
    if (a.dbg != 2) {
        const int oh = oh_lo + wave;
        const bool in_img = oh >= 0 && oh < 32;
        // The row's 32 positions x C floats are one contiguous 8 KB (C = 64) run: stage it with full-line LDS-DMA into
        // this wave's own P region (overwritten by P only after the fragments are in registers).  The 16-B chunk
        // index is XOR-swizzled with the position (on the source side) so the b128 fragment reads are conflict free.
        f32x4 av[C / 8];
        {
            char* stage = reinterpret_cast<char*>(sP + wave * 32 * CE_NKP);
            constexpr int CH = C / 4;                          // 16-B chunks per position
            constexpr int NI = 32 * CH / 64;                    // DMA instructions per tile
            if (in_img) {
                const char* src = reinterpret_cast<const char*>(hrow + (long long)oh * 32 * C);
#pragma unroll
                for (int q = 0; q < NI; ++q) {
                    const int slot = q * 64 + lane;
                    const int pos = slot / CH, c = slot % CH;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4)),
                        (__attribute__((address_space(3))) void*)(stage + q * 1024), 16, 0, 0);
                }
            }
        }
        // Halo rows only feed part of the band's outputs: the top halo row (local 0) reaches them through kh >= 3
        // (kappa >= 45: tiles 1, 2), the bottom one (local 5) through kh = 0 (kappa < 15: tile 0).
        const int t_lo = wave == 0 ? 1 : 0, t_hi = wave == 5 ? 1 : 3;
        f32x4 w0[1][C / 8], w1[1][C / 8];
        tail_load_fwd_weights_packed<C>(a.F6p, w0, t_lo, lane);
        if (t_lo + 1 < t_hi) tail_load_fwd_weights_packed<C>(a.F6p, w1, t_lo + 1, lane);
        {
            const char* stage = reinterpret_cast<const char*>(sP + wave * 32 * CE_NKP);
            constexpr int CH = C / 4;
            const int frow = lane & 31, fh = lane >> 5;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < C / 8; ++kk) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (in_img) v = *reinterpret_cast<const f32x4*>(stage + frow * (C * 4) + (((kk * 2 + fh) ^ (frow & (CH - 1))) << 4));
                av[kk] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        tail_fwd_compute<C, 1>(av, wave * 32, w0, sP, CE_NKP, t_lo * 32, lane);
        if (t_lo + 1 < t_hi) {
            if (t_lo + 2 < t_hi) tail_load_fwd_weights_packed<C>(a.F6p, w0, t_lo + 2, lane);
            tail_fwd_compute<C, 1>(av, wave * 32, w1, sP, CE_NKP, (t_lo + 1) * 32, lane);
            if (t_lo + 2 < t_hi) tail_fwd_compute<C, 1>(av, wave * 32, w0, sP, CE_NKP, (t_lo + 2) * 32, lane);
        }
    }
    __syncthreads();
    const float* xrow = a.x + (long long)b * 12288;
    float* grow = a.g6 + (long long)n * 12288;
    float* yrow = a.y ? a.y + (long long)n * 12288 : nullptr;
    const float gscale = 2.0f / 12288.0f;
    float sq = 0.f;
    if (a.dbg != 1) {
        float sum[4], xv[4];
        int oidx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = tid + r * 384;                           // 1536 outputs = 4 x 384 threads
            const int co = p % 3, pix = p / 3;
            const int il = pix >> 6, j = pix & 63;
            const int i = 8 * band + il;
            const int kh0 = (i + 1) & 1, kw0 = (j + 1) & 1;
            float sacc = 0.f;
#pragma unroll
            for (int ah = 0; ah < 3; ++ah) {
                const int kh = kh0 + 2 * ah;
                const int oh = (i + 1 - kh) >> 1;
                if (kh > 4 || oh < 0 || oh >= 32) continue;
                const int lr = oh - oh_lo;
#pragma unroll
                for (int aw = 0; aw < 3; ++aw) {
                    const int kw = kw0 + 2 * aw;
                    const int ow = (j + 1 - kw) >> 1;
                    if (kw > 4 || ow < 0 || ow >= 32) continue;
                    sacc += sP[(lr * 32 + ow) * CE_NKP + (kh * 5 + kw) * 3 + co];
                }
            }
            oidx[r] = (i * 64 + j) * 3 + co;
            sum[r] = sacc + a.b6[co];
            xv[r] = xrow[oidx[r]];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float y = tanhf(sum[r]);
            const float d = y - xv[r];
            sq = __builtin_fmaf(d, d, sq);
            grow[oidx[r]] = gscale * d * (1.0f - y * y);
            if (yrow) yrow[oidx[r]] = y;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
    if (lane == 0) sred[wave] = sq;
    __syncthreads();
    if (tid == 0) a.loss_part[(long long)n * 8 + band] = ((sred[0] + sred[1]) + (sred[2] + sred[3])) + (sred[4] + sred[5]);
}

template <int C>
__global__ __launch_bounds__(256) void celeba_tail_bwd_mfma_kernel(CelebaTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sg = reinterpret_cast<float*>(smem);                  // [11][68][3]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x >> 3, band = blockIdx.x & 7;
    const float* grow = a.g6 + (long long)n * 12288;
    const int i_lo = 8 * band - 1;
    for (int i = tid; i < CE_GROWS * CE_GWP * 3; i += 256) {
        const int co = i % 3, rc = i / 3;
        const int lr = rc / CE_GWP, lc = rc - lr * CE_GWP;
        const int ii = i_lo + lr, jj = lc - 1;
        sg[i] = (ii >= 0 && ii < 64 && jj >= 0 && jj < 64) ? grow[(ii * 64 + jj) * 3 + co] : 0.f;
    }
    BwdWeights<C, 3, CE_GWP> bw;
    bw.load(a.F6, lane);
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    float* hrow = a.h5 + (long long)n * (1024 * C);
    // 4 position tiles: local input row ohl = wave, 32 positions each
    {
        const int ohl = wave, ow = frow;
        f32x16 acc[C / 32];
        tail_bwd_tile<C, 3, CE_GWP>(sg, ((2 * ohl) * CE_GWP + 2 * ow) * 3, true, bw, acc, lane);
        const int oh = 4 * band + ohl;
#pragma unroll
        for (int u = 0; u < C / 32; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int owr = (e & 3) + 8 * (e >> 2) + 4 * fh;
                hrow[(oh * 32 + owr) * C + u * 32 + frow] = acc[u][e];
            }
    }
}

void launch_celeba_tail_fwd_mfma(const CelebaTailArgs& a, hipStream_t s) {
    const int lds = (192 * CE_NKP + 8) * 4;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_mfma_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    if (a.C == 64) hipLaunchKernelGGL((celeba_tail_fwd_mfma_kernel<64>), dim3(a.n_rows * 8), dim3(384), lds, s, a);
    else hipLaunchKernelGGL((celeba_tail_fwd_mfma_kernel<128>), dim3(a.n_rows * 8), dim3(384), lds, s, a);
}

void launch_celeba_tail_bwd_mfma(const CelebaTailArgs& a, hipStream_t s) {
    const int lds = CE_GROWS * CE_GWP * 3 * 4;
    if (a.C == 64) hipLaunchKernelGGL((celeba_tail_bwd_mfma_kernel<64>), dim3(a.n_rows * 8), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((celeba_tail_bwd_mfma_kernel<128>), dim3(a.n_rows * 8), dim3(256), lds, s, a);
}

}  // namespace dg
