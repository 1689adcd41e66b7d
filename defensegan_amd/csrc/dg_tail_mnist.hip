// MNIST / F-MNIST generator tail (Generator.5: 64 -> 1 channels, 28 x 28, sigmoid, loss, backward to da3), gfx950.
// Formulation and the shared device helpers: dg_tail_common.h.  Reference: models/dataset_models.py:66-69, models/gan.py:410-414.
#include "dg_tail_common.h"

namespace dg {

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
        if (DG_DBG(a) != 2) tail_fwd_compute<C, 1>(a0, wave * 32, w, sP, MN_NKP, 0, lane);
        if (second) {
            store_mask(a1, q1);
            if (DG_DBG(a) != 2) tail_fwd_compute<C, 1>(a1, wave * 32 + 128, w, sP, MN_NKP, 0, lane);
        }
    }
    __syncthreads();

    // ---- gather (taps of matching parity only) + sigmoid + loss + da5 ------------------------------------------------
    const float* xrow = a.x + (long long)b * 784;
    const float bias = a.b5[0];
    const float gscale = 2.0f / 784.0f;
    float sq = 0.f;
    for (int p = tid; p < (DG_DBG(a) == 1 ? 0 : 784); p += 256) {
        const int i = p / 28, j = p - i * 28;
        const int kh0 = (i + 1) & 1, kw0 = (j + 1) & 1;
        float s = 0.f;
        // all 9 candidate taps are read unconditionally (no divergent LDS round trips): a tap that does not exist
        // reads P[0][31], a zero-filter pad column
#pragma unroll
        for (int ah = 0; ah < 3; ++ah) {
            const int kh = kh0 + 2 * ah;
            const int oh = (i + 1 - kh) >> 1;
            const bool okh = !(kh > 4 || oh < 0 || oh >= 14);
#pragma unroll
            for (int aw = 0; aw < 3; ++aw) {
                const int kw = kw0 + 2 * aw;
                const int ow = (j + 1 - kw) >> 1;
                const bool ok = okh && !(kw > 4 || ow < 0 || ow >= 14);
                s += sP[ok ? (oh * 14 + ow) * MN_NKP + kh * 5 + kw : 31];
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
    for (int mt = wave; mt < (DG_DBG(a) == 3 ? 0 : 7); mt += 4) {
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

// ---- pipelined variant: one persistent 12-wave workgroup per CU, MFMA waves and gather waves work on different rows ----
// The fused kernel above runs load -> forward GEMM -> gather -> backward GEMM -> store strictly in sequence per latent
// row; the MFMA pipe idles during the gather and both memory phases.  Here waves 0-7 ("M", tile = wave, tile 7 idle)
// own the two GEMMs and waves 8-11 ("G") own the gather/sigmoid/loss.  In step t (one barrier per step)
//     M:  A fragments of row t+1 from the wave's LDS stage, LDS-DMA of row t+2 into it | backward GEMM + masked store of
//         row t-1 | forward GEMM of row t+1
//     G:  gather + sigmoid + loss + da5 image of row t
// with P, the da5 image and the ReluGrad bits double-buffered by row parity, so the three stages of three consecutive
// rows overlap and the A fragments have a whole step to arrive.  Rows of a workgroup: blockIdx.x + k * gridDim.x.
// (Round 3: the barriers here are __syncthreads(), whose fence also waits for vmcnt(0) -- the row stores' acknowledgements and
// the DMA of row t + 2; LDS-only barriers (lds_barrier) measured the same, 66.2 vs 66.5 us: by the end of a step they have arrived.)
// Measured (tools/tail_trace_mnist.py, N = 2560): 86 -> 78 us; a step is ~17 k cycles of which the 8 DMA instructions
// take 2.4 k and the 32 row stores ~3 k to ISSUE (the memory pipes are saturated in bursts: 100 KB per CU per step);
// spreading them between the forward MFMA groups made it worse (125 us: every stalled VMEM issue then blocks MFMAs);
// offsetting the workgroups' start times to de-phase the bursts changed nothing.
template <int C>
__global__ __launch_bounds__(768) void mnist_tail_pipe_kernel(MnistTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PSZ = 224 * MN_NKP, GSZ = MN_GR * MN_GWP, MSZ = 224 * (C / 32);
    float* sP = reinterpret_cast<float*>(smem);                  // [2][224][MN_NKP]
    float* sg = sP + 2 * PSZ;                                    // [2][31][32]
    unsigned* smask = reinterpret_cast<unsigned*>(sg + 2 * GSZ); // [2][224][C/32]
    float* sred = reinterpret_cast<float*>(smask + 2 * MSZ);     // [2][4]
    float* sWf = sred + 8;                                       // forward filter fragments [C/8][64][4]
    float* sWb = sWf + (C / 8) * 256;                            // backward filter fragments [13][64][C/32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const bool mrole = wave < 8;
    const bool mwork = wave < 7;                                 // 7 position tiles
    char* stage = reinterpret_cast<char*>(sWb + 13 * 64 * (C / 32)) + (wave & 7) * (32 * C * 4);   // this M wave's A tile [32][C], LDS-DMA target
    const int tile = wave;
    const int gt = tid - 512;                                    // gather thread id (G waves)
    const int gw = wave - 8;
    const int n_my = ((int)a.n_rows - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto row_of = [&](int k) { return (long long)blockIdx.x + (long long)k * gridDim.x; };

    for (int i = tid; i < 2 * GSZ; i += 768) sg[i] = 0.f;       // zero borders of both da5 images, written once

    for (int i = tid; i < (C / 8) * 256; i += 768) {
        const int e = i & 3, l = (i >> 2) & 63, kk = i >> 8;
        const int kappa = l & 31, c = kk * 8 + (l >> 5) * 4 + e;
        sWf[i] = kappa < 25 ? a.F5[kappa * C + c] : 0.f;
    }
    for (int i = tid; i < 13 * 64 * (C / 32); i += 768) {
        const int u = i % (C / 32), l = (i / (C / 32)) & 63, st = i / (64 * (C / 32));
        const int kappa = 2 * st + (l >> 5);
        sWb[i] = kappa < 25 ? a.F5[kappa * C + u * 32 + (l & 31)] : 0.f;
    }
    const int q = tile * 32 + frow;                              // this lane's position in the M role
    const bool qvalid = q < 196;
    // A tile of row k: 32 positions x C floats = one contiguous 8 KB run, staged with full-line LDS-DMA into this wave's
    // private region (16-B chunk index XOR-swizzled with the position on the source side: conflict-free b128 reads).
    constexpr int CH = C / 4;                                    // 16-B chunks per position
    constexpr int NI = 32 * CH / 64;                             // DMA instructions per full tile
    auto stage_row = [&](int k) {
        const char* src = reinterpret_cast<const char*>(a.h3 + row_of(k) * (196 * C) + (long long)tile * 32 * C);
        const int ni = tile < 6 ? NI : (196 - 192) * CH / 64;    // the last tile holds 4 positions
#pragma unroll
        for (int qi = 0; qi < NI; ++qi) {
            if (qi >= ni) break;
            const int slot = qi * 64 + lane;
            const int pos = slot / CH, c = slot % CH;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4)),
                (__attribute__((address_space(3))) void*)(stage + qi * 1024), 16, 0, 0);
        }
    };
    // younger_stores: the wave issued its 32 backward row stores AFTER the DMA being waited for.  VMEM operations retire
    // in order, so the DMA has landed once at most those 32 are outstanding -- waiting for vmcnt(0) would also wait for
    // the stores' HBM acknowledgements.  (Tile 6 issues fewer stores: it waits for everything.)
    auto read_frags = [&](f32x4 (&av)[C / 8], bool younger_stores) {
        if (younger_stores && tile < 6) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < C / 8; ++kk) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (qvalid) v = *reinterpret_cast<const f32x4*>(stage + frow * (C * 4) + (((kk * 2 + fh) ^ (frow & (CH - 1))) << 4));
            av[kk] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // ReluGrad bits, transposed: word mk[tile*C + c] has bit p set iff h3[position 32*tile + p][channel c] > 0.  A wave
    // ballot over "fragment element > 0" is exactly two such words (lanes 0-31 = the 32 positions for channel 8kk+e,
    // lanes 32-63 for channel 8kk+4+e), and the backward epilogue needs one word per accumulator block instead of 16.
    auto fwd = [&](int k, const f32x4 (&av)[C / 8]) {
        unsigned* mk = smask + (k & 1) * MSZ + tile * C;
        // word 8*kk + 4*half + e of the tile's C words = one half of a ballot; each is dropped into the lane that will store
        // it (v_writelane), then ONE ds_write_b32 per 64 words (a per-ballot "if (lane < 2) store" costs a divergent
        // branch and a store instruction for each of the C/2 ballots).  A VALU-written SGPR (v_cmp) is NOT safe as the data
        // operand of a v_writelane issued right behind it on gfx950 (wrong masks without wait states; measured): the four
        // ballots of a k-step are formed first, then 4 wait states, then their eight writes.
        static_assert(C % 64 == 0, "C words per tile in groups of 64");
#pragma unroll
        for (int w0 = 0; w0 < C; w0 += 64) {
            int word = 0;
#pragma unroll
            for (int kk = w0 / 8; kk < w0 / 8 + 8; ++kk) {
                unsigned lo[4], hi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned long long bal = __ballot(av[kk][e] > 0.f);
                    lo[e] = __builtin_amdgcn_readfirstlane((unsigned)bal);
                    hi[e] = __builtin_amdgcn_readfirstlane((unsigned)(bal >> 32));
                }
                asm volatile("s_nop 3\n\t"
                             "v_writelane_b32 %0, %1, %9\n\tv_writelane_b32 %0, %2, %10\n\t"
                             "v_writelane_b32 %0, %3, %11\n\tv_writelane_b32 %0, %4, %12\n\t"
                             "v_writelane_b32 %0, %5, %13\n\tv_writelane_b32 %0, %6, %14\n\t"
                             "v_writelane_b32 %0, %7, %15\n\tv_writelane_b32 %0, %8, %16"
                             : "+v"(word)
                             : "s"(lo[0]), "s"(lo[1]), "s"(lo[2]), "s"(lo[3]), "s"(hi[0]), "s"(hi[1]), "s"(hi[2]), "s"(hi[3]),
                               "i"(8 * kk - w0), "i"(8 * kk + 1 - w0), "i"(8 * kk + 2 - w0), "i"(8 * kk + 3 - w0),
                               "i"(8 * kk + 4 - w0), "i"(8 * kk + 5 - w0), "i"(8 * kk + 6 - w0), "i"(8 * kk + 7 - w0));
            }
            mk[w0 + lane] = (unsigned)word;
        }
        tail_fwd_compute_ldsw<C>(av, tile * 32, sWf, sP + (k & 1) * PSZ, MN_NKP, lane);
    };
    auto bwd = [&](int k) {
        const unsigned* mk = smask + (k & 1) * MSZ + tile * C;
        float* hrow = a.h3 + row_of(k) * (196 * C);
        const int qq = qvalid ? q : 0;
        const int oh = qq / 14, ow = qq - oh * 14;
        f32x16 acc[C / 32];
        tail_bwd_tile_ldsw<C, 1, MN_GWP>(sg + (k & 1) * GSZ, (2 * oh) * MN_GWP + 2 * ow, qvalid, sWb, acc, lane);
        // masked results go through this wave's tile of the P buffer of the same parity (free until the forward GEMM later
        // in this step rewrites it) so that a lane owns 4 consecutive channels: 4 b128 row stores per 32-channel block
        // instead of 16 dword ones
        float* tb = sP + (k & 1) * PSZ + tile * 32 * MN_NKP;         // >= 32 x 32 floats
        const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
        for (int u = 0; u < C / 32; ++u) {
            const unsigned mw = mk[u * 32 + frow];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = (e & 3) + 8 * (e >> 2) + 4 * fh;
                tb[pr * 32 + frow] = ((mw >> pr) & 1u) ? acc[u][e] : 0.f;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int qr = tile * 32 + p * 8 + er;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
                if (qr < 196) *reinterpret_cast<f32x4*>(hrow + qr * C + u * 32 + ec) = v;
            }
        }
    };
    const float bias = a.b5[0];
    const float gscale = 2.0f / 784.0f;
    auto load_x = [&](int k, float (&xv)[4]) {
        const float* xrow = a.x + (long long)((unsigned)row_of(k) / (unsigned)a.R) * 784;     // rows < 2^24: 32-bit division
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = gt + 256 * r;
            xv[r] = p < 784 ? xrow[p] : 0.f;
        }
    };
    auto gather = [&](int k, const float (&xv)[4]) {
        const float* pP = sP + (k & 1) * PSZ;
        float* pg = sg + (k & 1) * GSZ;
        const long long n = row_of(k);
        float sq = 0.f;
        // all 9 candidate taps of this thread's (up to) 4 pixels are read first and waited for once (hipcc otherwise reads,
        // waits and sums pixel by pixel: 4 exposed LDS round trips under the M waves' LDS traffic); a missing tap reads the
        // zero pad column 31.  The sums keep their order.
        float tv[4][9];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = gt + 256 * r;
            const int i = p / 28, j = p - i * 28;
            const int kh0 = (i + 1) & 1, kw0 = (j + 1) & 1;
#pragma unroll
            for (int ah = 0; ah < 3; ++ah) {
                const int kh = kh0 + 2 * ah;
                const int oh = (i + 1 - kh) >> 1;
                const bool okh = p < 784 && !(kh > 4 || oh < 0 || oh >= 14);
#pragma unroll
                for (int aw = 0; aw < 3; ++aw) {
                    const int kw = kw0 + 2 * aw;
                    const int ow = (j + 1 - kw) >> 1;
                    const bool ok = okh && !(kw > 4 || ow < 0 || ow >= 14);
                    tv[r][ah * 3 + aw] = pP[ok ? (oh * 14 + ow) * MN_NKP + kh * 5 + kw : 31];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(tv[r][t]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = gt + 256 * r;
            if (p >= 784) break;
            const int i = p / 28, j = p - i * 28;
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) sacc += tv[r][t];
            const float y = 1.0f / (1.0f + expf(-(sacc + bias)));
            const float d = y - xv[r];
            sq = __builtin_fmaf(d, d, sq);
            pg[(i + 1) * MN_GWP + (j + 1)] = gscale * d * y * (1.0f - y);
            if (a.y) a.y[n * 784 + p] = y;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        if (lane == 0) sred[(k & 1) * 4 + gw] = sq;
    };
    auto finish_loss = [&](int k) {       // after the barrier that follows gather(k)
        const float* r4 = sred + (k & 1) * 4;
        a.loss[row_of(k)] = ((r4[0] + r4[1]) + (r4[2] + r4[3])) * (1.0f / 784.0f);
    };

    // ---- the two roles run separate loops with the same barrier sequence: sg-zero | prologue | one per step ------------
    if (mrole) {
        f32x4 A[C / 8];
        if (mwork) stage_row(0);
        __syncthreads();                                         // sg zeroed, filter fragments in LDS
        if (mwork) {
            read_frags(A, false);
            if (n_my > 1) stage_row(1);
            fwd(0, A);
        }
        __syncthreads();
        const bool tr = DG_TRACE_PTR(a) != nullptr && wave == 0;
        long long ph[5] = {0, 0, 0, 0, 0};
        const long long tb = tr ? (long long)__builtin_readcyclecounter() : 0, wb = tr ? (long long)wall_clock64() : 0;
        for (int t = 0; t <= n_my; ++t) {
            long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
            if (tr) c0 = (long long)__builtin_readcyclecounter();
            if (mwork) {
                if (t + 1 < n_my) {
                    read_frags(A, t >= 2);                       // row t+1 (staged one step ago, before bwd(t-2)'s stores)
                    if (tr) c1 = (long long)__builtin_readcyclecounter();
                    if (t + 2 < n_my) stage_row(t + 2);          // lands during this step's two GEMMs
                }
                if (tr) c2 = (long long)__builtin_readcyclecounter();
                if (t >= 1) bwd(t - 1);
                if (tr) c3 = (long long)__builtin_readcyclecounter();
                if (t + 1 < n_my) fwd(t + 1, A);
                if (tr) c4 = (long long)__builtin_readcyclecounter();
            }
            __syncthreads();
            if (tr && t >= 2 && t + 2 < n_my) {
                ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3;
                ph[4] += (long long)__builtin_readcyclecounter() - c4;
            }
        }
        if (tr && lane == 0 && blockIdx.x < 2048) {
            long long* o = DG_TRACE_PTR(a) + (long long)blockIdx.x * 16;
            for (int i = 0; i < 5; ++i) o[i] = ph[i];
            o[5] = n_my > 4 ? n_my - 4 : 0;
            o[6] = (long long)__builtin_readcyclecounter() - tb;
            o[7] = (long long)wall_clock64() - wb;
        }
    } else {
        float xv0[4], xv1[4];
        load_x(0, xv0);
        __syncthreads();
        __syncthreads();
        const bool tr = DG_TRACE_PTR(a) != nullptr && wave == 8;
        long long gph[2] = {0, 0};
        auto step = [&](int t, float (&xc)[4], float (&xn)[4]) {
            long long c0 = 0, c1 = 0;
            if (tr) c0 = (long long)__builtin_readcyclecounter();
            if (t >= 1 && gt == 0) finish_loss(t - 1);
            if (t < n_my) {
                if (t + 1 < n_my) load_x(t + 1, xn);
                gather(t, xc);
            }
            if (tr) c1 = (long long)__builtin_readcyclecounter();
            __syncthreads();
            if (tr && t >= 2 && t + 2 < n_my) { gph[0] += c1 - c0; gph[1] += (long long)__builtin_readcyclecounter() - c1; }
        };
        for (int t = 0; t <= n_my; t += 2) {
            step(t, xv0, xv1);
            if (t + 1 <= n_my) step(t + 1, xv1, xv0);
        }
        if (tr && lane == 0 && blockIdx.x < 2048) {
            long long* o = DG_TRACE_PTR(a) + (long long)blockIdx.x * 16;
            o[8] = gph[0];
            o[9] = gph[1];
        }
    }
}


#ifdef DG_MEASURE   // superseded by the third generation: a cross-check only (option tail_pipe_version = 2 needs the measurement build)
// ---- pipelined variant, second generation: the matrix work levelled over the four SIMDs ---------------------------------
// In mnist_tail_pipe_kernel wave w < 7 owns position tile w for both GEMMs: waves (0,4), (1,5), (2,6) share a SIMD, so three
// SIMDs carry two tiles = 116 MFMAs per step (7.4 k cycles of matrix pipe) and the fourth one tile -- and tile 6 is 4 real
// positions (192..195) padded to 32.  The M waves are the step's critical path (~11 k of 13 k cycles, tools/tail_trace_mnist.py),
// mostly waiting for each other's MFMAs.  Here
//   waves 0-3   forward + backward of tile w                     (58 MFMAs)
//   waves 4, 5  forward of tile w only                           (32)
//   waves 6, 7  backward of tiles 4, 5 (masks, da5 image and filter fragments are in LDS: any wave can do it; the masked
//               tile goes through a 4 KB scratch in the wave's own, otherwise unused staging region)          (26)
//   the 4 positions of "tile 6" leave the matrix pipe: the gather waves compute their 4 x 25 P entries, their ReluGrad bits and
//   their 4 x 64 gradients with v_fma chains in the MFMA's k order (bit-identical: an MFMA is a k-ordered fma chain), from a
//   1 KB image one of them stages by LDS-DMA two rows ahead
// so every SIMD carries 84-90 MFMAs per step (5.8 k cycles) and no MFMA is spent on padding rows.  Same barrier sequence, same
// buffers and the same arithmetic per element as mnist_tail_pipe_kernel (tests/test_gpu_variants.py: bit-identical).
template <int C>
__global__ __launch_bounds__(768) void mnist_tail_pipe2_kernel(MnistTailArgs a) {
    static_assert(C == 64, "64 channels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PSZ = 224 * MN_NKP, GSZ = MN_GR * MN_GWP, MSZ = 224 * (C / 32);
    float* sP = reinterpret_cast<float*>(smem);                  // [2][224][MN_NKP]
    float* sg = sP + 2 * PSZ;                                    // [2][31][32]
    unsigned* smask = reinterpret_cast<unsigned*>(sg + 2 * GSZ); // [2][224][C/32]: tiles 0-5
    float* sred = reinterpret_cast<float*>(smask + 2 * MSZ);     // [2][4]
    float* sWf = sred + 8;                                       // forward filter fragments [C/8][64][4]
    float* sWb = sWf + (C / 8) * 256;                            // backward filter fragments [13][64][C/32]
    char* stages = reinterpret_cast<char*>(sWb + 13 * 64 * (C / 32));     // [8 waves][8 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const bool mrole = wave < 8;
    const bool has_fwd = wave < 6;                               // forward GEMM of tile `wave`
    const bool has_bwd = wave < 4 || wave == 6 || wave == 7;     // backward GEMM of tile btile
    const int tile = wave;                                       // forward tile
    const int btile = wave < 4 ? wave : wave - 2;                // backward tile (waves 6, 7 -> tiles 4, 5)
    char* stage = stages + (wave & 7) * (32 * C * 4);            // waves 0-5: A tile [32][C], LDS-DMA target
    // wave 6's region: [2][4 positions][C] images of positions 192..195 (rows t+1 / t+2), then wave 6's scratch; wave 7's: scratch
    float* la = reinterpret_cast<float*>(stages + 6 * (32 * C * 4));
    float* scratch = reinterpret_cast<float*>(stages + (wave & 7) * (32 * C * 4) + 2048);
    unsigned* lmask = reinterpret_cast<unsigned*>(stages + 7 * (32 * C * 4) + 2048 + 4096);   // [3][C] ReluGrad bits of 192..195, by row % 3
    // ReluGrad bits of tiles 4, 5, by row % 3: their backward (waves 6, 7, row t-1) runs while their forward (waves 4, 5,
    // row t+1 -- the same parity) writes the next bits; tiles 0-3 are read and rewritten by one wave in sequence (parity is enough)
    unsigned* xmask = reinterpret_cast<unsigned*>(stages + 6 * (32 * C * 4) + 2048 + 4096);   // [3][2][C]
    const int gt = tid - 512;                                    // gather thread id (G waves)
    const int gw = wave - 8;
    const int n_my = ((int)a.n_rows - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto row_of = [&](int k) { return (long long)blockIdx.x + (long long)k * gridDim.x; };

    for (int i = tid; i < 2 * GSZ; i += 768) sg[i] = 0.f;       // zero borders of both da5 images, written once
    for (int i = tid; i < (C / 8) * 256; i += 768) {
        const int e = i & 3, l = (i >> 2) & 63, kk = i >> 8;
        const int kappa = l & 31, c = kk * 8 + (l >> 5) * 4 + e;
        sWf[i] = kappa < 25 ? a.F5[kappa * C + c] : 0.f;
    }
    for (int i = tid; i < 13 * 64 * (C / 32); i += 768) {
        const int u = i % (C / 32), l = (i / (C / 32)) & 63, st = i / (64 * (C / 32));
        const int kappa = 2 * st + (l >> 5);
        sWb[i] = kappa < 25 ? a.F5[kappa * C + u * 32 + (l & 31)] : 0.f;
    }
    const int q = tile * 32 + frow;                              // this lane's position in the forward role (tiles 0-5: always valid)
    constexpr int CH = C / 4;                                    // 16-B chunks per position
    constexpr int NI = 32 * CH / 64;                             // DMA instructions per tile
    auto stage_row = [&](int k) {
        const char* src = reinterpret_cast<const char*>(a.h3 + row_of(k) * (196 * C) + (long long)tile * 32 * C);
#pragma unroll
        for (int qi = 0; qi < NI; ++qi) {
            const int slot = qi * 64 + lane;
            const int pos = slot / CH, c = slot % CH;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4)),
                (__attribute__((address_space(3))) void*)(stage + qi * 1024), 16, 0, 0);
        }
    };
    auto read_frags = [&](f32x4 (&av)[C / 8]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < C / 8; ++kk)
            av[kk] = *reinterpret_cast<const f32x4*>(stage + frow * (C * 4) + (((kk * 2 + fh) ^ (frow & (CH - 1))) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto fwd = [&](int k, const f32x4 (&av)[C / 8]) {
        unsigned* mk = tile < 4 ? smask + (k & 1) * MSZ + tile * C : xmask + ((k % 3) * 2 + (tile - 4)) * C;
        int word = 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            unsigned lo[4], hi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long bal = __ballot(av[kk][e] > 0.f);
                lo[e] = __builtin_amdgcn_readfirstlane((unsigned)bal);
                hi[e] = __builtin_amdgcn_readfirstlane((unsigned)(bal >> 32));
            }
            // (a v_cmp result SGPR is not safe as the data operand of a v_writelane issued right behind it on gfx950: 4 wait states)
            asm volatile("s_nop 3\n\t"
                         "v_writelane_b32 %0, %1, %9\n\tv_writelane_b32 %0, %2, %10\n\t"
                         "v_writelane_b32 %0, %3, %11\n\tv_writelane_b32 %0, %4, %12\n\t"
                         "v_writelane_b32 %0, %5, %13\n\tv_writelane_b32 %0, %6, %14\n\t"
                         "v_writelane_b32 %0, %7, %15\n\tv_writelane_b32 %0, %8, %16"
                         : "+v"(word)
                         : "s"(lo[0]), "s"(lo[1]), "s"(lo[2]), "s"(lo[3]), "s"(hi[0]), "s"(hi[1]), "s"(hi[2]), "s"(hi[3]),
                           "i"(8 * kk), "i"(8 * kk + 1), "i"(8 * kk + 2), "i"(8 * kk + 3),
                           "i"(8 * kk + 4), "i"(8 * kk + 5), "i"(8 * kk + 6), "i"(8 * kk + 7));
        }
        mk[lane] = (unsigned)word;
        tail_fwd_compute_ldsw<C>(av, tile * 32, sWf, sP + (k & 1) * PSZ, MN_NKP, lane);
    };
    auto bwd = [&](int k) {
        const unsigned* mk = btile < 4 ? smask + (k & 1) * MSZ + btile * C : xmask + ((k % 3) * 2 + (btile - 4)) * C;
        float* hrow = a.h3 + row_of(k) * (196 * C);
        const int qq = btile * 32 + frow;                        // tiles 0-5: every position exists
        const int oh = qq / 14, ow = qq - oh * 14;
        f32x16 acc[C / 32];
        tail_bwd_tile_ldsw<C, 1, MN_GWP>(sg + (k & 1) * GSZ, (2 * oh) * MN_GWP + 2 * ow, true, sWb, acc, lane);
        // transposition scratch: waves 0-3 their own tile of the P buffer of the same parity (free until their forward GEMM
        // later in this step rewrites it), waves 6 / 7 a private 4 KB (the P tiles 4 / 5 belong to waves 4 / 5, who are
        // writing them right now)
        float* tb = wave < 4 ? sP + (k & 1) * PSZ + btile * 32 * MN_NKP : scratch;
        const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
        for (int u = 0; u < C / 32; ++u) {
            const unsigned mw = mk[u * 32 + frow];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = (e & 3) + 8 * (e >> 2) + 4 * fh;
                tb[pr * 32 + frow] = ((mw >> pr) & 1u) ? acc[u][e] : 0.f;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int qr = btile * 32 + p * 8 + er;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
                *reinterpret_cast<f32x4*>(hrow + qr * C + u * 32 + ec) = v;
            }
        }
    };
    const float bias = a.b5[0];
    const float gscale = 2.0f / 784.0f;
    auto load_x = [&](int k, float (&xv)[4]) {
        const float* xrow = a.x + (long long)((unsigned)row_of(k) / (unsigned)a.R) * 784;     // rows < 2^24: 32-bit division
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = gt + 256 * r;
            xv[r] = p < 784 ? xrow[p] : 0.f;
        }
    };
    auto gather = [&](int k, const float (&xv)[4]) {
        const float* pP = sP + (k & 1) * PSZ;
        float* pg = sg + (k & 1) * GSZ;
        const long long n = row_of(k);
        float sq = 0.f;
        float tv[4][9];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = gt + 256 * r;
            const int i = p / 28, j = p - i * 28;
            const int kh0 = (i + 1) & 1, kw0 = (j + 1) & 1;
#pragma unroll
            for (int ah = 0; ah < 3; ++ah) {
                const int kh = kh0 + 2 * ah;
                const int oh = (i + 1 - kh) >> 1;
                const bool okh = p < 784 && !(kh > 4 || oh < 0 || oh >= 14);
#pragma unroll
                for (int aw = 0; aw < 3; ++aw) {
                    const int kw = kw0 + 2 * aw;
                    const int ow = (j + 1 - kw) >> 1;
                    const bool ok = okh && !(kw > 4 || ow < 0 || ow >= 14);
                    tv[r][ah * 3 + aw] = pP[ok ? (oh * 14 + ow) * MN_NKP + kh * 5 + kw : 31];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(tv[r][t]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = gt + 256 * r;
            if (p >= 784) break;
            const int i = p / 28, j = p - i * 28;
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) sacc += tv[r][t];
            const float y = 1.0f / (1.0f + expf(-(sacc + bias)));
            const float d = y - xv[r];
            sq = __builtin_fmaf(d, d, sq);
            pg[(i + 1) * MN_GWP + (j + 1)] = gscale * d * y * (1.0f - y);
            if (a.y) a.y[n * 784 + p] = y;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        if (lane == 0) sred[(k & 1) * 4 + gw] = sq;
    };
    auto finish_loss = [&](int k) {       // after the barrier that follows gather(k)
        const float* r4 = sred + (k & 1) * 4;
        a.loss[row_of(k)] = ((r4[0] + r4[1]) + (r4[2] + r4[3])) * (1.0f / 784.0f);
    };
    // ---- positions 192..195 on the gather waves (plain fma chains in the MFMA's k order) ---------------------------------
    auto stage_left = [&](int k) {        // one wave: 4 positions x C floats = 1 KB, contiguous in h3
        if (gw != 0) return;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a.h3 + row_of(k) * (196 * C) + 192 * C + lane * 4),
            (__attribute__((address_space(3))) void*)(la + (k & 1) * (4 * C)), 16, 0, 0);
    };
    auto left_fwd = [&](int k) {
        const float* A = la + (k & 1) * (4 * C);
        if (gt < C) {                     // ReluGrad bits of the 4 positions, one word per channel (bit p = position 192 + p)
            unsigned w = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) w |= (A[p * C + gt] > 0.f ? 1u : 0u) << p;
            lmask[(k % 3) * C + gt] = w;
        }
        if (gt < 100) {                   // P[192 + qp][kappa] = sum_c h[c] F[kappa][c], c in the order of the MFMA k-steps
            const int qp = gt / 25, kappa = gt - qp * 25;
            float acc = 0.f;
#pragma unroll
            for (int kk = 0; kk < C / 8; ++kk) {
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(A + qp * C + kk * 8);
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(A + qp * C + kk * 8 + 4);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + kappa) * 4);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + 32 + kappa) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_fmaf(h0[e], w0[e], acc);
                    acc = __builtin_fmaf(h1[e], w1[e], acc);
                }
            }
            sP[(k & 1) * PSZ + (192 + qp) * MN_NKP + kappa] = acc;
        }
    };
    auto left_bwd = [&](int k) {          // dH[192 + qp][c] = sum_kappa G[kappa] F[kappa][c], kappa ascending (the MFMA k order)
        const int qp = gt >> 6, c = gt & 63;
        const float* pg = sg + (k & 1) * GSZ + (2 * 13) * MN_GWP + 2 * (10 + qp);      // position 192 + qp = (oh 13, ow 10 + qp)
        float acc = 0.f;
#pragma unroll
        for (int kappa = 0; kappa < 25; ++kappa) {
            const float gv = pg[(kappa / 5) * MN_GWP + (kappa % 5)];
            const float wv = sWb[((kappa >> 1) * 64 + (kappa & 1) * 32 + (c & 31)) * (C / 32) + (c >> 5)];
            acc = __builtin_fmaf(gv, wv, acc);
        }
        const unsigned mw = lmask[(k % 3) * C + c];
        a.h3[row_of(k) * (196 * C) + (192 + qp) * C + c] = ((mw >> qp) & 1u) ? acc : 0.f;
    };

    // ---- the roles run separate loops with the same barrier sequence: sg-zero | prologue | one per step ------------
    if (mrole) {
        f32x4 A[C / 8];
        if (has_fwd) stage_row(0);
        __syncthreads();                                         // sg zeroed, filter fragments in LDS
        if (has_fwd) {
            read_frags(A);
            if (n_my > 1) stage_row(1);
            fwd(0, A);
        }
        __syncthreads();
        for (int t = 0; t <= n_my; ++t) {
            if (has_fwd && t + 1 < n_my) {
                read_frags(A);                                   // row t+1 (staged one step ago)
                if (t + 2 < n_my) stage_row(t + 2);              // lands during this step
            }
            if (has_bwd && t >= 1) bwd(t - 1);
            if (has_fwd && t + 1 < n_my) fwd(t + 1, A);
            __syncthreads();
        }
    } else {
        float xv0[4], xv1[4];
        load_x(0, xv0);
        stage_left(0);
        if (n_my > 1) stage_left(1);
        __syncthreads();
        left_fwd(0);
        __syncthreads();
        auto step = [&](int t, float (&xc)[4], float (&xn)[4]) {
            if (t >= 1 && gt == 0) finish_loss(t - 1);
            if (t < n_my) {
                if (t + 1 < n_my) load_x(t + 1, xn);
                gather(t, xc);
            }
            if (t >= 1) left_bwd(t - 1);
            if (t + 1 < n_my) left_fwd(t + 1);                   // its image was staged two steps ago
            if (t + 2 < n_my) stage_left(t + 2);                 // into the buffer left_fwd(t) read in the previous step
            __syncthreads();
        };
        for (int t = 0; t <= n_my; t += 2) {
            step(t, xv0, xv1);
            if (t + 1 <= n_my) step(t + 1, xv1, xv0);
        }
    }
}
#endif  // DG_MEASURE

// ---- pipelined variant, third generation: ONE GEMM per wave ------------------------------------------------------------
// mnist_tail_pipe2_kernel levelled the MFMAs over the SIMDs and the step stayed as long (65.5 vs 66.6 us): what bounds a step is
// not the matrix pipe but the dependent chain inside the waves that run BOTH GEMMs of a tile one after the other (wait for the
// DMA, read fragments, issue the next DMA, gather + multiply + mask + transpose + store the backward tile, ballots + multiply +
// write P: ~11 k cycles).  Here the workgroup has 16 waves and every GEMM of a row has a wave of its own:
//   waves 0-5    forward GEMM of tile w (stage, fragments, ReluGrad bits, P)
//   waves 6-11   backward GEMM of tile w-6 (masked tile through a private 4 KB scratch, row stores)
//   waves 12-15  gather / sigmoid / loss and the 4 positions 192..195 (as in pipe2)
// so the longest chain in a step is one GEMM (or the gather).  ReluGrad bits are kept for three rows (written for row t+1 while
// read for row t-1), everything else as in pipe2; 128 VGPRs per wave (4 waves per SIMD).
// The second-generation text follows.
// ---- (second generation) the matrix work levelled over the four SIMDs
// In mnist_tail_pipe_kernel wave w < 7 owns position tile w for both GEMMs: waves (0,4), (1,5), (2,6) share a SIMD, so three
// SIMDs carry two tiles = 116 MFMAs per step (7.4 k cycles of matrix pipe) and the fourth one tile -- and tile 6 is 4 real
// positions (192..195) padded to 32.  The M waves are the step's critical path (~11 k of 13 k cycles, tools/tail_trace_mnist.py),
// mostly waiting for each other's MFMAs.  Here
//   waves 0-3   forward + backward of tile w                     (58 MFMAs)
//   waves 4, 5  forward of tile w only                           (32)
//   waves 6, 7  backward of tiles 4, 5 (masks, da5 image and filter fragments are in LDS: any wave can do it; the masked
//               tile goes through a 4 KB scratch in the wave's own, otherwise unused staging region)          (26)
//   the 4 positions of "tile 6" leave the matrix pipe: the gather waves compute their 4 x 25 P entries, their ReluGrad bits and
//   their 4 x 64 gradients with v_fma chains in the MFMA's k order (bit-identical: an MFMA is a k-ordered fma chain), from a
//   1 KB image one of them stages by LDS-DMA two rows ahead
// so every SIMD carries 84-90 MFMAs per step (5.8 k cycles) and no MFMA is spent on padding rows.  Same barrier sequence, same
// buffers and the same arithmetic per element as mnist_tail_pipe_kernel (tests/test_gpu_variants.py: bit-identical).
// BN (round 6, MnistTailArgs::bn_pre): the input map is the PRE-ACTIVATIONS of Generator.3's Batchnorm layer; the forward waves apply
// relu(bn(.)) to their fragments as they leave LDS (the float expression of bn_apply_fwd_kernel: the activation image is never written
// or read -- one pass of 2 x 128 MB less per step at 2560 rows), and the backward waves add up that layer's backward sums (dy and
// dy * xhat per channel, xhat re-formed from the pre-activations they fetch beside their MFMAs) over all their rows; every wave
// leaves one [2][C] record at the end (6 tiles + positions 192..195), added up in LDS to ONE per workgroup (MnistTailArgs::bn_sums), which
// launch_bn_backward_from_blocks finalizes -- the statistics pass of the Batchnorm backward (another 2 x 128 MB) is not run either.
template <int C, bool BN>
__global__ __launch_bounds__(1024) void mnist_tail_pipe3_kernel(MnistTailArgs a) {
    static_assert(C == 64, "64 channels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PSZ = 196 * MN_NKP, GSZ = MN_GR * MN_GWP;
    // BN: [4][C] mean, rstd, scale, offset at the START of LDS -- ds_read takes a 16-bit immediate offset, and addresses beyond it
    // cost one register each (32 of them in the forward waves' conversion: measured as 200 spilled registers)
    float* sBN = reinterpret_cast<float*>(smem);
    float* sP = reinterpret_cast<float*>(smem) + (BN ? 4 * C : 0);     // [2][196][MN_NKP]
    float* sg = sP + 2 * PSZ;                                    // [2][31][32]
    unsigned* xmask = reinterpret_cast<unsigned*>(sg + 2 * GSZ); // [3][6 tiles][C] ReluGrad bits by row % 3
    float* sred = reinterpret_cast<float*>(xmask + 3 * 6 * C);   // [2][4]
    float* sWf = sred + 8;                                       // forward filter fragments [C/8][64][4]
    float* sWb = sWf + (C / 8) * 256;                            // backward filter fragments [13][64][C/32]
    char* stages = reinterpret_cast<char*>(sWb + 13 * 64 * (C / 32));     // [6 forward waves][8 KB], then [6 backward waves][4 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const bool mrole = wave < 12;
    const bool has_fwd = wave < 6;                               // forward GEMM of tile `wave`
    const bool has_bwd = wave >= 6 && wave < 12;                 // backward GEMM of tile wave - 6
    const int tile = wave;                                       // forward tile
    const int btile = wave - 6;                                  // backward tile
    char* stage = stages + (wave < 6 ? wave : 0) * (32 * C * 4); // waves 0-5: A tile [32][C], LDS-DMA target
    float* scratch = reinterpret_cast<float*>(stages + 6 * (32 * C * 4) + (has_bwd ? btile : 0) * 4096);
    float* la = reinterpret_cast<float*>(stages + 6 * (32 * C * 4) + 6 * 4096);   // [2][4 positions][C] images of positions 192..195
    unsigned* lmask = reinterpret_cast<unsigned*>(la + 2 * 4 * C);                // [3][C] ReluGrad bits of 192..195, by row % 3
    const float* in_map = BN ? a.bn_pre : a.h3;                                   // what the forward waves stage
    float* lh = reinterpret_cast<float*>(lmask + 3 * C);                          // BN: [2 waves][2 positions][C] relu(bn(.)) of 192..195
    unsigned* lmaskb = reinterpret_cast<unsigned*>(lh + 2 * 2 * C);               // BN: [3][2 waves][C] ReluGrad bits, two positions per word
    const int gt = tid - 768;                                    // gather thread id (G waves)
    const int gw = wave - 12;
    const int n_my = ((int)a.n_rows - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto row_of = [&](int k) { return (long long)blockIdx.x + (long long)k * gridDim.x; };

    for (int i = tid; i < 2 * GSZ; i += 1024) sg[i] = 0.f;       // zero borders of both da5 images, written once
    if constexpr (BN) {
        if (tid < 2 * C) sBN[tid] = a.bn_fstats[tid];
        else if (tid < 3 * C) sBN[tid] = a.bn_scale[tid - 2 * C];
        else if (tid < 4 * C) sBN[tid] = a.bn_offset[tid - 3 * C];
    }
    // relu(bn(v)) for 4 consecutive channels from c0 (dg_bn.hip bn_apply_fwd_kernel's expression, bit for bit)
    auto bn_relu4 = [&](f32x4 v, int c0) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(sBN + c0), rs = *reinterpret_cast<const f32x4*>(sBN + C + c0);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(sBN + 2 * C + c0), be = *reinterpret_cast<const f32x4*>(sBN + 3 * C + c0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = __builtin_fmaf((v[e] - mu[e]) * rs[e], gm[e], be[e]);
            o[e] = t > 0.f ? t : 0.f;
        }
        return o;
    };
    for (int i = tid; i < (C / 8) * 256; i += 1024) {
        const int e = i & 3, l = (i >> 2) & 63, kk = i >> 8;
        const int kappa = l & 31, c = kk * 8 + (l >> 5) * 4 + e;
        sWf[i] = kappa < 25 ? a.F5[kappa * C + c] : 0.f;
    }
    for (int i = tid; i < 13 * 64 * (C / 32); i += 1024) {
        const int u = i % (C / 32), l = (i / (C / 32)) & 63, st = i / (64 * (C / 32));
        const int kappa = 2 * st + (l >> 5);
        sWb[i] = kappa < 25 ? a.F5[kappa * C + u * 32 + (l & 31)] : 0.f;
    }
    const int q = tile * 32 + frow;                              // this lane's position in the forward role (tiles 0-5: always valid)
    constexpr int CH = C / 4;                                    // 16-B chunks per position
    constexpr int NI = 32 * CH / 64;                             // DMA instructions per tile
    auto stage_row = [&](int k) {
        const char* src = reinterpret_cast<const char*>(in_map + row_of(k) * (196 * C) + (long long)tile * 32 * C);
#pragma unroll
        for (int qi = 0; qi < NI; ++qi) {
            const int slot = qi * 64 + lane;
            const int pos = slot / CH, c = slot % CH;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4)),
                (__attribute__((address_space(3))) void*)(stage + qi * 1024), 16, 0, 0);
        }
    };
    auto read_frags = [&](f32x4 (&av)[C / 8]) {          // (the stage landed before the barrier that ended the previous step)
#pragma unroll
        for (int kk = 0; kk < C / 8; ++kk)
            av[kk] = *reinterpret_cast<const f32x4*>(stage + frow * (C * 4) + (((kk * 2 + fh) ^ (frow & (CH - 1))) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (BN) {
            // two k-steps at a time: left alone hipcc fetches the constants of all eight first (128 registers; the kernel has 128)
#pragma unroll
            for (int kk = 0; kk < C / 8; kk += 2) {
                av[kk] = bn_relu4(av[kk], kk * 8 + fh * 4);
                av[kk + 1] = bn_relu4(av[kk + 1], kk * 8 + 8 + fh * 4);
                asm volatile("" : "+v"(av[kk]), "+v"(av[kk + 1]) :: "memory");      // (the results exist here, and no LDS read moves across)
            }
        }
    };
    auto fwd = [&](int k, const f32x4 (&av)[C / 8]) {
        unsigned* mk = xmask + ((k % 3) * 6 + tile) * C;
        int word = 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            unsigned lo[4], hi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long bal = __ballot(av[kk][e] > 0.f);
                lo[e] = __builtin_amdgcn_readfirstlane((unsigned)bal);
                hi[e] = __builtin_amdgcn_readfirstlane((unsigned)(bal >> 32));
            }
            // (a v_cmp result SGPR is not safe as the data operand of a v_writelane issued right behind it on gfx950: 4 wait states)
            asm volatile("s_nop 3\n\t"
                         "v_writelane_b32 %0, %1, %9\n\tv_writelane_b32 %0, %2, %10\n\t"
                         "v_writelane_b32 %0, %3, %11\n\tv_writelane_b32 %0, %4, %12\n\t"
                         "v_writelane_b32 %0, %5, %13\n\tv_writelane_b32 %0, %6, %14\n\t"
                         "v_writelane_b32 %0, %7, %15\n\tv_writelane_b32 %0, %8, %16"
                         : "+v"(word)
                         : "s"(lo[0]), "s"(lo[1]), "s"(lo[2]), "s"(lo[3]), "s"(hi[0]), "s"(hi[1]), "s"(hi[2]), "s"(hi[3]),
                           "i"(8 * kk), "i"(8 * kk + 1), "i"(8 * kk + 2), "i"(8 * kk + 3),
                           "i"(8 * kk + 4), "i"(8 * kk + 5), "i"(8 * kk + 6), "i"(8 * kk + 7));
        }
        mk[lane] = (unsigned)word;
        tail_fwd_compute_ldsw<C>(av, tile * 32, sWf, sP + (k & 1) * PSZ, MN_NKP, lane);
    };
    // BN: this lane's backward sums over all its rows -- after the transposition below a lane owns 4 consecutive channels (ec) of the
    // tile's rows er, er + 8, er + 16, er + 24, for both 32-channel groups
    f32x4 bs1[C / 32], bs2[C / 32];
#pragma unroll
    for (int u = 0; u < C / 32; ++u) { bs1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; bs2[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto bwd = [&](int k) {
        const unsigned* mk = xmask + ((k % 3) * 6 + btile) * C;
        float* hrow = a.h3 + row_of(k) * (196 * C);
        const int qq = btile * 32 + frow;                        // tiles 0-5: every position exists
        const int oh = qq / 14, ow = qq - oh * 14;
        // BN: the pre-activations of this lane's output elements (L2: the tile's forward wave staged them two steps ago): those of
        // the first channel group are fetched before the MFMAs, those of the second behind them, when the operand registers are free
        f32x4 pv[C / 32][4];
        const float* prow = a.bn_pre + row_of(k) * (196 * C) + (long long)(btile * 32 + (lane >> 3)) * C + (lane & 7) * 4;
        int lane_k = lane;
        if constexpr (BN) {
            // (the lane id as this step sees it: 13 per-lane LDS offsets of the gather are then formed here, two VALU each, instead of
            // living in registers across the loop -- with the sums and the pre-activations the loop did not fit its 128 registers)
            asm volatile("" : "+v"(lane_k));
#pragma unroll
            for (int p = 0; p < 4; ++p) pv[0][p] = *reinterpret_cast<const f32x4*>(prow + p * 8 * C);
        }
        f32x16 acc[C / 32];
        int gbase = (2 * oh) * MN_GWP + 2 * ow;
        if constexpr (BN) {               // (formed from this step's lane id as well: kept across the loop it was the one spilled register,
            const int qk = btile * 32 + (lane_k & 31), ohk = qk / 14;     // whose reload waits for vmcnt(0) = for the loads just issued)
            gbase = (2 * ohk) * MN_GWP + 2 * (qk - ohk * 14);
        }
        tail_bwd_tile_ldsw<C, 1, MN_GWP>(sg + (k & 1) * GSZ, gbase, true, sWb, acc, lane_k);
        float* tb = scratch;                                     // private 4 KB: the P tile of this parity is being rewritten by the forward wave
        const int er = lane >> 3, ec = (lane & 7) * 4;
        if constexpr (BN) {
#pragma unroll
            for (int u = 1; u < C / 32; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) pv[u][p] = *reinterpret_cast<const f32x4*>(prow + p * 8 * C + u * 32);
        }
#pragma unroll
        for (int u = 0; u < C / 32; ++u) {
            const unsigned mw = mk[u * 32 + frow];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = (e & 3) + 8 * (e >> 2) + 4 * fh;
                tb[pr * 32 + frow] = ((mw >> pr) & 1u) ? acc[u][e] : 0.f;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int qr = btile * 32 + p * 8 + er;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
                *reinterpret_cast<f32x4*>(hrow + qr * C + u * 32 + ec) = v;
                if constexpr (BN) {
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(sBN + u * 32 + ec), rs = *reinterpret_cast<const f32x4*>(sBN + C + u * 32 + ec);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bs1[u][e] += v[e];
                        bs2[u][e] = __builtin_fmaf(v[e], (pv[u][p][e] - mu[e]) * rs[e], bs2[u][e]);
                    }
                }
            }
        }
    };
    // one [2][C] record of a backward wave: the 8 lanes that share 4 channels hold different rows -- a fixed xor tree over lane bits 3..5.
    // The workgroup's 10 records (6 tiles, positions 192..195) meet in LDS (the P images are dead by then) and leave as ONE
    float* srec = sP;                     // [10][2][C]
    auto flush_bwd_sums = [&]() {
        if constexpr (BN) {
            float* rec = srec + btile * (2 * C);
#pragma unroll
            for (int u = 0; u < C / 32; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 8; m < 64; m <<= 1) { bs1[u][e] += __shfl_xor(bs1[u][e], m, 64); bs2[u][e] += __shfl_xor(bs2[u][e], m, 64); }
                if (lane < 8) {
                    *reinterpret_cast<f32x4*>(rec + u * 32 + lane * 4) = bs1[u];
                    *reinterpret_cast<f32x4*>(rec + C + u * 32 + lane * 4) = bs2[u];
                }
            }
        }
    };
    const float bias = a.b5[0];
    const float gscale = 2.0f / 784.0f;
    // ---- gather: the 784 output pixels by PARITY CLASS.  Pixel (i, j) = (2u + a, 2v + b) has 2 (a = 0: kh 1, 3) or 3 (a = 1:
    // kh 0, 2, 4) filter rows and likewise columns: 4 / 6 / 6 / 9 terms.  Every gather wave takes 49 pixels (u, v) of each of
    // the four classes, one class per round, so a round reads exactly the terms that exist (25 LDS reads per thread and step
    // instead of 4 x 9 with a zero pad entry for the missing ones -- the trace had the gather waves as the step's critical
    // path); the terms are added in ascending (kh, kw) as everywhere else, so every pre-activation, hence y and da5, is bit-identical
    // to the other tail kernels'.  This kernel does NOT reduce the per-row loss: the projection loop never reads it (the launch
    // that needs it -- the last forward, dg_loss_grad -- runs a kernel that does, MnistTailArgs.want_loss).
    const int gq = gw * 49 + (lane < 49 ? lane : 48);             // (u, v) index of this lane, 0..195
    const int gu = gq / 14, gv = gq - gu * 14;
    auto load_x = [&](int k, float (&xv)[4]) {
        const float* xrow = a.x + (long long)((unsigned)row_of(k) / (unsigned)a.R) * 784;     // rows < 2^24: 32-bit division
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) xv[cls] = xrow[(2 * gu + (cls >> 1)) * 28 + 2 * gv + (cls & 1)];
    };
    auto gather = [&](int k, const float (&xv)[4]) {
        const float* pP = sP + (k & 1) * PSZ;
        float* pg = sg + (k & 1) * GSZ;
        // a term whose input position does not exist (image border) reads the zero pad entry P[0][31] as before
        float tv[25];
        int nt = 0;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int pa = cls >> 1, pb = cls & 1;
#pragma unroll
            for (int ah = 0; ah < 2 + pa; ++ah) {
                const int kh = (1 - pa) + 2 * ah;
                const int oh = gu + ((pa + 1 - kh) >> 1);          // (i + 1 - kh) / 2 with i = 2u + a
                const bool okh = oh >= 0 && oh < 14;
#pragma unroll
                for (int aw = 0; aw < 2 + pb; ++aw) {
                    const int kw = (1 - pb) + 2 * aw;
                    const int ow = gv + ((pb + 1 - kw) >> 1);
                    const bool ok = okh && ow >= 0 && ow < 14;
                    tv[nt++] = pP[ok ? (oh * 14 + ow) * MN_NKP + kh * 5 + kw : 31];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 25; ++t) asm volatile("" : "+v"(tv[t]));
        nt = 0;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int pa = cls >> 1, pb = cls & 1;
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < (2 + pa) * (2 + pb); ++t) sacc += tv[nt++];
            const float y = 1.0f / (1.0f + expf(-(sacc + bias)));
            const float d = y - xv[cls];
            if (lane < 49) pg[(2 * gu + pa + 1) * MN_GWP + (2 * gv + pb + 1)] = gscale * d * y * (1.0f - y);
        }
    };
    // ---- positions 192..195 on the gather waves (plain fma chains in the MFMA's k order) ---------------------------------
    auto stage_left = [&](int k) {        // forward wave 5: 4 positions x C floats = 1 KB, contiguous in h3
        if (wave != 5) return;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(in_map + row_of(k) * (196 * C) + 192 * C + lane * 4),
            (__attribute__((address_space(3))) void*)(la + (k & 1) * (4 * C)), 16, 0, 0);
    };
    // (run by the FORWARD waves after their GEMM: the trace showed the gather waves as the step's critical path -- 10.9 k cycles with
    // this work against 5.3 k for a forward wave; lt = thread index inside the group of waves that shares the piece)
    auto left_fwd = [&](int k, int lt) {
        const float* A = la + (k & 1) * (4 * C);
        if constexpr (BN) {
            // wave w2 (0 / 1) takes positions 192 + 2 w2 and + 1: lane = channel converts the two pre-activations once (25 kappa threads
            // per position would each convert all 64), parks them in LDS and notes the gate bits; the wave then reads its own image
            // (LDS operations of one wave complete in order) for the 2 x 25 dot products
            const int w2 = lt >> 6, c = lt & 63;
            float* img = lh + w2 * (2 * C);
            const float mu = sBN[c], rs = sBN[C + c], gm = sBN[2 * C + c], be = sBN[3 * C + c];
            unsigned bits = 0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float t = __builtin_fmaf((A[(2 * w2 + p) * C + c] - mu) * rs, gm, be);
                img[p * C + c] = t > 0.f ? t : 0.f;
                bits |= (t > 0.f ? 1u : 0u) << p;
            }
            lmaskb[((k % 3) * 2 + w2) * C + c] = bits;
            if (c < 50) {
                const int ql = c / 25, kappa = c - ql * 25;
                float acc = 0.f;
#pragma unroll
                for (int kk = 0; kk < C / 8; ++kk) {
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(img + ql * C + kk * 8);
                    const f32x4 h1 = *reinterpret_cast<const f32x4*>(img + ql * C + kk * 8 + 4);
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + kappa) * 4);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + 32 + kappa) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc = __builtin_fmaf(h0[e], w0[e], acc);
                        acc = __builtin_fmaf(h1[e], w1[e], acc);
                    }
                }
                sP[(k & 1) * PSZ + (192 + 2 * w2 + ql) * MN_NKP + kappa] = acc;
            }
            return;
        }
        if (lt < C) {                     // ReluGrad bits of the 4 positions, one word per channel (bit p = position 192 + p)
            unsigned w = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                w |= (A[p * C + lt] > 0.f ? 1u : 0u) << p;
            }
            lmask[(k % 3) * C + lt] = w;
        }
        if (lt < 100) {                   // P[192 + qp][kappa] = sum_c h[c] F[kappa][c], c in the order of the MFMA k-steps
            const int qp = lt / 25, kappa = lt - qp * 25;
            // (measured: fetching the operands in one or two batches, each waited for once, made this piece SLOWER -- 9.7 k vs 8.7 k
            // cycles for the wave's step, the bursts queue behind the GEMM waves' LDS traffic; one k-step at a time it is)
            float acc = 0.f;
#pragma unroll
            for (int kk = 0; kk < C / 8; ++kk) {
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(A + qp * C + kk * 8);
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(A + qp * C + kk * 8 + 4);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + kappa) * 4);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWf + (kk * 64 + 32 + kappa) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_fmaf(h0[e], w0[e], acc);
                    acc = __builtin_fmaf(h1[e], w1[e], acc);
                }
            }
            sP[(k & 1) * PSZ + (192 + qp) * MN_NKP + kappa] = acc;
        }
    };
    float ls1 = 0.f, ls2 = 0.f;           // BN: backward sums of this thread's (position 192 + qp, channel c) over its rows
    // BN: the pre-activation of this thread's element of row k.  Fetched at the TOP of the step, before the step's LDS-DMAs: vmcnt
    // counts in issue order, so a load issued behind them could only be waited for together with them (the whole HBM latency of
    // the row two steps ahead: measured as +3 k cycles per step on waves 2-5)
    auto left_pre = [&](int k, int lt) { return a.bn_pre[row_of(k) * (196 * C) + (192 + (lt >> 6)) * C + (lt & 63)]; };
    auto left_bwd = [&](int k, int lt, float pre) {  // dH[192 + qp][c] = sum_kappa G[kappa] F[kappa][c], kappa ascending (the MFMA k order)
        const int qp = lt >> 6, c = lt & 63;
        const float* pg = sg + (k & 1) * GSZ + (2 * 13) * MN_GWP + 2 * (10 + qp);      // position 192 + qp = (oh 13, ow 10 + qp)
        float acc = 0.f;
#pragma unroll
        for (int kappa = 0; kappa < 25; ++kappa) {
            const float gv = pg[(kappa / 5) * MN_GWP + (kappa % 5)];
            const float wv = sWb[((kappa >> 1) * 64 + (kappa & 1) * 32 + (c & 31)) * (C / 32) + (c >> 5)];
            acc = __builtin_fmaf(gv, wv, acc);
        }
        const unsigned mw = BN ? lmaskb[((k % 3) * 2 + (qp >> 1)) * C + c] >> (qp & 1) : lmask[(k % 3) * C + c] >> qp;
        const float dv = (mw & 1u) ? acc : 0.f;
        a.h3[row_of(k) * (196 * C) + (192 + qp) * C + c] = dv;
        if constexpr (BN) { ls1 += dv; ls2 = __builtin_fmaf(dv, (pre - sBN[c]) * sBN[C + c], ls2); }
    };

    // ---- the roles run separate loops with the same barrier sequence: sg-zero | prologue | one per step ------------
    (void)mrole;
    // Barriers are LDS-only (lds_barrier: lgkmcnt(0) + s_barrier).  __syncthreads() would also wait for vmcnt(0), i.e. for the
    // acknowledgement of every row store a backward wave has just issued -- nobody in this kernel reads them, and their latency
    // then sits on every step's critical path.  The only global -> LDS traffic a barrier has to publish are the LDS-DMAs of the
    // forward waves, issued at the START of a step and waited for (long landed) right before the barrier that ends it.
    auto dma_landed_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lds_barrier(); };
#ifdef DG_MEASURE
    // phase stamps (tools/tail_trace_mnist.py): per role, cycles between the barriers (work) and inside them (wait), steps 2 .. n-3
    const bool tr = DG_TRACE_PTR(a) != nullptr && lane == 0 && (wave == 0 || wave == 3 || wave == 6 || wave == 12) && blockIdx.x < 2048;
    long long tw = 0, tb = 0, tc0 = 0, tc1 = 0;
    int tn = 0;
#define TR_BEGIN() do { if (tr) tc0 = (long long)__builtin_readcyclecounter(); } while (0)
#define TR_MID() do { if (tr) tc1 = (long long)__builtin_readcyclecounter(); } while (0)
#define TR_END(t) do { if (tr && (t) >= 2 && (t) + 2 < n_my) { tw += tc1 - tc0; tb += (long long)__builtin_readcyclecounter() - tc1; ++tn; } } while (0)
#define TR_FLUSH(slot) do { if (tr) { long long* o = DG_TRACE_PTR(a) + (long long)blockIdx.x * 16 + (slot) * 3; o[0] = tw; o[1] = tb; o[2] = tn; } } while (0)
#else
#define TR_BEGIN() do {} while (0)
#define TR_MID() do {} while (0)
#define TR_END(t) do {} while (0)
#define TR_FLUSH(slot) do {} while (0)
#endif
    if (has_fwd) {
        // (a loop per role: with both GEMMs in one loop body the forward wave's fragments stay live across the backward code it
        // never runs, and the kernel does not fit the 128 VGPRs of four waves per SIMD)
        f32x4 A[C / 8];
        stage_row(0);
        stage_left(0);
        dma_landed_barrier();                                    // sg zeroed, filter fragments in LDS, row 0 staged
        read_frags(A);
        if (n_my > 1) { stage_row(1); stage_left(1); }
        fwd(0, A);
        if (wave < 2) left_fwd(0, tid);
        dma_landed_barrier();
        for (int t = 0; t <= n_my; ++t) {
            TR_BEGIN();
            float lpre = 0.f;
            if constexpr (BN) { if (wave >= 2 && t >= 1) lpre = left_pre(t - 1, tid - 128); }
            if (t + 1 < n_my) {
                read_frags(A);                                   // row t+1 (staged one step ago)
                if (t + 2 < n_my) { stage_row(t + 2); stage_left(t + 2); }     // land during this step
                fwd(t + 1, A);
            }
            // positions 192..195: waves 0, 1 their forward entries of row t+1, waves 2-5 their gradients of row t-1
            if (wave < 2) { if (t + 1 < n_my) left_fwd(t + 1, tid); }
            else if (t >= 1) left_bwd(t - 1, tid - 128, lpre);
            TR_MID();
            dma_landed_barrier();
            TR_END(t);
        }
        if constexpr (BN) {
            if (wave >= 2) {              // waves 2..5 = positions 192..195, lane = channel
                float* rec = srec + (6 + (wave - 2)) * (2 * C);
                rec[lane] = ls1; rec[C + lane] = ls2;
            }
        }
        TR_FLUSH(wave == 0 ? 0 : 3);
    } else if (has_bwd) {
        lds_barrier();
        lds_barrier();
        for (int t = 0; t <= n_my; ++t) {
            TR_BEGIN();
            if (t >= 1) bwd(t - 1);
            TR_MID();
            lds_barrier();                                       // the row stores stay in flight
            TR_END(t);
        }
        flush_bwd_sums();
        TR_FLUSH(1);
    } else {
        float xv0[4], xv1[4];
        load_x(0, xv0);
        lds_barrier();
        lds_barrier();
        auto step = [&](int t, float (&xc)[4], float (&xn)[4]) {
            TR_BEGIN();
            if (t < n_my) {
                if (t + 1 < n_my) load_x(t + 1, xn);
                gather(t, xc);
            }
            TR_MID();
            lds_barrier();
            TR_END(t);
        };
        for (int t = 0; t <= n_my; t += 2) {
            step(t, xv0, xv1);
            if (t + 1 <= n_my) step(t + 1, xv1, xv0);
        }
        TR_FLUSH(2);
    }
    if constexpr (BN) {
        // (every role has passed the same number of barriers; nobody reads P after the last one)
        lds_barrier();
        if (tid < 2 * C) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 10; ++r) acc += srec[r * (2 * C) + tid];
            a.bn_sums[(long long)blockIdx.x * (2 * C) + tid] = acc;
        }
    }
#undef TR_BEGIN
#undef TR_MID
#undef TR_END
#undef TR_FLUSH
}

void launch_mnist_tail_mfma(const MnistTailArgs& a, hipStream_t s) {
    if (a.pipe && a.do_backward && a.C == 64 && a.n_rows >= 2 * a.pipe) {
        constexpr int C = 64;
        const int lds = (2 * 224 * MN_NKP + 2 * MN_GR * MN_GWP + 2 * 224 * (C / 32) + 8 + (C / 8) * 256 + 13 * 64 * (C / 32)) * 4 + 8 * 32 * C * 4;
        static PerDeviceOnce attr;
        if (attr.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mnist_tail_pipe_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef DG_MEASURE
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mnist_tail_pipe2_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#endif
        }
        if (a.pipe_version == 3) {
            const int lds3 = (2 * 196 * MN_NKP + 2 * MN_GR * MN_GWP + 3 * 6 * C + 8 + (C / 8) * 256 + 13 * 64 * (C / 32)) * 4 + 6 * 32 * C * 4 + 6 * 4096 +
                             2 * 4 * C * 4 + 3 * C * 4;
            const int lds3_bn = (4 * C + 2 * 2 * C + 3 * 2 * C) * 4;      // Batchnorm form: constants, converted images and gate bits of 192..195
            static PerDeviceOnce attr3;
            if (attr3.need()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mnist_tail_pipe3_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mnist_tail_pipe3_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3 + lds3_bn);
            }
            if (a.bn_pre) hipLaunchKernelGGL((mnist_tail_pipe3_kernel<64, true>), dim3(a.pipe), dim3(1024), lds3 + lds3_bn, s, a);
            else hipLaunchKernelGGL((mnist_tail_pipe3_kernel<64, false>), dim3(a.pipe), dim3(1024), lds3, s, a);
        }
#ifdef DG_MEASURE
        else if (a.pipe_version == 2) hipLaunchKernelGGL((mnist_tail_pipe2_kernel<64>), dim3(a.pipe), dim3(768), lds, s, a);
#endif
        else hipLaunchKernelGGL((mnist_tail_pipe_kernel<64>), dim3(a.pipe), dim3(768), lds, s, a);
        return;
    }
    const int lds = (224 * MN_NKP + MN_GR * MN_GWP + 224 * (a.C / 32) + 4) * 4;
    if (a.C == 64) hipLaunchKernelGGL((mnist_tail_mfma_kernel<64>), dim3(a.n_rows), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mnist_tail_mfma_kernel<128>), dim3(a.n_rows), dim3(256), lds, s, a);
}

}  // namespace dg
