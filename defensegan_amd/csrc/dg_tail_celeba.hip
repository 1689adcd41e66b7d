// CelebA generator tail (Generator.6: 64 -> 3 channels, 64 x 64, tanh, loss, backward to da5), gfx950.
// Formulation and the shared device helpers: dg_tail_common.h.  Reference: models/dataset_models.py:160-163, models/gan.py:410-414.
#include <type_traits>
#include "dg_tail_common.h"

namespace dg {

// =================================================================================================================
// CelebA: Generator.6 (C -> 3, 32x32 -> 64x64) + tanh + loss ; backward to da5 (no ReluGrad: Generator.5 is linear)
// =================================================================================================================
constexpr int CE_NKP = 99;       // 75 kappa columns padded to 96 (+3: gather reads are <= 2-way bank conflicted)
constexpr int CE_GWP = 68;       // da6 image pitch (cols are image index + 1, 67 used)

#ifdef DG_MEASURE   // the 32x32x2 formulation, superseded by celeba_tail_fwd16_kernel: kept as a cross-check (option tail_fwd16 = 0)
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
    if (DG_DBG(a) != 2) {
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
    if (DG_DBG(a) != 1) {
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
    if (tid < 4) a.loss_part[((long long)n * 8 + band) * 4 + tid] = tid == 0 ? ((sred[0] + sred[1]) + (sred[2] + sred[3])) + (sred[4] + sred[5]) : 0.f;
}

#endif  // DG_MEASURE

// ---- forward tail, second formulation: v_mfma_f32_16x16x4_f32 with kh-aligned kappa tiles ------------------------------
// The 75 filter columns are regrouped per filter row kh: tile kh holds kappa' = kw*3 + co (15 columns, padded to 16).
// A band of 8 output rows i = 2*oh + kh - 1 needs, of its 6 input rows (local lr = oh - (4*band - 1)), only
//     lr0: kh 3,4   lr1: kh 1..4   lr2, lr3: kh 0..4   lr4: kh 0..2   lr5: kh 0          (20 of 30 (row, kh) units)
// so a third of the padded GEMM of the 32-wide formulation is never issued, and the 20 units split 5/5/5/5 over four
// waves: w0 = lr2, w1 = lr3, w2 = lr1 + lr5, w3 = lr4 + lr0.  Each unit's P block [32 positions][16] (pitch 17) has its
// own LDS slot; slot = CE16_BASE[lr] + kh.
// Measured alternatives to this three-workgroups-per-CU shape (188 us at N = 1280): one persistent workgroup per CU with
// dedicated GEMM and gather waves and a double-buffered P -- (a) 12 waves, rows staged through a single 48 KB LDS stage
// (all that fits next to two P buffers): 225 us, the DMA latency sits on every step's critical path; (b) 8 waves, next
// band's A fragments prefetched into registers by fragment-shaped global loads: 211 us, the single GEMM wave per SIMD
// spends longer issuing its 16 loads than multiplying.  Both were bit-identical to this kernel and removed.
// Round 2: (c) all five units' filter fragments requested up front (80 registers, no memory wait inside the GEMM phase):
// 217 us -- the kernel is bound by VMEM issue bursts, not by the per-unit filter round trip; (d) distinct wave priorities
// per resident workgroup (wg_priority) to stagger the three workgroups' phases: 190-193 us.  Phase costs (tail_dbg): without
// the gather 153 us, without the GEMM phase 64 us, without the staging DMA 161 us.  Phase trace of wave 0 (TRACE instantiation,
// tools/tail_trace.py fwd): a workgroup lives 29.6 k cycles (12.9 us; 13.3 workgroups per slot): issuing its x / DMA / filter
// loads 15.8 %, waiting for the staged rows 5.7 %, fragment reads 4.2 %, barrier 5.9 %, the five GEMM units 26.0 % (7.7 k cycles
// for 5.1 k cycles of MFMA issue), barrier 2.5 %, gather + tanh + stores 30.2 %, loss reduction and exit 9.8 %.  Three
// workgroups per CU (LDS) = 3 waves per SIMD overlap these serial phases to 51 % MFMA occupancy.  (e) a persistent form that
// keeps all filter fragments in registers needs 80 + 64 + 16 registers before addressing: spills at 3 waves per SIMD.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// da6 = d(loss)/d(pre-activation) of one output: (2/P) (y - x) (1 - y^2), as three rounded products / one difference.  Pinned
// (no fused multiply-add contraction) so that every formulation of the forward tail produces the same bits.
__device__ __forceinline__ float celeba_da6(float gscale, float d, float y) {
#pragma clang fp contract(off)
    const float yy = y * y;
    return (gscale * d) * (1.0f - yy);
}
constexpr int CE16_PITCH = 17;
constexpr int CE16_UNIT = 32 * CE16_PITCH;                 // floats per (row, kh) unit
constexpr int CE16_UNITS = 20;
#ifndef CE16_DMA_AUX
#define CE16_DMA_AUX 2                                       // nt: the streamed activation rows must not evict the filters from L1
#endif
// base slot per local row, 5 bits each: lr0 -> 15 (kh 3,4 -> 18,19), lr1 -> 9 (kh 1..4 -> 10..13), lr2 -> 0, lr3 -> 5,
// lr4 -> 15 (kh 0..2 -> 15..17), lr5 -> 14
constexpr unsigned CE16_BASE = 15u | (9u << 5) | (0u << 10) | (5u << 15) | (15u << 20) | (14u << 25);

// TRACE (option tail_trace with tail_dbg = 8; tools/tail_trace.py fwd): wave 0 of workgroups 3072 .. 6143 (past the launch's
// ramp) records the cycle counter at the phase boundaries.  A separate instantiation: the 18 extra registers must not reach
// the product kernel (126 + 8 registers: 3 waves per SIMD).
template <int C, bool TRACE>
__global__ __launch_bounds__(256) void celeba_tail_fwd16_kernel(CelebaTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KK = C / 16;                                  // channel groups of 16 (4 MFMAs each)
    constexpr int CH = C / 4;                                   // 16-B chunks per position
    constexpr int ROWB = 32 * C * 4;                            // bytes of one staged input row
    float* sP = reinterpret_cast<float*>(smem);                // [20][32][17], aliases the staging area
    constexpr int MAINF = (6 * ROWB > CE16_UNITS * CE16_UNIT * 4 ? 6 * ROWB : CE16_UNITS * CE16_UNIT * 4) / 4;
#ifdef DG_MEASURE
    wg_priority(a.prio);
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x >> 3, band = blockIdx.x & 7;
    const bool tr = TRACE && DG_TRACE_PTR(a) != nullptr && wave == 0 && blockIdx.x >= 3072 && blockIdx.x < 6144;
    long long tc[TRACE ? 9 : 1];
    auto mark = [&](int i) { if constexpr (TRACE) { if (tr) tc[i] = (long long)__builtin_readcyclecounter(); } };
    mark(0);
    const int b = n / a.R;
    const float* hrow = a.h5 + (long long)n * (1024 * C);
    const int oh_lo = 4 * band - 1;
    const float* xrow = a.x + (long long)b * 12288;
    // Output ownership: waves 0-2 own one (column j, channel co) pair each (192 pairs) for output rows 0..5 of the band,
    // wave 3 owns three pairs per lane for rows 6, 7 -- the row is wave-uniform, the column terms are per-thread constants.
    // x is fetched now and consumed after the GEMM phase.
    const bool tailw = wave == 3;
    int cjs[3];
    cjs[0] = tailw ? 3 * lane : tid;
    cjs[1] = tailw ? 3 * lane + 1 : tid;
    cjs[2] = tailw ? 3 * lane + 2 : tid;
    float xv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int il = tailw ? 6 + r / 3 : r;
        xv[r] = xrow[(8 * band + il) * 192 + cjs[tailw ? r % 3 : 0]];
    }

    // wave roles: (first row, kh range), (second row, kh range)
    const int lrA = wave == 0 ? 2 : wave == 1 ? 3 : wave == 2 ? 1 : 4;
    const int lrB = wave == 2 ? 5 : wave == 3 ? 0 : -1;
    const int khA_lo = wave == 2 ? 1 : 0, khA_hi = wave == 3 ? 3 : 5;
    const int ohA = oh_lo + lrA, ohB = oh_lo + lrB;
    const bool inA = ohA >= 0 && ohA < 32;                      // always true (lr 1..4), kept for symmetry
    const bool inB = lrB >= 0 && ohB >= 0 && ohB < 32;

    // ---- stage this wave's input rows with full-line LDS-DMA (chunk index XOR-swizzled on the source side) -----------
    constexpr int NI = 32 * CH / 64;                            // DMA instructions per row
    char* stA = smem + (wave < 2 ? wave : wave == 2 ? 2 : 4) * ROWB;
    char* stB = smem + (wave == 2 ? 3 : 5) * ROWB;
    if (DG_DBG(a) != 2 && DG_DBG(a) != 4) {
        if (inA) {
            const char* src = reinterpret_cast<const char*>(hrow + (long long)ohA * 32 * C);
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const int slot = q * 64 + lane;
                const int pos = slot / CH, c = slot % CH;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4)),
                    (__attribute__((address_space(3))) void*)(stA + q * 1024), 16, 0, CE16_DMA_AUX);
            }
        }
        if (inB) {
            const char* src = reinterpret_cast<const char*>(hrow + (long long)ohB * 32 * C);
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const int slot = q * 64 + lane;
                const int pos = slot / CH, c = slot % CH;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4)),
                    (__attribute__((address_space(3))) void*)(stB + q * 1024), 16, 0, CE16_DMA_AUX);
            }
        }
    }
    // filter fragments of the first unit are requested before the staging wait
    auto kh_of = [&](int st) { return wave == 2 ? (st + 1) % 5 : st; };
    auto load_w = [&](f32x4v (&w)[KK], int kh) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            w[kk] = *reinterpret_cast<const f32x4v*>(a.F6p + (((long long)kh * KK + kk) * 64 + lane) * 4);
    };
    f32x4v wa[KK], wb[KK];
    load_w(wa, kh_of(0));
    // A fragments: lane (i = lane & 15, g = lane >> 4) of position tile m holds channels 16*kk + 4*g + e
    const int fi = lane & 15, fg = lane >> 4;
    f32x4v avA[2][KK], avB[2][KK];
    mark(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mark(2);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int pos = 16 * m + fi;
            const int off = pos * (C * 4) + (((4 * kk + fg) ^ (pos & (CH - 1))) << 4);
            f32x4v va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (inA) va = *reinterpret_cast<const f32x4v*>(stA + off);
            if (inB) vb = *reinterpret_cast<const f32x4v*>(stB + off);
            avA[m][kk] = va;
            avB[m][kk] = vb;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mark(3);
    __syncthreads();                                            // staging is dead: P may overwrite it
    mark(4);

    // ---- GEMM: 5 (row, kh) units per wave, each 2 position tiles x KK*4 MFMAs (two independent accumulation chains);
    // the next unit's filter fragments are in flight while the current one runs -------------------------------------------
    // unit sequence: kh = (wave == 2 ? 1,2,3,4,0 : 0,1,2,3,4); the second row takes over at step nA
    if (DG_DBG(a) != 2) {
        const int nA = khA_hi - khA_lo;
        auto unit = [&](const f32x4v (&av)[2][KK], const f32x4v (&w)[KK], int lr, int kh) {
            f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][kk][e], w[kk][e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][kk][e], w[kk][e], acc1, 0, 0, 0);
                }
            // D layout: col = lane & 15 (kappa'), row = 4 * (lane >> 4) + reg (position within the tile)
            float* pu = sP + (((CE16_BASE >> (5 * lr)) & 31) + kh) * CE16_UNIT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pu[(4 * fg + r) * CE16_PITCH + fi] = acc0[r];
                pu[(16 + 4 * fg + r) * CE16_PITCH + fi] = acc1[r];
            }
        };
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            f32x4v (&wc)[KK] = (st & 1) ? wb : wa;
            f32x4v (&wn)[KK] = (st & 1) ? wa : wb;
            if (st + 1 < 5) load_w(wn, kh_of(st + 1));
            const int kh = kh_of(st);
            if (st < nA) { if (inA) unit(avA, wc, lrA, kh); }
            else if (inB) unit(avB, wc, lrB, kh);
        }
    }
    mark(5);
    __syncthreads();
    mark(6);

    // ---- gather (taps of matching parity) + tanh + loss + da6 ---------------------------------------------------------
    float* grow = a.g6 + (long long)n * 12288;
    float* yrow = a.y ? a.y + (long long)n * 12288 : nullptr;
    const float gscale = 2.0f / 12288.0f;
    float sq = 0.f;
    if (DG_DBG(a) != 1) {
        // per-pair column terms: offsets ow*17 + kw*3 + co of the <= 3 taps kw = kw0 + 2*aw
        int colofs[3][3];
        float bias[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            const int cj = cjs[c3];
            const int j = cj / 3, co = cj - 3 * j;
            const int kw0 = (j + 1) & 1;
            bias[c3] = a.b6[co];
#pragma unroll
            for (int aw = 0; aw < 3; ++aw) {
                const int kw = kw0 + 2 * aw;
                const int ow = (j + 1 - kw) >> 1;
                // a tap that does not exist reads the unit's zero pad column (kappa' = 15, zero filter column) instead
                colofs[c3][aw] = (kw > 4 || ow < 0 || ow >= 32) ? 15 : ow * CE16_PITCH + kw * 3 + co;
            }
        }
        // All 54 candidate taps of this thread's 6 outputs are read first and waited for once; left alone hipcc reads, waits
        // and sums output by output (6 exposed LDS round trips while two other workgroups of the CU hammer the LDS).  The
        // A fragments of the GEMM phase are dead by now, so the registers are there.  Sums keep their order.
        float tv[6][9];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int il = tailw ? 6 + r / 3 : r;                  // wave-uniform
            const int c3 = tailw ? r % 3 : 0;
            const int i = 8 * band + il;
            const int kh0 = (i + 1) & 1;
#pragma unroll
            for (int ah = 0; ah < 3; ++ah) {
                const int kh = kh0 + 2 * ah;
                const int oh = (i + 1 - kh) >> 1;
                const bool skip = kh > 4 || oh < 0 || oh >= 32;     // uniform; a skipped row reads slot 0 and is not summed
                const int lr = skip ? 2 : oh - oh_lo;
                const float* pu = sP + (((CE16_BASE >> (5 * lr)) & 31) + (skip ? 0 : kh)) * CE16_UNIT;
#pragma unroll
                for (int aw = 0; aw < 3; ++aw) tv[r][ah * 3 + aw] = pu[tailw ? colofs[c3][aw] : colofs[0][aw]];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) asm volatile("" : "+v"(tv[r][t9]));
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int il = tailw ? 6 + r / 3 : r;                  // wave-uniform
            const int c3 = tailw ? r % 3 : 0;
            const int i = 8 * band + il;
            const int kh0 = (i + 1) & 1;
            float sacc = 0.f;
#pragma unroll
            for (int ah = 0; ah < 3; ++ah) {
                const int kh = kh0 + 2 * ah;
                const int oh = (i + 1 - kh) >> 1;
                if (kh > 4 || oh < 0 || oh >= 32) continue;          // uniform
#pragma unroll
                for (int aw = 0; aw < 3; ++aw) sacc += tv[r][ah * 3 + aw];
            }
            // tanh(v) = sign(v) * (1 - t) / (1 + t), t = exp(-2|v|) = exp2(-2 log2(e) |v|): v_exp_f32 and v_rcp_f32 (1 ulp each)
            // instead of the library expf and an IEEE division -- absolute error <= 3 ulp(1) = 3.6e-7 (the gate on y against the
            // float64 oracle is 2e-6, tests/test_gpu_celeba_bn.py), 40 instructions fewer per output
            const float v = sacc + (tailw ? bias[c3] : bias[0]);
            const float t = __builtin_amdgcn_exp2f(-2.8853900817779268f * __builtin_fabsf(v));
            const float y = __builtin_copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
            const float d = y - xv[r];
            sq = __builtin_fmaf(d, d, sq);
            const int oi = i * 192 + (tailw ? cjs[c3] : cjs[0]);
            grow[oi] = celeba_da6(gscale, d, y);
            if (yrow) yrow[oi] = y;
        }
    }
    mark(7);
    // per-wave partial sums: celeba_loss_finish_kernel adds them as ((w0 + w1) + (w2 + w3)), band by band -- the order a
    // workgroup-level reduction here used to have, without its LDS round trip and barrier at the end of every workgroup.
    // Only when somebody reads the loss of this launch (the last forward pass of a projection, dg_loss_grad).
    if (a.want_loss) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        if (lane == 0) a.loss_part[((long long)n * 8 + band) * 4 + wave] = sq;
    }
    if constexpr (TRACE) {
        if (tr && lane == 0) {
            mark(8);
            long long* o = DG_TRACE_PTR(a) + (long long)(blockIdx.x - 3072 + 1024) * 8;     // rows 1024 .. 4095 of the [4096][8] buffer
            for (int i = 0; i < 8; ++i) o[i] = tc[i + 1] - tc[i];
        }
    }
}

// ---- forward tail, third formulation: persistent, ROLE-SPLIT workgroups on half-bands -----------------------------------
// celeba_tail_fwd16_kernel runs stage -> GEMM -> gather strictly in sequence inside a workgroup and leaves it to three resident
// workgroups per CU to overlap each other's phases (MFMA pipe 43 % busy).  Here a workgroup is 8 waves with two roles and
// walks a strided list of items (latent row n, half-band hb = 4 output rows):
//     waves 0-3 ("M"): stage the item's input rows by LDS-DMA, multiply;   waves 4-7 ("G"): gather + tanh + loss + da6
// In step s the M waves multiply item s + 1 (results stay in registers) while the G waves gather item s from P; then
// barrier, the M waves drop their 5 accumulators into P, barrier.  A half-band's 4 output rows i = 4 hb + il need the input rows
// oh = 2 hb - 1 + lr, lr = 0..3, through 10 (lr, kh) units
//     lr0: kh 3,4     lr1: kh 1..4     lr2: kh 0..2     lr3: kh 0               (i = 2 oh + kh - 1)
// = 20 half-units of 16 positions, 5 per M wave: wave w = 2 c + m takes position half m of the rows of class c
//     c = 0: (lr1; kh 1,2,3,4) + (lr3; kh 0)          c = 1: (lr2; kh 0,1,2) + (lr0; kh 3,4)
// so every wave stages exactly the two half-rows it multiplies (8 LDS-DMA pieces; no other wave reads them, so no barrier
// sits between staging and the fragment reads), and its five half-units are five independent accumulation chains.  The
// filter fragments (all five kh, 20 KB) live in LDS for the workgroup's life: the M waves issue no ordinary load inside the
// loop, so their vmcnt counts LDS-DMA pieces only and item s + 2's rows are in flight for a whole step (a filter load
// issued behind a DMA would have to wait for it: VMEM returns in order).  LDS: 32 KB stage + 21.8 KB P + 20 KB filters = 73.3 KB,
// two workgroups per CU.  G wave g owns output row il = g (its kh terms are wave-uniform), lane l the column j = l with its three
// channels (12 contiguous bytes per lane for x, y and da6).  Sums keep the order of celeba_tail_fwd16_kernel (kh ascending, kw
// ascending; one k-ordered MFMA chain per P entry): y and da6 are bit-identical to it.
// Measured (N = 1280, one box, A/B): 176.9 -> 137.8 us at 512 workgroups (2 per CU; 448 / 480 / 768: 169 / 159 / 157).  In-kernel
// trace (tools/tail_trace_split.py, cycles per step at ~2.0 GHz): M waves 4.06 k for fragment reads + 8 DMA pieces + 80 MFMAs
// (2.56 k of matrix-pipe time), 1.6 k waiting for the G waves, 1.6 k for the P stores and the second barrier; G waves 5.4 k per
// gather: the G waves are the longer side, and the SIMD's issue rate is the resource (a step issues ~2600 wave-instructions per
// workgroup; a SIMD issues one per ~4 cycles): each round of instruction-count reduction (per-row / per-border instantiations
// with immediate LDS offsets, buffer-descriptor DMA without address VALU, the squared-error reduction only when the loss is
// read) bought 3-5 us.  The step is two barriers; a double-buffered P (one barrier) does not fit two workgroups per CU.
constexpr int CES_UNITS = 10;
constexpr int CES_PBUF = CES_UNITS * CE16_UNIT;          // floats of the P buffer
// P slot of unit (lr, kh): lr0 -> kh - 3, lr1 -> 1 + kh, lr2 -> 6 + kh, lr3 -> 9
__device__ __forceinline__ constexpr int ces_slot(int lr, int kh) { return lr == 0 ? kh - 3 : lr == 1 ? 1 + kh : lr == 2 ? 6 + kh : 9; }

// (the body is a __device__ function: hipcc's host pass instantiates the body of a __global__ template and knows neither
// __amdgpu_buffer_rsrc_t nor the buffer-load builtins -- the kernel would silently lose its host stub)
template <int C>
__device__ __forceinline__ void celeba_tail_fwd_split_body(const CelebaTailArgs& a, int n_items, char* smem) {
    static_assert(C == 64, "position half = 16 positions x 64 channels = one 4 KB run; other widths use celeba_tail_fwd16_kernel");
    constexpr int KK = C / 16, CH = C / 4, ROWB = 32 * C * 4;
    char* stage = smem;                                               // [4 rows][32 positions][C]
    float* sP = reinterpret_cast<float*>(smem + 4 * ROWB);            // [CES_UNITS][32][17]
    float* sW = sP + CES_PBUF;                                        // filter fragments [5 kh][KK][64 lanes][4] (= the pack's layout)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_my = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto item_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };
    for (int i = tid; i < 5 * KK * 64; i += 512)
        reinterpret_cast<f32x4v*>(sW)[i] = reinterpret_cast<const f32x4v*>(a.F6p)[i];
    if (tid < CES_UNITS) sP[tid * CE16_UNIT + 16] = 0.f;          // the word taps that do not exist read (see the G role)
    // (Measured and removed: delaying one of a CU's two workgroups by 2-8 k cycles to de-phase them, 152.0-153.4 vs 152.1 us;
    // s_setprio 2 for the M waves, 149.6 vs 150.3 us.)
    if (wave < 4) {
        // ================================================ M role ================================================
        const int m = wave & 1;
        const bool cls0 = wave < 2;
        const int lrA = cls0 ? 1 : 2, lrB = cls0 ? 3 : 0;
        const int fi = lane & 15, fg = lane >> 4;
        char* stA = stage + lrA * ROWB + m * 16 * (C * 4);
        char* stB = stage + lrB * ROWB + m * 16 * (C * 4);
        auto row_ok = [&](int item, int lr) { const int oh = 2 * (item & 15) - 1 + lr; return oh >= 0 && oh < 32; };
        // LDS-DMA through a buffer descriptor (base = the latent row's input map): the per-lane byte offsets of the four 1 KB
        // pieces of a half-row are computed once, the half-row's offset rides in an SGPR -- no address VALU per piece
        unsigned voff[16 * CH / 64];
#pragma unroll
        for (int q = 0; q < 16 * CH / 64; ++q) {
            const int slot = q * 64 + lane;
            const int pos = slot / CH, c = slot % CH;
            voff[q] = (unsigned)(pos * (C * 4) + ((c ^ (pos & (CH - 1))) << 4));
        }
        auto stage_half = [&](int item, int lr, char* dst) {
            if (!row_ok(item, lr)) return;
            const int n = item >> 4, oh = 2 * (item & 15) - 1 + lr;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.h5 + (long long)n * (1024 * C)), 0, 1024 * C * 4, 0x00020000);
            const int soff = (oh * 32 + 16 * m) * (C * 4);
#pragma unroll
            for (int q = 0; q < 16 * CH / 64; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, voff[q], soff, 0,
                                                         CE16_DMA_AUX);
        };
        auto read_frags = [&](const char* st, f32x4v (&av)[KK]) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                av[kk] = *reinterpret_cast<const f32x4v*>(st + fi * (C * 4) + (((4 * kk + fg) ^ (fi & (CH - 1))) << 4));
        };
        const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4v acc[5];
        bool okB = false;
        // five half-units = five independent accumulation chains; chain c multiplies row (c < NA ? A : B) by filter row KH[c]
        auto compute = [&](int item) {
            f32x4v avA[KK], avB[KK];
            okB = row_ok(item, lrB);
            read_frags(stA, avA);                                      // row A always exists (lr 1, 2)
            if (okB) read_frags(stB, avB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (item + (int)gridDim.x < n_items) {                     // the next item's half-rows go out now: a whole step to land
                stage_half(item + (int)gridDim.x, lrA, stA);
                stage_half(item + (int)gridDim.x, lrB, stB);
            }
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = z4;
            // Specialised on (class, row B present): no branch sits between the MFMAs; the filter fragments of k-group kk + 1 are
            // requested before the 20 (16) MFMAs of k-group kk and pinned there (left alone hipcc emits read - wait - multiply).
            auto chains = [&](auto cls_tag, auto hasb_tag) {
                constexpr bool C0 = decltype(cls_tag)::value;
                constexpr bool HB = decltype(hasb_tag)::value;
                constexpr int NA = C0 ? 4 : 3;
                constexpr int NC = HB ? 5 : NA;
                f32x4v wk[2][5];
                auto read_w = [&](int kk, f32x4v (&dst)[5]) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int kh = C0 ? (c < 4 ? c + 1 : 0) : c;
                        dst[c] = *reinterpret_cast<const f32x4v*>(sW + (((kh * KK + kk) * 64 + lane) << 2));
                    }
                };
                read_w(0, wk[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, NC, 0);
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    if (kk + 1 < KK) read_w(kk + 1, wk[(kk + 1) & 1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            if (c < NA) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(avA[kk][e], wk[kk & 1][c][e], acc[c], 0, 0, 0);
                            else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(avB[kk][e], wk[kk & 1][c][e], acc[c], 0, 0, 0);
                        }
                    if (kk + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, NC, 0);      // DS reads of kk + 1 first ...
                    __builtin_amdgcn_sched_group_barrier(0x8, 4 * NC, 0);                     // ... then this k-group's MFMAs
                }
            };
            if (cls0) { if (okB) chains(std::true_type(), std::true_type()); else chains(std::true_type(), std::false_type()); }
            else { if (okB) chains(std::false_type(), std::true_type()); else chains(std::false_type(), std::false_type()); }
        };
        float* const pw = sP + (16 * m + 4 * fg) * CE16_PITCH + fi;     // this lane's first P entry inside a unit
        auto store_cls = [&](auto cls_tag) {
            constexpr bool C0 = decltype(cls_tag)::value;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                constexpr int NA = C0 ? 4 : 3;
                const int slot = C0 ? (c < 4 ? ces_slot(1, c + 1) : ces_slot(3, 0)) : (c < 3 ? ces_slot(2, c) : ces_slot(0, c));
                if (c < NA || okB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pw[slot * CE16_UNIT + r * CE16_PITCH] = acc[c][r];   // immediate offsets
                }
            }
        };
        auto store_p = [&]() { if (cls0) store_cls(std::true_type()); else store_cls(std::false_type()); };
        // prologue: rows of item 0; every compute(item k) sends item k + 1's rows out as soon as it has read its own
        stage_half(item_of(0), lrA, stA);
        stage_half(item_of(0), lrB, stB);
        lds_barrier();                                               // filter fragments are in LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        compute(item_of(0));
        store_p();
        lds_barrier();
#ifdef DG_MEASURE
        const bool tr = a.trace != nullptr && wave == 0 && blockIdx.x < 2048;
        long long ph[4] = {0, 0, 0, 0};
#endif
        for (int s = 0; s < n_my; ++s) {
            const bool more = s + 1 < n_my;
#ifdef DG_MEASURE
            long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (tr) c0 = (long long)__builtin_readcyclecounter();
#endif
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // item s + 1's half-rows (issued one step ago) have landed
#ifdef DG_MEASURE
                if (tr) c1 = (long long)__builtin_readcyclecounter();
#endif
                compute(item_of(s + 1));
            }
#ifdef DG_MEASURE
            if (tr) c2 = (long long)__builtin_readcyclecounter();
#endif
            lds_barrier();                                           // the G waves have read P of item s
#ifdef DG_MEASURE
            if (tr) c3 = (long long)__builtin_readcyclecounter();
#endif
            if (more) store_p();
            lds_barrier();
#ifdef DG_MEASURE
            if (tr && more && s >= 2) {
                ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += (long long)__builtin_readcyclecounter() - c3;
            }
#endif
        }
#ifdef DG_MEASURE
        if (tr && lane == 0) {
            long long* o = a.trace + (long long)blockIdx.x * 16;
            for (int i = 0; i < 4; ++i) o[i] = ph[i];
            o[4] = n_my > 3 ? n_my - 3 : 0;
        }
#endif
    } else {
        // ================================================ G role ================================================
        const int g = wave - 4;                                          // output row il of the half-band
        const int j = lane;                                              // output column; channels co = 0..2
        const float gscale = 2.0f / 12288.0f;
        // column terms: offsets ow * 17 + kw * 3 + co of the <= 3 taps kw = kw0 + 2 aw; a tap that does not exist reads word 16 of
        // the unit -- the pitch pad of position 0, which no MFMA result is ever stored to and which is zeroed explicitly when the
        // workgroup starts (a COMPUTED zero, position 0 times the pack's zero 16th column, would turn an Inf / NaN in that
        // input position into NaNs along the whole image border)
        int colofs[3][3];
        float bias[3];
        {
            const int kw0 = (j + 1) & 1;
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                bias[co] = a.b6[co];
#pragma unroll
                for (int aw = 0; aw < 3; ++aw) {
                    const int kw = kw0 + 2 * aw;
                    const int ow = (j + 1 - kw) >> 1;
                    colofs[co][aw] = (kw > 4 || ow < 0 || ow >= 32) ? 16 : ow * CE16_PITCH + kw * 3 + co;
                }
            }
        }
        // kh terms of row il = g in ascending kh (the order celeba_tail_fwd16_kernel adds them in): P slot and input row lr
        //   il 0: kh 1 (lr1), 3 (lr0)      il 1: kh 0 (lr2), 2 (lr1), 4 (lr0)      il 2: kh 1 (lr2), 3 (lr1)      il 3: kh 0 (lr3), 2 (lr2), 4 (lr1)
        // The gather is instantiated per row (G is a compile-time constant): the P slots become immediate offsets of the reads.
        // image of a latent row: n / R by a 40-bit magic multiply on the scalar unit (exact for n < 2^24, R < 2^16)
        const unsigned long long magicR = ((1ULL << 40) + (unsigned)a.R - 1) / (unsigned)a.R;
        const int j3 = 3 * j;
        auto load_x = [&](int item, float (&xv)[3]) {
            const int n = item >> 4, hb = item & 15;
            const long long b = (long long)(((unsigned long long)(unsigned)n * magicR) >> 40);
            const float* xrow = a.x + (b * 12288 + (4 * hb + g) * 192);          // wave-uniform base, lane offset j3
#pragma unroll
            for (int co = 0; co < 3; ++co) xv[co] = xrow[j3 + co];
        };
        // Instantiated per (row G, missing input row): every P slot is an immediate offset of its read and nothing is masked.
        // MISS = 1: hb == 0, input row lr0 (oh = -1) does not exist;  MISS = 2: hb == 15, lr3 (oh = 32) does not exist.
        auto gather_g = [&](auto g_tag, auto miss_tag, int item, const float (&xv)[3]) {
            constexpr int G = decltype(g_tag)::value;
            constexpr int MISS = decltype(miss_tag)::value;
            constexpr int NT = (G & 1) ? 3 : 2;
            constexpr int TS[4][3] = {{ces_slot(1, 1), ces_slot(0, 3), ces_slot(1, 1)}, {ces_slot(2, 0), ces_slot(1, 2), ces_slot(0, 4)},
                                      {ces_slot(2, 1), ces_slot(1, 3), ces_slot(2, 1)}, {ces_slot(3, 0), ces_slot(2, 2), ces_slot(1, 4)}};
            constexpr int TL[4][3] = {{1, 0, 1}, {2, 1, 0}, {2, 1, 2}, {3, 2, 1}};
            const int n = item >> 4, hb = item & 15;
            float tv[3][9];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if ((MISS == 1 && TL[G][t] == 0) || (MISS == 2 && TL[G][t] == 3)) continue;
                const float* pu = sP + TS[G][t] * CE16_UNIT;
#pragma unroll
                for (int co = 0; co < 3; ++co)
#pragma unroll
                    for (int aw = 0; aw < 3; ++aw) tv[co][t * 3 + aw] = pu[colofs[co][aw]];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float sq = 0.f;
            float yv[3], gv[3];
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                float sacc = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if ((MISS == 1 && TL[G][t] == 0) || (MISS == 2 && TL[G][t] == 3)) continue;
#pragma unroll
                    for (int aw = 0; aw < 3; ++aw) sacc += tv[co][t * 3 + aw];
                }
                const float v = sacc + bias[co];
                const float tt = __builtin_amdgcn_exp2f(-2.8853900817779268f * __builtin_fabsf(v));
                const float y = __builtin_copysignf((1.0f - tt) * __builtin_amdgcn_rcpf(1.0f + tt), v);
                const float d = y - xv[co];
                sq = __builtin_fmaf(d, d, sq);
                yv[co] = y;
                gv[co] = celeba_da6(gscale, d, y);
            }
            const long long ob = (long long)n * 12288 + (4 * hb + G) * 192;           // wave-uniform
            float* grow = a.g6 + ob;
#pragma unroll
            for (int co = 0; co < 3; ++co) grow[j3 + co] = gv[co];
            if (a.y) {
                float* yrow = a.y + ob;
#pragma unroll
                for (int co = 0; co < 3; ++co) yrow[j3 + co] = yv[co];
            }
            if (a.want_loss) {          // read only after the last forward pass of a projection and by dg_loss_grad
#pragma unroll
                for (int mm = 32; mm >= 1; mm >>= 1) sq += __shfl_xor(sq, mm, 64);
                if (lane == 0) a.loss_part[((long long)n * 16 + hb) * 4 + G] = sq;
            }
        };
        auto gather_m = [&](auto g_tag, int item, const float (&xv)[3]) {
            const int hb = item & 15;
            if (hb == 0) gather_g(g_tag, std::integral_constant<int, 1>(), item, xv);
            else if (hb == 15) gather_g(g_tag, std::integral_constant<int, 2>(), item, xv);
            else gather_g(g_tag, std::integral_constant<int, 0>(), item, xv);
        };
        auto gather = [&](int item, const float (&xv)[3]) {
            if (g == 0) gather_m(std::integral_constant<int, 0>(), item, xv);
            else if (g == 1) gather_m(std::integral_constant<int, 1>(), item, xv);
            else if (g == 2) gather_m(std::integral_constant<int, 2>(), item, xv);
            else gather_m(std::integral_constant<int, 3>(), item, xv);
        };
        lds_barrier();
        lds_barrier();
#ifdef DG_MEASURE
        const bool tr = a.trace != nullptr && wave == 4 && blockIdx.x < 2048;
        long long gph[3] = {0, 0, 0};
#endif
        for (int s = 0; s < n_my; ++s) {
            // x of this item is requested first and used last (after the LDS reads and the tanh): its L2 latency sits under them.
            // (Requesting it a step ahead does not help: the wait for it is a vmcnt(0), which would then also wait for the
            // request just issued for the step after.)
#ifdef DG_MEASURE
            long long c0 = 0, c1 = 0, c2 = 0;
            if (tr) c0 = (long long)__builtin_readcyclecounter();
#endif
            float xv[3];
            load_x(item_of(s), xv);
            gather(item_of(s), xv);
#ifdef DG_MEASURE
            if (tr) c1 = (long long)__builtin_readcyclecounter();
#endif
            lds_barrier();
#ifdef DG_MEASURE
            if (tr) c2 = (long long)__builtin_readcyclecounter();
#endif
            lds_barrier();
#ifdef DG_MEASURE
            if (tr && s >= 2 && s + 1 < n_my) { gph[0] += c1 - c0; gph[1] += c2 - c1; gph[2] += (long long)__builtin_readcyclecounter() - c2; }
#endif
        }
#ifdef DG_MEASURE
        if (tr && lane == 0) {
            long long* o = a.trace + (long long)blockIdx.x * 16 + 8;
            for (int i = 0; i < 3; ++i) o[i] = gph[i];
        }
#endif
    }
}

template <int C>
__global__ __launch_bounds__(512, 4) void celeba_tail_fwd_split_kernel(CelebaTailArgs a, int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    celeba_tail_fwd_split_body<C>(a, n_items, smem);
}

#ifdef DG_MEASURE   // the per-band backward kernel, superseded by the persistent one: kept as a cross-check (option tail_bwd_persist = 0)
// NB consecutive 4-input-row bands per workgroup: the filter fragments (76 registers) and the launch/ramp cost are
// paid once per NB * 128 positions.
template <int C, int NB>
__global__ __launch_bounds__(256) void celeba_tail_bwd_mfma_kernel(CelebaTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GROWS = 8 * NB + 3;
    float* sg = reinterpret_cast<float*>(smem);                  // [GROWS][68][3]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int GPR = 8 / NB;                                  // workgroups per latent row
    const int n = blockIdx.x / GPR, grp = blockIdx.x % GPR;
    const float* grow = a.g6 + (long long)n * 12288;
    const int i_lo = 8 * NB * grp - 1;
    for (int i = tid; i < GROWS * CE_GWP * 3; i += 256) {
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
    // 4 * NB position tiles: local input row ohl = wave + 4 * t, 32 positions each
#pragma unroll 1
    for (int t = 0; t < NB; ++t) {
        const int ohl = wave + 4 * t, ow = frow;
        f32x16 acc[C / 32];
        tail_bwd_tile<C, 3, CE_GWP>(sg, ((2 * ohl) * CE_GWP + 2 * ow) * 3, true, bw, acc, lane);
        const int oh = 4 * NB * grp + ohl;
#pragma unroll
        for (int u = 0; u < C / 32; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int owr = (e & 3) + 8 * (e >> 2) + 4 * fh;
                hrow[(oh * 32 + owr) * C + u * 32 + frow] = acc[u][e];
            }
    }
}

#endif  // DG_MEASURE

// ---- backward tail, persistent and software-pipelined ---------------------------------------------------------------
// The per-band workgroup above is three serial phases (fetch da6 + filters, 76 MFMAs per wave, 32 row stores) and every
// workgroup on the chip runs them in lockstep, so the MFMA pipe idles during the other two.  Here a workgroup keeps the
// filter fragments in registers for its whole life, walks a strided list of (latent row, band) items, fetches item k+1's
// da6 image into registers while item k's MFMAs run (double-buffered LDS image, one barrier per item), and leaves its
// stores in flight.
constexpr int CEB_ROWS = 11, CEB_ROWF = CE_GWP * 3;          // image rows per band, floats per image row (204)
constexpr int CEB_IMG = CEB_ROWS * CEB_ROWF;                // 2244 floats
constexpr int CEB_PF = (CEB_IMG + 255) / 256;               // prefetch registers per thread (9)

template <int C>
__global__ __launch_bounds__(256) void celeba_tail_bwd_persist_kernel(CelebaTailArgs a, int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sg0 = reinterpret_cast<float*>(smem);                // two images [11][68][3]
#ifdef DG_MEASURE
    wg_priority(a.prio);
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    float* tbuf = sg0 + 2 * CEB_IMG + wave * (32 * C);          // this wave's [32][C] store-transpose slice
    BwdWeights<C, 3, CE_GWP> bw;
    bw.load(a.F6, lane);
    // pin the fragments: without this the compiler re-issues the (invariant) filter loads inside the item loop
#pragma unroll
    for (int u = 0; u < C / 32; ++u)
#pragma unroll
        for (int k = 0; k < BwdWeights<C, 3, CE_GWP>::NS; ++k) asm volatile("" : "+v"(bw.w[u][k]));

    // element e = tid + 256*r of the image: row lr = e / 204, c = e % 204 -> da6[(i_lo + lr), c/3 - 1, c%3] = row base + c - 3
    auto fetch = [&](int item, float (&v)[CEB_PF]) {
        const int n = item >> 3, band = item & 7;
        const float* grow = a.g6 + (long long)n * 12288;
        const int i_lo = 8 * band - 1;
#pragma unroll
        for (int r = 0; r < CEB_PF; ++r) {
            const int e = tid + 256 * r;
            const int lr = e / CEB_ROWF, c = e - lr * CEB_ROWF;
            const int ii = i_lo + lr;
            const bool ok = lr < CEB_ROWS && ii >= 0 && ii < 64 && c >= 3 && c < 195;
            v[r] = ok ? grow[ii * 192 + c - 3] : 0.f;
        }
    };
    auto park = [&](float* sg, const float (&v)[CEB_PF]) {
#pragma unroll
        for (int r = 0; r < CEB_PF; ++r)
            if (tid + 256 * r < CEB_IMG) sg[tid + 256 * r] = v[r];
    };
    // Static strided item list.  Measured alternatives: an atomic item queue (one device-scope counter: 10k same-address
    // atomics cost more than the imbalance they remove, 132 -> 189 us), filter fragments in LDS at four workgroups per CU
    // (lifetimes spread 88..141 us, kernel 141 us), two bands per workgroup non-persistent (register pressure).
    float pf[CEB_PF];
    int item = blockIdx.x;
    if (item < n_items) { fetch(item, pf); park(sg0, pf); }
    __syncthreads();
    int buf = 0;
    // optional phase timing (wave 0): [0] fetch issue, [1] gather reads + MFMA issue, [2] store issue,
    // [3] wait for the prefetch + park, [4] absolute start (100 MHz), [5] items, [6] cycles, [7] 100 MHz ticks
    const bool tr = DG_TRACE_PTR(a) != nullptr && DG_DBG(a) != 8;          // dbg 8: the forward kernel owns the trace buffer
    long long ph[5] = {0, 0, 0, 0, 0}, t_begin = tr ? (long long)__builtin_readcyclecounter() : 0, nit = 0;
    const long long w_begin = tr ? (long long)wall_clock64() : 0;     // constant 100 MHz counter
    while (item < n_items) {
        long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (tr) t0 = (long long)__builtin_readcyclecounter();
        const int nxt = item + (int)gridDim.x;
        if (nxt < n_items && DG_DBG(a) != 7) fetch(nxt, pf);
        if (tr) t1 = (long long)__builtin_readcyclecounter();
        const float* sg = sg0 + buf * CEB_IMG;
        const int n = item >> 3, band = item & 7;
        float* hrow = a.h5 + (long long)n * (1024 * C);
        f32x16 acc[C / 32];
        tail_bwd_tile<C, 3, CE_GWP>(sg, ((2 * wave) * CE_GWP + 2 * frow) * 3, true, bw, acc, lane);
        if (tr) t2 = (long long)__builtin_readcyclecounter();
        const int oh = 4 * band + wave;
        bool do_store = true;
        if (DG_DBG(a) == 5 || DG_DBG(a) == 7) {                            // timing experiments: no stores (7: no fetch either)
            do_store = false;
#pragma unroll
            for (int u = 0; u < C / 32; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) do_store |= acc[u][e] == 12345.678f;
        }
        if (do_store) {
            // transpose the wave's [32 positions][C] result through its LDS slice: a lane then owns 4 consecutive channels and
            // one store instruction writes 1 KB contiguous (8 b128 stores per item instead of 32 dword ones -- under load every
            // VMEM instruction costs the issuing wave 100-300 cycles)
#pragma unroll
            for (int u = 0; u < C / 32; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) tbuf[((e & 3) + 8 * (e >> 2) + 4 * fh) * C + u * 32 + frow] = acc[u][e];
            constexpr int PPI = 256 / C;                           // positions per store instruction (64 lanes x 4 floats)
            float* orow = hrow + (long long)oh * 32 * C;
#pragma unroll
            for (int p = 0; p < 32 / PPI; ++p) {
                const int off = (p * PPI) * C + lane * 4;
                *reinterpret_cast<f32x4*>(orow + off) = *reinterpret_cast<const f32x4*>(tbuf + off);
            }
        }
        if (tr) t3 = (long long)__builtin_readcyclecounter();
        if (nxt < n_items && DG_DBG(a) != 7) park(sg0 + (buf ^ 1) * CEB_IMG, pf);
        if (tr) t4 = (long long)__builtin_readcyclecounter();
        __syncthreads();
        if (tr) {
            ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;
            ++nit;
        }
        item = nxt;
        buf ^= 1;
    }
    if (tr && tid == 0 && blockIdx.x < 4096) {
        long long* o = DG_TRACE_PTR(a) + (long long)blockIdx.x * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = ph[q];
        o[4] = w_begin;
        o[5] = nit;
        o[6] = (long long)__builtin_readcyclecounter() - t_begin;
        o[7] = (long long)wall_clock64() - w_begin;
    }
}

void launch_celeba_tail_fwd_mfma(const CelebaTailArgs& a, hipStream_t s) {
    static PerDeviceOnce attr;
    const int main16 = 6 * 32 * a.C * 4 > CE16_UNITS * CE16_UNIT * 4 ? 6 * 32 * a.C * 4 : CE16_UNITS * CE16_UNIT * 4;
    const int lds16 = main16 + 32;
    if (attr.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd16_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 32 * 64 * 4 + 32);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd16_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 32 * 128 * 4 + 32);
#ifdef DG_MEASURE
        const int lds32 = (192 * CE_NKP + 8) * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds32);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_mfma_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds32);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd16_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 32 * 64 * 4 + 32);
#endif
    }
#ifdef DG_MEASURE
    if (!a.fwd16) {
        const int lds32 = (192 * CE_NKP + 8) * 4;
        if (a.C == 64) hipLaunchKernelGGL((celeba_tail_fwd_mfma_kernel<64>), dim3(a.n_rows * 8), dim3(384), lds32, s, a);
        else hipLaunchKernelGGL((celeba_tail_fwd_mfma_kernel<128>), dim3(a.n_rows * 8), dim3(384), lds32, s, a);
        return;
    }
    if (a.C == 64 && a.trace && a.dbg == 8) {
        hipLaunchKernelGGL((celeba_tail_fwd16_kernel<64, true>), dim3(a.n_rows * 8), dim3(256), lds16, s, a);
        return;
    }
#endif
    if (a.C == 64 && a.fwd_split > 0) {
        const int n_items = a.n_rows * 16;
        const int grid = n_items < a.fwd_split ? n_items : a.fwd_split;
        const int lds = 4 * 32 * 64 * 4 + CES_PBUF * 4 + 5 * (64 / 16) * 64 * 16;
        static PerDeviceOnce attr2;
        if (attr2.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_split_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((celeba_tail_fwd_split_kernel<64>), dim3(grid), dim3(512), lds, s, a, n_items);
        return;
    }
    if (a.C == 64) hipLaunchKernelGGL((celeba_tail_fwd16_kernel<64, false>), dim3(a.n_rows * 8), dim3(256), lds16, s, a);
    else hipLaunchKernelGGL((celeba_tail_fwd16_kernel<128, false>), dim3(a.n_rows * 8), dim3(256), lds16, s, a);
}

#ifdef DG_MEASURE
template <int NB>
static void launch_celeba_tail_bwd_nb(const CelebaTailArgs& a, hipStream_t s) {
    const int lds = (8 * NB + 3) * CE_GWP * 3 * 4;
    const unsigned grid = (unsigned)(a.n_rows * (8 / NB));
    if (a.C == 64) hipLaunchKernelGGL((celeba_tail_bwd_mfma_kernel<64, NB>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((celeba_tail_bwd_mfma_kernel<128, NB>), dim3(grid), dim3(256), lds, s, a);
}
#endif

void launch_celeba_tail_bwd_mfma(const CelebaTailArgs& a, hipStream_t s) {
#ifdef DG_MEASURE
    if (a.bwd_persist <= 0) {
        switch (a.bwd_bands) {
            case 1: launch_celeba_tail_bwd_nb<1>(a, s); break;
            case 4: launch_celeba_tail_bwd_nb<4>(a, s); break;
            default: launch_celeba_tail_bwd_nb<2>(a, s); break;
        }
        return;
    }
#endif
    const int n_items = a.n_rows * 8;
    const int want = a.bwd_persist > 0 ? a.bwd_persist : 512;
    const int grid = n_items < want ? n_items : want;
    const int lds = (2 * CEB_IMG + 4 * 32 * a.C) * 4;
    static PerDeviceOnce attr;
    if (attr.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_bwd_persist_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (2 * CEB_IMG + 4 * 32 * 128) * 4);
    if (a.C == 64) hipLaunchKernelGGL((celeba_tail_bwd_persist_kernel<64>), dim3(grid), dim3(256), lds, s, a, n_items);
    else hipLaunchKernelGGL((celeba_tail_bwd_persist_kernel<128>), dim3(grid), dim3(256), lds, s, a, n_items);
}

}  // namespace dg
