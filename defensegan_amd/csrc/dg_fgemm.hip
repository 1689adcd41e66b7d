// LDS-free gathered implicit GEMM for gfx950 (MI355X), round 6: operands in MFMA FRAGMENT ORDER, fetched straight into the
// registers v_mfma_f32_32x32x2_f32 reads.
//
// Covers the forward transposed convolutions of the Defense-GAN generators (tflib/ops/deconv2d.py:100-117, call sites
// models/dataset_models.py:36-71 / 127-165) whose INPUT activation lives in fragment order (dg_kernels.h "fragment order"):
// an activation row block = 32 consecutive latent rows; per 8 consecutive floats of the NHWC row one contiguous KB
// [k-half 2][row 32][4 floats] -- exactly the B operand of one k8-step (4 MFMAs) for 32 rows.  The filters are packed the same
// way (32 output channels x 8 k per KB).  Every operand fetch is ONE fully coalesced buffer_load_dwordx4 per wave:
// no LDS staging, no ds_read, no barrier in the K loop, waves independent of each other.  Measured on MI355X
// (tools/exp_fraggemm.hip, profiles/r06_exp_fraggemm.txt): 147-149 TFLOP/s of a 150-153 pure-MFMA ceiling in steady state,
// against 137-142 for the LDS-DMA-staged loop of dg_gemm.hip (whose 8-10 DMA pieces per 64 MFMAs each cost most of an MFMA slot).
//
// Roles: the filters are the FIRST MFMA operand (rows = output channels), the activations the second (columns = latent rows):
// an accumulator then holds, per lane, 4 CONSECUTIVE channels of ONE latent row in 4 consecutive registers -- the output is
// already in the fragment order of the next layer's input (1 KB contiguous stores, no transposition), or 16-byte pieces of an
// NHWC row for the layers the tails read.
//
// Work: the planner's tap classes as in dg_gemm.hip; the M axis of a class is cut into M BLOCKS (one position j of one row
// block nb, mblk = nb * s + j); a wave tile = TN M blocks x TW 32-channel blocks (4 x 2: 128 rows x 64 channels, 8 accumulators).
// One tile on ONE SIMD would be a dependent chain 2-4 x longer than dg_gemm.hip's 4-wave tiles, so the K axis of a tile is split
// over `ksplit` waves of the workgroup (1, 2 or 4 -- a CONSTANT of the tap class, dg_plan.h frag_ksplit: the result does not
// depend on the batch or the job list): wave q takes the q-th contiguous part of the flattened (tap, k8) sequence, the partial
// accumulators meet in LDS after the loop in a fixed tree ((q0 + q1) + (q2 + q3)), wave 0 runs the epilogue.
// Every output element = that fixed tree of k-ordered fp32 fma chains.
#include <type_traits>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20;
constexpr int FRING = 4;                  // k8-steps of operands in flight per wave

// PERSIST: the record is ONE wave tile of this wave's own list (fgemm_persist_kernel): no K split, no other wave involved.
template <int TW, int TN, int MODE, bool OUTFRAG, bool PERSIST>
__device__ __forceinline__ void frag_body(const FragArgs& g, const FragJob jb, char* smem, int trace_slot) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = PERSIST ? 1 : __builtin_amdgcn_readfirstlane(jb.ksplit);
    const int q = wave & (ks - 1);                                   // K part of this wave
    const int tile = PERSIST ? 0 : (ks == 4 ? 0 : (ks == 2 ? wave >> 1 : wave));     // wave tile inside the job
    int nvalid = jb.n_mblk - tile * TN;
    nvalid = nvalid > TN ? TN : nvalid;
    const bool active = nvalid > 0;
#ifdef DG_MEASURE
    long long tr[5] = {0, 0, 0, 0, 0};
    const bool tron = g.trace != nullptr;
    const int dbg = g.dbg;
#else
    long long tr[5] = {0, 0, 0, 0, 0};
    constexpr bool tron = false;
    constexpr int dbg = 0;
#endif
    if (tron) tr[0] = (long long)__builtin_readcyclecounter();
    if constexpr (PERSIST) __builtin_amdgcn_s_setprio(0);            // (the previous tile of this wave ended at priority 3)
    auto dump = [&]() {
        if (!tron || lane != 0) return;
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* t = DG_TRACE_PTR(g) + (long long)trace_slot * 8;
        t[0] = tr[0]; t[1] = tr[1]; t[2] = tr[2]; t[3] = tr[3]; t[4] = (long long)__builtin_readcyclecounter(); t[5] = hwid; t[6] = (xcc & 15) | (ks << 8) | (q << 16); t[7] = jb.n_taps * (g.kch >> 3) / ks;
    };
    if (ks == 1 && !active) return;                                  // (no barrier below for ks == 1)

    // ---- the tile's M blocks: (row block, position) -> offsets inside the input / output rows (floats), all scalar
    const int s_cnt = jb.s;
    int nb[TN], a_pos[TN], o_pos[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int mblk = jb.mblk0 + tile * TN + (j < nvalid ? j : (nvalid > 0 ? nvalid - 1 : 0));
        mblk = __builtin_amdgcn_readfirstlane(mblk);
        const int b = (int)__umulhi((unsigned)mblk << 1, jb.s_magic);
        const int jj = mblk - b * s_cnt;
        const int jh = (int)__umulhi((unsigned)jj << 1, jb.wc_magic);
        const int jw = jj - jh * jb.wc;
        nb[j] = b;
        a_pos[j] = jb.a_base + jh * jb.a_rs + jw * jb.a_cs;
        o_pos[j] = jb.o_base + jh * jb.o_rs + jw * jb.o_cs;
    }
    // descriptors: A based at the tile's first row block (64-bit), offsets relative to it stay small; filters from their start
    const float* a_base = g.A + (long long)nb[0] * 32 * g.a_rowstride;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_base), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wp), 0, 0x7ffffff0, 0x00020000);
    int abase[TN];                                                   // byte offset of the block's fragment row: (floats) * 128
#pragma unroll
    for (int j = 0; j < TN; ++j) abase[j] = ((nb[j] - nb[0]) * (int)g.a_rowstride + a_pos[j]) * 128;
    const int voff = lane * 16;
    const int wcb = g.kch * 128;                                     // bytes between two 32-channel blocks of a filter slab
    const int wbase = jb.cb0 * wcb;

    const int kc8 = g.kch >> 3;
    const int n_my = (jb.n_taps * kc8) >> (ks == 4 ? 2 : (ks == 2 ? 1 : 0));      // k8-steps of this wave (a multiple of FRING: planner)
    const int start = q * n_my;
    // The taps of a class are a (rows x columns) grid of filter taps, both offsets affine in (u, v) = (t / tap_nw, t % tap_nw)
    // (dg_plan.h frag_tap_grid): the walk through them is scalar arithmetic -- a table fetch inside the loop (s_load + lgkmcnt)
    // made hipcc serialise the operand ring (measured in the ISA: loads landed in spare registers behind vmcnt(0) and were copied)
    const int t0 = start >> g.kc8_log2;
    int ld_k8 = start & (kc8 - 1), ld_t = t0;
    const int u0 = (int)__umulhi((unsigned)t0 << 1, jb.tap_nw_magic);
    int tcol = t0 - u0 * jb.tap_nw;
    int cur_a = (jb.a0 + u0 * jb.a_u + tcol * jb.a_v) * 128;
    int cur_w = (jb.w0 + u0 * jb.w_u + tcol * jb.w_v) * 4 + wbase;
    const int a_cstep = jb.a_v * 128, a_rstep = (jb.a_u - (jb.tap_nw - 1) * jb.a_v) * 128;
    const int w_cstep = jb.w_v * 4, w_rstep = (jb.w_u - (jb.tap_nw - 1) * jb.w_v) * 4;

    f32x16 acc[TW][TN];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 fw[FRING][TW], fa[FRING][TN];
    auto load_step = [&](int d) {
        const int k8off = ld_k8 * 1024;
#pragma unroll
        for (int i = 0; i < TW; ++i)
            fw[d][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, cur_w + i * wcb + k8off, 0));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fa[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voff, abase[j] + cur_a + k8off, 0));
        // branch-free advance (a scalar branch per step would cut the loop body into blocks the schedule pins cannot span).
        // Steps past the wave's last one fetch the next K part's operands (or, past the last tap, that tap again): valid addresses,
        // nobody multiplies what they bring.
        // (0 / 1 integers and multiplies, not && / ?: -- hipcc turned the short-circuit form into four scalar branches per iteration)
        const int wrap = ld_k8 + 1 == kc8 ? 1 : 0;
        const int adv = wrap & (ld_t + 1 < jb.n_taps ? 1 : 0);        // (past the last tap the walk re-reads it: addresses stay valid)
        const int roww = adv & (tcol + 1 == jb.tap_nw ? 1 : 0);
        ld_k8 = (ld_k8 + 1) * (1 - wrap);
        ld_t += adv;
        cur_a += adv * a_cstep + roww * (a_rstep - a_cstep);
        cur_w += adv * w_cstep + roww * (w_rstep - w_cstep);
        tcol = (tcol + adv) * (1 - roww);
    };
    if (tron) tr[1] = (long long)__builtin_readcyclecounter();
    if (active) {
#pragma unroll
        for (int d = 0; d < FRING - 1; ++d) { load_step(d); __builtin_amdgcn_sched_barrier(0); }   // ring order (the loop's waits merge with this path)
        for (int s0 = 0; s0 < n_my; s0 += FRING) {
#pragma unroll
            for (int d = 0; d < FRING; ++d) {
                // refill the stage consumed one step ago with step s0 + d + FRING - 1 (steps past the end re-read valid data
                // that nobody multiplies), then the MFMAs of stage d: lane = channel / row, e = k inside the half
                load_step((d + FRING - 1) % FRING);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TW; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[d][i][e], fa[d][j][e], acc[i][j], 0, 0, 0);
                constexpr int NL = TW + TN, NM = 4 * TW * TN;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM / NL, 0);
                }
                if (NM % NL) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM % NL, 0);
            }
        }
    }

    if (tron) tr[2] = (long long)__builtin_readcyclecounter();
    // the reduction and the epilogue are VALU / LDS / store work: beside a co-resident wave's MFMA stream each of their instructions
    // waited for a gap between two MFMAs (profiles/r06_frag_path_ab.txt: 8-13 us per tile); from here on this wave goes first
    __builtin_amdgcn_s_setprio(3);
    // ---- K-split reduction through LDS: image = [TW][TN][4 quads][64 lanes][4 floats] (b128 per lane: conflict-free)
    constexpr int IMG_BYTES = TW * TN * 16 * 256;
    auto put = [&](char* img) {
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {acc[i][j][q4 * 4 + 0], acc[i][j][q4 * 4 + 1], acc[i][j][q4 * 4 + 2], acc[i][j][q4 * 4 + 3]};
                    *reinterpret_cast<f32x4*>(img + (((i * TN + j) * 4 + q4) * 64 + lane) * 16) = v;
                }
    };
    auto add = [&](const char* img) {
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(img + (((i * TN + j) * 4 + q4) * 64 + lane) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][q4 * 4 + e] += v[e];
                }
    };
    if (dbg & 2) { if (q != 0) { dump(); return; } } else
    if (ks > 1) {
        // round 1: the odd parts hand their accumulators to the even ones (ks = 2: one image per tile; ks = 4: q1 -> q0, q3 -> q2)
        char* const img1 = smem + (ks == 4 ? (q >> 1) : tile) * IMG_BYTES;
        if ((q & 1) && active) put(img1);
        __syncthreads();
        if (q & 1) { dump(); return; }
        if (active) add(img1);
        if (ks == 4) {                             // round 2: (q0 + q1) + (q2 + q3)
            __syncthreads();
            if (q == 2) put(smem);
            __syncthreads();
            if (q == 2) { dump(); return; }
            add(smem);
        }
    }
    if (tron) tr[3] = (long long)__builtin_readcyclecounter();
    if (!active) { dump(); return; }
    if (dbg & 1) { dump(); return; }              // TIMING EXPERIMENT ONLY: no epilogue

    // ---- epilogue (one wave per tile).  Accumulator (i, j): lane = (k-half fh, latent row r of the block), register 4 * q4 + e =
    // channel 32 * (cb0 + i) + 8 * q4 + 4 * fh + e: f32x4 pieces of 4 consecutive channels of one row.
    const int fh = lane >> 5, r = lane & 31;
    f32x4 bv[TW][4];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            bv[i][q4] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_RELU) bv[i][q4] = *reinterpret_cast<const f32x4*>(g.bias + (jb.cb0 + i) * 32 + 8 * q4 + 4 * fh);
        }
    // NHWC output: the tile of one M block (32 rows x 64 channels) is transposed through this wave's own LDS slice (row pitch
    // 272 B: the 16 lanes of a b128 write pass hit 16 different bank quads) so that 16 lanes store the 256 contiguous bytes of one
    // row -- full 128-B lines; 16-B pieces scattered over 32 rows cost the single epilogue wave 13 us per tile (profiles/r06_frag_path_ab.txt)
    constexpr int TPITCH = 32 * TW * 4 + 16;
    char* const tslice = smem + wave * (32 * TPITCH);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (j >= nvalid) break;
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const int ch0 = (jb.cb0 + i) * 32;
            unsigned bits = 0u;
            f32x4 v[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][q4 * 4 + e] + bv[i][q4][e];
                    if constexpr (MODE == EPI_BIAS_RELU) {
                        bits |= (t > 0.f ? 1u : 0u) << (8 * q4 + 4 * fh + e);
                        t = t > 0.f ? t : 0.f;
                    }
                    v[q4][e] = t;
                }
            }
            if constexpr (OUTFRAG) {
                float* ob = g.Out + ((long long)nb[j] * 32 * g.out_rowstride);
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    // (offset in the VGPR, immediate soffset: see dg_linear.hip frag_store -- the 16-byte-store data hazard)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[q4]), ro, voff + (o_pos[j] + ch0 + 8 * q4) * 128, 0, 0);
            } else {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(tslice + r * TPITCH + (i * 32 + 8 * q4 + 4 * fh) * 4) = v[q4];
            }
            if constexpr (MODE == EPI_BIAS_RELU) {
                if (g.gate_bits) {                   // ReluGrad gates of this layer's output, one bit per element (uniform branch)
                    bits |= (unsigned)__shfl_xor((int)bits, 32, 64);
                    if (fh == 0) g.gate_bits[((long long)nb[j] * 32 + r) * g.gate_words + ((o_pos[j] + ch0) >> 5)] = bits;
                }
            }
        }
        if constexpr (!OUTFRAG) {
            // 16 lanes per row (TW * 8 = 16 pieces of 16 B), 4 rows per instruction
            constexpr int LPR = TW * 8;
            const int rl = lane / LPR, cl = lane % LPR;
            float* obase = g.Out + (long long)nb[j] * 32 * g.out_rowstride + o_pos[j] + jb.cb0 * 32 + cl * 4;
#pragma unroll
            for (int k = 0; k < 32 / (64 / LPR); ++k) {
                const int row = k * (64 / LPR) + rl;
                const f32x4 t = *reinterpret_cast<const f32x4*>(tslice + row * TPITCH + cl * 16);
                *reinterpret_cast<f32x4*>(obase + (long long)row * g.out_rowstride) = t;
            }
        }
    }
    dump();
}

template <int TW, int TN, int MODE, bool OUTFRAG>
__global__ __launch_bounds__(256, 2) void fgemm_kernel(FragArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two accumulator images (K-split reduction) / epilogue slices
    frag_body<TW, TN, MODE, OUTFRAG, false>(g, g.jobs[blockIdx.x], smem, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
}

// Persistent form (engine option frag_path = 2): 2 workgroups per CU, every WAVE walks its own list of whole wave tiles
// (dg_plan.h build_frag_tiles: longest-first bin packing of the layer's tiles over the wave slots, no tile split along K) -- no
// dispatch granularity, no reduction, no barrier anywhere; a wave's turnover between two tiles is covered by the wave it shares
// its SIMD with.
template <int TW, int TN, int MODE, bool OUTFRAG>
__global__ __launch_bounds__(256, 2) void fgemm_persist_kernel(FragArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the waves' epilogue slices
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    const int t0 = g.wave_begin[gw], t1 = g.wave_begin[gw + 1];
    for (int t = t0; t < t1; ++t) frag_body<TW, TN, MODE, OUTFRAG, true>(g, g.jobs[t], smem, t);
}

template <int MODE, bool OUTFRAG>
void launch_mo(const FragArgs& a, hipStream_t s) {
    constexpr int TW = 2, TN = 4;
    const int lds = 2 * TW * TN * 16 * 256;
    if (a.wave_begin) {
        static PerDeviceOnce attrp;
        if (attrp.need())
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fgemm_persist_kernel<TW, TN, MODE, OUTFRAG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((fgemm_persist_kernel<TW, TN, MODE, OUTFRAG>), dim3((unsigned)a.n_wgs), dim3(256), lds, s, a);
        return;
    }
    static PerDeviceOnce attr;
    if (attr.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fgemm_kernel<TW, TN, MODE, OUTFRAG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((fgemm_kernel<TW, TN, MODE, OUTFRAG>), dim3((unsigned)a.n_jobs), dim3(256), lds, s, a);
}

template <int MODE>
void launch_m(const FragArgs& a, hipStream_t s) {
    if (a.out_frag) launch_mo<MODE, true>(a, s); else launch_mo<MODE, false>(a, s);
}

}  // namespace

void launch_fgemm(const FragArgs& a0, hipStream_t s) {
    if (a0.n_jobs <= 0) return;
    FragArgs a = a0;
#ifdef DG_MEASURE
    // Measurement build only (profiles/r06_frag_*): DG_FRAG_DBG = 1 no epilogue / 3 no K-split reduction either (wrong results, timing
    // only); DG_FRAG_TRACE_KCH = <K extent per tap> dumps the per-wave cycle stamps of that layer's 20th launch to DG_FRAG_TRACE_FILE
    static const int dbg = getenv("DG_FRAG_DBG") ? atoi(getenv("DG_FRAG_DBG")) : 0;
    a.dbg = dbg;
    a.trace = nullptr;
    static const int trace_kch = getenv("DG_FRAG_TRACE_KCH") ? atoi(getenv("DG_FRAG_TRACE_KCH")) : 0;
    static int seen = 0;
    static long long* d_trace = nullptr;
    bool dumping = false;
    if (trace_kch && a.kch == trace_kch && ++seen == 20) {
        (void)hipMalloc(&d_trace, (size_t)a.n_jobs * 4 * 8 * sizeof(long long));
        (void)hipMemset(d_trace, 0, (size_t)a.n_jobs * 4 * 8 * sizeof(long long));
        (void)hipStreamSynchronize(s);
        a.trace = d_trace;
        dumping = true;
    }
#endif
    switch (a.mode) {
        case EPI_BIAS_RELU: launch_m<EPI_BIAS_RELU>(a, s); break;
        case EPI_BIAS: launch_m<EPI_BIAS>(a, s); break;
        default: launch_m<EPI_STORE>(a, s); break;
    }
#ifdef DG_MEASURE
    if (dumping) {
        (void)hipStreamSynchronize(s);
        std::vector<long long> hbuf((size_t)a.n_jobs * 4 * 8);
        (void)hipMemcpy(hbuf.data(), d_trace, hbuf.size() * sizeof(long long), hipMemcpyDeviceToHost);
        FILE* f = fopen(getenv("DG_FRAG_TRACE_FILE") ? getenv("DG_FRAG_TRACE_FILE") : "/tmp/frag_trace.bin", "wb");
        if (f) { fwrite(hbuf.data(), sizeof(long long), hbuf.size(), f); fclose(f); }
    }
#endif
}

}  // namespace dg
