// The "latent turn" of a projection step on gfx950 (MI355X): Linear backward (split-K partials of dz), then -- after the
// momentum update (dg_small.hip) -- Linear forward + BiasAdd + ReLU.  Reference call sites: tflib/ops/linear.py:129-142 inside
// the loop body of models/gan.py:409-437.
//
// Both are [rows x K] . [K x 128-column tile] products with a SHORT K (128 forward, one 256-wide slice backward), which
// the position-batched kernel (dg_gemm.hip) runs as thousands of 4- / 8-chunk jobs whose start-up and write-out are a third
// of each job.  Here the weights are STATIONARY: a workgroup keeps the fragments of its 128 output columns (forward: a
// column tile of W^T; backward: one K slice of W for all 128 latent columns) in REGISTERS for its whole life -- 64 / 128
// VGPRs per lane, loaded from a fragment-order copy of the weights in fully coalesced runs -- and streams 32-row blocks of the other operand through a double-buffered LDS image (full 128-B lines by
// buffer_load ... lds, XOR-swizzled on the source side exactly as in dg_gemm.hip).  One barrier, one 16 / 32 KB block and
// 64 / 128 MFMAs per wave and block; the next block's DMA is in flight for a whole block; the B operand is never staged
// (LDS-DMA pieces per MFMA: 0.0625 against 0.19-0.25 for the small job shapes).
//
// Arithmetic: every output element is the same k-ordered fp32 fma chain as in dg_gemm.hip (chunks of 32 in ascending order,
// inside a chunk the MFMA k-pairs (8kk + e, 8kk + 4 + e), e = 0..3, kk = 0..3) and the same epilogue expressions, so the
// results are BIT-IDENTICAL to the generic kernel's (tests/test_gpu_variants.py) and, like them, independent of the batch a
// row is in.  The backward keeps the engine's fixed K slices: slice sums are still added in slice order by the update.
// Optional (engine option update_fold, off: measured slower, profiles/r04_ab_update_fold.txt): the update itself inside the backward
// launch, done by the workgroup that delivers a row block's last slice ("folded update" below).
#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

__device__ __forceinline__ int lswz(int row) { return (row >> 1) & 7; }

// KCH: 32-float K chunks per block row (forward: latent / 32; backward: slice width / 32)
// FOLD (backward only): the momentum update of a 32-row block rides in the workgroup that delivers the block's last K slice
// ("folded update" below).
// FRAG (forward only): the output leaves in FRAGMENT ORDER (dg_types.h) for dg_fgemm.hip, with the ReluGrad gates as one bit per
// element.  The MFMA operand roles are swapped for it (weights first): an accumulator then holds 4 consecutive columns of one row per
// lane in 4 consecutive registers, which IS the fragment order -- 1 KB contiguous stores, no transposition through LDS.  Same
// products, same k order per element.
template <int KCH, int MODE, bool FOLD = false, bool FRAG = false>
__global__ __launch_bounds__(256, KCH <= 4 ? 3 : (KCH <= 6 ? 2 : 1)) void lin_stationary_kernel(LinArgs g) {
    static_assert(!FOLD || MODE == EPI_STORE, "the folded update belongs to the split-K backward");
    static_assert(!FRAG || MODE == EPI_BIAS_RELU, "fragment-order output: the forward with ReLU");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BLK_BYTES = KCH * 4096;             // one 32-row block: [KCH chunks][32 rows][128 B]
    char* const epi = smem + 2 * BLK_BYTES;           // [4 waves][32 x 32 floats] transposition tiles
    unsigned* const sflag = reinterpret_cast<unsigned*>(epi + 4 * 4096);      // FOLD: "this workgroup was the last arriver", 2 slots
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;

    const int unit = (int)blockIdx.x % g.units;       // column tile (forward) / K slice (backward)
    const int grp = (int)blockIdx.x / g.units;        // row group: blocks grp, grp + groups, ...
    const int n_blocks = (g.n_rows + 31) >> 5;
    const int n_my = grp < n_blocks ? (n_blocks - grp + g.groups - 1) / g.groups : 0;
    if (n_my == 0) return;
#ifdef DG_MEASURE
    long long tr[4] = {0, 0, 0, 0};
    const bool tron = g.trace != nullptr && tid == 0;
    if (tron) tr[0] = (long long)__builtin_readcyclecounter();
#endif

    // ---- streamed operand: rows of A, one 32-float chunk of a row = one 128-B line, 8 rows per DMA instruction; wave w
    // stages rows 8w .. 8w+7 of every chunk.  LDS slot s of a row receives source chunk s ^ swz(row).
    const float* const a_unit = g.A + (long long)unit * g.a_unit;
    const int srow = wave * 8 + (lane >> 3);
    const unsigned sslot = (unsigned)(((lane & 7) ^ lswz(srow)) << 4);
    auto stage = [&](int blk, char* dst) {
        const int row0 = blk << 5;
        int r = row0 + srow;
        r = r < g.n_rows ? r : g.n_rows - 1;                       // ragged last block: clamp loads, mask stores
        // descriptor based at the block's first row (64-bit), per-lane offset inside the block, chunk offset in an SGPR
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a_unit + (long long)row0 * g.a_rowstride), 0, 0x7ffffff0, 0x00020000);
        const unsigned voff = (unsigned)(r - row0) * (unsigned)(g.a_rowstride * 4) + sslot;
#pragma unroll
        for (int c = 0; c < KCH; ++c)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DG_LDS_PTR(dst + c * 4096 + wave * 1024), 16, voff, c * 128, 0, 0);
    };
    stage(grp, smem);
    if (n_my > 1) stage(grp + g.groups, smem + BLK_BYTES);       // both buffers are free: the second block rides with the weights

    // ---- stationary operand: this wave's 32 output columns x K, as MFMA B fragments (lane = column + 32 * k-half)
    f32x4 wf[KCH][4];
    {
        const float* wp = g.Wp + lin_pack_index(unit, wave, KCH, 0, 0, lane, 0);
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wf[c][kk] = *reinterpret_cast<const f32x4*>(wp + (c * 4 + kk) * 256);
    }
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const int ocol = unit * g.out_unit + wave * 32 + ec;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_RELU) bv = *reinterpret_cast<const f32x4*>(g.bias + ocol);
    // the bias is the load issued last: once it is there everything before it is (loads retire in order) -- told to the
    // compiler here so that it does not wait for "outstanding" loads again inside the loop
    asm volatile("" : "+v"(bv));
    float* const tb = reinterpret_cast<float*>(epi + wave * 4096);
    const int a_rd = frow * 128;
    const int a_sw = lswz(frow);

    // block 0 and the weights are there; every wave's pieces of it too
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (FOLD) {
        // (the weights ARE in their registers now; hipcc does not see the asm wait and, with this variant's single loop body,
        // would otherwise wait for "outstanding" weight loads -- in fact for the next block's DMA -- after the first chunk of every block)
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(wf[c][kk]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- write-out of a finished 32 x 32 tile, in three pieces that ride INSIDE the next block's MFMA stream (a wave that wrote
    // its tile out between two blocks left the matrix pipe idle for a fifth of a block, and two workgroups sharing a CU ran
    // that phase in lock step): transpose through this wave's LDS slice so that a lane owns 4 consecutive columns of one row,
    // then b128 stores; same expressions as dg_gemm.hip
    auto out_write = [&](const f32x16& acc) {
#pragma unroll
        for (int e = 0; e < 16; ++e) tb[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + frow] = acc[e];
    };
    auto out_read = [&](f32x4 (&v)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
    };
    auto out_store = [&](f32x4 (&v)[4], int blk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the next chunk's fragment reads were issued before these)
        const int row0 = blk << 5;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            asm volatile("" : "+v"(v[p]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = v[p][q] + bv[q];
                if constexpr (MODE == EPI_BIAS_RELU) t = t > 0.f ? t : 0.f;
                v[p][q] = t;
            }
            const int r = row0 + p * 8 + er;
            if constexpr (FOLD) {
                // write-through (sc1): another workgroup, on any XCD, sums these slices inside this launch
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                    g.Out + (long long)row0 * g.out_rowstride, 0, 0x7ffffff0, 0x00020000);
                if (r < g.n_rows)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[p]), orsrc,
                                                           (int)((unsigned)(p * 8 + er) * (unsigned)(g.out_rowstride * 4) + (unsigned)ocol * 4u), 0, 16);
            } else {
                if (r < g.n_rows) *reinterpret_cast<f32x4*>(g.Out + (long long)r * g.out_rowstride + ocol) = v[p];
            }
        }
    };

    // FRAG: bias of the 4 column quads this lane holds, and the write-out of a finished tile (bias, ReLU, gates, 4 x 1 KB stores)
    f32x4 bvf[4];
    if constexpr (FRAG) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            bvf[q4] = *reinterpret_cast<const f32x4*>(g.bias + unit * g.out_unit + wave * 32 + 8 * q4 + 4 * fh);
            asm volatile("" : "+v"(bvf[q4]));
        }
    }
    auto frag_store = [&](const f32x16& acc, int blk) {
        const int col0 = unit * g.out_unit + wave * 32;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            g.out_frag + (long long)blk * 32 * g.out_rowstride, 0, 0x7ffffff0, 0x00020000);
        unsigned bits = 0u;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[q4 * 4 + e] + bvf[q4][e];
                bits |= (t > 0.f ? 1u : 0u) << (8 * q4 + 4 * fh + e);
                v[e] = t > 0.f ? t : 0.f;
            }
            // (offset in the VGPR, soffset an immediate: with an SGPR soffset hipcc inserts no wait state between a 16-byte store and
            // the next VALU write of its data registers -- on gfx950 the next quad's bias add then overwrote element 0 of lanes
            // 12-15 of every 16 before the store had read them, whenever a second workgroup delayed the store's issue)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, lane * 16 + (col0 + 8 * q4) * 128, 0, 0);
        }
        // (always: the engine passes the gate buffer whenever it asks for fragment order; the wait below counts this store)
        bits |= (unsigned)__shfl_xor((int)bits, 32, 64);
        if (fh == 0) g.gate_bits[((long long)blk * 32 + frow) * g.gate_words + (col0 >> 5)] = bits;
    };

    // ---- folded update (FOLD).  A 32-row block of dz is complete when all `units` K-slice workgroups of its row group have
    // stored their partial tiles.  Every workgroup draws a ticket from the block's counter once ALL its waves' stores of that
    // block have been acknowledged (write-through stores, s_waitcnt vmcnt(0) in every wave, barrier, one relaxed agent-scope
    // fetch_add: the hand-off recipe of the programming guide, section 6 guideline 16 / split-K reduction); the workgroup that
    // draws units - 1 reads all slices back with sc1 loads (L1 bypassed), adds them in SLICE ORDER from zero exactly as
    // momentum_update_kernel does and applies m <- momentum * m + g, z <- z - lr * m (the same two fmas).
    // The ticket is drawn one block late and read another block later, so its round trip never stalls the MFMA stream; only a
    // workgroup's last two blocks are settled after its last multiply.  No workgroup ever waits for another one.
    unsigned tk = 0;
    auto arrive = [&](int blk) {
        // (atomic inc with wrap at units - 1: the counter is back at zero after the last arrival.  Not fetch_add: hipcc rewrites
        // an add in a divergent branch into a wave reduction that waits for the returned value on the spot)
        if (tid == 0) tk = __builtin_amdgcn_atomic_inc32(g.upd_count + blk, (unsigned)(g.units - 1), __ATOMIC_RELAXED, "agent");
    };
    auto post_flag = [&](int slot) {                  // before a barrier: was the ticket drawn last time the block's last one?
        if (tid == 0) sflag[slot & 1] = tk == (unsigned)(g.units - 1) ? 1u : 0u;
    };
    auto update_block = [&](int blk) {
        const int row0 = blk << 5;
        const int ur = tid >> 3, uq = (tid & 7) * 4;   // 8 lanes per row: one 128-B line per (slice, 32-column group)
        const int r = row0 + ur;
        if (r >= g.n_rows) return;
        const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
            g.Out + (long long)row0 * g.out_rowstride, 0, 0x7ffffff0, 0x00020000);
        const unsigned rbase = (unsigned)ur * (unsigned)(g.out_rowstride * 4);
#pragma unroll 1
        for (int j = 0; j < 4; j += 2) {                     // two 32-column groups at a time: 16 loads in flight per lane
            const int col = uq + 32 * j;
            f32x4 sum0 = {0.f, 0.f, 0.f, 0.f}, sum1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int s0 = 0; s0 < g.units; s0 += 8) {        // units % 8 == 0 (launch_lin_stationary)
                f32x4 pv[2][8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const unsigned off = rbase + (unsigned)((s0 + s) * g.out_unit + col) * 4u;
                    pv[0][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)off, 0, 16));
                    pv[1][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)off, 128, 16));
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) { sum0 += pv[0][s]; sum1 += pv[1][s]; }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x4 sum = jj ? sum1 : sum0;
                float* mp = g.upd_m + (long long)r * 128 + col + 32 * jj;
                float* zp = g.upd_z + (long long)r * 128 + col + 32 * jj;
                f32x4 mv = *reinterpret_cast<const f32x4*>(mp);
                f32x4 zv = *reinterpret_cast<const f32x4*>(zp);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mv[q] = __builtin_fmaf(g.upd_momentum, mv[q], sum[q]);
                    zv[q] = __builtin_fmaf(-g.upd_lr, mv[q], zv[q]);
                }
                *reinterpret_cast<f32x4*>(mp) = mv;
                *reinterpret_cast<f32x4*>(zp) = zv;
            }
        }
    };

    f32x16 done;                       // the previous block's tile, written out while this block multiplies
    for (int i = 0; i < n_my; ++i) {
        const int blk = grp + i * g.groups;
#ifdef DG_MEASURE
        if (tron && i == 0) tr[1] = (long long)__builtin_readcyclecounter();
#endif
        if (i > 0 && i + 1 < n_my) stage(blk + g.groups, smem + ((i + 1) & 1) * BLK_BYTES);     // in flight for the whole block
        const char* st = smem + (i & 1) * BLK_BYTES + a_rd;

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        f32x4 a[2][4];
        f32x4 v[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[0][kk] = *reinterpret_cast<const f32x4*>(st + (((kk * 2 + fh) ^ a_sw) << 4));
        if constexpr (!FRAG) { if (i > 0) out_write(done); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            if (c + 1 < KCH) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    a[(c + 1) & 1][kk] = *reinterpret_cast<const f32x4*>(st + (c + 1) * 4096 + (((kk * 2 + fh) ^ a_sw) << 4));
            }
            if constexpr (!FRAG) { if (c == 0 && i > 0) out_read(v); }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (FRAG) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[c][kk][e], a[c & 1][kk][e], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][kk][e], wf[c][kk][e], acc, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FRAG) { if (c == (KCH > 2 ? 1 : 0) && i > 0) frag_store(done, blk - g.groups); }
            else { if (c == (KCH > 2 ? 1 : 0) && i > 0) out_store(v, blk - g.groups); }
        }
        done = acc;
#ifdef DG_MEASURE
        if (tron && i == 0) tr[2] = (long long)__builtin_readcyclecounter();
        if (tron && i == 1) tr[3] = (long long)__builtin_readcyclecounter();
#endif
        if constexpr (FOLD) {
            // block i-1's stores (issued seven chunks ago) and block i-2's ticket are back; after the barrier that holds for
            // every wave: draw block i-1's ticket, settle block i-2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (i >= 2) post_flag(i);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (i >= 1) arrive(blk - g.groups);
            if (i >= 2 && __builtin_amdgcn_readfirstlane((int)sflag[i & 1])) update_block(blk - 2 * g.groups);
        } else if (i + 1 < n_my) {
            // this wave's pieces of block i+1 have landed once at most the 4 row stores of block i-1 (issued after them, inside
            // this block's stream; VMEM operations retire in order) are outstanding; then the barrier: every wave's pieces are
            // there, and every wave is done reading the buffer the block after that will be staged into
            if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (FRAG) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // (4 x 1 KB stores + the gate words)
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const int last_blk = grp + (n_my - 1) * g.groups;
    if constexpr (FRAG) {
        frag_store(done, last_blk);
    } else {
        f32x4 v[4];
        out_write(done);
        out_read(v);
        out_store(v, last_blk);
    }
    if constexpr (FOLD) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the last block's stores, the ticket of the one before
        if (n_my >= 2) post_flag(n_my);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        arrive(last_blk);
        if (n_my >= 2 && __builtin_amdgcn_readfirstlane((int)sflag[n_my & 1])) update_block(last_blk - g.groups);
        post_flag(n_my + 1);                                             // (waits for the ticket just drawn)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (__builtin_amdgcn_readfirstlane((int)sflag[(n_my + 1) & 1])) update_block(last_blk);
    }
#ifdef DG_MEASURE
    if (tron) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* t = g.trace + (long long)blockIdx.x * 8;
        t[0] = tr[0]; t[1] = tr[1]; t[2] = tr[2]; t[3] = tr[3]; t[4] = (long long)__builtin_readcyclecounter(); t[5] = hwid; t[6] = n_my; t[7] = xcc & 15;
    }
#endif
}

template <int KCH, int MODE, bool FOLD = false, bool FRAG = false>
void launch_km(const LinArgs& a, hipStream_t s) {
    const int lds = 2 * KCH * 4096 + 4 * 4096 + (FOLD ? 16 : 0);
    static PerDeviceOnce attr;
    if (attr.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lin_stationary_kernel<KCH, MODE, FOLD, FRAG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((lin_stationary_kernel<KCH, MODE, FOLD, FRAG>), dim3((unsigned)(a.units * a.groups)), dim3(256), lds, s, a);
}

template <int KCH>
void launch_k(const LinArgs& a, hipStream_t s) {
    if constexpr (KCH == 8) {
        if (a.upd_z) launch_km<KCH, EPI_STORE, true>(a, s);  // + the folded update (lin_fold_supported: the caller's check)
        else launch_km<KCH, EPI_STORE>(a, s);                // backward: split-K partials
    } else {
        if (a.mode == EPI_BIAS) launch_km<KCH, EPI_BIAS>(a, s);      // forward with Batchnorm behind it
        else if (a.out_frag) launch_km<KCH, EPI_BIAS_RELU, false, true>(a, s);
        else launch_km<KCH, EPI_BIAS_RELU>(a, s);
    }
}

}  // namespace

bool lin_stationary_supported(int kch, int mode) {
    return kch == 8 ? mode == EPI_STORE : ((kch == 2 || kch == 4 || kch == 6) && (mode == EPI_BIAS || mode == EPI_BIAS_RELU));
}

// the folded update reads the slices eight at a time and addresses z / m as [rows][128]
bool lin_fold_supported(int units, int latent) { return units >= 8 && units % 8 == 0 && latent == 128; }

void launch_lin_stationary(const LinArgs& a, hipStream_t s) {
    if (a.n_rows <= 0 || a.units <= 0 || a.groups <= 0) return;
    switch (a.kch) {
        case 2: launch_k<2>(a, s); break;
        case 4: launch_k<4>(a, s); break;
        case 6: launch_k<6>(a, s); break;
        case 8: launch_k<8>(a, s); break;
        default: break;
    }
}

}  // namespace dg
