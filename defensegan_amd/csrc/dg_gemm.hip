// Position-batched gathered implicit GEMM for gfx950 (MI355X): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Every output element is one k-ordered fp32 fma chain over (tap, channel), taps in the planner's order (bit-identical to
// the round-1 per-position kernel, profiles/r02_gemm2_bit_identity_vs_r01_kernel.txt).  Covers Linear fwd/bwd and every 5x5 stride-2 transposed conv fwd /
// backward-to-input of the Defense-GAN generators (reference call sites: tflib/ops/linear.py:129-142,
// tflib/ops/deconv2d.py:100-117).
//
// Structure:
//  * M axis = (latent row, output position) pairs of one tap CLASS (dg_plan.cpp): positions with the same relative tap
//    pattern share the filter slabs, so an M tile is dense for ANY batch size and every tile of a class has the same K.
//  * One workgroup = one JOB from a host-built list ordered longest first; the hardware dispatcher is the (dynamic) queue.
//    Jobs late in the list are the same tiles cut in halves / quarters along M and N -- never along K, except for the fixed
//    two-half split of the K-pair classes (dg_plan.h class_is_paired: a constant of the layer plan) -- so every output element
//    keeps its summation tree whatever the batch and the list look like: big tiles for the MFMA rate, small ones to level the
//    end of the launch.
//  * 128x128 tiles (2x2 waves; 256x64 with 4x1 waves for the 64-column layers; 64x64 per wave, four independent accumulators): 8 operand lines staged per 64 MFMAs
//    instead of 8 per 32; operand DMA through buffer_load ... lds with the chunk offset in an SGPR (no address VALU).
//  * The fragment reads of k-step kk+1 are issued before the MFMAs of k-step kk and pinned there.
#include <type_traits>

#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

int gemm_lds_bytes(int family, int min_level);

namespace {

constexpr int BK = 32;                 // floats per K chunk = one 128-B line per row
constexpr int ROW_BYTES = BK * 4;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// LLVM SchedGroupMask bits
constexpr int SG_MFMA = 0x8, SG_VMEM = 0x10, SG_DSREAD = 0x100;

// (FAM, TAG only make every call site its own specialization: hipcc's host pass rejects a second reference to one)
template <int TM, int TN, int WM, int WN, int MODE, int FAM, int TAG, bool PAIR>
__device__ __forceinline__ void run_job(const GemmArgs& g, const JobDesc jb, char* smem) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;   // waves are WM x WN, each owns TM x TN 32x32 accumulator tiles
    constexpr int SA = BM / 32, SB = BN / 32;      // staging slots (one 1 KB wave-instruction each) per wave
    constexpr int NS = SA + SB;
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef DG_MEASURE
    long long tr0 = 0;
    if (g.trace) tr0 = wall_clock64();
#endif

    const int s_cnt = jb.pos_count;
    const unsigned magic = jb.magic;
    const int m_valid = jb.m_valid;
    // M row r of the job -> (latent row n_first + q, position j)
    auto split = [&](int r, int& q, int& j) {
        const unsigned jj = (unsigned)(jb.j_first + r);
        q = (int)__umulhi(jj << 1, magic);             // jj / s, magic = ceil(2^31 / s): exact for jj < 2^20, branch-free for s = 1
        j = (int)jj - q * s_cnt;
    };
    // position j of the class -> float offsets inside an A row / an output row.  A class with taps is a rectangular grid
    // (dg_types.h ClassDesc): computed from the job record -- no table fetch between the job record and the first operand DMA
    const bool grid = jb.wc != 0;
    auto pos_of_a = [&](int j) -> int {
        if (!grid) return g.pos_a[jb.pos_begin + j];
        const int jh = (int)__umulhi((unsigned)j << 1, jb.wc_magic);
        return jb.a_base + jh * jb.a_rs + (j - jh * jb.wc) * jb.a_cs;
    };
    auto pos_of_out = [&](int j) -> int {
        if (!grid) return g.pos_out[jb.pos_begin + j];
        const int jh = (int)__umulhi((unsigned)j << 1, jb.wc_magic);
        return jb.o_base + jh * jb.o_rs + (j - jh * jb.wc) * jb.o_cs;
    };

    // ---- operand descriptors.  A: base = first latent row of the job; per-lane byte offset of its staging rows.
    const float* a_base = g.A + (long long)jb.n_first * g.a_rowstride;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_base), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, 0x7ffffff0, 0x00020000);
    unsigned voff_a[SA], voff_w[SB];
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        const int r = (s * 4 + wave) * 8 + (lane >> 3);           // row inside the A tile
        const int c = (lane & 7) ^ swz(r);                         // source 16-B chunk for LDS slot lane&7
        const int rc = r < m_valid ? r : m_valid - 1;              // ragged M: clamp loads, mask stores
        int q, j;
        split(rc, q, j);
        voff_a[s] = (unsigned)(q * (int)g.a_rowstride + pos_of_a(j)) * 4u + (unsigned)c * 16u;
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int r = (s * 4 + wave) * 8 + (lane >> 3);           // output column inside the tile
        const int c = (lane & 7) ^ swz(r);
        voff_w[s] = (unsigned)((jb.n0 + r) * g.w_rowstride) * 4u + (unsigned)c * 16u;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int n_taps = jb.n_taps;
    const int nchunks = jb.nchunks;
    const TapEntry* taps = g.taps + jb.tap_begin;

    // Staging plan: the slots of chunk c+1 are issued BETWEEN the MFMA groups of chunk c (slot s rides with k-step s % 4).
    int ld_tap = 0, ld_k = 0;
    TapEntry te_nxt = n_taps > 0 ? taps[n_taps > 1 ? 1 : 0] : TapEntry{0, 0};
    int cur_a = jb.tap0_a_off;                 // the first tap rides in the job record
    int cur_w = jb.tap0_w_off;
    int aoff = 0, woff = 0;                    // operand byte offsets of the chunk being staged (SGPRs)
    auto next_chunk_offsets = [&]() {
        aoff = (cur_a + ld_k) * 4;
        woff = (cur_w + ld_k) * 4;
        ld_k += BK;
        if (ld_k == g.kch) {
            ld_k = 0;
            ++ld_tap;
            cur_a = __builtin_amdgcn_readfirstlane(te_nxt.a_off);
            cur_w = __builtin_amdgcn_readfirstlane(te_nxt.w_off);
            const int nx = ld_tap + 1 < n_taps ? ld_tap + 1 : n_taps - 1;
            te_nxt = taps[nx];
        }
    };
    auto issue_slot = [&](int s, char* stage_base) {
        if (s < SA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, DG_LDS_PTR(stage_base + (s * 4 + wave) * 1024), 16,
                                                     voff_a[s < SA ? s : 0], aoff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, DG_LDS_PTR(stage_base + BM * ROW_BYTES + ((s - SA) * 4 + wave) * 1024), 16,
                                                     voff_w[s >= SA ? s - SA : 0], woff, 0, 0);
    };

    // fragment read addresses (byte offsets inside a stage), fixed per thread
    const int frow = lane & 31;
    const int fh = lane >> 5;
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / WM) + i * 32 + frow;
        a_rd[i] = r * ROW_BYTES;
        a_sw[i] = swz(r);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / WN) + j * 32 + frow;
        b_rd[j] = BM * ROW_BYTES + r * ROW_BYTES;
        b_sw[j] = swz(r);
    }

    // Output rows of this lane in the epilogue: pass p of tile row-block i covers tile row wm*BM/WM + i*32 + p*8 + (lane >> 3),
    // columns 4*(lane & 7)..+3 of each 32-column block.
    const int er = lane >> 3, ec = (lane & 7) * 4;
    float* out_base = g.Out + (long long)jb.n_first * g.out_rowstride + jb.n0;
    unsigned orow[TM][4];                          // float offset of the row inside the job's output window
    bool ovalid[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = wm * (BM / WM) + i * 32 + p * 8 + er;
            ovalid[i][p] = r < m_valid;
            int q, j;
            split(ovalid[i][p] ? r : 0, q, j);
            orow[i][p] = (unsigned)(q * (int)g.out_rowstride + pos_of_out(j));
        }

    // ReluGrad epilogue: the activation values that gate the result are fetched during the LAST K chunk.
    f32x4 oldv[TM][TN][4];
    // EPI_MASK: the activation the gradient overwrites; EPI_MASK_STATS: the pre-activations of the Batchnorm layer in front of it
    const float* gate_src = out_base;
    if constexpr (MODE == EPI_MASK_STATS) gate_src = g.bn_pre + (long long)jb.n_first * g.out_rowstride + jb.n0;
    auto prefetch_mask = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (BN / WN) + j * 32 + ec;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (ovalid[i][p]) v = *reinterpret_cast<const f32x4*>(gate_src + orow[i][p] + col);
                    oldv[i][j][p] = v;
                }
        }
    };
    // EPI_MASK_BITS: the gate words of this lane's rows (one 32-column group per (j)), fetched during the last K chunk
    unsigned gatew[TM][TN][4];
    const unsigned* gate_base = g.gate_bits + (long long)jb.n_first * g.gate_words;
    auto prefetch_gates = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int colb = jb.n0 + wn * (BN / WN) + j * 32;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    unsigned w = 0u;
                    if (ovalid[i][p]) w = gate_base[(orow[i][p] + (unsigned)colb) >> 5];
                    gatew[i][j][p] = w;
                }
        }
    };
    if (MODE == EPI_MASK_BITS && nchunks == 0) prefetch_gates();
    if ((MODE == EPI_MASK || MODE == EPI_MASK_STATS) && nchunks == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) oldv[i][j][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (nchunks > 0) {
        next_chunk_offsets();
        // the filter slots need nothing but the job record: they go out while the position-table loads that the A slots'
        // addresses wait for are still in flight
#pragma unroll
        for (int s = SA; s < NS; ++s) issue_slot(s, smem);
#pragma unroll
        for (int s = 0; s < SA; ++s) issue_slot(s, smem);
    }
    // One K chunk: wait for its operands, then 4 k-steps; the fragments of k-step kk+1 are read before the MFMAs of kk, and
    // the next chunk's DMA (or, in the LAST chunk of a ReluGrad tile, the gate prefetch) rides between the MFMA groups.
    auto chunk_body = [&](int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // chunk c landed for every wave; stage (c+1)&1 is free
        if constexpr (!LAST) next_chunk_offsets();
        const char* st = smem + (c & 1) * STAGE_BYTES;
        char* nx = smem + ((c + 1) & 1) * STAGE_BYTES;
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((fh ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((fh ^ b_sw[j]) << 4));
        __builtin_amdgcn_sched_group_barrier(SG_DSREAD, TM + TN, 0);     // the k-step 0 fragments
        auto kstep = [&](auto kk_tag) {
            constexpr int kk = decltype(kk_tag)::value;
            constexpr int cur = kk & 1, nxt = cur ^ 1;
            if constexpr (kk < 3) {                // fragments of k-step kk+1 while kk computes
                const int chunk = (kk + 1) * 2 + fh;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[nxt][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((chunk ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[nxt][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
            }
            constexpr int per_kk = (NS + 3) / 4;   // DMA slots riding with a k-step
            constexpr int first = kk * per_kk;
            constexpr int slots_here = LAST ? 0 : (first + per_kk <= NS ? per_kk : (NS > first ? NS - first : 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
                if constexpr (!LAST) {
                    if (e < slots_here) issue_slot(first + e, nx);        // slot first+e rides behind MFMA group e
                } else if constexpr (MODE == EPI_MASK || MODE == EPI_MASK_STATS) {
                    if (kk == 0 && e == 0) prefetch_mask();
                } else if constexpr (MODE == EPI_MASK_BITS) {
                    if (kk == 0 && e == 0) prefetch_gates();
                }
            }
            // pin the order: [fragment reads of kk+1] ([MFMA group] [DMA]) x slots_here [remaining MFMAs]
            if constexpr (kk < 3) __builtin_amdgcn_sched_group_barrier(SG_DSREAD, TM + TN, 0);
            if constexpr (slots_here >= 1) { __builtin_amdgcn_sched_group_barrier(SG_MFMA, TM * TN, 0); __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0); }
            if constexpr (slots_here >= 2) { __builtin_amdgcn_sched_group_barrier(SG_MFMA, TM * TN, 0); __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0); }
            if constexpr (slots_here >= 3) { __builtin_amdgcn_sched_group_barrier(SG_MFMA, TM * TN, 0); __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, TM * TN * (4 - (slots_here > 3 ? 3 : slots_here)), 0);
        };
        kstep(std::integral_constant<int, 0>());
        kstep(std::integral_constant<int, 1>());
        kstep(std::integral_constant<int, 2>());
        kstep(std::integral_constant<int, 3>());
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c, std::false_type());
    if (nchunks > 0) chunk_body(nchunks - 1, std::true_type());

#ifdef DG_MEASURE
    if (g.trace && tid == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        long long* t = g.trace + (long long)blockIdx.x * 4;
        t[0] = tr0; t[1] = wall_clock64(); t[2] = hwid; t[3] = nchunks;
    }
#endif
    // ---- K-pair jobs (dg_types.h): this job covered half of the class's taps.  Its raw accumulators go to the pair's scratch
    // image (write-through stores: the partner may sit on another XCD), every wave drains its stores, barrier, ONE ticket per
    // workgroup from the pair's counter (the hand-off recipe of the programming guide, guideline 16 / split-K seam).  The first
    // arriver is done; the second adds the partner's image (sc1 loads: L1 bypassed) to its own accumulators -- a + b = b + a, so
    // either arrival order gives the same bits -- and runs the epilogue.  Nobody ever waits for anybody.
    // Memory ordering, spelled out (advisor, round 5): the image leaves with sc1 (agent-scope write-through) stores; the asm
    // s_waitcnt vmcnt(0) of EVERY wave (a "memory" clobber: the compiler moves no access across it) followed by the workgroup barrier
    // means all of the image is acknowledged at the agent's coherence point before lane 0 issues the ticket; the ticket is an
    // agent-scope atomic (relaxed: the ordering is carried by the waits, not by the atomic); the second arriver's loads are sc1
    // (never this CU's L1) and are issued after its own ticket returned and a barrier.  This is the "{sc1 stores, sc1 loads}" form
    // the programming guide lists as valid on gfx950 without buffer_wbl2 / buffer_inv (each of which costs 1.7-6.5 us per use here).
    // tests/test_gpu_variants.py repeats a paired launch sequence bit for bit, also with poisoned counters.
    // (compiled only into the PAIR instantiations, which run the lists that hold such jobs: with this block present hipcc allocates
    // and schedules the main loop of EVERY instantiation differently -- Generator.3's forward, which has no pair, lost 1 % to it)
    if constexpr (PAIR) if (jb.pair_id != 0) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int TILE_FLOATS = BM * BN;
        float* const img = g.pair_scratch + (long long)jb.pair_off * 256;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(img, 0, 0x7ffffff0, 0x00020000);
        // image layout: [role][wave][i][j][quad of 4 accumulator registers][lane][4] -- every store / load instruction moves one
        // contiguous KB
        const unsigned mine = (unsigned)(jb.pair_role * TILE_FLOATS + wave * (TM * TN * 1024)) * 4u + (unsigned)lane * 16u;
        const unsigned theirs = (unsigned)((1 - jb.pair_role) * TILE_FLOATS + wave * (TM * TN * 1024)) * 4u + (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = {acc[i][j][q4 * 4 + 0], acc[i][j][q4 * 4 + 1], acc[i][j][q4 * 4 + 2], acc[i][j][q4 * 4 + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs, (int)(mine + (unsigned)(((i * TN + j) * 4 + q4) * 1024)), 0, 16);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const flag = reinterpret_cast<unsigned*>(smem);          // (both stages are free: every wave is past its last chunk)
        if (tid == 0) *flag = __builtin_amdgcn_atomic_inc32(g.pair_count + (jb.pair_id - 1), 1u, __ATOMIC_RELAXED, "agent");
        __syncthreads();
        const unsigned ticket = __builtin_amdgcn_readfirstlane(*flag);
        if (ticket == 0) return;                                            // the partner will find this half in the scratch
        __syncthreads();                                                    // (smem is reused by the epilogue's transposition tiles)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 o[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    o[q4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, (int)(theirs + (unsigned)(((i * TN + j) * 4 + q4) * 1024)), 0, 16));
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][q4 * 4 + e] += o[q4][e];
            }
    }
    // ---- epilogue: each 32x32 accumulator tile is transposed through this wave's 4 KB slice of the stage the last chunk
    // did NOT use, so that a lane owns 4 consecutive channels of one row: b128 stores (and b128 gate loads).
    float* tb = reinterpret_cast<float*>(smem + (nchunks & 1) * STAGE_BYTES + wave * 4096);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * (BN / WN) + j * 32 + ec;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_RELU || MODE == EPI_BIAS_STATS) bv = *reinterpret_cast<const f32x4*>(g.bias + jb.n0 + col);
        f32x4 bn_mu = bv, bn_rs = bv, bn_g = bv, bn_be = bv;           // EPI_MASK_STATS: the Batchnorm constants of this lane's 4 columns
        if constexpr (MODE == EPI_MASK_STATS) {
            bn_mu = *reinterpret_cast<const f32x4*>(g.bn_fstats + jb.n0 + col);
            bn_rs = *reinterpret_cast<const f32x4*>(g.bn_fstats + g.stats_cols + jb.n0 + col);
            bn_g = *reinterpret_cast<const f32x4*>(g.bn_scale + jb.n0 + col);
            bn_be = *reinterpret_cast<const f32x4*>(g.bn_offset + jb.n0 + col);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};     // EPI_BIAS_STATS: this lane's rows of the 32-row block
#pragma unroll
            for (int e = 0; e < 16; ++e) tb[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + frow] = acc[i][j][e];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 v = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // EPI_BIAS_STATS: the sums are taken of the accumulator BEFORE the bias (launch_bn_forward_from_blocks adds it back to
                    // the mean in float64): the float32 block sums of x and x^2 then carry no bias-sized offset, which
                    // E[x^2] - mean^2 would have to cancel
                    if constexpr (MODE == EPI_BIAS_STATS) {
                        if (ovalid[i][p]) { s1[q] += v[q]; s2[q] = __builtin_fmaf(v[q], v[q], s2[q]); }
                    }
                    float t = v[q] + bv[q];
                    if constexpr (MODE == EPI_BIAS_RELU) t = t > 0.f ? t : 0.f;
                    if constexpr (MODE == EPI_MASK) t = oldv[i][j][p][q] > 0.f ? t : 0.f;
                    if constexpr (MODE == EPI_MASK_BITS) t = ((gatew[i][j][p] >> (ec + q)) & 1u) ? t : 0.f;
                    if constexpr (MODE == EPI_MASK_STATS) {
                        // xhat and relu(bn(pre)) > 0 by the float expressions of bn_apply_fwd_kernel / bn_apply_bwd_kernel (dg_bn.hip)
                        const float xh = (oldv[i][j][p][q] - bn_mu[q]) * bn_rs[q];
                        t = __builtin_fmaf(xh, bn_g[q], bn_be[q]) > 0.f ? t : 0.f;
                        if (ovalid[i][p]) { s1[q] += t; s2[q] = __builtin_fmaf(t, xh, s2[q]); }
                    }
                    v[q] = t;
                }
                if (ovalid[i][p]) *reinterpret_cast<f32x4*>(out_base + orow[i][p] + col) = v;
            }
            if constexpr (MODE == EPI_BIAS_STATS || MODE == EPI_MASK_STATS) {
                // the 8 lanes that share this lane's 4 columns hold the other rows of the 32-row block (er = lane >> 3): a fixed
                // xor tree over lane bits 3..5, the same for every job shape -- a block's sums do not depend on the job list
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int m = 8; m < 64; m <<= 1) { s1[q] += __shfl_xor(s1[q], m, 64); s2[q] += __shfl_xor(s2[q], m, 64); }
                }
                const int row0 = wm * (BM / WM) + i * 32;                     // first row of this wave's 32-row block inside the job
                if (er == 0 && row0 < m_valid) {
                    const long long blk = (long long)jb.stat_base + ((jb.n_first * s_cnt + jb.j_first + row0) >> 5);
                    float* sp = g.stats + (blk * 2) * g.stats_cols + jb.n0 + col;
                    *reinterpret_cast<f32x4*>(sp) = s1;
                    *reinterpret_cast<f32x4*>(sp + g.stats_cols) = s2;
                }
            }
        }
    }
}

// FAM 0: layers with >= 128 output columns: job shapes 128x128 / 64x128 / 64x64 (waves 2 x 2).
// FAM 1: 64-column layers: 256x64 (waves 4 x 1: every wave reads the same 64 filter rows) / 128x64 / 64x64.
// MINLEVEL = the smallest shape code in the launch's job list: a list without full tiles needs less LDS and fewer
// registers, so more workgroups are resident per CU (launches too small to fill the chip with big tiles).
template <int FAM, int MODE, int MINLEVEL, bool PAIR>
__global__ __launch_bounds__(256, MINLEVEL == 0 ? 2 : (MINLEVEL == 1 ? 3 : 4)) void gemm_batched_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const JobDesc jb = g.jobs[blockIdx.x];
    const int shape = __builtin_amdgcn_readfirstlane(jb.shape);
    // wave priority by predicted job length (dg_types.h JobDesc::prio; s_setprio takes an immediate).  Lists without
    // priorities carry 0 everywhere: three scalar compares, not taken.
    {
        const int pr = __builtin_amdgcn_readfirstlane(jb.prio);
        if (pr == 3) __builtin_amdgcn_s_setprio(3);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (FAM == 0) {
        if constexpr (MINLEVEL <= 0) {
            if (shape == 0) { run_job<2, 2, 2, 2, MODE, FAM, MINLEVEL, PAIR>(g, jb, smem); return; }
        }
        if constexpr (MINLEVEL <= 1) {
            if (shape == 1) { run_job<1, 2, 2, 2, MODE, FAM, MINLEVEL, PAIR>(g, jb, smem); return; }
        }
        run_job<1, 1, 2, 2, MODE, FAM, MINLEVEL, PAIR>(g, jb, smem);
    } else {
        if constexpr (MINLEVEL <= 0) {
            if (shape == 0) { run_job<2, 2, 4, 1, MODE, FAM, MINLEVEL, PAIR>(g, jb, smem); return; }
        }
        if constexpr (MINLEVEL <= 1) {
            if (shape == 1) { run_job<2, 1, 2, 2, MODE, FAM, MINLEVEL, PAIR>(g, jb, smem); return; }
        }
        run_job<1, 1, 2, 2, MODE, FAM, MINLEVEL, PAIR>(g, jb, smem);
    }
}

template <int FAM, int MODE, int MINLEVEL, bool PAIR>
void launch_fmlp(const GemmArgs& a, hipStream_t s) {
    const int lds = gemm_lds_bytes(FAM, MINLEVEL);
    static PerDeviceOnce attr;
    if (attr.need(lds))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_batched_kernel<FAM, MODE, MINLEVEL, PAIR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((gemm_batched_kernel<FAM, MODE, MINLEVEL, PAIR>), dim3((unsigned)a.n_jobs), dim3(256), lds, s, a);
}

// a list with K-pair jobs (its scratch is set) runs the PAIR instantiation, every other list the plain one
template <int FAM, int MODE, int MINLEVEL>
void launch_fml(const GemmArgs& a, hipStream_t s) {
    if (a.pair_scratch) launch_fmlp<FAM, MODE, MINLEVEL, true>(a, s);
    else launch_fmlp<FAM, MODE, MINLEVEL, false>(a, s);
}

template <int FAM, int MODE>
void launch_fm(const GemmArgs& a, hipStream_t s) {
    if (a.min_level <= 0) launch_fml<FAM, MODE, 0>(a, s);
    else if (a.min_level == 1) launch_fml<FAM, MODE, 1>(a, s);
    else launch_fml<FAM, MODE, 2>(a, s);
}

template <int FAM>
void launch_f(const GemmArgs& a, hipStream_t s) {
    switch (a.mode) {
        case EPI_STORE: launch_fm<FAM, EPI_STORE>(a, s); break;
        case EPI_BIAS: launch_fm<FAM, EPI_BIAS>(a, s); break;
        case EPI_BIAS_RELU: launch_fm<FAM, EPI_BIAS_RELU>(a, s); break;
        case EPI_BIAS_STATS: launch_fm<FAM, EPI_BIAS_STATS>(a, s); break;
        case EPI_MASK_BITS: launch_fm<FAM, EPI_MASK_BITS>(a, s); break;
        case EPI_MASK_STATS: launch_fm<FAM, EPI_MASK_STATS>(a, s); break;
        default: launch_fm<FAM, EPI_MASK>(a, s); break;
    }
}

}  // namespace

// dynamic LDS of a launch: two stages of the largest job shape in its list
int gemm_lds_bytes(int family, int min_level) {
    const int rows = family == 0 ? (min_level <= 0 ? 256 : min_level == 1 ? 192 : 128)
                                 : (min_level <= 0 ? 320 : min_level == 1 ? 192 : 128);     // BM + BN of the largest shape
    return 2 * rows * ROW_BYTES;
}

void launch_gemm(int family, const GemmArgs& a, hipStream_t s) {
    if (a.n_jobs <= 0) return;
    if (family == 0) launch_f<0>(a, s); else launch_f<1>(a, s);
}

}  // namespace dg
