// Gathered implicit-GEMM kernel for gfx950 (MI355X): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Covers Linear fwd/bwd and every 5x5 stride-2 transposed conv fwd / backward-to-input of the
// Defense-GAN generators (reference call sites: tflib/ops/linear.py:129-142,
// tflib/ops/deconv2d.py:100-117).  M = latent rows (B*R), N = output channels of ONE output
// position, K = (valid taps of that position) x (input channels): the tap list is resolved on the
// host per position (dg_plan.cpp), so border taps are skipped exactly and zeros are never multiplied.
//
// Data movement: both operands are K-contiguous in HBM (NHWC activations; filters [tap][n][k]), so a
// 32-float K chunk of a row is one 128-B line.  Lines go HBM/L2 -> LDS with global_load_lds
// (16 B/lane, no VGPR round trip), double buffered; the 16-B slot inside each LDS row is XOR-swizzled
// through the per-lane SOURCE address so the ds_read_b128 fragment reads are bank-conflict free.
// One b128 fragment read feeds four MFMAs: lane half h = lane>>5 holds k = 8*kk + 4*h + e for
// e = 0..3, the same K permutation on A and B.
#include <type_traits>

#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DG_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define DG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int BK = 32;                 // floats per K chunk = one 128-B line per row
constexpr int ROW_BYTES = BK * 4;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_gather_kernel(GemmArgs g) {
    constexpr int TM = BM / 64;        // 32x32 sub-tiles per wave (waves are 2 x 2)
    constexpr int TN = BN / 64;
    constexpr int SA = BM / 32;        // staging slots (16 B each) per thread, A operand
    constexpr int SB = BN / 32;
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // measurement hook: shader-clock and 100 MHz real-time stamps of workgroup 0 (effective clock under load)
    long long clk0 = 0, rt0 = 0;
    if (g.clk && blockIdx.x == 0) { clk0 = clock64(); rt0 = wall_clock64(); }
    long long tr0 = 0;
    if (g.trace) tr0 = wall_clock64();

    // Block -> (M tile, position) map.  xcd_map: workgroups are observed to land on XCD blockIdx % 8 (speed
    // only, never correctness): XCD x walks M tiles x, x+8, ... and, for each, every output position
    // (longest K first), so one M tile's input rows (1-3 MB) and the layer's filters stay in that XCD's 4 MB L2
    // while all positions that re-read them are processed.
    // Persistent mode (g.sched_off != nullptr): the grid is one resident set of workgroups, each walking its own
    // host-built tile list (longest K first, balanced over the workgroups): a finished tile is followed by the next
    // one without a trip through the hardware dispatcher (measured gap between two workgroups in one CU slot:
    // median 5.7 us, profiles/r01_v3_timeline_F2_per_workgroup.csv).
    unsigned sched_i = 0, sched_end = 0;
    int tile_id = blockIdx.x;
    if (g.sched_off) {
        sched_i = g.sched_off[blockIdx.x];
        sched_end = g.sched_off[blockIdx.x + 1];
        if (sched_i >= sched_end) return;
        tile_id = (int)g.sched_list[sched_i];
    }
  for (;;) {
    int pn, mt;
    if (g.xcd_map) {
        const int xcd = tile_id & 7;
        const int local = tile_id >> 3;
        const int mg = local / g.n_pos;
        pn = local - mg * g.n_pos;
        mt = mg * 8 + xcd;
        if (mt >= g.n_mtiles) return;
    } else {
        pn = tile_id / g.n_mtiles;
        mt = tile_id - pn * g.n_mtiles;
    }
    const PosEntry pe = g.pos[pn];
    const int pe_out_off = __builtin_amdgcn_readfirstlane(pe.out_off);
    const int pe_n0 = __builtin_amdgcn_readfirstlane(pe.n0);
    const int pe_tap_begin = __builtin_amdgcn_readfirstlane(pe.tap_begin);
    const int pe_tap_count = __builtin_amdgcn_readfirstlane(pe.tap_count);
    const int m0 = mt * BM;

    // ---- per-thread staging sources (rows are fixed for the whole tile) -------------------------
    const float* asrc[SA];
    const float* wsrc[SB];
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        const int r = (s * 4 + wave) * 8 + (lane >> 3);           // row inside the A tile
        const int c = (lane & 7) ^ swz(r);                         // source 16-B chunk for LDS slot lane&7
        int n = m0 + r;
        n = n < g.n_rows ? n : g.n_rows - 1;                       // ragged M: clamp loads, mask stores
        asrc[s] = g.A + (long long)n * g.a_rowstride + c * 4;
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int r = (s * 4 + wave) * 8 + (lane >> 3);           // output column inside the tile
        const int c = (lane & 7) ^ swz(r);
        wsrc[s] = g.W + (long long)(pe_n0 + r) * g.w_rowstride + c * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int chunks_per_tap = g.kch / BK;
    const int nchunks = pe_tap_count * chunks_per_tap;
    const TapEntry* taps = g.taps + pe_tap_begin;

    // Staging plan.  Slot s < SA stages 8 A rows per wave, slot s >= SA stages 8 filter rows; the slots of chunk
    // c+1 are issued BETWEEN the MFMA groups of chunk c (slot s rides with k-step s % 4) instead of in one burst
    // behind the barrier, where they would delay the first fragment reads of the chunk (measured: +12 % MFMA
    // rate, tools/ubench/stage_cost4.hip).  The tap record of the NEXT tap is fetched one tap early (plain load).
    constexpr int NS = SA + SB;
    int ld_tap = 0, ld_k = 0;
    TapEntry te_nxt = pe_tap_count > 0 ? taps[pe_tap_count > 1 ? 1 : 0] : TapEntry{0, 0};
    const TapEntry te0 = pe_tap_count > 0 ? taps[0] : TapEntry{0, 0};
    int cur_a = __builtin_amdgcn_readfirstlane(te0.a_off);
    int cur_w = __builtin_amdgcn_readfirstlane(te0.w_off);
    int aoff = 0, woff = 0;                    // operand offsets of the chunk being staged
    auto next_chunk_offsets = [&]() {
        aoff = cur_a + ld_k;
        woff = cur_w + ld_k;
        ld_k += BK;
        if (ld_k == g.kch) {
            ld_k = 0;
            ++ld_tap;
            cur_a = __builtin_amdgcn_readfirstlane(te_nxt.a_off);
            cur_w = __builtin_amdgcn_readfirstlane(te_nxt.w_off);
            const int nx = ld_tap + 1 < pe_tap_count ? ld_tap + 1 : pe_tap_count - 1;
            te_nxt = taps[nx];
        }
    };
    auto issue_slot = [&](int s, char* stage_base) {
        if (s < SA)
            __builtin_amdgcn_global_load_lds(DG_GLOBAL_PTR(asrc[s < SA ? s : 0] + aoff),
                                             DG_LDS_PTR(stage_base + (s * 4 + wave) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(DG_GLOBAL_PTR(wsrc[s >= SA ? s - SA : 0] + woff),
                                             DG_LDS_PTR(stage_base + BM * ROW_BYTES + ((s - SA) * 4 + wave) * 1024), 16, 0, 0);
    };

    // fragment read addresses (byte offsets inside a stage), fixed per thread
    const int frow = lane & 31;
    const int fh = lane >> 5;
    int a_rd[TM], b_rd[TN], a_sw[TM], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / 2) + i * 32 + frow;
        a_rd[i] = r * ROW_BYTES;
        a_sw[i] = swz(r);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / 2) + j * 32 + frow;
        b_rd[j] = BM * ROW_BYTES + r * ROW_BYTES;
        b_sw[j] = swz(r);
    }

    // ReluGrad epilogue: the activation values that gate the result are fetched during the LAST K chunk (instead of
    // DMA for a next chunk), so their HBM/L2 latency hides under that chunk's MFMAs.
    // Store layout of the epilogue (see below): pass p of tile (i, j) covers rows p*8 + (lane >> 3), columns 4*(lane & 7)..+3.
    const int er = lane >> 3, ec = (lane & 7) * 4;
    f32x4 oldv[TM][TN][4];
    auto prefetch_mask = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = pe_n0 + wn * (BN / 2) + j * 32 + ec;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + wm * (BM / 2) + i * 32 + er;
                const float* obase = g.Out + (long long)row0 * g.out_rowstride + pe_out_off + col;
                const unsigned rs = (unsigned)g.out_rowstride;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (row0 + p * 8 < g.n_rows) v = *reinterpret_cast<const f32x4*>(obase + (unsigned)(p * 8) * rs);
                    oldv[i][j][p] = v;
                }
            }
        }
    };
    if (MODE == EPI_MASK && nchunks == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) oldv[i][j][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (nchunks > 0) {                         // zero-tap positions (BN mode: cropped outputs) just store zeros
        next_chunk_offsets();
#pragma unroll
        for (int s = 0; s < NS; ++s) issue_slot(s, smem);
    }
    // One K chunk: wait for its operands, then 4 k-steps of MFMAs with the next chunk's DMA (or, in the LAST chunk of a
    // ReluGrad tile, the mask prefetch) slotted between the MFMA groups.  The last chunk is peeled so the steady
    // loop carries no "is there a next chunk" branches.
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
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3) {                          // fragments of k-step kk+1 while kk computes
                const int chunk = (kk + 1) * 2 + fh;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[nxt][i] = *reinterpret_cast<const f32x4*>(st + a_rd[i] + ((chunk ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[nxt][j] = *reinterpret_cast<const f32x4*>(st + b_rd[j] + ((chunk ^ b_sw[j]) << 4));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
            if constexpr (!LAST) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if ((s & 3) == kk) issue_slot(s, nx);
            } else if constexpr (MODE == EPI_MASK) {
                if (kk == 0) prefetch_mask();
            }
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
        }
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c, std::false_type());
    if (nchunks > 0) chunk_body(nchunks - 1, std::true_type());

    if (g.clk && blockIdx.x == 0 && tid == 0) {
        g.clk[0] = clock64() - clk0;
        g.clk[1] = wall_clock64() - rt0;
    }
    if (g.trace && tid == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        long long* t = g.trace + (long long)blockIdx.x * 4;
        t[0] = tr0; t[1] = wall_clock64(); t[2] = hwid; t[3] = nchunks;
    }
    // ---- epilogue -------------------------------------------------------------------------------------------------
    // D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Each 32x32 accumulator tile is transposed through
    // this wave's 4 KB slice of the stage buffer the last chunk did NOT use (nobody reads it any more: the barrier at the
    // start of the last chunk retired its readers and the last chunk issues no DMA), so that a lane owns 4 consecutive
    // channels of one row: 4 b128 stores (and, for ReluGrad, 4 b128 gate loads) per tile instead of 16 dword ones.
    float* tb = reinterpret_cast<float*>(smem + (nchunks & 1) * STAGE_BYTES + wave * 4096);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = pe_n0 + wn * (BN / 2) + j * 32 + ec;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_RELU) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) tb[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + frow] = acc[i][j][e];
            const int row0 = m0 + wm * (BM / 2) + i * 32 + er;
            float* obase = g.Out + (long long)row0 * g.out_rowstride + pe_out_off + col;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 v = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = v[q] + bv[q];
                    if constexpr (MODE == EPI_BIAS_RELU) t = t > 0.f ? t : 0.f;
                    if constexpr (MODE == EPI_MASK) t = oldv[i][j][p][q] > 0.f ? t : 0.f;
                    v[q] = t;
                }
                if (row0 + p * 8 < g.n_rows) *reinterpret_cast<f32x4*>(obase + (long long)(p * 8) * g.out_rowstride) = v;
            }
        }
    }
    if (!g.sched_off) return;
    if (++sched_i >= sched_end) return;
    tile_id = (int)g.sched_list[sched_i];
    __syncthreads();                             // every wave is done with the LDS stages of this tile
  }
}

template <int BM, int BN, int MODE>
static void launch_tm(const GemmArgs& a, int n_pos, hipStream_t s) {
    const int lds = 2 * (BM + BN) * ROW_BYTES + (a.lds_pad > 0 ? a.lds_pad : 0);   // 64x64: exactly 32 KB, five fit in 160 KB
    static PerDeviceOnce attr;
    if (attr.need(lds)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_gather_kernel<BM, BN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    unsigned grid = a.xcd_map ? 8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)n_pos
                              : (unsigned)n_pos * (unsigned)a.n_mtiles;
    if (a.sched_off) grid = (unsigned)a.sched_grid;
    hipLaunchKernelGGL((gemm_gather_kernel<BM, BN, MODE>), dim3(grid), dim3(256), lds, s, a);
}

template <int BM, int BN>
static void launch_t(const GemmArgs& a, int n_pos, hipStream_t s) {
    switch (a.mode) {
        case EPI_STORE: launch_tm<BM, BN, EPI_STORE>(a, n_pos, s); break;
        case EPI_BIAS: launch_tm<BM, BN, EPI_BIAS>(a, n_pos, s); break;
        case EPI_BIAS_RELU: launch_tm<BM, BN, EPI_BIAS_RELU>(a, n_pos, s); break;
        default: launch_tm<BM, BN, EPI_MASK>(a, n_pos, s); break;
    }
}

static const int kTileBM[4] = {128, 64, 128, 64};
static const int kTileBN[4] = {128, 128, 64, 64};
int gemm_tile_bm(int tile) { return kTileBM[tile & 3]; }
int gemm_tile_bn(int tile) { return kTileBN[tile & 3]; }

void launch_gemm(int tile, const GemmArgs& a, int n_pos, hipStream_t s) {
    switch (tile & 3) {
        case 0: launch_t<128, 128>(a, n_pos, s); break;
        case 1: launch_t<64, 128>(a, n_pos, s); break;
        case 2: launch_t<128, 64>(a, n_pos, s); break;
        default: launch_t<64, 64>(a, n_pos, s); break;
    }
}

}  // namespace dg
