// The latent turn of a projection step as ONE launch on gfx950 (MI355X): Linear backward (split-K partials of dz) -> momentum
// update -> Linear forward + BiasAdd + ReLU of the NEXT step.  Reference call sites: tflib/ops/linear.py:129-142 inside the loop
// body of models/gan.py:409-437 (ApplyMomentum: gan.py:389-391).  Engine option turn_fused.
//
// Why it was built: as three launches (dg_linear.hip, dg_small.hip) the turn costs 62 us per GD iteration at 2560 rows for 34 us
// of matrix pipe: two kernel boundaries, three ramps, and each Linear launch waits for its stationary weights before its first
// multiply.  Here the 16 x 16 workgroups of the backward launch stay: workgroup (K slice s, row group g) multiplies its slice of
// the group's 32-row blocks exactly as lin_stationary_kernel<8, EPI_STORE> does, the 16 workgroups of a ROW GROUP then meet at a
// barrier of their own (no grid-wide barrier: row groups never exchange data), each updates 1/16 of the group's rows
// (momentum_update_kernel's arithmetic: slices added from zero in slice order, the same two fmas), they meet again, and
// workgroup (s, g) computes the forward column tiles 2s and 2s + 1 of the group's blocks -- TWO accumulators per wave over ONE
// staged image of z (the forward's weights arrive while the workgroup waits at the second barrier).
// What it measured (profiles/r06_ab_turn_fused.txt): 62-69 us -- at parity with the three launches.  The seam inside the launch
// (drain, arrive, poll, sc1 loads, drain, arrive, poll, sc1 DMA) is 17 us of serialized round trips; OFF by default.
//
// Hand-off (programming guide, section 6 guideline 16, {sc1 stores, sc1 loads} form): partials and z leave with write-through
// (sc1) 16-byte stores, every wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds one to the barrier's monotonic
// arrival counter (relaxed, agent scope) and polls it with relaxed agent-scope (sc1) loads + s_sleep until its episode is
// complete, workgroup barrier, then the readers use sc1 loads (partials) / sc1 LDS-DMA (z).
// Residency: 80 KB of LDS and at most 256 registers per lane admit TWO of these workgroups per CU; the engine keeps the
// workgroups of all turn launches that may be in flight at once (concurrent row groups of one call) within 2 x CUs, and a
// workgroup only waits for the others of its row group; kernels of other streams finish without this one.  A poll that does not
// end within ~0.5 s gives up and raises TurnArgs::err (the engine fails the handle's next call) instead of hanging the device.
//
// Arithmetic: every element of the partials, of z / m and of the activations is produced by the same k-ordered fp32 fma chain
// and the same expressions as by the three separate kernels: BIT-IDENTICAL (tests/test_gpu_variants.py).
#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int KB = 8;                 // 32-float chunks of a backward K slice (slice = 256 features)
constexpr int KF = 4;                 // 32-float chunks of the latent (128)
constexpr unsigned kSpinLimit = 1u << 19;

__device__ __forceinline__ int lswz(int row) { return (row >> 1) & 7; }

// Barrier of the `n` workgroups of a row group on ONE monotonic arrival counter (zeroed by the engine at the start of every
// call): the arrival that draws ticket t belongs to episode t / n and waits until the counter has reached (t / n + 1) * n.
// Caller: every wave has drained its write-through stores and the workgroup has passed a barrier.  Returns with the workgroup
// re-converged; `between` runs after the arrival, before the wait.
template <class F>
__device__ __forceinline__ void group_barrier(unsigned* cnt, unsigned n, unsigned* err, F between) {
    unsigned target = 0;
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target = (t / n + 1u) * n;
    }
    between();
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

// ApplyMomentum on workgroup `unit`'s share of row group `grp`: of each of the group's blocks the rows unit * (32 / NS) ..., a
// thread per (row, 4 latent components), ITEMS of them per thread and round with every load issued before the first add (the
// phase is one memory round trip per round).  momentum_update_kernel's arithmetic: slices added from zero in slice order, then
// m <- fma(momentum, m, g), z <- fma(-lr, m, z).  The partials were written through by other workgroups of this launch: sc1
// loads; z leaves write-through for the forward phase of the whole row group.
template <int NS, int ITEMS>
__device__ __forceinline__ void turn_update(const TurnArgs& g, int tid, int unit, int grp, int n_my) {
    constexpr int rows_per = 32 / NS;
    constexpr int per_blk = rows_per * 32;
    constexpr int prow = NS * 128;
    const int n_items = n_my * per_blk;
    // (descriptors of the whole buffers, the row in the per-lane offset: a per-lane descriptor would be a waterfall loop.
    // 32-bit byte offsets: kTurnMaxRows)
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(g.part, 0, 0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(g.z, 0, 0xfffffff0u, 0x00020000);
#pragma unroll 1
    for (int base = 0; base < n_items; base += 256 * ITEMS) {
        f32x4 pv[ITEMS][NS];
        f32x4 mv[ITEMS], zv[ITEMS];
        bool ok[ITEMS];
        float* mp[ITEMS];
        int zoff[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            int it = base + j * 256 + tid;
            ok[j] = it < n_items;
            it = ok[j] ? it : 0;
            const int i = it / per_blk, rem = it - i * per_blk;
            int r = ((grp + i * g.groups) << 5) + unit * rows_per + (rem >> 5);
            ok[j] = ok[j] && r < g.n_rows;
            r = r < g.n_rows ? r : g.n_rows - 1;                       // (loads of a masked item stay inside the buffers)
            const int col = (rem & 31) * 4;
            const unsigned pbase = (unsigned)r * (unsigned)(prow * 4) + (unsigned)col * 4u;
            mp[j] = g.m + (long long)r * 128 + col;
            zoff[j] = (int)((unsigned)r * 512u + (unsigned)col * 4u);
            mv[j] = *reinterpret_cast<const f32x4*>(mp[j]);
            zv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrsrc, zoff[j], 0, 16));
#pragma unroll
            for (int s = 0; s < NS; ++s)
                pv[j][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)pbase, s * 512, 16));
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) sum += pv[j][s];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mv[j][q] = __builtin_fmaf(g.momentum, mv[j][q], sum[q]);
                zv[j][q] = __builtin_fmaf(-g.lr, mv[j][q], zv[j][q]);
            }
            if (ok[j]) {
                *reinterpret_cast<f32x4*>(mp[j]) = mv[j];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zv[j]), zrsrc, zoff[j], 0, 16);
            }
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256, 2) void latent_turn_kernel(TurnArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BLKB = KB * 4096;                   // one 32-row block of the backward operand: [8 chunks][32 rows][128 B]
    constexpr int BLKF = KF * 4096;                   // one 32-row block of z
    // transposition tiles: backward [4 waves][32 x 32 floats] behind its two blocks; forward [4 waves][2 tiles][32 x 32] in the
    // half its two (smaller) blocks leave free -- 80 KB in all, so that two of these workgroups fit a CU (concurrent row groups)
    char* const epi_b = smem + 2 * BLKB;
    char* const epi_f = smem + BLKB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;

    const int unit = (int)blockIdx.x % g.nsplit;      // K slice (backward) / pair of column tiles (forward)
    const int grp = (int)blockIdx.x / g.nsplit;       // row group: blocks grp, grp + groups, ...
    const int n_blocks = (g.n_rows + 31) >> 5;
    const int n_my = (n_blocks - grp + g.groups - 1) / g.groups;          // >= 1: groups <= n_blocks (launch_latent_turn)
    unsigned* const bar = g.bar + grp * 2;
#ifdef DG_MEASURE
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool tron = g.trace != nullptr && tid == 0;
// (the 100 MHz real-time counter: one time base for all XCDs, unlike the shader-clock counter)
#define DG_TURN_STAMP(k) do { if (tron) tr[k] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DG_TURN_STAMP(k) do { } while (0)
#endif
    DG_TURN_STAMP(0);

    const int srow = wave * 8 + (lane >> 3);
    const unsigned sslot = (unsigned)(((lane & 7) ^ lswz(srow)) << 4);
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const int a_rd = frow * 128;
    const int a_sw = lswz(frow);

    // ================= phase B: split-K partials of dz (lin_stationary_kernel<8, EPI_STORE>, write-through stores) =================
    {
        const float* const a_unit = g.dA + (long long)unit * (KB * 32);
        auto stage = [&](int blk, char* dst) {
            const int row0 = blk << 5;
            int r = row0 + srow;
            r = r < g.n_rows ? r : g.n_rows - 1;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a_unit + (long long)row0 * g.features), 0, 0x7ffffff0, 0x00020000);
            const unsigned voff = (unsigned)(r - row0) * (unsigned)(g.features * 4) + sslot;
#pragma unroll
            for (int c = 0; c < KB; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DG_LDS_PTR(dst + c * 4096 + wave * 1024), 16, voff, c * 128, 0, 0);
        };
        stage(grp, smem);
        f32x4 wf[KB][4];
        {
            const float* wp = g.Wb + lin_pack_index(unit, wave, KB, 0, 0, lane, 0);
#pragma unroll
            for (int c = 0; c < KB; ++c)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wf[c][kk] = *reinterpret_cast<const f32x4*>(wp + (c * 4 + kk) * 256);
        }
        float* const tb = reinterpret_cast<float*>(epi_b + wave * 4096);
        const int ocol = unit * 128 + wave * 32 + ec;
        const int prow = g.nsplit * 128;                                   // floats per row of the partials
        // block 0 and the weights are there; block 1 (8 DMA instructions, issued last) still travels
        if (n_my > 1) { stage(grp + g.groups, smem + BLKB); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < KB; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(wf[c][kk]));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DG_TURN_STAMP(1);

        auto out_write = [&](const f32x16& acc) {
#pragma unroll
            for (int e = 0; e < 16; ++e) tb[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + frow] = acc[e];
        };
        auto out_read = [&](f32x4 (&v)[4]) {
#pragma unroll
            for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const f32x4*>(tb + (p * 8 + er) * 32 + ec);
        };
        auto out_store = [&](f32x4 (&v)[4], int blk) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int row0 = blk << 5;
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                g.part + (long long)row0 * prow, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                asm volatile("" : "+v"(v[p]));
#pragma unroll
                for (int q = 0; q < 4; ++q) v[p][q] = v[p][q] + 0.f;      // (EPI_STORE adds a zero bias in the separate kernel)
                const int r = row0 + p * 8 + er;
                if (r < g.n_rows)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[p]), orsrc,
                                                           (int)((unsigned)(p * 8 + er) * (unsigned)(prow * 4) + (unsigned)ocol * 4u), 0, 16);
            }
        };

        // The hand-over to the next block sits INSIDE the last chunk of this one: once this wave has fetched its last fragments
        // of the current buffer, it waits for its own pieces of block i+1 (everything but the 4 row stores of block i-1 issued
        // after them), meets the other waves (their pieces are there too, and nobody reads the current buffer any more), sends
        // block i+2 into the buffer just freed and fetches block i+1's first fragments under the last chunk's multiplies -- the
        // matrix pipe does not see the block boundary.
        f32x16 done;
        f32x4 a[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[0][kk] = *reinterpret_cast<const f32x4*>(smem + a_rd + (((kk * 2 + fh) ^ a_sw) << 4));
        for (int i = 0; i < n_my; ++i) {
            const int blk = grp + i * g.groups;
            const char* st = smem + (i & 1) * BLKB + a_rd;
            const char* stn = smem + ((i + 1) & 1) * BLKB + a_rd;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            f32x4 v[4];
            if (i > 0) out_write(done);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < KB; ++c) {
                if (c + 1 < KB) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        a[(c + 1) & 1][kk] = *reinterpret_cast<const f32x4*>(st + (c + 1) * 4096 + (((kk * 2 + fh) ^ a_sw) << 4));
                } else if (i + 1 < n_my) {
                    if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (i + 2 < n_my) stage(blk + 2 * g.groups, smem + (i & 1) * BLKB);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) a[0][kk] = *reinterpret_cast<const f32x4*>(stn + (((kk * 2 + fh) ^ a_sw) << 4));
                }
                if (c == 0 && i > 0) out_read(v);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][kk][e], wf[c][kk][e], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (c == 1 && i > 0) out_store(v, blk - g.groups);
            }
            done = acc;
        }
        DG_TURN_STAMP(2);
        f32x4 v[4];
        out_write(done);
        out_read(v);
        out_store(v, grp + (n_my - 1) * g.groups);
    }
    // every wave's partials are acknowledged; then the workgroup; then the row group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    DG_TURN_STAMP(3);
    group_barrier(bar + 0, (unsigned)g.nsplit, g.err, [] {});
    DG_TURN_STAMP(4);

    // ================= phase U: ApplyMomentum on this workgroup's share of the group's rows =================
    if (g.nsplit == 16) turn_update<16, 2>(g, tid, unit, grp, n_my);
    else if (g.nsplit == 32) turn_update<32, 1>(g, tid, unit, grp, n_my);
    else turn_update<8, 2>(g, tid, unit, grp, n_my);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    DG_TURN_STAMP(5);

    // ================= phase F: h = relu(z . W^T + b), column tiles 2 * unit and 2 * unit + 1 of the group's blocks =================
    f32x4 wf0[KF][4], wf1[KF][4];
    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
    const int tile0 = 2 * unit;
    // (the forward's weights travel while the workgroup waits for the rest of its row group)
    group_barrier(bar + 1, (unsigned)g.nsplit, g.err, [&] {
        const float* wp0 = g.Wf + lin_pack_index(tile0, wave, KF, 0, 0, lane, 0);
        const float* wp1 = g.Wf + lin_pack_index(tile0 + 1, wave, KF, 0, 0, lane, 0);
#pragma unroll
        for (int c = 0; c < KF; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                wf0[c][kk] = *reinterpret_cast<const f32x4*>(wp0 + (c * 4 + kk) * 256);
                wf1[c][kk] = *reinterpret_cast<const f32x4*>(wp1 + (c * 4 + kk) * 256);
            }
        bv0 = *reinterpret_cast<const f32x4*>(g.bias + tile0 * 128 + wave * 32 + ec);
        bv1 = *reinterpret_cast<const f32x4*>(g.bias + (tile0 + 1) * 128 + wave * 32 + ec);
    });
    DG_TURN_STAMP(6);
    {
        auto stage = [&](int blk, char* dst) {
            const int row0 = blk << 5;
            int r = row0 + srow;
            r = r < g.n_rows ? r : g.n_rows - 1;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                g.z + (long long)row0 * 128, 0, 0x7ffffff0, 0x00020000);
            const unsigned voff = (unsigned)(r - row0) * 512u + sslot;
#pragma unroll
            for (int c = 0; c < KF; ++c)     // sc1: z was written through by other workgroups of this launch
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DG_LDS_PTR(dst + c * 4096 + wave * 1024), 16, voff, c * 128, 0, 16);
        };
        stage(grp, smem);
        // block 0 and the weights (issued before the barrier wait) are there; block 1 (4 DMA instructions) still travels
        if (n_my > 1) { stage(grp + g.groups, smem + BLKF); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < KF; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { asm volatile("" : "+v"(wf0[c][kk])); asm volatile("" : "+v"(wf1[c][kk])); }
        asm volatile("" : "+v"(bv0));
        asm volatile("" : "+v"(bv1));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        float* const tb = reinterpret_cast<float*>(epi_f + wave * 8192);
        float* const tb1 = tb + 1024;
        auto out_write = [&](const f32x16& acc, float* t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) t[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + frow] = acc[e];
        };
        auto out_read = [&](f32x4 (&v)[4], const float* t) {
#pragma unroll
            for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const f32x4*>(t + (p * 8 + er) * 32 + ec);
        };
        auto out_store = [&](f32x4 (&v)[4], const f32x4& bv, int blk, int ocol) {
            const int row0 = blk << 5;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                asm volatile("" : "+v"(v[p]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = v[p][q] + bv[q];
                    t = t > 0.f ? t : 0.f;
                    v[p][q] = t;
                }
                const int r = row0 + p * 8 + er;
                if (r < g.n_rows) *reinterpret_cast<f32x4*>(g.H + (long long)r * g.features + ocol) = v[p];
            }
        };
        const int ocol0 = tile0 * 128 + wave * 32 + ec;

        // (the hand-over to the next block inside the last chunk, as in the backward phase; 8 row stores per block here)
        f32x16 done0, done1;
        f32x4 a[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[0][kk] = *reinterpret_cast<const f32x4*>(smem + a_rd + (((kk * 2 + fh) ^ a_sw) << 4));
        for (int i = 0; i < n_my; ++i) {
            const int blk = grp + i * g.groups;
            const char* st = smem + (i & 1) * BLKF + a_rd;
            const char* stn = smem + ((i + 1) & 1) * BLKF + a_rd;
            f32x16 acc0, acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
            f32x4 v0[4], v1[4];
            if (i > 0) { out_write(done0, tb); out_write(done1, tb1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < KF; ++c) {
                if (c + 1 < KF) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        a[(c + 1) & 1][kk] = *reinterpret_cast<const f32x4*>(st + (c + 1) * 4096 + (((kk * 2 + fh) ^ a_sw) << 4));
                } else if (i + 1 < n_my) {
                    if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (i + 2 < n_my) stage(blk + 2 * g.groups, smem + (i & 1) * BLKF);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) a[0][kk] = *reinterpret_cast<const f32x4*>(stn + (((kk * 2 + fh) ^ a_sw) << 4));
                }
                if (c == 0 && i > 0) { out_read(v0, tb); out_read(v1, tb1); }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][kk][e], wf0[c][kk][e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][kk][e], wf1[c][kk][e], acc1, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (c == 1 && i > 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    out_store(v0, bv0, blk - g.groups, ocol0);
                    out_store(v1, bv1, blk - g.groups, ocol0 + 128);
                }
            }
            done0 = acc0;
            done1 = acc1;
        }
        const int last_blk = grp + (n_my - 1) * g.groups;
        f32x4 v0[4], v1[4];
        out_write(done0, tb);
        out_write(done1, tb1);
        out_read(v0, tb);
        out_read(v1, tb1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        out_store(v0, bv0, last_blk, ocol0);
        out_store(v1, bv1, last_blk, ocol0 + 128);
    }
#ifdef DG_MEASURE
    if (tron) {
        tr[7] = (long long)__builtin_amdgcn_s_memrealtime();
        long long* t = g.trace + (long long)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = tr[k];
    }
#endif
}

// nsplit K slices of 256 features, two forward column tiles per slice, 128 latent components; the update reads the slices eight
// at a time and splits a block's 32 rows over the nsplit workgroups of a row group
bool turn_fused_supported(int nsplit, int latent, int features) {
    return latent == 128 && nsplit >= 8 && nsplit % 8 == 0 && 32 % nsplit == 0 && features == nsplit * 256;
}

void launch_latent_turn(const TurnArgs& a, hipStream_t s) {
    if (a.n_rows <= 0 || a.groups <= 0 || a.n_rows > kTurnMaxRows) return;
    const int lds = 2 * KB * 4096 + 4 * 4096;
    static PerDeviceOnce attr;
    if (attr.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_turn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(latent_turn_kernel, dim3((unsigned)(a.nsplit * a.groups)), dim3(256), lds, s, a);
}

}  // namespace dg
