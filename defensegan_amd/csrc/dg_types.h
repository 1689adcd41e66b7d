// POD tables shared by the host planner (dg_plan.cpp, no HIP dependency) and the device kernels.
#pragma once
#include <stdint.h>

namespace dg {

struct PosEntry {        // one (output position, column tile); sorted by descending work
    int out_off;         // float offset of the position inside an output row
    int n0;              // first output column of this tile
    int tap_begin;       // index into the tap table
    int tap_count;
};
struct TapEntry {
    int a_off;           // float offset inside an input row
    int w_off;           // float offset into the weight buffer
};


// ---- position-batched plan (dg_gemm.hip, second generation) -----------------------------------
// Output positions whose valid taps are the same relative pattern form a CLASS; inside a class the GEMM's M axis is the
// list of (latent row n, position j) pairs, m = n * pos_count + j, so an M tile of any height is dense whatever the batch
// size, every tile of a class has the same K, and the filter slabs of a tap are shared by all its rows.
struct ClassDesc {
    int pos_begin;       // first entry of this class in the position tables
    int pos_count;       // s: positions in the class
    int tap_begin;       // first entry of this class in the tap table (a_off relative to the row's pos_a base, >= 0)
    int nchunks;         // K chunks (32 floats each) of every tile of the class = taps * kch / 32
    unsigned magic;      // ceil(2^31 / s): m / s = mulhi(2 * m, magic) for m < 2^20 (also for s == 1)
    // The positions of a class with taps are a rectangular sub-grid of the layer's maps: with wc positions per grid row,
    // pos_a[j] = a_base + (j / wc) * a_rs + (j % wc) * a_cs and likewise pos_out -- the device computes them from these numbers
    // (copied into every JobDesc) instead of fetching the tables: one dependent memory round trip less at every job start.
    // wc = 0: not a grid (the zero-tap border class of the Batchnorm crop): the tables are read.
    int wc;
    unsigned wc_magic;   // ceil(2^31 / wc)
    int a_base, a_rs, a_cs, o_base, o_rs, o_cs;
    int pad[3];
};
struct JobDesc {         // one workgroup = one job: (class, M range, column range, tile shape)
    int cls;
    int shape;           // cutting level: 0 = full tile (128x128 / 256x64), 1 = half, 2 = quarter (64x64); same arithmetic per element
    int n0;              // first output column
    int n_first;         // latent row of the job's first M row
    int j_first;         // position index (inside the class) of the job's first M row
    int m_valid;         // valid M rows of the job (<= tile height)
    // copy of the job's ClassDesc and of its first tap: one 64-byte scalar load brings everything the workgroup needs before
    // its first operand DMA (a separate class record was one more dependent L2 round trip at every job start)
    int pos_begin, pos_count, tap_begin, nchunks;
    unsigned magic;
    int tap0_a_off, tap0_w_off;
    int n_taps;          // nchunks / (kch / 32)
    int wc;              // the class's grid (ClassDesc): 0 = read the position tables
    unsigned wc_magic;
    // ---- second 64 bytes (the two scalar loads are issued together: one round trip)
    int a_base, a_rs, a_cs, o_base, o_rs, o_cs;
    // Wave priority of the job's workgroup (s_setprio, 0..3).  A job of many K chunks is one dependent chain: sharing a SIMD's
    // matrix pipe equally with its neighbours it lasts most of the launch (CelebA's 4x4 <- 8x8 backward at 1280 rows: 100-chunk
    // jobs of 170 us in a 204 us launch, whatever their tile size), and what is dispatched behind it can only start then.  With
    // priority by predicted length the long jobs take the pipe first and end early, the short ones -- which have slack -- fill in:
    // completion times spread out instead of piling up at the end of the first dispatch round.  Results cannot change.
    int prio;
    // K-pair jobs.  A tile of a class with many K chunks (dg_plan.h kPairMinChunks) is ONE dependent chain that lasts as long as
    // the whole launch when a SIMD is shared three ways; such a tile is computed by TWO jobs, each over half of the class's taps
    // (tap_begin / n_taps / nchunks / tap0_* above describe the job's own half): both leave their raw accumulators in the launch's
    // scratch (write-through stores), draw a ticket from the pair's counter, and the one that arrives second adds its partner's
    // half and runs the epilogue.  out = half0 + half1, each half a k-ordered chain from zero: float addition commutes, so the
    // result does not depend on which half arrives second, nor on the job list, nor on the batch (the classes that are
    // paired are a property of the layer plan).
    int pair_id;         // 0 = not paired; else 1 + the pair's counter index
    int pair_role;       // 0 / 1: which half of the taps
    int pair_off;        // offset of the pair's two accumulator images in the scratch, in units of 256 floats (images are multiples of
                         // 4096 floats; a float offset would pass 2^31 at ~800 000 latent rows): role r at 256 * pair_off + r * rows * columns
    int stat_base;       // EPI_BIAS_STATS: index of the class's first 32-row statistics block (classes in order, ceil(M_c / 32) blocks each)
    int pad[5];
};
static_assert(sizeof(JobDesc) == 128, "two 64-byte scalar loads");

// ---- fragment-order path (dg_fgemm.hip, round 6) ------------------------------------------------------------------------
// FRAGMENT ORDER of an activation buffer with F floats per latent row (NHWC inside the row): rows in blocks of 32; the 8 floats
// [8u, 8u + 8) of the 32 rows of block nb are ONE contiguous KB at byte (nb * F + 8u) * 128, laid out [k-half 2][row 32][4 floats]
// (float f of row n: lane = ((f % 8) / 4) * 32 + n % 32, element f % 4) -- the operand of 4 consecutive v_mfma_f32_32x32x2_f32
// for 32 rows.  Buffers are allocated for a multiple of 32 rows; the rows past the batch hold whatever the kernels computed there
// (rows are independent; nothing reads them back).
// One job = one workgroup of 4 waves = `4 / ksplit` wave tiles of TN = 4 M blocks x 2 channel blocks, each computed by `ksplit`
// waves over contiguous parts of the tile's K axis.  M block mblk of a class = (row block nb = mblk / s, position j = mblk % s).
struct FragJob {
    int mblk0;           // first M block of the job
    int n_mblk;          // valid M blocks (1 .. 4 * 4 / ksplit)
    int cb0;             // first 32-channel block of the job's 64 output channels
    int s;               // positions of the class
    unsigned s_magic;    // ceil(2^31 / s)
    int wc;              // the class's position grid (ClassDesc), >= 1
    unsigned wc_magic;
    int a_base, a_rs, a_cs, o_base, o_rs, o_cs;   // floats inside an input / output row
    int n_taps;
    int ksplit;          // 1, 2 or 4 (dg_plan.h frag_ksplit)
    // the class's taps as a grid (dg_plan.h frag_tap_grid): tap t = (u, v) = (t / tap_nw, t % tap_nw) reads the input at float
    // offset a0 + u * a_u + v * a_v behind the position's base and the filter slab at float offset w0 + u * w_u + v * w_v
    int tap_nw;
    unsigned tap_nw_magic;
    int a0, a_u, a_v, w0, w_u, w_v;
    int pad[9];
};
static_assert(sizeof(FragJob) == 128, "two 64-byte scalar loads");

}  // namespace dg
