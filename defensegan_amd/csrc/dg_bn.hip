// Batch normalisation with BATCH statistics at inference, as the generator's tflib Batchnorm takes its
// `else` branch (/root/reference/tflib/ops/batchnorm.py:80-93): mean, var = tf.nn.moments(x, axes) (biased
// variance), y = (x - mean) * rsqrt(var + 1e-5) * scale + offset; is_training / moving averages are ignored.
// An activation buffer is viewed as [rows, C] (rows = latent rows x spatial positions, C = channels, or
// rows = latent rows, C = 4096 features for BN1).  All kernels are HBM-bound streaming passes:
//   partial column sums in float64 (<= 1024 row blocks x column groups of 1024) -> finalize (float64) -> apply.
// The deconv / Linear GEMM writes the PRE-ACTIVATIONS into their own buffer (BnArgs.xhat); the forward pass reads them twice
// (statistics, apply) and writes relu(bn(.)) into the activation buffer -- 3 passes, no separate xhat image (round 5; it was 4) --
// and the backward pass re-forms xhat = (pre - mean) * rstd from them on the fly (the float expression the forward used to
// store, so nothing changes in the results):  da = (scale*rstd) * (dy - mean(dy) - xhat * mean(dy*xhat)).
//
// Layout of the streaming passes (gfx950): a thread owns 4 consecutive channels (b128 loads), the C/4 channel quads of a row
// sit on adjacent lanes (a wave reads whole 256 B .. 1 KB runs), the remaining lanes of the workgroup take different rows
// ("row lanes"), and every thread keeps 4 rows in flight before it starts adding.  Sums are float64 from the first add, so
// the order in which rows are visited does not show in the float32 statistics.
#include "dg_kernels.h"

namespace dg {

constexpr int BN_MAX_BLOCKS = 1024;  // row blocks of the partial sums (upper bound; bn_part is sized for it)
constexpr int BN_MIN_ROWS = 32;      // rows per block, at least
constexpr int BN_COLS = 1024;        // channels per workgroup (256 threads x 4)
constexpr int BN_FSPLIT = 16;        // finalize: threads per channel

static inline long long bn_rows_per_block(long long rows) {
    const long long r = (rows + BN_MAX_BLOCKS - 1) / BN_MAX_BLOCKS;
    return r < BN_MIN_ROWS ? BN_MIN_ROWS : r;
}

// part[blk][0][c] = sum_r a[r][c];  part[blk][1][c] = sum_r a[r][c] * (b ? b[r][c] : a[r][c])      (rows of block blk)
// HAS_B: b holds the layer's PRE-ACTIVATIONS and fstats its forward statistics (mean, rstd): the second factor is
// xhat = (b - mean) * rstd, formed on the fly -- the same float expression the forward pass used to store
template <bool HAS_B>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ fstats,
                                                         double* __restrict__ part, long long rows, int C,
                                                         long long rows_per_blk) {
    __shared__ double red[8][256];
    const int tid = threadIdx.x;
    const int cw = C < BN_COLS ? C : BN_COLS;              // channels this workgroup covers
    const int quads = cw >> 2;                             // threads per row
    const int lanes = 256 / quads;                         // row lanes (>= 1)
    const int q = tid % quads, rl = tid / quads;
    const int c = blockIdx.y * BN_COLS + 4 * q;
    const bool live = rl < lanes && c < C;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < rows ? r0 + rows_per_blk : rows;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (live) {
        const float* pa = a + c;
        const float* pb = HAS_B ? b + c : nullptr;
        float4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
        if (HAS_B) { mu = *reinterpret_cast<const float4*>(fstats + c); rs = *reinterpret_cast<const float4*>(fstats + C + c); }
        auto xhat = [&](float4 v) { float4 o; o.x = (v.x - mu.x) * rs.x; o.y = (v.y - mu.y) * rs.y; o.z = (v.z - mu.z) * rs.z; o.w = (v.w - mu.w) * rs.w; return o; };
        long long r = r0 + rl;
        for (; r + 3LL * lanes < r1; r += 4LL * lanes) {   // 4 rows in flight
            float4 v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const float4*>(pa + (r + (long long)u * lanes) * C);
                if (HAS_B) w[u] = xhat(*reinterpret_cast<const float4*>(pb + (r + (long long)u * lanes) * C));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!HAS_B) w[u] = v[u];
                s1[0] += (double)v[u].x; s2[0] += (double)v[u].x * (double)w[u].x;
                s1[1] += (double)v[u].y; s2[1] += (double)v[u].y * (double)w[u].y;
                s1[2] += (double)v[u].z; s2[2] += (double)v[u].z * (double)w[u].z;
                s1[3] += (double)v[u].w; s2[3] += (double)v[u].w * (double)w[u].w;
            }
        }
        for (; r < r1; r += lanes) {
            const float4 v = *reinterpret_cast<const float4*>(pa + r * C);
            const float4 w = HAS_B ? xhat(*reinterpret_cast<const float4*>(pb + r * C)) : v;
            s1[0] += (double)v.x; s2[0] += (double)v.x * (double)w.x;
            s1[1] += (double)v.y; s2[1] += (double)v.y * (double)w.y;
            s1[2] += (double)v.z; s2[2] += (double)v.z * (double)w.z;
            s1[3] += (double)v.w; s2[3] += (double)v.w * (double)w.w;
        }
    }
    if (lanes > 1) {                                        // fold the row lanes (fixed order)
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[e][tid] = s1[e]; red[4 + e][tid] = s2[e]; }
        __syncthreads();
        if (rl == 0) {
            for (int k = 1; k < lanes; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += red[e][k * quads + q]; s2[e] += red[4 + e][k * quads + q]; }
        }
    }
    if (rl == 0 && c < C) {
        double* p1 = part + ((long long)blockIdx.x * 2 + 0) * C + c;
        double* p2 = part + ((long long)blockIdx.x * 2 + 1) * C + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { p1[e] = s1[e]; p2[e] = s2[e]; }
    }
}

// forward: stats[0][c] = mean, stats[1][c] = rstd = 1/sqrt(var + eps)
// backward: stats[0][c] = mean(dy), stats[1][c] = mean(dy * xhat)
// 16 channels per workgroup, 16 threads per channel: thread j of a channel adds blocks j, j+16, ... (4 loads in flight), the
// 16 sums are folded in LDS in a fixed order.
template <typename PartT>          // double: bn_partial_kernel's sums; float: the GEMM epilogue's per-32-row-block sums (EPI_BIAS_STATS)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const PartT* __restrict__ part, int nblk, long long rows, int C,
                                                          float* __restrict__ stats, int forward, const float* __restrict__ shift = nullptr) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const int cl = tid & 15, j = tid >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int k = j;
        for (; k + 3 * BN_FSPLIT < nblk; k += 4 * BN_FSPLIT) {
            double x1[4], x2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x1[u] = (double)part[((long long)(k + u * BN_FSPLIT) * 2 + 0) * C + c];
                x2[u] = (double)part[((long long)(k + u * BN_FSPLIT) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += x1[u]; s2 += x2[u]; }
        }
        for (; k < nblk; k += BN_FSPLIT) {
            s1 += (double)part[((long long)k * 2 + 0) * C + c];
            s2 += (double)part[((long long)k * 2 + 1) * C + c];
        }
    }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    if (j != 0 || c >= C) return;
    for (int k = 1; k < BN_FSPLIT; ++k) { s1 += red[0][k * 16 + cl]; s2 += red[1][k * 16 + cl]; }
    const double m1 = s1 / (double)rows, m2 = s2 / (double)rows;
    if (forward) {
        double var = m2 - m1 * m1;
        if (var < 0.0) var = 0.0;
        // (block sums of the GEMM epilogue are taken before the bias: shift[c] puts it back; the variance does not see it)
        stats[c] = (float)(m1 + (shift ? (double)shift[c] : 0.0));
        stats[C + c] = (float)(1.0 / sqrt(var + 1e-5));
    } else {
        stats[c] = (float)m1;
        stats[C + c] = (float)m2;
    }
}

// pre: pre-activations in (kept for the backward pass, which re-forms xhat from them); a: relu(bn(pre)) out (when relu)
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(float* __restrict__ a, const float* __restrict__ pre,
                                                           const float* __restrict__ stats, const float* __restrict__ scale,
                                                           const float* __restrict__ offset, long long total, int C, int relu) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const int c = (int)(i % C);
    float4 v = *reinterpret_cast<const float4*>(pre + i);
    const float4 mu = *reinterpret_cast<const float4*>(stats + c);
    const float4 rs = *reinterpret_cast<const float4*>(stats + C + c);
    const float4 g = *reinterpret_cast<const float4*>(scale + c);
    const float4 be = *reinterpret_cast<const float4*>(offset + c);
    float4 xh, o;
    xh.x = (v.x - mu.x) * rs.x; xh.y = (v.y - mu.y) * rs.y; xh.z = (v.z - mu.z) * rs.z; xh.w = (v.w - mu.w) * rs.w;
    // (explicit fma: EPI_MASK_STATS and the Batchnorm form of the MNIST tail re-form the ReLU gate from `pre` with this expression)
    o.x = __builtin_fmaf(xh.x, g.x, be.x); o.y = __builtin_fmaf(xh.y, g.y, be.y); o.z = __builtin_fmaf(xh.z, g.z, be.z); o.w = __builtin_fmaf(xh.w, g.w, be.w);
    if (relu) {
        o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    }
    *reinterpret_cast<float4*>(a + i) = o;
}

// dy (already ReluGrad-masked) in, da out (in place)
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(float* __restrict__ dy, const float* __restrict__ pre,
                                                           const float* __restrict__ fstats, const float* __restrict__ bstats,
                                                           const float* __restrict__ scale, long long total, int C) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float4 d = *reinterpret_cast<const float4*>(dy + i);
    const float4 pv = *reinterpret_cast<const float4*>(pre + i);
    const float4 mu = *reinterpret_cast<const float4*>(fstats + c);
    const float4 rs = *reinterpret_cast<const float4*>(fstats + C + c);
    float4 xh;
    xh.x = (pv.x - mu.x) * rs.x; xh.y = (pv.y - mu.y) * rs.y; xh.z = (pv.z - mu.z) * rs.z; xh.w = (pv.w - mu.w) * rs.w;
    const float4 m1 = *reinterpret_cast<const float4*>(bstats + c);
    const float4 m2 = *reinterpret_cast<const float4*>(bstats + C + c);
    const float4 g = *reinterpret_cast<const float4*>(scale + c);
    float4 o;
    o.x = (g.x * rs.x) * (d.x - m1.x - xh.x * m2.x);
    o.y = (g.y * rs.y) * (d.y - m1.y - xh.y * m2.y);
    o.z = (g.z * rs.z) * (d.z - m1.z - xh.z * m2.z);
    o.w = (g.w * rs.w) * (d.w - m1.w - xh.w * m2.w);
    *reinterpret_cast<float4*>(dy + i) = o;
}

int bn_max_blocks() { return BN_MAX_BLOCKS; }

static void launch_bn_stats(const float* a, const float* b, const BnArgs& args, float* stats, int forward, hipStream_t s) {
    const long long rpb = bn_rows_per_block(args.rows);
    const int nblk = (int)((args.rows + rpb - 1) / rpb);
    const dim3 grid((unsigned)nblk, (unsigned)((args.C + BN_COLS - 1) / BN_COLS));
    if (b) hipLaunchKernelGGL(bn_partial_kernel<true>, grid, dim3(256), 0, s, a, b, (const float*)args.fstats, args.part, (long long)args.rows, args.C, rpb);
    else hipLaunchKernelGGL(bn_partial_kernel<false>, grid, dim3(256), 0, s, a, b, (const float*)nullptr, args.part, (long long)args.rows, args.C, rpb);
    hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3((args.C + 15) / 16), dim3(256), 0, s, (const double*)args.part, nblk, (long long)args.rows,
                       args.C, stats, forward);
}

void launch_bn_forward(const BnArgs& a, int relu, hipStream_t s) {
    launch_bn_stats(a.xhat, nullptr, a, a.fstats, 1, s);          // statistics of the pre-activations the GEMM left in a.xhat
    const long long total = (long long)a.rows * a.C;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a.a, a.xhat, a.fstats,
                       a.scale, a.offset, total, a.C, relu);
}

// part[range][k][c] (double) = sum over the blocks of the range of sums[blk][k][c] (float), k = 0, 1: the GEMM epilogue's
// per-32-row-block sums (thousands of blocks for a 14x14 map) folded to at most BN_FOLD_RANGES float64 partials, coalesced over
// the channels, before the per-channel finalize (which has C / 16 workgroups only)
constexpr int BN_FOLD_RANGES = 256;
__global__ __launch_bounds__(256) void bn_fold_blocks_kernel(const float* __restrict__ sums, double* __restrict__ part, int nblk, int C,
                                                            int blocks_per_range) {
    // thread = one float4 of the [2][C] record of a block; the 2C/4 quads of a record sit on adjacent lanes, the other lanes of
    // the workgroup take other blocks of the range
    const int quads = (2 * C) / 4;
    const int per = quads < 256 ? quads : 256;             // quads this workgroup covers (2C <= 1024 floats: one pass; BN1's 8192: grid.y)
    const int lanes = 256 / per;
    const int tid = threadIdx.x;
    const int q = blockIdx.y * per + tid % per, bl = tid / per;
    const int b0 = blockIdx.x * blocks_per_range;
    const int b1 = b0 + blocks_per_range < nblk ? b0 + blocks_per_range : nblk;
    __shared__ double red[4][256];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (q < quads && bl < lanes) {
        for (int b = b0 + bl; b < b1; b += lanes) {
            const float4 v = *reinterpret_cast<const float4*>(sums + (long long)b * 2 * C + 4 * q);
            acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
        }
    }
    if (lanes > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[e][tid] = acc[e];
        __syncthreads();
        if (bl == 0 && q < quads)
            for (int k = 1; k < lanes; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += red[e][k * per + tid % per];
    }
    if (bl == 0 && q < quads) {
        double* o = part + (long long)blockIdx.x * 2 * C + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[e];
    }
}

// mean / rstd (forward) or mean(dy) / mean(dy * xhat) from nblk float block sums.  More than BN_FOLD_RANGES blocks are first folded
// to that many float64 partials (coalesced over the channels); fewer go to the finalize kernel as they are -- the same additions in
// the same order (a range would hold one block), one launch less (BN1: 80 blocks at 2560 rows; the MNIST tail: one per workgroup)
static void stats_from_blocks(const BnArgs& a, const float* block_sums, int nblk, float* stats, int forward, const float* shift, hipStream_t s);

// block sums -> at most BN_FOLD_RANGES float64 partials in a.part; returns the number of ranges
static int fold_blocks(const BnArgs& a, const float* block_sums, int nblk, hipStream_t s) {
    const int per_range = (nblk + BN_FOLD_RANGES - 1) / BN_FOLD_RANGES;
    const int ranges = (nblk + per_range - 1) / per_range;
    const int quads = (2 * a.C) / 4;
    hipLaunchKernelGGL(bn_fold_blocks_kernel, dim3((unsigned)ranges, (unsigned)((quads + 255) / 256)), dim3(256), 0, s, block_sums, a.part, nblk, a.C, per_range);
    return ranges;
}

static void stats_from_blocks(const BnArgs& a, const float* block_sums, int nblk, float* stats, int forward, const float* shift, hipStream_t s) {
    if (nblk <= BN_FOLD_RANGES) {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3((a.C + 15) / 16), dim3(256), 0, s, block_sums, nblk, (long long)a.rows, a.C, stats, forward, shift);
        return;
    }
    const int ranges = fold_blocks(a, block_sums, nblk, s);
    hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3((a.C + 15) / 16), dim3(256), 0, s, (const double*)a.part, ranges, (long long)a.rows, a.C, stats, forward,
                       shift);
}

void launch_bn_forward_from_blocks(const BnArgs& a, const float* block_sums, int nblk, int relu, hipStream_t s, const float* shift) {
    stats_from_blocks(a, block_sums, nblk, a.fstats, 1, shift, s);
    if (relu < 0) return;              // statistics only: the consumer applies relu(bn(.)) itself (the MNIST tail's Batchnorm form)
    const long long total = (long long)a.rows * a.C;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a.a, a.xhat, a.fstats,
                       a.scale, a.offset, total, a.C, relu);
}

void launch_bn_backward_from_blocks(const BnArgs& a, const float* block_sums, int nblk, hipStream_t s) {
    stats_from_blocks(a, block_sums, nblk, a.bstats, 0, nullptr, s);
    const long long total = (long long)a.rows * a.C;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a.a, a.xhat, a.fstats,
                       a.bstats, a.scale, total, a.C);
}

void launch_bn_backward(const BnArgs& a, hipStream_t s) {
    launch_bn_stats(a.a, a.xhat, a, a.bstats, 0, s);
    const long long total = (long long)a.rows * a.C;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a.a, a.xhat, a.fstats,
                       a.bstats, a.scale, total, a.C);
}

}  // namespace dg
