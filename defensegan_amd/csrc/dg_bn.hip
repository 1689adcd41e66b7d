// Batch normalisation with BATCH statistics at inference, as the generator's tflib Batchnorm takes its
// `else` branch (/root/reference/tflib/ops/batchnorm.py:80-93): mean, var = tf.nn.moments(x, axes) (biased
// variance), y = (x - mean) * rsqrt(var + 1e-5) * scale + offset; is_training / moving averages are ignored.
// An activation buffer is viewed as [rows, C] (rows = latent rows x spatial positions, C = channels, or
// rows = latent rows, C = 4096 features for BN1).  All kernels are HBM-bound streaming passes:
//   partial column sums in float64 (one workgroup per block of rows) -> finalize (float64) -> apply.
// Forward keeps xhat for the backward:  da = (scale*rstd) * (dy - mean(dy) - xhat * mean(dy*xhat)).
#include "dg_kernels.h"

namespace dg {

constexpr int BN_RB = 128;       // rows per partial-sum workgroup

// part[blk][0][c] = sum_r a[r][c];  part[blk][1][c] = sum_r a[r][c] * (b ? b[r][c] : a[r][c])
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         double* __restrict__ part, long long rows, int C) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const int cpt = C < 256 ? C : 256;          // columns per pass
    const int rs = 256 / cpt;                   // row sub-lanes when C < 256
    const int c_in = tid % cpt, rsub = tid / cpt;
    const long long r0 = (long long)blockIdx.x * BN_RB;
    const long long r1 = r0 + BN_RB < rows ? r0 + BN_RB : rows;
    for (int c0 = 0; c0 < C; c0 += cpt) {
        const int c = c0 + c_in;
        double s1 = 0.0, s2 = 0.0;
        if (rsub < rs) {
            for (long long r = r0 + rsub; r < r1; r += rs) {
                const float v = a[r * C + c];
                const float w = b ? b[r * C + c] : v;
                s1 += (double)v;
                s2 += (double)v * (double)w;
            }
        }
        if (rs > 1) {
            red[0][tid] = s1; red[1][tid] = s2;
            __syncthreads();
            if (rsub == 0) {
                for (int k = 1; k < rs; ++k) { s1 += red[0][k * cpt + c_in]; s2 += red[1][k * cpt + c_in]; }
            }
            __syncthreads();
        }
        if (rsub == 0) {
            part[((long long)blockIdx.x * 2 + 0) * C + c] = s1;
            part[((long long)blockIdx.x * 2 + 1) * C + c] = s2;
        }
    }
}

// forward: stats[0][c] = mean, stats[1][c] = rstd = 1/sqrt(var + eps)
// backward: stats[0][c] = mean(dy), stats[1][c] = mean(dy * xhat)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int nblk, long long rows, int C,
                                                          float* __restrict__ stats, int forward) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nblk; ++k) {
        s1 += part[((long long)k * 2 + 0) * C + c];
        s2 += part[((long long)k * 2 + 1) * C + c];
    }
    const double m1 = s1 / (double)rows, m2 = s2 / (double)rows;
    if (forward) {
        double var = m2 - m1 * m1;
        if (var < 0.0) var = 0.0;
        stats[c] = (float)m1;
        stats[C + c] = (float)(1.0 / sqrt(var + 1e-5));
    } else {
        stats[c] = (float)m1;
        stats[C + c] = (float)m2;
    }
}

// a: pre-activation in, relu(bn(a)) out (when relu) ; xhat out
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(float* __restrict__ a, float* __restrict__ xhat,
                                                           const float* __restrict__ stats, const float* __restrict__ scale,
                                                           const float* __restrict__ offset, long long total, int C, int relu) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const int c = (int)(i % C);
    float4 v = *reinterpret_cast<const float4*>(a + i);
    const float4 mu = *reinterpret_cast<const float4*>(stats + c);
    const float4 rs = *reinterpret_cast<const float4*>(stats + C + c);
    const float4 g = *reinterpret_cast<const float4*>(scale + c);
    const float4 be = *reinterpret_cast<const float4*>(offset + c);
    float4 xh, o;
    xh.x = (v.x - mu.x) * rs.x; xh.y = (v.y - mu.y) * rs.y; xh.z = (v.z - mu.z) * rs.z; xh.w = (v.w - mu.w) * rs.w;
    o.x = xh.x * g.x + be.x; o.y = xh.y * g.y + be.y; o.z = xh.z * g.z + be.z; o.w = xh.w * g.w + be.w;
    if (relu) {
        o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    }
    *reinterpret_cast<float4*>(xhat + i) = xh;
    *reinterpret_cast<float4*>(a + i) = o;
}

// dy (already ReluGrad-masked) in, da out (in place)
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(float* __restrict__ dy, const float* __restrict__ xhat,
                                                           const float* __restrict__ fstats, const float* __restrict__ bstats,
                                                           const float* __restrict__ scale, long long total, int C) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float4 d = *reinterpret_cast<const float4*>(dy + i);
    const float4 xh = *reinterpret_cast<const float4*>(xhat + i);
    const float4 rs = *reinterpret_cast<const float4*>(fstats + C + c);
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

int bn_num_blocks(int64_t rows) { return (int)((rows + BN_RB - 1) / BN_RB); }

void launch_bn_forward(const BnArgs& a, int relu, hipStream_t s) {
    const int nblk = bn_num_blocks(a.rows);
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nblk), dim3(256), 0, s, a.a, (const float*)nullptr, a.part, (long long)a.rows, a.C);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.C + 255) / 256), dim3(256), 0, s, a.part, nblk, (long long)a.rows, a.C, a.fstats, 1);
    const long long total = (long long)a.rows * a.C;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a.a, a.xhat, a.fstats,
                       a.scale, a.offset, total, a.C, relu);
}

void launch_bn_backward(const BnArgs& a, hipStream_t s) {
    const int nblk = bn_num_blocks(a.rows);
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nblk), dim3(256), 0, s, a.a, (const float*)a.xhat, a.part, (long long)a.rows, a.C);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.C + 255) / 256), dim3(256), 0, s, a.part, nblk, (long long)a.rows, a.C, a.bstats, 0);
    const long long total = (long long)a.rows * a.C;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a.a, a.xhat, a.fstats,
                       a.bstats, a.scale, total, a.C);
}

}  // namespace dg
