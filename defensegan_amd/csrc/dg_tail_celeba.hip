// CelebA tail of the projection step, gfx950:
//   Generator.6 (Deconv2D C -> 3, 32x32 -> 64x64) + tanh                 dataset_models.py:160-163
//   image_rec_loss = mean_pix (G(z) - x)^2                               gan.py:410-414
//   backward: dY = 2(y-x)/P, TanhGrad (1-y^2), Conv2D s2 (grad of the transpose) -> da5 (Generator.5 has no
//   nonlinearity, so no ReluGrad here)
// Cout = 3 keeps this off the MFMA path (VALU out of LDS, like the MNIST tail), but one row's input map is
// 256 KB, so the work is banded: forward = 8 output rows per workgroup (6 input rows + zero halo in LDS),
// backward = 4 input rows per workgroup (11 rows of da6 + zero halo in LDS).  da6 crosses band borders, hence
// two kernels with da6 parked in HBM (49 KB per row) between them.  The loss is reduced per band, then over
// the 8 bands in a fixed order.
#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CB_IN_ROWS = 6;     // input rows per forward band (4 + halo)
constexpr int CB_IN_COLS = 34;    // 32 + zero border
constexpr int CB_G_ROWS = 11;     // da6 rows per backward band
constexpr int CB_G_COLS = 68;     // 64 + zero border (67 used), padded

template <int C, int PI, int PJ>
__device__ __forceinline__ void celeba_fwd_class(const float* sin, const float* __restrict__ F6, const float* __restrict__ b6,
                                                 const float* __restrict__ xrow, float* __restrict__ grow,
                                                 float* __restrict__ yrow, int i0, int c4, int slot, float gscale,
                                                 float& sq) {
    constexpr int G = C / 4, SLOTS = 256 / G;
    constexpr int NH = PI ? 3 : 2, NW = PJ ? 3 : 2;
    f32x4 w[NH][NW][3];
#pragma unroll
    for (int a = 0; a < NH; ++a)
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const int kh = PI ? 2 * a : 2 * a + 1, kw = PJ ? 2 * b : 2 * b + 1;
#pragma unroll
            for (int co = 0; co < 3; ++co)
                w[a][b][co] = *reinterpret_cast<const f32x4*>(F6 + ((kh * 5 + kw) * 3 + co) * C + c4 * 4);
        }
    for (int p0 = 0; p0 < 128; p0 += SLOTS) {
        const int p = p0 + slot;
        const int t = p >> 5, u = p & 31;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NH; ++a) {
            const int lr = t + (PI ? 2 : 1) - a;
#pragma unroll
            for (int b = 0; b < NW; ++b) {
                const int lc = u + (PJ ? 2 : 1) - b;
                const f32x4 h = *reinterpret_cast<const f32x4*>(sin + (lr * CB_IN_COLS + lc) * C + c4 * 4);
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const f32x4 ww = w[a][b][co];
                    acc[co] = __builtin_fmaf(h[0], ww[0], acc[co]);
                    acc[co] = __builtin_fmaf(h[1], ww[1], acc[co]);
                    acc[co] = __builtin_fmaf(h[2], ww[2], acc[co]);
                    acc[co] = __builtin_fmaf(h[3], ww[3], acc[co]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
            for (int m = 1; m < G; m <<= 1) acc[co] += __shfl_xor(acc[co], m, 64);
        if (c4 < 3) {
            const float pre = (c4 == 0 ? acc[0] : (c4 == 1 ? acc[1] : acc[2])) + b6[c4];
            const float y = tanhf(pre);
            const int i = i0 + 2 * t + PI, j = 2 * u + PJ;
            const int o = (i * 64 + j) * 3 + c4;
            const float d = y - xrow[o];
            sq = __builtin_fmaf(d, d, sq);
            grow[o] = gscale * d * (1.0f - y * y);
            if (yrow) yrow[o] = y;
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void celeba_tail_fwd_kernel(CelebaTailArgs a) {
    constexpr int G = C / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sin = reinterpret_cast<float*>(smem);               // [6][34][C]
    float* sred = sin + CB_IN_ROWS * CB_IN_COLS * C;           // [4]
    const int tid = threadIdx.x;
    const int n = blockIdx.x >> 3, band = blockIdx.x & 7;
    const int b = n / a.R;
    const int c4 = tid % G, slot = tid / G;
    const float* hrow = a.h5 + (long long)n * (1024 * C);
    const int oh_lo = 4 * band - 1;
    for (int i = tid; i < CB_IN_ROWS * CB_IN_COLS * G; i += 256) {
        const int g = i % G, pc = i / G;
        const int lr = pc / CB_IN_COLS, lc = pc % CB_IN_COLS;
        const int oh = oh_lo + lr, ow = lc - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (oh >= 0 && oh < 32 && ow >= 0 && ow < 32) v = *reinterpret_cast<const f32x4*>(hrow + (oh * 32 + ow) * C + g * 4);
        *reinterpret_cast<f32x4*>(sin + pc * C + g * 4) = v;
    }
    __syncthreads();
    const float* xrow = a.x + (long long)b * 12288;
    float* grow = a.g6 + (long long)n * 12288;
    float* yrow = a.y ? a.y + (long long)n * 12288 : nullptr;
    const float gscale = 2.0f / 12288.0f;
    float sq = 0.f;
    const int i0 = 8 * band;
    celeba_fwd_class<C, 0, 0>(sin, a.F6, a.b6, xrow, grow, yrow, i0, c4, slot, gscale, sq);
    celeba_fwd_class<C, 0, 1>(sin, a.F6, a.b6, xrow, grow, yrow, i0, c4, slot, gscale, sq);
    celeba_fwd_class<C, 1, 0>(sin, a.F6, a.b6, xrow, grow, yrow, i0, c4, slot, gscale, sq);
    celeba_fwd_class<C, 1, 1>(sin, a.F6, a.b6, xrow, grow, yrow, i0, c4, slot, gscale, sq);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) a.loss_part[(long long)n * 8 + band] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

template <int C>
__global__ __launch_bounds__(256) void celeba_tail_bwd_kernel(CelebaTailArgs a) {
    constexpr int G = C / 4, SLOTS = 256 / G, ROUNDS = 128 / SLOTS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sg = reinterpret_cast<float*>(smem);                // [11][68][4]: da6 with zero halo, 4th lane zero
    const int tid = threadIdx.x;
    const int n = blockIdx.x >> 3, band = blockIdx.x & 7;
    const int c4 = tid % G, slot = tid / G;
    const float* grow = a.g6 + (long long)n * 12288;
    const int i_lo = 8 * band - 1;                             // local row lr <-> image row i_lo + lr
    for (int i = tid; i < CB_G_ROWS * CB_G_COLS; i += 256) {
        const int lr = i / CB_G_COLS, lc = i % CB_G_COLS;
        const int ii = i_lo + lr, jj = lc - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ii >= 0 && ii < 64 && jj >= 0 && jj < 64) {
            const float* gp = grow + (ii * 64 + jj) * 3;
            v[0] = gp[0]; v[1] = gp[1]; v[2] = gp[2];
        }
        *reinterpret_cast<f32x4*>(sg + i * 4) = v;
    }
    __syncthreads();
    f32x4 acc[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int co = 0; co < 3; ++co) {
        f32x4 w[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) w[t] = *reinterpret_cast<const f32x4*>(a.F6 + (t * 3 + co) * C + c4 * 4);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int p = r * SLOTS + slot;
            const int ohl = p >> 5, ow = p & 31;
            const float* gp = sg + ((2 * ohl) * CB_G_COLS + 2 * ow) * 4 + co;
#pragma unroll
            for (int kh = 0; kh < 5; ++kh)
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) {
                    const float gv = gp[(kh * CB_G_COLS + kw) * 4];
                    const f32x4 ww = w[kh * 5 + kw];
                    acc[r][0] = __builtin_fmaf(gv, ww[0], acc[r][0]);
                    acc[r][1] = __builtin_fmaf(gv, ww[1], acc[r][1]);
                    acc[r][2] = __builtin_fmaf(gv, ww[2], acc[r][2]);
                    acc[r][3] = __builtin_fmaf(gv, ww[3], acc[r][3]);
                }
        }
    }
    float* hrow = a.h5 + (long long)n * (1024 * C);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int p = r * SLOTS + slot;
        const int oh = 4 * band + (p >> 5), ow = p & 31;
        *reinterpret_cast<f32x4*>(hrow + (oh * 32 + ow) * C + c4 * 4) = acc[r];
    }
}

__global__ __launch_bounds__(256) void celeba_loss_finish_kernel(const float* __restrict__ part, float* __restrict__ loss,
                                                                 int n_rows, int nparts, float inv_p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_rows) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long long)n * nparts + k];
    loss[n] = s * inv_p;
}

void launch_celeba_tail_fwd(const CelebaTailArgs& a, hipStream_t s) {
    const int lds = (CB_IN_ROWS * CB_IN_COLS * a.C + 4) * 4;
    if (a.C == 64) {
        static PerDeviceOnce attr;
        if (attr.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((celeba_tail_fwd_kernel<64>), dim3(a.n_rows * 8), dim3(256), lds, s, a);
    } else {
        static PerDeviceOnce attr;
        if (attr.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(celeba_tail_fwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((celeba_tail_fwd_kernel<128>), dim3(a.n_rows * 8), dim3(256), lds, s, a);
    }
}

void launch_celeba_tail_bwd(const CelebaTailArgs& a, hipStream_t s) {
    const int lds = CB_G_ROWS * CB_G_COLS * 4 * 4;
    if (a.C == 64) hipLaunchKernelGGL((celeba_tail_bwd_kernel<64>), dim3(a.n_rows * 8), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((celeba_tail_bwd_kernel<128>), dim3(a.n_rows * 8), dim3(256), lds, s, a);
}

void launch_celeba_loss_finish(const float* loss_part, float* loss, int n_rows, int nparts, hipStream_t s) {
    hipLaunchKernelGGL(celeba_loss_finish_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, loss_part, loss, n_rows,
                       nparts, 1.0f / 12288.0f);
}

}  // namespace dg
