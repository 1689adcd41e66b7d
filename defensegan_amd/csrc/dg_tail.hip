// MNIST / F-MNIST tail of the projection step, gfx950:
//   Generator.5 (Deconv2D C -> 1, 14x14 -> 28x28) + sigmoid            dataset_models.py:66-69
//   image_rec_loss = mean_pix (G(z) - x)^2                             gan.py:410-414
//   backward: dY = 2(y-x)/P, SigmoidGrad, Conv2D s2 (grad of the transpose), ReluGrad -> da3
// One 256-thread workgroup per latent row.  Cout = 1 makes this GEMV-shaped (1.7 % of the FLOPs), so it
// runs on the VALU out of LDS: the row's h3 map is staged once into a zero-bordered 16x16xC LDS
// image (no bounds tests in the tap loops), every thread owns 4 channels (its 25x4 filter taps live
// in registers), the 16 threads of a pixel reduce with wave shuffles, the loss with a block reduce.
// The backward overwrites h3 in place with da3 (same row, same workgroup).
#include "dg_kernels.h"

namespace dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HP = 16;           // padded h3 extent (14 + 2)
constexpr int GW = 32;           // padded g5 row pitch (31 used: i+1 for i in [-1, 29])
constexpr int GR = 31;           // padded g5 rows

template <int PI, int PJ>
__device__ __forceinline__ float tail_fwd_pixel(const float* sh3, const f32x4 (&w)[25], int C, int ti, int tj,
                                                int c4) {
    constexpr int NH = PI ? 3 : 2;
    constexpr int NW = PJ ? 3 : 2;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < NH; ++a) {
        constexpr int dummy = 0; (void)dummy;
        const int kh = PI ? 2 * a : 2 * a + 1;
        const int ohp = ti + (PI ? 2 : 1) - a;               // padded input row
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const int kw = PJ ? 2 * b : 2 * b + 1;
            const int owp = tj + (PJ ? 2 : 1) - b;
            const f32x4 h = *reinterpret_cast<const f32x4*>(sh3 + (ohp * HP + owp) * C + c4 * 4);
            const f32x4 ww = w[kh * 5 + kw];
            acc = __builtin_fmaf(h[0], ww[0], acc);
            acc = __builtin_fmaf(h[1], ww[1], acc);
            acc = __builtin_fmaf(h[2], ww[2], acc);
            acc = __builtin_fmaf(h[3], ww[3], acc);
        }
    }
    return acc;
}

template <int C>
__global__ __launch_bounds__(256) void mnist_tail_kernel(MnistTailArgs a) {
    constexpr int G = C / 4;             // channel groups (threads per pixel)
    constexpr int SLOTS = 256 / G;       // pixels / positions processed concurrently
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sh3 = reinterpret_cast<float*>(smem);                 // [16][16][C], zero border
    float* sg = sh3 + HP * HP * C;                               // [GR][GW] da5, zero border
    float* sred = sg + GR * GW;                                  // [4]

    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const int b = n / a.R;
    const int c4 = tid % G;
    const int slot = tid / G;
    float* hrow = a.h3 + (long long)n * (196 * C);

    // ---- stage: zero the border + sg, copy the 14x14xC interior ----------------------------------
    for (int i = tid; i < GR * GW; i += 256) sg[i] = 0.f;
    for (int i = tid; i < 60 * G; i += 256) {
        const int bp = i / G, g = i % G;
        int ph, pw;                                              // 60 border positions of the 16x16 frame
        if (bp < 16) { ph = 0; pw = bp; }
        else if (bp < 32) { ph = 15; pw = bp - 16; }
        else if (bp < 46) { ph = bp - 32 + 1; pw = 0; }
        else { ph = bp - 46 + 1; pw = 15; }
        *reinterpret_cast<f32x4*>(sh3 + (ph * HP + pw) * C + g * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < 196 * G; i += 256) {
        const int q = i / G, g = i % G;
        const int oh = q / 14, ow = q % 14;
        *reinterpret_cast<f32x4*>(sh3 + ((oh + 1) * HP + ow + 1) * C + g * 4) =
            *reinterpret_cast<const f32x4*>(hrow + q * C + g * 4);
    }
    // this thread's filter taps: F5[kh,kw,0,c4*4 .. +3]
    f32x4 w[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = *reinterpret_cast<const f32x4*>(a.F5 + t * C + c4 * 4);
    const float bias = a.b5[0];
    __syncthreads();

    // ---- forward: 4 parity classes x 196 pixels, SLOTS pixels per round ---------------------------
    const float* xrow = a.x + (long long)b * 784;
    float sq = 0.f;
    const float gscale = 2.0f / 784.0f;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int pi = cls >> 1, pj = cls & 1;
        for (int u0 = 0; u0 < 196; u0 += SLOTS) {
            const int u = u0 + slot;
            const bool valid = u < 196;
            const int uu = valid ? u : 0;
            const int ti = uu / 14, tj = uu % 14;
            float s;
            if (cls == 0) s = tail_fwd_pixel<0, 0>(sh3, w, C, ti, tj, c4);
            else if (cls == 1) s = tail_fwd_pixel<0, 1>(sh3, w, C, ti, tj, c4);
            else if (cls == 2) s = tail_fwd_pixel<1, 0>(sh3, w, C, ti, tj, c4);
            else s = tail_fwd_pixel<1, 1>(sh3, w, C, ti, tj, c4);
            // reduce over the G consecutive lanes of this pixel
#pragma unroll
            for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m, 64);
            if (valid && c4 == 0) {
                const int i = 2 * ti + pi, j = 2 * tj + pj;
                const float pre = s + bias;
                const float y = 1.0f / (1.0f + expf(-pre));
                const float d = y - xrow[i * 28 + j];
                sq = __builtin_fmaf(d, d, sq);
                sg[(i + 1) * GW + (j + 1)] = gscale * d * y * (1.0f - y);
                if (a.y) a.y[(long long)n * 784 + i * 28 + j] = y;
            }
        }
    }
    // ---- loss: block reduce (fixed order) ---------------------------------------------------------
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) a.loss[n] = ((sred[0] + sred[1]) + (sred[2] + sred[3])) * (1.0f / 784.0f);
    if (!a.do_backward) return;

    // ---- backward: da3[q,c] = [h3>0] * sum_{kh,kw} da5[2oh+kh-1, 2ow+kw-1] * F5[kh,kw,c] ---------
    for (int q0 = 0; q0 < 196; q0 += SLOTS) {
        const int q = q0 + slot;
        if (q >= 196) break;
        const int oh = q / 14, ow = q % 14;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* gp = sg + (2 * oh) * GW + 2 * ow;           // padded coords: row 2oh+kh, col 2ow+kw
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const float gv = gp[kh * GW + kw];
                const f32x4 ww = w[kh * 5 + kw];
                acc[0] = __builtin_fmaf(gv, ww[0], acc[0]);
                acc[1] = __builtin_fmaf(gv, ww[1], acc[1]);
                acc[2] = __builtin_fmaf(gv, ww[2], acc[2]);
                acc[3] = __builtin_fmaf(gv, ww[3], acc[3]);
            }
        const f32x4 h = *reinterpret_cast<const f32x4*>(sh3 + ((oh + 1) * HP + ow + 1) * C + c4 * 4);
        f32x4 o;
        o[0] = h[0] > 0.f ? acc[0] : 0.f;
        o[1] = h[1] > 0.f ? acc[1] : 0.f;
        o[2] = h[2] > 0.f ? acc[2] : 0.f;
        o[3] = h[3] > 0.f ? acc[3] : 0.f;
        *reinterpret_cast<f32x4*>(hrow + q * C + c4 * 4) = o;
    }
}

template <int C>
static void launch_tail_c(const MnistTailArgs& a, hipStream_t s) {
    constexpr int lds = (HP * HP * C + GR * GW + 4) * 4;
    static PerDeviceOnce attr;
    if (attr.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mnist_tail_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    hipLaunchKernelGGL((mnist_tail_kernel<C>), dim3(a.n_rows), dim3(256), lds, s, a);
}

void launch_mnist_tail(const MnistTailArgs& a, hipStream_t s) {
    if (a.C == 64) launch_tail_c<64>(a, s);
    else launch_tail_c<128>(a, s);
}

}  // namespace dg
