// Small kernels of the projection loop (gfx950): momentum update, restart selection, latent init.
#include "dg_kernels.h"

namespace dg {

// ---- ApplyMomentum (tf.train.MomentumOptimizer, non-Nesterov; gan.py:389-391, 416-417) ------------
//   g = sum_s part[n][s][d]  (split-K partials of dz = da1 . W^T, fixed summation order)
//   m <- momentum*m + g ;  z <- z - lr*m
// One thread owns 4 consecutive latent components of one row; the partials of up to 8 slices are fetched before the first add
// (the adds keep the slice order: bit-identical for any nsplit grouping).
__global__ __launch_bounds__(256) void momentum_update_kernel(float* __restrict__ z, float* __restrict__ m,
                                                              const float* __restrict__ part, int nsplit,
                                                              long long n_quads, int latent, float lr,
                                                              float momentum, float* __restrict__ dz_out) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_quads) return;
    const unsigned qrow = (unsigned)latent >> 2;
    const long long n = q / qrow;
    const int d = (int)(q - n * qrow) << 2;
    const float* p = part + n * (long long)nsplit * latent + d;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nsplit; s0 += 8) {
        float4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s0 + s < nsplit) v[s] = *reinterpret_cast<const float4*>(p + (long long)(s0 + s) * latent);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s0 + s < nsplit) { g.x += v[s].x; g.y += v[s].y; g.z += v[s].z; g.w += v[s].w; }
    }
    const long long i = n * latent + d;
    if (dz_out) { dz_out[i] = g.x; dz_out[i + 1] = g.y; dz_out[i + 2] = g.z; dz_out[i + 3] = g.w; return; }   // caller's pointer: no alignment assumed
    const float4 m0 = *reinterpret_cast<const float4*>(m + i);
    const float4 z0 = *reinterpret_cast<const float4*>(z + i);
    float4 mm, zz;
    mm.x = momentum * m0.x + g.x; mm.y = momentum * m0.y + g.y; mm.z = momentum * m0.z + g.z; mm.w = momentum * m0.w + g.w;
    zz.x = z0.x - lr * mm.x; zz.y = z0.y - lr * mm.y; zz.z = z0.z - lr * mm.z; zz.w = z0.w - lr * mm.w;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(z + i) = zz;
}

void launch_momentum_update(float* z, float* m, const float* part, int nsplit, int64_t n_rows, int latent,
                            float lr, float momentum, float* dz_out, hipStream_t s) {
    const long long n_quads = (long long)n_rows * (latent >> 2);          // latent % 64 == 0 (dg_create)
    if (n_quads == 0) return;
    const unsigned grid = (unsigned)((n_quads + 255) / 256);
    hipLaunchKernelGGL(momentum_update_kernel, dim3(grid), dim3(256), 0, s, z, m, part, nsplit, n_quads, latent, lr,
                       momentum, dz_out);
}

// ---- selection: first argmin over the R restarts of each image, then gather (gan.py:438-449) ------
// One 256-thread workgroup per image: wave 0 finds the first minimum (wave shuffles), everybody copies the selected row
// (b128 when both rows are 16-byte aligned: P % 4 == 0 and aligned bases; dwords otherwise).
__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ loss, const float* __restrict__ y,
                                                     int R, int P, float* __restrict__ out_rec,
                                                     int32_t* __restrict__ out_idx, int vec4) {
    __shared__ int s_bi;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {
        float best = __builtin_inff();
        int bi = 0x7fffffff;
        for (int r = lane; r < R; r += 64) {
            const float v = loss[(long long)b * R + r];
            // strict "<" keeps the first minimum; a NaN loss never wins (all-NaN rows select restart 0)
            if (v < best || (v == best && r < bi)) { best = v; bi = r; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) bi = 0;
        if (lane == 0) {
            s_bi = bi;
            if (out_idx) out_idx[b] = bi;
        }
    }
    __syncthreads();
    const float* src = y + ((long long)b * R + s_bi) * P;
    float* dst = out_rec + (long long)b * P;
    if (vec4) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = tid; i < (P >> 2); i += 256) d4[i] = s4[i];
    } else {
        for (int i = tid; i < P; i += 256) dst[i] = src[i];
    }
}

void launch_select(const float* loss, const float* y, int B, int R, int P, float* out_rec, int32_t* out_idx,
                   hipStream_t s) {
    const int vec4 = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out_rec)) % 16 == 0);
    hipLaunchKernelGGL(select_kernel, dim3(B), dim3(256), 0, s, loss, y, R, P, out_rec, out_idx, vec4);
}

// ---- latent init: z ~ N(0, std^2), Philox4x32-10 + Box-Muller ------------------------------------
// counter = (global row lo, global row hi, column/4, 0), key = (seed lo, seed hi): the draw of a row
// depends only on (seed, global row), never on batching or on the GPU count.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void init_latents_kernel(float* __restrict__ z, long long n_rows, int latent,
                                                           unsigned long long seed, long long first_row,
                                                           float std) {
    const int q = latent / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * q) return;
    const long long row = i / q;
    const int cq = (int)(i - row * q);
    const unsigned long long grow = (unsigned long long)(first_row + row);
    uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)cq, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // uniforms in (0,1]: (x + 1) * 2^-32
    const float u0 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f;
    const float u1 = ((float)c[1] + 1.0f) * 2.3283064365386963e-10f;
    const float u2 = ((float)c[2] + 1.0f) * 2.3283064365386963e-10f;
    const float u3 = ((float)c[3] + 1.0f) * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.0f * logf(u0 < 1.0f ? u0 : 1.0f)) * std;
    const float r1 = sqrtf(-2.0f * logf(u2 < 1.0f ? u2 : 1.0f)) * std;
    const float t0 = 6.283185307179586f * u1;
    const float t1 = 6.283185307179586f * u3;
    float4 o;
    o.x = r0 * cosf(t0);
    o.y = r0 * sinf(t0);
    o.z = r1 * cosf(t1);
    o.w = r1 * sinf(t1);
    *reinterpret_cast<float4*>(z + row * latent + cq * 4) = o;
}

void launch_init_latents(float* z, int64_t n_rows, int latent, uint64_t seed, int64_t first_row, float std,
                         hipStream_t s) {
    const long long total = (long long)n_rows * (latent / 4);
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(init_latents_kernel, dim3(grid), dim3(256), 0, s, z, (long long)n_rows, latent,
                       (unsigned long long)seed, (long long)first_row, std);
}

// ---- CelebA loss: sum of the per-(band, wave) partial squared errors of a latent row, / P (gan.py:410-414) ----
// part [n_rows][nparts bands][4 waves]; fixed order ((w0 + w1) + (w2 + w3)) per band, bands in sequence.
__global__ __launch_bounds__(256) void celeba_loss_finish_kernel(const float* __restrict__ part, float* __restrict__ loss,
                                                                 int n_rows, int nparts, float inv_p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_rows) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(part + ((long long)n * nparts + k) * 4);
        s += (q.x + q.y) + (q.z + q.w);
    }
    loss[n] = s * inv_p;
}

void launch_celeba_loss_finish(const float* loss_part, float* loss, int n_rows, int nparts, int P, hipStream_t s) {
    hipLaunchKernelGGL(celeba_loss_finish_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, loss_part, loss, n_rows,
                       nparts, 1.0f / (float)P);
}

// ---- fragment order -> plain rows (dg_types.h "fragment order"); one thread per 4 floats ---------------------------------
__global__ __launch_bounds__(256) void unfrag_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n_quads,
                                                     long long row_floats) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_quads) return;
    const long long qrow = row_floats >> 2;
    const long long n = q / qrow;
    const long long f = (q - n * qrow) << 2;
    const long long blk = ((n >> 5) * row_floats + (f & ~7LL)) * 32;          // floats before the KB of (row block, f / 8)
    const float4 v = *reinterpret_cast<const float4*>(src + blk + ((((f & 7) >> 2) * 32 + (n & 31)) << 2));
    *reinterpret_cast<float4*>(dst + n * row_floats + f) = v;
}

void launch_unfrag(const float* src, float* dst, int64_t n_rows, int64_t row_floats, hipStream_t s) {
    const long long n_quads = (long long)n_rows * (row_floats >> 2);
    if (n_quads <= 0) return;
    hipLaunchKernelGGL(unfrag_kernel, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, s, src, dst, n_quads, (long long)row_floats);
}

}  // namespace dg
