// The step AFTER the projection: the classifier forward of the reference's cleverhans-style
// MLP (utils/network_builder.py:129-331: Conv2D = tf.nn.conv2d(x, kernels[kh,kw,cin,cout], strides, padding) + b, ReLU,
// Flatten, Linear = x @ W[in,out] + b, Softmax, Dropout = identity at inference) and the per-batch reduction of
// model_eval_gan (utils/gan_defense.py:91-179 with diff_op of blackbox.py:569-572): preds = argmax, correct count,
// per-image mean squared difference between the classifier's input and the original image.
//
// This path is <0.3 % of a defended evaluation (a few MFLOP per image against 132 GFLOP of projection), so the kernels are
// plain direct loops with coalesced channel-fastest indexing; they exist so that configuration 5 runs end to end on the
// device without a host round trip per batch.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/defensegan_hip.h"

extern "C" __attribute__((visibility("hidden"))) void dg_set_error_message(const char* msg);   // dg_engine.cpp (dg_last_error storage), library-internal

namespace {

enum LayerKind { L_CONV = 0, L_RELU = 1, L_LINEAR = 2, L_FLATTEN = 3, L_SOFTMAX = 4, L_DROPOUT = 5 };

struct ClfLayer {
    int kind = 0;
    // conv
    int kh = 0, kw = 0, sh = 1, sw = 1, same = 0, cin = 0, cout = 0, pad_t = 0, pad_l = 0;
    // shapes (per image)
    int ih = 0, iw = 0, ic = 0, oh = 0, ow = 0, oc = 0;
    bool fused_relu = false;     // the ReLU that follows is applied in this layer's kernel
    bool skip = false;           // a ReLU folded into its predecessor / identity layers
    float* W = nullptr;
    float* b = nullptr;
    bool have_w = false;
};

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    dg_set_error_message(buf);
    return code;
}

#define CLF_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(DG_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

// y[b, oh, ow, co] = bias[co] + sum_{kh,kw,ci} x[b, oh*sh + kh - pad_t, ow*sw + kw - pad_l, ci] * K[kh, kw, ci, co]
// (cross-correlation, tf.nn.conv2d; out-of-range taps are the zero padding).  One thread per output, co fastest.
__global__ __launch_bounds__(256) void clf_conv2d_kernel(const float* __restrict__ x, const float* __restrict__ K,
                                                          const float* __restrict__ bias, float* __restrict__ y, long long total,
                                                          int ih, int iw, int ic, int oh, int ow, int oc, int kh, int kw, int sh,
                                                          int sw, int pad_t, int pad_l, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % oc);
    long long r = i / oc;
    const int xo = (int)(r % ow);
    r /= ow;
    const int yo = (int)(r % oh);
    const long long b = r / oh;
    const float* xb = x + b * (long long)ih * iw * ic;
    float acc = bias[co];
    for (int a = 0; a < kh; ++a) {
        const int yi = yo * sh + a - pad_t;
        if (yi < 0 || yi >= ih) continue;
        for (int c = 0; c < kw; ++c) {
            const int xi = xo * sw + c - pad_l;
            if (xi < 0 || xi >= iw) continue;
            const float* xp = xb + ((long long)yi * iw + xi) * ic;
            const float* kp = K + ((long long)(a * kw + c) * ic) * oc + co;
            for (int ci = 0; ci < ic; ++ci) acc = __builtin_fmaf(xp[ci], kp[(long long)ci * oc], acc);
        }
    }
    y[i] = relu ? (acc > 0.f ? acc : 0.f) : acc;
}

// y[b, o] = bias[o] + sum_i x[b, i] * W[i, o]; one thread per output, o fastest
__global__ __launch_bounds__(256) void clf_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ y, long long total,
                                                          int n_in, int n_out, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int o = (int)(i % n_out);
    const long long b = i / n_out;
    const float* xb = x + b * n_in;
    float acc = bias[o];
    for (int k = 0; k < n_in; ++k) acc = __builtin_fmaf(xb[k], W[(long long)k * n_out + o], acc);
    y[i] = relu ? (acc > 0.f ? acc : 0.f) : acc;
}

__global__ __launch_bounds__(256) void clf_relu_kernel(const float* __restrict__ x, float* __restrict__ y, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) y[i] = x[i] > 0.f ? x[i] : 0.f;
}

// tf.nn.softmax over the last axis; one thread per row (the class count is small)
__global__ __launch_bounds__(64) void clf_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int n) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* r = x + (long long)b * n;
    float m = r[0];
    for (int k = 1; k < n; ++k) m = r[k] > m ? r[k] : m;
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(r[k] - m);
    const float inv = 1.0f / s;
    for (int k = 0; k < n; ++k) y[(long long)b * n + k] = expf(r[k] - m) * inv;
}

// One workgroup per image: preds[b] = first argmax of scores[b, :], correct += (preds == labels), diffs[b] =
// mean((rec - orig)^2) (fixed-order block reduction).
__global__ __launch_bounds__(256) void clf_eval_kernel(const float* __restrict__ scores, int ncls, const float* __restrict__ rec,
                                                        const float* __restrict__ orig, int P, const int32_t* __restrict__ labels,
                                                        int32_t* __restrict__ preds, float* __restrict__ diffs,
                                                        int32_t* __restrict__ n_correct) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sq = 0.f;
    if (rec && orig && diffs) {
        const float* r = rec + (long long)b * P;
        const float* o = orig + (long long)b * P;
        for (int i = tid; i < P; i += 256) {
            const float d = o[i] - r[i];
            sq = __builtin_fmaf(d, d, sq);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        if (lane == 0) red[wave] = sq;
    }
    __syncthreads();
    if (tid == 0) {
        if (rec && orig && diffs) diffs[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)P;
        const float* s = scores + (long long)b * ncls;
        int best = 0;
        for (int k = 1; k < ncls; ++k)
            if (s[k] > s[best]) best = k;
        if (preds) preds[b] = best;
        if (labels && n_correct && labels[b] == best) atomicAdd(n_correct, 1);
    }
}

// ---- backward to the input (FGSM) ----------------------------------------------------------------------
// g = dCE/dlogits = softmax(logits) - onehot(label); label = the model's own first argmax when labels == nullptr (what
// cleverhans' FastGradientMethod does when no y is given, to avoid label leaking).
__global__ __launch_bounds__(64) void clf_ce_grad_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                          float* __restrict__ g, int B, int n) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* r = logits + (long long)b * n;
    float m = r[0];
    int best = 0;
    for (int k = 1; k < n; ++k)
        if (r[k] > m) { m = r[k]; best = k; }
    const int y = labels ? labels[b] : best;
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(r[k] - m);
    const float inv = 1.0f / s;
    for (int k = 0; k < n; ++k) g[(long long)b * n + k] = expf(r[k] - m) * inv - (k == y ? 1.0f : 0.0f);
}

// dx[b, i] = sum_o gm[b, o] * W[i, o], gm = g masked by the layer's own ReLU (out > 0) when it was fused
__global__ __launch_bounds__(256) void clf_linear_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                              const float* __restrict__ W, float* __restrict__ dx, long long total,
                                                              int n_in, int n_out, int relu) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int i = (int)(t % n_in);
    const long long b = t / n_in;
    const float* gb = g + b * n_out;
    const float* ob = out + b * n_out;
    const float* wr = W + (long long)i * n_out;
    float acc = 0.f;
    for (int o = 0; o < n_out; ++o) {
        const float gv = (relu && !(ob[o] > 0.f)) ? 0.f : gb[o];
        acc = __builtin_fmaf(gv, wr[o], acc);
    }
    dx[t] = acc;
}

// dx[b, yi, xi, ci] = sum over (a, c) with yo*sh + a - pad_t == yi, xo*sw + c - pad_l == xi, and co of gm[b,yo,xo,co] * K[a,c,ci,co]
__global__ __launch_bounds__(256) void clf_conv2d_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                              const float* __restrict__ K, float* __restrict__ dx, long long total,
                                                              int ih, int iw, int ic, int oh, int ow, int oc, int kh, int kw, int sh,
                                                              int sw, int pad_t, int pad_l, int relu) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int ci = (int)(t % ic);
    long long r = t / ic;
    const int xi = (int)(r % iw);
    r /= iw;
    const int yi = (int)(r % ih);
    const long long b = r / ih;
    float acc = 0.f;
    for (int a = 0; a < kh; ++a) {
        const int ty = yi + pad_t - a;
        if (ty < 0 || ty % sh) continue;
        const int yo = ty / sh;
        if (yo >= oh) continue;
        for (int c = 0; c < kw; ++c) {
            const int tx = xi + pad_l - c;
            if (tx < 0 || tx % sw) continue;
            const int xo = tx / sw;
            if (xo >= ow) continue;
            const long long obase = ((b * oh + yo) * ow + xo) * oc;
            const float* kp = K + ((long long)(a * kw + c) * ic + ci) * oc;
            for (int co = 0; co < oc; ++co) {
                const float gv = (relu && !(out[obase + co] > 0.f)) ? 0.f : g[obase + co];
                acc = __builtin_fmaf(gv, kp[co], acc);
            }
        }
    }
    dx[t] = acc;
}

__global__ __launch_bounds__(256) void clf_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                            float* __restrict__ dx, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) dx[i] = out[i] > 0.f ? g[i] : 0.f;
}

// x_adv = clip(x + eps * sign(grad), lo, hi)   (cleverhans fgm with ord = inf; sign(0) = 0 as tf.sign)
__global__ __launch_bounds__(256) void clf_fgsm_kernel(const float* __restrict__ x, const float* __restrict__ grad,
                                                        float* __restrict__ xadv, long long total, float eps, float lo, float hi) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float gsign = grad[i] > 0.f ? 1.f : (grad[i] < 0.f ? -1.f : 0.f);
    float v = x[i] + eps * gsign;
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    xadv[i] = v;
}

}  // namespace

struct dg_clf {
    int device = 0;
    int in_h = 0, in_w = 0, in_c = 0;
    std::vector<ClfLayer> layers;
    bool planned = false;
    int cur_h = 0, cur_w = 0, cur_c = 0;      // running shape while layers are added (flat: h = w = 1, c = width)
    bool flat = false;
    float* buf[2] = {nullptr, nullptr};
    size_t buf_floats = 0;
    float* scores = nullptr;                  // [B, n_out] scratch of dg_eval_batch
    std::vector<float*> acts;                 // per-layer outputs kept by the gradient path
    std::vector<size_t> acts_floats;
    float* gbuf[2] = {nullptr, nullptr};      // gradient ping-pong
    size_t gbuf_floats = 0;
    size_t scores_floats = 0;
    int n_out = 0;
};

extern "C" {

int dg_clf_create(int device, int in_h, int in_w, int in_c, dg_clf** out) {
    if (!out || in_h <= 0 || in_w <= 0 || in_c <= 0) return fail(DG_E_INVALID, "dg_clf_create: bad input shape");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(DG_E_HIP, "no HIP device");
    if (device < 0 || device >= n) return fail(DG_E_INVALID, "device %d out of range", device);
    dg_clf* h = new dg_clf();
    h->device = device;
    h->in_h = h->cur_h = in_h;
    h->in_w = h->cur_w = in_w;
    h->in_c = h->cur_c = in_c;
    *out = h;
    return DG_OK;
}

int dg_clf_destroy(dg_clf* h) {
    if (!h) return DG_OK;
    (void)hipSetDevice(h->device);
    for (auto& l : h->layers) {
        if (l.W) (void)hipFree(l.W);
        if (l.b) (void)hipFree(l.b);
    }
    for (float* p : h->buf)
        if (p) (void)hipFree(p);
    if (h->scores) (void)hipFree(h->scores);
    for (float* p : h->acts)
        if (p) (void)hipFree(p);
    for (float* p : h->gbuf)
        if (p) (void)hipFree(p);
    delete h;
    return DG_OK;
}

int dg_clf_add_layer(dg_clf* h, int kind, int p0, int p1, int p2, int p3, int p4, int p5) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    CLF_TRY(hipSetDevice(h->device));
    ClfLayer l;
    l.kind = kind;
    l.ih = h->cur_h; l.iw = h->cur_w; l.ic = h->cur_c;
    switch (kind) {
        case L_CONV: {
            if (h->flat) return fail(DG_E_INVALID, "Conv2D after Flatten");
            l.cout = p0; l.kh = p1; l.kw = p2; l.sh = p3; l.sw = p4; l.same = p5; l.cin = h->cur_c;
            if (l.cout <= 0 || l.kh <= 0 || l.kw <= 0 || l.sh <= 0 || l.sw <= 0) return fail(DG_E_INVALID, "bad Conv2D parameters");
            if (l.same) {       // TF SAME: out = ceil(in / stride), pad_before = pad_total / 2
                l.oh = (l.ih + l.sh - 1) / l.sh;
                l.ow = (l.iw + l.sw - 1) / l.sw;
                const int pt = std::max((l.oh - 1) * l.sh + l.kh - l.ih, 0), pl = std::max((l.ow - 1) * l.sw + l.kw - l.iw, 0);
                l.pad_t = pt / 2;
                l.pad_l = pl / 2;
            } else {            // VALID
                if (l.ih < l.kh || l.iw < l.kw) return fail(DG_E_INVALID, "VALID Conv2D kernel larger than its input");
                l.oh = (l.ih - l.kh) / l.sh + 1;
                l.ow = (l.iw - l.kw) / l.sw + 1;
            }
            l.oc = l.cout;
            CLF_TRY(hipMalloc(&l.W, (size_t)l.kh * l.kw * l.cin * l.cout * sizeof(float)));
            CLF_TRY(hipMalloc(&l.b, (size_t)l.cout * sizeof(float)));
            break;
        }
        case L_LINEAR: {
            if (!h->flat) return fail(DG_E_INVALID, "Linear needs a Flatten before it");
            l.cin = h->cur_c; l.cout = p0;
            if (l.cout <= 0) return fail(DG_E_INVALID, "bad Linear width");
            l.oh = l.ow = 1; l.oc = l.cout;
            CLF_TRY(hipMalloc(&l.W, (size_t)l.cin * l.cout * sizeof(float)));
            CLF_TRY(hipMalloc(&l.b, (size_t)l.cout * sizeof(float)));
            break;
        }
        case L_FLATTEN:
            l.oh = l.ow = 1; l.oc = l.ih * l.iw * l.ic; l.skip = true;       // NHWC row-major: a reshape, no data movement
            h->flat = true;
            break;
        case L_RELU: case L_SOFTMAX: case L_DROPOUT:
            l.oh = l.ih; l.ow = l.iw; l.oc = l.ic;
            if (kind == L_DROPOUT) l.skip = true;                             // K.learning_phase() == 0 at evaluation
            if (kind == L_RELU && !h->layers.empty()) {
                // fold the ReLU into the producing Conv2D / Linear (skipping identity layers in between)
                for (int j = (int)h->layers.size() - 1; j >= 0; --j) {
                    ClfLayer& p = h->layers[j];
                    if (p.skip) continue;
                    if ((p.kind == L_CONV || p.kind == L_LINEAR) && !p.fused_relu) { p.fused_relu = true; l.skip = true; }
                    break;
                }
            }
            break;
        default:
            return fail(DG_E_INVALID, "unknown layer kind %d", kind);
    }
    h->cur_h = l.oh; h->cur_w = l.ow; h->cur_c = l.oc;
    h->layers.push_back(l);
    return (int)h->layers.size() - 1;
}

int dg_clf_output_width(dg_clf* h) { return h ? h->cur_h * h->cur_w * h->cur_c : 0; }

int dg_clf_set_weights(dg_clf* h, int layer, const float* W, const int64_t* wshape, int wndim, const float* b, int64_t blen,
                       int is_device) {
    if (!h || !W || !b || !wshape) return fail(DG_E_INVALID, "null argument");
    if (layer < 0 || layer >= (int)h->layers.size()) return fail(DG_E_INVALID, "layer %d out of range", layer);
    ClfLayer& l = h->layers[layer];
    CLF_TRY(hipSetDevice(h->device));
    size_t n = 0;
    if (l.kind == L_CONV) {
        if (wndim != 4 || wshape[0] != l.kh || wshape[1] != l.kw || wshape[2] != l.cin || wshape[3] != l.cout)
            return fail(DG_E_INVALID, "layer %d: Conv2D kernels must be [%d,%d,%d,%d]", layer, l.kh, l.kw, l.cin, l.cout);
        n = (size_t)l.kh * l.kw * l.cin * l.cout;
    } else if (l.kind == L_LINEAR) {
        if (wndim != 2 || wshape[0] != l.cin || wshape[1] != l.cout)
            return fail(DG_E_INVALID, "layer %d: Linear W must be [%d,%d]", layer, l.cin, l.cout);
        n = (size_t)l.cin * l.cout;
    } else {
        return fail(DG_E_INVALID, "layer %d has no parameters", layer);
    }
    if (blen != l.cout) return fail(DG_E_INVALID, "layer %d: bias must have %d entries", layer, l.cout);
    const hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    CLF_TRY(hipMemcpy(l.W, W, n * sizeof(float), kind));
    CLF_TRY(hipMemcpy(l.b, b, (size_t)l.cout * sizeof(float), kind));
    l.have_w = true;
    return DG_OK;
}

static int clf_run(dg_clf* h, const float* x, int B, float* logits, float* probs, hipStream_t s, bool keep = false) {
    for (size_t j = 0; j < h->layers.size(); ++j)
        if ((h->layers[j].kind == L_CONV || h->layers[j].kind == L_LINEAR) && !h->layers[j].have_w)
            return fail(DG_E_STATE, "classifier layer %d has no weights", (int)j);
    size_t need = 0;
    for (const auto& l : h->layers) need = std::max(need, (size_t)B * l.oh * l.ow * l.oc);
    if (need > h->buf_floats) {
        for (float*& p : h->buf) {
            if (p) (void)hipFree(p);
            p = nullptr;
            CLF_TRY(hipMalloc(&p, need * sizeof(float)));
        }
        h->buf_floats = need;
    }
    // the logits are the output of the last parameterised layer ("logits" = layers[-2] when the model ends in Softmax,
    // network_builder.py:148-153)
    int last_param = -1, softmax_at = -1;
    for (int j = 0; j < (int)h->layers.size(); ++j) {
        if (h->layers[j].kind == L_SOFTMAX) softmax_at = j;
        if (!h->layers[j].skip && h->layers[j].kind != L_SOFTMAX) last_param = j;
    }
    if (last_param < 0) return fail(DG_E_STATE, "classifier has no layers");
    const float* cur = x;
    int which = 0;
    for (int j = 0; j < (int)h->layers.size(); ++j) {
        const ClfLayer& l = h->layers[j];
        if (l.skip || l.kind == L_SOFTMAX) continue;
        const long long total = (long long)B * l.oh * l.ow * l.oc;
        float* out = (j == last_param && logits) ? logits : h->buf[which];
        if (keep) {                                      // the gradient path needs every layer's output afterwards
            if (h->acts.size() < h->layers.size()) { h->acts.resize(h->layers.size(), nullptr); h->acts_floats.resize(h->layers.size(), 0); }
            if ((size_t)total > h->acts_floats[j]) {
                if (h->acts[j]) (void)hipFree(h->acts[j]);
                h->acts[j] = nullptr;
                CLF_TRY(hipMalloc(&h->acts[j], (size_t)total * sizeof(float)));
                h->acts_floats[j] = (size_t)total;
            }
            out = h->acts[j];
        }
        const unsigned grid = (unsigned)((total + 255) / 256);
        if (l.kind == L_CONV)
            hipLaunchKernelGGL(clf_conv2d_kernel, dim3(grid), dim3(256), 0, s, cur, l.W, l.b, out, total, l.ih, l.iw, l.ic, l.oh,
                               l.ow, l.oc, l.kh, l.kw, l.sh, l.sw, l.pad_t, l.pad_l, l.fused_relu ? 1 : 0);
        else if (l.kind == L_LINEAR)
            hipLaunchKernelGGL(clf_linear_kernel, dim3(grid), dim3(256), 0, s, cur, l.W, l.b, out, total, l.cin, l.cout,
                               l.fused_relu ? 1 : 0);
        else
            hipLaunchKernelGGL(clf_relu_kernel, dim3(grid), dim3(256), 0, s, cur, out, total);
        cur = out;
        which ^= 1;
    }
    h->n_out = h->layers[last_param].oh * h->layers[last_param].ow * h->layers[last_param].oc;
    if (probs) {
        if (softmax_at >= 0)
            hipLaunchKernelGGL(clf_softmax_kernel, dim3((B + 63) / 64), dim3(64), 0, s, cur, probs, B, h->n_out);
        else
            CLF_TRY(hipMemcpyAsync(probs, cur, (size_t)B * h->n_out * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    CLF_TRY(hipGetLastError());
    return DG_OK;
}

int dg_clf_forward(dg_clf* h, const float* x, int B, float* logits, float* probs, void* stream) {
    if (!h || !x || B <= 0) return fail(DG_E_INVALID, "dg_clf_forward: bad argument");
    CLF_TRY(hipSetDevice(h->device));
    return clf_run(h, x, B, logits, probs, (hipStream_t)stream);
}

int dg_eval_batch(dg_clf* h, const float* rec, const float* orig, const int32_t* labels, int B, int32_t* preds, float* diffs,
                  int32_t* n_correct, void* stream) {
    if (!h || !rec || B <= 0) return fail(DG_E_INVALID, "dg_eval_batch: bad argument");
    CLF_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int ncls = dg_clf_output_width(h);
    if ((size_t)B * ncls > h->scores_floats) {
        if (h->scores) (void)hipFree(h->scores);
        h->scores = nullptr;
        CLF_TRY(hipMalloc(&h->scores, (size_t)B * ncls * sizeof(float)));
        h->scores_floats = (size_t)B * ncls;
    }
    float* scratch = h->scores;
    int rc = clf_run(h, rec, B, scratch, nullptr, s);      // argmax(probs) == argmax(logits): softmax is monotone
    if (rc) return rc;
    const int P = h->in_h * h->in_w * h->in_c;
    hipLaunchKernelGGL(clf_eval_kernel, dim3(B), dim3(256), 0, s, scratch, ncls, rec, orig, P, labels, preds, diffs, n_correct);
    CLF_TRY(hipGetLastError());
    return DG_OK;
}

// d(sum_b CE(softmax(logits_b), y_b))/dx for x [B,H,W,C]; y = labels or, when NULL, the model's own prediction.
static int clf_input_gradient(dg_clf* h, const float* x, const int32_t* labels, int B, float** grad_out, hipStream_t s) {
    int rc = clf_run(h, x, B, nullptr, nullptr, s, /*keep=*/true);
    if (rc) return rc;
    size_t need = (size_t)B * h->in_h * h->in_w * h->in_c;
    for (const auto& l : h->layers) need = std::max(need, (size_t)B * l.oh * l.ow * l.oc);
    if (need > h->gbuf_floats) {
        for (float*& p : h->gbuf) {
            if (p) (void)hipFree(p);
            p = nullptr;
            CLF_TRY(hipMalloc(&p, need * sizeof(float)));
        }
        h->gbuf_floats = need;
    }
    int last = -1;
    for (int j = 0; j < (int)h->layers.size(); ++j)
        if (!h->layers[j].skip && h->layers[j].kind != L_SOFTMAX) last = j;
    const int n = h->layers[last].oh * h->layers[last].ow * h->layers[last].oc;
    int which = 0;
    float* g = h->gbuf[which];
    hipLaunchKernelGGL(clf_ce_grad_kernel, dim3((B + 63) / 64), dim3(64), 0, s, h->acts[last], labels, g, B, n);
    for (int j = last; j >= 0; --j) {
        const ClfLayer& l = h->layers[j];
        if (l.skip || l.kind == L_SOFTMAX) continue;
        const long long total = (long long)B * l.ih * l.iw * l.ic;
        float* dx = h->gbuf[which ^ 1];
        const unsigned grid = (unsigned)((total + 255) / 256);
        if (l.kind == L_LINEAR)
            hipLaunchKernelGGL(clf_linear_bwd_kernel, dim3(grid), dim3(256), 0, s, g, h->acts[j], l.W, dx, total, l.cin, l.cout,
                               l.fused_relu ? 1 : 0);
        else if (l.kind == L_CONV)
            hipLaunchKernelGGL(clf_conv2d_bwd_kernel, dim3(grid), dim3(256), 0, s, g, h->acts[j], l.W, dx, total, l.ih, l.iw, l.ic,
                               l.oh, l.ow, l.oc, l.kh, l.kw, l.sh, l.sw, l.pad_t, l.pad_l, l.fused_relu ? 1 : 0);
        else
            hipLaunchKernelGGL(clf_relu_bwd_kernel, dim3(grid), dim3(256), 0, s, g, h->acts[j], dx, total);
        which ^= 1;
        g = dx;
    }
    CLF_TRY(hipGetLastError());
    *grad_out = g;
    return DG_OK;
}

int dg_clf_input_gradient(dg_clf* h, const float* x, const int32_t* labels, int B, float* grad, void* stream) {
    if (!h || !x || !grad || B <= 0) return fail(DG_E_INVALID, "dg_clf_input_gradient: bad argument");
    CLF_TRY(hipSetDevice(h->device));
    float* g = nullptr;
    int rc = clf_input_gradient(h, x, labels, B, &g, (hipStream_t)stream);
    if (rc) return rc;
    CLF_TRY(hipMemcpyAsync(grad, g, (size_t)B * h->in_h * h->in_w * h->in_c * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DG_OK;
}

int dg_fgsm(dg_clf* h, const float* x, const int32_t* labels, int B, float eps, float clip_min, float clip_max, float* x_adv,
            void* stream) {
    if (!h || !x || !x_adv || B <= 0) return fail(DG_E_INVALID, "dg_fgsm: bad argument");
    CLF_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    float* g = nullptr;
    int rc = clf_input_gradient(h, x, labels, B, &g, s);
    if (rc) return rc;
    const long long total = (long long)B * h->in_h * h->in_w * h->in_c;
    hipLaunchKernelGGL(clf_fgsm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, g, x_adv, total, eps, clip_min, clip_max);
    CLF_TRY(hipGetLastError());
    return DG_OK;
}

}  // extern "C"
