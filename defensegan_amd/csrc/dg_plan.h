// Host-side layer plans: per-output-position tap tables for the gathered implicit GEMM.
// Pure C++ (no HIP) so the tables can be unit-tested on a CPU-only box (tests/test_plan.py).
//
// Index map of tf.nn.conv2d_transpose(5x5, stride 2, SAME) as used by tflib Deconv2D
// (/root/reference/tflib/ops/deconv2d.py:100-117): out index i = 2*o + k - 1.
#pragma once
#include <string>
#include <vector>
#include "dg_types.h"

namespace dg {

struct LayerPlan {
    std::string name;
    std::vector<PosEntry> pos;     // sorted by descending tap_count (longest tiles dispatch first)
    std::vector<TapEntry> taps;
    long long a_rowstride = 0;     // floats per latent row of the input buffer
    long long out_rowstride = 0;   // floats per latent row of the output buffer
    int w_rowstride = 0;           // floats between consecutive output columns in the weight slab
    int kch = 0;                   // K extent per tap
    int ncols = 0;                 // output columns per position
    int bn = 0;                    // column tile
    long long macs_per_row = 0;    // multiply-accumulates per latent row (valid taps only)
};

// Linear forward: out[n, f] = sum_d z[n,d] * Wt[f,d]           (Wt = W^T, [features][latent])
LayerPlan plan_linear_fwd(int latent, int features, int bn);
// Linear backward: part[n, s, d] = sum_{f in split s} da[n,f] * W[d,f]   (W native [latent][features])
LayerPlan plan_linear_bwd(int latent, int features, int nsplit, int bn);
// Deconv forward: input grid h_in x h_in (row pitch in_pitch positions), outputs e_out x e_out
// (row pitch out_pitch); filters F[kh,kw,cout,cin] (reference layout).
LayerPlan plan_deconv_fwd(int h_in, int in_pitch, int e_out, int out_pitch, int cin, int cout, int bn);
// Deconv backward-to-input: dh[n,oh,ow,ci] = sum da[n,2oh+kh-1,2ow+kw-1,co] * Ft[kh,kw,ci,co];
// da valid extent e_out (pitch a_pitch), dh grid h_in (pitch out_pitch).
LayerPlan plan_deconv_bwd(int h_in, int out_pitch, int e_out, int a_pitch, int cin, int cout, int bn);
// Adds zero-tap entries for the positions of a (pitch x pitch) grid outside the leading (used x used) block:
// the epilogue then writes zeros there (gradient of the MNIST crop = zero padding, needed by BN statistics).
void plan_add_zero_positions(LayerPlan& p, int used, int pitch, int ncols);

}  // namespace dg
