#include "dg_plan.h"

#include <algorithm>

namespace dg {

static void sort_positions(LayerPlan& p) {
    std::stable_sort(p.pos.begin(), p.pos.end(),
                     [](const PosEntry& a, const PosEntry& b) { return a.tap_count > b.tap_count; });
}

LayerPlan plan_linear_fwd(int latent, int features, int bn) {
    LayerPlan p;
    p.name = "linear_fwd";
    p.a_rowstride = latent;
    p.out_rowstride = features;
    p.w_rowstride = latent;
    p.kch = latent;
    p.ncols = features;
    p.bn = bn;
    p.taps.push_back(TapEntry{0, 0});
    for (int n0 = 0; n0 < features; n0 += bn) p.pos.push_back(PosEntry{0, n0, 0, 1});
    p.macs_per_row = (long long)latent * features;
    return p;
}

LayerPlan plan_linear_bwd(int latent, int features, int nsplit, int bn) {
    LayerPlan p;
    p.name = "linear_bwd";
    const int kc = features / nsplit;
    p.a_rowstride = features;
    p.out_rowstride = (long long)nsplit * latent;
    p.w_rowstride = features;
    p.kch = kc;
    p.ncols = latent;
    p.bn = bn;
    for (int s = 0; s < nsplit; ++s) {
        p.taps.push_back(TapEntry{s * kc, s * kc});
        for (int n0 = 0; n0 < latent; n0 += bn) p.pos.push_back(PosEntry{s * latent, n0, s, 1});
    }
    p.macs_per_row = (long long)latent * features;
    return p;
}

LayerPlan plan_deconv_fwd(int h_in, int in_pitch, int e_out, int out_pitch, int cin, int cout, int bn) {
    LayerPlan p;
    p.name = "deconv_fwd";
    p.a_rowstride = (long long)in_pitch * in_pitch * cin;
    p.out_rowstride = (long long)out_pitch * out_pitch * cout;
    p.w_rowstride = cin;
    p.kch = cin;
    p.ncols = cout;
    p.bn = bn;
    for (int i = 0; i < e_out; ++i)
        for (int j = 0; j < e_out; ++j) {
            const int tb = (int)p.taps.size();
            for (int kh = 0; kh < 5; ++kh) {
                const int th = i + 1 - kh;
                if (th < 0 || (th & 1)) continue;
                const int oh = th >> 1;
                if (oh >= h_in) continue;
                for (int kw = 0; kw < 5; ++kw) {
                    const int tw = j + 1 - kw;
                    if (tw < 0 || (tw & 1)) continue;
                    const int ow = tw >> 1;
                    if (ow >= h_in) continue;
                    p.taps.push_back(TapEntry{(oh * in_pitch + ow) * cin, (kh * 5 + kw) * cout * cin});
                }
            }
            const int tc = (int)p.taps.size() - tb;
            p.macs_per_row += (long long)tc * cin * cout;
            for (int n0 = 0; n0 < cout; n0 += bn)
                p.pos.push_back(PosEntry{(i * out_pitch + j) * cout, n0, tb, tc});
        }
    sort_positions(p);
    return p;
}

LayerPlan plan_deconv_bwd(int h_in, int out_pitch, int e_out, int a_pitch, int cin, int cout, int bn) {
    LayerPlan p;
    p.name = "deconv_bwd";
    p.a_rowstride = (long long)a_pitch * a_pitch * cout;
    p.out_rowstride = (long long)out_pitch * out_pitch * cin;
    p.w_rowstride = cout;
    p.kch = cout;
    p.ncols = cin;
    p.bn = bn;
    for (int oh = 0; oh < h_in; ++oh)
        for (int ow = 0; ow < h_in; ++ow) {
            const int tb = (int)p.taps.size();
            for (int kh = 0; kh < 5; ++kh) {
                const int i = 2 * oh + kh - 1;
                if (i < 0 || i >= e_out) continue;
                for (int kw = 0; kw < 5; ++kw) {
                    const int j = 2 * ow + kw - 1;
                    if (j < 0 || j >= e_out) continue;
                    p.taps.push_back(TapEntry{(i * a_pitch + j) * cout, (kh * 5 + kw) * cin * cout});
                }
            }
            const int tc = (int)p.taps.size() - tb;
            p.macs_per_row += (long long)tc * cin * cout;
            for (int n0 = 0; n0 < cin; n0 += bn)
                p.pos.push_back(PosEntry{(oh * out_pitch + ow) * cin, n0, tb, tc});
        }
    sort_positions(p);
    return p;
}

void plan_add_zero_positions(LayerPlan& p, int used, int pitch, int ncols) {
    for (int i = 0; i < pitch; ++i)
        for (int j = 0; j < pitch; ++j) {
            if (i < used && j < used) continue;
            for (int n0 = 0; n0 < ncols; n0 += p.bn) p.pos.push_back(PosEntry{(i * pitch + j) * ncols, n0, 0, 0});
        }
}

}  // namespace dg
