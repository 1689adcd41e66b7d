// TEST SUPPORT (never linked into the product library): executes a dg::LayerPlan with plain host
// loops so the per-position tap tables can be checked against the oracle on a CPU-only box.
#include <cstring>
#include <string>
#include <vector>

#include "dg_plan.h"

extern "C" {

void* dgp_build(const char* kind, int p0, int p1, int p2, int p3, int p4, int p5, int p6) {
    dg::LayerPlan* p = new dg::LayerPlan();
    const std::string k(kind);
    if (k == "linear_fwd") *p = dg::plan_linear_fwd(p0, p1, p2);
    else if (k == "linear_bwd") *p = dg::plan_linear_bwd(p0, p1, p2, p3);
    else if (k == "deconv_fwd") *p = dg::plan_deconv_fwd(p0, p1, p2, p3, p4, p5, p6);
    else if (k == "deconv_bwd") *p = dg::plan_deconv_bwd(p0, p1, p2, p3, p4, p5, p6);
    else { delete p; return nullptr; }
    return p;
}

void dgp_free(void* h) { delete static_cast<dg::LayerPlan*>(h); }

// info[0..7] = n_pos, n_taps, a_rowstride, out_rowstride, w_rowstride, kch, ncols, macs_per_row
void dgp_info(void* h, long long* info) {
    const dg::LayerPlan& p = *static_cast<dg::LayerPlan*>(h);
    info[0] = (long long)p.pos.size();
    info[1] = (long long)p.taps.size();
    info[2] = p.a_rowstride;
    info[3] = p.out_rowstride;
    info[4] = p.w_rowstride;
    info[5] = p.kch;
    info[6] = p.ncols;
    info[7] = p.macs_per_row;
}

void dgp_tap_counts(void* h, int* counts) {
    const dg::LayerPlan& p = *static_cast<dg::LayerPlan*>(h);
    for (size_t i = 0; i < p.pos.size(); ++i) counts[i] = p.pos[i].tap_count;
}

// mode: 0 store, 1 bias, 2 bias+relu, 3 mask (in place over Out)
void dgp_apply(void* h, const double* A, const double* W, const double* bias, double* Out, int n_rows, int mode) {
    const dg::LayerPlan& p = *static_cast<dg::LayerPlan*>(h);
    for (int n = 0; n < n_rows; ++n)
        for (const dg::PosEntry& pe : p.pos)
            for (int c = 0; c < p.bn; ++c) {
                const int col = pe.n0 + c;
                double acc = 0.0;
                for (int t = 0; t < pe.tap_count; ++t) {
                    const dg::TapEntry& te = p.taps[pe.tap_begin + t];
                    const double* a = A + (long long)n * p.a_rowstride + te.a_off;
                    const double* w = W + te.w_off + (long long)col * p.w_rowstride;
                    for (int k = 0; k < p.kch; ++k) acc += a[k] * w[k];
                }
                double* o = Out + (long long)n * p.out_rowstride + pe.out_off + col;
                if (mode == 1 || mode == 2) acc += bias[col];
                if (mode == 2) acc = acc > 0 ? acc : 0;
                if (mode == 3) acc = *o > 0 ? acc : 0;
                *o = acc;
            }
}

}  // extern "C"
