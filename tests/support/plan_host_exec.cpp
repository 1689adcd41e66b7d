// TEST SUPPORT (never linked into the product library): executes a dg::LayerPlan with plain host
// loops so the per-position tap tables can be checked against the oracle on a CPU-only box.
#include <algorithm>
#include <cstring>
#include <map>
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


// ---- position-batched form: the job list of dg::build_jobs executed with host loops -------------------------------
struct Batched {
    dg::BatchedPlan plan;
    std::vector<dg::JobDesc> jobs;
    int family = 0;
    double predicted_us = 0.0;
};

void* dgp2_build(void* layer_plan) {
    Batched* b = new Batched();
    b->plan = dg::make_batched(*static_cast<dg::LayerPlan*>(layer_plan));
    b->family = b->plan.ncols % 128 == 0 ? 0 : 1;
    return b;
}
void dgp2_free(void* h) { delete static_cast<Batched*>(h); }

// info = n_classes, n_positions, n_taps, family
void dgp2_info(void* h, long long* info) {
    const Batched& b = *static_cast<Batched*>(h);
    info[0] = (long long)b.plan.cls.size();
    info[1] = (long long)b.plan.pos_a.size();
    info[2] = (long long)b.plan.taps.size();
    info[3] = b.family;
}
void dgp2_classes(void* h, int* out) {        // per class: pos_count, nchunks
    const Batched& b = *static_cast<Batched*>(h);
    for (size_t i = 0; i < b.plan.cls.size(); ++i) { out[2 * i] = b.plan.cls[i].pos_count; out[2 * i + 1] = b.plan.cls[i].nchunks; }
}
void dgp2_class_grids(void* h, int* out) {    // per class: wc (0 = not a grid), then whether the tables equal the grid formula
    const Batched& b = *static_cast<Batched*>(h);
    for (size_t i = 0; i < b.plan.cls.size(); ++i) {
        const dg::ClassDesc& c = b.plan.cls[i];
        int same = 1;
        for (int j = 0; j < c.pos_count && c.wc; ++j)
            same &= b.plan.pos_a[c.pos_begin + j] == c.a_base + (j / c.wc) * c.a_rs + (j % c.wc) * c.a_cs &&
                    b.plan.pos_out[c.pos_begin + j] == c.o_base + (j / c.wc) * c.o_rs + (j % c.wc) * c.o_cs;
        out[2 * i] = c.wc; out[2 * i + 1] = same;
    }
}
int dgp2_make_jobs(void* h, int n_rows, int slots, double alpha) {
    Batched& b = *static_cast<Batched*>(h);
    b.jobs = dg::build_jobs(b.plan, n_rows, b.family, slots, alpha, dg::JobModel(), &b.predicted_us);
    return (int)b.jobs.size();
}
// re-arranges the list built last (dg_plan.h order_for_xcd) and returns its simulated makespan on `slots` slots
double dgp2_order_for_xcd(void* h, int n_rows, double head_frac, int n_xcd, int slots) {
    Batched& b = *static_cast<Batched*>(h);
    dg::order_for_xcd(b.jobs, n_rows, head_frac, n_xcd);
    return dg::simulate_jobs(b.plan, b.jobs, b.family, slots, dg::JobModel());
}
double dgp2_predicted_us(void* h) { return static_cast<Batched*>(h)->predicted_us; }

// ---- tuning records (dg_export_tuning / dg_import_tuning, dg_plan.h TuneRecord) ---------------------------------------
// Builds the list of (row count, level, threshold, order variant) the way the engine's timing does, keeps it as the current
// list and writes its record line into `line`; returns the list's length.
int dgp2_make_recorded(void* h, const char* op, int n_rows, int cus, int slots_per_cu, int min_level, double slack, int snake,
                       double xcd_head, double taper, char* line, int cap) {
    Batched& b = *static_cast<Batched*>(h);
    dg::TuneRecord r;
    r.op = op; r.n_rows = n_rows; r.min_level = min_level; r.slack = slack; r.snake = snake;
    r.xcd_order = xcd_head > 0.0; r.xcd_head = xcd_head;
    r.taper = taper;
    b.jobs = dg::jobs_from_record(b.plan, b.family, cus, slots_per_cu, r, dg::JobModel(), &b.predicted_us);
    r.n_jobs = (int)b.jobs.size();
    r.measured_us = 123.456;
    const std::string t = dg::format_tune_record(r);
    if ((int)t.size() + 1 > cap) return -1;
    std::memcpy(line, t.c_str(), t.size() + 1);
    return r.n_jobs;
}
// The work-balanced single-round list (dg_plan.h jobs_balanced) becomes the current list; returns its length (0 = does not fit one round)
int dgp2_make_balanced(void* h, int n_rows, int cus, int slots_per_cu, int min_level) {
    Batched& b = *static_cast<Batched*>(h);
    b.jobs = dg::jobs_balanced(b.plan, n_rows, b.family, cus, slots_per_cu, min_level);
    return (int)b.jobs.size();
}
// Wave priorities by predicted job length on the current list (dg_plan.h assign_priorities); out = the priority of every job
void dgp2_assign_priorities(void* h, int cus, int slots_per_cu, int mode, int* out) {
    Batched& b = *static_cast<Batched*>(h);
    dg::assign_priorities(b.plan, b.jobs, b.family, cus * slots_per_cu, dg::JobModel(), mode);
    for (size_t i = 0; i < b.jobs.size(); ++i) out[i] = b.jobs[i].prio;
}
// The current list's record line with the given priority mode (the list itself is left alone)
int dgp2_format_with_prio(void* h, const char* line_in, int prio, char* line, int cap) {
    const char* p = line_in;
    dg::TuneRecord r;
    if (!dg::parse_tune_record(&p, &r)) return -1;
    r.prio = prio;
    const std::string t = dg::format_tune_record(r);
    if ((int)t.size() + 1 > cap) return -1;
    std::memcpy(line, t.c_str(), t.size() + 1);
    (void)h;
    return 0;
}
// Parses `text` (record lines), rebuilds the list of its record number `which` and compares it byte for byte with the
// current list: 1 equal, 0 different, -1 malformed / no such record, -2 the record's job count does not match.
int dgp2_rebuild_matches(void* h, const char* text, int which, int cus, int slots_per_cu) {
    Batched& b = *static_cast<Batched*>(h);
    const char* p = text;
    dg::TuneRecord r;
    for (int i = 0; i <= which; ++i)
        if (!dg::parse_tune_record(&p, &r)) return -1;
    const std::vector<dg::JobDesc> jobs = dg::jobs_from_record(b.plan, b.family, cus, slots_per_cu, r);
    if ((int)jobs.size() != r.n_jobs) return -2;
    if (jobs.size() != b.jobs.size()) return 0;
    return std::memcmp(jobs.data(), b.jobs.data(), jobs.size() * sizeof(dg::JobDesc)) == 0 ? 1 : 0;
}
void dgp2_jobs(void* h, int* out) {           // per job: cls, shape, n0, n_first, j_first, m_valid
    const Batched& b = *static_cast<Batched*>(h);
    for (size_t i = 0; i < b.jobs.size(); ++i) {
        const dg::JobDesc& j = b.jobs[i];
        int* o = out + 6 * i;
        o[0] = j.cls; o[1] = j.shape; o[2] = j.n0; o[3] = j.n_first; o[4] = j.j_first; o[5] = j.m_valid;
    }
}
void dgp2_job_pairs(void* h, int* out) {      // per job: its own K chunks, pair id (0 = not a K-pair job), role, float offset of the pair's images
    const Batched& b = *static_cast<Batched*>(h);
    for (size_t i = 0; i < b.jobs.size(); ++i) {
        const dg::JobDesc& j = b.jobs[i];
        int* o = out + 4 * i;
        o[0] = j.nchunks; o[1] = j.pair_id; o[2] = j.pair_role; o[3] = j.pair_off;
    }
}
// Executes the job list the way the device kernel addresses it (row split by the class's magic multiplier, a_off relative
// to pos_a); `touched` (same shape as Out, int32) counts the writes per output element.
void dgp2_apply(void* h, const double* A, const double* W, const double* bias, double* Out, int* touched, int mode) {
    const Batched& b = *static_cast<Batched*>(h);
    const dg::BatchedPlan& p = b.plan;
    // K-pair jobs (dg_types.h JobDesc::pair_id): each covers half of the class's taps; the half that comes first in the list
    // leaves its sums here, the other adds them and runs the epilogue (on the device: whichever ARRIVES second; a + b = b + a)
    std::map<long long, double> half;
    for (const dg::JobDesc& jb : b.jobs) {
        const dg::ClassDesc& cd = p.cls[jb.cls];
        const int bn = b.family == 0 ? (jb.shape == 2 ? 64 : 128) : 64;      // columns of the job (dg_plan.cpp kShapeBN)
        const int n_taps = jb.n_taps;                                          // the job's OWN taps (jb.tap_begin ...)
        for (int r = 0; r < jb.m_valid; ++r) {
            const unsigned jj = (unsigned)(jb.j_first + r);
            const int q = (int)(((unsigned long long)(jj << 1) * cd.magic) >> 32);
            const int j = (int)jj - q * cd.pos_count;
            const long long n = (long long)jb.n_first + q;
            // the device's position arithmetic (dg_gemm.hip pos_of_a / pos_of_out): from the JOB RECORD's copy of the class grid
            int pa, po;
            if (jb.wc) {
                const int jh = (int)(((unsigned long long)((unsigned)j << 1) * jb.wc_magic) >> 32);
                pa = jb.a_base + jh * jb.a_rs + (j - jh * jb.wc) * jb.a_cs;
                po = jb.o_base + jh * jb.o_rs + (j - jh * jb.wc) * jb.o_cs;
            } else {
                pa = p.pos_a[cd.pos_begin + j];
                po = p.pos_out[cd.pos_begin + j];
            }
            for (int c = 0; c < bn; ++c) {
                const int col = jb.n0 + c;
                double acc = 0.0;
                for (int t = 0; t < n_taps; ++t) {
                    const dg::TapEntry& te = p.taps[jb.tap_begin + t];
                    const double* a = A + n * p.a_rowstride + pa + te.a_off;
                    const double* w = W + te.w_off + (long long)col * p.w_rowstride;
                    for (int k = 0; k < p.kch; ++k) acc += a[k] * w[k];
                }
                const long long o = n * p.out_rowstride + po + col;
                if (jb.pair_id) {
                    auto it = half.find(o);
                    if (it == half.end()) { half[o] = acc; continue; }       // first half of this element: nothing is written yet
                    acc += it->second;
                    half.erase(it);
                }
                if (mode == 1 || mode == 2) acc += bias[col];
                if (mode == 2) acc = acc > 0 ? acc : 0;
                if (mode == 3) acc = Out[o] > 0 ? acc : 0;
                Out[o] = acc;
                touched[o] += 1;
            }
        }
    }
}

// ---- fragment-order lists (dg_fgemm.hip): build_frag_jobs / build_frag_tiles executed with host loops ------------------------------
// n_wgs == 0: the per-job list; n_wgs > 0: the persistent form.  Addresses are formed the way the device forms them -- M block ->
// (row block, position) by the record's magic multipliers, taps by the record's affine grid -- in the logical NHWC layout (the
// fragment order is a permutation of the same elements).  info = n_jobs, supported, max ksplit, heaviest wave (K chunks), mean wave.
int dgp2_frag_apply(void* h, int n_rows, int n_wgs, const double* A, const double* W, const double* bias, double* Out, int* touched,
                    int mode, double* info) {
    const Batched& b = *static_cast<Batched*>(h);
    const dg::BatchedPlan& p = b.plan;
    info[1] = dg::frag_supported(p) ? 1 : 0;
    if (!dg::frag_supported(p)) return 0;
    std::vector<int> begin;
    std::vector<dg::FragJob> jobs = n_wgs > 0 ? dg::build_frag_tiles(p, n_rows, n_wgs, &begin) : dg::build_frag_jobs(p, n_rows);
    info[0] = (double)jobs.size(); info[2] = 0; info[3] = 0; info[4] = 0;
    if (n_wgs > 0) {
        if ((int)begin.size() != n_wgs * 4 + 1 || begin.front() != 0 || begin.back() != (int)jobs.size()) return -1;
        double tot = 0;
        for (int w = 0; w < n_wgs * 4; ++w) {
            if (begin[w + 1] < begin[w]) return -1;
            double l = 0;
            for (int i = begin[w]; i < begin[w + 1]; ++i) l += jobs[i].n_taps * (p.kch / 32) + 1.5;
            info[3] = std::max(info[3], l);
            tot += l;
        }
        info[4] = tot / (n_wgs * 4);
    }
    const int kc8 = p.kch / 8;
    for (const dg::FragJob& jb : jobs) {
        info[2] = std::max(info[2], (double)jb.ksplit);
        if ((jb.n_taps * kc8) % (4 * jb.ksplit)) return -2;                  // a wave's K part is whole turns of the operand ring
        for (int m = 0; m < jb.n_mblk; ++m) {
            const unsigned mblk = (unsigned)(jb.mblk0 + m);
            const int rb = (int)(((unsigned long long)(mblk << 1) * jb.s_magic) >> 32);
            const int jj = (int)mblk - rb * jb.s;
            const int jh = (int)(((unsigned long long)((unsigned)jj << 1) * jb.wc_magic) >> 32);
            const int jw = jj - jh * jb.wc;
            const int pa = jb.a_base + jh * jb.a_rs + jw * jb.a_cs, po = jb.o_base + jh * jb.o_rs + jw * jb.o_cs;
            for (int r = 0; r < 32; ++r) {
                const long long n = (long long)rb * 32 + r;
                if (n >= n_rows) break;
                for (int c = 0; c < 64; ++c) {
                    const int col = jb.cb0 * 32 + c;
                    double acc = 0.0;
                    for (int t = 0; t < jb.n_taps; ++t) {
                        const int u = (int)(((unsigned long long)((unsigned)t << 1) * jb.tap_nw_magic) >> 32), v = t - u * jb.tap_nw;
                        const double* a = A + n * p.a_rowstride + pa + jb.a0 + u * jb.a_u + v * jb.a_v;
                        const double* w = W + jb.w0 + u * jb.w_u + v * jb.w_v + (long long)col * p.w_rowstride;
                        for (int k = 0; k < p.kch; ++k) acc += a[k] * w[k];
                    }
                    const long long o = n * p.out_rowstride + po + col;
                    if (mode == 1 || mode == 2) acc += bias[col];
                    if (mode == 2) acc = acc > 0 ? acc : 0;
                    if (mode == 3) acc = Out[o] > 0 ? acc : 0;
                    Out[o] = acc;
                    touched[o] += 1;
                }
            }
        }
    }
    return (int)jobs.size();
}

}  // extern "C"
