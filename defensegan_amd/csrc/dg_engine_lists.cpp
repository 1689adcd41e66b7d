// Job lists of the engine (dg_engine.h): building and TIMING the candidate lists of a layer (get_jobs), their K-pair scratch,
// the fragment-order lists, and the tuning export / import of the C ABI.
#include "dg_engine.h"
#include <cstring>

#pragma GCC visibility push(hidden)
namespace dge {

// forget every tuned job list (an option that changes how the lists are built or timed was set)
void drop_job_lists(dg_handle* h) {
    ++h->list_epoch;
    for (GemmOp* op : {&h->F1, &h->B1}) { for (auto& jl : op->jobs) (void)hipFree(jl.d_jobs); op->jobs.clear(); }
    for (auto* vec : {&h->Fd, &h->Bd})
        for (auto& op : *vec) { for (auto& jl : op.jobs) (void)hipFree(jl.d_jobs); op.jobs.clear(); }
}

void free_batched(GemmOp& op) {
    for (auto& jl : op.jobs)
        if (jl.d_jobs) (void)hipFree(jl.d_jobs);
    op.jobs.clear();
    for (auto& fl : op.fjobs)
        if (fl.d_jobs) (void)hipFree(fl.d_jobs);
    op.fjobs.clear();
    if (op.d_cls) { (void)hipFree(op.d_cls); op.d_cls = nullptr; }
    if (op.d_btaps) { (void)hipFree(op.d_btaps); op.d_btaps = nullptr; }
    if (op.d_pos_a) { (void)hipFree(op.d_pos_a); op.d_pos_a = nullptr; }
    if (op.d_pos_out) { (void)hipFree(op.d_pos_out); op.d_pos_out = nullptr; }
}

// `base` = the layer planned with one PosEntry per position (bn == ncols)
int upload_batched(GemmOp& op, const dg::LayerPlan& base) {
    free_batched(op);
    // the job shapes are 64 / 128 columns wide and the kernel never masks columns or K: anything else would read and write
    // out of bounds (dg_create's latent_dim % 64 / net_dim % 64 checks guarantee this for the two generators)
    if (base.ncols % 64 != 0 || base.kch % 32 != 0 || base.kch <= 0)
        return fail(DG_E_INVALID, "layer %s: %d output columns / K %d per tap (need multiples of 64 / 32)", op.name.c_str(), base.ncols, base.kch);
    op.bplan = dg::make_batched(base);
    op.family = (base.ncols % 128 == 0) ? 0 : 1;
    const dg::BatchedPlan& b = op.bplan;
    auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIP_TRY(up((void**)&op.d_cls, b.cls.data(), b.cls.size() * sizeof(dg::ClassDesc)));
    HIP_TRY(up((void**)&op.d_btaps, b.taps.data(), b.taps.size() * sizeof(dg::TapEntry)));
    HIP_TRY(up((void**)&op.d_pos_a, b.pos_a.data(), b.pos_a.size() * sizeof(int)));
    HIP_TRY(up((void**)&op.d_pos_out, b.pos_out.data(), b.pos_out.size() * sizeof(int)));
    return DG_OK;
}

dg::GemmArgs gemm_args(dg_handle* h, const GemmOp& op, const JobList& jl, const float* A, float* Out, int group) {
    dg::GemmArgs a;
    a.A = A;
    a.W = op.W;
    a.Out = Out;
    a.bias = op.bias;
    a.jobs = jl.d_jobs;
    a.pair_scratch = jl.d_pair ? jl.d_pair + (size_t)group * jl.pair_stride : nullptr;
    a.pair_count = jl.d_pair_count ? jl.d_pair_count + (size_t)group * jl.pair_count_stride : nullptr;
    // a list without pairs that was timed faster on the PAIR instantiation (same arithmetic, another register allocation): any
    // non-null pointer selects it, nothing reads it
    if (jl.pair_kernel && !a.pair_scratch) a.pair_scratch = reinterpret_cast<float*>(jl.d_jobs);
    a.cls = op.d_cls;
    a.taps = op.d_btaps;
    a.pos_a = op.d_pos_a;
    a.pos_out = op.d_pos_out;
    a.a_rowstride = op.bplan.a_rowstride;
    a.out_rowstride = op.bplan.out_rowstride;
    a.w_rowstride = op.bplan.w_rowstride;
    a.kch = op.bplan.kch;
    a.mode = op.mode;
    a.stats = op.stats;
    a.stats_cols = op.bplan.ncols;
    a.gate_bits = nullptr;
    a.gate_words = 0;
    a.bn_pre = a.bn_fstats = a.bn_scale = a.bn_offset = nullptr;
    if (op.bn_act >= 0) {
        const ActInfo& bn = h->ai[(size_t)op.bn_act];
        a.bn_pre = bn.xhat; a.bn_fstats = bn.fstats; a.bn_scale = bn.scale; a.bn_offset = bn.offset;
    }
    a.n_jobs = jl.n_jobs;
    a.min_level = jl.min_level;
#ifdef DG_MEASURE
    // the trace buffer holds kJobTraceCap records (one per workgroup): larger launches are not traced
    a.trace = (h->d_job_trace && op.name == h->job_trace_op && jl.n_jobs <= kJobTraceCap) ? h->d_job_trace : nullptr;
#endif
    return a;
}

// copies of a list's K-pair scratch: one per row group that may run concurrently (option two_streams)
int pair_copies(const dg_handle* h) { return h->two_streams > 1 ? std::min(h->two_streams, (int)dg_handle::kMaxGroups) : 1; }

bool upload_jobs(JobList& jl, const std::vector<dg::JobDesc>& jobs, int family, int copies) {
    jl.n_jobs = (int)jobs.size();
    // one allocation: [job records][pair counters x copies][pair accumulator images x copies] (every free of d_jobs frees all of
    // it).  A copy per row group that may launch this list concurrently on its own stream (two groups of equal size share a list)
    const dg::PairNeeds pn = dg::pair_needs(jobs, family);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t jobs_bytes = up((jobs.size() + 1) * sizeof(dg::JobDesc));
    const size_t count_bytes = up((size_t)pn.pairs * sizeof(unsigned));
    const size_t img_bytes = up((size_t)pn.floats * sizeof(float));
    if (!pn.pairs) copies = 0;
    char* base = nullptr;
    if (hipMalloc(&base, jobs_bytes + (count_bytes + img_bytes) * (size_t)copies) != hipSuccess) return false;
    jl.d_jobs = reinterpret_cast<dg::JobDesc*>(base);
    jl.d_pair_count = pn.pairs ? reinterpret_cast<unsigned*>(base + jobs_bytes) : nullptr;
    jl.d_pair = pn.pairs ? reinterpret_cast<float*>(base + jobs_bytes + count_bytes * (size_t)copies) : nullptr;
    jl.pair_count_stride = count_bytes / sizeof(unsigned);
    jl.pair_stride = img_bytes / sizeof(float);
    jl.pair_copies = copies;
    if (hipMemcpy(jl.d_jobs, jobs.data(), jobs.size() * sizeof(dg::JobDesc), hipMemcpyHostToDevice) != hipSuccess ||
        (pn.pairs && hipMemset(jl.d_pair_count, 0, count_bytes * (size_t)copies) != hipSuccess)) {
        (void)hipFree(jl.d_jobs);
        jl.d_jobs = nullptr; jl.d_pair_count = nullptr; jl.d_pair = nullptr;
        return false;
    }
    return true;
}

// The arrival counters of a list's K-pair jobs return to zero by themselves (the second arrival wraps them), but a launch that died
// between the two arrivals of a pair -- a device fault, a killed process that shared the handle's memory -- would leave a counter
// at 1, and the FIRST arriver of the next call would then add a stale image and run the epilogue: silently wrong numbers.  Every
// call therefore clears the counters of the lists it is about to launch, on its own stream (a few hundred bytes per list: free,
// and capturable).  Same treatment as the folded update's counters.
int clear_pair_counters(dg_handle* h, int n_rows, hipStream_t s) {
    for (auto* vec : {&h->Fd, &h->Bd})
        for (auto& op : *vec)
            for (auto& jl : op.jobs)
                if (jl.n_rows == n_rows && jl.d_pair_count && jl.pair_copies > 0)
                    HIP_TRY(hipMemsetAsync(jl.d_pair_count, 0, jl.pair_count_stride * sizeof(unsigned) * (size_t)jl.pair_copies, s));
    for (GemmOp* op : {&h->F1, &h->B1})
        for (auto& jl : op->jobs)
            if (jl.n_rows == n_rows && jl.d_pair_count && jl.pair_copies > 0)
                HIP_TRY(hipMemsetAsync(jl.d_pair_count, 0, jl.pair_count_stride * sizeof(unsigned) * (size_t)jl.pair_copies, s));
    return DG_OK;
}

// Job list of `op` for this row count (built on first use, kept on the device).
//
// Candidates: a list that starts with full tiles leaves room for 2 (family 0) / 3 (family 1) workgroups per CU; lists cut
// to halves or quarters from the start need less LDS and registers (dg_gemm.hip, MINLEVEL) and get more slots; each with a
// few cutting thresholds (dg_plan.h build_jobs).  Every candidate computes bit-identical results (cuts are along M / N
// only), so the choice is purely one of speed: with `job_tune` the candidates are TIMED on the layer's real operands (the
// launch is repeated on the actual input; an in-place ReluGrad layer writes to a scratch copy of its output) and the fastest
// is kept; without it the cost model's simulated makespan decides.
const JobList* find_jobs(const GemmOp& op, int n_rows) {
    for (const auto& jl : op.jobs)
        if (jl.n_rows == n_rows) return &jl;
    return nullptr;
}

const JobList* get_jobs(dg_handle* h, GemmOp& op, int n_rows, const float* A, float* Out, hipStream_t s) {
    if (const JobList* have = find_jobs(op, n_rows)) return have;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    struct Cand { JobList jl; std::vector<dg::JobDesc> jobs; float ms = 0.f; };
    std::vector<Cand> cands;
    const int n_levels = 3;
    const bool tune = h->job_tune && h->job_slack <= 0.0 && A && Out;
    // (slack, taper) pairs offered to the timing: the cutting thresholds as before, plus tapered lists (dg_plan.h JobModel::taper)
    const double slacks_tune[] = {1e30, 0.85, 0.92, 0.97, 1.0, 1.04, 1.1, 1e30, 1e30, 1e30, 1.0, 1.0, 1.0};
    const double tapers_tune[] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.5, 0.65, 0.8, 0.5, 0.65, 0.8};
    const double slack_one[] = {h->job_slack};
    const double taper_one[] = {h->job_model.taper};
    const double* slacks = tune ? slacks_tune : slack_one;
    const double* tapers = tune ? tapers_tune : taper_one;
    const int n_slacks = tune ? (h->job_taper_tune ? 13 : 7) : 1;
    for (int lvl = 0; lvl < n_levels; ++lvl) {
        if (h->job_min_level >= 0 && lvl != std::min(h->job_min_level, n_levels - 1)) continue;
        for (int k = 0; k < n_slacks; ++k) {
            Cand c;
            c.jl.n_rows = n_rows;
            c.jl.min_level = lvl;
            c.jl.slack = slacks[k];
            c.jl.taper = tapers[k];
            dg::JobModel jm = h->job_model;
            jm.taper = tapers[k];
            c.jobs = dg::build_jobs(op.bplan, n_rows, op.family, cus * h->job_slots_per_cu[op.family][lvl], slacks[k],
                                    jm, &c.jl.predicted_us, lvl);
            if (h->job_pair_kernel >= 2) c.jl.pair_kernel = 1;
            if (h->job_prio >= 2) {
                c.jl.prio = 1;
                dg::assign_priorities(op.bplan, c.jobs, op.family, cus * h->job_slots_per_cu[op.family][lvl], jm, 1);
            }
            auto add = [&](Cand&& x) {
                for (const Cand& o : cands)
                    if (o.jl.min_level == x.jl.min_level && o.jobs.size() == x.jobs.size() &&
                        std::memcmp(o.jobs.data(), x.jobs.data(), x.jobs.size() * sizeof(dg::JobDesc)) == 0) return;
                cands.push_back(std::move(x));
            };
            // A list that fits the resident slots is dispatched in one go, workgroup i to CU ~ i mod #CUs: with the jobs in
            // descending order CU 0 collects the longest of every round and the last CU the shortest.  Second candidate:
            // every other round of #CUs jobs reversed (boustrophedon), which evens the per-CU sums out (small batches).
            const size_t slots = (size_t)cus * h->job_slots_per_cu[op.family][lvl];
            // Lists of several dispatch rounds: second candidate in XCD-locality order (a permutation; kept only if it is timed
            // faster -- without timing the cost model cannot see the difference, so it is not offered)
            if (tune && h->job_xcd_head > 0.0 && c.jobs.size() > (size_t)cus) {
                Cand lx;
                lx.jl = c.jl;
                lx.jl.xcd_order = 1;
                lx.jl.xcd_head = h->job_xcd_head;
                lx.jobs = c.jobs;
                dg::order_for_xcd(lx.jobs, n_rows, h->job_xcd_head);
                lx.jl.predicted_us = dg::simulate_jobs(op.bplan, lx.jobs, op.family, (int)slots, h->job_model);
                // row-major order gives up longest-first: with only 2-3 dispatch rounds (MNIST at 2560 rows) the long jobs of
                // the last rows then end the launch 10-60 % late in the simulation -- such lists are not worth timing; with ten
                // rounds (CelebA's 32x32 layers) the order costs nothing
                if (lx.jl.predicted_us <= 1.03 * c.jl.predicted_us) add(std::move(lx));
            }
            if (tune && c.jobs.size() <= slots && c.jobs.size() > (size_t)cus) {
                Cand sn;
                sn.jl = c.jl;
                sn.jl.snake = 1;
                sn.jobs = c.jobs;
                dg::snake_order(sn.jobs, cus);
                // third candidate: the jobs partitioned into per-CU sets of equal predicted work (dg_plan.h balance_order)
                Cand bl;
                bl.jl = c.jl;
                bl.jl.snake = 2;
                bl.jobs = c.jobs;
                dg::balance_order(op.bplan, bl.jobs, op.family, cus, h->job_slots_per_cu[op.family][lvl], jm);
                add(std::move(c));
                add(std::move(sn));
                if (h->job_balance) add(std::move(bl));
            } else {
                add(std::move(c));
            }
        }
    }
    // lists BUILT for one dispatch round with the work balanced over the CUs (dg_plan.h jobs_balanced), one per starting level
    if (tune && h->job_balance >= 1) {
        for (int lvl = 0; lvl < n_levels; ++lvl) {
            if (h->job_min_level >= 0 && lvl != std::min(h->job_min_level, n_levels - 1)) continue;
            Cand c;
            c.jl.n_rows = n_rows;
            c.jl.min_level = lvl;
            c.jl.snake = 4;
            c.jobs = dg::jobs_balanced(op.bplan, n_rows, op.family, cus, h->job_slots_per_cu[op.family][lvl], lvl, h->job_model);
            if (c.jobs.empty()) continue;
            c.jl.predicted_us = dg::simulate_jobs(op.bplan, c.jobs, op.family, cus * h->job_slots_per_cu[op.family][lvl], h->job_model);
            if (h->job_pair_kernel >= 2) c.jl.pair_kernel = 1;
            if (h->job_prio >= 2) {
                c.jl.prio = 1;
                dg::assign_priorities(op.bplan, c.jobs, op.family, cus * h->job_slots_per_cu[op.family][lvl], h->job_model, 1);
            }
            cands.push_back(std::move(c));
        }
    }
    if (cands.empty()) return nullptr;
    size_t best = 0;
    for (size_t i = 1; i < cands.size(); ++i)
        if (cands[i].jl.predicted_us < cands[best].jl.predicted_us) best = i;
    // launches of several milliseconds have thousands of jobs per slot wave: the lists differ by < 1 % there, not worth timing
    if (tune && cands.size() > 1 && cands[best].jl.predicted_us < 3000.0) {
        float* scratch = nullptr;
        float* out = Out;
        const size_t out_bytes = (size_t)n_rows * op.bplan.out_rowstride * sizeof(float);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        if (ok && op.mode == dg::EPI_MASK) {          // in place over its gates: time it on a copy
            ok = hipMalloc(&scratch, out_bytes) == hipSuccess &&
                 hipMemcpyAsync(scratch, Out, out_bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
            out = scratch;
        }
        // One untimed launch keeps the stream busy while the timed ones are queued behind it, so the interval between the
        // two events holds no host submission gaps; short layers are repeated more often.
        auto time_list = [&](const JobList& jl, int scale, float* ms_out) {
            dg::GemmArgs a = gemm_args(h, op, jl, A, out);
            if (h->tune_gates && op.mode == dg::EPI_MASK) {          // the launch this layer will really make (fragment-order path)
                a.mode = dg::EPI_MASK_BITS;
                a.gate_bits = h->tune_gates;
                a.gate_words = (int)(op.bplan.out_rowstride / 32);
            }
#ifdef DG_MEASURE
            a.trace = nullptr;                       // candidate launches are not the traced ones
#endif
            const int reps = scale * std::max(2, std::min(16, (int)(1500.0 / std::max(jl.predicted_us, 1.0))));
            dg::launch_gemm(op.family, a, s);
            (void)hipEventRecord(e0, s);
            for (int rep = 0; rep < reps; ++rep) dg::launch_gemm(op.family, a, s);
            (void)hipEventRecord(e1, s);
            float ms = 0.f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
            *ms_out = ms / reps;
            return true;
        };
        // Two stages (the tapered lists doubled the candidates, and every first use of a call shape -- the ragged last batch of
        // an evaluation included -- pays for them): the plain lists of every starting level first; tapered lists are then timed
        // only for the levels whose best plain list came within 2 % of the best overall (a taper re-cuts the END of a list, it
        // does not make up for a starting level that is 5-10 % behind).
        float level_best[3] = {1e30f, 1e30f, 1e30f};
        for (size_t i = 0; ok && i < cands.size(); ++i) {
            Cand& c = cands[i];
            if (c.jl.taper > 0.0) continue;
            ok = upload_jobs(c.jl, c.jobs, op.family, pair_copies(h)) && time_list(c.jl, 1, &c.ms);
            if (ok && c.ms < level_best[c.jl.min_level]) level_best[c.jl.min_level] = c.ms;
        }
        const float plain_best = std::min(level_best[0], std::min(level_best[1], level_best[2]));
        for (size_t i = 0; ok && i < cands.size(); ++i) {
            Cand& c = cands[i];
            if (c.jl.taper <= 0.0) continue;
            if (level_best[c.jl.min_level] > 1.02f * plain_best) { c.ms = 1e30f; continue; }      // never uploaded, never kept
            ok = upload_jobs(c.jl, c.jobs, op.family, pair_copies(h)) && time_list(c.jl, 1, &c.ms);
        }
        if (ok && h->job_spread) {
            // the three fastest multi-round lists so far, once more in spread order (dg_plan.h spread_order: same jobs, desynchronised)
            std::vector<size_t> top;
            for (size_t i = 0; i < cands.size(); ++i)
                if (cands[i].ms < 1e29f && cands[i].jl.snake == 0 && !cands[i].jl.xcd_order &&
                    cands[i].jobs.size() > (size_t)cus * h->job_slots_per_cu[op.family][cands[i].jl.min_level]) top.push_back(i);
            std::sort(top.begin(), top.end(), [&](size_t x, size_t y) { return cands[x].ms < cands[y].ms; });
            if (top.size() > 3) top.resize(3);
            for (size_t k = 0; ok && k < top.size(); ++k) {
                Cand c;
                c.jl = cands[top[k]].jl;
                c.jl.d_jobs = nullptr;
                c.jl.snake = 3;
                c.jobs = cands[top[k]].jobs;
                dg::JobModel jm = h->job_model;
                jm.taper = c.jl.taper;
                dg::spread_order(op.bplan, c.jobs, op.family, cus * h->job_slots_per_cu[op.family][c.jl.min_level], jm);
                ok = upload_jobs(c.jl, c.jobs, op.family, pair_copies(h)) && time_list(c.jl, 1, &c.ms);
                cands.push_back(std::move(c));
            }
        }
        if (ok && h->job_pair_kernel >= 1) {
            // the two fastest lists without K-pair jobs, once more on the PAIR instantiation of the kernel
            std::vector<size_t> top;
            for (size_t i = 0; i < cands.size(); ++i)
                if (cands[i].ms < 1e29f && cands[i].jl.d_jobs && !cands[i].jl.d_pair) top.push_back(i);
            std::sort(top.begin(), top.end(), [&](size_t x, size_t y) { return cands[x].ms < cands[y].ms; });
            if (top.size() > 2) top.resize(2);
            for (size_t k = 0; ok && k < top.size(); ++k) {
                Cand c;
                c.jl = cands[top[k]].jl;
                c.jl.d_jobs = nullptr;
                c.jl.pair_kernel = 1;
                c.jobs = cands[top[k]].jobs;
                ok = upload_jobs(c.jl, c.jobs, op.family, pair_copies(h)) && time_list(c.jl, 1, &c.ms);
                cands.push_back(std::move(c));
            }
        }
        if (ok && h->job_prio == 1) {
            // the three fastest lists so far, once more with wave priorities by predicted job length (same jobs, same order)
            std::vector<size_t> top;
            for (size_t i = 0; i < cands.size(); ++i) if (cands[i].ms < 1e29f) top.push_back(i);
            std::sort(top.begin(), top.end(), [&](size_t x, size_t y) { return cands[x].ms < cands[y].ms; });
            if (top.size() > 3) top.resize(3);
            for (size_t k = 0; ok && k < top.size(); ++k) {
                Cand c;
                c.jl = cands[top[k]].jl;
                c.jl.d_jobs = nullptr;
                c.jl.prio = 1;
                c.jobs = cands[top[k]].jobs;
                dg::JobModel jm = h->job_model;
                jm.taper = c.jl.taper;
                dg::assign_priorities(op.bplan, c.jobs, op.family, cus * h->job_slots_per_cu[op.family][c.jl.min_level], jm, 1);
                bool any = false;
                for (const dg::JobDesc& j : c.jobs) any = any || j.prio != 0;
                if (!any) continue;
                ok = upload_jobs(c.jl, c.jobs, op.family, pair_copies(h)) && time_list(c.jl, 1, &c.ms);
                cands.push_back(std::move(c));
            }
        }
        if (ok) {
            for (size_t i = 0; i < cands.size(); ++i)
                if (cands[i].ms < cands[best].ms) best = i;
            // The first pass is a few launches per candidate while the clocks may still be settling: candidates within 3 % of
            // its winner (at most 4) are timed again, longer, once in order and once in reverse; the sum decides.
            std::vector<size_t> fin;
            for (size_t i = 0; i < cands.size(); ++i)
                if (cands[i].ms <= 1.03f * cands[best].ms) fin.push_back(i);
            std::sort(fin.begin(), fin.end(), [&](size_t x, size_t y) { return cands[x].ms < cands[y].ms; });
            if (fin.size() > 4) fin.resize(4);
            if (fin.size() > 1) {
                std::vector<float> sum(fin.size(), 0.f);
                bool ok2 = true;
                for (int pass = 0; ok2 && pass < 2; ++pass)
                    for (size_t k = 0; ok2 && k < fin.size(); ++k) {
                        const size_t q = pass == 0 ? k : fin.size() - 1 - k;
                        float ms = 0.f;
                        ok2 = time_list(cands[fin[q]].jl, 2, &ms);
                        sum[q] += ms;
                    }
                if (ok2) {
                    size_t w = 0, pref = 0;
                    for (size_t k = 0; k < fin.size(); ++k) {
                        cands[fin[k]].ms = 0.5f * sum[k];
                        if (sum[k] < sum[w]) w = k;
                        if (cands[fin[k]].jl.predicted_us < cands[fin[pref]].jl.predicted_us) pref = k;
                    }
                    // finalists within 0.7 % of each other are a coin toss from run to run (seen: Generator.2's backward taking a
                    // level-0 list in one process and a level-1 list in the next): then the cost model's favourite among them is
                    // kept, so that two runs on the same device make the same choice unless one list is measurably faster
                    // (not between the two kernel forms of ONE list: there the timing compares like with like, and either choice
                    // gives the same results)
                    auto same_list = [&](const JobList& a, const JobList& b) {
                        return a.min_level == b.min_level && a.slack == b.slack && a.taper == b.taper && a.snake == b.snake &&
                               a.xcd_order == b.xcd_order && a.prio == b.prio && a.n_jobs == b.n_jobs;
                    };
                    if (sum[pref] <= 1.007f * sum[w] && !same_list(cands[fin[pref]].jl, cands[fin[w]].jl)) w = pref;
                    best = fin[w];
                }
            }
            cands[best].jl.measured_us = cands[best].ms * 1e3;
            if (getenv("DG_TUNE_VERBOSE")) {
                for (size_t i = 0; i < cands.size(); ++i)
                    if (cands[i].ms < 1e29f)
                    fprintf(stderr, "[dg tune] %s rows %d level %d%s%s%s slack %-6.3g taper %-4.2f jobs %5d model %8.1f us measured %8.1f us%s\n", op.name.c_str(),
                            n_rows, cands[i].jl.min_level, cands[i].jl.xcd_order ? " xcd" : "    ", cands[i].jl.snake == 4 ? " built" : cands[i].jl.snake == 3 ? " sprd " : cands[i].jl.snake == 2 ? " balan" : cands[i].jl.snake ? " snake" : "      ", cands[i].jl.pair_kernel ? " pk  " : cands[i].jl.prio ? " prio" : "     ",
                            cands[i].jl.slack, cands[i].jl.taper, (int)cands[i].jobs.size(), cands[i].jl.predicted_us, cands[i].ms * 1e3,
                            i == best ? "  <- kept" : "");
            }
        }
        for (size_t i = 0; i < cands.size(); ++i)
            if (i != best && cands[i].jl.d_jobs) { (void)hipFree(cands[i].jl.d_jobs); cands[i].jl.d_jobs = nullptr; }
        if (scratch) (void)hipFree(scratch);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    JobList jl = cands[best].jl;
    if (!jl.d_jobs && !upload_jobs(jl, cands[best].jobs, op.family, pair_copies(h))) return nullptr;
    if (op.jobs.size() >= 16) {                 // callers with many distinct batch sizes: keep the table bounded
        (void)hipFree(op.jobs.front().d_jobs);
        op.jobs.erase(op.jobs.begin());
    }
    op.jobs.push_back(jl);
    ++h->list_epoch;
    return &op.jobs.back();
}

const FragList* find_frag_jobs(const GemmOp& op, int n_rows) {
    for (const auto& fl : op.fjobs)
        if (fl.n_rows == n_rows) return &fl;
    return nullptr;
}

const FragList* get_frag_jobs(GemmOp& op, int n_rows, int persist_wgs) {
    if (const FragList* have = find_frag_jobs(op, n_rows)) return have;
    std::vector<int> begin;
    const std::vector<dg::FragJob> jobs = persist_wgs > 0 ? dg::build_frag_tiles(op.bplan, n_rows, persist_wgs, &begin)
                                                         : dg::build_frag_jobs(op.bplan, n_rows);
    if (jobs.empty()) return nullptr;
    FragList fl;
    fl.n_rows = n_rows;
    fl.n_jobs = (int)jobs.size();
    fl.n_wgs = persist_wgs;
    const size_t jb = jobs.size() * sizeof(dg::FragJob), bb = begin.size() * sizeof(int);
    char* base = nullptr;
    if (hipMalloc(&base, jb + bb + 16) != hipSuccess) return nullptr;
    fl.d_jobs = reinterpret_cast<dg::FragJob*>(base);
    if (hipMemcpy(fl.d_jobs, jobs.data(), jb, hipMemcpyHostToDevice) != hipSuccess ||
        (bb && hipMemcpy(base + jb, begin.data(), bb, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipFree(base);
        return nullptr;
    }
    fl.d_begin = bb ? reinterpret_cast<int*>(base + jb) : nullptr;
    if (op.fjobs.size() >= 16) { (void)hipFree(op.fjobs.front().d_jobs); op.fjobs.erase(op.fjobs.begin()); }
    op.fjobs.push_back(fl);
    return &op.fjobs.back();
}

}  // namespace dge
#pragma GCC visibility pop

extern "C" {

// ---- tuning export / import ---------------------------------------------------------------------------------------------
// The job list a layer runs with for a row count is chosen by TIMING candidates (get_jobs), so two processes -- the bench and a
// profiler pass, or the ranks of a multi-GPU run -- can settle on different lists for the same layer.  Every list is a pure
// function of (layer plan, row count, starting level, cutting threshold, order variant): exporting those few numbers and
// importing them elsewhere reproduces the lists exactly, without timing.
int64_t dg_export_tuning(dg_handle* h, char* buf, int64_t cap) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    std::string out;
    char line[256];
    snprintf(line, sizeof line, "dgtune 1 arch %d latent %d net_dim %d use_bn %d nsplit %d cus %d\n", h->arch, h->latent, h->net_dim,
             h->use_bn, h->nsplit, h->cu_count);
    out += line;
    // the timed one-group / several-groups choice of every call shape (option two_streams = "auto"): "groups B R n"
    for (const auto& kv : h->group_choice) {
        snprintf(line, sizeof line, "groups %d %d %d\n", kv.first.first, kv.first.second, kv.second);
        out += line;
    }
    auto dump = [&](const GemmOp& op) {
        for (const JobList& jl : op.jobs) {
            dg::TuneRecord r;
            r.op = op.name; r.n_rows = jl.n_rows; r.min_level = jl.min_level; r.slack = jl.slack; r.snake = jl.snake;
            r.xcd_order = jl.xcd_order; r.xcd_head = jl.xcd_head; r.n_jobs = jl.n_jobs; r.measured_us = jl.measured_us; r.taper = jl.taper;
            r.prio = jl.prio; r.pair_kernel = jl.pair_kernel;
            out += dg::format_tune_record(r);
        }
    };
    dump(h->F1);
    for (const auto& op : h->Fd) dump(op);
    for (const auto& op : h->Bd) dump(op);
    dump(h->B1);
    const int64_t need = (int64_t)out.size() + 1;
    if (buf && cap >= need) std::memcpy(buf, out.c_str(), (size_t)need);
    else if (buf && cap > 0) buf[0] = 0;
    return need;
}

int dg_import_tuning(dg_handle* h, const char* text) {
    if (!h || !text) return fail(DG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    const char* p = text;
    int ver = 0, arch = 0, latent = 0, net_dim = 0, use_bn = 0, nsplit = 0, cus = 0, used = 0;
    if (sscanf(p, "dgtune %d arch %d latent %d net_dim %d use_bn %d nsplit %d cus %d%n", &ver, &arch, &latent, &net_dim, &use_bn, &nsplit, &cus, &used) != 7 || ver != 1)
        return fail(DG_E_INVALID, "not a dg_export_tuning text (header)");
    if (arch != h->arch || latent != h->latent || net_dim != h->net_dim || use_bn != h->use_bn || nsplit != h->nsplit || cus != h->cu_count)
        return fail(DG_E_INVALID, "tuning was exported for another configuration (arch %d latent %d net_dim %d use_bn %d nsplit %d, %d CUs)",
                    arch, latent, net_dim, use_bn, nsplit, cus);
    p += used;
    std::vector<GemmOp*> ops = {&h->F1, &h->B1};
    for (auto& op : h->Fd) ops.push_back(&op);
    for (auto& op : h->Bd) ops.push_back(&op);
    HIP_TRY(hipDeviceSynchronize());      // lists that are replaced may still be in use by queued launches
    int n_imported = 0;
    for (;;) {
        while (*p == '\n' || *p == ' ' || *p == '\r' || *p == '\t') ++p;
        if (std::strncmp(p, "groups ", 7) == 0) {          // a call shape's row-group choice: installed instead of being timed again
            int gb = 0, gr = 0, gn = 0, gu = 0;
            if (sscanf(p, "groups %d %d %d%n", &gb, &gr, &gn, &gu) != 3 || gb < 1 || gr < 1 || gn < 1 || gn > dg_handle::kMaxGroups)
                return fail(DG_E_INVALID, "malformed row-group record near '%.40s'", p);
            h->group_choice[std::make_pair(gb, gr)] = gn;
            p += gu;
            continue;
        }
        dg::TuneRecord r;
        if (!dg::parse_tune_record(&p, &r)) {
            if (*p) return fail(DG_E_INVALID, "malformed tuning record near '%.40s'", p);
            break;
        }
        GemmOp* op = nullptr;
        for (GemmOp* o : ops) if (o->name == r.op) op = o;
        if (!op || r.n_rows < 1 || r.n_rows > (1 << 24) || r.min_level < 0 || r.min_level > 2 || r.prio < 0 || r.prio > 1 || r.pair_kernel < 0 || r.pair_kernel > 1)
            return fail(DG_E_INVALID, "tuning record for unknown layer '%s' / bad row count %d / level %d", r.op.c_str(), r.n_rows, r.min_level);
        JobList jl;
        jl.n_rows = r.n_rows; jl.min_level = r.min_level; jl.slack = r.slack; jl.snake = r.snake; jl.xcd_order = r.xcd_order;
        jl.xcd_head = r.xcd_head; jl.measured_us = r.measured_us; jl.taper = r.taper; jl.prio = r.prio; jl.pair_kernel = r.pair_kernel;
        const std::vector<dg::JobDesc> jobs = dg::jobs_from_record(op->bplan, op->family, h->cu_count, h->job_slots_per_cu[op->family][r.min_level],
                                                                   r, h->job_model, &jl.predicted_us);
        if ((int)jobs.size() != r.n_jobs)
            return fail(DG_E_INVALID, "layer %s, %d rows: the record describes %d jobs, this build makes %d (other cost model or planner)",
                        r.op.c_str(), r.n_rows, r.n_jobs, (int)jobs.size());
        if (!upload_jobs(jl, jobs, op->family, pair_copies(h))) return fail(DG_E_NOMEM, "cannot upload the job list of layer %s", r.op.c_str());
        ++h->list_epoch;                 // from here on lists are replaced: a captured loop may point at one (also when a later record fails)
        for (auto it = op->jobs.begin(); it != op->jobs.end();)
            if (it->n_rows == r.n_rows) { (void)hipFree(it->d_jobs); it = op->jobs.erase(it); } else ++it;
        if (op->jobs.size() >= 16) { (void)hipFree(op->jobs.front().d_jobs); op->jobs.erase(op->jobs.begin()); }
        op->jobs.push_back(jl);
        ++n_imported;
    }
    return n_imported;
}

}  // extern "C"
