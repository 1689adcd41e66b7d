// dg_set_option and dg_debug_read of the C ABI (dg_engine.h).
#include "dg_engine.h"

extern "C" {

int64_t dg_debug_read(dg_handle* h, const char* what, float* dst, int64_t n) {
    if (!h || !what || !dst) return fail(DG_E_INVALID, "null argument");
    const std::string w(what);
    const float* src = nullptr;
    int64_t avail = 0;
    if (w == "z") { src = h->z; avail = h->cap_rows * h->latent; }
    else if (w == "m") { src = h->m; avail = h->cap_rows * h->latent; }
    else if (w == "loss") { src = h->loss; avail = h->cap_rows; }
    else if (w == "y") { src = h->y; avail = h->cap_rows * h->P; }
    else if (w == "part") { src = h->part; avail = h->cap_rows * h->nsplit * h->latent; }
#ifdef DG_MEASURE
    else if (w == "job_trace" && h->d_job_trace) { src = reinterpret_cast<const float*>(h->d_job_trace); avail = (int64_t)kJobTraceCap * 4 * 2; }
    else if (w == "tail_trace" && h->d_tail_trace) { src = reinterpret_cast<const float*>(h->d_tail_trace); avail = 4096 * 8 * 2; }
#endif
    else if (w.size() == 4 && w.compare(0, 3, "act") == 0) {
        const int d = w[3] - '0';
        if (d >= 0 && d < (int)h->act.size()) { src = h->act[d]; avail = h->cap_rows * h->act_row[d]; }
        if (src && frag_on(h) && d + 1 < (int)h->act.size()) {
            // the fragment-order path keeps this activation in fragment order: hand it over as NHWC rows
            const int64_t cnt = n < avail ? n : avail;
            const int64_t rows = (cnt + h->act_row[d] - 1) / h->act_row[d];
            float* tmp = nullptr;
            HIP_TRY(hipMalloc(&tmp, (size_t)rows * h->act_row[d] * sizeof(float)));
            dg::launch_unfrag(h->actf[d], tmp, rows, h->act_row[d], nullptr);
            hipError_t e = hipMemcpy(dst, tmp, (size_t)cnt * sizeof(float), hipMemcpyDeviceToDevice);
            (void)hipFree(tmp);
            if (e != hipSuccess) return fail(DG_E_HIP, "hipMemcpy: %s", hipGetErrorString(e));
            return cnt;
        }
    }
    if (!src) return fail(DG_E_INVALID, "unknown buffer '%s'", what);
    const int64_t cnt = n < avail ? n : avail;
    HIP_TRY(hipMemcpy(dst, src, (size_t)cnt * sizeof(float), hipMemcpyDeviceToDevice));
    return cnt;
}

static int set_option(dg_handle* h, const char* key, const char* value);

int dg_set_option(dg_handle* h, const char* key, const char* value) {
    if (!h || !key || !value) return fail(DG_E_INVALID, "null argument");
    const int rc = set_option(h, key, value);
    if (rc == DG_OK) {
        ++h->list_epoch;                 // an accepted option: captured loops are rebuilt on their next use (a refused one changes nothing)
        h->group_choice.clear();         // ... and the one-group / several-groups choice of every call shape is timed again
    }
    return rc;
}

static int set_option(dg_handle* h, const char* key, const char* value) {
    const std::string k(key);
    if (k == "two_streams") {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        // number of concurrent row groups (0 = off, 2..8; historic "1" = two), or "auto": two groups where timing the call shape
        // finds them faster (CelebA's default)
        h->two_streams_auto = std::string(value) == "auto";
        h->two_streams = h->two_streams_auto ? 2 : atoi(value);
        if (h->two_streams == 1) h->two_streams = 2;
        drop_job_lists(h);                     // a list's K-pair scratch is sized by the number of groups that may launch it at once
        return DG_OK;
    }
    if (k == "lr_schedule") {
        const std::string v(value);
        if (v != "constant" && v != "intended") return fail(DG_E_INVALID, "lr_schedule: 'constant' or 'intended'");
        h->lr_intended = v == "intended";
        return DG_OK;
    }
    if (k == "two_stream_split") {       // two row groups: percent of the images in the first one (0 = halves)
        h->two_stream_split = atoi(value);
        return DG_OK;
    }
    if (k == "two_stream_min_rows") {
        h->two_stream_min_rows = atoi(value);
        return DG_OK;
    }
    if (k == "graph_max_rows") {         // call shapes of at most this many latent rows replay a captured graph of the loop; 0 = never
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->graph_max_rows = atoi(value) > 0 ? atoi(value) : 0;
        h->graph_broken = false;
        drop_graphs(h);
        free_workspace(h);               // the staging copy of the images is sized by this option
        return DG_OK;
    }
    if (k == "update_fold") {            // 1 = the momentum update rides in the Linear backward launch (needs latent_turn)
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->update_fold = atoi(value) != 0;
        return DG_OK;
    }
    if (k == "turn_fused") {             // 1 = Linear backward + update + next Linear forward as one launch (dg_turn.hip)
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->turn_fused = atoi(value) != 0;
        ++h->list_epoch;                 // captured loops hold the other launch sequence
        return DG_OK;
    }
    if (k == "bn_fused") {               // Batchnorm sums from the producing GEMM's epilogue: 2 = forward and backward (default), 1 = forward, 0 = passes
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->bn_fused = atoi(value) < 0 ? 0 : (atoi(value) > 2 ? 2 : atoi(value));
        free_workspace(h);               // the block-sum buffers exist (and are sized) by it
        drop_job_lists(h);
        return rebuild_plans(h);
    }
    if (k == "frag_path") {              // 1 = fragment-order forward path (dg_fgemm.hip; default), 0 = every GEMM on dg_gemm.hip
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->frag_path = atoi(value) < 0 ? 0 : (atoi(value) > 2 ? 2 : atoi(value));       // 2 = persistent waves (dg_fgemm.hip fgemm_persist_kernel)
        free_workspace(h);               // the fragment-order buffers exist only with it
        drop_job_lists(h);               // (the backward layers' lists were timed with the other epilogue)
        for (GemmOp& op : h->Fd) {       // the two forms keep different fragment-order lists
            for (auto& fl : op.fjobs) if (fl.d_jobs) (void)hipFree(fl.d_jobs);
            op.fjobs.clear();
        }
        return DG_OK;
    }
    if (k == "latent_turn") {            // 1 = weight-stationary Linear kernels (default), 0 = position-batched kernel
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->latent_turn = atoi(value) != 0;
        ++h->list_epoch;
        return DG_OK;                    // job lists the other setting needs are built by the next prepare / call
    }
    if (k == "lin_groups_fwd" || k == "lin_groups_bwd") {
        const int v = atoi(value);
        if (v < 0 || v > 4096) return fail(DG_E_INVALID, "%s: 0 (auto) .. 4096", key);
        (k == "lin_groups_fwd" ? h->lin_groups_fwd : h->lin_groups_bwd) = v;
        ++h->list_epoch;
        return DG_OK;
    }
    if (k == "tail_pipe") {
        if (h->use_bn) {                 // the Batchnorm form of the tail leaves one set of records per workgroup (ensure_workspace)
            HIP_TRY(hipSetDevice(h->device));
            HIP_TRY(hipDeviceSynchronize());
            free_workspace(h);
        }
        h->tail_pipe = atoi(value);
        return DG_OK;
    }
    if (k == "tail_pipe_version") {
        if (atoi(value) < 1 || atoi(value) > 3) return fail(DG_E_INVALID, "tail_pipe_version: 1, 2 or 3");
#ifndef DG_MEASURE
        if (atoi(value) == 2) return fail(DG_E_INVALID, "tail_pipe_version = 2 (the superseded second-generation kernel) needs the measurement build");
#endif
        h->tail_pipe_version = atoi(value);
        return DG_OK;
    }
    if (k == "tail_fwd_split") {
        h->tail_fwd_split = atoi(value) > 0 ? atoi(value) : 0;
        return DG_OK;
    }
    if (k == "tail_bwd_persist") {
#ifndef DG_MEASURE
        if (atoi(value) <= 0) return fail(DG_E_INVALID, "tail_bwd_persist = 0 (the per-band backward kernel) needs the measurement build");
#endif
        h->tail_bwd_persist = atoi(value);
        return DG_OK;
    }
    if (k == "jobs.slack" || k == "jobs.slots0" || k == "jobs.slots1" || k == "jobs.rate0" || k == "jobs.rate1" ||
        k == "jobs.rate2" || k == "jobs.fixed_us" || k == "jobs.min_level" || k == "jobs.tune" || k == "jobs.xcd_head" || k == "jobs.taper" ||
        k == "jobs.taper_tune" || k == "jobs.prio" || k == "jobs.balance" || k == "jobs.spread" || k == "jobs.pair_kernel") {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        const double v = atof(value);
        if (k == "jobs.slack") h->job_slack = v;
        else if (k == "jobs.slots0") h->job_slots_per_cu[0][0] = (int)v > 0 ? (int)v : 1;
        else if (k == "jobs.slots1") h->job_slots_per_cu[1][0] = (int)v > 0 ? (int)v : 1;
        else if (k == "jobs.min_level") h->job_min_level = (int)v;
        else if (k == "jobs.tune") h->job_tune = v != 0.0;
        else if (k == "jobs.xcd_head") h->job_xcd_head = v;
        else if (k == "jobs.taper") h->job_model.taper = v;
        else if (k == "jobs.taper_tune") h->job_taper_tune = v != 0.0;
        else if (k == "jobs.balance") h->job_balance = v != 0.0;
        else if (k == "jobs.pair_kernel") h->job_pair_kernel = (int)v < 0 ? 0 : ((int)v > 2 ? 2 : (int)v);
        else if (k == "jobs.spread") h->job_spread = v != 0.0;
        else if (k == "jobs.prio") h->job_prio = (int)v < 0 ? 0 : ((int)v > 2 ? 2 : (int)v);
        else if (k == "jobs.fixed_us") { for (auto& f : h->job_model.fixed_us) for (double& x : f) x = v; }
        else { for (auto& r : h->job_model.rate) r[k.back() - '0'] = v > 0 ? v : 1.0; }
        drop_job_lists(h);
        return DG_OK;
    }
    // ---- measurement options: the kernels behind them exist only in the -DDG_MEASURE build of the library
    if (k == "tail_trace" || k == "tail_fwd16" || k == "tail_bwd_bands" || k == "tail_prio" || k == "tail_dbg" || k == "job_trace") {
#ifdef DG_MEASURE
    if (k == "tail_trace") {     // read back with dg_debug_read("tail_trace") (int64 pairs viewed as floats)
        HIP_TRY(hipSetDevice(h->device));
        if (atoi(value)) {
            if (!h->d_tail_trace) HIP_TRY(hipMalloc(&h->d_tail_trace, 4096 * 8 * sizeof(long long)));
            HIP_TRY(hipMemset(h->d_tail_trace, 0, 4096 * 8 * sizeof(long long)));
        } else if (h->d_tail_trace) {
            (void)hipFree(h->d_tail_trace);
            h->d_tail_trace = nullptr;
        }
        return DG_OK;
    }
    if (k == "tail_fwd16") {
        h->tail_fwd16 = atoi(value);
        return DG_OK;
    }
    if (k == "tail_bwd_bands") {
        h->tail_bwd_bands = atoi(value);
        return DG_OK;
    }
    if (k == "tail_prio") {
        const int v = atoi(value);
        if (v < 0 || v > 3) return fail(DG_E_INVALID, "%s: 0..3", key);
        h->tail_prio = v;
        return DG_OK;
    }
    if (k == "tail_dbg") {
        h->tail_dbg = atoi(value);
        return DG_OK;
    }
    if (k == "job_trace") {      // value = op name ("F2"); read back with dg_debug_read("job_trace") (int64 viewed as floats)
        HIP_TRY(hipSetDevice(h->device));
        if (!h->d_job_trace) HIP_TRY(hipMalloc(&h->d_job_trace, (size_t)kJobTraceCap * 4 * sizeof(long long)));
        HIP_TRY(hipMemset(h->d_job_trace, 0, (size_t)kJobTraceCap * 4 * sizeof(long long)));
        h->job_trace_op = value;
        return DG_OK;
    }
#else
        return fail(DG_E_INVALID, "option '%s' needs the measurement build of the library (libdefensegan_hip_measure.so, -DDG_MEASURE)", key);
#endif
    }
    if (k == "debug.poison_pair_counters") {
        // test hook (tests/test_gpu_variants.py): leaves every K-pair arrival counter of every list at `value`, the state a launch
        // that died between the two arrivals of a pair would leave behind.  The next call must not care.
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        int n = 0;
        std::vector<GemmOp*> ops = {&h->F1, &h->B1};
        for (auto& op : h->Fd) ops.push_back(&op);
        for (auto& op : h->Bd) ops.push_back(&op);
        for (GemmOp* op : ops)
            for (auto& jl : op->jobs) {
                if (!jl.d_pair_count || jl.pair_copies <= 0) continue;
                const std::vector<unsigned> v(jl.pair_count_stride * (size_t)jl.pair_copies, (unsigned)atoi(value));
                HIP_TRY(hipMemcpy(jl.d_pair_count, v.data(), v.size() * sizeof(unsigned), hipMemcpyHostToDevice));
                ++n;
            }
        if (!n) return fail(DG_E_STATE, "debug.poison_pair_counters: no prepared job list holds K-pair jobs");
        return DG_OK;
    }
    if (k == "nsplit") {
        const int v = atoi(value);
        if (v < 1 || v > 64 || (h->lin_out / v) % 32 || h->lin_out % v) return fail(DG_E_INVALID, "nsplit must divide lin_out into multiples of 32");
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->nsplit = v;
        free_workspace(h);
        const int rcp = build_lin_packs(h);
        if (rcp) return rcp;
        return rebuild_plans(h);
    }
    return fail(DG_E_INVALID, "unknown option '%s'", key);
}

}  // extern "C"
