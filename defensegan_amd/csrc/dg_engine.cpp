// Host side of the C ABI (include/defensegan_hip.h): handle, weights, workspace, launch sequence.
//
// The launch sequence restates DefenseGANBase.reconstruct (/root/reference/models/gan.py:333-449) as a
// fixed per-iteration kernel chain on one HIP stream:
//   F1 Linear+ReLU -> Fk Deconv+ReLU ... -> tail (last Deconv + sigmoid/tanh + loss [+ its backward])
//   -> Bk Deconv backward (+ReluGrad, in place over the activation) ... -> B1 Linear backward (split-K)
//   -> momentum update.
// The L-th iteration runs the forward half only (the reference discards the L-th update), then the
// per-image first-argmin over restarts gathers the output.  No host round trip inside the loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <vector>

#include "../../include/defensegan_hip.h"
#include "dg_kernels.h"
#include "dg_plan.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(DG_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

struct DeconvSpec {
    const char* name;   // reference layer name
    int cin, cout, h_in, e_used;
    int act;            // 0 relu, 1 none, 2 final (sigmoid / tanh in the tail)
    const char* bn;     // BN layer applied to this layer's output when use_bn ("" = none)
};

// One activation buffer: act[0] = Linear output [N, 16 positions, 4*net_dim]; act[d+1] = output of deconv d.
struct ActInfo {
    int pitch = 0;          // stored positions per spatial dimension
    int valid = 0;          // leading positions that are consumed downstream (7 of 8 after the MNIST crop)
    int C = 0;              // channels
    int64_t row_floats = 0; // floats per latent row
    bool has_bn = false;
    std::string bn_name;
    int bn_C = 0;           // BN columns (4096 features for BN1, channels otherwise)
    int64_t bn_rows = 0;    // BN rows per latent row (1 for BN1, pitch^2 otherwise)
    float* buf = nullptr;
    float* xhat = nullptr;
    float* scale = nullptr;   // [bn_C]
    float* offset = nullptr;  // [bn_C]
    float* fstats = nullptr;  // [2, bn_C]
    float* bstats = nullptr;  // [2, bn_C]
    float* block_sums = nullptr;  // [blocks][2][bn_C]: per-32-row-block column sums left by the producing GEMM's epilogue (EPI_BIAS_STATS)
    int64_t block_cap = 0;        // blocks block_sums holds
};

// Job list of a position-batched launch (dg_gemm.hip), one per row count a layer has been run with.
struct JobList {
    int n_rows = 0, n_jobs = 0, min_level = 0;
    int xcd_order = 0;             // 1 = head of the list re-arranged for XCD locality (dg_plan.h order_for_xcd)
    int snake = 0;                 // 1 = every other round of #CUs jobs reversed (boustrophedon)
    double slack = 0.0;            // the cutting threshold the list was built with (dg_plan.h build_jobs)
    double taper = 0.0;            // JobModel::taper the list was built with
    int prio = 0;                  // 1 = wave priorities by predicted job length (dg_plan.h assign_priorities)
    int pair_kernel = 0;           // 1 = launched as the PAIR instantiation although it holds no pair (dg_plan.h TuneRecord::pair_kernel)
    double xcd_head = 0.0;         // head fraction of the XCD-locality order
    double predicted_us = 0.0;     // simulated makespan of the cost model
    double measured_us = 0.0;      // duration measured when the list was chosen by timing (0 = chosen by the model)
    dg::JobDesc* d_jobs = nullptr;
    // K-pair jobs (dg_types.h): arrival counters and accumulator images, in the SAME allocation behind the job records
    unsigned* d_pair_count = nullptr;
    float* d_pair = nullptr;
    size_t pair_count_stride = 0, pair_stride = 0;   // elements per copy: one copy per concurrent row group (option two_streams)
    int pair_copies = 0;           // copies of the counters / images behind the job records (0 = the list has no pair)
};

// Job list of a fragment-order launch (dg_fgemm.hip): a pure function of (layer plan, row count), no timing
struct FragList {
    int n_rows = 0, n_jobs = 0;
    dg::FragJob* d_jobs = nullptr;
};

struct GemmOp {
    std::string name;
    dg::BatchedPlan bplan;
    std::vector<FragList> fjobs;
    int family = 0;
    dg::ClassDesc* d_cls = nullptr;
    dg::TapEntry* d_btaps = nullptr;
    int* d_pos_a = nullptr;
    int* d_pos_out = nullptr;
    std::vector<JobList> jobs;
    int mode = 0;
    const float* W = nullptr;
    const float* bias = nullptr;
    float* stats = nullptr;   // EPI_BIAS_STATS: the block sums of the activation this layer produces (ActInfo::block_sums)
    int64_t stats_cap = 0;    // blocks that buffer holds
};

constexpr int kJobTraceCap = 65536;

struct ProfEntry {
    std::string name;
    int64_t launches = 0;
    double ms = 0.0;
    double flops = 0.0;   // algorithmic FLOP of the measured launches
};
struct ProfPending {
    int entry;
    int e0, e1;          // indices into dg_handle::prof_events
};

}  // namespace

struct dg_handle {
    int arch = 0, latent = 0, net_dim = 0, use_bn = 0, device = 0;
    int img_h = 0, img_c = 0, P = 0;
    int lin_out = 0;
    std::vector<DeconvSpec> dec;

    // weights (device, engine-owned)
    float* lin_w = nullptr;    // [latent][lin_out]   reference layout; K-contiguous operand of the backward
    float* lin_wt = nullptr;   // [lin_out][latent]   K-contiguous operand of the forward
    float* lin_b = nullptr;
    float* lin_pack_fwd = nullptr; // lin_wt / lin_w in the MFMA fragment order of dg_linear.hip (dg_kernels.h lin_pack_index), or nullptr
    float* lin_pack_bwd = nullptr;
    std::vector<float> lin_w_host; // [latent][lin_out]: the packs are rebuilt when nsplit changes
    std::vector<float*> F, Ft, bias;   // per deconv: [25][cout][cin], [25][cin][cout], [cout]
    std::vector<float*> Fp;            // per non-final deconv: the forward filters in fragment order (dg_fgemm.hip), per tap slab
                                       // [cout / 32][cin / 8][64][4]
    // The fragment-order forward path (round 6; option frag_path, default on; conditions: frag_active()): F1 writes h1 in fragment
    // order + gate bits, every forward deconv runs on dg_fgemm.hip (fragment-order input; output in fragment order, or NHWC for the
    // layer the tail reads), the backward GEMMs take their ReluGrad gates from the bits and write the gradients into the NHWC buffers.
    int frag_path = 0;
    const unsigned* tune_gates = nullptr;   // set while prepare_rows times a backward layer's lists: the gate bits it will run with
    std::vector<float*> actf;          // per activation d < nd - 1: fragment-order buffer (rows padded to 32)
    std::vector<unsigned*> gate;       // per activation d < nd - 1: [rows][row_floats / 32] gate bits
    float* tail_pack = nullptr;        // last deconv's filters in MFMA fragment order (forward tail GEMM)
    float* tail_pack16 = nullptr;      // same, 16x16x4 fragments of the kh-aligned tiles (CelebA forward tail)
    std::map<std::string, bool> have;

    // ops
    GemmOp F1, B1;
    std::vector<GemmOp> Fd, Bd;   // per non-final deconv
    // K slices of the Linear backward (one fixed value for every row count: the slice sums are added in slice order, so the
    // result does not depend on the batch).  Measured 8 vs 16 (MI355X): 2560 rows 977.6 vs 975.5 img/s, 500 rows (the
    // reference's default batch) 756.6 vs 777.9, CelebA 305.2 vs 305.2.
    int nsplit = 16;
    // The latent turn (Linear backward -> update -> Linear forward) on the weight-stationary kernels of dg_linear.hip when the
    // shapes allow (any latent_dim that is a multiple of 32 up to 192 forward; latent_dim 128 and 256-wide K slices backward);
    // 0 = the position-batched kernel as for every other layer.  Bit-identical either way (same fma chains, same K slices).
    int latent_turn = 1;
    int update_fold = 0;           // momentum update folded into the Linear backward launch (dg_linear.hip); needs latent_turn
    unsigned* upd_count = nullptr; // one arrival counter per 32-row block (+ one per row group), zero between launches
    int lin_groups_fwd = 0, lin_groups_bwd = 0;   // workgroups per column tile / K slice; 0 = pick from the CU count
    int cu_count = 256;
    double job_slack = 0.0;        // job cutting threshold (dg_plan.h build_jobs); 0 = pick by simulated makespan
    // Resident workgroups per CU by (family, smallest level in the list) = what LDS admits: 160 KB / (64 | 80, 48, 32 KB of
    // gemm_lds_bytes) = 2, 3, 5.  The kernel's __launch_bounds__(256, 2 / 3 / 4) is the MINIMUM occupancy the register allocator
    // must leave room for, not a cap: the level-2 instantiations use 62-68 VGPRs, so registers admit 7 and LDS decides (5).
    int job_slots_per_cu[2][3] = {{2, 3, 5}, {2, 3, 5}};
    int job_min_level = -1;        // >= 0 forces the starting level of every list (measurement)
    int job_tune = 1;              // 1 = time the candidate job lists on first use of a row count and keep the fastest
    int job_taper_tune = 1;        // 1 = tapered lists (dg_plan.h JobModel::taper) are among the timed candidates
    // Wave priorities by predicted job length (dg_types.h JobDesc::prio): 1 = the best lists of the timing are timed again with
    // priorities and the faster form is kept, 0 = never (default), 2 = every list carries them (measurement, bit-identity tests)
    int job_prio = 0;              // (measured, profiles/r05_ab_prio.txt: the arbiter follows the priorities, the launches last the same)
    int job_spread = 0;            // 1 = the fastest multi-round lists are also timed in spread order (dg_plan.h spread_order); measured
                                   // slower on every layer (profiles/r05_ab_list_orders.txt): off
    // Batchnorm forward statistics from the producing GEMM's epilogue (per-32-row-block column sums, EPI_BIAS_STATS) instead of a
    // pass over the pre-activations; 0 = the separate pass (cross-check)
    int bn_fused = 1;
    // The kernel has two instantiations per (family, epilogue, level): with and without the K-pair hand-off code.  A list without
    // pairs needs neither, and hipcc allocates and schedules their main loops differently: measured on MNIST at 2560 rows the PAIR
    // form is 0.7 % FASTER on Generator.3's backward and 0.6 % on Generator.2's forward, 0.3 % slower on Generator.3's forward
    // (profiles/r05_ab_pair_kernel.txt).  1 = the two fastest lists without pairs are timed on both and the faster form is kept
    // (default), 0 = never, 2 = always.
    int job_pair_kernel = 1;
    int job_balance = 1;           // 1 = lists that fit the resident slots are also offered in balance_order (dg_plan.h)
    // > 0: lists are also offered to the timing in XCD-locality order (dg_plan.h order_for_xcd) with this head fraction.  Off:
    // measured in round 3 (profiles/r03_exp_xcd_order.txt) -- the timing kept it for CelebA's Generator.5 backward only, the
    // launch took the same time (465 vs 466 us), fetched the same bytes across the L2/fabric boundary (907 vs 910 MB raw) and
    // clocked the same: one latent row of that layer's input is 256 KB, so the rows even a row-ordered resident set touches
    // (~50 per XCD) are three times the 4 MB L2 -- the re-reads of the 25-tap pattern are served by the Infinity Cache either way.
    double job_xcd_head = 0.0;
    dg::JobModel job_model;
    long long* d_job_trace = nullptr;
    std::string job_trace_op;
    int tail_dbg = 0;
    int tail_prio = 0;       // wg_priority mode of the CelebA tail launches (dg_device.h): measured, no gain; off
    int tail_bwd_bands = 1;
    int tail_fwd16 = 1;
    int tail_bwd_persist = 512;
    int tail_fwd_split = 512;      // CelebA forward tail (64 channels): workgroups of the role-split persistent kernel, two per CU
                                   // (0 = celeba_tail_fwd16_kernel, which also serves NET_DIM 128)
    int tail_pipe = 256;           // MNIST tail: persistent pipelined kernel, workgroups (0 = fused per-row kernel)
    int tail_pipe_version = 3;     // mnist_tail_pipe3_kernel / _pipe2_ / _pipe_kernel (dg_tail_mnist.hip)
    long long* d_tail_trace = nullptr;   // [4096][8] phase cycle totals, allocated by option tail_trace
    // 0: lr == rec_lr for every step -- what the reference executes (its decay's step variable is never advanced, gan.py:362-386).
    // 1: the schedule the reference's code asks for, exponential_decay(rec_lr, k, ceil(0.8 L), 0.1, staircase) (base_model.py:186-192)
    int lr_intended = 0;
    // number of concurrent row groups (each on its own stream).  Off: measured again in round 3 on one box with 2 .. 8 groups,
    // tuned and whole-tile job lists (profiles/r03_exp_stream_groups.txt): MNIST 2560 rows 966.7 vs 966.6 img/s with 2 groups,
    // slower with 3+ (a queue only gets workgroup slots as the other's kernel retires them, so two MFMA-bound kernels do not
    // overlap beyond their launch ends, and the half-size launches are less efficient); CelebA +0.9 %; 500 rows -13 %.
    int two_streams = 0;
    int two_stream_min_rows = 1024;
    static constexpr int kMaxGroups = 8;
    hipStream_t side_stream[kMaxGroups - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxGroups - 1] = {};

    // workspace
    int64_t cap_rows = 0;
    float *z = nullptr, *m = nullptr, *part = nullptr, *loss = nullptr, *y = nullptr;
    float* xzero = nullptr;        // [P] zeros: stand-in target for dg_generate
    std::vector<ActInfo> ai;       // per activation buffer (sizes, BN parameters)
    std::vector<float*> act;       // act[0] = h1 [N, lin_out]; act[d+1] = output of deconv d (non-final)
    std::vector<int64_t> act_row;  // floats per latent row
    double* bn_part = nullptr;     // BN partial sums scratch
    float* g6 = nullptr;           // CelebA: da6 [N, 64*64*3]
    float* loss_part = nullptr;    // CelebA: [N, 8 bands, 4 waves] partial sums of squared error

    // Replayed graphs of the L-step loop (option graph_max_rows): for call shapes of at most that many latent rows -- where a
    // kernel lasts tens of microseconds and the ~1600 host enqueues of a call are a visible share -- dg_reconstruct captures the
    // loop once per (B, R, L, lr, momentum, schedule) on an internal stream and replays it on the caller's.  Graph nodes hold
    // fixed pointers: the loop reads the call's images from a staging copy (xbuf), and a graph dies with the job lists /
    // workspace it points into (list_epoch).
    struct LoopGraph { int B = 0, R = 0, L = 0; float lr = 0.f, momentum = 0.f; int lr_intended = 0; uint64_t epoch = 0; hipGraphExec_t exec = nullptr; };
    std::vector<LoopGraph> graphs;
    uint64_t list_epoch = 0;
    // OFF by default (0).  Measured in round 4 on the reference's default batch (500 rows): 780.4 img/s replayed vs 779.5 enqueued
    // (profiles/r04_exp_loop_graph.txt) -- the loop is not launch-bound (a kernel lasts 40 us on average) -- and on ROCm 7.2 a graph
    // the CALLER captured of a call on this handle (tests/test_gpu_prepare.py) replays with wrong results once an internal replay
    // has run between its capture and its replay (4 runs in 5; eager launches in between are harmless; tools/graph_interplay_repro.py).
    int graph_max_rows = 0;
    bool graph_broken = false;     // a capture / instantiate failed once: stay on the eager path
    float* xbuf = nullptr;
    int64_t xbuf_floats = 0;
    hipStream_t cap_stream = nullptr;

    // profiling
    int prof_stride = 0;
    std::vector<ProfEntry> prof;
    std::vector<ProfPending> pending;
    std::map<std::string, int> prof_index;
    // Markers of the profiled launches, in stream order.  Consecutive launches SHARE the marker between them (the end of one
    // is the start of the next), so the durations of a profiled step add up to its wall time exactly -- a separate event pair
    // per launch counted every dispatch boundary twice (round 2: the breakdown summed 1.3 % above the timed step).
    std::vector<hipEvent_t> prof_events;
    hipStream_t prof_chain_stream = nullptr;
    int prof_chain_last = -1;      // index of the marker recorded after the previous profiled launch, -1 = chain broken
};

namespace {

int prof_slot(dg_handle* h, const std::string& name) {
    auto it = h->prof_index.find(name);
    if (it != h->prof_index.end()) return it->second;
    h->prof.push_back(ProfEntry{name, 0, 0.0, 0.0});
    h->prof_index[name] = (int)h->prof.size() - 1;
    return (int)h->prof.size() - 1;
}

struct ProfScope {   // brackets one launch with stream markers when sampling is on for this iteration
    dg_handle* h;
    hipStream_t s;
    bool on;
    int entry = -1;
    int i0 = -1;
    double flops;
    static int new_marker(dg_handle* h, hipStream_t s) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        if (hipEventRecord(e, s) != hipSuccess) { (void)hipEventDestroy(e); return -1; }
        h->prof_events.push_back(e);
        return (int)h->prof_events.size() - 1;
    }
    ProfScope(dg_handle* h_, hipStream_t s_, bool on_, const std::string& name, double flops_)
        : h(h_), s(s_), on(on_), flops(flops_) {
        if (!on) return;
        entry = prof_slot(h, name);
        // the marker after the previous profiled launch on this stream is this launch's start
        i0 = (h->prof_chain_last >= 0 && h->prof_chain_stream == s) ? h->prof_chain_last : new_marker(h, s);
        if (i0 < 0) on = false;
    }
    ~ProfScope() {
        if (!on) return;
        const int i1 = new_marker(h, s);
        h->prof_chain_stream = s;
        h->prof_chain_last = i1;
        if (i1 < 0) return;
        h->pending.push_back(ProfPending{entry, i0, i1});
        h->prof[entry].flops += flops;
    }
};

void prof_collect(dg_handle* h) {
    for (auto& p : h->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(h->prof_events[p.e1]) == hipSuccess &&
            hipEventElapsedTime(&ms, h->prof_events[p.e0], h->prof_events[p.e1]) == hipSuccess) {
            h->prof[p.entry].ms += ms;
            h->prof[p.entry].launches += 1;
        }
    }
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    h->prof_events.clear();
    h->pending.clear();
    h->prof_chain_last = -1;
}

// A failed launch (bad configuration, LDS limit, ...) is reported by the layer it happened in, not at the end of the call.
int launch_check(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DG_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return DG_OK;
}

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

// Fills h->ai (geometry of every activation buffer) from the architecture and use_bn.
void describe_activations(dg_handle* h) {
    const int nd = (int)h->dec.size();
    if ((int)h->ai.size() == nd) return;        // geometry depends on (arch, use_bn) only; keep the BN buffers
    h->ai.assign(nd, ActInfo());
    ActInfo& a0 = h->ai[0];
    a0.pitch = 4; a0.valid = 4; a0.C = h->lin_out / 16; a0.row_floats = h->lin_out;
    a0.has_bn = h->use_bn != 0; a0.bn_name = "Generator.BN1"; a0.bn_C = h->lin_out; a0.bn_rows = 1;
    for (int d = 0; d + 1 < nd; ++d) {
        const DeconvSpec& s = h->dec[d];
        ActInfo& a = h->ai[d + 1];
        a.has_bn = h->use_bn && s.bn[0] != 0;
        a.valid = s.e_used;
        // BN statistics cover the full 2h x 2h map; the MNIST crop comes after the ReLU (dataset_models.py:52-59)
        a.pitch = a.has_bn ? 2 * s.h_in : s.e_used;
        a.C = s.cout;
        a.row_floats = (int64_t)a.pitch * a.pitch * a.C;
        a.bn_name = s.bn;
        a.bn_C = s.cout;
        a.bn_rows = (int64_t)a.pitch * a.pitch;
    }
}

int build_plans(dg_handle* h) {
    describe_activations(h);
    {
        GemmOp& op = h->F1;
        op.name = "F1";
        op.mode = h->use_bn ? (h->bn_fused ? dg::EPI_BIAS_STATS : dg::EPI_BIAS) : dg::EPI_BIAS_RELU;
        int rc = upload_batched(op, dg::plan_linear_fwd(h->latent, h->lin_out, h->lin_out));
        if (rc) return rc;
    }
    {
        GemmOp& op = h->B1;
        op.name = "B1";
        op.mode = dg::EPI_STORE;
        int rc = upload_batched(op, dg::plan_linear_bwd(h->latent, h->lin_out, h->nsplit, h->latent));
        if (rc) return rc;
    }
    const int nd = (int)h->dec.size();
    for (auto* v : {&h->Fd, &h->Bd})
        for (auto& o : *v) free_batched(o);
    h->Fd.assign(nd - 1, GemmOp());
    h->Bd.assign(nd - 1, GemmOp());
    for (int d = 0; d + 1 < nd; ++d) {
        const DeconvSpec& s = h->dec[d];
        const ActInfo& in = h->ai[d];
        const ActInfo& out = h->ai[d + 1];
        {
            GemmOp& op = h->Fd[d];
            op.name = std::string("F") + s.name[10];     // "Generator.N" -> "FN"
            op.mode = out.has_bn ? (h->bn_fused ? dg::EPI_BIAS_STATS : dg::EPI_BIAS) : (s.act == 0 ? dg::EPI_BIAS_RELU : dg::EPI_BIAS);
            // with BN every stored position is computed (statistics need the cropped row/column too)
            int rc = upload_batched(op, dg::plan_deconv_fwd(in.valid, in.pitch, out.has_bn ? out.pitch : out.valid, out.pitch,
                                                            s.cin, s.cout, s.cout));
            if (rc) return rc;
        }
        {
            GemmOp& op = h->Bd[d];
            op.name = std::string("B") + s.name[10];
            op.mode = dg::EPI_MASK;      // every backward output lands on a ReLU activation (h1, h2, h3)
            // the incoming gradient is non-zero on the whole stored map after a BN backward, else only on the used block
            dg::LayerPlan base = dg::plan_deconv_bwd(in.valid, in.pitch, out.has_bn ? out.pitch : out.valid, out.pitch, s.cin,
                                                     s.cout, s.cin);
            if (in.pitch > in.valid) dg::plan_add_zero_positions(base, in.valid, in.pitch, s.cin);
            int rc = upload_batched(op, base);
            if (rc) return rc;
        }
    }
    return DG_OK;
}

void free_workspace(dg_handle* h) {
    auto fr = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    fr(h->z); fr(h->m); fr(h->part); fr(h->loss); fr(h->y); fr(h->g6); fr(h->loss_part); fr(h->xbuf);
    h->xbuf_floats = 0;
    ++h->list_epoch;                 // captured loops point into these buffers
    for (auto& a : h->act) fr(a);
    for (auto& a : h->actf) fr(a);
    for (auto& g : h->gate) if (g) { (void)hipFree(g); g = nullptr; }
    for (auto& a : h->ai) { a.buf = nullptr; fr(a.xhat); fr(a.block_sums); a.block_cap = 0; }
    if (h->bn_part) { (void)hipFree(h->bn_part); h->bn_part = nullptr; }
    if (h->upd_count) { (void)hipFree(h->upd_count); h->upd_count = nullptr; }
    h->cap_rows = 0;
}

int ensure_workspace(dg_handle* h, int64_t rows) {
    if (rows <= h->cap_rows) return DG_OK;
    HIP_TRY(hipDeviceSynchronize());
    free_workspace(h);
    const int64_t cap = rows;
    HIP_TRY(hipMalloc(&h->z, cap * h->latent * sizeof(float)));
    HIP_TRY(hipMalloc(&h->m, cap * h->latent * sizeof(float)));
    HIP_TRY(hipMalloc(&h->part, cap * h->nsplit * h->latent * sizeof(float)));
    {
        const size_t n_count = (size_t)(cap / 32 + 16);          // blocks + up to 8 row groups + slack
        HIP_TRY(hipMalloc(&h->upd_count, n_count * sizeof(unsigned)));
        HIP_TRY(hipMemset(h->upd_count, 0, n_count * sizeof(unsigned)));
    }
    HIP_TRY(hipMalloc(&h->loss, cap * sizeof(float)));
    HIP_TRY(hipMalloc(&h->y, cap * h->P * sizeof(float)));
    if (h->graph_max_rows > 0) {
        h->xbuf_floats = std::min<int64_t>(cap, h->graph_max_rows) * h->P;      // B <= B * R rows
        HIP_TRY(hipMalloc(&h->xbuf, (size_t)h->xbuf_floats * sizeof(float)));
    }
    if (h->arch == DG_ARCH_CELEBA64) {
        HIP_TRY(hipMalloc(&h->g6, cap * h->P * sizeof(float)));
        HIP_TRY(hipMalloc(&h->loss_part, cap * 64 * sizeof(float)));
    }
    const int nd = (int)h->dec.size();
    h->act.assign(nd, nullptr);
    h->actf.assign(nd, nullptr);
    h->gate.assign(nd, nullptr);
    h->act_row.assign(nd, 0);
    size_t part_doubles = 0;
    // (rows padded to whole 32-row blocks: the fragment-order kernels compute and store whole blocks)
    const int64_t cap32 = (cap + 31) / 32 * 32;
    for (int d = 0; d < nd; ++d) {
        ActInfo& a = h->ai[d];
        h->act_row[d] = a.row_floats;
        HIP_TRY(hipMalloc(&h->act[d], cap32 * a.row_floats * sizeof(float)));
        if (h->frag_path && !h->use_bn && d + 1 < nd) {
            HIP_TRY(hipMalloc(&h->actf[d], cap32 * a.row_floats * sizeof(float)));
            HIP_TRY(hipMalloc(&h->gate[d], cap32 * (a.row_floats / 32) * sizeof(unsigned)));
        }
        a.buf = h->act[d];
        if (a.has_bn) {
            HIP_TRY(hipMalloc(&a.xhat, cap * a.row_floats * sizeof(float)));
            if (h->bn_fused) {
                // 32-row statistics blocks of the producing GEMM at `cap` rows (dg_plan.h stat_blocks: ceil(rows * positions / 32)
                // per tap class); bn_forward refuses a launch that would need more
                const GemmOp& producer = d == 0 ? h->F1 : h->Fd[(size_t)d - 1];
                const size_t blocks = (size_t)dg::stat_blocks(producer.bplan, (int)std::min<int64_t>(cap, 1 << 24));
                HIP_TRY(hipMalloc(&a.block_sums, blocks * 2 * (size_t)a.bn_C * sizeof(float)));
                a.block_cap = (int64_t)blocks;
            }
            const size_t need = (size_t)dg::bn_max_blocks() * 2 * a.bn_C;
            if (need > part_doubles) part_doubles = need;
        }
    }
    if (part_doubles) HIP_TRY(hipMalloc(&h->bn_part, part_doubles * sizeof(double)));
    h->F1.stats = h->ai[0].block_sums; h->F1.stats_cap = h->ai[0].block_cap;
    for (int d = 0; d + 1 < nd; ++d) { h->Fd[(size_t)d].stats = h->ai[d + 1].block_sums; h->Fd[(size_t)d].stats_cap = h->ai[d + 1].block_cap; }
    h->cap_rows = cap;
    return DG_OK;
}

dg::GemmArgs gemm_args(dg_handle* h, const GemmOp& op, const JobList& jl, const float* A, float* Out, int group = 0) {
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

// F1 / B1 on the weight-stationary kernels (dg_linear.hip)?
bool lin_stationary(const dg_handle* h, const GemmOp& op) {
    if (!h->latent_turn) return false;
    if (&op == &h->F1) return h->lin_pack_fwd != nullptr && dg::lin_stationary_supported(h->latent / 32, op.mode);
    if (&op == &h->B1) return h->lin_pack_bwd != nullptr && dg::lin_stationary_supported(h->lin_out / h->nsplit / 32, op.mode);
    return false;
}

// Does this handle run the fragment-order forward path (dg_fgemm.hip)?  Batchnorm couples rows through statistics passes that
// read NHWC pre-activations; concurrent row groups start at rows that are not multiples of 32; everything else is a matter of
// the layer shapes (the two generators at their default widths qualify).
bool frag_active(const dg_handle* h) {
    if (!h->frag_path || h->use_bn || h->two_streams > 1) return false;
    return lin_stationary(h, h->F1) && h->F1.mode == dg::EPI_BIAS_RELU;
}
bool frag_shapes_ok(const dg_handle* h) {
    const size_t nd = h->dec.size();
    for (size_t d = 0; d + 1 < nd; ++d) {
        if (!h->Fp[d] || !dg::frag_supported(h->Fd[d].bplan)) return false;
        if (h->Fd[d].mode != dg::EPI_BIAS_RELU && h->Fd[d].mode != dg::EPI_BIAS) return false;
        if (h->Fd[d].mode == dg::EPI_BIAS && d + 2 < nd) return false;    // (a layer without ReLU may only feed the tail)
        if (h->ai[d].row_floats % 32) return false;
    }
    return h->lin_out % 128 == 0;
}
bool frag_on(const dg_handle* h) { return frag_active(h) && frag_shapes_ok(h) && h->actf.size() == h->dec.size() && h->actf[0] != nullptr; }

const FragList* find_frag_jobs(const GemmOp& op, int n_rows) {
    for (const auto& fl : op.fjobs)
        if (fl.n_rows == n_rows) return &fl;
    return nullptr;
}

const FragList* get_frag_jobs(GemmOp& op, int n_rows) {
    if (const FragList* have = find_frag_jobs(op, n_rows)) return have;
    const std::vector<dg::FragJob> jobs = dg::build_frag_jobs(op.bplan, n_rows);
    if (jobs.empty()) return nullptr;
    FragList fl;
    fl.n_rows = n_rows;
    fl.n_jobs = (int)jobs.size();
    if (hipMalloc(&fl.d_jobs, jobs.size() * sizeof(dg::FragJob)) != hipSuccess) return nullptr;
    if (hipMemcpy(fl.d_jobs, jobs.data(), jobs.size() * sizeof(dg::FragJob), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(fl.d_jobs);
        return nullptr;
    }
    if (op.fjobs.size() >= 16) { (void)hipFree(op.fjobs.front().d_jobs); op.fjobs.erase(op.fjobs.begin()); }
    op.fjobs.push_back(fl);
    return &op.fjobs.back();
}

// One forward deconv on dg_fgemm.hip: A in fragment order; Out in fragment order (+ gate bits) or NHWC (the layer the tail reads)
int run_frag(dg_handle* h, GemmOp& op, int d, const float* A, float* Out, bool out_frag, unsigned* gates, int n_rows, hipStream_t s, bool prof) {
    const FragList* fl = find_frag_jobs(op, n_rows);
    if (!fl) return fail(DG_E_STATE, "layer %s has no fragment-order job list for %d rows (prepare_rows was skipped)", op.name.c_str(), n_rows);
    dg::FragArgs a;
    a.A = A;
    a.Wp = h->Fp[(size_t)d];
    a.Out = Out;
    a.bias = op.bias;
    a.gate_bits = gates;
    a.jobs = fl->d_jobs;
    a.taps = op.d_btaps;
    a.a_rowstride = op.bplan.a_rowstride;
    a.out_rowstride = op.bplan.out_rowstride;
    a.gate_words = (int)(op.bplan.out_rowstride / 32);
    a.kch = op.bplan.kch;
    a.kc8_log2 = op.bplan.kch == 64 ? 3 : (op.bplan.kch == 128 ? 4 : 5);
    a.mode = op.mode;
    a.out_frag = out_frag ? 1 : 0;
    a.n_jobs = fl->n_jobs;
    char sym[64];
    snprintf(sym, sizeof sym, "@fgemm_kernel<2, 4, %d, %s>", op.mode, out_frag ? "true" : "false");
    {
        ProfScope ps(h, s, prof, op.name + sym, 2.0 * (double)op.bplan.macs_per_row * n_rows);
        dg::launch_fgemm(a, s);
    }
    return launch_check(op.name.c_str());
}

// Fragment-order copies of the Linear weights for dg_linear.hip (forward: 128-feature column tiles of W^T; backward: the
// nsplit K slices of W, all 128 latent columns each), built from the host copy kept by dg_set_weights.
int build_lin_packs(dg_handle* h) {
    auto fr = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    HIP_TRY(hipDeviceSynchronize());
    fr(h->lin_pack_fwd); fr(h->lin_pack_bwd);
    ++h->list_epoch;
    if (h->lin_w_host.empty()) return DG_OK;
    const int K = h->latent, F = h->lin_out;
    const float* W = h->lin_w_host.data();               // [K][F]
    std::vector<float> pk((size_t)K * F);
    if (F % 128 == 0 && K % 32 == 0 && dg::lin_stationary_supported(K / 32, dg::EPI_BIAS_RELU)) {
        const int kch = K / 32;
        for (int u = 0; u < F / 128; ++u)
            for (int w = 0; w < 4; ++w)
                for (int c = 0; c < kch; ++c)
                    for (int kk = 0; kk < 4; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int f = u * 128 + w * 32 + (lane & 31), k = c * 32 + (kk * 2 + (lane >> 5)) * 4 + e;
                                pk[(size_t)dg::lin_pack_index(u, w, kch, c, kk, lane, e)] = W[(size_t)k * F + f];      // W^T[f][k]
                            }
        HIP_TRY(hipMalloc(&h->lin_pack_fwd, pk.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->lin_pack_fwd, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (K == 128 && F % (h->nsplit * 32) == 0 && dg::lin_stationary_supported(F / h->nsplit / 32, dg::EPI_STORE)) {
        const int ks = F / h->nsplit, kch = ks / 32;
        for (int s = 0; s < h->nsplit; ++s)
            for (int w = 0; w < 4; ++w)
                for (int c = 0; c < kch; ++c)
                    for (int kk = 0; kk < 4; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int d = w * 32 + (lane & 31), f = s * ks + c * 32 + (kk * 2 + (lane >> 5)) * 4 + e;
                                pk[(size_t)dg::lin_pack_index(s, w, kch, c, kk, lane, e)] = W[(size_t)d * F + f];
                            }
        HIP_TRY(hipMalloc(&h->lin_pack_bwd, pk.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->lin_pack_bwd, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return DG_OK;
}

// The momentum update riding in the Linear backward launch: the rows' z / m, their arrival counters and the step's constants.
struct UpdateFold {
    float* z; float* m; unsigned* count; float lr, momentum;
};

int run_lin_stationary(dg_handle* h, GemmOp& op, const float* A, float* Out, int n_rows, hipStream_t s, bool prof,
                       const UpdateFold* uf = nullptr, float* out_frag = nullptr, unsigned* gates = nullptr) {
    const bool fwd = &op == &h->F1;
    dg::LinArgs a;
    a.out_frag = fwd ? out_frag : nullptr;
    a.gate_bits = fwd ? gates : nullptr;
    a.gate_words = h->lin_out / 32;
    a.upd_z = nullptr; a.upd_m = nullptr; a.upd_count = nullptr; a.upd_lr = 0.f; a.upd_momentum = 0.f;
    if (uf && !fwd) { a.upd_z = uf->z; a.upd_m = uf->m; a.upd_count = uf->count; a.upd_lr = uf->lr; a.upd_momentum = uf->momentum; }
    a.A = A;
    a.Wp = fwd ? h->lin_pack_fwd : h->lin_pack_bwd;
    a.Out = Out;
    a.bias = op.bias;
    a.n_rows = n_rows;
    a.mode = op.mode;
    const int n_blocks = (n_rows + 31) / 32;
    int want;
    if (fwd) {
        a.a_rowstride = h->latent; a.a_unit = 0;
        a.out_rowstride = h->lin_out; a.out_unit = 128;
        a.units = h->lin_out / 128;
        a.kch = h->latent / 32;
        want = h->lin_groups_fwd > 0 ? h->lin_groups_fwd : std::max(1, 2 * h->cu_count / a.units);     // two workgroups per CU
    } else {
        const int ks = h->lin_out / h->nsplit;
        a.a_rowstride = h->lin_out; a.a_unit = ks;
        a.out_rowstride = (long long)h->nsplit * h->latent; a.out_unit = h->latent;
        a.units = h->nsplit;
        a.kch = ks / 32;
        want = h->lin_groups_bwd > 0 ? h->lin_groups_bwd : std::max(1, h->cu_count / a.units);         // one workgroup per CU
    }
    a.groups = std::min(n_blocks, want);
#ifdef DG_MEASURE
    a.trace = (h->d_job_trace && op.name == h->job_trace_op && a.units * a.groups * 2 <= kJobTraceCap) ? h->d_job_trace : nullptr;
#endif
    char sym[64];
    snprintf(sym, sizeof sym, a.upd_z ? "@lin_stationary_kernel<%d, %d, true>" : a.out_frag ? "@lin_stationary_kernel<%d, %d, false, true>" : "@lin_stationary_kernel<%d, %d>", a.kch, a.mode);
    {
        ProfScope ps(h, s, prof, op.name + sym, 2.0 * (double)op.bplan.macs_per_row * n_rows);
        dg::launch_lin_stationary(a, s);
    }
    return launch_check(op.name.c_str());
}

// One GEMM layer: one launch with the job list prepare_rows() left for this row count.  Nothing here allocates or waits.
int run_gemm(dg_handle* h, GemmOp& op, const float* A, float* Out, int n_rows, hipStream_t s, bool prof, const unsigned* gates = nullptr) {
    if (lin_stationary(h, op)) return run_lin_stationary(h, op, A, Out, n_rows, s, prof);
    const JobList* jl = find_jobs(op, n_rows);
    if (!jl) return fail(DG_E_STATE, "layer %s has no job list for %d rows (prepare_rows was skipped)", op.name.c_str(), n_rows);
    if (op.mode == dg::EPI_BIAS_STATS && (!op.stats || dg::stat_blocks(op.bplan, n_rows) > op.stats_cap))
        return fail(DG_E_STATE, "layer %s: %lld statistics blocks at %d rows, the buffer holds %lld", op.name.c_str(),
                    (long long)dg::stat_blocks(op.bplan, n_rows), n_rows, (long long)op.stats_cap);
    int group = 0;                                   // the row group launching: its own copy of the list's pair scratch
    for (int i = 0; i < dg_handle::kMaxGroups - 1; ++i)
        if (s == h->side_stream[i] && s != nullptr) group = i + 1;
    dg::GemmArgs a = gemm_args(h, op, *jl, A, Out, group);
    if (gates) {                                      // fragment-order path: ReluGrad from the gate bits, Out holds gradients only
        a.mode = dg::EPI_MASK_BITS;
        a.gate_bits = gates;
        a.gate_words = (int)(op.bplan.out_rowstride / 32);
    }
    char sym[64];
    snprintf(sym, sizeof sym, "@gemm_batched_kernel<%d, %d, %d, %s>", op.family, a.mode, jl->min_level, a.pair_scratch ? "true" : "false");
    {
        ProfScope ps(h, s, prof, op.name + sym, 2.0 * (double)op.bplan.macs_per_row * n_rows);
        dg::launch_gemm(op.family, a, s);
    }
    return launch_check(op.name.c_str());
}

// A row group = a contiguous range of latent rows (whole images) processed on one stream.  Rows are independent
// (use_bn = False), so a batch may be split into groups that run concurrently on two streams: one group's
// VALU/LDS-bound tail and short kernels then overlap the other group's MFMA-bound GEMMs.
struct RowGroup {
    int row0 = 0, n_rows = 0;
    hipStream_t s = nullptr;
};

// Row groups of a call over B images x R restarts (whole images per group; the streams are assigned by the caller).
int split_groups(const dg_handle* h, int B, int R, RowGroup* grp) {
    const int n_rows = B * R;
    int ngroups = 1;
    grp[0].row0 = 0; grp[0].n_rows = n_rows;
    if (h->two_streams > 1 && !h->use_bn && n_rows >= h->two_stream_min_rows && h->prof_stride == 0) {
        ngroups = h->two_streams < B ? h->two_streams : B;
        if (ngroups > dg_handle::kMaxGroups) ngroups = dg_handle::kMaxGroups;
        int b_done = 0;
        for (int gi = 0; gi < ngroups; ++gi) {
            const int nb = (B - b_done + (ngroups - gi) - 1) / (ngroups - gi);     // images of this group
            grp[gi].row0 = b_done * R;
            grp[gi].n_rows = nb * R;
            b_done += nb;
        }
    }
    return ngroups;
}

// Everything a call needs before its first kernel: workspace for `cap_rows` latent rows and, for each row count in
// `rows[0..n)`, the job list of every GEMM layer (built and, with jobs.tune, timed on the layer's own buffers).  This is the
// only place on the compute path that allocates device memory or waits for the device; once it has run for a row count,
// calls with that row count only enqueue kernels.  A stream that is being captured cannot be prepared on.
int prepare_rows(dg_handle* h, int64_t cap_rows, const int* rows, int n, hipStream_t s) {
    bool missing = cap_rows > h->cap_rows;
    const int nd = (int)h->dec.size();
    auto each_op = [&](auto&& fn) -> int {
        int rc = fn(h->F1, 0, -1, 0);                 // (op, kind, deconv index)
        for (int d = 0; !rc && d + 1 < nd; ++d) rc = fn(h->Fd[d], 1, d, 0);
        for (int d = nd - 2; !rc && d >= 0; --d) rc = fn(h->Bd[d], 2, d, 0);
        if (!rc) rc = fn(h->B1, 3, -1, 0);
        return rc;
    };
    for (int i = 0; i < n && !missing; ++i)
        each_op([&](GemmOp& op, int kind, int, int) {
            if (lin_stationary(h, op)) return 0;
            if (kind == 1 && frag_on(h) ? !find_frag_jobs(op, rows[i]) : !find_jobs(op, rows[i])) missing = true;
            return 0;
        });
    if (!missing) return DG_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    // (an error from the query -- the legacy NULL stream while another stream captures in global mode -- is treated as "capturing":
    // what follows synchronises the device and allocates)
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return fail(DG_E_STATE, "this call shape has not been prepared and the stream is being captured: call dg_prepare(B, R) "
                                "before the capture (it allocates workspace and times the job lists)");
    }
    // One handle serves ONE stream at a time (include/defensegan_hip.h): the timing launches below write the handle's own
    // activation buffers, so whatever an earlier call left queued on another stream must have finished first
    HIP_TRY(hipDeviceSynchronize());
    int rc = ensure_workspace(h, cap_rows);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const int nr = rows[i];
        rc = each_op([&](GemmOp& op, int kind, int d, int) {
            const float* A = kind == 0 ? h->z : kind == 1 ? h->act[d] : kind == 2 ? h->act[d + 1] : h->act[0];
            float* Out = kind == 0 ? h->act[0] : kind == 1 ? h->act[d + 1] : kind == 2 ? h->act[d] : h->part;
            if (lin_stationary(h, op)) return (int)DG_OK;          // no job list: dg_linear.hip derives its grid from the row count
            if (kind == 1 && frag_on(h)) {                         // forward deconv on dg_fgemm.hip: its own kind of list, not timed
                if (!get_frag_jobs(op, nr)) return fail(DG_E_NOMEM, "cannot build the fragment-order job list of layer %s for %d rows", op.name.c_str(), nr);
                return (int)DG_OK;
            }
            h->tune_gates = (kind == 2 && frag_on(h)) ? h->gate[(size_t)d] : nullptr;
            const JobList* got = get_jobs(h, op, nr, A, Out, s);
            h->tune_gates = nullptr;
            if (!got) return fail(DG_E_NOMEM, "cannot build the job list of layer %s for %d rows", op.name.c_str(), nr);
            return (int)DG_OK;
        });
        if (rc) return rc;
    }
    return DG_OK;
}

// prepare_rows for a projection call of B images x R restarts: its row groups' sizes
int prepare_call(dg_handle* h, int B, int R, hipStream_t s) {
    RowGroup grp[dg_handle::kMaxGroups];
    const int ng = split_groups(h, B, R, grp);
    int rows[dg_handle::kMaxGroups];
    for (int gi = 0; gi < ng; ++gi) rows[gi] = grp[gi].n_rows;
    return prepare_rows(h, (int64_t)B * R, rows, ng, s);
}

dg::BnArgs bn_args(dg_handle* h, const ActInfo& a, int n_rows) {
    dg::BnArgs b;
    b.a = a.buf;
    b.xhat = a.xhat;
    b.part = h->bn_part;
    b.fstats = a.fstats;
    b.bstats = a.bstats;
    b.scale = a.scale;
    b.offset = a.offset;
    b.rows = (int64_t)n_rows * a.bn_rows;
    b.C = a.bn_C;
    return b;
}

// forward chain at the current h->z; fills activations, loss, (y when want_y); when tail_backward the tail also
// leaves the gradient w.r.t. the last GEMM activation in place.  x points at image 0 of the CALL (row0 / R
// images are skipped inside).  With use_bn the whole call is one row group (batch statistics couple all rows).
// want_loss: the per-row loss is only read after the last step (selection) and by dg_loss_grad; the CelebA tail leaves
// per-band partial sums, whose reduction is skipped when nobody reads the result.
int run_forward(dg_handle* h, const float* x, const RowGroup& g, int R, bool want_y, bool want_loss, bool tail_backward, bool prof) {
    const int n_rows = g.n_rows;
    hipStream_t s = g.s;
    const int64_t r0 = g.row0;
    // a layer with Batchnorm behind it leaves its pre-activations in the layer's own buffer (ActInfo::xhat): the BN pass writes
    // the activation
    auto out_of = [&](int d) { return (h->ai[d].has_bn ? h->ai[d].xhat : h->act[d]) + r0 * h->act_row[d]; };
    const bool frag = frag_on(h) && r0 == 0;
    int rc = frag ? run_lin_stationary(h, h->F1, h->z, h->act[0], n_rows, s, prof, nullptr, h->actf[0], h->gate[0])
                  : run_gemm(h, h->F1, h->z + r0 * h->latent, out_of(0), n_rows, s, prof);
    if (rc) return rc;
    auto bn_forward = [&](int d, const GemmOp& producer) {
        ProfScope ps(h, s, prof, "BNf", 0.0);
        if (producer.mode == dg::EPI_BIAS_STATS)
            dg::launch_bn_forward_from_blocks(bn_args(h, h->ai[d], n_rows), h->ai[d].block_sums, (int)dg::stat_blocks(producer.bplan, n_rows), 1, s,
                                              producer.bias);
        else
            dg::launch_bn_forward(bn_args(h, h->ai[d], n_rows), 1, s);
    };
    if (h->ai[0].has_bn) bn_forward(0, h->F1);
    const int nd = (int)h->dec.size();
    for (int d = 0; d + 1 < nd; ++d) {
        if (frag) {
            // the last of these layers feeds the tail, which reads NHWC; the others feed the next fragment-order layer
            const bool to_tail = d + 2 == nd;
            rc = run_frag(h, h->Fd[d], d, h->actf[d], to_tail ? h->act[d + 1] : h->actf[d + 1], !to_tail, to_tail ? nullptr : h->gate[d + 1],
                          n_rows, s, prof);
            if (rc) return rc;
            continue;
        }
        rc = run_gemm(h, h->Fd[d], h->act[d] + r0 * h->act_row[d], out_of(d + 1), n_rows, s, prof);
        if (rc) return rc;
        if (h->ai[d + 1].has_bn) bn_forward(d + 1, h->Fd[d]);
    }
    const DeconvSpec& last = h->dec[nd - 1];
    if (h->arch == DG_ARCH_MNIST28) {
        dg::MnistTailArgs t;
        t.h3 = h->act[nd - 1] + r0 * h->act_row[nd - 1];
        t.F5 = h->F[nd - 1];
        t.b5 = h->bias[nd - 1];
        t.x = x + (r0 / R) * h->P;
        t.loss = h->loss + r0;
        t.y = want_y ? h->y + r0 * h->P : nullptr;
        t.n_rows = n_rows;
        t.R = R;
        t.C = last.cin;
        t.do_backward = tail_backward ? 1 : 0;
        t.pipe = h->tail_pipe;
        t.want_loss = want_loss ? 1 : 0;
        // the third-generation kernel writes neither the per-row loss nor y: a launch whose loss or image is read (dg_loss_grad
        // with out_loss and / or out_y) runs the first
        t.pipe_version = (h->tail_pipe_version == 3 && (want_loss || want_y)) ? 1 : h->tail_pipe_version;
#ifdef DG_MEASURE
        t.dbg = h->tail_dbg;
        t.trace = h->d_tail_trace;
#endif
        const double macs = 67.0 * 67.0 * last.cin;   // valid taps 14 -> 28
        const bool piped = t.pipe > 0 && tail_backward && t.C == 64 && n_rows >= 2 * t.pipe;   // launch_mnist_tail_mfma
        ProfScope ps(h, s, prof, !tail_backward ? "T5f@mnist_tail_mfma_kernel" : piped ? (t.pipe_version == 3 ? "T5fb@mnist_tail_pipe3_kernel" : t.pipe_version == 2 ? "T5fb@mnist_tail_pipe2_kernel" : "T5fb@mnist_tail_pipe_kernel") : "T5fb@mnist_tail_mfma_kernel",
                     (tail_backward ? 4.0 : 2.0) * macs * n_rows);
        dg::launch_mnist_tail_mfma(t, s);
    }
    if (h->arch == DG_ARCH_MNIST28) {
        const int rc2 = launch_check("the MNIST tail (Generator.5 + loss)");
        if (rc2) return rc2;
    } else {
        dg::CelebaTailArgs t;
        t.h5 = h->act[nd - 1] + r0 * h->act_row[nd - 1];
        t.F6 = h->F[nd - 1];
        t.F6p = h->tail_pack16;
        t.bwd_persist = h->tail_bwd_persist;
        t.fwd_split = (last.cin == 64) ? h->tail_fwd_split : 0;
        t.want_loss = want_loss ? 1 : 0;
#ifdef DG_MEASURE
        t.F6p = h->tail_fwd16 ? h->tail_pack16 : h->tail_pack;
        t.fwd16 = h->tail_fwd16;
        if (!h->tail_fwd16) t.fwd_split = 0;          // the 32-wide cross-check kernel was asked for
        t.trace = h->d_tail_trace;
        t.dbg = h->tail_dbg;
        t.bwd_bands = h->tail_bwd_bands;
        t.prio = h->tail_prio;
#endif
        t.b6 = h->bias[nd - 1];
        t.x = x + (r0 / R) * h->P;
        t.loss_part = h->loss_part + r0 * 64;
        t.y = want_y ? h->y + r0 * h->P : nullptr;
        t.g6 = h->g6 + r0 * h->P;
        t.n_rows = n_rows;
        t.R = R;
        t.C = last.cin;
        t.do_backward = tail_backward ? 1 : 0;
        const double macs = 157.0 * 157.0 * last.cin * 3.0;   // valid taps 32 -> 64
        {
#ifdef DG_MEASURE
            ProfScope ps(h, s, prof, h->tail_fwd16 ? "T6f@celeba_tail_fwd16_kernel" : "T6f@celeba_tail_fwd_mfma_kernel", 2.0 * macs * n_rows);
#else
            ProfScope ps(h, s, prof, t.fwd_split > 0 ? "T6f@celeba_tail_fwd_split_kernel" : "T6f@celeba_tail_fwd16_kernel", 2.0 * macs * n_rows);
#endif
            dg::launch_celeba_tail_fwd_mfma(t, s);
        }
        if (want_loss) dg::launch_celeba_loss_finish(t.loss_part, h->loss + r0, n_rows, t.fwd_split > 0 ? 16 : 8, h->P, s);
        if (tail_backward) {
#ifdef DG_MEASURE
            ProfScope ps(h, s, prof, h->tail_bwd_persist > 0 ? "T6b@celeba_tail_bwd_persist_kernel" : "T6b@celeba_tail_bwd_mfma_kernel", 2.0 * macs * n_rows);
#else
            ProfScope ps(h, s, prof, "T6b@celeba_tail_bwd_persist_kernel", 2.0 * macs * n_rows);
#endif
            dg::launch_celeba_tail_bwd_mfma(t, s);
        }
        const int rc2 = launch_check("the CelebA tail (Generator.6 + loss)");
        if (rc2) return rc2;
    }
    return DG_OK;
}

bool update_folds(const dg_handle* h) {
    return h->update_fold && h->upd_count && lin_stationary(h, h->B1) && dg::lin_fold_supported(h->nsplit, h->latent);
}

int run_backward(dg_handle* h, const RowGroup& g, bool prof, const UpdateFold* uf = nullptr) {
    const int nd = (int)h->dec.size();
    const int64_t r0 = g.row0;
    for (int d = nd - 2; d >= 0; --d) {
        if (h->ai[d + 1].has_bn) {
            ProfScope ps(h, g.s, prof, "BNb", 0.0);
            dg::launch_bn_backward(bn_args(h, h->ai[d + 1], g.n_rows), g.s);
        }
        int rc = run_gemm(h, h->Bd[d], h->act[d + 1] + r0 * h->act_row[d + 1], h->act[d] + r0 * h->act_row[d], g.n_rows, g.s, prof,
                          frag_on(h) && r0 == 0 ? h->gate[d] : nullptr);
        if (rc) return rc;
    }
    if (h->ai[0].has_bn) {
        ProfScope ps(h, g.s, prof, "BNb", 0.0);
        dg::launch_bn_backward(bn_args(h, h->ai[0], g.n_rows), g.s);
    }
    if (uf) return run_lin_stationary(h, h->B1, h->act[0] + r0 * h->act_row[0], h->part + r0 * h->nsplit * h->latent, g.n_rows, g.s, prof, uf);
    return run_gemm(h, h->B1, h->act[0] + r0 * h->act_row[0], h->part + r0 * h->nsplit * h->latent, g.n_rows, g.s, prof);
}

// (Re)builds every layer plan and re-attaches the weight pointers (dg_create, tuning options).
int rebuild_plans(dg_handle* h) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    int rc = build_plans(h);
    if (rc) return rc;
    const size_t ndec = h->dec.size();
    h->F1.W = h->lin_wt; h->F1.bias = h->lin_b;
    h->F1.stats = h->ai.empty() ? nullptr : h->ai[0].block_sums;
    h->F1.stats_cap = h->ai.empty() ? 0 : h->ai[0].block_cap;
    h->B1.W = h->lin_w;
    for (size_t d = 0; d + 1 < ndec; ++d) {
        h->Fd[d].W = h->F[d]; h->Fd[d].bias = h->bias[d];
        h->Fd[d].stats = h->ai[d + 1].block_sums;
        h->Fd[d].stats_cap = h->ai[d + 1].block_cap;
        h->Bd[d].W = h->Ft[d];
    }
    return DG_OK;
}

// The L-step loop of DefenseGANBase.reconstruct (gan.py:409-437) as launches on the row groups' streams: L forwards, L - 1
// backward + update (the reference's L-th update is dead work), the last forward also leaves y and the per-row loss.
int enqueue_steps(dg_handle* h, const float* x, int R, int L, float lr, float momentum, const RowGroup* grp, int ngroups) {
    const int steps = L > 1 ? L : 1;
    const int decay_iter = L > 0 ? (int)std::ceil(0.8 * (double)L) : 1;
    for (int k = 0; k < steps; ++k) {
        const bool last = (k == steps - 1);
        const bool prof = h->prof_stride > 0 && (k % h->prof_stride) == 0;
        // a step that is not sampled breaks the marker chain: the next sampled launch starts from its own marker, not from the
        // one recorded after the last sampled step (which would charge it with every skipped step in between)
        if (!prof) h->prof_chain_last = -1;
        const float lr_k = h->lr_intended ? lr * std::pow(0.1f, (float)(k / decay_iter)) : lr;
        for (int gi = 0; gi < ngroups; ++gi) {
            const RowGroup& g = grp[gi];
            int rc = run_forward(h, x, g, R, /*want_y=*/last, /*want_loss=*/last, /*tail_backward=*/!last, prof);
            if (rc) return rc;
            if (last) continue;
            const int64_t r0 = g.row0;
            if (update_folds(h)) {
                // row groups are whole images, not whole 32-row blocks: group gi's counters start at r0 / 32 + gi (disjoint)
                const UpdateFold uf = {h->z + r0 * h->latent, h->m + r0 * h->latent, h->upd_count + r0 / 32 + gi, lr_k, momentum};
                rc = run_backward(h, g, prof, &uf);
                if (rc) return rc;
                continue;
            }
            rc = run_backward(h, g, prof);
            if (rc) return rc;
            ProfScope ps(h, g.s, prof, "UPD@momentum_update_kernel", 0.0);
            dg::launch_momentum_update(h->z + r0 * h->latent, h->m + r0 * h->latent, h->part + r0 * h->nsplit * h->latent,
                                       h->nsplit, g.n_rows, h->latent, lr_k, momentum, nullptr, g.s);
        }
    }
    return DG_OK;
}

// Graph of enqueue_steps for one call shape, reading the images from h->xbuf; nullptr = use the eager path.
hipGraphExec_t loop_graph(dg_handle* h, int B, int R, int L, float lr, float momentum) {
    for (auto it = h->graphs.begin(); it != h->graphs.end();) {
        if (it->epoch != h->list_epoch) { (void)hipGraphExecDestroy(it->exec); it = h->graphs.erase(it); continue; }
        if (it->B == B && it->R == R && it->L == L && it->lr == lr && it->momentum == momentum && it->lr_intended == h->lr_intended)
            return it->exec;
        ++it;
    }
    if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) { h->graph_broken = true; return nullptr; }
    RowGroup g;
    g.row0 = 0; g.n_rows = B * R; g.s = h->cap_stream;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (ok) {
        const int rc = enqueue_steps(h, h->xbuf, R, L, lr, momentum, &g, 1);
        ok = hipStreamEndCapture(h->cap_stream, &graph) == hipSuccess && rc == DG_OK && graph != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();
        h->graph_broken = true;
        return nullptr;
    }
    if (h->graphs.size() >= 8) { (void)hipGraphExecDestroy(h->graphs.front().exec); h->graphs.erase(h->graphs.begin()); }
    dg_handle::LoopGraph lg;
    lg.B = B; lg.R = R; lg.L = L; lg.lr = lr; lg.momentum = momentum; lg.lr_intended = h->lr_intended; lg.epoch = h->list_epoch; lg.exec = exec;
    h->graphs.push_back(lg);
    return exec;
}

void drop_graphs(dg_handle* h) {
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

// One call processes at most 2^24 latent rows (32-bit tile arithmetic in the launchers).
int check_rows(int B, int R, int* n_rows) {
    if (B < 1 || R < 1) return fail(DG_E_INVALID, "need B >= 1 and R >= 1 (got %d, %d)", B, R);
    const int64_t rows64 = (int64_t)B * R;
    if (rows64 > (1 << 24)) return fail(DG_E_INVALID, "B*R = %lld is too large for one call", (long long)rows64);
    *n_rows = (int)rows64;
    return DG_OK;
}

int check_ready(dg_handle* h) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    if (!dg_weights_complete(h)) return fail(DG_E_STATE, "generator weights are not completely set (dg_set_weights)");
    return DG_OK;
}

}  // namespace

extern "C" {

int dg_version(void) { return DG_ABI_VERSION; }
const char* dg_last_error(void) { return g_err.c_str(); }

// internal: lets the other translation units of the library (dg_clf.hip) report through dg_last_error()
__attribute__((visibility("hidden"))) void dg_set_error_message(const char* msg) { g_err = msg ? msg : ""; }

int dg_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail(DG_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return DG_E_HIP; }
    return n;
}

int dg_device_info(int device, char* name, int name_len, int* cu_count, int64_t* hbm_bytes) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return DG_OK;
}

int dg_create(int arch, int latent_dim, int net_dim, int use_bn, int device, dg_handle** out) {
    if (!out) return fail(DG_E_INVALID, "out is null");
    *out = nullptr;
    if (arch != DG_ARCH_MNIST28 && arch != DG_ARCH_CELEBA64) return fail(DG_E_INVALID, "unknown arch %d", arch);
    if (latent_dim <= 0 || latent_dim % 64) return fail(DG_E_INVALID, "latent_dim must be a positive multiple of 64 (got %d)", latent_dim);
    if (net_dim <= 0 || net_dim % 64 || net_dim > 128) return fail(DG_E_INVALID, "net_dim must be 64 or 128 (got %d)", net_dim);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DG_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    dg_handle* h = new dg_handle();
    h->arch = arch;
    h->latent = latent_dim;
    h->net_dim = net_dim;
    h->use_bn = use_bn;
    h->device = device;
    h->lin_out = 4 * 4 * 4 * net_dim;
    (void)hipDeviceGetAttribute(&h->cu_count, hipDeviceAttributeMultiprocessorCount, device);
    if (h->cu_count <= 0) h->cu_count = 256;
    const int nd = net_dim;
    if (arch == DG_ARCH_MNIST28) {
        h->img_h = 28; h->img_c = 1;
        h->dec = {{"Generator.2", 4 * nd, 2 * nd, 4, 7, 0, "Generator.BN2"}, {"Generator.3", 2 * nd, nd, 7, 14, 0, "Generator.BN3"},
                  {"Generator.5", nd, 1, 14, 28, 2, ""}};
    } else {
        h->img_h = 64; h->img_c = 3;
        h->dec = {{"Generator.2", 4 * nd, 2 * nd, 4, 8, 0, "Generator.BN2"}, {"Generator.3", 2 * nd, nd, 8, 16, 0, "Generator.BN3"},
                  {"Generator.5", nd, nd, 16, 32, 1, ""}, {"Generator.6", nd, 3, 32, 64, 2, ""}};
    }
    h->P = h->img_h * h->img_h * h->img_c;
    const size_t ndec = h->dec.size();
    h->F.assign(ndec, nullptr);
    h->Fp.assign(ndec, nullptr);
    h->Ft.assign(ndec, nullptr);
    h->bias.assign(ndec, nullptr);
    hipError_t e = hipSuccess;
    auto dmalloc = [&](float** p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, n * sizeof(float)); };
    dmalloc(&h->lin_w, (size_t)h->latent * h->lin_out);
    dmalloc(&h->lin_wt, (size_t)h->latent * h->lin_out);
    dmalloc(&h->lin_b, h->lin_out);
    dmalloc(&h->xzero, h->P);
    for (size_t d = 0; d < ndec; ++d) {
        dmalloc(&h->F[d], (size_t)25 * h->dec[d].cout * h->dec[d].cin);
        dmalloc(&h->Ft[d], (size_t)25 * h->dec[d].cout * h->dec[d].cin);
        dmalloc(&h->bias[d], h->dec[d].cout);
    }
    if (e == hipSuccess) e = hipMemset(h->xzero, 0, (size_t)h->P * sizeof(float));
    if (e != hipSuccess) { dg_destroy(h); return fail(DG_E_NOMEM, "hipMalloc(weights): %s", hipGetErrorString(e)); }
    bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < dg_handle::kMaxGroups - 1; ++i)
        ok = hipStreamCreateWithFlags(&h->side_stream[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        dg_destroy(h);
        return fail(DG_E_HIP, "cannot create the side streams / events");
    }
    int rc = rebuild_plans(h);
    if (rc) { dg_destroy(h); return rc; }
    for (auto& a : h->ai) {
        if (!a.has_bn) continue;
        hipError_t e2 = hipSuccess;
        for (float** pp : {&a.scale, &a.offset}) if (e2 == hipSuccess) e2 = hipMalloc(pp, (size_t)a.bn_C * sizeof(float));
        for (float** pp : {&a.fstats, &a.bstats}) if (e2 == hipSuccess) e2 = hipMalloc(pp, (size_t)2 * a.bn_C * sizeof(float));
        if (e2 != hipSuccess) { dg_destroy(h); return fail(DG_E_NOMEM, "hipMalloc(BN parameters): %s", hipGetErrorString(e2)); }
    }
    *out = h;
    return DG_OK;
}

int dg_destroy(dg_handle* h) {
    if (!h) return DG_OK;
    (void)hipSetDevice(h->device);
    prof_collect(h);
    drop_graphs(h);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    free_workspace(h);
    auto fr = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    for (auto& a : h->ai) { fr(a.scale); fr(a.offset); fr(a.fstats); fr(a.bstats); }
    fr(h->lin_w); fr(h->lin_wt); fr(h->lin_b); fr(h->lin_pack_fwd); fr(h->lin_pack_bwd); fr(h->xzero); fr(h->tail_pack); fr(h->tail_pack16);
    if (h->d_tail_trace) (void)hipFree(h->d_tail_trace);
    if (h->d_job_trace) (void)hipFree(h->d_job_trace);
    for (int i = 0; i < dg_handle::kMaxGroups - 1; ++i) {
        if (h->side_stream[i]) { (void)hipStreamSynchronize(h->side_stream[i]); (void)hipStreamDestroy(h->side_stream[i]); }
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (auto& p : h->F) fr(p);
    for (auto& p : h->Fp) fr(p);
    for (auto& p : h->Ft) fr(p);
    for (auto& p : h->bias) fr(p);
    auto frop = [](GemmOp& op) { free_batched(op); };
    frop(h->F1); frop(h->B1);
    for (auto& o : h->Fd) frop(o);
    for (auto& o : h->Bd) frop(o);
    delete h;
    return DG_OK;
}

int dg_set_weights(dg_handle* h, const char* name, const float* data, const int64_t* shape, int ndim, int is_device) {
    if (!h || !name || !data || !shape) return fail(DG_E_INVALID, "null argument");
    const std::string nm(name);
    // Resolve the name to the shape it must have BEFORE anything is sized or copied from the caller's pointer.
    std::vector<int64_t> want;
    bool any_shape = false;          // BN scale / offset: the reference stores them with the keep_dims shape of the moments
    if (nm == "Generator.Input.W") want = {h->latent, h->lin_out};
    else if (nm == "Generator.Input.b") want = {h->lin_out};
    for (const DeconvSpec& s : h->dec) {
        if (nm == std::string(s.name) + ".Filters") want = {5, 5, s.cout, s.cin};
        else if (nm == std::string(s.name) + ".Biases") want = {s.cout};
    }
    for (const ActInfo& a : h->ai)
        if (a.has_bn && (nm == a.bn_name + ".scale" || nm == a.bn_name + ".offset")) { want = {a.bn_C}; any_shape = true; }
    if (want.empty()) return fail(DG_E_INVALID, "unknown weight name '%s'", name);
    int64_t n_want = 1;
    for (int64_t w : want) n_want *= w;
    if (ndim < 1 || ndim > 8) return fail(DG_E_INVALID, "%s: ndim %d out of range", name, ndim);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0 || shape[i] > n_want) return fail(DG_E_INVALID, "%s: dimension %d is %lld", name, i, (long long)shape[i]);
        n *= shape[i];
        if (n > n_want) break;
    }
    bool shape_ok = n == n_want;
    if (shape_ok && !any_shape) {
        shape_ok = ndim == (int)want.size();
        for (int i = 0; shape_ok && i < ndim; ++i) shape_ok = shape[i] == want[i];
    }
    if (!shape_ok) {
        std::string w;
        for (size_t i = 0; i < want.size(); ++i) w += (i ? "," : "") + std::to_string(want[i]);
        return fail(DG_E_INVALID, any_shape ? "%s: expected %s values" : "%s: expected shape [%s]", name, w.c_str());
    }
    HIP_TRY(hipSetDevice(h->device));
    std::vector<float> host((size_t)n);
    HIP_TRY(hipMemcpy(host.data(), data, (size_t)n * sizeof(float), is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
    if (nm == "Generator.Input.W") {
        std::vector<float> t((size_t)n);
        for (int d = 0; d < h->latent; ++d)
            for (int f = 0; f < h->lin_out; ++f) t[(size_t)f * h->latent + d] = host[(size_t)d * h->lin_out + f];
        HIP_TRY(hipMemcpy(h->lin_w, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->lin_wt, t.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        h->lin_w_host = host;
        const int rcp = build_lin_packs(h);
        if (rcp) return rcp;
        h->have[nm] = true;
        return DG_OK;
    }
    if (nm == "Generator.Input.b") {
        HIP_TRY(hipMemcpy(h->lin_b, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        h->have[nm] = true;
        return DG_OK;
    }
    for (size_t d = 0; d < h->dec.size(); ++d) {
        const DeconvSpec& s = h->dec[d];
        if (nm == std::string(s.name) + ".Filters") {
            std::vector<float> t((size_t)n);
            for (int k = 0; k < 25; ++k)
                for (int co = 0; co < s.cout; ++co)
                    for (int ci = 0; ci < s.cin; ++ci)
                        t[((size_t)k * s.cin + ci) * s.cout + co] = host[((size_t)k * s.cout + co) * s.cin + ci];
            HIP_TRY(hipMemcpy(h->F[d], host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->Ft[d], t.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            if (d + 1 < h->dec.size() && s.cout % 32 == 0 && s.cin % 8 == 0) {
                // forward filters in fragment order (dg_fgemm.hip): the slab of tap k keeps its float offset k * cout * cin; inside it
                // [cout / 32][cin / 8][64 lanes][4]: lane = ((ci % 8) / 4) * 32 + co % 32, element ci % 4
                std::vector<float> pk((size_t)n);
                const int kc8 = s.cin / 8;
                for (int k = 0; k < 25; ++k)
                    for (int co = 0; co < s.cout; ++co)
                        for (int ci = 0; ci < s.cin; ++ci)
                            pk[(size_t)k * s.cout * s.cin + (((size_t)(co / 32) * kc8 + ci / 8) * 64 + ((ci % 8) / 4) * 32 + co % 32) * 4 + ci % 4] =
                                host[((size_t)k * s.cout + co) * s.cin + ci];
                if (!h->Fp[d]) HIP_TRY(hipMalloc(&h->Fp[d], (size_t)n * sizeof(float)));
                HIP_TRY(hipMemcpy(h->Fp[d], pk.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            }
            if (d + 1 == h->dec.size()) {
                // tail forward GEMM: B fragments of v_mfma_f32_32x32x2_f32 in load order.  kappa = (kh*5+kw)*cout + co is
                // the row of host[] (layout [25][cout][cin] == [kappa][c]); lane = (kappa & 31) + 32*half holds
                // c = 8*kk + 4*half + e of kappa tile t.
                const int nk = 25 * s.cout, nt = (nk + 31) / 32, kkn = s.cin / 8;
                std::vector<float> pk((size_t)nt * kkn * 64 * 4, 0.f);
                for (int tt = 0; tt < nt; ++tt)
                    for (int kk = 0; kk < kkn; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int kappa = tt * 32 + (lane & 31), c = kk * 8 + (lane >> 5) * 4 + e;
                                if (kappa < nk) pk[(((size_t)tt * kkn + kk) * 64 + lane) * 4 + e] = host[(size_t)kappa * s.cin + c];
                            }
#ifdef DG_MEASURE      // only the superseded 32-wide CelebA forward tail reads this pack
                if (!h->tail_pack) HIP_TRY(hipMalloc(&h->tail_pack, pk.size() * sizeof(float)));
                HIP_TRY(hipMemcpy(h->tail_pack, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
#else
                (void)pk;
#endif
                if (s.cout == 3) {
                    // B fragments of v_mfma_f32_16x16x4_f32, one 16-column tile per filter row kh: column
                    // j = kw*3 + co (< 15) is kappa = 15*kh + j; lane = j + 16*g holds c = 16*kk + 4*g + e.
                    const int kk16 = s.cin / 16;
                    std::vector<float> p16((size_t)5 * kk16 * 64 * 4, 0.f);
                    for (int kh = 0; kh < 5; ++kh)
                        for (int kk = 0; kk < kk16; ++kk)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 4; ++e) {
                                    const int j = lane & 15, c = kk * 16 + (lane >> 4) * 4 + e;
                                    if (j < 15) p16[(((size_t)kh * kk16 + kk) * 64 + lane) * 4 + e] = host[(size_t)(15 * kh + j) * s.cin + c];
                                }
                    if (!h->tail_pack16) HIP_TRY(hipMalloc(&h->tail_pack16, p16.size() * sizeof(float)));
                    HIP_TRY(hipMemcpy(h->tail_pack16, p16.data(), p16.size() * sizeof(float), hipMemcpyHostToDevice));
                }
            }
            h->have[nm] = true;
            return DG_OK;
        }
        if (nm == std::string(s.name) + ".Biases") {
            HIP_TRY(hipMemcpy(h->bias[d], host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            h->have[nm] = true;
            return DG_OK;
        }
    }
    for (auto& a : h->ai) {
        if (!a.has_bn) continue;
        for (int which = 0; which < 2; ++which) {
            if (nm != a.bn_name + (which ? ".offset" : ".scale")) continue;
            HIP_TRY(hipMemcpy(which ? a.offset : a.scale, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            h->have[nm] = true;
            return DG_OK;
        }
    }
    return fail(DG_E_INVALID, "unknown weight name '%s'", name);
}

int dg_weights_complete(dg_handle* h) {
    if (!h) return 0;
    if (!h->have.count("Generator.Input.W") || !h->have.count("Generator.Input.b")) return 0;
    for (auto& s : h->dec) {
        if (!h->have.count(std::string(s.name) + ".Filters")) return 0;
        if (!h->have.count(std::string(s.name) + ".Biases")) return 0;
    }
    for (auto& a : h->ai)
        if (a.has_bn && (!h->have.count(a.bn_name + ".scale") || !h->have.count(a.bn_name + ".offset"))) return 0;
    return 1;
}

int dg_init_latents(dg_handle* h, float* z, int64_t n_rows, uint64_t seed, int64_t first_row, float std, void* stream) {
    if (!h || !z || n_rows < 0) return fail(DG_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (std <= 0.f) std = std::sqrt(1.0f / (float)h->latent);
    if (n_rows) dg::launch_init_latents(z, n_rows, h->latent, seed, first_row, std, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

int dg_reconstruct(dg_handle* h, const float* x, const float* z0, uint64_t seed, int64_t first_row, int B, int R, int L,
                   float lr, float momentum, float* out_rec, int32_t* out_idx, float* out_loss, float* out_z,
                   void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!x || !out_rec) return fail(DG_E_INVALID, "x and out_rec must be non-null");
    if (L < 0) return fail(DG_E_INVALID, "need L >= 0 (got %d)", L);
    int n_rows = 0;
    rc = check_rows(B, R, &n_rows);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_call(h, B, R, s);          // a no-op (no allocation, no wait) once this shape has been prepared
    if (rc) return rc;
    h->prof_chain_last = -1;                // profile markers never span two calls (copies and the latent draw sit between)
    const size_t zbytes = (size_t)n_rows * h->latent * sizeof(float);
    if (z0) HIP_TRY(hipMemcpyAsync(h->z, z0, zbytes, hipMemcpyDeviceToDevice, s));
    else dg::launch_init_latents(h->z, n_rows, h->latent, seed, first_row, std::sqrt(1.0f / (float)h->latent), s);
    HIP_TRY(hipMemsetAsync(h->m, 0, zbytes, s));
    // (the arrival counters of the folded update return to zero by themselves; cleared per call all the same, so that a call
    // that died half-way cannot poison the next one)
    if (update_folds(h)) HIP_TRY(hipMemsetAsync(h->upd_count, 0, (size_t)(n_rows / 32 + 16) * sizeof(unsigned), s));
    const int steps = L > 1 ? L : 1;
    // the batch split (by image) into row groups on separate streams (option two_streams)
    RowGroup grp[dg_handle::kMaxGroups];
    const int ngroups = split_groups(h, B, R, grp);
    grp[0].s = s;
    // (likewise the arrival counters of the K-pair jobs, before the side streams fork off this one)
    for (int gi = 0; gi < ngroups; ++gi) {
        bool seen = false;
        for (int gj = 0; gj < gi; ++gj) seen = seen || grp[gj].n_rows == grp[gi].n_rows;
        if (!seen) { rc = clear_pair_counters(h, grp[gi].n_rows, s); if (rc) return rc; }
    }
    if (ngroups > 1) {
        for (int gi = 1; gi < ngroups; ++gi) grp[gi].s = h->side_stream[gi - 1];
        HIP_TRY(hipEventRecord(h->ev_fork, s));
        for (int gi = 1; gi < ngroups; ++gi) HIP_TRY(hipStreamWaitEvent(grp[gi].s, h->ev_fork, 0));
    }
    // Small prepared shapes replay a captured graph of the loop instead of enqueuing its ~8 L launches one by one
    bool replayed = false;
    if (h->graph_max_rows > 0 && n_rows <= h->graph_max_rows && !h->graph_broken && h->prof_stride == 0 && ngroups == 1 &&
        steps >= 2 && (int64_t)B * h->P <= h->xbuf_floats) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            hipGraphExec_t exec = loop_graph(h, B, R, L, lr, momentum);
            if (exec) {
                // The graph runs on the engine's own stream, tied to the caller's by events on both sides (measured: a graph
                // launched straight into the legacy NULL stream -- torch's default stream -- was not ordered against the
                // launches that followed it there)
                HIP_TRY(hipMemcpyAsync(h->xbuf, x, (size_t)B * h->P * sizeof(float), hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipEventRecord(h->ev_fork, s));
                HIP_TRY(hipStreamWaitEvent(h->cap_stream, h->ev_fork, 0));
                HIP_TRY(hipGraphLaunch(exec, h->cap_stream));
                HIP_TRY(hipEventRecord(h->ev_join[0], h->cap_stream));
                HIP_TRY(hipStreamWaitEvent(s, h->ev_join[0], 0));
                replayed = true;
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (!replayed) {
        rc = enqueue_steps(h, x, R, L, lr, momentum, grp, ngroups);
        if (rc) return rc;
    }
    for (int gi = 1; gi < ngroups; ++gi) {
        HIP_TRY(hipEventRecord(h->ev_join[gi - 1], grp[gi].s));
        HIP_TRY(hipStreamWaitEvent(s, h->ev_join[gi - 1], 0));
    }
    dg::launch_select(h->loss, h->y, B, R, h->P, out_rec, out_idx, s);
    if (out_loss) HIP_TRY(hipMemcpyAsync(out_loss, h->loss, (size_t)n_rows * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_z) HIP_TRY(hipMemcpyAsync(out_z, h->z, zbytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

int dg_prepare(dg_handle* h, int B, int R, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    int n_rows = 0;
    rc = check_rows(B, R, &n_rows);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_call(h, B, R, s);
    if (rc) return rc;
    // dg_loss_grad / dg_generate and the profiled (single-group) form of the same shape run all rows as one group
    rc = prepare_rows(h, n_rows, &n_rows, 1, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return DG_OK;
}

int dg_generate(dg_handle* h, const float* z, int N, float* out_y, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!z || !out_y) return fail(DG_E_INVALID, "bad argument");
    int n_checked = 0;
    rc = check_rows(N, 1, &n_checked);
    if (rc) return rc;
    // every row is compared with image 0 by passing R = N: the tails' row -> image division (a 40-bit magic multiply) is exact
    // for row * R < 2^40 only
    if (N > (1 << 20)) return fail(DG_E_INVALID, "dg_generate: at most %d rows per call (got %d): split the batch", 1 << 20, N);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_rows(h, N, &N, 1, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->z, z, (size_t)N * h->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    // the loss is discarded here: every row is compared with one all-zero image (R = N -> image 0)
    RowGroup g; g.n_rows = N; g.s = s;
    rc = run_forward(h, h->xzero, g, /*R=*/N, /*want_y=*/true, /*want_loss=*/false, /*tail_backward=*/false, false);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_y, h->y, (size_t)N * h->P * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

int dg_loss_grad(dg_handle* h, const float* x, const float* z, int B, int R, float* out_y, float* out_loss,
                 float* out_dz, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!x || !z) return fail(DG_E_INVALID, "bad argument");
    int n_rows = 0;
    rc = check_rows(B, R, &n_rows);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_rows(h, n_rows, &n_rows, 1, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->z, z, (size_t)n_rows * h->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    rc = clear_pair_counters(h, n_rows, s);
    if (rc) return rc;
    RowGroup g; g.n_rows = n_rows; g.s = s;
    rc = run_forward(h, x, g, R, /*want_y=*/out_y != nullptr, /*want_loss=*/out_loss != nullptr, /*tail_backward=*/out_dz != nullptr, false);
    if (rc) return rc;
    if (out_y) HIP_TRY(hipMemcpyAsync(out_y, h->y, (size_t)n_rows * h->P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_loss) HIP_TRY(hipMemcpyAsync(out_loss, h->loss, (size_t)n_rows * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_dz) {
        rc = run_backward(h, g, false);
        if (rc) return rc;
        dg::launch_momentum_update(nullptr, nullptr, h->part, h->nsplit, n_rows, h->latent, 0.f, 0.f, out_dz, s);
    }
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

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

int dg_profile_enable(dg_handle* h, int on) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    h->prof_stride = on > 0 ? on : 0;
    return DG_OK;
}
int dg_profile_count(dg_handle* h) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    prof_collect(h);
    return (int)h->prof.size();
}
int dg_profile_read(dg_handle* h, int i, char* name, int name_len, int64_t* launches, double* total_ms, double* flops) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    prof_collect(h);
    if (i < 0 || i >= (int)h->prof.size()) return fail(DG_E_INVALID, "profile index out of range");
    const ProfEntry& p = h->prof[i];
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.name.c_str());
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = p.ms;
    if (flops) *flops = p.flops;
    return DG_OK;
}
int dg_profile_reset(dg_handle* h) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    prof_collect(h);
    h->prof.clear();
    h->prof_index.clear();
    return DG_OK;
}

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
    if (rc == DG_OK) ++h->list_epoch;    // an accepted option: captured loops are rebuilt on their next use (a refused one changes nothing)
    return rc;
}

static int set_option(dg_handle* h, const char* key, const char* value) {
    const std::string k(key);
    if (k == "two_streams") {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->two_streams = atoi(value);          // number of concurrent row groups (0/1 = off, 2..8)
        if (h->two_streams == 1) h->two_streams = 2;   // historic meaning of "1": two groups
        drop_job_lists(h);                     // a list's K-pair scratch is sized by the number of groups that may launch it at once
        return DG_OK;
    }
    if (k == "lr_schedule") {
        const std::string v(value);
        if (v != "constant" && v != "intended") return fail(DG_E_INVALID, "lr_schedule: 'constant' or 'intended'");
        h->lr_intended = v == "intended";
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
    if (k == "bn_fused") {               // 1 = Batchnorm forward statistics from the GEMM epilogue (default), 0 = a separate pass
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->bn_fused = atoi(value) != 0;
        drop_job_lists(h);
        return rebuild_plans(h);
    }
    if (k == "frag_path") {              // 1 = fragment-order forward path (dg_fgemm.hip; default), 0 = every GEMM on dg_gemm.hip
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->frag_path = atoi(value) != 0;
        free_workspace(h);               // the fragment-order buffers exist only with it
        drop_job_lists(h);               // (the backward layers' lists were timed with the other epilogue)
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
