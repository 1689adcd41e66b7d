// Internal header of the engine's translation units (dg_engine*.cpp): the handle, the per-layer records and the functions the
// units share.  Nothing here is part of the C ABI (include/defensegan_hip.h).
//   dg_engine.cpp        handle life cycle, weights, workspace, the launch sequence, the compute entry points
//   dg_engine_lists.cpp  job lists: building, timing (get_jobs), K-pair scratch, fragment-order lists, tuning export / import
//   dg_engine_prof.cpp   per-launch event profile
//   dg_engine_opts.cpp   dg_set_option, dg_debug_read
#pragma once
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

#pragma GCC visibility push(hidden)
namespace dge {

extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);

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
    int* d_begin = nullptr;        // persistent form (frag_path = 2): per-wave ranges of d_jobs, in the same allocation
    int n_wgs = 0;
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
    float* stats = nullptr;   // EPI_BIAS_STATS / EPI_MASK_STATS: the block sums of the activation this layer writes (ActInfo::block_sums)
    int64_t stats_cap = 0;    // blocks that buffer holds
    int bn_act = -1;          // EPI_MASK_STATS: index (dg_handle::ai) of the activation whose ReLU gradient this layer writes
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

}  // namespace dge
#pragma GCC visibility pop
using namespace dge;

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
    // The whole latent turn (Linear backward -> update -> next step's Linear forward) as ONE launch (dg_turn.hip): option
    // "turn_fused"; needs the weight-stationary shapes, no Batchnorm behind the Linear layer, the NHWC path.  Bit-identical.
    int turn_fused = 0;
    static constexpr int kTurnBarWords = 1024;    // arrival counters of one concurrent row group ([workgroup row groups][2])
    unsigned* turn_bar = nullptr;  // [kMaxGroups][kTurnBarWords] + the error word; zeroed at the start of every call
    unsigned* turn_err_host = nullptr;   // pinned copy of the error word, fetched behind every call
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
    int bn_fused = 2;         // option "bn_fused": 0 = separate statistics passes, 1 = forward sums from the GEMM epilogue, 2 = backward sums too
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
    float* tail_bn_sums = nullptr; // Batchnorm form of the MNIST tail: [tail_bn_sums_wgs][2][C] backward sums of the last Batchnorm layer
    int tail_bn_sums_wgs = 0;
    bool tail_left_bn_sums = false; // the last run_forward's tail ran in that form (run_backward takes the sums from it)
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
    int two_streams_auto = 0;      // 1: whether a call shape runs as ONE group or as two_streams groups is timed once per shape (prepare_call)
    std::map<std::pair<int, int>, int> group_choice;   // (B, R) -> number of row groups that timing chose
    double group_timing_ms[2] = {0.0, 0.0};            // the last timing: nine loop steps as one group / as two_streams groups
    int two_stream_min_rows = 1024;
    int two_stream_split = 0;      // two groups: percent of the images in the first (0 = halves)
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

#pragma GCC visibility push(hidden)
namespace dge {

int prof_slot(dg_handle* h, const std::string& name);
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

void prof_collect(dg_handle* h);
int launch_check(const char* what);
// ---- dg_engine_lists.cpp
void drop_job_lists(dg_handle* h);
void free_batched(GemmOp& op);
int upload_batched(GemmOp& op, const dg::LayerPlan& base);
dg::GemmArgs gemm_args(dg_handle* h, const GemmOp& op, const JobList& jl, const float* A, float* Out, int group = 0);
int pair_copies(const dg_handle* h);
bool upload_jobs(JobList& jl, const std::vector<dg::JobDesc>& jobs, int family, int copies);
int clear_pair_counters(dg_handle* h, int n_rows, hipStream_t s);
const JobList* find_jobs(const GemmOp& op, int n_rows);
const JobList* get_jobs(dg_handle* h, GemmOp& op, int n_rows, const float* A, float* Out, hipStream_t s);
const FragList* find_frag_jobs(const GemmOp& op, int n_rows);
const FragList* get_frag_jobs(GemmOp& op, int n_rows, int persist_wgs = 0);
// ---- dg_engine.cpp
bool lin_stationary(const dg_handle* h, const GemmOp& op);
bool frag_on(const dg_handle* h);
int build_lin_packs(dg_handle* h);
int rebuild_plans(dg_handle* h);
void free_workspace(dg_handle* h);
void drop_graphs(dg_handle* h);

}  // namespace dge
#pragma GCC visibility pop
