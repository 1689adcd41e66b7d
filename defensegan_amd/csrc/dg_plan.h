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


// ---- position-batched form (dg_gemm.hip) -------------------------------------------------------
// The same layer regrouped: output positions whose valid taps form the same RELATIVE pattern (same filter taps, same
// input offsets relative to the position) are one class; the tap order inside a class is the per-position order above, so
// every output element keeps its summation order.
struct BatchedPlan {
    std::string name;
    std::vector<ClassDesc> cls;    // sorted by descending K
    std::vector<TapEntry> taps;    // per class: a_off >= 0 relative to pos_a of the row's position, w_off as above
    std::vector<int> pos_a;        // per position (class-major): float offset of the position's first-tap base in an A row
    std::vector<int> pos_out;      // per position: float offset inside an output row
    long long a_rowstride = 0, out_rowstride = 0;
    int w_rowstride = 0, kch = 0, ncols = 0;
    long long macs_per_row = 0;
};
// `p` must have been planned with bn == ncols (one PosEntry per position).
BatchedPlan make_batched(const LayerPlan& p);

// Job list of one launch: every class's M axis (n_rows * positions) cut into tiles, columns into tiles, ordered longest
// first (the hardware dispatcher hands workgroups out in this order).  family 0: full tile 128x128, family 1: 256x64.
// Built by simulating that dispatch (greedy list scheduling on `slots` equal servers, cost model below): jobs are taken
// longest first, and one that would end later than `slack` x (total cost / slots) is cut in two along M (family 0's second
// cut: along N) -- never along K (the K-pair classes below are split in two FIXED halves, whatever the list) -- whose pieces queue up again, down to 64x64.  slack <= 0 picks, from a fixed ladder,
// the value with the smallest simulated makespan; slack >= 1e20 never cuts.  min_level > 0 starts every tile cut to that
// level (1 halves, 2 quarters): such a list needs less LDS and registers per workgroup, so the caller may pass more slots.
// Classes of at least this many K chunks (with at least two taps) are computed by K-pair jobs (dg_types.h JobDesc::pair_id):
// 80 = the 20- / 25-tap classes of the 4x4 <-> 7x7 / 8x8 backward layers (80 / 100 chunks).  Measured with 64 (which also pairs
// their 16-tap class and the 9-tap class, 72 chunks, of the forward layers): the forward layers lose 1-2 % to it, the backward
// layers gain 3 % (MNIST, 2560 rows) / 8 % (CelebA, 1280 rows) -- profiles/r05_ab_k_pair.txt.  A constant of the build: which
// classes are paired must not depend on the row count or the list.  (Also measured and dropped: pairing the 50-chunk, few-position
// classes of the 7x7 <- 14x14 / 8x8 <- 16x16 backward layers as well: Generator.3's backward 315 -> 324.5 us on MNIST, 230 -> 232.6
// on CelebA -- pairs only pay where one chain lasts most of the launch.)
constexpr int kPairMinChunks = 80;
inline bool class_is_paired(const BatchedPlan& p, int cls) {
    return p.cls[(size_t)cls].nchunks >= kPairMinChunks && p.cls[(size_t)cls].nchunks / (p.kch / 32) >= 2;
}
// taps of the first half of a paired class (the second half takes the rest)
inline int pair_first_taps(const BatchedPlan& p, int cls) { return p.cls[(size_t)cls].nchunks / (p.kch / 32) / 2; }
// scratch a job list needs: floats of accumulator images, number of pair counters
struct PairNeeds { long long floats = 0; int pairs = 0; };
PairNeeds pair_needs(const std::vector<JobDesc>& jobs, int family);
// 32-row statistics blocks of a layer at n_rows latent rows (JobDesc::stat_base): sum over the classes of ceil(n_rows * s_c / 32)
long long stat_blocks(const BatchedPlan& p, int n_rows);

struct JobModel {
    // measured on MI355X at 12 500 rows (profiles/r02_*): TFLOP/s of a chip full of jobs of one shape, by (family, level),
    // at the residency that shape allows (2 / 3 / 5 workgroups per CU), and the prologue + epilogue of one job in
    // microseconds of its slot's time (from the K = 128 Linear layer, where they are a third of a job)
    double rate[2][3] = {{141.5, 138.5, 134.0}, {139.0, 136.5, 131.5}};
    double fixed_us[2][3] = {{7.5, 5.0, 3.4}, {7.5, 5.0, 3.4}};
    // what a K-pair job adds to its slot's time: its accumulator image written through (64 / 32 / 16 KB: the guide's publish-large
    // row prices 64 KB at 3 us), the ticket, and for the second arriver the partner's image read back
    double pair_us[3] = {5.0, 3.5, 2.5};
    // Taper (round 4): real jobs do not run at the model's speed (a workgroup on a CU with fewer neighbours is faster, operands
    // miss or hit), so where the model sees a level end the CUs finish 3-7 % of the launch apart (tools/job_trace.py: mean last-job
    // end 273 of 294 us for Generator.2's backward) -- by about a tenth of the duration of the jobs that were started LAST.  With
    // taper > 0 a piece may only START before taper x (the ideal makespan) at level 0 and before the midpoint between that and
    // the end at level 1; later it is cut further whatever its predicted end: the launch ends on progressively smaller jobs, and a
    // mis-predicted speed moves its end by a fraction of a SMALL job.  0 = off.  One more dimension of the timed choice.
    double taper = 0.0;
};
std::vector<JobDesc> build_jobs(const BatchedPlan& p, int n_rows, int family, int slots, double slack,
                                const JobModel& model = JobModel(), double* predicted_us = nullptr, int min_level = 0);
// Locality order for the 8 XCDs of an MI355X (each with its own 4 MB L2; the dispatcher hands consecutive workgroups to
// consecutive XCDs, workgroup i -> XCD i mod n_xcd): the first `head_frac` of a dispatch-ordered list -- the long whole-tile
// jobs, which in cost order are class-major, i.e. every XCD sweeps the whole batch once per class and re-fetches each input
// row from the fabric for every class and tap -- is re-arranged so that XCD x receives the jobs of latent rows
// [x, x + 1) * n_rows / n_xcd in ascending row order, all classes (and column tiles) of a row range next to each other in
// time: the input rows a resident set of jobs touches then fit the XCD's L2.  The tail of the list (short jobs, cut pieces)
// keeps its longest-first order, which is what levels the end of the launch.  A pure permutation: results cannot change.
void order_for_xcd(std::vector<JobDesc>& jobs, int n_rows, double head_frac, int n_xcd = 8);

// ---- tuning records -------------------------------------------------------------------------------------------------
// A job list is a pure function of (layer plan, row count, family, slots, starting level, cutting threshold, order variant):
// these few numbers are what dg_export_tuning writes and dg_import_tuning reads (one text line per layer and row count), so
// that another process reproduces the timed choice of this one exactly, without timing.
struct TuneRecord {
    std::string op;            // layer name ("F2", "B3", ...)
    int n_rows = 0;
    int min_level = 0;
    double slack = 0.0;        // build_jobs' cutting threshold (<= 0: its ladder, 1e30: never cut)
    int snake = 0;             // 1: every other round of `cus` jobs reversed; 2: balance_order; 3: spread_order; 4: jobs_balanced (a list of its own)
    int xcd_order = 0;         // order_for_xcd with head fraction xcd_head
    double xcd_head = 0.0;
    int n_jobs = 0;            // length of the list (checked on import: another planner / cost model makes another list)
    double measured_us = 0.0;  // informational
    double taper = 0.0;        // JobModel::taper the list was built with (tenth field of the line; absent in round-4a texts = 0)
    int prio = 0;              // 1 = wave priorities by predicted job length (assign_priorities; eleventh field, absent = 0)
    int pair_kernel = 0;       // 1 = the list runs the PAIR instantiation of the kernel although it holds no K-pair job (twelfth field,
                               // absent = 0): the same arithmetic, another register allocation / schedule -- a timed choice like the rest
};
std::string format_tune_record(const TuneRecord& r);                 // one line, '\n'-terminated
// Parses the record at *p and advances *p behind it; false on a malformed record (nothing consumed) or at the end of the text.
bool parse_tune_record(const char** p, TuneRecord* r);
// The list the record describes (slots_per_cu = resident workgroups per CU at the record's level).
std::vector<JobDesc> jobs_from_record(const BatchedPlan& p, int family, int cus, int slots_per_cu, const TuneRecord& r,
                                      const JobModel& model = JobModel(), double* predicted_us = nullptr);
void snake_order(std::vector<JobDesc>& jobs, int cus);
// A list that fits the resident slots (jobs <= cus * slots_per_cu) is dispatched in one go, workgroup i to CU ~ i mod cus, and a
// CU shares its matrix pipes among its resident jobs whatever their sizes: when the launch ends is decided by the CU with the
// most WORK, not by the longest job.  balance_order partitions the jobs into `cus` bins of equal cardinality (+- 1) and nearly
// equal predicted work (longest job first into the lightest bin that still has room) and writes bin b's jobs to the positions
// b, b + cus, b + 2 cus, ...  (TuneRecord::snake = 2).  A pure permutation; lists that do not fit are left alone.
void balance_order(const BatchedPlan& p, std::vector<JobDesc>& jobs, int family, int cus, int slots_per_cu, const JobModel& model);
// A list BUILT for one dispatch round (TuneRecord::snake = 4): every job resident from the start, so a CU is done when its own
// jobs' work is done, whatever their lengths.  Tiles start at `min_level`; longest piece first into the lightest CU that still has a
// slot; while the heaviest CU carries more than (1 + tol) x the mean, its largest piece is cut in two (along M, then N -- never
// K) and everything is placed again -- as long as the list still fits cus * slots_per_cu jobs.  CU b's jobs go to the positions b, b + cus, ...
// (CUs with the most jobs first, so every round of `cus` positions is a prefix).  Empty when the layer does not fit one round.
std::vector<JobDesc> jobs_balanced(const BatchedPlan& p, int n_rows, int family, int cus, int slots_per_cu, int min_level,
                                   const JobModel& model = JobModel(), double tol = 0.03);
// Lists of several dispatch rounds.  In longest-first order the first round is `slots` jobs of ONE length: they end together,
// their successors start together (all in their start-up and first operand burst at once) and end together again.  spread_order
// hands `frac` of the first round's slots to ALL lengths of the list in proportion to their counts (a job of duration D only has
// to start before T - D, and the displaced long jobs start as soon as the first short ones end), the rest of the round and
// everything behind it stays longest first: jobs then end and start one by one from the first turnover on.  A pure permutation
// (TuneRecord::snake = 3).
void spread_order(const BatchedPlan& p, std::vector<JobDesc>& jobs, int family, int slots, const JobModel& model, double frac = 0.5);
// JobDesc::prio by predicted length: a job whose predicted duration (cost model, at the list's residency) is in the top quarter
// of the longest job's gets priority 3, the next quarter 2, ... ; mode 0 clears them.  Order and arithmetic are untouched.
void assign_priorities(const BatchedPlan& p, std::vector<JobDesc>& jobs, int family, int slots, const JobModel& model, int mode);

// Makespan (microseconds) of greedy list scheduling of `jobs` in order on `slots` servers of 1/slots of the chip each.
double simulate_jobs(const BatchedPlan& p, const std::vector<JobDesc>& jobs, int family, int slots, const JobModel& model);

// ---- fragment-order path (dg_fgemm.hip; layouts and FragJob: dg_types.h) ------------------------------------------------------
// The taps of a class as a grid: tap t = (u, v) = (t / nw, t % nw), a_off = a0 + u * a_u + v * a_v, w_off = w0 + u * w_u + v * w_v
// (the valid (kh, kw) of a position are a product of two arithmetic progressions; Linear layers: one row).  false = not affine.
struct TapGrid { int nw = 0, a0 = 0, a_u = 0, a_v = 0, w0 = 0, w_u = 0, w_v = 0; };
bool frag_tap_grid(const BatchedPlan& p, int cls, TapGrid* g);
// Waves that share one tile's K axis: a constant of the class (its K chunks and the layer's K extent per tap), never of the row count
// or the list -- every output element is one fixed tree of k-ordered chains.  Every wave's part is a multiple of 4 k8-steps.
int frag_ksplit(const BatchedPlan& p, int cls);
// Can dg_fgemm.hip run this layer?  (every class a position grid with an affine tap grid, K extent 64 / 128 / 256, columns % 64 == 0,
// dense filter slabs)
bool frag_supported(const BatchedPlan& p);
// Job list of one launch at n_rows latent rows (ceil(n_rows / 32) row blocks), longest jobs first.
std::vector<FragJob> build_frag_jobs(const BatchedPlan& p, int n_rows);
// The persistent form: every wave tile of the layer (4 M blocks x 64 channels, never split along K) assigned to one of
// `n_wgs * 4` waves -- longest first into the lightest PAIR of waves that share a SIMD (wave w of workgroups b and b + n_wgs / 2: the
// dispatcher hands workgroup i to CU i mod #CUs), then alternately to the pair's two waves -- and returned grouped by wave;
// begin[k] .. begin[k + 1] = the records of wave k = 4 * workgroup + wave (begin has n_wgs * 4 + 1 entries).
std::vector<FragJob> build_frag_tiles(const BatchedPlan& p, int n_rows, int n_wgs, std::vector<int>* begin);

}  // namespace dg
