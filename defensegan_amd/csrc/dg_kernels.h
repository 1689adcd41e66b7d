// Device-kernel launch interface of the projection engine (gfx950 only).
// Host code (dg_engine.cpp) fills these argument blocks from the layer plans (dg_plan.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dg_types.h"

namespace dg {

// hipFuncSetAttribute applies to the CURRENT device: "set it once" guards in the launchers are kept per device, so that a
// process driving several GPUs (one handle each) raises the dynamic-LDS limit on every one of them.
struct PerDeviceOnce {
    int level[64] = {};
    bool need(int want = 1) {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 63;
        if (level[d] >= want) return false;
        level[d] = want;
        return true;
    }
};


// Measurement scaffolding (in-kernel phase traces, phase-removal switches, wave-priority hooks, the superseded tail
// formulations kept as cross-checks) is compiled only with -DDG_MEASURE, into a SECOND library
// (lib/libdefensegan_hip_measure.so, same sources): the product library's hot kernels carry none of it -- not in their
// argument blocks either (a switched-off hook changed the register allocation of a GEMM instantiation).
#ifdef DG_MEASURE
#define DG_DBG(a) ((a).dbg)
#define DG_TRACE_PTR(a) ((a).trace)
#else
#define DG_DBG(a) 0
#define DG_TRACE_PTR(a) (static_cast<long long*>(nullptr))
#endif

enum EpiMode : int {
    EPI_STORE = 0,       // out = acc
    EPI_BIAS = 1,        // out = acc + bias[col]
    EPI_BIAS_RELU = 2,   // out = max(acc + bias[col], 0)          (BiasAdd + Relu)
    EPI_MASK = 3,        // out = out_old > 0 ? acc : 0            (ReluGrad, in place over the activation)
    EPI_BIAS_STATS = 4,  // out = acc + bias[col], and per 32-row block of the class's M axis the column sums of out and out^2
                         // (GemmArgs::stats): the Batchnorm forward statistics without a pass over the pre-activations
    EPI_MASK_BITS = 5,   // out = gate bit ? acc : 0: ReluGrad with the gates as one bit per element (GemmArgs::gate_bits) -- the
                         // fragment-order path, whose forward activations do not live in the buffer the gradient is written to
    EPI_MASK_STATS = 6,  // ReluGrad behind a Batchnorm layer, with that layer's BACKWARD sums: the gate is re-formed from the layer's
                         // pre-activations (GemmArgs::bn_pre; relu(bn(pre)) > 0 by the float expression bn_apply_fwd_kernel stored), and
                         // per 32-row block the column sums of dy and dy * xhat go to GemmArgs::stats -- the statistics pass of
                         // launch_bn_backward (a read of dy and of the pre-activations) is not run (launch_bn_backward_from_blocks)
};

// ---- position-batched gathered implicit GEMM (dg_gemm.hip) --------------------------------------
// Linear fwd/bwd (tflib/ops/linear.py:129-142), every Deconv2D fwd (tf.nn.conv2d_transpose, tflib/ops/deconv2d.py:100-117)
// and its backward-to-input (a stride-2 SAME conv), with the 5x5 taps resolved per output position on the host so that no
// zero is ever multiplied.  M axis = (latent row, output position) pairs of one tap class (dg_types.h), one workgroup per
// JobDesc:
//   Out[n, pos_out[j] + n0 + c] = epi( sum_{t in taps(class)} sum_{k < kch} A[n*a_rowstride + pos_a[j] + a_off(t) + k]
//                                                                         * W[w_off(t) + (n0 + c)*w_rowstride + k] )
struct GemmArgs {
    const float* A;
    const float* W;
    float* Out;
    const float* bias;
    const JobDesc* jobs;
    const ClassDesc* cls;
    const TapEntry* taps;
    const int* pos_a;        // per position: float offset of its input base inside an A row
    const int* pos_out;      // per position: float offset inside an output row
    long long a_rowstride;
    long long out_rowstride;
    int w_rowstride;
    int kch;                 // K extent per tap, multiple of 32
    int mode;                // EpiMode
    int n_jobs;
    int min_level;           // smallest shape code among the jobs (0 = the list holds full tiles)
    // K-pair jobs (dg_types.h JobDesc::pair_id): accumulator images of the list's pairs and their arrival counters (zero between
    // launches: the second arrival wraps a counter back to zero); nullptr when the list has no pair
    float* pair_scratch;
    unsigned* pair_count;
    // EPI_BIAS_STATS: [blocks][2][stats_cols] floats; block = JobDesc::stat_base + (row of the class's M axis) / 32 -- a fixed
    // partition of the layer's rows whatever the job list looks like, every block summed in a fixed order: deterministic
    float* stats;
    int stats_cols;          // output columns per position (the Batchnorm channels)
    // EPI_MASK_BITS: [rows][gate_words] words, bit f % 32 of word f / 32 = (activation f of the row > 0); gate_words = out_rowstride / 32
    const unsigned* gate_bits;
    int gate_words;
    // EPI_MASK_STATS: the Batchnorm layer in front of the ReLU whose gradient this launch writes -- its pre-activations (geometry of
    // Out), forward statistics [2][stats_cols] (mean, rstd), scale and offset [stats_cols]
    const float* bn_pre;
    const float* bn_fstats;
    const float* bn_scale;
    const float* bn_offset;
#ifdef DG_MEASURE
    long long* trace;        // optional [n_jobs][4] per-workgroup {start, end (100 MHz ticks), HW_ID, chunks}
#endif
};
// family 0: layers with >= 128 output columns (job shapes 128x128 / 64x128 / 64x64); family 1: 64 columns (256x64 / 128x64 / 64x64)
void launch_gemm(int family, const GemmArgs& a, hipStream_t s);
int gemm_lds_bytes(int family, int min_level);

// ---- LDS-free gathered implicit GEMM on fragment-order operands (dg_fgemm.hip; layouts: dg_types.h "fragment order") --------
// Forward transposed convolutions whose input activation is in fragment order:
//   Out[n, o_pos(j) + c] = epi( sum_{t in taps(class)} sum_{k < kch} A[n, a_pos(j) + a_off(t) + k] * W[w_off(t) + c * kch + k] )
struct FragArgs {
    const float* A;          // fragment order, a_rowstride floats per latent row
    const float* Wp;         // filters: per tap slab (same float offsets w_off as the [cout][cin] slabs) [cout / 32][kch / 8][64][4]
    float* Out;              // fragment order (out_frag) or plain NHWC rows
    const float* bias;
    unsigned* gate_bits;     // optional (EPI_BIAS_RELU): [rows][gate_words] one bit per output element, bit f % 32 of word f / 32 = out > 0
    const FragJob* jobs;
    const TapEntry* taps;
    long long a_rowstride, out_rowstride;   // floats per latent row
    int gate_words;          // out_rowstride / 32
    int kch;                 // K extent per tap (input channels): 64, 128 or 256
    int kc8_log2;            // log2(kch / 8)
    int mode;                // EPI_BIAS_RELU / EPI_BIAS / EPI_STORE
    int out_frag;
    int n_jobs;
    // persistent form: jobs = the wave tiles grouped by wave, wave w of workgroup b walks jobs[wave_begin[4 b + w] .. wave_begin[4 b + w + 1])
    const int* wave_begin;   // nullptr = one workgroup per job record (fgemm_kernel)
    int n_wgs;
#ifdef DG_MEASURE
    int dbg;                 // timing experiments (env DG_FRAG_DBG): 1 = no epilogue, 2 = no K-split reduction
    long long* trace;        // [n_jobs * 4][8] per-wave cycle stamps (tools/frag_trace.py)
#endif
};
void launch_fgemm(const FragArgs& a, hipStream_t s);

// ---- weight-stationary Linear kernels of the latent turn (dg_linear.hip) ------------------------
// Out[n, unit*out_unit + c] = epi( sum_{k < 32*kch} A[n*a_rowstride + unit*a_unit + k] * Wunit[c][k] ),
// c < 128.  Forward (tflib/ops/linear.py:129-142): unit = a 128-feature column tile of W^T [features][latent], kch = latent/32.
// Backward: unit = one of the engine's fixed K slices of W [latent][features] (kch = slice/32), Out = the split-K partials
// [n][slice][latent] that momentum_update_kernel adds in slice order.  Same fma chains as launch_gemm (bit-identical).
// The weights arrive in MFMA FRAGMENT ORDER (lin_pack_index, built once per weight upload by dg_engine.cpp): the registers a
// lane keeps for the workgroup's life are consecutive 16-byte pieces of one array, so the 64 / 128 KB a workgroup loads are
// fully coalesced 1 KB runs (the reference layouts would be 32 partial lines per load instruction, 16 KB apart).
struct LinArgs {
    const float* A;
    const float* Wp;         // [units][4 waves][kch][4 k-steps][64 lanes][4]: lane = column (lane & 31) + 32 * k-half
    float* Out;
    const float* bias;       // forward only
    long long a_rowstride;   // floats per row of A
    long long out_rowstride; // floats per row of Out
    int a_unit, out_unit;    // float offsets per unit
    int n_rows;
    int units;               // column tiles (forward) / K slices (backward)
    int groups;              // workgroups per unit; group g owns the 32-row blocks g, g + groups, ...
    int kch;                 // 32-float K chunks per row: 2, 4 or 6 forward, 8 backward
    int mode;                // EpiMode (EPI_BIAS_RELU / EPI_BIAS forward, EPI_STORE backward)
    // backward only, optional: the momentum update folded into this launch (dg_linear.hip, "folded update").  The workgroup
    // that stores the LAST of a 32-row block's `units` K slices sums the slices in slice order and applies ApplyMomentum to
    // upd_z / upd_m (both [n_rows][128], the rows of Out); upd_count = one arrival counter per 32-row block, zero between launches
    // forward only, optional: the output in FRAGMENT ORDER (dg_types.h; rows padded to a multiple of 32) instead of Out, and the
    // ReluGrad gates as one bit per element ([rows][gate_words] words, bit f % 32 of word f / 32)
    float* out_frag;
    unsigned* gate_bits;
    int gate_words;
    float* upd_z;            // nullptr: partials only (momentum_update_kernel follows as its own launch)
    float* upd_m;
    unsigned* upd_count;
    float upd_lr, upd_momentum;
#ifdef DG_MEASURE
    long long* trace;        // optional [grid][8] shader-clock stamps of every workgroup: start, first block ready, first block
                             // multiplied, first block stored, end, HW_ID, blocks, -
#endif
};
bool lin_stationary_supported(int kch, int mode);
bool lin_fold_supported(int units, int latent);
// float index inside Wp of element e of the fragment (unit, wave, chunk c, k-step kk, lane): the weight of output column
// wave*32 + (lane & 31) of the unit at k = c*32 + (kk*2 + (lane >> 5))*4 + e
__host__ __device__ inline long long lin_pack_index(int unit, int wave, int kch, int c, int kk, int lane, int e) {
    return ((((long long)(unit * 4 + wave) * kch + c) * 4 + kk) * 64 + lane) * 4 + e;
}
void launch_lin_stationary(const LinArgs& a, hipStream_t s);

// ---- the latent turn as one launch (dg_turn.hip; engine option turn_fused) ----------------------
// Linear backward (partials [n][nsplit][128] of dz from dA [n][features], K slices of 256 features) -> ApplyMomentum on z / m
// [n][128] -> Linear forward + BiasAdd + ReLU into H [n][features] (dA and H may be the same buffer: a workgroup's forward
// writes come after its row group's backward reads).  Grid nsplit x groups workgroups; `bar` = two arrival counters per row
// group ([groups][2]), zeroed by the caller at the start of every call; `err` is raised when a barrier poll gives up.
// Bit-identical to launch_lin_stationary(backward) + launch_momentum_update + launch_lin_stationary(forward).
struct TurnArgs {
    const float* dA;
    const float* Wb;         // lin_pack_index order, [nsplit][4][8]...   (the backward's K slices)
    float* part;
    float* z;
    float* m;
    const float* Wf;         // lin_pack_index order, [features / 128][4][4]...   (the forward's column tiles)
    const float* bias;
    float* H;
    unsigned* bar;
    unsigned* err;
    float lr, momentum;
    int features;
    int n_rows;
    int nsplit;
    int groups;              // <= ceil(n_rows / 32)
#ifdef DG_MEASURE
    long long* trace;        // optional [grid][8] stamps of the 100 MHz real-time counter: start, backward ready, backward multiplied,
                             // partials drained, past barrier 1, updated + z drained, past barrier 2, end (tools/turn_trace.py)
#endif
};
constexpr int kTurnMaxRows = 1 << 17;        // the update addresses the partials by 32-bit byte offsets: rows x nsplit x 512 B < 4 GB
bool turn_fused_supported(int nsplit, int latent, int features);
void launch_latent_turn(const TurnArgs& a, hipStream_t s);

// ---- MNIST tail: Generator.5 (64 -> 1, 28x28) + sigmoid + loss + backward to da3 --------------
// dataset_models.py:66-69, gan.py:410-414.  One workgroup per latent row.
struct MnistTailArgs {
    float* h3;           // [N,14,14,C] in: relu output; out (if backward): da3 = dh3 * [h3 > 0]
    const float* F5;     // [5,5,1,C]
    const float* b5;     // [1]
    const float* x;      // [B,28,28,1]
    float* loss;         // [N]
    float* y;            // [N,784] or nullptr
    int n_rows;
    int R;
    int C;               // net_dim (64)
    int do_backward;
    int pipe;            // > 0: persistent pipelined kernel with this many workgroups when n_rows >= 2 * pipe (C = 64)
    int want_loss;       // 0: nobody reads this launch's per-row loss (every launch of a projection but the last forward): the third-
                         // generation pipelined kernel, which does not reduce it, may run
    // Batchnorm form of the third-generation pipelined kernel (mnist_tail_pipe3_kernel<C, true>; nullptr = plain): the input map is
    // bn_pre, the PRE-ACTIVATIONS of Generator.3's Batchnorm layer (h3 is then written only: da3), with that layer's forward statistics
    // [2][C] (mean, rstd), scale and offset [C]; bn_sums receives [pipe][2][C] backward sums (dy, dy * xhat) for
    // launch_bn_backward_from_blocks.  Other kernels of this launcher ignore the fields: callers set them only when that kernel runs
    // (mnist_tail_runs_pipe3)
    const float* bn_pre;
    const float* bn_fstats;
    const float* bn_scale;
    const float* bn_offset;
    float* bn_sums;
    int pipe_version;    // 3: mnist_tail_pipe3_kernel (one GEMM per wave, 16 waves; no loss); 2: mnist_tail_pipe2_kernel (matrix work levelled over the SIMDs, positions 192..195 on the gather waves); 1: mnist_tail_pipe_kernel
#ifdef DG_MEASURE
    int dbg;             // timing experiments only: 1 skip gather, 2 skip forward GEMM, 3 skip backward GEMM
    long long* trace;    // optional phase cycle totals [grid][16] of the pipelined kernel (tools/tail_trace.py), or nullptr
#endif
};
void launch_mnist_tail_mfma(const MnistTailArgs& a, hipStream_t s);   // dg_tail_mnist.hip
// does this launch run mnist_tail_pipe3_kernel (the only one with a Batchnorm form)?
inline bool mnist_tail_runs_pipe3(const MnistTailArgs& a) { return a.pipe > 0 && a.do_backward && a.C == 64 && a.n_rows >= 2 * a.pipe && a.pipe_version == 3; }

// ---- CelebA tail: Generator.6 (64 -> 3, 64x64) + tanh + loss + backward to da5 ----------------
struct CelebaTailArgs {
    float* h5;           // [N,32,32,C] in: Generator.5 output (no nonlinearity); out: da5
    const float* F6;     // [5,5,3,C]
    const float* F6p;    // forward filter fragments in MFMA fragment order: 16x16x4 tiles per filter row kh (dg_engine.cpp)
    const float* b6;     // [3]
    const float* x;      // [B,64,64,3]
    float* loss_part;    // [N, nparts, 4] partial sums of squared error; nparts = 8 bands (fwd16) or 16 half-bands (role-split kernel); celeba_loss_finish_kernel adds them in a fixed order
    float* y;            // [N,64,64,3] or nullptr
    float* g6;           // [N,64,64,3] scratch for da6 (needed across band borders)
    int n_rows;
    int R;
    int C;
    int do_backward;
    int bwd_persist;     // backward tail: workgroups of the persistent pipelined kernel (> 0)
    int want_loss;       // 0: nobody reads the per-row loss of this launch (199 of the 200 forward passes of a projection): the
                         // forward tail skips its squared-error reduction and leaves loss_part untouched
    int fwd_split;       // forward tail (C = 64): > 0 = role-split persistent kernel on half-bands with this many workgroups; 0 = celeba_tail_fwd16_kernel
#ifdef DG_MEASURE
    int dbg;             // timing experiments only: 1 = skip the gather phase, 2 = skip the GEMM phase
    int bwd_bands;       // bwd_persist == 0: per-band backward kernel, 4-input-row bands per workgroup (1, 2 or 4)
    int fwd16;           // forward tail: 1 = 16x16x4 kh-aligned formulation (F6p = its pack), 0 = 32x32x2 (F6p = the 32-wide pack)
    long long* trace;    // optional per-workgroup phase cycle totals [grid][8], or nullptr
    int prio;            // wave priority per workgroup slot (see wg_priority); 0 = all equal
#endif
};
void launch_celeba_tail_fwd_mfma(const CelebaTailArgs& a, hipStream_t s);  // dg_tail_mnist.hip
void launch_celeba_tail_bwd_mfma(const CelebaTailArgs& a, hipStream_t s);
void launch_celeba_loss_finish(const float* loss_part, float* loss, int n_rows, int nparts, int P, hipStream_t s);   // loss = sum(parts) / P

// ---- small kernels ----------------------------------------------------------------------------
// m = momentum*m + sum_s part[n][s][:];  z -= lr*m     (ApplyMomentum, gan.py:389-391)
void launch_momentum_update(float* z, float* m, const float* part, int nsplit, int64_t n_elems_rows,
                            int latent, float lr, float momentum, float* dz_out, hipStream_t s);
// first-argmin over R restarts + gather (gan.py:438-449)
void launch_select(const float* loss, const float* y, int B, int R, int P, float* out_rec, int32_t* out_idx,
                   hipStream_t s);
// z ~ N(0, std^2), Philox4x32-10 keyed by (seed), counter (global row, column/4)
void launch_init_latents(float* z, int64_t n_rows, int latent, uint64_t seed, int64_t first_row, float std,
                         hipStream_t s);

// fragment order (dg_types.h) -> plain rows: dst[n * row_floats + f], n < n_rows (debug reads, tests)
void launch_unfrag(const float* src, float* dst, int64_t n_rows, int64_t row_floats, hipStream_t s);

// ---- BatchNorm with batch statistics (tflib/ops/batchnorm.py:80-93) ---------------------------
// An activation buffer viewed as [rows, C] (rows = latent rows x positions); per-column mean / biased variance,
// eps 1e-5, statistics reduced in float64.
struct BnArgs {
    float* a;            // fwd: [relu](bn(pre)) out | bwd: masked dy in, da out (in place)
    float* xhat;         // [rows, C] the layer's PRE-ACTIVATIONS (written by the GEMM, read by fwd and bwd; xhat = (pre - mean) * rstd
                         // is re-formed from them where it is needed)
    double* part;        // [nblk <= bn_max_blocks(), 2, C] scratch
    float* fstats;       // [2, C] mean, rstd (written by fwd)
    float* bstats;       // [2, C] mean(dy), mean(dy*xhat) (written by bwd)
    const float* scale;  // [C]
    const float* offset; // [C]
    int64_t rows;
    int C;
};
int bn_max_blocks();        // row blocks of the partial sums, upper bound: part holds bn_max_blocks() * 2 * C doubles
void launch_bn_forward(const BnArgs& a, int relu, hipStream_t s);
// the forward pass when the GEMM epilogue (EPI_BIAS_STATS) has left per-block column sums in `block_sums` [nblk][2][C] (float):
// finalize (float64) + apply -- no pass over the pre-activations for the statistics.  The block sums are those of (pre - shift[c]):
// the producing GEMM takes them before it adds its bias (shift = that bias, per column), so that the float32 sums of x and x^2
// carry no bias-sized offset
// relu < 0: statistics only (a.fstats), nothing is applied -- the consumer of the layer does that itself
void launch_bn_forward_from_blocks(const BnArgs& a, const float* block_sums, int nblk, int relu, hipStream_t s, const float* shift);
void launch_bn_backward(const BnArgs& a, hipStream_t s);
// the backward pass when the producer of dy (a GEMM epilogue in EPI_MASK_STATS, or the MNIST tail in its Batchnorm form) has left
// per-block column sums of dy and dy * xhat in `block_sums` [nblk][2][C] (float): fold + finalize (float64) + apply
void launch_bn_backward_from_blocks(const BnArgs& a, const float* block_sums, int nblk, hipStream_t s);

}  // namespace dg
