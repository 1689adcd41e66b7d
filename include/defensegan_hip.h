/*
 * defensegan_hip.h -- C ABI of the MI355X-native Defense-GAN latent-projection engine.
 *
 * The reference has NO FFI for this path: the boundary is a Python method on a TF-graph-owning
 * object,
 *     DefenseGANBase.reconstruct(self, images, batch_size=None, back_prop=True,
 *                                reconstructor_id=0, z_init_val=None)
 *                                              /root/reference/models/gan.py:333-449
 * configured by attributes rec_iters / rec_rr / rec_lr / latent_dim / net_dim / use_bn / image_dim
 *                                              /root/reference/models/gan.py:41-68
 * with weights arriving through load_generator()   /root/reference/models/gan.py:80-87.
 * These entry points are what a ctypes binding of that method binds instead of building the
 * tf.while_loop graph (INTEGRATION.md shows the stub).  Plain pointers and sizes only; every tensor
 * argument of the compute calls is a DEVICE pointer (HIP), caller-owned, fp32 unless stated, in the
 * reference's layouts (NHWC images, Linear W[in,out], Deconv2D filters [5,5,Cout,Cin]).
 *
 * All functions return 0 on success, a negative DG_E_* code otherwise; dg_last_error() returns a
 * thread-local message.  A handle is bound to one device and is not thread-safe.
 */
#ifndef DEFENSEGAN_HIP_H
#define DEFENSEGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_ABI_VERSION 1

/* architectures: /root/reference/models/dataset_models.py:36-71 (mnist, f-mnist), :127-165 (celeba) */
#define DG_ARCH_MNIST28 0
#define DG_ARCH_CELEBA64 1

#define DG_OK 0
#define DG_E_INVALID (-1)      /* bad argument / unsupported configuration            */
#define DG_E_STATE (-2)        /* weights missing, handle destroyed ...               */
#define DG_E_HIP (-3)          /* a HIP runtime call failed (message has the detail)  */
#define DG_E_NOMEM (-4)

typedef struct dg_handle dg_handle;

int dg_version(void);
const char* dg_last_error(void);

/* number of HIP devices visible; <0 on error (no driver / no GPU). */
int dg_device_count(void);

/* Fills name (<= name_len bytes, NUL-terminated), CU count and HBM bytes of `device`. */
int dg_device_info(int device, char* name, int name_len, int* cu_count, int64_t* hbm_bytes);

/*
 * Replaces GAN(cfg).generator_fn construction (gan.py:657-665 mnist, :726-735 celeba).
 * latent_dim: LATENT_DIM (default.yml:4), net_dim: NET_DIM (default.yml:7), use_bn: USE_BN (default.yml:3).
 * Constraints of this build: latent_dim % 64 == 0, net_dim % 64 == 0.
 */
int dg_create(int arch, int latent_dim, int net_dim, int use_bn, int device, dg_handle** out);
int dg_destroy(dg_handle* h);

/*
 * Replaces load_generator() / tf.train.Saver restore of the `Generator*` variables
 * (gan.py:80-87, base_model.py:294-335).  `name` is the tflib parameter name
 * (tflib/__init__.py:9-33): "Generator.Input.W" [latent, 4*4*4*net_dim], "Generator.Input.b",
 * "Generator.{2,3,5,6}.Filters" [5,5,Cout,Cin], "Generator.N.Biases",
 * "Generator.BN{1,2,3}.{scale,offset}".  `data` holds prod(shape) fp32 values; is_device != 0 means
 * a device pointer on the handle's device, else host memory.  The engine keeps its own copy.
 */
int dg_set_weights(dg_handle* h, const char* name, const float* data, const int64_t* shape,
                   int ndim, int is_device);

/* 1 when every weight the configuration needs has been set. */
int dg_weights_complete(dg_handle* h);

/*
 * The hot path: R restarts x L momentum-GD steps on sum_j mean_pix (G(z_j) - x_j)^2, then the
 * per-image first-argmin over restarts -- DefenseGANBase.reconstruct, gan.py:333-449.
 *
 *   x        [B, H, W, C]  images already in generator range ([0,1] mnist, [-1,1] celeba; gan.py:684-685, 764-765)
 *   z0       [B*R, latent] initial latents, row j = b*R + r (gan.py:348-359), or NULL: rows are then drawn
 *            N(0, 1/latent) (gan.py:370-375) from a counter-based generator keyed by (seed, first_row + j),
 *            i.e. independent of how the image list is batched or sharded over GPUs
 *   L        rec_iters: L forwards, L-1 applied updates; L = 0 and L = 1 both return G(z0) (gan.py:409-437)
 *   lr       rec_lr, constant over the loop (the reference's decay never fires, gan.py:362-386)
 *   momentum 0.7 in the reference (gan.py:390), non-Nesterov, slots zero-initialised
 *   out_rec  [B, H, W, C]  G(z_{L-1}) of the selected restart
 *   out_idx  [B] int32     selected restart r*(b) in [0,R)  (first minimum, gan.py:438-445)   (may be NULL)
 *   out_loss [B*R]         image_rec_loss of every restart at step L-1                        (may be NULL)
 *   out_z    [B*R, latent] z_{L-1}                                                            (may be NULL)
 *   stream   hipStream_t (NULL = default stream).  Once the call shape (B, R) has been prepared -- by dg_prepare or by an
 *            earlier call of the same shape -- the call only enqueues kernels and copies on `stream`: it neither allocates
 *            device memory nor waits for the device, and may be recorded by a stream capture.  The FIRST call of a new shape
 *            prepares itself and therefore BLOCKS (see dg_prepare); under a stream capture it fails with DG_E_STATE instead.
 *
 * Stateless per call (fresh z / momentum each batch = model_eval_gan's behaviour,
 * /root/reference/utils/gan_defense.py:119); any B >= 1 is accepted (ragged last batch).
 */
int dg_reconstruct(dg_handle* h, const float* x, const float* z0, uint64_t seed, int64_t first_row,
                   int B, int R, int L, float lr, float momentum,
                   float* out_rec, int32_t* out_idx, float* out_loss, float* out_z, void* stream);

/*
 * Prepares calls of B images x R restarts (no reference counterpart: the reference builds its static graph for exactly
 * batch_size * rec_rr rows at this point, gan.py:345-377): sizes the workspace for B*R latent rows and builds the job list of
 * every GEMM layer for this shape -- with "jobs.tune" (default) by timing the candidate lists on the device, ~0.1-0.3 s.
 * Blocking: allocates, launches timing runs on `stream` and waits for them.  Afterwards dg_reconstruct / dg_loss_grad with
 * this (B, R) and dg_generate with N = B*R are asynchronous in the strict sense documented at dg_reconstruct.  Shapes stay
 * prepared until an option that changes the lists is set (at most 16 row counts are kept per layer, oldest dropped first).
 * Results never depend on which list was chosen (tiles are only cut along M / N): preparing is about latency only.
 * A handle serves ONE stream at a time: dg_prepare (and the self-preparation of a call with a new shape) waits for the
 * device and then times launches on the handle's own activation buffers -- work of the same handle still queued on another
 * stream is waited for, not overlapped.
 */
int dg_prepare(dg_handle* h, int B, int R, void* stream);
/* Number of row groups a dg_reconstruct call of B images x R restarts runs as (1, or option "two_streams" when the call is large
 * enough -- with "auto" the timed choice of this shape, which exists once the shape has been prepared or called; 1 before that:
 * the groups are whole images, run concurrently on the engine's own streams forked from / joined to the caller's, and the
 * results do not depend on the split); -1 on an invalid shape (dg_last_error). */
int dg_call_row_groups(dg_handle* h, int B, int R);

/* G(z): z [N, latent] -> y [N, H, W, C], N <= 2^20 rows per call.  (generator_fn(z, is_training=False), gan.py:399) */
int dg_generate(dg_handle* h, const float* z, int N, float* out_y, void* stream);

/*
 * One forward + backward-to-z at fixed z (the body of the loop without the update, gan.py:409-417):
 * x [B,H,W,C], z [B*R, latent] -> out_y [B*R,H,W,C] (may be NULL), out_loss [B*R] (may be NULL),
 * out_dz [B*R, latent] = d(sum_j loss_j)/dz (may be NULL).
 */
int dg_loss_grad(dg_handle* h, const float* x, const float* z, int B, int R,
                 float* out_y, float* out_loss, float* out_dz, void* stream);

/*
 * The job list a GEMM layer runs with for a row count is chosen by timing candidates on first use (dg_prepare), so two
 * processes can settle on different -- equally correct, bit-identical in their results -- lists.  dg_export_tuning writes the
 * choices made so far as text (one line per layer and row count: starting level, cutting threshold, order variant) into
 * buf and returns the bytes needed including the terminating NUL (call with buf = NULL to size); dg_import_tuning rebuilds
 * exactly those lists in another handle of the same configuration without timing and returns the number of lists installed
 * (>= 0) or a negative DG_E_* code.  Use: the ranks of a multi-GPU run import rank 0's text; a profiler pass imports the
 * bench's (defensegan_amd/gan.py: export_tuning / import_tuning / the DG_TUNING_CACHE file).
 */
int64_t dg_export_tuning(dg_handle* h, char* buf, int64_t cap);
int dg_import_tuning(dg_handle* h, const char* text);

/* Fills z [n_rows, latent] with the same N(0, std^2) draw dg_reconstruct uses for z0 == NULL
 * (std <= 0 selects sqrt(1/latent)). */
int dg_init_latents(dg_handle* h, float* z, int64_t n_rows, uint64_t seed, int64_t first_row,
                    float std, void* stream);

/*
 * Measurement hooks (bench.py's roofline leg).  When enabled, every dg_reconstruct brackets each
 * kernel family with hipEvents on the launch stream; dg_profile_read returns, for family `i`
 * (0 <= i < count), its name, number of launches and total milliseconds since the last reset.
 */
int dg_profile_enable(dg_handle* h, int on);
int dg_profile_count(dg_handle* h);
int dg_profile_read(dg_handle* h, int i, char* name, int name_len, int64_t* launches, double* total_ms,
                    double* flops);
int dg_profile_reset(dg_handle* h);

/* Copies an internal activation buffer of the last forward ("h1","h2","h3","h5" ...) to `dst`
 * (device pointer, capacity n floats); returns the number of floats copied.  Test hook only. */
int64_t dg_debug_read(dg_handle* h, const char* what, float* dst, int64_t n);

/*
 * Tuning / measurement overrides (no reference counterpart; defaults are the measured optimum on MI355X):
 *   "jobs.tune"        1 (default): on first use of a row count every GEMM layer times its candidate job lists on the real
 *                      operands and keeps the fastest; 0: the cost model's simulated makespan decides
 *   "jobs.slack"       > 0 fixes the cutting threshold of the job lists (dg_plan.h build_jobs; 1e30 = whole tiles only)
 *   "jobs.min_level"   >= 0 forces every list to start cut to halves (1) / quarters (2)
 *   "jobs.xcd_head"    fraction (default 0 = off; measured, no gain) of a list that is ALSO offered to the timing in XCD-locality order: every
 *                      XCD gets the jobs of one contiguous range of latent rows, all tap classes of a row range next to each other
 *                      in time, so that their shared input rows stay in that XCD's L2 (a permutation of the list)
 *   "jobs.balance"     1 (default): layers that fit ONE dispatch round are also offered to the timing as lists whose jobs are placed
 *                      -- and, where the round has room, cut -- so that every CU carries the same predicted work (dg_plan.h
 *                      balance_order, jobs_balanced; +2.2 % at the reference's 500 rows, profiles/r05_ab_list_orders.txt)
 *   "jobs.pair_kernel" the kernel has an instantiation with and one without the K-pair hand-off code; lists without pairs can run on
 *                      either (same arithmetic, another register allocation): 1 (default) the two fastest lists are timed on both
 *                      and the faster form is kept (+0.3 % on the MNIST loop, profiles/r05_ab_pair_kernel.txt), 0 never, 2 always
 *   "jobs.prio"        wave priorities by predicted job length (s_setprio per job): 0 (default) never, 1 the fastest lists are timed
 *                      again with them, 2 always.  Measured: the launches last the same (profiles/r05_ab_prio.txt)
 *   "jobs.spread"      1: the first dispatch round of a multi-round list mixes all job lengths (dg_plan.h spread_order).  Default 0:
 *                      slower on every layer (profiles/r05_ab_list_orders.txt)
 *   "bn_fused"         with use_bn, where the Batchnorm sums come from.  1: the forward statistics are per-32-row-block column sums taken
 *                      in the producing GEMM's epilogue; 2 (default): so are the backward sums (of dy and dy * xhat) of the layers
 *                      normalised over rows x positions, taken in the epilogue of the GEMM that writes dy (which re-forms the ReLU gate
 *                      and xhat from the layer's pre-activations), and inside the projection loop the MNIST tail takes its Batchnorm form
 *                      (it reads the last Batchnorm layer's pre-activations, applies relu(bn(.)) itself and leaves that layer's backward
 *                      sums: the layer's activation image is not written in those steps -- dg_debug_read("act2") then returns the image
 *                      of the last launch that did write it); 0: separate passes (float64 sums from the first add)
 *   "jobs.slots0/1", "jobs.rate0..2", "jobs.fixed_us"   cost-model parameters
 *   "lr_schedule"      "constant" (default): lr == rec_lr at every step, which is what the reference executes -- the step variable
 *                      of its decay is never advanced (gan.py:362-386, 416-417); "intended": the schedule its code asks for,
 *                      tf.train.exponential_decay(rec_lr, k, ceil(0.8 * rec_iters), 0.1, staircase=True) (base_model.py:186-192)
 *   "nsplit"           split-K factor of the Linear backward (default 16)
 *   "two_streams"      number of concurrent row groups (0 = off, 2..8) of calls with at least "two_stream_min_rows" (1024) latent
 *                      rows, or "auto": two groups for the call shapes where that is faster -- timed once per (B, R) when the shape
 *                      is prepared (nine loop steps of each form, three times; kept when >= 1 % faster).  Default "auto" for CelebA
 *                      (+2-3 % at 1280 rows: its gather-bound tails run beside the other group's GEMMs; -4.5 % at 5120), 0 for
 *                      MNIST (-0.9 %).  Results are bit-identical either way (rows are independent).  "two_stream_split": two
 *                      groups, percent of the images in the first (default halves; unequal halves measured slower).  While the
 *                      per-launch profile is on (dg_profile_enable) the groups run one after the other on the caller's stream
 *   "latent_turn"      1 (default): Linear backward / forward on the weight-stationary kernels (dg_linear.hip) where the shapes
 *                      allow; 0: on the position-batched kernel like every other layer.  "lin_groups_fwd" / "lin_groups_bwd":
 *                      their workgroups per column tile / K slice (0 = from the CU count)
 *   "update_fold"      1: the momentum update rides in the Linear backward launch -- the workgroup that delivers the last K slice
 *                      of a 32-row block sums the slices (write-through partials, one arrival counter per block) and applies the
 *                      update; needs latent_turn, latent 128, nsplit a multiple of 8.  Bit-identical to the separate kernel and
 *                      1.6 % SLOWER on the MNIST loop (the reducer of a row group becomes the last arriver of every later block of
 *                      the group: profiles/r04_ab_update_fold.txt).  Default 0
 *   "turn_fused"       1: the whole latent turn -- Linear backward, momentum update and the NEXT step's Linear forward -- is ONE launch
 *                      (dg_turn.hip): the nsplit workgroups of a row group meet at two barriers of their own (write-through partials
 *                      and latents, one monotonic arrival counter per barrier, cleared at the start of every call).  Needs
 *                      latent_turn, latent 128, K slices of 256 features, no Batchnorm, frag_path 0.  Bit-identical to the three
 *                      launches and AT PARITY with them (MNIST -0.5 ... +0.2 %, -1.5 % beside CelebA's second row group: the seam
 *                      inside the launch costs what the two kernel boundaries did, profiles/r06_ab_turn_fused.txt).  A barrier poll
 *                      that does not end within ~0.5 s gives up (the next dg_reconstruct on the handle fails with DG_E_HIP) instead
 *                      of hanging the device.  Residency is guaranteed for the row groups of ONE call (two workgroups fit a CU, the
 *                      engine sizes the launches of concurrent row groups for that); do not run more than two handles with this
 *                      option at once on one device.  Default 0
 *   "graph_max_rows"   > 0: call shapes of at most this many latent rows replay a captured hipGraph of the L-step loop instead of
 *                      enqueuing its launches one by one (built on the first call with a new (B, R, L, lr, momentum); never while
 *                      the caller's stream is itself capturing).  Default 0 = always enqueue: measured no gain at the reference's
 *                      default batch (500 rows: 780 vs 779 img/s), and on ROCm 7.2 a graph the CALLER captured of a call on the same
 *                      handle replays wrongly after an internal replay ran in between (tools/graph_interplay_repro.py) -- do not
 *                      combine the two on one handle
 *   "frag_path"        1: the FORWARD deconvs run on the LDS-free fragment-order kernel (dg_fgemm.hip: both operands in MFMA fragment
 *                      order, fetched straight into registers; the Linear forward writes fragment order, the backward GEMMs take
 *                      their ReluGrad gates from one bit per element).  Agrees with the default path to float32 rounding (another
 *                      fixed summation tree per tap class: a tile's K axis is split over 1 / 2 / 4 waves), rows stay independent of
 *                      their batch.  Default 0: its loop is faster in isolation (147-149 vs 137-142 TFLOP/s) but the path is 2-10 %
 *                      SLOWER at 1280 ... 12 500 rows (profiles/r06_frag_path_ab.txt).  Not with use_bn or two_streams
 *   "debug.poison_pair_counters"   test hook: leaves every K-pair arrival counter of every prepared job list at `value` -- the state a
 *                      launch that died between a pair's two arrivals would leave.  Every call clears the counters of the lists it
 *                      launches, so the next call must not care (tests/test_gpu_variants.py)
 *   "tail_pipe"        MNIST tail: workgroups of the pipelined kernel (0 = fused per-row kernel)
 *   "tail_fwd_split"   CelebA forward tail (NET_DIM 64): workgroups of the role-split persistent kernel (default 512 = two per CU);
 *                      0 = the per-band kernel (celeba_tail_fwd16_kernel; same y and da6 bit for bit, the per-row loss to rounding)
 *   "tail_bwd_persist" CelebA backward tail: workgroups of the persistent kernel
 * Only in the measurement build of the library (same sources with -DDG_MEASURE -> libdefensegan_hip_measure.so; the product
 * library refuses them): "tail_pipe_version" = 2 (second-generation MNIST tail), "tail_fwd16" = 0 (32x32x2 CelebA forward tail), "tail_bwd_persist" = 0 + "tail_bwd_bands" (per-band
 * backward tail), "tail_trace", "tail_dbg", "tail_prio", "job_trace" (in-kernel phase traces and phase-removal switches, tools/)
 * Every job-list / launch-shape option leaves the results bit-identical (tests/test_gpu_variants.py): GEMM tiles are only
 * ever cut along M and N; along K only the classes of >= 80 K chunks, and those ALWAYS, in two fixed halves whose sums are added
 * once (K-pair jobs, dg_plan.h kPairMinChunks: a + b = b + a, so the arrival order does not show); "nsplit", "tail_fwd16" and
 * "bn_fused" and "frag_path" change a summation order (agreement to rounding).
 */
int dg_set_option(dg_handle* h, const char* key, const char* value);

/* =====================================================================================================
 * The step after the projection: classifier forward + the per-batch reduction of
 * model_eval_gan.  Replaces, for evaluation, `model(reconstructed_tensors)` over the cleverhans-style MLP
 * of /root/reference/utils/network_builder.py:129-331 (models A-F, :333-521) and the `acc_value / cur_preds /
 * diff_op` fetches of /root/reference/utils/gan_defense.py:91-179, /root/reference/blackbox.py:569-572.
 * ===================================================================================================== */
typedef struct dg_clf dg_clf;

#define DG_LAYER_CONV2D 0     /* p0..p5 = output_channels, kernel h, kernel w, stride h, stride w, 1 = "SAME" / 0 = "VALID" */
#define DG_LAYER_RELU 1
#define DG_LAYER_LINEAR 2     /* p0 = num_hid                                                                    */
#define DG_LAYER_FLATTEN 3
#define DG_LAYER_SOFTMAX 4
#define DG_LAYER_DROPOUT 5    /* identity: K.learning_phase() is 0 at evaluation (network_builder.py:296-297)     */

/* An empty MLP for NHWC inputs [*, in_h, in_w, in_c] (network_builder.py:129-162). */
int dg_clf_create(int device, int in_h, int in_w, int in_c, dg_clf** out);
int dg_clf_destroy(dg_clf* h);
/* Appends a layer; returns its index (>= 0) or a negative DG_E_* code.  Shapes follow tf.nn.conv2d's SAME / VALID
 * rules (SAME: out = ceil(in / stride), pad_before = pad_total / 2). */
int dg_clf_add_layer(dg_clf* h, int kind, int p0, int p1, int p2, int p3, int p4, int p5);
/* Width of the last layer's output per image (the class count once the model is complete). */
int dg_clf_output_width(dg_clf* h);
/* Parameters of a Conv2D (kernels [kh,kw,cin,cout], network_builder.py:213-221) or Linear (W [in,out], :196-203) layer
 * and its bias; host or device pointers. */
int dg_clf_set_weights(dg_clf* h, int layer, const float* W, const int64_t* wshape, int wndim, const float* b,
                       int64_t blen, int is_device);
/* x [B,in_h,in_w,in_c] -> logits [B,n] (the layer before Softmax, may be NULL) and probs [B,n] (may be NULL; equals the
 * logits when the model has no Softmax).  Device pointers; asynchronous on `stream`. */
int dg_clf_forward(dg_clf* h, const float* x, int B, float* logits, float* probs, void* stream);
/* One evaluation batch: preds[b] = argmax_k model(rec)[b,k] (first maximum), *n_correct += #(preds == labels) (device
 * int32, caller zeroes it), diffs[b] = mean((orig[b] - rec[b])^2).  labels/preds/diffs/n_correct/orig may be NULL. */
int dg_eval_batch(dg_clf* h, const float* rec, const float* orig, const int32_t* labels, int B, int32_t* preds,
                  float* diffs, int32_t* n_correct, void* stream);

/*
 * The step before the path: the FGSM inputs of /root/reference/whitebox.py:198-210 and
 * /root/reference/blackbox.py:521-534 (cleverhans FastGradientMethod, ord = inf; cleverhans itself is an un-pinned,
 * empty submodule of the reference, its published fgm is restated):
 *     x_adv = clip(x + eps * sign(d CE(softmax(logits(x)), y) / dx), clip_min, clip_max)
 * y = labels, or the model's own prediction when labels == NULL (cleverhans' default, avoids label leaking).
 */
int dg_clf_input_gradient(dg_clf* h, const float* x, const int32_t* labels, int B, float* grad, void* stream);
int dg_fgsm(dg_clf* h, const float* x, const int32_t* labels, int B, float eps, float clip_min, float clip_max,
            float* x_adv, void* stream);

/*
 * The path's one collective: every rank projects and classifies its contiguous shard of the image list
 * (no collective on the data path) and ONE all_gather assembles the evaluation message -- per rank `count` int32 words, e.g.
 * [n | labels (cap) | preds (cap) | diffs (cap, float32 bits)] as defensegan_amd/gan_defense.py:model_eval_gan_sharded builds it
 * (the reduction of /root/reference/utils/gan_defense.py:166-179 over ranks; the reference itself is single-process).  RCCL
 * over xGMI, reachable WITHOUT torch.distributed: librccl.so is opened on first use.  Rank 0 draws a unique id, the caller
 * carries its 128 bytes to the other ranks (file, MPI, environment), every rank creates its communicator -- one process per GPU.
 * send [count], recv [nranks * count]: device pointers; asynchronous on `stream`.
 */
typedef struct { char internal[128]; } DgUniqueId;          /* = ncclUniqueId */
typedef struct dg_comm dg_comm;
int dg_comm_unique_id(DgUniqueId* id);
int dg_comm_create(int nranks, const DgUniqueId* id, int rank, int device, dg_comm** out);
int dg_comm_destroy(dg_comm* c);
int dg_gather_eval(dg_comm* c, const int32_t* send, int32_t* recv, int64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEFENSEGAN_HIP_H */
