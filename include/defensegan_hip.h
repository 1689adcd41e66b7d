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
 *   stream   hipStream_t (NULL = default stream); the call is asynchronous on it
 *
 * Stateless per call (fresh z / momentum each batch = model_eval_gan's behaviour,
 * /root/reference/utils/gan_defense.py:119); any B >= 1 is accepted (ragged last batch).
 */
int dg_reconstruct(dg_handle* h, const float* x, const float* z0, uint64_t seed, int64_t first_row,
                   int B, int R, int L, float lr, float momentum,
                   float* out_rec, int32_t* out_idx, float* out_loss, float* out_z, void* stream);

/* G(z): z [N, latent] -> y [N, H, W, C].  (generator_fn(z, is_training=False), gan.py:399) */
int dg_generate(dg_handle* h, const float* z, int N, float* out_y, void* stream);

/*
 * One forward + backward-to-z at fixed z (the body of the loop without the update, gan.py:409-417):
 * x [B,H,W,C], z [B*R, latent] -> out_y [B*R,H,W,C] (may be NULL), out_loss [B*R] (may be NULL),
 * out_dz [B*R, latent] = d(sum_j loss_j)/dz (may be NULL).
 */
int dg_loss_grad(dg_handle* h, const float* x, const float* z, int B, int R,
                 float* out_y, float* out_loss, float* out_dz, void* stream);

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

/* Tuning override for experiments: key=value pairs, e.g. "tile.F2=64x128". */
int dg_set_option(dg_handle* h, const char* key, const char* value);

#ifdef __cplusplus
}
#endif
#endif /* DEFENSEGAN_HIP_H */
