#!/usr/bin/env python
"""Hand-over kit for the parity pin that cannot be made in the build container: runs the REFERENCE itself
(kabkabm/defensegan, Python 2.7 + TensorFlow 1.7) on the synthetic weights and inputs of this repository's parity tests and
writes small fixtures (inputs + the reference's outputs) that tests/test_tf_fixtures.py picks up when present.

Why: the reference holds no tests or golden vectors, and neither Python 2 nor TensorFlow 1.x exists in the build image, so
oracle/defensegan_oracle.py is pinned only internally ("parity unpinned").  A maintainer with a TF-1 machine runs

    python2 tools/make_tf_fixtures.py --reference /path/to/defensegan --out tests/golden

and commits tests/golden/tf_*.npz and tests/golden/tf_ckpt/ (a few hundred KB + one ~6 MB checkpoint).  Nothing of the
reference travels: only arrays it computed.  The script is self-contained (it does NOT import defensegan_amd) and runs under
Python 2.7 and 3.x; `--self-check` (no TensorFlow needed) prints the SHA-256 of every case's weights / z0 so that
tests/test_tf_fixtures.py can assert, on the CPU, that the generator embedded here still equals defensegan_amd/synth.py.

What is recorded per case (weights: tflib initialisers x gain from RandomState(wseed), as defensegan_amd.synth.make_weights;
z_true, z0 ~ N(0, 1/latent) from RandomState(zseed); x = the reference generator's own G(z_true), clean or +-0.3 sign noise):
  * for L in --iters (default 1 5 10): `rec_L`  = sess.run(model.reconstruct(x_pl, batch_size=B, z_init_val=z0)), the call the
    callers make (models/gan.py:333-449), with rec_rr = R restarts and the argmin selection inside the graph;
  * `rows_L` = the same call with rec_rr = 1 on the images tiled R times (row b*R + r), i.e. every restart's G(z_{L-1}): the
    per-restart losses mean((rows - x)^2) and the first-argmin index follow in NumPy (`loss_L`, `idx_L`);
  * one tf.train.Saver checkpoint of the generator variables (tf_ckpt/<case>/GAN.model-0) + the weights it holds, for the
    TensorFlow-free checkpoint reader (defensegan_amd/tf_checkpoint.py).
"""
from __future__ import print_function

import argparse
import hashlib
import json
import os
import sys

import numpy as np

# (name, arch, wseed, gain, bias_range, B, R, adversarial, zseed, use_bn): B*R rows per case; B is also the reference's
# batch_size, which must be divisible by rec_rr (models/gan.py:101-104).  The three regimes of the path in ONE run: no Batchnorm
# (the shipped configs, default.yml:3), USE_BN True on both architectures (batch statistics at inference, the `else` branch of
# tflib/ops/batchnorm.py:80-93 -- the B*R rows of a call share their statistics, so these cases also pin WHICH rows those are),
# and the CelebA generator; every case also leaves a tf.train.Saver checkpoint for the TensorFlow-free reader.
CASES = [
    ("mnist_clean", "mnist", 1234, 2.0, 0.1, 6, 3, False, 21, False),
    ("mnist_adv", "mnist", 1234, 3.0, 0.1, 4, 2, True, 22, False),
    ("fmnist_clean", "f-mnist", 4321, 2.0, 0.0, 4, 4, False, 23, False),
    ("celeba_clean", "celeba", 1234, 2.0, 0.1, 2, 2, False, 24, False),
    ("mnist_bn", "mnist", 1234, 2.0, 0.1, 6, 3, False, 25, True),
    ("celeba_bn", "celeba", 1234, 2.0, 0.1, 2, 2, False, 26, True),
]
BN_JITTER = 0.2
LATENT, NET_DIM = 128, 64


def deconvs(arch):
    """(name, cin, cout) per Deconv2D of models/dataset_models.py:36-71 (MNIST / F-MNIST) and :127-165 (CelebA)."""
    nd = NET_DIM
    if arch in ("mnist", "f-mnist"):
        return [("Generator.2", 4 * nd, 2 * nd), ("Generator.3", 2 * nd, nd), ("Generator.5", nd, 1)]
    return [("Generator.2", 4 * nd, 2 * nd), ("Generator.3", 2 * nd, nd), ("Generator.5", nd, nd), ("Generator.6", nd, 3)]


def image_dim(arch):
    return [28, 28, 1] if arch in ("mnist", "f-mnist") else [64, 64, 3]


def _uniform(rs, stdev, shape):
    lim = stdev * np.sqrt(3.0)
    return rs.uniform(low=-lim, high=lim, size=shape).astype(np.float32)


def bn_layers(arch):
    """(name, channels) of the generator's Batchnorm layers in registry order: BN1 over the 4096 Linear features (axes [0]),
    BN2 / BN3 per channel of Generator.2 / .3 (axes [0, 1, 2]); models/dataset_models.py:44-65, 135-156."""
    return [("Generator.BN1", 4 * 4 * 4 * NET_DIM), ("Generator.BN2", 2 * NET_DIM), ("Generator.BN3", NET_DIM)]


def make_weights(arch, seed, gain, bias_range, use_bn=False):
    """tflib initialisers (tflib/ops/linear.py:55-60 glorot, tflib/ops/deconv2d.py:46-76 he) x gain, drawn in layer order from
    RandomState(seed), biases after all filters, then (use_bn) scale = 1 + U(+-0.2) and offset = U(+-0.2) per Batchnorm layer --
    the same stream as defensegan_amd/synth.py:make_weights(..., use_bn, bn_jitter=0.2)."""
    rs = np.random.RandomState(seed)
    lin_out = 4 * 4 * 4 * NET_DIM
    w = {}
    names = ["Generator.Input.W"]
    w["Generator.Input.W"] = _uniform(rs, np.sqrt(2.0 / (LATENT + lin_out)), (LATENT, lin_out)) * np.float32(gain)
    for name, cin, cout in deconvs(arch):
        fan_in, fan_out = cin * 25 / 4.0, cout * 25.0
        w[name + ".Filters"] = _uniform(rs, np.sqrt(4.0 / (fan_in + fan_out)), (5, 5, cout, cin)) * np.float32(gain)
        names.append(name + ".Filters")
    bias_shapes = [("Generator.Input.b", (lin_out,))] + [(name + ".Biases", (cout,)) for name, _, cout in deconvs(arch)]
    for name, shp in bias_shapes:
        w[name] = rs.uniform(-bias_range, bias_range, size=shp).astype(np.float32) if bias_range > 0 else np.zeros(shp, np.float32)
        names.append(name)
    if use_bn:
        for name, c in bn_layers(arch):
            w[name + ".scale"] = (np.ones((c,), np.float32) + rs.uniform(-BN_JITTER, BN_JITTER, size=(c,)).astype(np.float32)).astype(np.float32)
            w[name + ".offset"] = rs.uniform(-BN_JITTER, BN_JITTER, size=(c,)).astype(np.float32)
            names += [name + ".scale", name + ".offset"]
    return w, names


def make_latents(zseed, B, R):
    rs = np.random.RandomState(zseed)
    std = np.sqrt(1.0 / LATENT)
    zt = (rs.standard_normal((B, LATENT)) * std).astype(np.float32)
    z0 = (rs.standard_normal((B * R, LATENT)) * std).astype(np.float32)
    if R >= 3:
        z0[2] = z0[1]            # a duplicated restart: the first-minimum tie-break of tf.argmin
    return zt, z0


def sign_noise(zseed, shape):
    return np.sign(np.random.RandomState(zseed + 1).standard_normal(shape)).astype(np.float32)


def digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def self_check():
    out = {}
    for name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn in CASES:
        w, names = make_weights(arch, wseed, gain, bias_range, use_bn)
        zt, z0 = make_latents(zseed, B, R)
        out[name] = {"weights": digest([w[k] for k in names]), "z_true": digest([zt]), "z0": digest([z0]),
                     "noise": digest([sign_noise(zseed, [B] + image_dim(arch))])}
    return out


def run_reference(ref_root, out_dir, iters):
    sys.path.insert(0, ref_root)
    import tensorflow as tf                     # TensorFlow 1.x (the reference targets 1.7, README.md:46)
    import tflib
    from models import gan as ref_gan

    for name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn in CASES:
        w, names = make_weights(arch, wseed, gain, bias_range, use_bn)
        zt, z0 = make_latents(zseed, B, R)
        dim = image_dim(arch)
        lo = -1.0 if arch == "celeba" else 0.0
        result = {"arch": arch, "wseed": wseed, "gain": gain, "bias_range": bias_range, "R": R, "lr": 10.0, "momentum": 0.7,
                  "z0": z0, "iters": np.asarray(iters), "use_bn": int(use_bn)}
        x = None
        for pass_no, (rr, batch, tag) in enumerate([(R, B, "rec"), (1, B * R, "rows")]):
            for L in iters:
                tf.reset_default_graph()
                tflib.delete_all_params()
                cls = {"mnist": ref_gan.MnistDefenseGAN, "f-mnist": ref_gan.FmnistDefenseDefenseGAN,
                       "celeba": ref_gan.CelebADefenseGAN}[arch]

                class NoData(cls):              # the constructor would read the dataset from disk; the path needs none
                    def _load_dataset(self):
                        pass

                # AbstractModel.__init__ (models/base_model.py:29-84) reads cfg['cfg_path'] for its checkpoint directory and
                # resolves every attribute it is not given from tf.app.flags / cfg (None otherwise): give it all of them
                cfg = {"cfg_path": "experiments/cfgs/gans/%s.yml" % {"f-mnist": "fmnist"}.get(arch, arch), "DATASET_NAME": arch,
                       "BATCH_SIZE": batch, "USE_BN": bool(use_bn), "LATENT_DIM": LATENT, "NET_DIM": NET_DIM, "REC_ITERS": L, "REC_RR": rr,
                       "REC_LR": 10.0, "IMAGE_DIM": dim}
                model = NoData(cfg=cfg, test_mode=True, verbose=False, dataset_name=arch, batch_size=batch, test_batch_size=batch,
                               use_bn=bool(use_bn), latent_dim=LATENT, net_dim=NET_DIM, rec_iters=L, rec_rr=rr, rec_lr=10.0, image_dim=dim,
                               mode="gp-wgan", gradient_penalty_lambda=10.0, train_iters=1, critic_iters=5, input_transform_type=0,
                               debug=False, test_again=False, loss_type="l2", attribute="gender", tensorboard_log=False,
                               output_dir=os.path.join(out_dir, "tf_scratch"), num_gpus=1)
                x_pl = tf.placeholder(tf.float32, shape=[batch] + dim)
                z_pl = tf.placeholder(tf.float32, shape=[batch * rr, LATENT])
                rec_op = model.reconstruct(x_pl, batch_size=batch, z_init_val=z_pl)
                g_op = model.generator_fn(tf.constant(zt), is_training=False)
                sess = model.sess
                sess.run(tf.global_variables_initializer())
                params = {v.name.split(":")[0].split("/")[-1]: v for v in model.generator_vars}
                for k in names:               # (Batchnorm scale / offset are stored with the keep_dims shape of the moments)
                    sess.run(tf.assign(params[k], w[k].reshape(params[k].shape.as_list())))
                if x is None:
                    x = sess.run(g_op).reshape([B] + dim).astype(np.float32)
                    if adv:
                        x = np.clip(x + np.float32(0.3) * sign_noise(zseed, x.shape), lo, 1.0).astype(np.float32)
                    result["x"] = x
                    ck = os.path.join(out_dir, "tf_ckpt", name)
                    if not os.path.isdir(ck):
                        os.makedirs(ck)
                    tf.train.Saver(var_list=model.generator_vars).save(sess, os.path.join(ck, "GAN.model"), global_step=0)
                    np.savez(os.path.join(ck, "weights.npz"), **w)
                feed_x = x if rr == R else np.repeat(x, R, axis=0)
                sess.run(tf.local_variables_initializer())
                out = sess.run(rec_op, feed_dict={x_pl: feed_x, z_pl: z0})
                result["%s_%d" % (tag, L)] = np.asarray(out, np.float32)
                model.close_session()
        for L in iters:
            rows = result["rows_%d" % L].reshape(B * R, -1).astype(np.float64)
            xt = np.repeat(x.reshape(B, -1).astype(np.float64), R, axis=0)
            loss = ((rows - xt) ** 2).mean(axis=1)
            result["loss_%d" % L] = loss
            result["idx_%d" % L] = loss.reshape(B, R).argmin(axis=1).astype(np.int32)
        np.savez_compressed(os.path.join(out_dir, "tf_%s.npz" % name), **result)
        print("wrote", os.path.join(out_dir, "tf_%s.npz" % name))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", help="checkout of kabkabm/defensegan (Python 2.7 + TensorFlow 1.7 environment)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--iters", type=int, nargs="+", default=[1, 5, 10])
    ap.add_argument("--self-check", action="store_true", help="print the SHA-256 of every case's weights / latents (no TensorFlow)")
    args = ap.parse_args()
    if args.self_check or not args.reference:
        print(json.dumps(self_check(), indent=1, sort_keys=True))
        return
    run_reference(os.path.abspath(args.reference), args.out, args.iters)


if __name__ == "__main__":
    main()
