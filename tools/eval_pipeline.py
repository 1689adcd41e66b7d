#!/usr/bin/env python
"""BASELINE configuration 5 in miniature, end to end on one GPU with synthetic weights: FGSM inputs from a substitute
classifier -> Defense-GAN projection -> black-box classifier -> model_eval_gan reduction (the flow of
/root/reference/blackbox.py:521-575).  Prints where the time goes; with random weights the accuracies are meaningless.
    python tools/eval_pipeline.py [n_images] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import gan_defense, network_builder as nb, synth
from defensegan_amd.gan import dataset_gan_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 250
R, L = 10, 200
gan = dataset_gan_dict["mnist"](cfg={"USE_BN": False}, test_mode=True, rec_rr=R, rec_iters=L)
gan.set_weights(synth.make_weights("mnist", seed=1234, gain=2.0))
bbox, sub = nb.model_a(), nb.model_e()
bbox.init_like_reference(seed=1)
sub.init_like_reference(seed=2)
x = gan.generate(gan.init_latents(n, seed=5)).contiguous()               # clean in-range images, on the device
sync = torch.cuda.synchronize
sync(); t0 = time.perf_counter()
labels = bbox(x).argmax(dim=1).to(torch.int32)                            # black-box predictions as "ground truth"
sync(); t1 = time.perf_counter()
x_adv = nb.FastGradientMethod(sub).generate(x, eps=0.3, clip_min=0.0, clip_max=1.0)
sync(); t2 = time.perf_counter()
c_adv, _, _ = gan_defense.model_eval_gan(None, bbox, x_adv, labels.cpu().numpy(), batch_size=batch)
sync(); t3 = time.perf_counter()
c_def, _, roc = gan_defense.model_eval_gan(gan.reconstruct, bbox, x_adv, labels.cpu().numpy(), batch_size=batch, rec_rr=R)
sync(); t4 = time.perf_counter()
print("images %d  batch %d  R %d  L %d" % (n, batch, R, L))
print("classify clean        %8.3f s" % (t1 - t0))
print("FGSM (substitute)     %8.3f s" % (t2 - t1))
print("eval undefended       %8.3f s   agreement %.3f" % (t3 - t2, c_adv / n))
print("eval defended         %8.3f s   agreement %.3f   -> %.1f images/s end to end, mean rec error %.3e"
      % (t4 - t3, c_def / n, n / (t4 - t3), float(roc[2].mean())))
