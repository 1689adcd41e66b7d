#!/usr/bin/env python
"""GPU check: the position-batched GEMM (dg_gemm2.hip) against the per-position kernel (dg_gemm.hip) -- same summation order
per output element, so rec / loss / z must be BIT-identical -- for both architectures, ragged and large batches, and every
job-cutting policy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from defensegan_amd import archs, synth
from tests.helpers import make_gan

bad = 0
for arch, use_bn, sizes in (("mnist", False, (1, 7, 50, 256)), ("celeba", False, (3, 40, 128)), ("mnist", True, (6,)), ("celeba", True, (5,))):
    a = archs.make_arch(arch)
    R = 10
    ref_gan, p = make_gan(arch, gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=3, use_bn=use_bn)
    ref_gan.set_option("gemm2", 0)
    for B in sizes:
        rs = np.random.RandomState(B)
        x = np.asarray(ref_gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
        x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=8)
        z0 = synth.make_z(B * R, 128, seed=9)
        ref = ref_gan.reconstruct(x, z_init_val=z0, return_details=True)
        for opts in ({}, {"jobs.slack": 1e30}, {"jobs.slack": 0.01}, {"jobs.slack": 1.0, "jobs.min_level": 1}, {"jobs.min_level": 2}):
            g2, _ = make_gan(arch, gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=3, use_bn=use_bn)
            g2.set_option("gemm2", 1)
            for k, v in opts.items():
                g2.set_option(k, v)
            got = g2.reconstruct(x, z_init_val=z0, return_details=True)
            ok = all(np.array_equal(got[k], ref[k]) for k in ("rec", "idx", "loss", "z"))
            err = max(float(np.abs(got[k].astype(np.float64) - ref[k]).max()) for k in ("rec", "loss", "z"))
            print("%-7s bn=%d B=%4d %-40s %s  max|diff| %.3g" % (arch, use_bn, B, opts, "BIT-IDENTICAL" if ok else "DIFFERENT", err), flush=True)
            bad += 0 if ok else 1
            g2.close()
    ref_gan.close()
print("FAILED: %d configurations differ" % bad if bad else "all configurations bit-identical")
sys.exit(1 if bad else 0)
