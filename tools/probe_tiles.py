#!/usr/bin/env python
"""GPU experiment helper: per-kernel timings of the projection loop under engine options.

    python tools/probe_tiles.py --L 12 --set "jobs.tune=0" --set "jobs.slack=1e30,jobs.min_level=0" ...
Each --set is one configuration (comma-separated key=value engine options); prints avg us / TFLOP/s per kernel.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="mnist")
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--R", type=int, default=10)
ap.add_argument("--L", type=int, default=12)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--zero", action="store_true", help="all-zero weights (DVFS experiment)")
args = ap.parse_args()

a = archs.make_arch(args.arch)
params = synth.make_weights(args.arch, seed=1234, gain=2.0)
if args.zero:
    params = {k: np.zeros_like(v) for k, v in params.items()}
configs = args.set or [""]
for cfg in configs:
    gan = dataset_gan_dict[args.arch](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=args.R, rec_iters=args.L, device=0)
    gan.set_weights(params)
    for kv in [c for c in cfg.split(",") if c]:
        k, v = kv.split("=")
        gan.set_option(k, v)
    zt = gan.init_latents(args.B, seed=1)
    x = gan.generate(zt)
    x = torch.clamp(x + 0.3 * torch.sign(torch.randn_like(x)), a.in_lo, a.in_hi)
    gan.reconstruct(x, seed=1)
    torch.cuda.synchronize()
    gan.profile_reset(); gan.profile_enable(1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.reps):
        gan.reconstruct(x, seed=2)
    torch.cuda.synchronize()
    prof = gan.profile_read()
    gan.profile_enable(0)
    ev0.record()
    for _ in range(args.reps):
        gan.reconstruct(x, seed=2)
    ev1.record(); torch.cuda.synchronize()
    tot = ev0.elapsed_time(ev1) / args.reps
    line = {p["name"].split("@")[0]: (round(p["ms"] / p["launches"] * 1e3, 1), round(p["flops"] / p["ms"] / 1e9, 1)) for p in prof if p["launches"]}
    it_us = sum(v[0] for v in line.values())
    print(json.dumps({"cfg": cfg, "per_iter_us": round(it_us, 1), "loop_ms_unprofiled": round(tot, 2), "kernels": line }), flush=True)
    gan.close()
