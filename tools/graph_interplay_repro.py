#!/usr/bin/env python
"""Reproducer (ROCm 7.2, MI355X) of why the replayed loop graph (engine option graph_max_rows) is off by default: a graph the
CALLER captured of a dg_reconstruct call replays with wrong results once one of the engine's own graph replays has run between
its capture and its replay -- eager launches in between are harmless, the engine's own replays are always right.
    python tools/graph_interplay_repro.py plain|sync|noepoch|want_on_side|replay_twice|nograph      (nograph: the control)"""
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.helpers import make_gan
mode = sys.argv[1]
B, R, L = 24, 5, 6
gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
gan.set_option("graph_max_rows", 0 if mode == "nograph" else 1024)
dev = torch.device("cuda", 0)
x = gan.generate(gan.init_latents(B, seed=3)).contiguous()
z0 = gan.init_latents(B * R, seed=4)
gan.prepare(B)
eager = gan.reconstruct(x, z_init_val=z0, return_details=True)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        cap = gan.reconstruct(x, z_init_val=z0, return_details=True)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
r1 = torch.equal(cap["rec"], eager["rec"])
x2 = gan.generate(gan.init_latents(B, seed=5)).contiguous() if mode != "noepoch" else (x * 0.7 + 0.1).contiguous()
if mode == "sync_before_want":
    torch.cuda.synchronize()
if mode == "want_on_side":
    with torch.cuda.stream(side):
        want = gan.reconstruct(x2, z_init_val=z0, return_details=True)
    torch.cuda.synchronize()
else:
    want = gan.reconstruct(x2, z_init_val=z0, return_details=True)
if mode == "sync":
    torch.cuda.synchronize()
x.copy_(x2)
g.replay()
torch.cuda.synchronize()
if mode == "replay_twice":
    first = cap["rec"].clone()
    g.replay()
    torch.cuda.synchronize()
    print("second replay == first replay", torch.equal(first, cap["rec"]))
r2 = torch.equal(cap["rec"], want["rec"]) and torch.equal(cap["loss"], want["loss"])
want3 = gan.reconstruct(x2, z_init_val=z0, return_details=True)
torch.cuda.synchronize()
g0, _ = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
g0.set_option("graph_max_rows", 0)
ref2 = g0.reconstruct(x2, z_init_val=z0, return_details=True)
torch.cuda.synchronize()
print("cap==result for the OLD x:", torch.equal(cap["rec"], eager["rec"]), "x now == x2:", torch.equal(x, x2))
print("want==ref", torch.equal(want["rec"], ref2["rec"]), "cap==ref", torch.equal(cap["rec"], ref2["rec"]), "cap loss==ref", torch.equal(cap["loss"], ref2["loss"]), "cap z==ref", torch.equal(cap["z"], ref2["z"]), float((cap["rec"]-ref2["rec"]).abs().max()))
print(mode, "replay_equal", r1, "new_input_equal", r2, "want==want3", torch.equal(want["rec"], want3["rec"]), "cap==want3", torch.equal(cap["rec"], want3["rec"]))
