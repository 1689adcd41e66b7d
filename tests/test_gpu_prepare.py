"""-m gpu: the asynchronous contract of dg_reconstruct (include/defensegan_hip.h): once a call shape has been prepared
(dg_prepare) the call neither allocates device memory nor waits for the device -- shown by recording it into a graph under
HIP's global capture mode, where any hipMalloc / synchronisation from the capturing thread fails the capture -- and an
UNPREPARED shape fails loudly under capture instead of corrupting it.  Runs in a child process (a broken capture must not
take the test session's context with it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import json, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from defensegan_amd import synth, _native
from tests.helpers import make_gan
res = {}
B, R, L = 24, 5, 6
gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
dev = torch.device("cuda", 0)
x = gan.generate(gan.init_latents(B, seed=3)).contiguous()
z0 = gan.init_latents(B * R, seed=4)
gan.prepare(B)
eager = gan.reconstruct(x, z_init_val=z0, return_details=True)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):                       # capture mode "global": a hipMalloc here is an error
        cap = gan.reconstruct(x, z_init_val=z0, return_details=True)
torch.cuda.synchronize()
for t in cap.values():
    t.zero_()
g.replay()
torch.cuda.synchronize()
res["replay_equal"] = all(bool(torch.equal(cap[k], eager[k])) for k in ("rec", "idx", "loss", "z"))
# new inputs through the same graph: the captured call reads the caller's buffers at replay time
x2 = gan.generate(gan.init_latents(B, seed=5)).contiguous()
want = gan.reconstruct(x2, z_init_val=z0, return_details=True)
x.copy_(x2)
g.replay()
torch.cuda.synchronize()
res["replay_new_input_equal"] = bool(torch.equal(cap["rec"], want["rec"]) and torch.equal(cap["loss"], want["loss"]))
# seeded latents (z0 == NULL) are drawn inside the call: capturable too
gan.prepare(7)
xs = x[:7].contiguous()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g2, stream=side):
        cap2 = gan.reconstruct(xs, seed=11, first_row=70, return_details=True)
g2.replay()
torch.cuda.synchronize()
want2 = gan.reconstruct(xs, seed=11, first_row=70, return_details=True)
res["seeded_equal"] = bool(torch.equal(cap2["rec"], want2["rec"]) and torch.equal(cap2["idx"], want2["idx"]))
# an unprepared shape under capture: refused with a message, the capture is not poisoned by an allocation
g3 = torch.cuda.CUDAGraph()
msg = ""
try:
    with torch.cuda.stream(side):
        with torch.cuda.graph(g3, stream=side):
            gan.reconstruct(x[:13].contiguous(), z_init_val=z0[:13 * R].contiguous())
except _native.NativeError as e:
    msg = str(e)
except Exception as e:                                            # torch may wrap the failure of the capture itself
    msg = "other: " + str(e)
res["unprepared_message"] = msg
torch.cuda.synchronize()
after = gan.reconstruct(x2, z_init_val=z0, return_details=True)    # the handle still works, eagerly, afterwards
res["usable_after"] = bool(torch.equal(after["rec"], want["rec"]))
print("RESULT " + json.dumps(res))
"""


def test_prepared_call_is_capturable_and_unprepared_capture_fails_loudly():
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["replay_equal"] and res["replay_new_input_equal"] and res["seeded_equal"], res
    assert "dg_prepare" in res["unprepared_message"] and "captured" in res["unprepared_message"], res
    assert res["usable_after"], res
