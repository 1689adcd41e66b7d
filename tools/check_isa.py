#!/usr/bin/env python
"""Static checks of hipcc's gfx950 assembly of the kernel sources (no GPU needed): two things that cost a day in round 6.

1. The 16-byte-store data hazard.  `buffer_store_dwordx3/x4` with an SGPR soffset followed by a VALU write of its data registers:
   LLVM's hazard recognizer inserts no wait state for that form (it assumes the hazard needs an immediate soffset), gfx950 has it:
   the Linear forward's fragment-order stores lost element 0 of lanes 12-15 of every 16 whenever a second workgroup delayed the
   store's issue (tests/test_gpu_frag.py found it).  Rule: no dwordx3 / dwordx4 buffer store with a register soffset in any kernel.
2. The operand ring of dg_fgemm.hip.  Its K loop must be ONE basic block whose waits leave at least two k8-steps of loads in
   flight (vmcnt >= 12 for the 2 x 4 tile): hipcc's wait insertion merges the loop's back edge with the prologue path, and a
   prologue whose loads are issued in another order (or a scalar load inside the loop) silently collapses the ring to vmcnt(0-2).

    python tools/check_isa.py            -> prints findings, exit status 1 when a rule is broken
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "defensegan_amd", "csrc")
SOURCES = ["dg_gemm.hip", "dg_fgemm.hip", "dg_linear.hip", "dg_turn.hip", "dg_small.hip", "dg_bn.hip", "dg_clf.hip", "dg_tail_mnist.hip", "dg_tail_celeba.hip"]


def assembly(src, defines=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", "-I", CSRC,
                               os.path.join(CSRC, src), "-o", out] + list(defines), stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def kernels_of(lines):
    out, cur = [], None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = [m.group(1), []]
            out.append(cur)
        elif cur is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur[1].append(ln)
    return out


def store_hazards(lines):
    """(kernel, instruction) of every 12- / 16-byte buffer store whose soffset is a register."""
    bad = []
    for name, body in kernels_of(lines):
        for ln in body:
            m = re.match(r"\s*buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*(?:v\d+|off),\s*s\[\d+:\d+\],\s*(\S+)", ln)
            if m and re.match(r"^(s\d+|m0|ttmp\d+)", m.group(1)):
                bad.append((name, ln.strip()))
    return bad


def fgemm_loops(lines):
    """Per fgemm kernel: (name, MFMAs in the K loop, smallest vmcnt waited for inside it, branches inside it)."""
    out = []
    for name, body in kernels_of(lines):
        if "fgemm_" not in name:
            continue
        # the K loop: the loop (header annotation .. its backward branch) that holds the MFMAs
        best = None
        for start in [i for i, ln in enumerate(body) if "Loop Header" in ln]:
            end = next((i for i in range(start, len(body)) if re.match(r"\s*s_cbranch_(scc|vcc|exec)", body[i])), len(body) - 1)
            loop = body[start:end + 1]
            n = sum("v_mfma" in ln for ln in loop)
            if best is None or n > best[0]:
                best = (n, loop)
        if best is None:
            out.append((name, 0, None, 0))
            continue
        n, loop = best
        waits = [int(m.group(1)) for ln in loop for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", ln)] if m]
        out.append((name, n, min(waits) if waits else None, sum("s_cbranch" in ln for ln in loop) - 1))
    return out


def main():
    rc = 0
    for src in SOURCES:
        lines = assembly(src)
        for name, ins in store_hazards(lines):
            print("HAZARD %s: %s: %s" % (src, name, ins))
            rc = 1
        if src == "dg_fgemm.hip":
            for name, mfma, wmin, br in fgemm_loops(lines):
                ok = mfma == 128 and wmin is not None and wmin >= 12 and br == 0
                print("%s dg_fgemm.hip K loop %s: %d MFMAs, smallest vmcnt %s, %d inner branches" % ("ok    " if ok else "BROKEN", name, mfma, wmin, br))
                rc = rc if ok else 1
    print("no 16-byte buffer store with a register soffset" if rc == 0 else "rules broken")
    return rc


if __name__ == "__main__":
    sys.exit(main())
