#!/usr/bin/env python
"""Static instruction mix of the K loops of the GEMM kernel instantiations, from hipcc's own assembly (no GPU needed).

    python tools/gemm_loop_isa.py [substring of the demangled kernel name ...] > profiles/rNN_gemm_loop_isa.txt

A K loop = a loop (LLVM's own block annotations) that holds MFMA instructions; one iteration = one 32-float K chunk
of the job's tile shape (dg_gemm.hip run_job: one loop per tile shape an instantiation can run).  Per loop: instructions by class -- what DESIGN 4.1 states about the inner loop (per chunk and
wave 64 v_mfma_f32_32x32x2_f32 on a 128x128 tile, 16 ds_read_b128, 8 LDS-DMA loads, one barrier) can be read off here, and so can
the difference between the two forms (PAIR = true / false) of an instantiation that jobs.pair_kernel chooses between."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "defensegan_amd", "csrc")

CLASSES = [("mfma", r"v_mfma_"), ("ds_read", r"ds_read|ds_load"), ("ds_write", r"ds_write|ds_store"),
           ("lds_dma", r"buffer_load_dword.*\blds\b"), ("vmem_load", r"buffer_load|global_load"), ("vmem_store", r"buffer_store|global_store"),
           ("barrier", r"s_barrier"), ("waitcnt", r"s_waitcnt"), ("sched", r"sched_|s_nop|s_setprio"), ("branch", r"s_cbranch|s_branch"),
           ("valu", r"v_"), ("salu", r"s_")]


def classify(ins):
    for name, pat in CLASSES:
        if re.match(pat, ins) or (name == "lds_dma" and re.search(pat, ins)):
            return name
    return "other"


def main():
    wanted = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dg_gemm.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", "-I", CSRC,
                               os.path.join(CSRC, "dg_gemm.hip"), "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    kernels, cur = [], None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = [m.group(1), []]
            kernels.append(cur)
        elif cur is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur[1].append(ln)
    names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in kernels), capture_output=True, text=True).stdout.splitlines()
    cols = ["mfma", "ds_read", "lds_dma", "vmem_load", "vmem_store", "barrier", "waitcnt", "valu", "salu", "total"]
    print("%-44s %4s  " % ("kernel", "loop") + " ".join("%9s" % c for c in cols))
    for (mangled, body), name in zip(kernels, names):
        short = re.sub(r"\(.*\)$", "", name.replace("dg::(anonymous namespace)::", "").replace("void ", ""))
        if "gemm_batched_kernel" not in short or (wanted and not any(w in short for w in wanted)):
            continue
        # LLVM annotates every block with the loop it belongs to: "=>This Inner Loop Header" / "in Loop: Header=BBx_y"
        loops, order, cur_loop = {}, [], None
        for ln in body:
            m = re.match(r"^(?:(\.LBB\d+_\d+):|; %bb\.\d+:)(.*)$", ln)
            if m:
                note = m.group(2)
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
                if "This Inner Loop Header" in note or "This Loop Header" in note:
                    cur_loop = m.group(1)
                elif h:
                    cur_loop = ".L" + h.group(1)
                else:
                    cur_loop = None
                if cur_loop is not None and cur_loop not in loops:
                    loops[cur_loop] = []
                    order.append(cur_loop)
            elif cur_loop is not None and ln.startswith("\t") and not ln.startswith("\t.") and not ln.lstrip().startswith(";"):
                loops[cur_loop].append(ln.strip().split(";")[0].strip())
        n_loop = 0
        for label in order:
            ins = loops[label]
            cnt = {}
            for i in ins:
                c = classify(i)
                cnt[c] = cnt.get(c, 0) + 1
            if not cnt.get("mfma"):
                continue
            n_loop += 1
            cnt["total"] = len(ins)
            print("%-44s %4d  " % (short, n_loop) + " ".join("%9d" % cnt.get(c, 0) for c in cols))

if __name__ == "__main__":
    main()
