#!/usr/bin/env python
"""Register / LDS footprint of every kernel in the product library, from hipcc's own metadata (no GPU needed).

    python tools/kernel_regs.py [-DDG_MEASURE] > profiles/rNN_kernel_regs.txt

DESIGN section 8.7: the code generation of the GEMM kernel is sensitive to anything at its entry, so after touching
dg_gemm.hip / GemmArgs the VGPR / SGPR counts of the hot instantiations are compared with the previous build's table."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "defensegan_amd", "csrc")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    rows = []
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith(".hip"):
                continue
            out = os.path.join(td, f + ".s")
            procs.append((f, out, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only",
                                                     "-I", CSRC] + flags + [os.path.join(CSRC, f), "-o", out], stderr=subprocess.DEVNULL)))
        for f, out, p in procs:
            if p.wait() != 0:
                raise SystemExit("hipcc failed on " + f)
            txt = open(out).read()
            for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?"
                                 r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
                rows.append((f, m.group(3), int(m.group(5)), int(m.group(1)), int(m.group(4)), int(m.group(2)), int(m.group(6))))
    names = demangle([r[1] for r in rows])
    print("%-18s %5s %5s %5s %7s %6s  %s" % ("file", "vgpr", "agpr", "sgpr", "lds_B", "spill", "kernel"))
    for r, n in sorted(zip(rows, names), key=lambda t: (t[0][0], t[1])):
        n = n.replace("dg::(anonymous namespace)::", "").replace("dg::", "")
        n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
        print("%-18s %5d %5d %5d %7d %6d  %s" % (r[0], r[2], r[3], r[4], r[5], r[6], n))


if __name__ == "__main__":
    main()
