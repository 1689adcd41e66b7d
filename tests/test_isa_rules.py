"""CPU: static rules on hipcc's gfx950 assembly of the kernel sources (tools/check_isa.py; hipcc cross-compiles without a GPU).

Two code-generation accidents of round 6 that no CPU test could have seen and that the GPU tests only caught by luck of timing:
a 16-byte buffer store with a REGISTER soffset (no wait state before the next VALU write of its data registers: silent data
corruption under load), and the collapse of dg_fgemm.hip's operand ring (the K loop waiting for nearly every load in flight)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_detector_sees_the_hazardous_store_form():
    t = _tool()
    lines = ["_Z1kv:", "\tbuffer_store_dwordx4 v[0:3], v112, s[8:11], s36 offen", "\tv_add_f32_e32 v0, v4, v24",
             "\tbuffer_store_dwordx4 v[4:7], v112, s[8:11], 0 offen", "\tbuffer_store_dwordx2 v[4:5], v112, s[8:11], s36 offen",
             "\tbuffer_store_dwordx3 v[4:6], off, s[8:11], s3", "\t.end_amdhsa_kernel"]
    bad = t.store_hazards(lines)
    assert [b[1].split()[0] for b in bad] == ["buffer_store_dwordx4", "buffer_store_dwordx3"] and "s36" in bad[0][1]


def test_no_kernel_of_the_library_breaks_the_rules():
    t = _tool()
    for src in t.SOURCES:
        lines = t.assembly(src)
        assert t.store_hazards(lines) == [], src
        if src == "dg_fgemm.hip":
            loops = t.fgemm_loops(lines)
            assert len(loops) >= 4
            for name, mfma, wmin, branches in loops:
                assert mfma == 128 and branches == 0 and wmin is not None and wmin >= 12, (name, mfma, wmin, branches)
