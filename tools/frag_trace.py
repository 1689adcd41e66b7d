"""Per-wave cycle stamps of one dg_fgemm.hip launch (measurement build: DG_FRAG_TRACE_KCH / DG_FRAG_TRACE_FILE) -> phase statistics.
usage: python tools/frag_trace.py <trace.bin>"""
import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
t = t[t[:, 4] != 0]
t0 = t[:, 0].min()
clk = 2400.0  # cycles per us (approx)
st, lp0, lp1, red, en = [(t[:, i] - t0) / clk for i in range(5)]
ks = (t[:, 6] >> 8) & 0xff; q = (t[:, 6] >> 16) & 0xff; steps = t[:, 7]
hw = t[:, 5]; cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; xcc = t[:, 6] & 15
print("waves", len(t), "span us", en.max())
print("prologue (start->loop start) mean %.2f" % (lp0 - st).mean())
for k in (1, 2, 4):
    m = ks == k
    if not m.any(): continue
    per_step = (lp1[m] - lp0[m]) / steps[m]
    print("ks", k, "waves", m.sum(), "loop us/step mean %.3f p10 %.3f p90 %.3f ; loop us mean %.1f" % (per_step.mean(), np.percentile(per_step, 10), np.percentile(per_step, 90), (lp1[m]-lp0[m]).mean()))
    print("   after-loop (loop end -> wave end) mean %.2f p50 %.2f p90 %.2f max %.2f" % ((en[m]-lp1[m]).mean(), np.percentile(en[m]-lp1[m], 50), np.percentile(en[m]-lp1[m], 90), (en[m]-lp1[m]).max()))
    m0 = m & (q == 0)
    print("   q0: reduce (loop end -> epilogue start) mean %.2f p90 %.2f ; epilogue mean %.2f p90 %.2f" % ((red[m0]-lp1[m0]).mean(), np.percentile(red[m0]-lp1[m0], 90), (en[m0]-red[m0]).mean(), np.percentile(en[m0]-red[m0], 90)))
# skew inside a workgroup: loop end spread
wg = np.arange(len(t)) // 4 if len(t) % 4 == 0 else None
