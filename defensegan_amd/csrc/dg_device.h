// Device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dg {

// Workgroups that share a CU run the same phase sequence (wait for staged data, multiply, write out) and, started together,
// stay in lock step: they all wait at the same time and then compete for the matrix pipe at the same time.  Giving every
// resident workgroup its own wave priority lets the CU's arbiter finish one workgroup's multiply phase before the next
// one's, which staggers the phases: one workgroup's waits sit under another one's MFMAs.
// The priority comes from the workgroup's slot on the CU (HW_ID.TG_ID, bits 19:16 on gfx9): uniform over the waves of a
// workgroup (waves of one workgroup never wait for each other across priorities), distinct among the residents.
//   mode 1: slot & 3      mode 2: 3 * (slot & 1)      mode 3: 3 - (slot & 3)
// Measured on MI355X (round 2): the priorities do take effect (in the GEMM kernel, mode 1, a job in slot 3 ran 1.54 us per K
// chunk, one in slot 0 2.17 us; equal priorities 1.77-2.02 us) but the launches take the same time to within noise (MNIST
// 980.7 vs 980.8 img/s; CelebA forward tail 186 -> 190-193 us): the kernels are throughput-bound, not phase-locked.  The
// hook was REMOVED from the GEMM kernel again: even switched off, its presence changed the register allocation of the
// ReluGrad-epilogue instantiation (109 / 71 instead of 107 / 81 VGPRs / SGPRs) and cost 1.5-2.5 % on B3 / B2.  It stays in
// the CelebA tails (option tail_prio, off), whose timings did not move.
__device__ __forceinline__ void wg_priority(int mode) {
    if (mode == 0) return;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned slot = (hwid >> 16) & 15u;
    const unsigned pr = mode == 1 ? (slot & 3u) : mode == 2 ? 3u * (slot & 1u) : 3u - (slot & 3u);
    switch (pr) {
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
    }
}

}  // namespace dg
