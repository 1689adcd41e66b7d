// Probe of the production gathered-GEMM kernel (dg_gemm.hip) under synthetic plans, to separate
// memory-system effects from kernel-structure effects.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I defensegan_amd/csrc tools/ubench/gemm_probe.hip \
//         defensegan_amd/csrc/dg_gemm.hip defensegan_amd/csrc/dg_plan.cpp -o tools/ubench/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dg_kernels.h"
#include "dg_plan.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static long long* g_trace = nullptr;
static double run(const dg::LayerPlan& p, int tile, int mode, int n_rows, float* A, float* W, float* Out, float* bias,
                  long long a_rowstride_override, int reps, int xcd_map) {
    dg::PosEntry* dpos; dg::TapEntry* dtaps;
    CHECK(hipMalloc(&dpos, p.pos.size() * sizeof(dg::PosEntry)));
    CHECK(hipMalloc(&dtaps, p.taps.size() * sizeof(dg::TapEntry)));
    CHECK(hipMemcpy(dpos, p.pos.data(), p.pos.size() * sizeof(dg::PosEntry), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dtaps, p.taps.data(), p.taps.size() * sizeof(dg::TapEntry), hipMemcpyHostToDevice));
    dg::GemmArgs a;
    a.A = A; a.W = W; a.Out = Out; a.bias = bias; a.pos = dpos; a.taps = dtaps;
    a.a_rowstride = a_rowstride_override >= 0 ? a_rowstride_override : p.a_rowstride;
    a.out_rowstride = p.out_rowstride; a.w_rowstride = p.w_rowstride; a.kch = p.kch; a.n_rows = n_rows;
    const int bm = dg::gemm_tile_bm(tile);
    a.n_mtiles = (n_rows + bm - 1) / bm; a.mode = mode; a.n_pos = (int)p.pos.size(); a.xcd_map = xcd_map; a.lds_pad = 0; a.clk = nullptr; a.trace = g_trace; a.queue = nullptr; a.persist_wgs_per_cu = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dg::launch_gemm(tile, a, a.n_pos, 0);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) dg::launch_gemm(tile, a, a.n_pos, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipGetLastError());
    hipFree(dpos); hipFree(dtaps);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2560;
    const size_t abytes = (size_t)N * 12544 * 4 * 2;
    float *A, *W, *Out, *bias;
    CHECK(hipMalloc(&A, abytes)); CHECK(hipMalloc(&Out, abytes)); CHECK(hipMalloc(&W, 8 << 20)); CHECK(hipMalloc(&bias, 1 << 16));
    CHECK(hipMemset(A, 0, abytes)); CHECK(hipMemset(Out, 0, abytes)); CHECK(hipMemset(W, 0, 8 << 20)); CHECK(hipMemset(bias, 0, 1 << 16));
    if (argc > 2) {   // timeline dump of one layer: ./gemm_probe N F2|B3|B2 > csv
        const int tile = 3, bn = dg::gemm_tile_bn(tile);
        dg::LayerPlan p = !strcmp(argv[2], "F2") ? dg::plan_deconv_fwd(4, 4, 7, 7, 256, 128, bn)
                        : !strcmp(argv[2], "B3") ? dg::plan_deconv_bwd(7, 7, 14, 14, 128, 64, bn)
                                                 : dg::plan_deconv_bwd(4, 4, 7, 7, 256, 128, bn);
        const int mode = !strcmp(argv[2], "F2") ? dg::EPI_BIAS_RELU : dg::EPI_MASK;
        const int grid = (int)p.pos.size() * ((N + 63) / 64);
        CHECK(hipMalloc(&g_trace, (size_t)grid * 4 * sizeof(long long)));
        double ms = run(p, tile, mode, N, A, W, Out, bias, -1, 2, 0);
        std::vector<long long> h((size_t)grid * 4);
        CHECK(hipMemcpy(h.data(), g_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        long long t0 = h[0];
        for (int i = 0; i < grid; ++i) if (h[i * 4] < t0) t0 = h[i * 4];
        printf("# %s N=%d grid=%d kernel_us=%.1f\n", argv[2], N, grid, ms * 1e3);
        for (int i = 0; i < grid; ++i)
            printf("%d,%lld,%lld,%lld,%lld\n", i, h[i * 4] - t0, h[i * 4 + 1] - t0, h[i * 4 + 2], h[i * 4 + 3]);
        return 0;
    }
    {   // steady-state check: exactly one resident wave of workgroups (256 CUs x 3), every tile 1504 chunks long
        for (int tile : {3, 2, 1, 0}) {
            const int bn = dg::gemm_tile_bn(tile), bm = dg::gemm_tile_bm(tile);
            const int wg_per_cu = tile == 3 ? 5 : (tile == 0 ? 2 : 3);
            dg::LayerPlan p = dg::plan_deconv_fwd(4, 4, 7, 7, 256, 128, bn);
            p.out_rowstride = 128;          // keep the synthetic output inside the buffer
            p.pos.resize(wg_per_cu);
            p.taps.assign(188, dg::TapEntry{0, 0});
            for (auto& pe : p.pos) { pe.tap_begin = 0; pe.tap_count = 188; pe.n0 = 0; pe.out_off = 0; }
            const int rows = 256 * bm;
            double ms = run(p, tile, dg::EPI_BIAS_RELU, rows, A, W, Out, bias, 0, 3, 0);
            double fl = 2.0 * 188 * 256 * bn * (double)rows * wg_per_cu;
            printf("steady tile%d (%d WG/CU, K=188 taps, hot)      : %8.1f us %6.1f TF\n", tile, wg_per_cu, ms * 1e3, fl / ms / 1e9);
            // same total work in 8x shorter tiles (8 waves of workgroups)
            p.pos.resize(wg_per_cu * 8);
            for (auto& pe : p.pos) { pe.tap_begin = 0; pe.tap_count = 23; pe.n0 = 0; pe.out_off = 0; }
            ms = run(p, tile, dg::EPI_BIAS_RELU, rows, A, W, Out, bias, 0, 3, 0);
            fl = 2.0 * 23 * 256 * bn * (double)rows * wg_per_cu * 8;
            printf("steady tile%d (%d WG/CU x8 waves, K=23 taps)   : %8.1f us %6.1f TF\n", tile, wg_per_cu, ms * 1e3, fl / ms / 1e9);
            p.pos.resize(wg_per_cu * 32);
            for (auto& pe : p.pos) { pe.tap_begin = 0; pe.tap_count = 5; pe.n0 = 0; pe.out_off = 0; }
            ms = run(p, tile, dg::EPI_BIAS_RELU, rows, A, W, Out, bias, 0, 3, 0);
            fl = 2.0 * 5 * 256 * bn * (double)rows * wg_per_cu * 32;
            printf("steady tile%d (%d WG/CU x32 waves, K=5 taps)   : %8.1f us %6.1f TF\n", tile, wg_per_cu, ms * 1e3, fl / ms / 1e9);
        }
    }
    struct L { const char* name; dg::LayerPlan p; int mode; };
    for (int tile : {3}) {
        const int bn = dg::gemm_tile_bn(tile);
        std::vector<L> layers;
        layers.push_back({"F2", dg::plan_deconv_fwd(4, 4, 7, 7, 256, 128, bn), dg::EPI_BIAS_RELU});
        layers.push_back({"B3", dg::plan_deconv_bwd(7, 7, 14, 14, 128, 64, bn), dg::EPI_MASK});
        layers.push_back({"B2", dg::plan_deconv_bwd(4, 4, 7, 7, 256, 128, bn), dg::EPI_MASK});
        for (auto& l : layers) {
            const double fl = 2.0 * l.p.macs_per_row * N;
            double ms = run(l.p, tile, l.mode, N, A, W, Out, bias, -1, 5, 0);
            printf("%s tile%d real plan              : %8.1f us %6.1f TF\n", l.name, tile, ms * 1e3, fl / ms / 1e9);
            dg::LayerPlan hot = l.p;
            for (auto& t : hot.taps) { t.a_off = 0; t.w_off = 0; }
            ms = run(hot, tile, l.mode, N, A, W, Out, bias, 0, 5, 0);
            printf("%s tile%d real K, hot operands   : %8.1f us %6.1f TF\n", l.name, tile, ms * 1e3, fl / ms / 1e9);
            ms = run(hot, tile, dg::EPI_STORE, N, A, W, Out, bias, 0, 5, 0);
            printf("%s tile%d real K, hot, EPI_STORE : %8.1f us %6.1f TF\n", l.name, tile, ms * 1e3, fl / ms / 1e9);
            dg::LayerPlan uni = l.p;
            int mean = 0; for (auto& pe : uni.pos) mean += pe.tap_count; mean = (mean + (int)uni.pos.size() / 2) / (int)uni.pos.size();
            double macs = 0;
            for (auto& pe : uni.pos) { if (pe.tap_begin + mean > (int)uni.taps.size()) pe.tap_begin = (int)uni.taps.size() - mean; pe.tap_count = mean; macs += (double)mean * uni.kch * uni.bn; }
            ms = run(uni, tile, l.mode, N, A, W, Out, bias, -1, 5, 0);
            printf("%s tile%d uniform K (%2d), real   : %8.1f us %6.1f TF  (%d tiles)\n", l.name, tile, mean, ms * 1e3, 2.0 * macs * N / ms / 1e9,
                   (int)uni.pos.size() * ((N + dg::gemm_tile_bm(tile) - 1) / dg::gemm_tile_bm(tile)));
            for (auto& t : uni.taps) { t.a_off = 0; t.w_off = 0; }
            ms = run(uni, tile, l.mode, N, A, W, Out, bias, 0, 5, 0);
            printf("%s tile%d uniform K (%2d), hot    : %8.1f us %6.1f TF\n", l.name, tile, mean, ms * 1e3, 2.0 * macs * N / ms / 1e9);
        }
    }
    return 0;
}
