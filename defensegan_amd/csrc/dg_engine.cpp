// Host side of the C ABI (include/defensegan_hip.h): handle, weights, workspace, launch sequence.
//
// The launch sequence restates DefenseGANBase.reconstruct (/root/reference/models/gan.py:333-449) as a
// fixed per-iteration kernel chain on one HIP stream:
//   F1 Linear+ReLU -> Fk Deconv+ReLU ... -> tail (last Deconv + sigmoid/tanh + loss [+ its backward])
//   -> Bk Deconv backward (+ReluGrad, in place over the activation) ... -> B1 Linear backward (split-K)
//   -> momentum update.
// The L-th iteration runs the forward half only (the reference discards the L-th update), then the
// per-image first-argmin over restarts gathers the output.  No host round trip inside the loop.
#include "dg_engine.h"

#pragma GCC visibility push(hidden)
namespace dge {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// A failed launch (bad configuration, LDS limit, ...) is reported by the layer it happened in, not at the end of the call.
int launch_check(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DG_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return DG_OK;
}

// Fills h->ai (geometry of every activation buffer) from the architecture and use_bn.
void describe_activations(dg_handle* h) {
    const int nd = (int)h->dec.size();
    if ((int)h->ai.size() == nd) return;        // geometry depends on (arch, use_bn) only; keep the BN buffers
    h->ai.assign(nd, ActInfo());
    ActInfo& a0 = h->ai[0];
    a0.pitch = 4; a0.valid = 4; a0.C = h->lin_out / 16; a0.row_floats = h->lin_out;
    a0.has_bn = h->use_bn != 0; a0.bn_name = "Generator.BN1"; a0.bn_C = h->lin_out; a0.bn_rows = 1;
    for (int d = 0; d + 1 < nd; ++d) {
        const DeconvSpec& s = h->dec[d];
        ActInfo& a = h->ai[d + 1];
        a.has_bn = h->use_bn && s.bn[0] != 0;
        a.valid = s.e_used;
        // BN statistics cover the full 2h x 2h map; the MNIST crop comes after the ReLU (dataset_models.py:52-59)
        a.pitch = a.has_bn ? 2 * s.h_in : s.e_used;
        a.C = s.cout;
        a.row_floats = (int64_t)a.pitch * a.pitch * a.C;
        a.bn_name = s.bn;
        a.bn_C = s.cout;
        a.bn_rows = (int64_t)a.pitch * a.pitch;
    }
}

int build_plans(dg_handle* h) {
    describe_activations(h);
    {
        GemmOp& op = h->F1;
        op.name = "F1";
        op.mode = h->use_bn ? (h->bn_fused ? dg::EPI_BIAS_STATS : dg::EPI_BIAS) : dg::EPI_BIAS_RELU;
        int rc = upload_batched(op, dg::plan_linear_fwd(h->latent, h->lin_out, h->lin_out));
        if (rc) return rc;
    }
    {
        GemmOp& op = h->B1;
        op.name = "B1";
        op.mode = dg::EPI_STORE;
        int rc = upload_batched(op, dg::plan_linear_bwd(h->latent, h->lin_out, h->nsplit, h->latent));
        if (rc) return rc;
    }
    const int nd = (int)h->dec.size();
    for (auto* v : {&h->Fd, &h->Bd})
        for (auto& o : *v) free_batched(o);
    h->Fd.assign(nd - 1, GemmOp());
    h->Bd.assign(nd - 1, GemmOp());
    for (int d = 0; d + 1 < nd; ++d) {
        const DeconvSpec& s = h->dec[d];
        const ActInfo& in = h->ai[d];
        const ActInfo& out = h->ai[d + 1];
        {
            GemmOp& op = h->Fd[d];
            op.name = std::string("F") + s.name[10];     // "Generator.N" -> "FN"
            op.mode = out.has_bn ? (h->bn_fused ? dg::EPI_BIAS_STATS : dg::EPI_BIAS) : (s.act == 0 ? dg::EPI_BIAS_RELU : dg::EPI_BIAS);
            // with BN every stored position is computed (statistics need the cropped row/column too)
            int rc = upload_batched(op, dg::plan_deconv_fwd(in.valid, in.pitch, out.has_bn ? out.pitch : out.valid, out.pitch,
                                                            s.cin, s.cout, s.cout));
            if (rc) return rc;
        }
        {
            GemmOp& op = h->Bd[d];
            op.name = std::string("B") + s.name[10];
            op.mode = dg::EPI_MASK;      // every backward output lands on a ReLU activation (h1, h2, h3)
            // behind a Batchnorm over (rows x positions) the epilogue also leaves that layer's backward sums (BN1 normalises
            // every (position, channel) feature by itself -- its columns are not the GEMM's: separate pass)
            op.bn_act = -1;
            if (in.has_bn && in.bn_rows > 1 && h->bn_fused >= 2) { op.mode = dg::EPI_MASK_STATS; op.bn_act = d; }
            // the incoming gradient is non-zero on the whole stored map after a BN backward, else only on the used block
            dg::LayerPlan base = dg::plan_deconv_bwd(in.valid, in.pitch, out.has_bn ? out.pitch : out.valid, out.pitch, s.cin,
                                                     s.cout, s.cin);
            if (in.pitch > in.valid) dg::plan_add_zero_positions(base, in.valid, in.pitch, s.cin);
            int rc = upload_batched(op, base);
            if (rc) return rc;
        }
    }
    return DG_OK;
}

// arrival counters of the fused latent turn (dg_turn.hip): one block per concurrent row group, then the error word
constexpr size_t turn_bar_bytes() { return ((size_t)dg_handle::kMaxGroups * dg_handle::kTurnBarWords + 16) * sizeof(unsigned); }

void free_workspace(dg_handle* h) {
    auto fr = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    fr(h->z); fr(h->m); fr(h->part); fr(h->loss); fr(h->y); fr(h->g6); fr(h->loss_part); fr(h->xbuf);
    h->xbuf_floats = 0;
    ++h->list_epoch;                 // captured loops point into these buffers
    for (auto& a : h->act) fr(a);
    for (auto& a : h->actf) fr(a);
    for (auto& g : h->gate) if (g) { (void)hipFree(g); g = nullptr; }
    for (auto& a : h->ai) { a.buf = nullptr; fr(a.xhat); fr(a.block_sums); a.block_cap = 0; }
    if (h->bn_part) { (void)hipFree(h->bn_part); h->bn_part = nullptr; }
    fr(h->tail_bn_sums); h->tail_bn_sums_wgs = 0;
    if (h->upd_count) { (void)hipFree(h->upd_count); h->upd_count = nullptr; }
    if (h->turn_bar) { (void)hipFree(h->turn_bar); h->turn_bar = nullptr; }
    h->cap_rows = 0;
}

int ensure_workspace(dg_handle* h, int64_t rows) {
    if (rows <= h->cap_rows) return DG_OK;
    HIP_TRY(hipDeviceSynchronize());
    free_workspace(h);
    const int64_t cap = rows;
    HIP_TRY(hipMalloc(&h->z, cap * h->latent * sizeof(float)));
    HIP_TRY(hipMalloc(&h->m, cap * h->latent * sizeof(float)));
    HIP_TRY(hipMalloc(&h->part, cap * h->nsplit * h->latent * sizeof(float)));
    {
        const size_t n_count = (size_t)(cap / 32 + 16);          // blocks + up to 8 row groups + slack
        HIP_TRY(hipMalloc(&h->upd_count, n_count * sizeof(unsigned)));
        HIP_TRY(hipMemset(h->upd_count, 0, n_count * sizeof(unsigned)));
    }
    HIP_TRY(hipMalloc(&h->turn_bar, turn_bar_bytes()));
    HIP_TRY(hipMemset(h->turn_bar, 0, turn_bar_bytes()));
    HIP_TRY(hipMalloc(&h->loss, cap * sizeof(float)));
    HIP_TRY(hipMalloc(&h->y, cap * h->P * sizeof(float)));
    if (h->graph_max_rows > 0) {
        h->xbuf_floats = std::min<int64_t>(cap, h->graph_max_rows) * h->P;      // B <= B * R rows
        HIP_TRY(hipMalloc(&h->xbuf, (size_t)h->xbuf_floats * sizeof(float)));
    }
    if (h->arch == DG_ARCH_CELEBA64) {
        HIP_TRY(hipMalloc(&h->g6, cap * h->P * sizeof(float)));
        HIP_TRY(hipMalloc(&h->loss_part, cap * 64 * sizeof(float)));
    }
    const int nd = (int)h->dec.size();
    h->act.assign(nd, nullptr);
    h->actf.assign(nd, nullptr);
    h->gate.assign(nd, nullptr);
    h->act_row.assign(nd, 0);
    size_t part_doubles = 0;
    // (rows padded to whole 32-row blocks: the fragment-order kernels compute and store whole blocks)
    const int64_t cap32 = (cap + 31) / 32 * 32;
    for (int d = 0; d < nd; ++d) {
        ActInfo& a = h->ai[d];
        h->act_row[d] = a.row_floats;
        HIP_TRY(hipMalloc(&h->act[d], cap32 * a.row_floats * sizeof(float)));
        if (h->frag_path && !h->use_bn && d + 1 < nd) {
            HIP_TRY(hipMalloc(&h->actf[d], cap32 * a.row_floats * sizeof(float)));
            HIP_TRY(hipMalloc(&h->gate[d], cap32 * (a.row_floats / 32) * sizeof(unsigned)));
        }
        a.buf = h->act[d];
        if (a.has_bn) {
            HIP_TRY(hipMalloc(&a.xhat, cap * a.row_floats * sizeof(float)));
            if (h->bn_fused) {
                // 32-row statistics blocks of the producing GEMM at `cap` rows (dg_plan.h stat_blocks: ceil(rows * positions / 32)
                // per tap class); bn_forward refuses a launch that would need more
                const GemmOp& producer = d == 0 ? h->F1 : h->Fd[(size_t)d - 1];
                size_t blocks = (size_t)dg::stat_blocks(producer.bplan, (int)std::min<int64_t>(cap, 1 << 24));
                // (the backward GEMM that writes this activation's gradient leaves ITS block sums in the same buffer: the forward
                // sums are consumed right behind the forward GEMM)
                if (d + 1 < nd && h->Bd[(size_t)d].mode == dg::EPI_MASK_STATS)
                    blocks = std::max(blocks, (size_t)dg::stat_blocks(h->Bd[(size_t)d].bplan, (int)std::min<int64_t>(cap, 1 << 24)));
                HIP_TRY(hipMalloc(&a.block_sums, blocks * 2 * (size_t)a.bn_C * sizeof(float)));
                a.block_cap = (int64_t)blocks;
            }
            const size_t need = (size_t)dg::bn_max_blocks() * 2 * a.bn_C;
            if (need > part_doubles) part_doubles = need;
        }
    }
    if (part_doubles) HIP_TRY(hipMalloc(&h->bn_part, part_doubles * sizeof(double)));
    // Batchnorm form of the MNIST tail: one record [2][C] per workgroup (dg_tail_mnist.hip mnist_tail_pipe3_kernel<C, true>)
    if (h->arch == DG_ARCH_MNIST28 && h->ai[(size_t)nd - 1].has_bn && h->bn_fused >= 2 && h->tail_pipe > 0) {
        HIP_TRY(hipMalloc(&h->tail_bn_sums, (size_t)h->tail_pipe * 2 * (size_t)h->ai[(size_t)nd - 1].bn_C * sizeof(float)));
        h->tail_bn_sums_wgs = h->tail_pipe;
    }
    h->F1.stats = h->ai[0].block_sums; h->F1.stats_cap = h->ai[0].block_cap;
    for (int d = 0; d + 1 < nd; ++d) {
        h->Fd[(size_t)d].stats = h->ai[d + 1].block_sums; h->Fd[(size_t)d].stats_cap = h->ai[d + 1].block_cap;
        h->Bd[(size_t)d].stats = h->ai[d].block_sums; h->Bd[(size_t)d].stats_cap = h->ai[d].block_cap;
    }
    h->cap_rows = cap;
    return DG_OK;
}

// F1 / B1 on the weight-stationary kernels (dg_linear.hip)?
bool lin_stationary(const dg_handle* h, const GemmOp& op) {
    if (!h->latent_turn) return false;
    if (&op == &h->F1) return h->lin_pack_fwd != nullptr && dg::lin_stationary_supported(h->latent / 32, op.mode);
    if (&op == &h->B1) return h->lin_pack_bwd != nullptr && dg::lin_stationary_supported(h->lin_out / h->nsplit / 32, op.mode);
    return false;
}

// Does this handle run the fragment-order forward path (dg_fgemm.hip)?  Batchnorm couples rows through statistics passes that
// read NHWC pre-activations; concurrent row groups start at rows that are not multiples of 32; everything else is a matter of
// the layer shapes (the two generators at their default widths qualify).
bool frag_active(const dg_handle* h) {
    if (!h->frag_path || h->use_bn || h->two_streams > 1) return false;
    return lin_stationary(h, h->F1) && h->F1.mode == dg::EPI_BIAS_RELU;
}
bool frag_shapes_ok(const dg_handle* h) {
    const size_t nd = h->dec.size();
    for (size_t d = 0; d + 1 < nd; ++d) {
        if (!h->Fp[d] || !dg::frag_supported(h->Fd[d].bplan)) return false;
        if (h->Fd[d].mode != dg::EPI_BIAS_RELU && h->Fd[d].mode != dg::EPI_BIAS) return false;
        if (h->Fd[d].mode == dg::EPI_BIAS && d + 2 < nd) return false;    // (a layer without ReLU may only feed the tail)
        if (h->ai[d].row_floats % 32) return false;
    }
    return h->lin_out % 128 == 0;
}
bool frag_on(const dg_handle* h) { return frag_active(h) && frag_shapes_ok(h) && h->actf.size() == h->dec.size() && h->actf[0] != nullptr; }

// One forward deconv on dg_fgemm.hip: A in fragment order; Out in fragment order (+ gate bits) or NHWC (the layer the tail reads)
int run_frag(dg_handle* h, GemmOp& op, int d, const float* A, float* Out, bool out_frag, unsigned* gates, int n_rows, hipStream_t s, bool prof) {
    const FragList* fl = find_frag_jobs(op, n_rows);
    if (!fl) return fail(DG_E_STATE, "layer %s has no fragment-order job list for %d rows (prepare_rows was skipped)", op.name.c_str(), n_rows);
    dg::FragArgs a;
    a.A = A;
    a.Wp = h->Fp[(size_t)d];
    a.Out = Out;
    a.bias = op.bias;
    a.gate_bits = gates;
    a.jobs = fl->d_jobs;
    a.taps = op.d_btaps;
    a.a_rowstride = op.bplan.a_rowstride;
    a.out_rowstride = op.bplan.out_rowstride;
    a.gate_words = (int)(op.bplan.out_rowstride / 32);
    a.kch = op.bplan.kch;
    a.kc8_log2 = op.bplan.kch == 64 ? 3 : (op.bplan.kch == 128 ? 4 : 5);
    a.mode = op.mode;
    a.out_frag = out_frag ? 1 : 0;
    a.n_jobs = fl->n_jobs;
    a.wave_begin = fl->d_begin;
    a.n_wgs = fl->n_wgs;
    char sym[64];
    snprintf(sym, sizeof sym, fl->d_begin ? "@fgemm_persist_kernel<2, 4, %d, %s>" : "@fgemm_kernel<2, 4, %d, %s>", op.mode, out_frag ? "true" : "false");
    {
        ProfScope ps(h, s, prof, op.name + sym, 2.0 * (double)op.bplan.macs_per_row * n_rows);
        dg::launch_fgemm(a, s);
    }
    return launch_check(op.name.c_str());
}

// Fragment-order copies of the Linear weights for dg_linear.hip (forward: 128-feature column tiles of W^T; backward: the
// nsplit K slices of W, all 128 latent columns each), built from the host copy kept by dg_set_weights.
int build_lin_packs(dg_handle* h) {
    auto fr = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    HIP_TRY(hipDeviceSynchronize());
    fr(h->lin_pack_fwd); fr(h->lin_pack_bwd);
    ++h->list_epoch;
    if (h->lin_w_host.empty()) return DG_OK;
    const int K = h->latent, F = h->lin_out;
    const float* W = h->lin_w_host.data();               // [K][F]
    std::vector<float> pk((size_t)K * F);
    if (F % 128 == 0 && K % 32 == 0 && dg::lin_stationary_supported(K / 32, dg::EPI_BIAS_RELU)) {
        const int kch = K / 32;
        for (int u = 0; u < F / 128; ++u)
            for (int w = 0; w < 4; ++w)
                for (int c = 0; c < kch; ++c)
                    for (int kk = 0; kk < 4; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int f = u * 128 + w * 32 + (lane & 31), k = c * 32 + (kk * 2 + (lane >> 5)) * 4 + e;
                                pk[(size_t)dg::lin_pack_index(u, w, kch, c, kk, lane, e)] = W[(size_t)k * F + f];      // W^T[f][k]
                            }
        HIP_TRY(hipMalloc(&h->lin_pack_fwd, pk.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->lin_pack_fwd, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (K == 128 && F % (h->nsplit * 32) == 0 && dg::lin_stationary_supported(F / h->nsplit / 32, dg::EPI_STORE)) {
        const int ks = F / h->nsplit, kch = ks / 32;
        for (int s = 0; s < h->nsplit; ++s)
            for (int w = 0; w < 4; ++w)
                for (int c = 0; c < kch; ++c)
                    for (int kk = 0; kk < 4; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int d = w * 32 + (lane & 31), f = s * ks + c * 32 + (kk * 2 + (lane >> 5)) * 4 + e;
                                pk[(size_t)dg::lin_pack_index(s, w, kch, c, kk, lane, e)] = W[(size_t)d * F + f];
                            }
        HIP_TRY(hipMalloc(&h->lin_pack_bwd, pk.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->lin_pack_bwd, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return DG_OK;
}

// The momentum update riding in the Linear backward launch: the rows' z / m, their arrival counters and the step's constants.
struct UpdateFold {
    float* z; float* m; unsigned* count; float lr, momentum;
};

int run_lin_stationary(dg_handle* h, GemmOp& op, const float* A, float* Out, int n_rows, hipStream_t s, bool prof,
                       const UpdateFold* uf = nullptr, float* out_frag = nullptr, unsigned* gates = nullptr) {
    const bool fwd = &op == &h->F1;
    dg::LinArgs a;
    a.out_frag = fwd ? out_frag : nullptr;
    a.gate_bits = fwd ? gates : nullptr;
    a.gate_words = h->lin_out / 32;
    a.upd_z = nullptr; a.upd_m = nullptr; a.upd_count = nullptr; a.upd_lr = 0.f; a.upd_momentum = 0.f;
    if (uf && !fwd) { a.upd_z = uf->z; a.upd_m = uf->m; a.upd_count = uf->count; a.upd_lr = uf->lr; a.upd_momentum = uf->momentum; }
    a.A = A;
    a.Wp = fwd ? h->lin_pack_fwd : h->lin_pack_bwd;
    a.Out = Out;
    a.bias = op.bias;
    a.n_rows = n_rows;
    a.mode = op.mode;
    const int n_blocks = (n_rows + 31) / 32;
    int want;
    if (fwd) {
        a.a_rowstride = h->latent; a.a_unit = 0;
        a.out_rowstride = h->lin_out; a.out_unit = 128;
        a.units = h->lin_out / 128;
        a.kch = h->latent / 32;
        want = h->lin_groups_fwd > 0 ? h->lin_groups_fwd : std::max(1, 2 * h->cu_count / a.units);     // two workgroups per CU
    } else {
        const int ks = h->lin_out / h->nsplit;
        a.a_rowstride = h->lin_out; a.a_unit = ks;
        a.out_rowstride = (long long)h->nsplit * h->latent; a.out_unit = h->latent;
        a.units = h->nsplit;
        a.kch = ks / 32;
        want = h->lin_groups_bwd > 0 ? h->lin_groups_bwd : std::max(1, h->cu_count / a.units);         // one workgroup per CU
    }
    a.groups = std::min(n_blocks, want);
#ifdef DG_MEASURE
    a.trace = (h->d_job_trace && op.name == h->job_trace_op && a.units * a.groups * 2 <= kJobTraceCap) ? h->d_job_trace : nullptr;
#endif
    char sym[64];
    snprintf(sym, sizeof sym, a.upd_z ? "@lin_stationary_kernel<%d, %d, true>" : a.out_frag ? "@lin_stationary_kernel<%d, %d, false, true>" : "@lin_stationary_kernel<%d, %d>", a.kch, a.mode);
    {
        ProfScope ps(h, s, prof, op.name + sym, 2.0 * (double)op.bplan.macs_per_row * n_rows);
        dg::launch_lin_stationary(a, s);
    }
    return launch_check(op.name.c_str());
}

// One GEMM layer: one launch with the job list prepare_rows() left for this row count.  Nothing here allocates or waits.
int run_gemm(dg_handle* h, GemmOp& op, const float* A, float* Out, int n_rows, hipStream_t s, bool prof, const unsigned* gates = nullptr) {
    if (lin_stationary(h, op)) return run_lin_stationary(h, op, A, Out, n_rows, s, prof);
    const JobList* jl = find_jobs(op, n_rows);
    if (!jl) return fail(DG_E_STATE, "layer %s has no job list for %d rows (prepare_rows was skipped)", op.name.c_str(), n_rows);
    if ((op.mode == dg::EPI_BIAS_STATS || op.mode == dg::EPI_MASK_STATS) && (!op.stats || dg::stat_blocks(op.bplan, n_rows) > op.stats_cap))
        return fail(DG_E_STATE, "layer %s: %lld statistics blocks at %d rows, the buffer holds %lld", op.name.c_str(),
                    (long long)dg::stat_blocks(op.bplan, n_rows), n_rows, (long long)op.stats_cap);
    int group = 0;                                   // the row group launching: its own copy of the list's pair scratch
    for (int i = 0; i < dg_handle::kMaxGroups - 1; ++i)
        if (s == h->side_stream[i] && s != nullptr) group = i + 1;
    dg::GemmArgs a = gemm_args(h, op, *jl, A, Out, group);
    if (gates) {                                      // fragment-order path: ReluGrad from the gate bits, Out holds gradients only
        a.mode = dg::EPI_MASK_BITS;
        a.gate_bits = gates;
        a.gate_words = (int)(op.bplan.out_rowstride / 32);
    }
    char sym[64];
    snprintf(sym, sizeof sym, "@gemm_batched_kernel<%d, %d, %d, %s>", op.family, a.mode, jl->min_level, a.pair_scratch ? "true" : "false");
    {
        ProfScope ps(h, s, prof, op.name + sym, 2.0 * (double)op.bplan.macs_per_row * n_rows);
        dg::launch_gemm(op.family, a, s);
    }
    return launch_check(op.name.c_str());
}

// A row group = a contiguous range of latent rows (whole images) processed on one stream.  Rows are independent
// (use_bn = False), so a batch may be split into groups that run concurrently on two streams: one group's
// VALU/LDS-bound tail and short kernels then overlap the other group's MFMA-bound GEMMs.
struct RowGroup {
    int row0 = 0, n_rows = 0;
    hipStream_t s = nullptr;
};

// Row groups of a call over B images x R restarts (whole images per group; the streams are assigned by the caller).
int split_groups(const dg_handle* h, int B, int R, RowGroup* grp, int force = -1);

// may a call of B x R run as several row groups at all?
bool groups_eligible(const dg_handle* h, int B, int R) {
    return h->two_streams > 1 && !h->use_bn && B * R >= h->two_stream_min_rows && B > 1;
}

// force: -1 = what the call runs as (with two_streams_auto: the timed choice of this shape, ONE group while there is none),
// 1 = one group, > 1 = the concurrent form
int split_groups(const dg_handle* h, int B, int R, RowGroup* grp, int force) {
    const int n_rows = B * R;
    int ngroups = 1;
    grp[0].row0 = 0; grp[0].n_rows = n_rows;
    bool several = groups_eligible(h, B, R);
    if (several && force == 1) several = false;
    if (several && force < 0 && h->two_streams_auto) {
        const auto it = h->group_choice.find(std::make_pair(B, R));
        several = it != h->group_choice.end() && it->second > 1;
    }
    if (several) {
        ngroups = h->two_streams < B ? h->two_streams : B;
        if (ngroups > dg_handle::kMaxGroups) ngroups = dg_handle::kMaxGroups;
        int b_done = 0;
        for (int gi = 0; gi < ngroups; ++gi) {
            int nb = (B - b_done + (ngroups - gi) - 1) / (ngroups - gi);           // images of this group
            if (ngroups == 2 && gi == 0 && h->two_stream_split > 0 && h->two_stream_split < 100)
                nb = std::min(B - 1, std::max(1, (B * h->two_stream_split + 50) / 100));
            grp[gi].row0 = b_done * R;
            grp[gi].n_rows = nb * R;
            b_done += nb;
        }
    }
    return ngroups;
}

// Everything a call needs before its first kernel: workspace for `cap_rows` latent rows and, for each row count in
// `rows[0..n)`, the job list of every GEMM layer (built and, with jobs.tune, timed on the layer's own buffers).  This is the
// only place on the compute path that allocates device memory or waits for the device; once it has run for a row count,
// calls with that row count only enqueue kernels.  A stream that is being captured cannot be prepared on.
int prepare_rows(dg_handle* h, int64_t cap_rows, const int* rows, int n, hipStream_t s) {
    bool missing = cap_rows > h->cap_rows;
    const int nd = (int)h->dec.size();
    auto each_op = [&](auto&& fn) -> int {
        int rc = fn(h->F1, 0, -1, 0);                 // (op, kind, deconv index)
        for (int d = 0; !rc && d + 1 < nd; ++d) rc = fn(h->Fd[d], 1, d, 0);
        for (int d = nd - 2; !rc && d >= 0; --d) rc = fn(h->Bd[d], 2, d, 0);
        if (!rc) rc = fn(h->B1, 3, -1, 0);
        return rc;
    };
    for (int i = 0; i < n && !missing; ++i)
        each_op([&](GemmOp& op, int kind, int, int) {
            if (lin_stationary(h, op)) return 0;
            if (kind == 1 && frag_on(h) ? !find_frag_jobs(op, rows[i]) : !find_jobs(op, rows[i])) missing = true;
            return 0;
        });
    if (!missing) return DG_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    // (an error from the query -- the legacy NULL stream while another stream captures in global mode -- is treated as "capturing":
    // what follows synchronises the device and allocates)
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return fail(DG_E_STATE, "this call shape has not been prepared and the stream is being captured: call dg_prepare(B, R) "
                                "before the capture (it allocates workspace and times the job lists)");
    }
    // One handle serves ONE stream at a time (include/defensegan_hip.h): the timing launches below write the handle's own
    // activation buffers, so whatever an earlier call left queued on another stream must have finished first
    HIP_TRY(hipDeviceSynchronize());
    int rc = ensure_workspace(h, cap_rows);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const int nr = rows[i];
        rc = each_op([&](GemmOp& op, int kind, int d, int) {
            const float* A = kind == 0 ? h->z : kind == 1 ? h->act[d] : kind == 2 ? h->act[d + 1] : h->act[0];
            float* Out = kind == 0 ? h->act[0] : kind == 1 ? h->act[d + 1] : kind == 2 ? h->act[d] : h->part;
            if (lin_stationary(h, op)) return (int)DG_OK;          // no job list: dg_linear.hip derives its grid from the row count
            if (kind == 1 && frag_on(h)) {                         // forward deconv on dg_fgemm.hip: its own kind of list, not timed
                if (!get_frag_jobs(op, nr, h->frag_path >= 2 ? 2 * h->cu_count : 0)) return fail(DG_E_NOMEM, "cannot build the fragment-order job list of layer %s for %d rows", op.name.c_str(), nr);
                return (int)DG_OK;
            }
            h->tune_gates = (kind == 2 && frag_on(h)) ? h->gate[(size_t)d] : nullptr;
            const JobList* got = get_jobs(h, op, nr, A, Out, s);
            h->tune_gates = nullptr;
            if (!got) return fail(DG_E_NOMEM, "cannot build the job list of layer %s for %d rows", op.name.c_str(), nr);
            return (int)DG_OK;
        });
        if (rc) return rc;
    }
    return DG_OK;
}

int time_group_forms(dg_handle* h, int B, int R, hipStream_t s, int* best);

// prepare_rows for a projection call of B images x R restarts: its row groups' sizes.  With two_streams_auto a shape that may run
// as several groups is prepared in both forms once and the faster form is kept (timed: a few loop steps of each, alternating)
int prepare_call(dg_handle* h, int B, int R, hipStream_t s) {
    RowGroup grp[dg_handle::kMaxGroups];
    if (h->two_streams_auto && groups_eligible(h, B, R) && !h->group_choice.count(std::make_pair(B, R))) {
        int rows[dg_handle::kMaxGroups];
        for (int form : {1, h->two_streams}) {
            const int ngf = split_groups(h, B, R, grp, form);
            for (int gi = 0; gi < ngf; ++gi) rows[gi] = grp[gi].n_rows;
            const int rc = prepare_rows(h, (int64_t)B * R, rows, ngf, s);
            if (rc) return rc;
        }
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return fail(DG_E_STATE, "this call shape has not been prepared and the stream is being captured: call dg_prepare(B, R) "
                                    "before the capture (it times the one-group and the %d-group form of the call)", h->two_streams);
        }
        int best = 1;
        const int rc = time_group_forms(h, B, R, s, &best);
        if (rc) return rc;
        h->group_choice[std::make_pair(B, R)] = best;
    }
    const int ng = split_groups(h, B, R, grp);
    int rows[dg_handle::kMaxGroups];
    for (int gi = 0; gi < ng; ++gi) rows[gi] = grp[gi].n_rows;
    return prepare_rows(h, (int64_t)B * R, rows, ng, s);
}

dg::BnArgs bn_args(dg_handle* h, const ActInfo& a, int n_rows) {
    dg::BnArgs b;
    b.a = a.buf;
    b.xhat = a.xhat;
    b.part = h->bn_part;
    b.fstats = a.fstats;
    b.bstats = a.bstats;
    b.scale = a.scale;
    b.offset = a.offset;
    b.rows = (int64_t)n_rows * a.bn_rows;
    b.C = a.bn_C;
    return b;
}

// forward chain at the current h->z; fills activations, loss, (y when want_y); when tail_backward the tail also
// leaves the gradient w.r.t. the last GEMM activation in place.  x points at image 0 of the CALL (row0 / R
// images are skipped inside).  With use_bn the whole call is one row group (batch statistics couple all rows).
// want_loss: the per-row loss is only read after the last step (selection) and by dg_loss_grad; the CelebA tail leaves
// per-band partial sums, whose reduction is skipped when nobody reads the result.
int run_forward(dg_handle* h, const float* x, const RowGroup& g, int R, bool want_y, bool want_loss, bool tail_backward, bool prof,
                bool have_f1 = false) {
    const int n_rows = g.n_rows;
    hipStream_t s = g.s;
    const int64_t r0 = g.row0;
    // a layer with Batchnorm behind it leaves its pre-activations in the layer's own buffer (ActInfo::xhat): the BN pass writes
    // the activation
    auto out_of = [&](int d) { return (h->ai[d].has_bn ? h->ai[d].xhat : h->act[d]) + r0 * h->act_row[d]; };
    const bool frag = frag_on(h) && r0 == 0;
    // (have_f1: the previous step's fused latent turn has already left this step's Linear forward in act[0])
    int rc = have_f1 ? DG_OK
             : frag  ? run_lin_stationary(h, h->F1, h->z, h->act[0], n_rows, s, prof, nullptr, h->actf[0], h->gate[0])
                     : run_gemm(h, h->F1, h->z + r0 * h->latent, out_of(0), n_rows, s, prof);
    if (rc) return rc;
    const int nd = (int)h->dec.size();
    // The MNIST tail in its Batchnorm form (this launch runs mnist_tail_pipe3_kernel and the last GEMM is behind a Batchnorm whose sums
    // come from epilogues): the tail applies relu(bn(.)) to the pre-activations itself and leaves that layer's backward sums
    dg::MnistTailArgs t;
    bool tail_bn = false;
    if (h->arch == DG_ARCH_MNIST28) {
        t.n_rows = n_rows;
        t.C = h->dec[(size_t)nd - 1].cin;
        t.do_backward = tail_backward ? 1 : 0;
        t.pipe = h->tail_pipe;
        // the third-generation kernel writes neither the per-row loss nor y: a launch whose loss or image is read (dg_loss_grad
        // with out_loss and / or out_y) runs the first
        t.pipe_version = (h->tail_pipe_version == 3 && (want_loss || want_y)) ? 1 : h->tail_pipe_version;
        const ActInfo& la = h->ai[(size_t)nd - 1];
        tail_bn = la.has_bn && h->bn_fused >= 2 && h->Fd[(size_t)nd - 2].mode == dg::EPI_BIAS_STATS && dg::mnist_tail_runs_pipe3(t) &&
                  h->tail_bn_sums && h->tail_bn_sums_wgs == t.pipe && r0 == 0;
    }
    h->tail_left_bn_sums = tail_bn;
    auto bn_forward = [&](int d, const GemmOp& producer) {
        ProfScope ps(h, s, prof, "BNf", 0.0);
        if (producer.mode == dg::EPI_BIAS_STATS)
            dg::launch_bn_forward_from_blocks(bn_args(h, h->ai[d], n_rows), h->ai[d].block_sums, (int)dg::stat_blocks(producer.bplan, n_rows),
                                              (tail_bn && d == nd - 1) ? -1 : 1, s, producer.bias);
        else
            dg::launch_bn_forward(bn_args(h, h->ai[d], n_rows), 1, s);
    };
    if (h->ai[0].has_bn) bn_forward(0, h->F1);
    for (int d = 0; d + 1 < nd; ++d) {
        if (frag) {
            // the last of these layers feeds the tail, which reads NHWC; the others feed the next fragment-order layer
            const bool to_tail = d + 2 == nd;
            rc = run_frag(h, h->Fd[d], d, h->actf[d], to_tail ? h->act[d + 1] : h->actf[d + 1], !to_tail, to_tail ? nullptr : h->gate[d + 1],
                          n_rows, s, prof);
            if (rc) return rc;
            continue;
        }
        rc = run_gemm(h, h->Fd[d], h->act[d] + r0 * h->act_row[d], out_of(d + 1), n_rows, s, prof);
        if (rc) return rc;
        if (h->ai[d + 1].has_bn) bn_forward(d + 1, h->Fd[d]);
    }
    const DeconvSpec& last = h->dec[nd - 1];
    if (h->arch == DG_ARCH_MNIST28) {
        t.h3 = h->act[nd - 1] + r0 * h->act_row[nd - 1];
        t.F5 = h->F[nd - 1];
        t.b5 = h->bias[nd - 1];
        t.x = x + (r0 / R) * h->P;
        t.loss = h->loss + r0;
        t.y = want_y ? h->y + r0 * h->P : nullptr;
        t.R = R;
        t.want_loss = want_loss ? 1 : 0;
        t.bn_pre = t.bn_fstats = t.bn_scale = t.bn_offset = nullptr;
        t.bn_sums = nullptr;
        if (tail_bn) {
            const ActInfo& la = h->ai[(size_t)nd - 1];
            t.bn_pre = la.xhat; t.bn_fstats = la.fstats; t.bn_scale = la.scale; t.bn_offset = la.offset;
            t.bn_sums = h->tail_bn_sums;
        }
#ifdef DG_MEASURE
        t.dbg = h->tail_dbg;
        t.trace = h->d_tail_trace;
#endif
        const double macs = 67.0 * 67.0 * last.cin;   // valid taps 14 -> 28
        const bool piped = t.pipe > 0 && tail_backward && t.C == 64 && n_rows >= 2 * t.pipe;   // launch_mnist_tail_mfma
        ProfScope ps(h, s, prof, !tail_backward ? "T5f@mnist_tail_mfma_kernel" : piped ? (t.pipe_version == 3 ? "T5fb@mnist_tail_pipe3_kernel" : t.pipe_version == 2 ? "T5fb@mnist_tail_pipe2_kernel" : "T5fb@mnist_tail_pipe_kernel") : "T5fb@mnist_tail_mfma_kernel",
                     (tail_backward ? 4.0 : 2.0) * macs * n_rows);
        dg::launch_mnist_tail_mfma(t, s);
    }
    if (h->arch == DG_ARCH_MNIST28) {
        const int rc2 = launch_check("the MNIST tail (Generator.5 + loss)");
        if (rc2) return rc2;
    } else {
        dg::CelebaTailArgs t;
        t.h5 = h->act[nd - 1] + r0 * h->act_row[nd - 1];
        t.F6 = h->F[nd - 1];
        t.F6p = h->tail_pack16;
        t.bwd_persist = h->tail_bwd_persist;
        t.fwd_split = (last.cin == 64) ? h->tail_fwd_split : 0;
        t.want_loss = want_loss ? 1 : 0;
#ifdef DG_MEASURE
        t.F6p = h->tail_fwd16 ? h->tail_pack16 : h->tail_pack;
        t.fwd16 = h->tail_fwd16;
        if (!h->tail_fwd16) t.fwd_split = 0;          // the 32-wide cross-check kernel was asked for
        t.trace = h->d_tail_trace;
        t.dbg = h->tail_dbg;
        t.bwd_bands = h->tail_bwd_bands;
        t.prio = h->tail_prio;
#endif
        t.b6 = h->bias[nd - 1];
        t.x = x + (r0 / R) * h->P;
        t.loss_part = h->loss_part + r0 * 64;
        t.y = want_y ? h->y + r0 * h->P : nullptr;
        t.g6 = h->g6 + r0 * h->P;
        t.n_rows = n_rows;
        t.R = R;
        t.C = last.cin;
        t.do_backward = tail_backward ? 1 : 0;
        const double macs = 157.0 * 157.0 * last.cin * 3.0;   // valid taps 32 -> 64
        {
#ifdef DG_MEASURE
            ProfScope ps(h, s, prof, h->tail_fwd16 ? "T6f@celeba_tail_fwd16_kernel" : "T6f@celeba_tail_fwd_mfma_kernel", 2.0 * macs * n_rows);
#else
            ProfScope ps(h, s, prof, t.fwd_split > 0 ? "T6f@celeba_tail_fwd_split_kernel" : "T6f@celeba_tail_fwd16_kernel", 2.0 * macs * n_rows);
#endif
            dg::launch_celeba_tail_fwd_mfma(t, s);
        }
        if (want_loss) dg::launch_celeba_loss_finish(t.loss_part, h->loss + r0, n_rows, t.fwd_split > 0 ? 16 : 8, h->P, s);
        if (tail_backward) {
#ifdef DG_MEASURE
            ProfScope ps(h, s, prof, h->tail_bwd_persist > 0 ? "T6b@celeba_tail_bwd_persist_kernel" : "T6b@celeba_tail_bwd_mfma_kernel", 2.0 * macs * n_rows);
#else
            ProfScope ps(h, s, prof, "T6b@celeba_tail_bwd_persist_kernel", 2.0 * macs * n_rows);
#endif
            dg::launch_celeba_tail_bwd_mfma(t, s);
        }
        const int rc2 = launch_check("the CelebA tail (Generator.6 + loss)");
        if (rc2) return rc2;
    }
    return DG_OK;
}

bool update_folds(const dg_handle* h) {
    return h->update_fold && h->upd_count && lin_stationary(h, h->B1) && dg::lin_fold_supported(h->nsplit, h->latent);
}

int run_backward(dg_handle* h, const RowGroup& g, bool prof, const UpdateFold* uf = nullptr, bool without_b1 = false) {
    const int nd = (int)h->dec.size();
    const int64_t r0 = g.row0;
    // Batchnorm backward of activation k: the sums come from the epilogue of the GEMM that wrote dy (EPI_MASK_STATS) or from a pass
    auto bn_backward = [&](int k) {
        ProfScope ps(h, g.s, prof, "BNb", 0.0);
        const GemmOp* producer = k + 1 < nd ? &h->Bd[(size_t)k] : nullptr;
        if (!producer && h->tail_left_bn_sums)         // the tail wrote dy and its sums (run_forward of this step)
            dg::launch_bn_backward_from_blocks(bn_args(h, h->ai[k], g.n_rows), h->tail_bn_sums, h->tail_bn_sums_wgs, g.s);
        else if (producer && producer->mode == dg::EPI_MASK_STATS)
            dg::launch_bn_backward_from_blocks(bn_args(h, h->ai[k], g.n_rows), h->ai[k].block_sums, (int)dg::stat_blocks(producer->bplan, g.n_rows), g.s);
        else
            dg::launch_bn_backward(bn_args(h, h->ai[k], g.n_rows), g.s);
    };
    for (int d = nd - 2; d >= 0; --d) {
        if (h->ai[d + 1].has_bn) bn_backward(d + 1);
        int rc = run_gemm(h, h->Bd[d], h->act[d + 1] + r0 * h->act_row[d + 1], h->act[d] + r0 * h->act_row[d], g.n_rows, g.s, prof,
                          frag_on(h) && r0 == 0 ? h->gate[d] : nullptr);
        if (rc) return rc;
    }
    if (h->ai[0].has_bn) bn_backward(0);
    if (without_b1) return DG_OK;                    // the fused latent turn follows (run_latent_turn)
    if (uf) return run_lin_stationary(h, h->B1, h->act[0] + r0 * h->act_row[0], h->part + r0 * h->nsplit * h->latent, g.n_rows, g.s, prof, uf);
    return run_gemm(h, h->B1, h->act[0] + r0 * h->act_row[0], h->part + r0 * h->nsplit * h->latent, g.n_rows, g.s, prof);
}

// The latent turn as one launch (dg_turn.hip)?  Both Linear layers on their weight-stationary shapes, the Linear output feeding
// the first deconv directly (no Batchnorm pass, no fragment-order copy), the update not already folded into the backward.
bool turn_fuses(const dg_handle* h) {
    return h->turn_fused && h->turn_bar && lin_stationary(h, h->B1) && lin_stationary(h, h->F1) && h->F1.mode == dg::EPI_BIAS_RELU &&
           dg::turn_fused_supported(h->nsplit, h->latent, h->lin_out) && !h->ai[0].has_bn && !frag_on(h) && !update_folds(h) &&
           2 * h->cu_count / h->nsplit >= 1;
}

// Row group `gi` of `ngroups` concurrent ones: da1 in act[0] -> partials -> z, m -> act[0] = next step's relu(z W^T + b).
int run_latent_turn(dg_handle* h, const RowGroup& g, int gi, int ngroups, float lr, float momentum, bool prof) {
    const int64_t r0 = g.row0;
    dg::TurnArgs a;
    a.dA = h->act[0] + r0 * h->act_row[0];
    a.Wb = h->lin_pack_bwd;
    a.part = h->part + r0 * h->nsplit * h->latent;
    a.z = h->z + r0 * h->latent;
    a.m = h->m + r0 * h->latent;
    a.Wf = h->lin_pack_fwd;
    a.bias = h->F1.bias;
    a.H = h->act[0] + r0 * h->act_row[0];
    a.bar = h->turn_bar + (size_t)gi * dg_handle::kTurnBarWords;
    a.err = h->turn_bar + (size_t)dg_handle::kMaxGroups * dg_handle::kTurnBarWords;
    a.lr = lr; a.momentum = momentum;
    a.features = h->lin_out;
    a.n_rows = g.n_rows;
    a.nsplit = h->nsplit;
    // every workgroup of every turn launch that may be in flight at once must be resident: two fit a CU
    const int n_blocks = (g.n_rows + 31) / 32;
    int want = h->lin_groups_bwd > 0 ? h->lin_groups_bwd : std::max(1, h->cu_count / h->nsplit);
    want = std::min(want, std::max(1, 2 * h->cu_count / std::max(1, ngroups) / h->nsplit));
    want = std::min(want, dg_handle::kTurnBarWords / 2);
    a.groups = std::min(n_blocks, want);
#ifdef DG_MEASURE
    a.trace = (h->d_job_trace && h->job_trace_op == "TURN" && a.nsplit * a.groups * 2 <= kJobTraceCap) ? h->d_job_trace : nullptr;
#endif
    {
        ProfScope ps(h, g.s, prof, "TURN@latent_turn_kernel", 2.0 * (double)(h->B1.bplan.macs_per_row + h->F1.bplan.macs_per_row) * g.n_rows);
        dg::launch_latent_turn(a, g.s);
    }
    return launch_check("the fused latent turn");
}

// (Re)builds every layer plan and re-attaches the weight pointers (dg_create, tuning options).
int rebuild_plans(dg_handle* h) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    int rc = build_plans(h);
    if (rc) return rc;
    const size_t ndec = h->dec.size();
    h->F1.W = h->lin_wt; h->F1.bias = h->lin_b;
    h->F1.stats = h->ai.empty() ? nullptr : h->ai[0].block_sums;
    h->F1.stats_cap = h->ai.empty() ? 0 : h->ai[0].block_cap;
    h->B1.W = h->lin_w;
    for (size_t d = 0; d + 1 < ndec; ++d) {
        h->Fd[d].W = h->F[d]; h->Fd[d].bias = h->bias[d];
        h->Fd[d].stats = h->ai[d + 1].block_sums;
        h->Fd[d].stats_cap = h->ai[d + 1].block_cap;
        h->Bd[d].W = h->Ft[d];
        h->Bd[d].stats = h->ai[d].block_sums;
        h->Bd[d].stats_cap = h->ai[d].block_cap;
    }
    return DG_OK;
}

// The L-step loop of DefenseGANBase.reconstruct (gan.py:409-437) as launches on the row groups' streams: L forwards, L - 1
// backward + update (the reference's L-th update is dead work), the last forward also leaves y and the per-row loss.
int enqueue_steps(dg_handle* h, const float* x, int R, int L, float lr, float momentum, const RowGroup* grp, int ngroups) {
    const int steps = L > 1 ? L : 1;
    const int decay_iter = L > 0 ? (int)std::ceil(0.8 * (double)L) : 1;
    const bool fused_turn = turn_fuses(h);
    for (int k = 0; k < steps; ++k) {
        const bool last = (k == steps - 1);
        const bool prof = h->prof_stride > 0 && (k % h->prof_stride) == 0;
        // a step that is not sampled breaks the marker chain: the next sampled launch starts from its own marker, not from the
        // one recorded after the last sampled step (which would charge it with every skipped step in between)
        if (!prof) h->prof_chain_last = -1;
        const float lr_k = h->lr_intended ? lr * std::pow(0.1f, (float)(k / decay_iter)) : lr;
        for (int gi = 0; gi < ngroups; ++gi) {
            const RowGroup& g = grp[gi];
            const bool turn = fused_turn && g.n_rows <= dg::kTurnMaxRows;
            int rc = run_forward(h, x, g, R, /*want_y=*/last, /*want_loss=*/last, /*tail_backward=*/!last, prof, /*have_f1=*/turn && k > 0);
            if (rc) return rc;
            if (last) continue;
            const int64_t r0 = g.row0;
            if (turn) {
                rc = run_backward(h, g, prof, nullptr, /*without_b1=*/true);
                if (rc) return rc;
                rc = run_latent_turn(h, g, gi, ngroups, lr_k, momentum, prof);
                if (rc) return rc;
                continue;
            }
            if (update_folds(h)) {
                // row groups are whole images, not whole 32-row blocks: group gi's counters start at r0 / 32 + gi (disjoint)
                const UpdateFold uf = {h->z + r0 * h->latent, h->m + r0 * h->latent, h->upd_count + r0 / 32 + gi, lr_k, momentum};
                rc = run_backward(h, g, prof, &uf);
                if (rc) return rc;
                continue;
            }
            rc = run_backward(h, g, prof);
            if (rc) return rc;
            ProfScope ps(h, g.s, prof, "UPD@momentum_update_kernel", 0.0);
            dg::launch_momentum_update(h->z + r0 * h->latent, h->m + r0 * h->latent, h->part + r0 * h->nsplit * h->latent,
                                       h->nsplit, g.n_rows, h->latent, lr_k, momentum, nullptr, g.s);
        }
    }
    return DG_OK;
}

// enqueue_steps on the call's row groups: several groups run on the engine's side streams, forked off the caller's stream and joined
// to it again.  (While the per-launch profile is on, the groups run one after the other on the caller's stream: the same launches
// -- row counts, job lists -- as the concurrent form, each alone on the chip, so that a launch's duration is its rate.)
int run_groups(dg_handle* h, const float* x, int R, int L, float lr, float momentum, hipStream_t s, RowGroup* grp, int ngroups) {
    const bool forked = ngroups > 1 && h->prof_stride == 0;
    grp[0].s = s;
    for (int gi = 1; gi < ngroups; ++gi) grp[gi].s = forked ? h->side_stream[gi - 1] : s;
    if (forked) {
        HIP_TRY(hipEventRecord(h->ev_fork, s));
        for (int gi = 1; gi < ngroups; ++gi) HIP_TRY(hipStreamWaitEvent(grp[gi].s, h->ev_fork, 0));
    }
    const int rc = enqueue_steps(h, x, R, L, lr, momentum, grp, ngroups);
    if (rc) return rc;
    for (int gi = 1; forked && gi < ngroups; ++gi) {
        HIP_TRY(hipEventRecord(h->ev_join[gi - 1], grp[gi].s));
        HIP_TRY(hipStreamWaitEvent(s, h->ev_join[gi - 1], 0));
    }
    return DG_OK;
}

// One group or h->two_streams groups for calls of B x R?  Nine loop steps of each form on zeroed latents and images (the launches
// do not depend on the values), three times alternating, the best time of each; the concurrent form is kept when it is at least
// 1 % faster.  Runs once per call shape, from prepare_call (blocking, like the timing of the job lists).
int time_group_forms(dg_handle* h, int B, int R, hipStream_t s, int* best) {
    const int n_rows = B * R;
    float* xt = nullptr;
    HIP_TRY(hipMalloc(&xt, (size_t)B * h->P * sizeof(float)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DG_OK;
    const int saved_prof = h->prof_stride;
    h->prof_stride = 0;
    auto done = [&](int r) { h->prof_stride = saved_prof; if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); (void)hipFree(xt); return r; };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(fail(DG_E_HIP, "hipEventCreate failed"));
    if (hipMemsetAsync(xt, 0, (size_t)B * h->P * sizeof(float), s) != hipSuccess) return done(fail(DG_E_HIP, "hipMemsetAsync failed"));
    const int forms[2] = {1, h->two_streams};
    double t[2] = {1e30, 1e30};
    const size_t zbytes = (size_t)n_rows * h->latent * sizeof(float);
    for (int rep = 0; rep < 3; ++rep)
        for (int f = 0; f < 2; ++f) {
            if (hipMemsetAsync(h->z, 0, zbytes, s) != hipSuccess || hipMemsetAsync(h->m, 0, zbytes, s) != hipSuccess) return done(fail(DG_E_HIP, "hipMemsetAsync failed"));
            if (update_folds(h) && hipMemsetAsync(h->upd_count, 0, (size_t)(n_rows / 32 + 16) * sizeof(unsigned), s) != hipSuccess) return done(fail(DG_E_HIP, "hipMemsetAsync failed"));
            if (turn_fuses(h) && hipMemsetAsync(h->turn_bar, 0, turn_bar_bytes(), s) != hipSuccess) return done(fail(DG_E_HIP, "hipMemsetAsync failed"));
            RowGroup grp[dg_handle::kMaxGroups];
            const int ng = split_groups(h, B, R, grp, forms[f]);
            for (int gi = 0; gi < ng; ++gi) {
                bool seen = false;
                for (int gj = 0; gj < gi; ++gj) seen = seen || grp[gj].n_rows == grp[gi].n_rows;
                if (!seen) { rc = clear_pair_counters(h, grp[gi].n_rows, s); if (rc) return done(rc); }
            }
            (void)hipEventRecord(e0, s);
            rc = run_groups(h, xt, R, 9, 0.f, 0.7f, s, grp, ng);
            if (rc) return done(rc);
            (void)hipEventRecord(e1, s);
            if (hipEventSynchronize(e1) != hipSuccess) return done(fail(DG_E_HIP, "the timed loop steps failed: %s", hipGetErrorString(hipGetLastError())));
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < t[f]) t[f] = ms;                   // (the first round warms both forms up)
        }
    *best = t[1] < 0.99 * t[0] ? forms[1] : 1;
    h->group_timing_ms[0] = t[0]; h->group_timing_ms[1] = t[1];
    return done(DG_OK);
}

// Graph of enqueue_steps for one call shape, reading the images from h->xbuf; nullptr = use the eager path.
hipGraphExec_t loop_graph(dg_handle* h, int B, int R, int L, float lr, float momentum) {
    for (auto it = h->graphs.begin(); it != h->graphs.end();) {
        if (it->epoch != h->list_epoch) { (void)hipGraphExecDestroy(it->exec); it = h->graphs.erase(it); continue; }
        if (it->B == B && it->R == R && it->L == L && it->lr == lr && it->momentum == momentum && it->lr_intended == h->lr_intended)
            return it->exec;
        ++it;
    }
    if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) { h->graph_broken = true; return nullptr; }
    RowGroup g;
    g.row0 = 0; g.n_rows = B * R; g.s = h->cap_stream;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (ok) {
        const int rc = enqueue_steps(h, h->xbuf, R, L, lr, momentum, &g, 1);
        ok = hipStreamEndCapture(h->cap_stream, &graph) == hipSuccess && rc == DG_OK && graph != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();
        h->graph_broken = true;
        return nullptr;
    }
    if (h->graphs.size() >= 8) { (void)hipGraphExecDestroy(h->graphs.front().exec); h->graphs.erase(h->graphs.begin()); }
    dg_handle::LoopGraph lg;
    lg.B = B; lg.R = R; lg.L = L; lg.lr = lr; lg.momentum = momentum; lg.lr_intended = h->lr_intended; lg.epoch = h->list_epoch; lg.exec = exec;
    h->graphs.push_back(lg);
    return exec;
}

void drop_graphs(dg_handle* h) {
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

// One call processes at most 2^24 latent rows (32-bit tile arithmetic in the launchers).
int check_rows(int B, int R, int* n_rows) {
    if (B < 1 || R < 1) return fail(DG_E_INVALID, "need B >= 1 and R >= 1 (got %d, %d)", B, R);
    const int64_t rows64 = (int64_t)B * R;
    if (rows64 > (1 << 24)) return fail(DG_E_INVALID, "B*R = %lld is too large for one call", (long long)rows64);
    *n_rows = (int)rows64;
    return DG_OK;
}

int check_ready(dg_handle* h) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    if (!dg_weights_complete(h)) return fail(DG_E_STATE, "generator weights are not completely set (dg_set_weights)");
    return DG_OK;
}

}  // namespace dge
#pragma GCC visibility pop

extern "C" {

int dg_version(void) { return DG_ABI_VERSION; }
const char* dg_last_error(void) { return g_err.c_str(); }

// internal: lets the other translation units of the library (dg_clf.hip) report through dg_last_error()
__attribute__((visibility("hidden"))) void dg_set_error_message(const char* msg) { g_err = msg ? msg : ""; }

int dg_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail(DG_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return DG_E_HIP; }
    return n;
}

int dg_device_info(int device, char* name, int name_len, int* cu_count, int64_t* hbm_bytes) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return DG_OK;
}

int dg_create(int arch, int latent_dim, int net_dim, int use_bn, int device, dg_handle** out) {
    if (!out) return fail(DG_E_INVALID, "out is null");
    *out = nullptr;
    if (arch != DG_ARCH_MNIST28 && arch != DG_ARCH_CELEBA64) return fail(DG_E_INVALID, "unknown arch %d", arch);
    if (latent_dim <= 0 || latent_dim % 64) return fail(DG_E_INVALID, "latent_dim must be a positive multiple of 64 (got %d)", latent_dim);
    if (net_dim <= 0 || net_dim % 64 || net_dim > 128) return fail(DG_E_INVALID, "net_dim must be 64 or 128 (got %d)", net_dim);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DG_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    dg_handle* h = new dg_handle();
    h->arch = arch;
    h->latent = latent_dim;
    h->net_dim = net_dim;
    h->use_bn = use_bn;
    h->device = device;
    h->lin_out = 4 * 4 * 4 * net_dim;
    (void)hipDeviceGetAttribute(&h->cu_count, hipDeviceAttributeMultiprocessorCount, device);
    if (h->cu_count <= 0) h->cu_count = 256;
    const int nd = net_dim;
    if (arch == DG_ARCH_MNIST28) {
        h->img_h = 28; h->img_c = 1;
        h->dec = {{"Generator.2", 4 * nd, 2 * nd, 4, 7, 0, "Generator.BN2"}, {"Generator.3", 2 * nd, nd, 7, 14, 0, "Generator.BN3"},
                  {"Generator.5", nd, 1, 14, 28, 2, ""}};
    } else {
        h->img_h = 64; h->img_c = 3;
        // Two row groups on two streams where that is faster (round 6, profiles/r06_ab_celeba_row_groups.txt: +2.0 ... +2.9 % at
        // configs[3]'s 1280 rows on three boxes -- the gather-bound tails, 12 % of a step at ~0.6 of the matrix pipe, run beside the
        // other group's GEMMs -- +1.2 % at 2560, nothing at 1600 / 1920 / 3200, -4.5 % at 5120): a timed choice per call shape
        // (prepare_call).  3 / 4 groups and unequal halves lose; MNIST loses 0.9 % with two groups and keeps one.
        h->two_streams = 2;
        h->two_streams_auto = 1;
        h->dec = {{"Generator.2", 4 * nd, 2 * nd, 4, 8, 0, "Generator.BN2"}, {"Generator.3", 2 * nd, nd, 8, 16, 0, "Generator.BN3"},
                  {"Generator.5", nd, nd, 16, 32, 1, ""}, {"Generator.6", nd, 3, 32, 64, 2, ""}};
    }
    h->P = h->img_h * h->img_h * h->img_c;
    const size_t ndec = h->dec.size();
    h->F.assign(ndec, nullptr);
    h->Fp.assign(ndec, nullptr);
    h->Ft.assign(ndec, nullptr);
    h->bias.assign(ndec, nullptr);
    hipError_t e = hipSuccess;
    auto dmalloc = [&](float** p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, n * sizeof(float)); };
    dmalloc(&h->lin_w, (size_t)h->latent * h->lin_out);
    dmalloc(&h->lin_wt, (size_t)h->latent * h->lin_out);
    dmalloc(&h->lin_b, h->lin_out);
    dmalloc(&h->xzero, h->P);
    for (size_t d = 0; d < ndec; ++d) {
        dmalloc(&h->F[d], (size_t)25 * h->dec[d].cout * h->dec[d].cin);
        dmalloc(&h->Ft[d], (size_t)25 * h->dec[d].cout * h->dec[d].cin);
        dmalloc(&h->bias[d], h->dec[d].cout);
    }
    if (e == hipSuccess) e = hipMemset(h->xzero, 0, (size_t)h->P * sizeof(float));
    if (e != hipSuccess) { dg_destroy(h); return fail(DG_E_NOMEM, "hipMalloc(weights): %s", hipGetErrorString(e)); }
    bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < dg_handle::kMaxGroups - 1; ++i)
        ok = hipStreamCreateWithFlags(&h->side_stream[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        dg_destroy(h);
        return fail(DG_E_HIP, "cannot create the side streams / events");
    }
    int rc = rebuild_plans(h);
    if (rc) { dg_destroy(h); return rc; }
    for (auto& a : h->ai) {
        if (!a.has_bn) continue;
        hipError_t e2 = hipSuccess;
        for (float** pp : {&a.scale, &a.offset}) if (e2 == hipSuccess) e2 = hipMalloc(pp, (size_t)a.bn_C * sizeof(float));
        for (float** pp : {&a.fstats, &a.bstats}) if (e2 == hipSuccess) e2 = hipMalloc(pp, (size_t)2 * a.bn_C * sizeof(float));
        if (e2 != hipSuccess) { dg_destroy(h); return fail(DG_E_NOMEM, "hipMalloc(BN parameters): %s", hipGetErrorString(e2)); }
    }
    *out = h;
    return DG_OK;
}

int dg_destroy(dg_handle* h) {
    if (!h) return DG_OK;
    (void)hipSetDevice(h->device);
    prof_collect(h);
    if (h->turn_err_host) { (void)hipHostFree(h->turn_err_host); h->turn_err_host = nullptr; }
    drop_graphs(h);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    free_workspace(h);
    auto fr = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    for (auto& a : h->ai) { fr(a.scale); fr(a.offset); fr(a.fstats); fr(a.bstats); }
    fr(h->lin_w); fr(h->lin_wt); fr(h->lin_b); fr(h->lin_pack_fwd); fr(h->lin_pack_bwd); fr(h->xzero); fr(h->tail_pack); fr(h->tail_pack16);
    if (h->d_tail_trace) (void)hipFree(h->d_tail_trace);
    if (h->d_job_trace) (void)hipFree(h->d_job_trace);
    for (int i = 0; i < dg_handle::kMaxGroups - 1; ++i) {
        if (h->side_stream[i]) { (void)hipStreamSynchronize(h->side_stream[i]); (void)hipStreamDestroy(h->side_stream[i]); }
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (auto& p : h->F) fr(p);
    for (auto& p : h->Fp) fr(p);
    for (auto& p : h->Ft) fr(p);
    for (auto& p : h->bias) fr(p);
    auto frop = [](GemmOp& op) { free_batched(op); };
    frop(h->F1); frop(h->B1);
    for (auto& o : h->Fd) frop(o);
    for (auto& o : h->Bd) frop(o);
    delete h;
    return DG_OK;
}

int dg_set_weights(dg_handle* h, const char* name, const float* data, const int64_t* shape, int ndim, int is_device) {
    if (!h || !name || !data || !shape) return fail(DG_E_INVALID, "null argument");
    const std::string nm(name);
    // Resolve the name to the shape it must have BEFORE anything is sized or copied from the caller's pointer.
    std::vector<int64_t> want;
    bool any_shape = false;          // BN scale / offset: the reference stores them with the keep_dims shape of the moments
    if (nm == "Generator.Input.W") want = {h->latent, h->lin_out};
    else if (nm == "Generator.Input.b") want = {h->lin_out};
    for (const DeconvSpec& s : h->dec) {
        if (nm == std::string(s.name) + ".Filters") want = {5, 5, s.cout, s.cin};
        else if (nm == std::string(s.name) + ".Biases") want = {s.cout};
    }
    for (const ActInfo& a : h->ai)
        if (a.has_bn && (nm == a.bn_name + ".scale" || nm == a.bn_name + ".offset")) { want = {a.bn_C}; any_shape = true; }
    if (want.empty()) return fail(DG_E_INVALID, "unknown weight name '%s'", name);
    int64_t n_want = 1;
    for (int64_t w : want) n_want *= w;
    if (ndim < 1 || ndim > 8) return fail(DG_E_INVALID, "%s: ndim %d out of range", name, ndim);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0 || shape[i] > n_want) return fail(DG_E_INVALID, "%s: dimension %d is %lld", name, i, (long long)shape[i]);
        n *= shape[i];
        if (n > n_want) break;
    }
    bool shape_ok = n == n_want;
    if (shape_ok && !any_shape) {
        shape_ok = ndim == (int)want.size();
        for (int i = 0; shape_ok && i < ndim; ++i) shape_ok = shape[i] == want[i];
    }
    if (!shape_ok) {
        std::string w;
        for (size_t i = 0; i < want.size(); ++i) w += (i ? "," : "") + std::to_string(want[i]);
        return fail(DG_E_INVALID, any_shape ? "%s: expected %s values" : "%s: expected shape [%s]", name, w.c_str());
    }
    HIP_TRY(hipSetDevice(h->device));
    std::vector<float> host((size_t)n);
    HIP_TRY(hipMemcpy(host.data(), data, (size_t)n * sizeof(float), is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
    if (nm == "Generator.Input.W") {
        std::vector<float> t((size_t)n);
        for (int d = 0; d < h->latent; ++d)
            for (int f = 0; f < h->lin_out; ++f) t[(size_t)f * h->latent + d] = host[(size_t)d * h->lin_out + f];
        HIP_TRY(hipMemcpy(h->lin_w, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->lin_wt, t.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        h->lin_w_host = host;
        const int rcp = build_lin_packs(h);
        if (rcp) return rcp;
        h->have[nm] = true;
        return DG_OK;
    }
    if (nm == "Generator.Input.b") {
        HIP_TRY(hipMemcpy(h->lin_b, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        h->have[nm] = true;
        return DG_OK;
    }
    for (size_t d = 0; d < h->dec.size(); ++d) {
        const DeconvSpec& s = h->dec[d];
        if (nm == std::string(s.name) + ".Filters") {
            std::vector<float> t((size_t)n);
            for (int k = 0; k < 25; ++k)
                for (int co = 0; co < s.cout; ++co)
                    for (int ci = 0; ci < s.cin; ++ci)
                        t[((size_t)k * s.cin + ci) * s.cout + co] = host[((size_t)k * s.cout + co) * s.cin + ci];
            HIP_TRY(hipMemcpy(h->F[d], host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->Ft[d], t.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            if (d + 1 < h->dec.size() && s.cout % 32 == 0 && s.cin % 8 == 0) {
                // forward filters in fragment order (dg_fgemm.hip): the slab of tap k keeps its float offset k * cout * cin; inside it
                // [cout / 32][cin / 8][64 lanes][4]: lane = ((ci % 8) / 4) * 32 + co % 32, element ci % 4
                std::vector<float> pk((size_t)n);
                const int kc8 = s.cin / 8;
                for (int k = 0; k < 25; ++k)
                    for (int co = 0; co < s.cout; ++co)
                        for (int ci = 0; ci < s.cin; ++ci)
                            pk[(size_t)k * s.cout * s.cin + (((size_t)(co / 32) * kc8 + ci / 8) * 64 + ((ci % 8) / 4) * 32 + co % 32) * 4 + ci % 4] =
                                host[((size_t)k * s.cout + co) * s.cin + ci];
                if (!h->Fp[d]) HIP_TRY(hipMalloc(&h->Fp[d], (size_t)n * sizeof(float)));
                HIP_TRY(hipMemcpy(h->Fp[d], pk.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            }
            if (d + 1 == h->dec.size()) {
                // tail forward GEMM: B fragments of v_mfma_f32_32x32x2_f32 in load order.  kappa = (kh*5+kw)*cout + co is
                // the row of host[] (layout [25][cout][cin] == [kappa][c]); lane = (kappa & 31) + 32*half holds
                // c = 8*kk + 4*half + e of kappa tile t.
                const int nk = 25 * s.cout, nt = (nk + 31) / 32, kkn = s.cin / 8;
                std::vector<float> pk((size_t)nt * kkn * 64 * 4, 0.f);
                for (int tt = 0; tt < nt; ++tt)
                    for (int kk = 0; kk < kkn; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int kappa = tt * 32 + (lane & 31), c = kk * 8 + (lane >> 5) * 4 + e;
                                if (kappa < nk) pk[(((size_t)tt * kkn + kk) * 64 + lane) * 4 + e] = host[(size_t)kappa * s.cin + c];
                            }
#ifdef DG_MEASURE      // only the superseded 32-wide CelebA forward tail reads this pack
                if (!h->tail_pack) HIP_TRY(hipMalloc(&h->tail_pack, pk.size() * sizeof(float)));
                HIP_TRY(hipMemcpy(h->tail_pack, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
#else
                (void)pk;
#endif
                if (s.cout == 3) {
                    // B fragments of v_mfma_f32_16x16x4_f32, one 16-column tile per filter row kh: column
                    // j = kw*3 + co (< 15) is kappa = 15*kh + j; lane = j + 16*g holds c = 16*kk + 4*g + e.
                    const int kk16 = s.cin / 16;
                    std::vector<float> p16((size_t)5 * kk16 * 64 * 4, 0.f);
                    for (int kh = 0; kh < 5; ++kh)
                        for (int kk = 0; kk < kk16; ++kk)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 4; ++e) {
                                    const int j = lane & 15, c = kk * 16 + (lane >> 4) * 4 + e;
                                    if (j < 15) p16[(((size_t)kh * kk16 + kk) * 64 + lane) * 4 + e] = host[(size_t)(15 * kh + j) * s.cin + c];
                                }
                    if (!h->tail_pack16) HIP_TRY(hipMalloc(&h->tail_pack16, p16.size() * sizeof(float)));
                    HIP_TRY(hipMemcpy(h->tail_pack16, p16.data(), p16.size() * sizeof(float), hipMemcpyHostToDevice));
                }
            }
            h->have[nm] = true;
            return DG_OK;
        }
        if (nm == std::string(s.name) + ".Biases") {
            HIP_TRY(hipMemcpy(h->bias[d], host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            h->have[nm] = true;
            return DG_OK;
        }
    }
    for (auto& a : h->ai) {
        if (!a.has_bn) continue;
        for (int which = 0; which < 2; ++which) {
            if (nm != a.bn_name + (which ? ".offset" : ".scale")) continue;
            HIP_TRY(hipMemcpy(which ? a.offset : a.scale, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
            h->have[nm] = true;
            return DG_OK;
        }
    }
    return fail(DG_E_INVALID, "unknown weight name '%s'", name);
}

int dg_weights_complete(dg_handle* h) {
    if (!h) return 0;
    if (!h->have.count("Generator.Input.W") || !h->have.count("Generator.Input.b")) return 0;
    for (auto& s : h->dec) {
        if (!h->have.count(std::string(s.name) + ".Filters")) return 0;
        if (!h->have.count(std::string(s.name) + ".Biases")) return 0;
    }
    for (auto& a : h->ai)
        if (a.has_bn && (!h->have.count(a.bn_name + ".scale") || !h->have.count(a.bn_name + ".offset"))) return 0;
    return 1;
}

int dg_init_latents(dg_handle* h, float* z, int64_t n_rows, uint64_t seed, int64_t first_row, float std, void* stream) {
    if (!h || !z || n_rows < 0) return fail(DG_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (std <= 0.f) std = std::sqrt(1.0f / (float)h->latent);
    if (n_rows) dg::launch_init_latents(z, n_rows, h->latent, seed, first_row, std, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

int dg_reconstruct(dg_handle* h, const float* x, const float* z0, uint64_t seed, int64_t first_row, int B, int R, int L,
                   float lr, float momentum, float* out_rec, int32_t* out_idx, float* out_loss, float* out_z,
                   void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!x || !out_rec) return fail(DG_E_INVALID, "x and out_rec must be non-null");
    if (L < 0) return fail(DG_E_INVALID, "need L >= 0 (got %d)", L);
    int n_rows = 0;
    rc = check_rows(B, R, &n_rows);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_call(h, B, R, s);          // a no-op (no allocation, no wait) once this shape has been prepared
    if (rc) return rc;
    h->prof_chain_last = -1;                // profile markers never span two calls (copies and the latent draw sit between)
    const size_t zbytes = (size_t)n_rows * h->latent * sizeof(float);
    if (z0) HIP_TRY(hipMemcpyAsync(h->z, z0, zbytes, hipMemcpyDeviceToDevice, s));
    else dg::launch_init_latents(h->z, n_rows, h->latent, seed, first_row, std::sqrt(1.0f / (float)h->latent), s);
    HIP_TRY(hipMemsetAsync(h->m, 0, zbytes, s));
    // (the arrival counters of the folded update return to zero by themselves; cleared per call all the same, so that a call
    // that died half-way cannot poison the next one)
    if (update_folds(h)) HIP_TRY(hipMemsetAsync(h->upd_count, 0, (size_t)(n_rows / 32 + 16) * sizeof(unsigned), s));
    // (the fused latent turn's barrier counters are monotonic within a call and start every call at zero)
    if (turn_fuses(h)) {
        if (h->turn_err_host && *h->turn_err_host) {
            *h->turn_err_host = 0;
            return fail(DG_E_HIP, "a barrier of the fused latent turn (option turn_fused) gave up in an earlier call: its results were wrong");
        }
        HIP_TRY(hipMemsetAsync(h->turn_bar, 0, turn_bar_bytes(), s));
    }
    const int steps = L > 1 ? L : 1;
    // the batch split (by image) into row groups on separate streams (option two_streams)
    RowGroup grp[dg_handle::kMaxGroups];
    const int ngroups = split_groups(h, B, R, grp);
    grp[0].s = s;
    // (likewise the arrival counters of the K-pair jobs, before the side streams fork off this one)
    for (int gi = 0; gi < ngroups; ++gi) {
        bool seen = false;
        for (int gj = 0; gj < gi; ++gj) seen = seen || grp[gj].n_rows == grp[gi].n_rows;
        if (!seen) { rc = clear_pair_counters(h, grp[gi].n_rows, s); if (rc) return rc; }
    }

    // Small prepared shapes replay a captured graph of the loop instead of enqueuing its ~8 L launches one by one
    bool replayed = false;
    if (h->graph_max_rows > 0 && n_rows <= h->graph_max_rows && !h->graph_broken && h->prof_stride == 0 && ngroups == 1 &&
        steps >= 2 && (int64_t)B * h->P <= h->xbuf_floats) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            hipGraphExec_t exec = loop_graph(h, B, R, L, lr, momentum);
            if (exec) {
                // The graph runs on the engine's own stream, tied to the caller's by events on both sides (measured: a graph
                // launched straight into the legacy NULL stream -- torch's default stream -- was not ordered against the
                // launches that followed it there)
                HIP_TRY(hipMemcpyAsync(h->xbuf, x, (size_t)B * h->P * sizeof(float), hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipEventRecord(h->ev_fork, s));
                HIP_TRY(hipStreamWaitEvent(h->cap_stream, h->ev_fork, 0));
                HIP_TRY(hipGraphLaunch(exec, h->cap_stream));
                HIP_TRY(hipEventRecord(h->ev_join[0], h->cap_stream));
                HIP_TRY(hipStreamWaitEvent(s, h->ev_join[0], 0));
                replayed = true;
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (!replayed) {
        rc = run_groups(h, x, R, L, lr, momentum, s, grp, ngroups);
        if (rc) return rc;
    }
    dg::launch_select(h->loss, h->y, B, R, h->P, out_rec, out_idx, s);
    if (out_loss) HIP_TRY(hipMemcpyAsync(out_loss, h->loss, (size_t)n_rows * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_z) HIP_TRY(hipMemcpyAsync(out_z, h->z, zbytes, hipMemcpyDeviceToDevice, s));
    if (turn_fuses(h)) {          // the turn kernel's "a barrier gave up" word travels to the host behind the call; read by the next call
        if (!h->turn_err_host) HIP_TRY(hipHostMalloc((void**)&h->turn_err_host, sizeof(unsigned), hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(h->turn_err_host, h->turn_bar + (size_t)dg_handle::kMaxGroups * dg_handle::kTurnBarWords, sizeof(unsigned),
                               hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

int dg_call_row_groups(dg_handle* h, int B, int R) {
    if (check_ready(h)) return -1;
    int n_rows = 0;
    if (check_rows(B, R, &n_rows)) return -1;
    RowGroup grp[dg_handle::kMaxGroups];
    return split_groups(h, B, R, grp);
}

int dg_prepare(dg_handle* h, int B, int R, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    int n_rows = 0;
    rc = check_rows(B, R, &n_rows);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_call(h, B, R, s);
    if (rc) return rc;
    // dg_loss_grad / dg_generate run all rows as one group
    rc = prepare_rows(h, n_rows, &n_rows, 1, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return DG_OK;
}

int dg_generate(dg_handle* h, const float* z, int N, float* out_y, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!z || !out_y) return fail(DG_E_INVALID, "bad argument");
    int n_checked = 0;
    rc = check_rows(N, 1, &n_checked);
    if (rc) return rc;
    // every row is compared with image 0 by passing R = N: the tails' row -> image division (a 40-bit magic multiply) is exact
    // for row * R < 2^40 only
    if (N > (1 << 20)) return fail(DG_E_INVALID, "dg_generate: at most %d rows per call (got %d): split the batch", 1 << 20, N);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_rows(h, N, &N, 1, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->z, z, (size_t)N * h->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    // the loss is discarded here: every row is compared with one all-zero image (R = N -> image 0)
    RowGroup g; g.n_rows = N; g.s = s;
    rc = run_forward(h, h->xzero, g, /*R=*/N, /*want_y=*/true, /*want_loss=*/false, /*tail_backward=*/false, false);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_y, h->y, (size_t)N * h->P * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

int dg_loss_grad(dg_handle* h, const float* x, const float* z, int B, int R, float* out_y, float* out_loss,
                 float* out_dz, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!x || !z) return fail(DG_E_INVALID, "bad argument");
    int n_rows = 0;
    rc = check_rows(B, R, &n_rows);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = prepare_rows(h, n_rows, &n_rows, 1, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->z, z, (size_t)n_rows * h->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    rc = clear_pair_counters(h, n_rows, s);
    if (rc) return rc;
    RowGroup g; g.n_rows = n_rows; g.s = s;
    rc = run_forward(h, x, g, R, /*want_y=*/out_y != nullptr, /*want_loss=*/out_loss != nullptr, /*tail_backward=*/out_dz != nullptr, false);
    if (rc) return rc;
    if (out_y) HIP_TRY(hipMemcpyAsync(out_y, h->y, (size_t)n_rows * h->P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_loss) HIP_TRY(hipMemcpyAsync(out_loss, h->loss, (size_t)n_rows * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_dz) {
        rc = run_backward(h, g, false);
        if (rc) return rc;
        dg::launch_momentum_update(nullptr, nullptr, h->part, h->nsplit, n_rows, h->latent, 0.f, 0.f, out_dz, s);
    }
    HIP_TRY(hipGetLastError());
    return DG_OK;
}

}  // extern "C"
