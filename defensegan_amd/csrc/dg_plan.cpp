#include "dg_plan.h"

#include <algorithm>
#include <cstdio>
#include <functional>
#include <map>
#include <queue>
#include <utility>

namespace dg {

static void sort_positions(LayerPlan& p) {
    std::stable_sort(p.pos.begin(), p.pos.end(),
                     [](const PosEntry& a, const PosEntry& b) { return a.tap_count > b.tap_count; });
}

LayerPlan plan_linear_fwd(int latent, int features, int bn) {
    LayerPlan p;
    p.name = "linear_fwd";
    p.a_rowstride = latent;
    p.out_rowstride = features;
    p.w_rowstride = latent;
    p.kch = latent;
    p.ncols = features;
    p.bn = bn;
    p.taps.push_back(TapEntry{0, 0});
    for (int n0 = 0; n0 < features; n0 += bn) p.pos.push_back(PosEntry{0, n0, 0, 1});
    p.macs_per_row = (long long)latent * features;
    return p;
}

LayerPlan plan_linear_bwd(int latent, int features, int nsplit, int bn) {
    LayerPlan p;
    p.name = "linear_bwd";
    const int kc = features / nsplit;
    p.a_rowstride = features;
    p.out_rowstride = (long long)nsplit * latent;
    p.w_rowstride = features;
    p.kch = kc;
    p.ncols = latent;
    p.bn = bn;
    for (int s = 0; s < nsplit; ++s) {
        p.taps.push_back(TapEntry{s * kc, s * kc});
        for (int n0 = 0; n0 < latent; n0 += bn) p.pos.push_back(PosEntry{s * latent, n0, s, 1});
    }
    p.macs_per_row = (long long)latent * features;
    return p;
}

LayerPlan plan_deconv_fwd(int h_in, int in_pitch, int e_out, int out_pitch, int cin, int cout, int bn) {
    LayerPlan p;
    p.name = "deconv_fwd";
    p.a_rowstride = (long long)in_pitch * in_pitch * cin;
    p.out_rowstride = (long long)out_pitch * out_pitch * cout;
    p.w_rowstride = cin;
    p.kch = cin;
    p.ncols = cout;
    p.bn = bn;
    for (int i = 0; i < e_out; ++i)
        for (int j = 0; j < e_out; ++j) {
            const int tb = (int)p.taps.size();
            for (int kh = 0; kh < 5; ++kh) {
                const int th = i + 1 - kh;
                if (th < 0 || (th & 1)) continue;
                const int oh = th >> 1;
                if (oh >= h_in) continue;
                for (int kw = 0; kw < 5; ++kw) {
                    const int tw = j + 1 - kw;
                    if (tw < 0 || (tw & 1)) continue;
                    const int ow = tw >> 1;
                    if (ow >= h_in) continue;
                    p.taps.push_back(TapEntry{(oh * in_pitch + ow) * cin, (kh * 5 + kw) * cout * cin});
                }
            }
            const int tc = (int)p.taps.size() - tb;
            p.macs_per_row += (long long)tc * cin * cout;
            for (int n0 = 0; n0 < cout; n0 += bn)
                p.pos.push_back(PosEntry{(i * out_pitch + j) * cout, n0, tb, tc});
        }
    sort_positions(p);
    return p;
}

LayerPlan plan_deconv_bwd(int h_in, int out_pitch, int e_out, int a_pitch, int cin, int cout, int bn) {
    LayerPlan p;
    p.name = "deconv_bwd";
    p.a_rowstride = (long long)a_pitch * a_pitch * cout;
    p.out_rowstride = (long long)out_pitch * out_pitch * cin;
    p.w_rowstride = cout;
    p.kch = cout;
    p.ncols = cin;
    p.bn = bn;
    for (int oh = 0; oh < h_in; ++oh)
        for (int ow = 0; ow < h_in; ++ow) {
            const int tb = (int)p.taps.size();
            for (int kh = 0; kh < 5; ++kh) {
                const int i = 2 * oh + kh - 1;
                if (i < 0 || i >= e_out) continue;
                for (int kw = 0; kw < 5; ++kw) {
                    const int j = 2 * ow + kw - 1;
                    if (j < 0 || j >= e_out) continue;
                    p.taps.push_back(TapEntry{(i * a_pitch + j) * cout, (kh * 5 + kw) * cin * cout});
                }
            }
            const int tc = (int)p.taps.size() - tb;
            p.macs_per_row += (long long)tc * cin * cout;
            for (int n0 = 0; n0 < cin; n0 += bn)
                p.pos.push_back(PosEntry{(oh * out_pitch + ow) * cin, n0, tb, tc});
        }
    sort_positions(p);
    return p;
}

void plan_add_zero_positions(LayerPlan& p, int used, int pitch, int ncols) {
    for (int i = 0; i < pitch; ++i)
        for (int j = 0; j < pitch; ++j) {
            if (i < used && j < used) continue;
            for (int n0 = 0; n0 < ncols; n0 += p.bn) p.pos.push_back(PosEntry{(i * pitch + j) * ncols, n0, 0, 0});
        }
}


BatchedPlan make_batched(const LayerPlan& p) {
    BatchedPlan b;
    b.name = p.name;
    b.a_rowstride = p.a_rowstride;
    b.out_rowstride = p.out_rowstride;
    b.w_rowstride = p.w_rowstride;
    b.kch = p.kch;
    b.ncols = p.ncols;
    b.macs_per_row = p.macs_per_row;
    typedef std::vector<std::pair<int, int>> Sig;            // (a_off relative to the smallest one, w_off) per tap, in order
    std::map<Sig, int> index;
    std::vector<Sig> sigs;
    std::vector<std::vector<std::pair<int, int>>> members;   // per class: (pos_a, pos_out) of its positions
    for (const PosEntry& pe : p.pos) {
        int a_min = 0;
        for (int t = 0; t < pe.tap_count; ++t) {
            const int a = p.taps[pe.tap_begin + t].a_off;
            if (t == 0 || a < a_min) a_min = a;
        }
        Sig sig;
        for (int t = 0; t < pe.tap_count; ++t)
            sig.push_back(std::make_pair(p.taps[pe.tap_begin + t].a_off - a_min, p.taps[pe.tap_begin + t].w_off));
        auto it = index.find(sig);
        int c;
        if (it == index.end()) {
            c = (int)sigs.size();
            index[sig] = c;
            sigs.push_back(sig);
            members.push_back({});
        } else {
            c = it->second;
        }
        members[c].push_back(std::make_pair(a_min, pe.out_off));
    }
    std::vector<int> order(sigs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sigs[x].size() > sigs[y].size(); });
    const int cpt = p.kch / 32;
    for (int c : order) {
        ClassDesc cd = {};
        cd.pos_begin = (int)b.pos_a.size();
        cd.pos_count = (int)members[c].size();
        cd.tap_begin = (int)b.taps.size();
        cd.nchunks = (int)sigs[c].size() * cpt;
        cd.magic = (unsigned)(((1ULL << 31) + (unsigned)cd.pos_count - 1) / (unsigned)cd.pos_count);
        for (const auto& m : members[c]) { b.pos_a.push_back(m.first); b.pos_out.push_back(m.second); }
        for (const auto& t : sigs[c]) b.taps.push_back(TapEntry{t.first, t.second});
        // is the class a grid?  smallest row length wc (a divisor of the position count) for which both tables are affine in
        // (j / wc, j % wc)
        const std::vector<std::pair<int, int>>& mem = members[c];
        const int s = cd.pos_count;
        for (int wc = 1; wc <= s && cd.wc == 0; ++wc) {
            if (s % wc) continue;
            const int a_cs = wc > 1 ? mem[1].first - mem[0].first : 0, o_cs = wc > 1 ? mem[1].second - mem[0].second : 0;
            const int a_rs = s > wc ? mem[(size_t)wc].first - mem[0].first : 0, o_rs = s > wc ? mem[(size_t)wc].second - mem[0].second : 0;
            bool ok = true;
            for (int j = 0; j < s && ok; ++j)
                ok = mem[(size_t)j].first == mem[0].first + (j / wc) * a_rs + (j % wc) * a_cs &&
                     mem[(size_t)j].second == mem[0].second + (j / wc) * o_rs + (j % wc) * o_cs;
            if (!ok) continue;
            cd.wc = wc;
            cd.wc_magic = (unsigned)(((1ULL << 31) + (unsigned)wc - 1) / (unsigned)wc);
            cd.a_base = mem[0].first; cd.a_rs = a_rs; cd.a_cs = a_cs;
            cd.o_base = mem[0].second; cd.o_rs = o_rs; cd.o_cs = o_cs;
        }
        b.cls.push_back(cd);
    }
    return b;
}

namespace {

// job shapes (rows x columns) by (family, level); level + 1 is the same tile cut in two: along M, except family 0's last cut
const int kShapeBM[2][3] = {{128, 64, 64}, {256, 128, 64}};
const int kShapeBN[2][3] = {{128, 128, 64}, {64, 64, 64}};

int shape_bm(int family, int shape) { return kShapeBM[family][shape]; }
int shape_bn(int family, int shape) { return kShapeBN[family][shape]; }

double job_us(const BatchedPlan& p, const JobDesc& j, int family, int slots, const JobModel& m) {
    // a job occupies its tile's full MFMA footprint whatever m_valid is
    (void)p;
    const double flop = 2.0 * j.nchunks * 32.0 * shape_bm(family, j.shape) * shape_bn(family, j.shape);      // the job's own K range
    return flop / (m.rate[family][j.shape] * 1e6 / slots) + m.fixed_us[family][j.shape] + (j.pair_id ? m.pair_us[j.shape] : 0.0);
}

// Turns pieces (class, level, M range, column) into job records: one job, or -- classes of K-pair jobs -- the two half-K jobs, next to
// each other; hands out the pairs' counters and scratch images and the classes' statistics blocks.
struct JobEmitter {
    const BatchedPlan& p;
    int family;
    int cpt;
    int n_pairs = 0;
    long long pair_floats = 0;
    std::vector<int> stat_base;
    JobEmitter(const BatchedPlan& plan, int n_rows, int fam) : p(plan), family(fam), cpt(plan.kch / 32), stat_base(plan.cls.size(), 0) {
        long long at = 0;
        for (size_t c = 0; c < p.cls.size(); ++c) { stat_base[c] = (int)at; at += ((long long)n_rows * p.cls[c].pos_count + 31) / 32; }
    }
    int half_taps(int cls, int role) const {
        const int nt = p.cls[(size_t)cls].nchunks / cpt, first = pair_first_taps(p, cls);
        return role == 0 ? first : nt - first;
    }
    void emit(std::vector<JobDesc>& out, int cls, int level, long long m0, int rows, int n0) {
        const ClassDesc& cd = p.cls[(size_t)cls];
        const int s = cd.pos_count;
        const bool paired = class_is_paired(p, cls);
        for (int role = 0; role < (paired ? 2 : 1); ++role) {
            JobDesc j = {};
            j.cls = cls;
            j.shape = level;
            j.n0 = n0;
            j.n_first = (int)(m0 / s);
            j.j_first = (int)(m0 % s);
            j.m_valid = rows;
            j.pos_begin = cd.pos_begin;
            j.pos_count = cd.pos_count;
            j.magic = cd.magic;
            j.stat_base = stat_base[(size_t)cls];
            j.wc = cd.wc; j.wc_magic = cd.wc_magic;
            j.a_base = cd.a_base; j.a_rs = cd.a_rs; j.a_cs = cd.a_cs;
            j.o_base = cd.o_base; j.o_rs = cd.o_rs; j.o_cs = cd.o_cs;
            // the job's own taps: all of the class's, or -- K-pair jobs -- its half
            const int t0_tap = paired && role == 1 ? pair_first_taps(p, cls) : 0;
            j.n_taps = paired ? half_taps(cls, role) : cd.nchunks / cpt;
            j.tap_begin = cd.tap_begin + t0_tap;
            j.nchunks = j.n_taps * cpt;
            if (j.nchunks > 0) { j.tap0_a_off = p.taps[(size_t)j.tap_begin].a_off; j.tap0_w_off = p.taps[(size_t)j.tap_begin].w_off; }
            if (paired) {
                j.pair_id = n_pairs + 1;
                j.pair_role = role;
                j.pair_off = (int)(pair_floats / 256);
            }
            out.push_back(j);
        }
        if (paired) {
            ++n_pairs;
            pair_floats += 2LL * shape_bm(family, level) * shape_bn(family, level);
        }
    }
};

// One pass of "longest first with cutting on demand": jobs are handed to the earliest free of `slots` servers in descending
// cost; a job that would end after `target` is cut in two along M (then along N) and its pieces go back into the pool, so
// the launch starts with whole tiles and ends with small ones.  The order of assignment is the dispatch order.
std::vector<JobDesc> jobs_for_target(const BatchedPlan& p, int n_rows, int family, int slots, double slack, int min_level,
                                     const JobModel& model) {
    const int BM = kShapeBM[family][0], BN = kShapeBN[family][0];
    const int max_level = 2;
    if (min_level > max_level) min_level = max_level;
    auto level_for_rows = [&](int rows, int level) {          // a ragged last tile starts at the level that still holds it
        while (level < max_level && kShapeBM[family][level + 1] >= rows && kShapeBN[family][level + 1] == kShapeBN[family][level]) ++level;
        return level;
    };
    // (a piece of a paired class stands for BOTH its K-pair jobs: they are cut together -- their shapes must match -- and `us`
    // is the longer half's cost)
    struct Piece { double us; int cls; int level; long long m0; int rows; int n0; unsigned seq; };
    const int cpt = p.kch / 32;
    auto half_taps = [&](int cls, int role) {
        const int nt = p.cls[(size_t)cls].nchunks / cpt, first = pair_first_taps(p, cls);
        return role == 0 ? first : nt - first;
    };
    auto cmp = [](const Piece& a, const Piece& b) { return a.us < b.us || (a.us == b.us && a.seq > b.seq); };
    std::priority_queue<Piece, std::vector<Piece>, decltype(cmp)> pool(cmp);
    unsigned seq = 0;
    double total = 0.0;
    auto cost = [&](int cls, int level) {
        JobDesc j = {};
        j.cls = cls; j.shape = level;
        j.nchunks = p.cls[(size_t)cls].nchunks;
        if (class_is_paired(p, cls)) { j.pair_id = 1; j.nchunks = half_taps(cls, 1) * cpt; }
        return job_us(p, j, family, slots, model);
    };
    auto n_jobs_of = [&](int cls) { return class_is_paired(p, cls) ? 2 : 1; };
    for (int c = 0; c < (int)p.cls.size(); ++c) {
        const long long M = (long long)n_rows * p.cls[c].pos_count;
        for (long long m0 = 0; m0 < M; m0 += BM)
            for (int n0 = 0; n0 < p.ncols; n0 += BN) {
                const int rows = (int)std::min<long long>(BM, M - m0);
                const int level = level_for_rows(rows, min_level);
                const int bm = shape_bm(family, level), bn = shape_bn(family, level);
                for (int r0 = 0; r0 < rows; r0 += bm)
                    for (int c0 = 0; c0 < BN; c0 += bn) {
                        Piece pc = {cost(c, level), c, level, m0 + r0, std::min(bm, rows - r0), n0 + c0, seq++};
                        total += pc.us * n_jobs_of(c);
                        pool.push(pc);
                    }
            }
    }
    const double target = slack >= 1e20 ? 1e300 : total / slots * slack;
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int i = 0; i < slots; ++i) free_at.push(0.0);
    std::vector<JobDesc> out;
    JobEmitter em(p, n_rows, family);
    while (!pool.empty()) {
        Piece pc = pool.top();
        pool.pop();
        const double t0 = free_at.top();
        const double ideal = total / slots;
        const bool late = model.taper > 0.0 && pc.level < max_level &&
                          t0 > (pc.level == 0 ? model.taper : 0.5 * (model.taper + 1.0)) * ideal;
        if ((t0 + pc.us > target || late) && pc.level < max_level) {
            // cut in two: along M while the next level is lower, else along N
            const int nl = pc.level + 1;
            const int bm = shape_bm(family, nl), bn = shape_bn(family, nl);
            const int cols = shape_bn(family, pc.level);
            for (int r0 = 0; r0 < pc.rows; r0 += bm)
                for (int c0 = 0; c0 < cols; c0 += bn) {
                    Piece q = {cost(pc.cls, nl), pc.cls, nl, pc.m0 + r0, std::min(bm, pc.rows - r0), pc.n0 + c0, seq++};
                    pool.push(q);
                }
            continue;
        }
        for (int role = 0; role < n_jobs_of(pc.cls); ++role) {
            const double ts = free_at.top();                 // the second half of a K-pair takes the next free slot
            free_at.pop();
            free_at.push(ts + pc.us);
        }
        em.emit(out, pc.cls, pc.level, pc.m0, pc.rows, pc.n0);
    }
    return out;
}

}  // namespace

std::vector<JobDesc> jobs_balanced(const BatchedPlan& p, int n_rows, int family, int cus, int slots_per_cu, int min_level,
                                   const JobModel& model, double tol) {
    const int BM = kShapeBM[family][0], BN = kShapeBN[family][0];
    const int max_level = 2;
    if (min_level > max_level) min_level = max_level;
    const int slots = cus * slots_per_cu;
    struct Piece { double us; int cls; int level; long long m0; int rows; int n0; int njobs; };
    const int cpt = p.kch / 32;
    auto cost = [&](int cls, int level) {
        JobDesc j = {};
        j.cls = cls; j.shape = level;
        j.nchunks = p.cls[(size_t)cls].nchunks;
        if (class_is_paired(p, cls)) { j.pair_id = 1; j.nchunks = (p.cls[(size_t)cls].nchunks / cpt - pair_first_taps(p, cls)) * cpt; }
        return job_us(p, j, family, slots, model);
    };
    std::vector<Piece> pieces;
    long long n_jobs = 0;
    for (int c = 0; c < (int)p.cls.size(); ++c) {
        const long long M = (long long)n_rows * p.cls[c].pos_count;
        const int nj = class_is_paired(p, c) ? 2 : 1;
        for (long long m0 = 0; m0 < M; m0 += BM)
            for (int n0 = 0; n0 < p.ncols; n0 += BN) {
                const int rows = (int)std::min<long long>(BM, M - m0);
                int level = min_level;
                while (level < max_level && kShapeBM[family][level + 1] >= rows && kShapeBN[family][level + 1] == kShapeBN[family][level]) ++level;
                const int bm = shape_bm(family, level), bn = shape_bn(family, level);
                for (int r0 = 0; r0 < rows; r0 += bm)
                    for (int c0 = 0; c0 < BN; c0 += bn) {
                        pieces.push_back(Piece{cost(c, level) * nj, c, level, m0 + r0, std::min(bm, rows - r0), n0 + c0, nj});
                        n_jobs += nj;
                    }
            }
    }
    if (n_jobs > slots || pieces.size() <= (size_t)cus) return {};
    std::vector<std::vector<size_t>> bin;
    std::vector<double> load;
    std::vector<int> count;
    auto assign = [&]() -> bool {
        std::vector<size_t> idx(pieces.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return pieces[a].us > pieces[b].us; });
        bin.assign((size_t)cus, {});
        load.assign((size_t)cus, 0.0);
        count.assign((size_t)cus, 0);
        for (size_t i : idx) {
            int best = -1;
            for (int b = 0; b < cus; ++b)
                if (count[(size_t)b] + pieces[i].njobs <= slots_per_cu && (best < 0 || load[(size_t)b] < load[(size_t)best])) best = b;
            if (best < 0) return false;
            bin[(size_t)best].push_back(i);
            load[(size_t)best] += pieces[i].us;
            count[(size_t)best] += pieces[i].njobs;
        }
        return true;
    };
    if (!assign()) return {};
    for (int iter = 0; iter < 4096; ++iter) {
        double total = 0.0;
        int heavy = 0;
        for (int b = 0; b < cus; ++b) { total += load[(size_t)b]; if (load[(size_t)b] > load[(size_t)heavy]) heavy = b; }
        if (load[(size_t)heavy] <= (1.0 + tol) * total / cus) break;
        // the largest piece of the heaviest CU that can still be cut -- if the list has room for one more piece
        size_t pick = pieces.size();
        for (size_t i : bin[(size_t)heavy])
            if (pieces[i].level < max_level && (pick == pieces.size() || pieces[i].us > pieces[pick].us)) pick = i;
        if (pick == pieces.size() || n_jobs + pieces[pick].njobs > slots) break;
        const Piece pc = pieces[pick];
        const int nl = pc.level + 1;
        const int bm = shape_bm(family, nl), bn = shape_bn(family, nl), cols = shape_bn(family, pc.level);
        std::vector<Piece> parts;
        for (int r0 = 0; r0 < pc.rows; r0 += bm)
            for (int c0 = 0; c0 < cols; c0 += bn)
                parts.push_back(Piece{cost(pc.cls, nl) * pc.njobs, pc.cls, nl, pc.m0 + r0, std::min(bm, pc.rows - r0), pc.n0 + c0, pc.njobs});
        if (parts.size() < 2) {                      // a ragged piece that the next level holds whole: just relabel it
            load[(size_t)heavy] += parts[0].us - pc.us;
            pieces[pick] = parts[0];
            continue;
        }
        // the first part stays, the others move to the lightest CUs that have a slot -- if that lowers the heaviest load
        bool moved = false;
        std::vector<std::pair<size_t, int>> placed;                       // (piece index, bin)
        double heavy_load = load[(size_t)heavy] - pc.us + parts[0].us;
        bool ok = true;
        std::vector<double> load_try = load;
        std::vector<int> count_try = count;
        load_try[(size_t)heavy] = heavy_load;
        for (size_t k = 1; k < parts.size() && ok; ++k) {
            int best = -1;
            for (int b = 0; b < cus; ++b)
                if (b != heavy && count_try[(size_t)b] + pc.njobs <= slots_per_cu && (best < 0 || load_try[(size_t)b] < load_try[(size_t)best])) best = b;
            if (best < 0 || load_try[(size_t)best] + parts[k].us >= load[(size_t)heavy]) { ok = false; break; }
            load_try[(size_t)best] += parts[k].us;
            count_try[(size_t)best] += pc.njobs;
            placed.push_back(std::make_pair(k, best));
            moved = true;
        }
        if (!ok || !moved) break;                    // no CU can take a part without becoming the heaviest itself: done
        pieces[pick] = parts[0];
        for (const auto& pl : placed) {
            pieces.push_back(parts[pl.first]);
            bin[(size_t)pl.second].push_back(pieces.size() - 1);
        }
        load = load_try;
        count = count_try;
        n_jobs += (long long)placed.size() * pc.njobs;
    }
    // dispatch order: CU b's jobs at positions b, b + cus, ...; the CUs with the most jobs first, so that every round is a prefix
    std::vector<int> order((size_t)cus);
    for (int b = 0; b < cus; ++b) order[(size_t)b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return count[(size_t)a] > count[(size_t)b]; });
    JobEmitter em(p, n_rows, family);
    std::vector<std::vector<JobDesc>> per((size_t)cus);
    for (int b = 0; b < cus; ++b)
        for (size_t i : bin[(size_t)order[(size_t)b]])
            em.emit(per[(size_t)b], pieces[i].cls, pieces[i].level, pieces[i].m0, pieces[i].rows, pieces[i].n0);
    std::vector<JobDesc> out;
    for (int r = 0; r < slots_per_cu; ++r)
        for (int b = 0; b < cus; ++b)
            if ((size_t)r < per[(size_t)b].size()) out.push_back(per[(size_t)b][(size_t)r]);
    return out;
}

long long stat_blocks(const BatchedPlan& p, int n_rows) {
    long long at = 0;
    for (const ClassDesc& c : p.cls) at += ((long long)n_rows * c.pos_count + 31) / 32;
    return at;
}

PairNeeds pair_needs(const std::vector<JobDesc>& jobs, int family) {
    PairNeeds n;
    for (const JobDesc& j : jobs) {
        if (!j.pair_id) continue;
        n.pairs = std::max(n.pairs, j.pair_id);
        n.floats = std::max(n.floats, 256LL * j.pair_off + 2LL * shape_bm(family, j.shape) * shape_bn(family, j.shape));
    }
    return n;
}

void order_for_xcd(std::vector<JobDesc>& jobs, int n_rows, double head_frac, int n_xcd) {
    if (n_xcd < 2 || n_rows < n_xcd || head_frac <= 0.0) return;
    size_t head = (size_t)((double)jobs.size() * (head_frac < 1.0 ? head_frac : 1.0));
    head -= head % (size_t)n_xcd;
    if (head < (size_t)(2 * n_xcd)) return;
    std::vector<std::vector<JobDesc>> bucket((size_t)n_xcd);
    for (size_t i = 0; i < head; ++i) {
        long long x = (long long)jobs[i].n_first * n_xcd / n_rows;
        if (x >= n_xcd) x = n_xcd - 1;
        bucket[(size_t)x].push_back(jobs[i]);
    }
    for (auto& b : bucket)          // ascending latent row; jobs of one row keep their (cost) order
        std::stable_sort(b.begin(), b.end(), [](const JobDesc& a, const JobDesc& c) { return a.n_first < c.n_first; });
    // slot i of the list goes to XCD i mod n_xcd: deal the buckets out round-robin; a bucket that runs dry (row ranges need not
    // hold the same number of jobs) is skipped, the longer ones then share its slots
    std::vector<size_t> at((size_t)n_xcd, 0);
    size_t out = 0;
    while (out < head)
        for (int x = 0; x < n_xcd && out < head; ++x)
            if (at[(size_t)x] < bucket[(size_t)x].size()) jobs[out++] = bucket[(size_t)x][at[(size_t)x]++];
}

void snake_order(std::vector<JobDesc>& jobs, int cus) {
    if (cus <= 0) return;
    for (size_t g0 = (size_t)cus; g0 < jobs.size(); g0 += 2 * (size_t)cus)
        std::reverse(jobs.begin() + g0, jobs.begin() + std::min(jobs.size(), g0 + (size_t)cus));
}

void balance_order(const BatchedPlan& p, std::vector<JobDesc>& jobs, int family, int cus, int slots_per_cu, const JobModel& model) {
    const size_t n = jobs.size();
    if (cus <= 0 || n <= (size_t)cus || n > (size_t)cus * (size_t)slots_per_cu) return;
    const int slots = cus * slots_per_cu;
    std::vector<size_t> idx(n);
    std::vector<double> us(n);
    for (size_t i = 0; i < n; ++i) { idx[i] = i; us[i] = job_us(p, jobs[i], family, slots, model); }
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return us[a] > us[b]; });
    // bins 0 .. (n mod cus) - 1 take one job more: every round of `cus` positions but the last is full, so that position
    // r * cus + b really is bin b's r-th job
    const size_t lo = n / (size_t)cus, extra = n % (size_t)cus;
    std::vector<std::vector<size_t>> bin((size_t)cus);
    std::vector<double> sum((size_t)cus, 0.0);
    for (size_t k = 0; k < n; ++k) {
        size_t best = (size_t)cus;
        for (size_t b = 0; b < (size_t)cus; ++b) {
            const size_t cap = lo + (b < extra ? 1 : 0);
            if (bin[b].size() >= cap) continue;
            if (best == (size_t)cus || sum[b] < sum[best]) best = b;
        }
        bin[best].push_back(idx[k]);
        sum[best] += us[idx[k]];
    }
    std::vector<JobDesc> out;
    out.reserve(n);
    for (size_t r = 0; r <= lo; ++r)
        for (size_t b = 0; b < (size_t)cus; ++b)
            if (r < bin[b].size()) out.push_back(jobs[bin[b][r]]);
    jobs.swap(out);
}

void spread_order(const BatchedPlan& p, std::vector<JobDesc>& jobs, int family, int slots, const JobModel& model, double frac) {
    const size_t n = jobs.size();
    if (n <= (size_t)slots || slots <= 0) return;
    (void)p; (void)family; (void)model;
    // groups of equal length = (class, shape); `take` of the first round's slots go to the groups in proportion to their share of
    // the list (their first jobs), the other slots to the head of the list as before
    std::map<std::pair<int, int>, size_t> count, taken;
    for (const JobDesc& j : jobs) ++count[std::make_pair(j.cls, j.shape)];
    const size_t take = (size_t)(frac * slots);
    std::map<std::pair<int, int>, size_t> quota;
    for (const auto& kv : count) quota[kv.first] = (size_t)((double)take * (double)kv.second / (double)n + 0.5);
    std::vector<char> mixed(n, 0);
    size_t n_mixed = 0;
    for (size_t i = 0; i < n && n_mixed < take; ++i) {
        const auto key = std::make_pair(jobs[i].cls, jobs[i].shape);
        if (taken[key] < quota[key]) { ++taken[key]; mixed[i] = 1; ++n_mixed; }
    }
    // first round = the mixed jobs + the longest-first head of the others, in list (longest-first) order; then the rest
    std::vector<JobDesc> first, rest;
    first.reserve((size_t)slots);
    size_t head_left = (size_t)slots - n_mixed;
    for (size_t i = 0; i < n; ++i) {
        if (mixed[i]) first.push_back(jobs[i]);
        else if (head_left) { first.push_back(jobs[i]); --head_left; }
        else rest.push_back(jobs[i]);
    }
    first.insert(first.end(), rest.begin(), rest.end());
    jobs.swap(first);
}

void assign_priorities(const BatchedPlan& p, std::vector<JobDesc>& jobs, int family, int slots, const JobModel& model, int mode) {
    double longest = 0.0;
    for (const JobDesc& j : jobs) longest = std::max(longest, job_us(p, j, family, slots, model));
    for (JobDesc& j : jobs) {
        int pr = 0;
        if (mode && longest > 0.0) {
            pr = (int)(4.0 * job_us(p, j, family, slots, model) / longest);
            pr = pr > 3 ? 3 : (pr < 0 ? 0 : pr);
        }
        j.prio = pr;
    }
}

std::string format_tune_record(const TuneRecord& r) {
    char line[256];
    snprintf(line, sizeof line, "%s %d %d %.17g %d %d %.17g %d %.3f %.17g %d %d\n", r.op.c_str(), r.n_rows, r.min_level, r.slack, r.snake, r.xcd_order,
             r.xcd_head, r.n_jobs, r.measured_us, r.taper, r.prio, r.pair_kernel);
    return line;
}

bool parse_tune_record(const char** pp, TuneRecord* r) {
    const char* p = *pp;
    while (*p == '\n' || *p == ' ' || *p == '\r' || *p == '\t') ++p;
    if (!*p) { *pp = p; return false; }
    char name[32];
    int used = 0;
    TuneRecord t;
    if (sscanf(p, "%31s %d %d %lf %d %d %lf %d %lf%n", name, &t.n_rows, &t.min_level, &t.slack, &t.snake, &t.xcd_order, &t.xcd_head,
               &t.n_jobs, &t.measured_us, &used) != 9 || used <= 0)
        return false;
    {   // optional tenth field on the same line: the taper
        const char* q = p + used;
        while (*q == ' ' || *q == '\t') ++q;
        int u2 = 0;
        double tp = 0.0;
        if (*q && *q != '\n' && *q != '\r' && sscanf(q, "%lf%n", &tp, &u2) == 1 && u2 > 0) {
            t.taper = tp; used = (int)(q - p) + u2;
            // optional eleventh field: the priority mode
            q = p + used;
            while (*q == ' ' || *q == '\t') ++q;
            int pr = 0, u3 = 0;
            if (*q && *q != '\n' && *q != '\r' && sscanf(q, "%d%n", &pr, &u3) == 1 && u3 > 0) {
                t.prio = pr; used = (int)(q - p) + u3;
                // optional twelfth field: the kernel instantiation
                q = p + used;
                while (*q == ' ' || *q == '\t') ++q;
                int pk = 0, u4 = 0;
                if (*q && *q != '\n' && *q != '\r' && sscanf(q, "%d%n", &pk, &u4) == 1 && u4 > 0) { t.pair_kernel = pk; used = (int)(q - p) + u4; }
            }
        }
    }
    t.op = name;
    *r = t;
    *pp = p + used;
    return true;
}

std::vector<JobDesc> jobs_from_record(const BatchedPlan& p, int family, int cus, int slots_per_cu, const TuneRecord& r,
                                      const JobModel& model, double* predicted_us) {
    JobModel m = model;
    m.taper = r.taper;
    if (r.snake == 4) {                          // a work-balanced single-round list is built, not ordered
        std::vector<JobDesc> jb = jobs_balanced(p, r.n_rows, family, cus, slots_per_cu, r.min_level, m);
        if (predicted_us) *predicted_us = simulate_jobs(p, jb, family, cus * slots_per_cu, m);
        if (r.prio) assign_priorities(p, jb, family, cus * slots_per_cu, m, r.prio);       // (what the record says is what is launched)
        return jb;
    }
    std::vector<JobDesc> jobs = build_jobs(p, r.n_rows, family, cus * slots_per_cu, r.slack, m, predicted_us, r.min_level);
    if (r.xcd_order) order_for_xcd(jobs, r.n_rows, r.xcd_head);
    if (r.snake == 3) spread_order(p, jobs, family, cus * slots_per_cu, m);
    else if (r.snake == 2) balance_order(p, jobs, family, cus, slots_per_cu, m);
    else if (r.snake) snake_order(jobs, cus);
    if (r.prio) assign_priorities(p, jobs, family, cus * slots_per_cu, m, r.prio);
    return jobs;
}

double simulate_jobs(const BatchedPlan& p, const std::vector<JobDesc>& jobs, int family, int slots, const JobModel& model) {
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int i = 0; i < slots; ++i) free_at.push(0.0);
    double end = 0.0;
    for (const JobDesc& j : jobs) {
        const double t = free_at.top() + job_us(p, j, family, slots, model);
        free_at.pop();
        free_at.push(t);
        if (t > end) end = t;
    }
    return end;
}

std::vector<JobDesc> build_jobs(const BatchedPlan& p, int n_rows, int family, int slots, double slack, const JobModel& model,
                                double* predicted_us, int min_level) {
    std::vector<JobDesc> best;
    double best_t = 0.0;
    if (slack > 0.0) {
        best = jobs_for_target(p, n_rows, family, slots, slack, min_level, model);
        best_t = simulate_jobs(p, best, family, slots, model);
    } else {
        const double ladder[] = {1e30, 1.0, 1.02, 1.04, 1.07, 1.1, 1.15};
        for (double a : ladder) {
            std::vector<JobDesc> j = jobs_for_target(p, n_rows, family, slots, a, min_level, model);
            const double t = simulate_jobs(p, j, family, slots, model);
            if (best.empty() || t < best_t) { best.swap(j); best_t = t; }
        }
    }
    if (predicted_us) *predicted_us = best_t;
    return best;
}

// ---- fragment-order path -------------------------------------------------------------------------------------------------------
bool frag_tap_grid(const BatchedPlan& p, int cls, TapGrid* g) {
    const ClassDesc& cd = p.cls[(size_t)cls];
    const int cpt = p.kch / 32;
    const int nt = cd.nchunks / cpt;
    if (nt <= 0) return false;
    const TapEntry* t = p.taps.data() + cd.tap_begin;
    for (int nw = 1; nw <= nt; ++nw) {
        if (nt % nw) continue;
        TapGrid tg;
        tg.nw = nw;
        tg.a0 = t[0].a_off; tg.w0 = t[0].w_off;
        tg.a_v = nw > 1 ? t[1].a_off - t[0].a_off : 0;
        tg.w_v = nw > 1 ? t[1].w_off - t[0].w_off : 0;
        tg.a_u = nt > nw ? t[nw].a_off - t[0].a_off : 0;
        tg.w_u = nt > nw ? t[nw].w_off - t[0].w_off : 0;
        bool ok = true;
        for (int k = 0; k < nt && ok; ++k)
            ok = t[k].a_off == tg.a0 + (k / nw) * tg.a_u + (k % nw) * tg.a_v && t[k].w_off == tg.w0 + (k / nw) * tg.w_u + (k % nw) * tg.w_v;
        if (ok) { if (g) *g = tg; return true; }
    }
    return false;
}

int frag_ksplit(const BatchedPlan& p, int cls) {
    const int nchunks = p.cls[(size_t)cls].nchunks;
    const int kc8 = p.kch / 8;
    // (thresholds measured in round 6, profiles/r06_frag_path_ab.txt: 24 / 12 lose 1 % to 40 / 20 on both workloads -- every split costs a
    // reduction behind a barrier)
    int ks = nchunks >= 40 ? 4 : (nchunks >= 20 ? 2 : 1);
    // a wave's part = n_taps * kc8 / ks k8-steps must be a multiple of the operand ring (4): ks <= kc8 / 4
    while (ks > 1 && ks > kc8 / 4) ks >>= 1;
    return ks;
}

bool frag_supported(const BatchedPlan& p) {
    if (p.kch != 64 && p.kch != 128 && p.kch != 256) return false;
    if (p.ncols % 64 || p.w_rowstride != p.kch) return false;
    for (size_t c = 0; c < p.cls.size(); ++c) {
        if (p.cls[c].wc <= 0 || p.cls[c].nchunks <= 0) return false;
        if (!frag_tap_grid(p, (int)c, nullptr)) return false;
        // offsets are used in units of 8 floats
        if ((p.cls[c].a_base | p.cls[c].a_rs | p.cls[c].a_cs | p.cls[c].o_base | p.cls[c].o_rs | p.cls[c].o_cs) & 31) return false;
    }
    return p.a_rowstride % 32 == 0 && p.out_rowstride % 32 == 0;
}

std::vector<FragJob> build_frag_jobs(const BatchedPlan& p, int n_rows) {
    constexpr int TN = 4;
    const int nblk = (n_rows + 31) / 32;
    const int kc8 = p.kch / 8;
    struct Key { long long steps; int cls; };
    std::vector<FragJob> out;
    std::vector<int> order(p.cls.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    auto steps_of = [&](int c) { return (long long)p.cls[(size_t)c].nchunks * 4 / frag_ksplit(p, c); };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return steps_of(a) > steps_of(b); });
    for (int c : order) {
        const ClassDesc& cd = p.cls[(size_t)c];
        TapGrid tg;
        if (!frag_tap_grid(p, c, &tg)) return {};
        const int ks = frag_ksplit(p, c);
        const int per_job = TN * (4 / ks);                       // M blocks per job
        const long long mblks = (long long)nblk * cd.pos_count;
        for (long long m0 = 0; m0 < mblks; m0 += per_job)
            for (int cb = 0; cb < p.ncols / 32; cb += 2) {
                FragJob j = {};
                j.mblk0 = (int)m0;
                j.n_mblk = (int)std::min<long long>(per_job, mblks - m0);
                j.cb0 = cb;
                j.s = cd.pos_count;
                j.s_magic = cd.magic;
                j.wc = cd.wc; j.wc_magic = cd.wc_magic;
                j.a_base = cd.a_base; j.a_rs = cd.a_rs; j.a_cs = cd.a_cs;
                j.o_base = cd.o_base; j.o_rs = cd.o_rs; j.o_cs = cd.o_cs;
                j.n_taps = cd.nchunks / (p.kch / 32);
                j.ksplit = ks;
                j.tap_nw = tg.nw;
                j.tap_nw_magic = (unsigned)(((1ULL << 31) + (unsigned)tg.nw - 1) / (unsigned)tg.nw);
                j.a0 = tg.a0; j.a_u = tg.a_u; j.a_v = tg.a_v;
                j.w0 = tg.w0; j.w_u = tg.w_u; j.w_v = tg.w_v;
                out.push_back(j);
            }
    }
    (void)kc8;
    return out;
}

std::vector<FragJob> build_frag_tiles(const BatchedPlan& p, int n_rows, int n_wgs, std::vector<int>* begin) {
    constexpr int TN = 4;
    const int nblk = (n_rows + 31) / 32;
    std::vector<FragJob> tiles;
    std::vector<double> cost;
    for (size_t c = 0; c < p.cls.size(); ++c) {
        const ClassDesc& cd = p.cls[c];
        TapGrid tg;
        if (!frag_tap_grid(p, (int)c, &tg)) return {};
        const long long mblks = (long long)nblk * cd.pos_count;
        for (long long m0 = 0; m0 < mblks; m0 += TN)
            for (int cb = 0; cb < p.ncols / 32; cb += 2) {
                FragJob j = {};
                j.mblk0 = (int)m0;
                j.n_mblk = (int)std::min<long long>(TN, mblks - m0);
                j.cb0 = cb;
                j.s = cd.pos_count;
                j.s_magic = cd.magic;
                j.wc = cd.wc; j.wc_magic = cd.wc_magic;
                j.a_base = cd.a_base; j.a_rs = cd.a_rs; j.a_cs = cd.a_cs;
                j.o_base = cd.o_base; j.o_rs = cd.o_rs; j.o_cs = cd.o_cs;
                j.n_taps = cd.nchunks / (p.kch / 32);
                j.ksplit = 1;
                j.tap_nw = tg.nw;
                j.tap_nw_magic = (unsigned)(((1ULL << 31) + (unsigned)tg.nw - 1) / (unsigned)tg.nw);
                j.a0 = tg.a0; j.a_u = tg.a_u; j.a_v = tg.a_v;
                j.w0 = tg.w0; j.w_u = tg.w_u; j.w_v = tg.w_v;
                tiles.push_back(j);
                cost.push_back((double)cd.nchunks + 1.5);              // + start-up and epilogue, in K chunks (measured ~4 us / 2.7 us per chunk)
            }
    }
    const int n_waves = n_wgs * 4, n_pairs = n_waves / 2;
    std::vector<size_t> order(tiles.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
    // pair k = (workgroup b, wave w) and (workgroup b + n_wgs / 2, wave w), k = 4 b + w
    typedef std::pair<double, int> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int k = 0; k < n_pairs; ++k) heap.push(Load(0.0, k));
    std::vector<std::vector<size_t>> per_wave((size_t)n_waves);
    std::vector<double> wl((size_t)n_waves, 0.0);
    for (size_t i : order) {
        Load l = heap.top();
        heap.pop();
        const int w0 = l.second, w1 = l.second + n_pairs;           // the pair's two waves
        const int w = wl[(size_t)w0] <= wl[(size_t)w1] ? w0 : w1;
        per_wave[(size_t)w].push_back(i);
        wl[(size_t)w] += cost[i];
        heap.push(Load(l.first + cost[i], l.second));
    }
    std::vector<FragJob> out;
    begin->assign((size_t)n_waves + 1, 0);
    for (int w = 0; w < n_waves; ++w) {
        (*begin)[(size_t)w] = (int)out.size();
        for (size_t i : per_wave[(size_t)w]) out.push_back(tiles[i]);
    }
    (*begin)[(size_t)n_waves] = (int)out.size();
    return out;
}

}  // namespace dg
