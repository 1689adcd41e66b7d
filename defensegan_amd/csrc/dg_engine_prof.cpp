// Per-launch event profile of the engine (dg_engine.h ProfScope): collection and the dg_profile_* entry points.
#include "dg_engine.h"

#pragma GCC visibility push(hidden)
namespace dge {

int prof_slot(dg_handle* h, const std::string& name) {
    auto it = h->prof_index.find(name);
    if (it != h->prof_index.end()) return it->second;
    h->prof.push_back(ProfEntry{name, 0, 0.0, 0.0});
    h->prof_index[name] = (int)h->prof.size() - 1;
    return (int)h->prof.size() - 1;
}

void prof_collect(dg_handle* h) {
    for (auto& p : h->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(h->prof_events[p.e1]) == hipSuccess &&
            hipEventElapsedTime(&ms, h->prof_events[p.e0], h->prof_events[p.e1]) == hipSuccess) {
            h->prof[p.entry].ms += ms;
            h->prof[p.entry].launches += 1;
        }
    }
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    h->prof_events.clear();
    h->pending.clear();
    h->prof_chain_last = -1;
}

}  // namespace dge
#pragma GCC visibility pop

extern "C" {

int dg_profile_enable(dg_handle* h, int on) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    h->prof_stride = on > 0 ? on : 0;
    return DG_OK;
}
int dg_profile_count(dg_handle* h) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    prof_collect(h);
    return (int)h->prof.size();
}
int dg_profile_read(dg_handle* h, int i, char* name, int name_len, int64_t* launches, double* total_ms, double* flops) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    prof_collect(h);
    if (i < 0 || i >= (int)h->prof.size()) return fail(DG_E_INVALID, "profile index out of range");
    const ProfEntry& p = h->prof[i];
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.name.c_str());
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = p.ms;
    if (flops) *flops = p.flops;
    return DG_OK;
}
int dg_profile_reset(dg_handle* h) {
    if (!h) return fail(DG_E_INVALID, "null handle");
    prof_collect(h);
    h->prof.clear();
    h->prof_index.clear();
    return DG_OK;
}

}  // extern "C"
