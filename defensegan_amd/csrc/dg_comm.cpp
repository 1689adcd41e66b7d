// The path's ONE collective behind the C ABI (include/defensegan_hip.h dg_comm_*): the all_gather of the evaluation message
// (the reference has no multi-GPU path, utils/gan_defense.py:113-179 runs one process).  RCCL over xGMI, reached
// WITHOUT torch.distributed: librccl.so is opened at first use (no link-time dependency: a single-GPU user never loads it) and
// four of its entry points are bound -- ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy.  The caller moves
// the 128-byte unique id from rank 0 to the other ranks by whatever it has (a file, MPI, an environment variable).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/defensegan_hip.h"

extern "C" __attribute__((visibility("hidden"))) void dg_set_error_message(const char* msg);   // dg_engine.cpp

namespace {

struct Rccl {
    void* lib = nullptr;
    int (*get_unique_id)(void*) = nullptr;
    int (*comm_init_rank)(void**, int, DgUniqueId, int) = nullptr;       // ncclUniqueId is a 128-byte struct passed by value
    int (*all_gather)(const void*, void*, size_t, int, void*, void*) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    const char* (*get_error_string)(int) = nullptr;
};

int fail(int code, const char* fmt, const char* a = "", int b = 0) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    dg_set_error_message(buf);
    return code;
}

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.get_unique_id = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.comm_init_rank = reinterpret_cast<int (*)(void**, int, DgUniqueId, int)>(dlsym(r.lib, "ncclCommInitRank"));
            r.all_gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, void*)>(dlsym(r.lib, "ncclAllGather"));
            r.comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
            r.get_error_string = reinterpret_cast<const char* (*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
        }
    }
    if (!r.lib || !r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) return nullptr;
    return &r;
}

int check(Rccl* r, int rc, const char* what) {
    if (rc == 0) return DG_OK;
    return fail(DG_E_HIP, "%s failed: RCCL error %d", what, rc);
}

}  // namespace

struct dg_comm {
    void* comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

extern "C" {

int dg_comm_unique_id(DgUniqueId* id) {
    if (!id) return fail(DG_E_INVALID, "null id");
    Rccl* r = rccl();
    if (!r) return fail(DG_E_STATE, "librccl.so could not be loaded%s", "");
    return check(r, r->get_unique_id(id), "ncclGetUniqueId");
}

int dg_comm_create(int nranks, const DgUniqueId* id, int rank, int device, dg_comm** out) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(DG_E_INVALID, "bad argument%s", "");
    *out = nullptr;
    Rccl* r = rccl();
    if (!r) return fail(DG_E_STATE, "librccl.so could not be loaded%s", "");
    if (hipSetDevice(device) != hipSuccess) return fail(DG_E_HIP, "hipSetDevice(%s%d) failed", "", device);
    dg_comm* c = new dg_comm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    const int rc = r->comm_init_rank(&c->comm, nranks, *id, rank);
    if (rc != 0) { delete c; return check(r, rc, "ncclCommInitRank"); }
    *out = c;
    return DG_OK;
}

int dg_comm_destroy(dg_comm* c) {
    if (!c) return DG_OK;
    Rccl* r = rccl();
    int rc = DG_OK;
    if (r && c->comm) rc = check(r, r->comm_destroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

int dg_gather_eval(dg_comm* c, const int32_t* send, int32_t* recv, int64_t count, void* stream) {
    if (!c || !send || !recv || count < 0) return fail(DG_E_INVALID, "bad argument%s", "");
    Rccl* r = rccl();
    if (!r) return fail(DG_E_STATE, "librccl.so could not be loaded%s", "");
    if (hipSetDevice(c->device) != hipSuccess) return fail(DG_E_HIP, "hipSetDevice(%s%d) failed", "", c->device);
    return check(r, r->all_gather(send, recv, (size_t)count, /*ncclInt32*/ 2, c->comm, stream), "ncclAllGather");
}

}  // extern "C"
