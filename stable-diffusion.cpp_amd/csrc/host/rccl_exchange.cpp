// rccl_exchange.cpp — the CFG-pair reduction as a NATIVE collective: one in-place ncclAllReduce (SUM, f32) of the engine's eps buffer per sampler
// step, enqueued on the backend's HIP stream.  RCCL is loaded at run time with dlopen from the directory of the HIP runtime the backend plug-in
// is bound to (librccl.so ships with ROCm; the host library has no link-time dependency on it and none on torch).  This is the C++ counterpart of shard.make_pair_exchange (which goes through torch.distributed)
// — north_star keeps the host C++.
//
// What is exchanged (src/runtime/guidance.cpp:149-179: guided = uncond + s * (cond - uncond)): the cond rank holds s * eps_cond, the uncond rank
// (1 - s) * eps_uncond, so the SUM over the two ranks of a pair is the guided prediction; both ranks then take the same Euler(-A) update.
// One [N, C, H, W] f32 buffer per step (64 KB per SD1.5 image, 1 MB per DiT image) over a single xGMI link.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "ggml.h"
#include "sd-mi355x.h"

namespace {

struct NcclUniqueId {  // ncclUniqueId (nccl.h): 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char internal[128];
};
typedef void* NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, NcclComm, void* /*hipStream_t*/);
typedef int (*fn_comm_destroy)(NcclComm);
typedef const char* (*fn_error_string)(int);
typedef int (*fn_hip_set_device)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;  // nccl.h enum values (ncclFloat32, ncclSum)

struct Api {
    void* h = nullptr;
    fn_get_unique_id get_unique_id   = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce         = nullptr;
    fn_comm_destroy comm_destroy     = nullptr;
    fn_error_string error_string     = nullptr;
    fn_hip_set_device hip_set_device = nullptr;
    bool ok                          = false;
};
std::mutex g_mu;
Api g_api;
thread_local std::string g_err;

bool load_api() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_api.ok) return true;
    // RCCL must sit on the SAME HIP runtime as the backend plug-in (its streams are handed to ncclAllReduce); a process may hold a second
    // runtime + RCCL pair (the copies a torch wheel bundles).  The plug-in tells which runtime it is bound to; RCCL is taken from that directory.
    typedef const char* (*fn_hip_library)(void);
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    fn_hip_library hl      = reg ? (fn_hip_library)ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_hip_library") : nullptr;
    g_api.hip_set_device   = reg ? (fn_hip_set_device)ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_set_device") : nullptr;
    if (!hl) {
        g_err = "the MI355X backend plug-in is not loaded (sd_load_backend first)";
        return false;
    }
    std::string dir = hl();
    const size_t sl = dir.rfind('/');
    dir             = sl == std::string::npos ? std::string() : dir.substr(0, sl + 1);
    void* h = nullptr;
    for (const char* n : {"librccl.so.1", "librccl.so"})
        if ((h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!h) {
        const char* e = dlerror();
        g_err         = "librccl.so not found next to " + std::string(hl()) + ": " + (e ? e : "");
        return false;
    }
    g_api.h              = h;
    g_api.get_unique_id  = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_api.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_api.all_reduce     = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_api.comm_destroy   = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_api.error_string   = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!g_api.get_unique_id || !g_api.comm_init_rank || !g_api.all_reduce || !g_api.comm_destroy) {
        g_err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
        return false;
    }
    g_api.ok = true;
    return true;
}

std::string nccl_err(int r) { return g_api.error_string ? g_api.error_string(r) : ("ncclResult_t " + std::to_string(r)); }

// sd_pair_exchange_fn: user = the communicator
bool rccl_exchange(void* device_eps, int64_t count, void* stream, void* user) {
    if (!user || !stream || !g_api.ok) {
        g_err = "native pair exchange needs a communicator and the backend's HIP stream (host backends have none)";
        return false;
    }
    const int r = g_api.all_reduce(device_eps, device_eps, (size_t)count, NCCL_FLOAT32, NCCL_SUM, (NcclComm)user, stream);
    if (r != 0) {
        g_err = "ncclAllReduce: " + nccl_err(r);
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

const char* sd_rccl_last_error(void) { return g_err.c_str(); }

bool sd_rccl_get_unique_id(void* id128) {
    if (!id128 || !load_api()) return false;
    NcclUniqueId id;
    const int r = g_api.get_unique_id(&id);
    if (r != 0) {
        g_err = "ncclGetUniqueId: " + nccl_err(r);
        return false;
    }
    memcpy(id128, id.internal, sizeof(id.internal));
    return true;
}

void* sd_rccl_comm_create(int device, int nranks, int rank, const void* id128) {
    if (!id128 || !load_api()) return nullptr;
    if (device >= 0 && g_api.hip_set_device) (void)g_api.hip_set_device(device);  // the communicator binds to the calling thread's current device
    NcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    NcclComm comm = nullptr;
    const int r   = g_api.comm_init_rank(&comm, nranks, id, rank);
    if (r != 0) {
        g_err = "ncclCommInitRank: " + nccl_err(r);
        return nullptr;
    }
    return comm;
}

void sd_rccl_comm_destroy(void* comm) {
    if (comm && g_api.ok) (void)g_api.comm_destroy((NcclComm)comm);
}

bool sd_set_pair_exchange_rccl(sdm_ctx_t* ctx, void* comm, int branch) {
    if (!ctx) return false;
    if (!comm) {
        sd_set_pair_exchange(ctx, nullptr, nullptr, 0);
        return true;
    }
    if (!load_api()) return false;
    sd_set_pair_exchange(ctx, rccl_exchange, comm, branch);
    return true;
}

}  // extern "C"
