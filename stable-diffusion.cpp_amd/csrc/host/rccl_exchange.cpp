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
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
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
typedef int (*fn_hip_get_device)(void);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;  // nccl.h enum values (ncclFloat32, ncclSum)

struct Api {
    void* h = nullptr;
    fn_get_unique_id get_unique_id   = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce         = nullptr;
    fn_comm_destroy comm_destroy     = nullptr;
    fn_error_string error_string     = nullptr;
    fn_hip_set_device hip_set_device = nullptr;
    fn_hip_get_device hip_get_device = nullptr;
    bool ok                          = false;
    bool failed                      = false;  // a load attempt found no usable library: not repeated (the answer cannot change inside one process)
};
std::mutex g_mu;
Api g_api;
// the last error of ANY thread (a host typically creates the communicator on one thread and asks for the error on another); sd_rccl_last_error
// hands out a per-thread copy so the pointer it returns stays valid while other threads fail
std::mutex g_err_mu;
std::string g_err_text;
struct ErrSink {
    ErrSink& operator=(const std::string& s) {
        std::lock_guard<std::mutex> lk(g_err_mu);
        g_err_text = s;
        return *this;
    }
} g_err;
// communicators created here and not yet destroyed: the exchange callback refuses a communicator that is no longer in this set, so a context that
// still has a destroyed communicator installed fails its next step with an error instead of calling into freed RCCL state
std::set<void*> g_live;
std::map<void*, int> g_inflight;  // communicator -> all-reduce calls currently enqueueing on it (destroy waits for zero)
std::condition_variable g_idle;

bool load_api() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_api.ok) return true;
    if (g_api.failed) return false;
    // RCCL must sit on the SAME HIP runtime as the backend plug-in (its streams are handed to ncclAllReduce); a process may hold a second
    // runtime + RCCL pair (the copies a torch wheel bundles).  The plug-in tells which runtime it is bound to; RCCL is taken from that directory.
    typedef const char* (*fn_hip_library)(void);
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    fn_hip_library hl      = reg ? (fn_hip_library)ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_hip_library") : nullptr;
    g_api.hip_set_device   = reg ? (fn_hip_set_device)ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_set_device") : nullptr;
    g_api.hip_get_device   = reg ? (fn_hip_get_device)ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_get_device") : nullptr;
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
        g_api.failed  = true;
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
        dlclose(h);
        g_api        = Api{};
        g_api.failed = true;
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
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_live.count(user)) {
            g_err = "native pair exchange: the installed communicator was destroyed (sd_set_pair_exchange_rccl(ctx, NULL, 0) before sd_rccl_comm_destroy)";
            return false;
        }
        ++g_inflight[user];  // held across the enqueue: sd_rccl_comm_destroy waits for it (round-4 advice: the check alone left a window)
    }
    const int r = g_api.all_reduce(device_eps, device_eps, (size_t)count, NCCL_FLOAT32, NCCL_SUM, (NcclComm)user, stream);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (--g_inflight[user] == 0) {
            g_inflight.erase(user);
            g_idle.notify_all();
        }
    }
    if (r != 0) {
        g_err = "ncclAllReduce: " + nccl_err(r);
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

const char* sd_rccl_last_error(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_err_text;
    return copy.c_str();
}

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
    // the communicator binds to the calling thread's current device: switch for the call only, the caller's device is put back
    const int prev = (device >= 0 && g_api.hip_get_device) ? g_api.hip_get_device() : -1;
    if (device >= 0 && g_api.hip_set_device && g_api.hip_set_device(device) != 0) {
        g_err = "hipSetDevice(" + std::to_string(device) + ") failed: the communicator would bind to the wrong device";
        return nullptr;
    }
    NcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    NcclComm comm = nullptr;
    const int r   = g_api.comm_init_rank(&comm, nranks, id, rank);
    if (prev >= 0 && prev != device && g_api.hip_set_device) (void)g_api.hip_set_device(prev);
    if (r != 0) {
        g_err = "ncclCommInitRank: " + nccl_err(r);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live.insert(comm);
    return comm;
}

// Order of teardown: remove the exchange from every context that uses the communicator (sd_set_pair_exchange_rccl(ctx, NULL, 0)), then destroy it.
// A context that still has it installed does not crash afterwards — its next sampler step fails with an error (rccl_exchange checks g_live).
void sd_rccl_comm_destroy(void* comm) {
    if (!comm) return;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        if (!g_api.ok || !g_live.erase(comm)) return;  // not one of ours (or already destroyed): nothing to do; new exchanges are refused from here on
        g_idle.wait(lk, [&] { return g_inflight.find(comm) == g_inflight.end(); });  // an all-reduce being enqueued on another thread finishes first
    }
    (void)g_api.comm_destroy((NcclComm)comm);
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
