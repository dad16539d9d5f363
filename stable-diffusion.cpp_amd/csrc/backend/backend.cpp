// backend.cpp — libggml-mi355x.so: a ggml backend plug-in for AMD Instinct MI355X (gfx950 / CDNA4).
//
// This file is the DROP-IN BOUNDARY of the repo: it implements the ggml backend vtables declared in
// include/ggml-abi.h (registry -> device -> buffer type -> buffer -> backend/stream) and exports the
// dynamic-loading entry points `ggml_backend_init` / `ggml_backend_score` that a GGML_BACKEND_DL host finds
// via ggml_backend_load_all() (reference: src/core/ggml_extend_backend.cpp:302-320).  The host-visible
// contract per vtable slot is documented in include/ggml-mi355x.h; graph execution (fusion planner + plan
// cache) lives in planner.cpp, kernels in ../kernels/*.hip.
//
// Device naming: "MI355X<i>".  The name deliberately contains none of "ROCm"/"CUDA"/"Vulkan"/"SYCL": the
// reference string-matches device names to flip graph-build decisions and we want the
// DEFAULT graph (no force_prec_f32 special case — accumulation is always f32 here).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ggml-abi.h"
#include "ggml-abi-check.h"  // static_asserts: every layout fact of ggml's headers this backend was written against
#include <dlfcn.h>
#include <link.h>

#include "ggml-mi355x.h"
#include "ktime.h"
#include "planner.h"
#include "kernels.h"

namespace mi355x {

#define HIP_OK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) {                                                                               \
            fprintf(stderr, "[ggml-mi355x] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                                     \
    } while (0)

struct DeviceCtx {
    int id;
    std::string name, desc;
    ggml_backend_device dev;
    ggml_backend_buffer_type buft;
};
struct BufferCtx {
    int device;
    void* base;
};

static std::vector<DeviceCtx*> g_devices;
static ggml_backend_reg g_reg;
static ggml_guid g_guid = {{0x35, 0x5a, 0x0d, 0xcd, 0x4a, 0x09, 0x50, 0x11, 0xa1, 0xb2, 0xc3, 0xd4, 0xe5, 0xf6, 0x07, 0x18}};

// ---------------------------------------------------------------- buffers
static const char* buft_get_name(ggml_backend_buffer_type_t t) { return ((DeviceCtx*)t->context)->name.c_str(); }
static void buf_free(ggml_backend_buffer_t b) {
    BufferCtx* c = (BufferCtx*)b->context;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipDeviceSynchronize());
    planner_forget_range(c->base, b->size);
    HIP_OK(hipFree(c->base));
    delete c;
}
static void* buf_get_base(ggml_backend_buffer_t b) { return ((BufferCtx*)b->context)->base; }
static enum ggml_status buf_init_tensor(ggml_backend_buffer_t, ggml_tensor*) { return GGML_STATUS_SUCCESS; }
static void buf_memset(ggml_backend_buffer_t b, ggml_tensor* t, uint8_t v, size_t off, size_t sz) {
    HIP_OK(hipSetDevice(((BufferCtx*)b->context)->device));
    HIP_OK(hipMemset((char*)t->data + off, v, sz));
}
static void buf_set(ggml_backend_buffer_t b, ggml_tensor* t, const void* d, size_t off, size_t sz) {
    HIP_OK(hipSetDevice(((BufferCtx*)b->context)->device));
    if (b->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS) planner_forget_range((char*)t->data + off, sz);  // swizzled copies are stale now
    HIP_OK(hipMemcpy((char*)t->data + off, d, sz, hipMemcpyHostToDevice));
}
static void buf_get(ggml_backend_buffer_t b, const ggml_tensor* t, void* d, size_t off, size_t sz) {
    HIP_OK(hipSetDevice(((BufferCtx*)b->context)->device));
    HIP_OK(hipMemcpy(d, (const char*)t->data + off, sz, hipMemcpyDeviceToHost));
}
static bool buf_cpy(ggml_backend_buffer_t b, const ggml_tensor* src, ggml_tensor* dst) {
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    if (!sb || sb->iface.get_base != buf_get_base) return false;  // not one of ours
    if (!ggml_abi_is_contiguous(src) || !ggml_abi_is_contiguous(dst)) return false;
    HIP_OK(hipSetDevice(((BufferCtx*)b->context)->device));
    HIP_OK(hipMemcpy(dst->data, src->data, ggml_abi_nbytes(src), hipMemcpyDeviceToDevice));
    return true;
}
static void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    BufferCtx* c = (BufferCtx*)b->context;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemset(c->base, v, b->size));
}
static ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    DeviceCtx* d = (DeviceCtx*)t->context;
    HIP_OK(hipSetDevice(d->id));
    void* p = nullptr;
    if (hipMalloc(&p, size > 0 ? size : 256) != hipSuccess) {
        fprintf(stderr, "[ggml-mi355x] hipMalloc(%zu) failed\n", size);
        (void)hipGetLastError();
        return nullptr;  // host turns this into "alloc failed" (ggml_extend.hpp:2232-2236)
    }
    // GGML_MI355X_POISON=1 (debugging): every byte a kernel may read before anything wrote it is a NaN pattern — an uninitialised read shows up as NaN in the
    // result instead of depending on what the allocator handed back (round 6: the FLUX graphs' results depended on the buffer's previous contents)
    static const bool poison = getenv("GGML_MI355X_POISON") != nullptr;
    if (poison) (void)hipMemset(p, 0xFF, size > 0 ? size : 256);
    ggml_backend_buffer* b = new ggml_backend_buffer();
    memset(b, 0, sizeof(*b));
    b->iface.free_buffer   = buf_free;
    b->iface.get_base      = buf_get_base;
    b->iface.init_tensor   = buf_init_tensor;
    b->iface.memset_tensor = buf_memset;
    b->iface.set_tensor    = buf_set;
    b->iface.get_tensor    = buf_get;
    b->iface.cpy_tensor    = buf_cpy;
    b->iface.clear         = buf_clear;
    b->buft                = t;
    b->context             = new BufferCtx{d->id, p};
    b->size                = size;
    b->usage               = GGML_BACKEND_BUFFER_USAGE_ANY;
    return b;
}
static size_t buft_alignment(ggml_backend_buffer_type_t) { return 256; }
static size_t buft_max_size(ggml_backend_buffer_type_t t) {
    size_t fr = 0, tot = 0;
    HIP_OK(hipSetDevice(((DeviceCtx*)t->context)->id));
    HIP_OK(hipMemGetInfo(&fr, &tot));
    return tot;
}
static size_t buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor* t) { return ggml_abi_nbytes(t); }
static bool buft_is_host(ggml_backend_buffer_type_t) { return false; }

// ---------------------------------------------------------------- backend (one HIP stream)
// Pinned upload staging (option "pinned_uploads", default 0 until timed on hardware).  hipMemcpyAsync from PAGEABLE host memory is
// host-synchronous in ROCm: the call returns only when the copy — ordered behind everything already queued on the stream — has
// finished, so a host that uploads the next step's inputs stalls until the GPU has drained the previous graph.  With the option on,
// set_tensor_async copies the caller's bytes into a pinned double buffer and queues the DMA from there: the call returns at once, the
// caller may reuse its buffer, and a half is only rewritten after the event recorded behind its last DMA has completed.
struct UploadStaging {
    static constexpr size_t HALF = 16u << 20;
    char* base        = nullptr;  // 2 * HALF bytes of hipHostMalloc memory
    size_t used       = 0;        // bytes handed out in the current half
    int cur           = 0;
    hipEvent_t done[2] = {nullptr, nullptr};
    bool pending[2]   = {false, false};
};
static std::atomic<int> g_pinned_uploads{0};

struct BackendCtx {
    DeviceCtx* dev;
    hipStream_t stream;
    Planner* planner;
    UploadStaging up;
};
static const char* be_get_name(ggml_backend_t b) { return ((BackendCtx*)b->context)->dev->name.c_str(); }
static void be_free(ggml_backend_t b) {
    BackendCtx* c = (BackendCtx*)b->context;
    HIP_OK(hipSetDevice(c->dev->id));
    HIP_OK(hipStreamSynchronize(c->stream));
    planner_destroy(c->planner);
    if (c->up.base) {
        (void)hipEventDestroy(c->up.done[0]);
        (void)hipEventDestroy(c->up.done[1]);
        (void)hipHostFree(c->up.base);
    }
    HIP_OK(hipStreamDestroy(c->stream));
    delete c;
    delete b;
}
static void be_set_async(ggml_backend_t b, ggml_tensor* t, const void* d, size_t off, size_t sz) {
    BackendCtx* c = (BackendCtx*)b->context;
    HIP_OK(hipSetDevice(c->dev->id));
    UploadStaging& u   = c->up;
    const size_t need  = (sz + 255) & ~(size_t)255;
    if (g_pinned_uploads.load() && sz > 0 && need <= UploadStaging::HALF) {
        if (!u.base) {
            if (hipHostMalloc((void**)&u.base, 2 * UploadStaging::HALF, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                u.base = nullptr;
            } else {
                HIP_OK(hipEventCreateWithFlags(&u.done[0], hipEventDisableTiming));
                HIP_OK(hipEventCreateWithFlags(&u.done[1], hipEventDisableTiming));
            }
        }
        if (u.base) {
            if (u.used + need > UploadStaging::HALF) {
                // this half is full: mark the end of its DMAs and move to the other half once ITS DMAs have completed
                HIP_OK(hipEventRecord(u.done[u.cur], c->stream));
                u.pending[u.cur] = true;
                u.cur ^= 1;
                u.used = 0;
                if (u.pending[u.cur]) {
                    HIP_OK(hipEventSynchronize(u.done[u.cur]));
                    u.pending[u.cur] = false;
                }
            }
            char* slot = u.base + (size_t)u.cur * UploadStaging::HALF + u.used;
            memcpy(slot, d, sz);
            u.used += need;
            HIP_OK(hipMemcpyAsync((char*)t->data + off, slot, sz, hipMemcpyHostToDevice, c->stream));
            return;
        }
    }
    HIP_OK(hipMemcpyAsync((char*)t->data + off, d, sz, hipMemcpyHostToDevice, c->stream));
}
static void be_get_async(ggml_backend_t b, const ggml_tensor* t, void* d, size_t off, size_t sz) {
    BackendCtx* c = (BackendCtx*)b->context;
    HIP_OK(hipSetDevice(c->dev->id));
    HIP_OK(hipMemcpyAsync(d, (const char*)t->data + off, sz, hipMemcpyDeviceToHost, c->stream));
}
static void be_synchronize(ggml_backend_t b) {
    BackendCtx* c = (BackendCtx*)b->context;
    HIP_OK(hipSetDevice(c->dev->id));
    HIP_OK(hipStreamSynchronize(c->stream));
}
static enum ggml_status be_graph_compute(ggml_backend_t b, ggml_cgraph* g) {
    BackendCtx* c = (BackendCtx*)b->context;
    if (hipSetDevice(c->dev->id) != hipSuccess) return GGML_STATUS_FAILED;
    return planner_compute(c->planner, g, c->stream);
}

// ---------------------------------------------------------------- device
static const char* dev_get_name(ggml_backend_dev_t d) { return ((DeviceCtx*)d->context)->name.c_str(); }
static const char* dev_get_desc(ggml_backend_dev_t d) { return ((DeviceCtx*)d->context)->desc.c_str(); }
static void dev_get_memory(ggml_backend_dev_t d, size_t* fr, size_t* tot) {
    HIP_OK(hipSetDevice(((DeviceCtx*)d->context)->id));
    HIP_OK(hipMemGetInfo(fr, tot));
}
static enum ggml_backend_dev_type dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void dev_get_props(ggml_backend_dev_t d, ggml_backend_dev_props* p) {
    memset(p, 0, sizeof(*p));
    p->name        = dev_get_name(d);
    p->description = dev_get_desc(d);
    dev_get_memory(d, &p->memory_free, &p->memory_total);
    p->type                      = GGML_BACKEND_DEVICE_TYPE_GPU;
    p->caps.async                = true;
    p->caps.host_buffer          = false;
    p->caps.buffer_from_host_ptr = false;  // probed at ggml_extend_backend.cpp:759-761
    p->caps.events               = false;
}
static ggml_backend_t dev_init_backend(ggml_backend_dev_t d, const char*) {
    DeviceCtx* dc = (DeviceCtx*)d->context;
    if (hipSetDevice(dc->id) != hipSuccess) return nullptr;
    BackendCtx* c = new BackendCtx();
    c->dev        = dc;
    if (hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return nullptr;
    }
    c->planner       = planner_create(dc->id);
    ggml_backend* b  = new ggml_backend();
    memset(b, 0, sizeof(*b));
    b->guid                   = &g_guid;
    b->iface.get_name         = be_get_name;
    b->iface.free             = be_free;
    b->iface.set_tensor_async = be_set_async;
    b->iface.get_tensor_async = be_get_async;
    b->iface.synchronize      = be_synchronize;
    b->iface.graph_compute    = be_graph_compute;
    b->device                 = d;
    b->context                = c;
    return b;
}
static ggml_backend_buffer_type_t dev_get_buft(ggml_backend_dev_t d) { return &((DeviceCtx*)d->context)->buft; }
static bool dev_supports_op(ggml_backend_dev_t, const ggml_tensor* op) { return planner_supports_op(op); }
static bool dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t t) { return t == &((DeviceCtx*)d->context)->buft; }
static bool dev_offload_op(ggml_backend_dev_t, const ggml_tensor*) { return false; }

// ---------------------------------------------------------------- registry
static const char* reg_get_name(ggml_backend_reg_t) { return "MI355X"; }
static size_t reg_dev_count(ggml_backend_reg_t) { return g_devices.size(); }
static ggml_backend_dev_t reg_get_dev(ggml_backend_reg_t, size_t i) { return i < g_devices.size() ? &g_devices[i]->dev : nullptr; }
static void* reg_get_proc(ggml_backend_reg_t, const char* name) {
    // optional proc addresses the host probes: split buffer type, set_n_threads, get_features -> absent
    // (ggml_extend_backend.cpp:801-804, 440-446, 519-528).  Ours: planner statistics for tests / bench.
    if (strcmp(name, "ggml_backend_mi355x_get_stats") == 0) return (void*)ggml_backend_mi355x_get_stats;
    if (strcmp(name, "ggml_backend_mi355x_set_option") == 0) return (void*)ggml_backend_mi355x_set_option;
    if (strcmp(name, "ggml_backend_mi355x_kernel_timing_enable") == 0) return (void*)ggml_backend_mi355x_kernel_timing_enable;
    if (strcmp(name, "ggml_backend_mi355x_get_kernel_timing") == 0) return (void*)ggml_backend_mi355x_get_kernel_timing;
    if (strcmp(name, "ggml_backend_mi355x_get_kernel_timings") == 0) return (void*)ggml_backend_mi355x_get_kernel_timings;
    if (strcmp(name, "ggml_backend_mi355x_kernel_timing_enable_mask") == 0) return (void*)ggml_backend_mi355x_kernel_timing_enable_mask;
    if (strcmp(name, "ggml_backend_mi355x_get_stream") == 0) return (void*)ggml_backend_mi355x_get_stream;
    if (strcmp(name, "ggml_backend_mi355x_hip_library") == 0) return (void*)ggml_backend_mi355x_hip_library;
    if (strcmp(name, "ggml_backend_mi355x_set_device") == 0) return (void*)ggml_backend_mi355x_set_device;
    if (strcmp(name, "ggml_backend_mi355x_get_device") == 0) return (void*)ggml_backend_mi355x_get_device;
    return nullptr;
}

// ---------------------------------------------------------------- host enum numbering, resolved by name (VERDICT r4 task 4)
// The numeric values of ggml_op / ggml_unary_op in include/ggml-abi.h are this repo's recollection of upstream; the reference links a FORK of ggml that
// adds ops (src/core/ggml_extend.hpp:1059,1088,3492).  Every libggml-base exports ggml_op_name / ggml_unary_op_name / ggml_type_name: at init the
// translation tables of planner.cpp are rebuilt from the HOST's names, so a fork that inserted ops mid-enum dispatches the right kernels; a host that
// lacks a name the planner needs, or gives one name to two numbers, or numbers a tensor type differently, gets ZERO devices and a log line saying why.
typedef const char* (*name_of_fn)(int);
static std::string g_enum_status = "not resolved (ggml_backend_init not called)";
static bool g_enum_ok            = true;
static std::mutex g_enum_mu;
static bool g_enum_explicit = false;  // the host called ggml_backend_mi355x_resolve_enums() itself

static void* find_host_symbol(const char* sym) {
    if (void* p = dlsym(RTLD_DEFAULT, sym)) return p;
    // the host library may have been dlopen()ed RTLD_LOCAL (language bindings): walk the loaded objects
    struct Ctx {
        const char* sym;
        void* found;
    } ctx{sym, nullptr};
    dl_iterate_phdr(
        [](struct dl_phdr_info* info, size_t, void* data) -> int {
            Ctx* c = (Ctx*)data;
            if (!info->dlpi_name || !info->dlpi_name[0]) return 0;
            if (void* h = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_LAZY)) {
                void* p = dlsym(h, c->sym);
                dlclose(h);
                if (p) {
                    c->found = p;
                    return 1;
                }
            }
            return 0;
        },
        &ctx);
    return ctx.found;
}

static bool resolve_host_enums(name_of_fn opn, name_of_fn unn, name_of_fn tyn, const char* how) {
    char err[256] = {0};
    if (!opn && !unn && !tyn) {
        g_enum_status = "host exports no ggml_op_name / ggml_unary_op_name / ggml_type_name: enum numbering of include/ggml-abi.h assumed";
        g_enum_ok     = true;
        return true;
    }
    g_enum_ok = planner_build_op_maps(opn, unn, tyn, err, sizeof(err));
    if (g_enum_ok)
        g_enum_status = std::string("op / unary-op numbers translated by name, type numbers verified (") + how + ")";
    else
        g_enum_status = std::string("ENUM MISMATCH (") + how + "): " + err;
    return g_enum_ok;
}

static void init_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) {
            (void)hipGetLastError();
            n = 0;
        }
        std::lock_guard<std::mutex> elk(g_enum_mu);
        if (!g_enum_explicit &&
            !resolve_host_enums((name_of_fn)find_host_symbol("ggml_op_name"), (name_of_fn)find_host_symbol("ggml_unary_op_name"),
                                (name_of_fn)find_host_symbol("ggml_type_name"), "host symbols found at ggml_backend_init")) {
            fprintf(stderr, "[ggml-mi355x] %s -- registering NO devices (a wrong op table would dispatch wrong kernels silently)\n", g_enum_status.c_str());
            n = 0;
        }
        for (int i = 0; i < n; ++i) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
            // gfx950 only: this library contains no code object for anything else
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
                fprintf(stderr, "[ggml-mi355x] skipping device %d (%s): not gfx950\n", i, prop.gcnArchName);
                continue;
            }
            DeviceCtx* d = new DeviceCtx();
            d->id        = i;
            d->name      = "MI355X" + std::to_string((int)g_devices.size());
            d->desc      = std::string(prop.name) + " (" + prop.gcnArchName + ", " + std::to_string(prop.multiProcessorCount) + " CUs)";
            memset(&d->dev, 0, sizeof(d->dev));
            d->dev.iface.get_name        = dev_get_name;
            d->dev.iface.get_description = dev_get_desc;
            d->dev.iface.get_memory      = dev_get_memory;
            d->dev.iface.get_type        = dev_get_type;
            d->dev.iface.get_props       = dev_get_props;
            d->dev.iface.init_backend    = dev_init_backend;
            d->dev.iface.get_buffer_type = dev_get_buft;
            d->dev.iface.supports_op     = dev_supports_op;
            d->dev.iface.supports_buft   = dev_supports_buft;
            d->dev.iface.offload_op      = dev_offload_op;
            d->dev.reg                   = &g_reg;
            d->dev.context               = d;
            memset(&d->buft, 0, sizeof(d->buft));
            d->buft.iface.get_name       = buft_get_name;
            d->buft.iface.alloc_buffer   = buft_alloc;
            d->buft.iface.get_alignment  = buft_alignment;
            d->buft.iface.get_max_size   = buft_max_size;
            d->buft.iface.get_alloc_size = buft_alloc_size;
            d->buft.iface.is_host        = buft_is_host;
            d->buft.device               = &d->dev;
            d->buft.context              = d;
            g_devices.push_back(d);
        }
        memset(&g_reg, 0, sizeof(g_reg));
        g_reg.api_version            = GGML_BACKEND_API_VERSION;
        g_reg.iface.get_name         = reg_get_name;
        g_reg.iface.get_device_count = reg_dev_count;
        g_reg.iface.get_device       = reg_get_dev;
        g_reg.iface.get_proc_address = reg_get_proc;
    });
}

}  // namespace mi355x

extern "C" {
GGML_MI355X_API ggml_backend_reg_t ggml_backend_mi355x_reg(void) {
    mi355x::init_once();
    return &mi355x::g_reg;
}
GGML_MI355X_API ggml_backend_reg_t ggml_backend_init(void) { return ggml_backend_mi355x_reg(); }
GGML_MI355X_API int ggml_backend_score(void) {
    mi355x::init_once();
    return mi355x::g_devices.empty() ? 0 : 100;
}
GGML_MI355X_API int ggml_backend_mi355x_get_device_count(void) {
    mi355x::init_once();
    return (int)mi355x::g_devices.size();
}
GGML_MI355X_API void ggml_backend_mi355x_get_stats(struct ggml_backend_mi355x_stats* out) { mi355x::planner_get_stats(out); }
GGML_MI355X_API int ggml_backend_mi355x_calibrate(struct ggml_backend_mi355x_calibration* out) {
    mi355x::CalibrationResult r;
    if (!out || !mi355x::calibrate_device(nullptr, &r)) return -1;
    out->mfma_f16_tflops = r.mfma_f16_tflops;
    out->mfma_clock_mhz  = r.mfma_clock_mhz;
    out->copy_tbs        = r.copy_tbs;
    out->read_tbs        = r.read_tbs;
    out->compute_units   = r.compute_units;
    return 0;
}
// explicit form of the by-name enum resolution (a host that does not export the name functions globally, and the ABI tests): returns 0 and installs
// the translation, or -1 (tables unchanged; the reason is in ggml_backend_mi355x_enum_status()).  NULL arguments leave that table alone.
GGML_MI355X_API int ggml_backend_mi355x_resolve_enums(const char* (*op_name)(int), const char* (*unary_op_name)(int), const char* (*type_name)(int)) {
    std::lock_guard<std::mutex> lk(mi355x::g_enum_mu);
    const bool was_ok = mi355x::g_enum_ok;
    const std::string was = mi355x::g_enum_status;
    const bool ok = mi355x::resolve_host_enums(op_name, unary_op_name, type_name, "ggml_backend_mi355x_resolve_enums");
    if (ok) {
        mi355x::g_enum_explicit = true;
    } else {  // a failed explicit call changes nothing but the status text it returns through enum_status()
        mi355x::g_enum_ok     = was_ok;
        mi355x::g_enum_status = mi355x::g_enum_status + " [previous state kept: " + was + "]";
    }
    return ok ? 0 : -1;
}
GGML_MI355X_API const char* ggml_backend_mi355x_enum_status(void) {
    std::lock_guard<std::mutex> lk(mi355x::g_enum_mu);
    static thread_local std::string copy;
    copy = mi355x::g_enum_status;
    return copy.c_str();
}
GGML_MI355X_API void ggml_backend_mi355x_get_enum_maps(uint8_t* ops256, uint8_t* unary256) { mi355x::planner_get_op_maps(ops256, unary256); }
GGML_MI355X_API void ggml_backend_mi355x_set_option(const char* key, int value) {
    if (strcmp(key, "pinned_uploads") == 0) {
        mi355x::g_pinned_uploads.store(value);
        return;
    }
    mi355x::planner_set_option(key, value);
}
static void fill_timing(struct ggml_backend_mi355x_kernel_timing* o, const mi355x::KFamTiming& t, int family) {
    memset(o, 0, sizeof(*o));
    snprintf(o->kernel, sizeof(o->kernel), "%s", t.name);
    o->launches    = t.launches;
    o->total_ms    = t.total_ms;
    o->total_flops = t.total_flops;
    o->total_bytes = t.total_bytes;
    o->bound       = t.bound;
    o->family      = family;
}
GGML_MI355X_API void ggml_backend_mi355x_kernel_timing_enable(int enable) { mi355x::ktime_enable(enable ? 1u : 0u); }
GGML_MI355X_API void ggml_backend_mi355x_kernel_timing_enable_mask(uint32_t family_mask) { mi355x::ktime_enable(family_mask); }
GGML_MI355X_API void* ggml_backend_mi355x_get_stream(ggml_backend_t backend) {
    return backend ? (void*)((mi355x::BackendCtx*)backend->context)->stream : nullptr;
}
// The HIP runtime THIS plug-in is bound to (a process may hold a second copy, e.g. the one a torch wheel bundles): a companion library that
// must share streams with the backend — RCCL for the native CFG-pair exchange — is loaded from the same directory.
GGML_MI355X_API const char* ggml_backend_mi355x_hip_library(void) {
    static std::string path;
    if (path.empty()) {
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) path = info.dli_fname;
    }
    return path.c_str();
}
GGML_MI355X_API int ggml_backend_mi355x_set_device(int hip_device) { return hipSetDevice(hip_device) == hipSuccess ? 0 : -1; }
GGML_MI355X_API int ggml_backend_mi355x_get_device(void) {
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}
GGML_MI355X_API int ggml_backend_mi355x_get_kernel_timings(struct ggml_backend_mi355x_kernel_timing* out, int capacity) {
    mi355x::KFamTiming t[mi355x::KF_COUNT];
    int fam[mi355x::KF_COUNT];
    const int n = mi355x::ktime_read(t, mi355x::KF_COUNT, fam);
    int m       = 0;
    for (int i = 0; i < n && m < capacity; ++i) fill_timing(&out[m++], t[i], fam[i]);
    return m;
}
GGML_MI355X_API void ggml_backend_mi355x_get_kernel_timing(struct ggml_backend_mi355x_kernel_timing* out) {
    struct ggml_backend_mi355x_kernel_timing all[mi355x::KF_COUNT];
    const int n = ggml_backend_mi355x_get_kernel_timings(all, mi355x::KF_COUNT);
    if (n > 0) {
        *out = all[0];
    } else {
        memset(out, 0, sizeof(*out));
        snprintf(out->kernel, sizeof(out->kernel), "(no timed launches)");
    }
}
}
