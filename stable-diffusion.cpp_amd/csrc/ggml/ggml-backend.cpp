// ggml-backend.cpp — backend registry (static + GGML_BACKEND_DL dlopen loading), thin wrappers over the
// plug-in vtables of include/ggml-abi.h, and the graph allocator (gallocr).
// Reference call sites this mirrors: src/core/ggml_extend_backend.cpp:302-320 (load_all / enumerate),
// :393-418 (init by name), :466-509 (graph compute); src/core/ggml_extend.hpp:2212-2245 (gallocr
// reserve/alloc), :2347-2435 (tensor_set/get); src/model_manager.cpp:735-750 (weights buffers).
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "ggml.h"

namespace {
struct registry {
    std::mutex mu;
    std::vector<ggml_backend_reg_t> regs;
    std::vector<ggml_backend_dev_t> devs;
    std::vector<void*> handles;
};
registry& reg() {
    static registry r;
    return r;
}
bool iequals(const char* a, const char* b) {
    for (; *a && *b; ++a, ++b)
        if (tolower((unsigned char)*a) != tolower((unsigned char)*b)) return false;
    return *a == *b;
}
}  // namespace

extern "C" {

// ---- registry ----------------------------------------------------------------------------------
void ggml_backend_register(ggml_backend_reg_t r) {
    if (!r) return;
    std::lock_guard<std::mutex> lk(reg().mu);
    for (auto* e : reg().regs)
        if (e == r) return;
    reg().regs.push_back(r);
    const size_t n = r->iface.get_device_count(r);
    for (size_t i = 0; i < n; ++i) reg().devs.push_back(r->iface.get_device(r, i));
}

ggml_backend_reg_t ggml_backend_load(const char* path) {
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "ggml_backend_load: dlopen(%s) failed: %s\n", path, dlerror());
        return nullptr;
    }
    auto init = (ggml_backend_init_t)dlsym(h, "ggml_backend_init");
    if (!init) {
        fprintf(stderr, "ggml_backend_load: %s does not export ggml_backend_init\n", path);
        dlclose(h);
        return nullptr;
    }
    ggml_backend_reg_t r = init();
    if (!r || r->api_version != GGML_BACKEND_API_VERSION) {
        fprintf(stderr, "ggml_backend_load: %s: incompatible backend API version\n", path);
        dlclose(h);
        return nullptr;
    }
    ggml_backend_register(r);
    std::lock_guard<std::mutex> lk(reg().mu);
    reg().handles.push_back(h);
    return r;
}

size_t ggml_backend_reg_count(void) { return reg().regs.size(); }
ggml_backend_reg_t ggml_backend_reg_get(size_t index) { return index < reg().regs.size() ? reg().regs[index] : nullptr; }
ggml_backend_reg_t ggml_backend_reg_by_name(const char* name) {
    for (auto* r : reg().regs)
        if (iequals(r->iface.get_name(r), name)) return r;
    return nullptr;
}
const char* ggml_backend_reg_name(ggml_backend_reg_t r) { return r->iface.get_name(r); }
size_t ggml_backend_reg_dev_count(ggml_backend_reg_t r) { return r->iface.get_device_count(r); }
ggml_backend_dev_t ggml_backend_reg_dev_get(ggml_backend_reg_t r, size_t index) { return r->iface.get_device(r, index); }
void* ggml_backend_reg_get_proc_address(ggml_backend_reg_t r, const char* name) {
    return r->iface.get_proc_address ? r->iface.get_proc_address(r, name) : nullptr;
}
size_t ggml_backend_dev_count(void) { return reg().devs.size(); }
ggml_backend_dev_t ggml_backend_dev_get(size_t index) { return index < reg().devs.size() ? reg().devs[index] : nullptr; }
ggml_backend_dev_t ggml_backend_dev_by_name(const char* name) {
    for (auto* d : reg().devs)
        if (iequals(d->iface.get_name(d), name)) return d;
    return nullptr;
}
ggml_backend_dev_t ggml_backend_dev_by_type(enum ggml_backend_dev_type type) {
    for (auto* d : reg().devs)
        if (d->iface.get_type(d) == type) return d;
    return nullptr;
}
const char* ggml_backend_dev_name(ggml_backend_dev_t d) { return d->iface.get_name(d); }
const char* ggml_backend_dev_description(ggml_backend_dev_t d) { return d->iface.get_description(d); }
void ggml_backend_dev_memory(ggml_backend_dev_t d, size_t* free, size_t* total) { d->iface.get_memory(d, free, total); }
enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t d) { return d->iface.get_type(d); }
void ggml_backend_dev_get_props(ggml_backend_dev_t d, struct ggml_backend_dev_props* props) { d->iface.get_props(d, props); }
ggml_backend_reg_t ggml_backend_dev_backend_reg(ggml_backend_dev_t d) { return d->reg; }
ggml_backend_t ggml_backend_dev_init(ggml_backend_dev_t d, const char* params) { return d->iface.init_backend(d, params); }
ggml_backend_buffer_type_t ggml_backend_dev_buffer_type(ggml_backend_dev_t d) { return d->iface.get_buffer_type(d); }
bool ggml_backend_dev_supports_op(ggml_backend_dev_t d, const ggml_tensor* op) { return d->iface.supports_op(d, op); }
bool ggml_backend_dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t buft) { return d->iface.supports_buft(d, buft); }
ggml_backend_t ggml_backend_init_by_name(const char* name, const char* params) {
    ggml_backend_dev_t d = ggml_backend_dev_by_name(name);
    return d ? ggml_backend_dev_init(d, params) : nullptr;
}

ggml_backend_t ggml_backend_init_by_type(enum ggml_backend_dev_type type, const char* params) {
    ggml_backend_dev_t d = ggml_backend_dev_by_type(type);
    return d ? ggml_backend_dev_init(d, params) : nullptr;
}
ggml_backend_t ggml_backend_init_best(void) {
    ggml_backend_dev_t d = ggml_backend_dev_by_type(GGML_BACKEND_DEVICE_TYPE_GPU);
    if (!d && !reg().devs.empty()) d = reg().devs.front();
    return d ? ggml_backend_dev_init(d, nullptr) : nullptr;
}
void ggml_backend_load_all(void) {}
ggml_backend_dev_t ggml_backend_buft_get_device(ggml_backend_buffer_type_t buft) { return buft->device; }
ggml_backend_buffer_type_t ggml_backend_dev_host_buffer_type(ggml_backend_dev_t d) { return d->iface.get_host_buffer_type ? d->iface.get_host_buffer_type(d) : nullptr; }

// ---- backend (stream) ----------------------------------------------------------------------------
const char* ggml_backend_name(ggml_backend_t b) { return b ? b->iface.get_name(b) : "NULL"; }
void ggml_backend_free(ggml_backend_t b) {
    if (b) b->iface.free(b);
}
ggml_backend_dev_t ggml_backend_get_device(ggml_backend_t b) { return b->device; }
ggml_backend_buffer_type_t ggml_backend_get_default_buffer_type(ggml_backend_t b) { return ggml_backend_dev_buffer_type(b->device); }
ggml_backend_buffer_t ggml_backend_alloc_buffer(ggml_backend_t b, size_t size) {
    return ggml_backend_buft_alloc_buffer(ggml_backend_get_default_buffer_type(b), size);
}
size_t ggml_backend_get_alignment(ggml_backend_t b) { return ggml_backend_buft_get_alignment(ggml_backend_get_default_buffer_type(b)); }
bool ggml_backend_supports_op(ggml_backend_t b, const ggml_tensor* op) { return ggml_backend_dev_supports_op(b->device, op); }
void ggml_backend_synchronize(ggml_backend_t b) {
    if (b->iface.synchronize) b->iface.synchronize(b);
}
enum ggml_status ggml_backend_graph_compute_async(ggml_backend_t b, ggml_cgraph* g) { return b->iface.graph_compute(b, g); }
enum ggml_status ggml_backend_graph_compute(ggml_backend_t b, ggml_cgraph* g) {
    enum ggml_status st = ggml_backend_graph_compute_async(b, g);
    ggml_backend_synchronize(b);
    return st;
}

// ---- buffer types / buffers ----------------------------------------------------------------------
const char* ggml_backend_buft_name(ggml_backend_buffer_type_t t) { return t->iface.get_name(t); }
ggml_backend_buffer_t ggml_backend_buft_alloc_buffer(ggml_backend_buffer_type_t t, size_t size) { return t->iface.alloc_buffer(t, size); }
size_t ggml_backend_buft_get_alignment(ggml_backend_buffer_type_t t) { return t->iface.get_alignment(t); }
size_t ggml_backend_buft_get_max_size(ggml_backend_buffer_type_t t) { return t->iface.get_max_size ? t->iface.get_max_size(t) : SIZE_MAX; }
size_t ggml_backend_buft_get_alloc_size(ggml_backend_buffer_type_t t, const ggml_tensor* tensor) {
    return t->iface.get_alloc_size ? t->iface.get_alloc_size(t, tensor) : ggml_nbytes(tensor);
}
bool ggml_backend_buft_is_host(ggml_backend_buffer_type_t t) { return t->iface.is_host ? t->iface.is_host(t) : false; }

void ggml_backend_buffer_free(ggml_backend_buffer_t b) {
    if (!b) return;
    if (b->iface.free_buffer) b->iface.free_buffer(b);
    delete b;
}
void* ggml_backend_buffer_get_base(ggml_backend_buffer_t b) { return b->size == 0 ? nullptr : b->iface.get_base(b); }
size_t ggml_backend_buffer_get_size(ggml_backend_buffer_t b) { return b->size; }
void ggml_backend_buffer_clear(ggml_backend_buffer_t b, uint8_t value) {
    if (b->size && b->iface.clear) b->iface.clear(b, value);
}
void ggml_backend_buffer_set_usage(ggml_backend_buffer_t b, enum ggml_backend_buffer_usage usage) { b->usage = usage; }
enum ggml_backend_buffer_usage ggml_backend_buffer_get_usage(ggml_backend_buffer_t b) { return b->usage; }
ggml_backend_buffer_type_t ggml_backend_buffer_get_type(ggml_backend_buffer_t b) { return b->buft; }
bool ggml_backend_buffer_is_host(ggml_backend_buffer_t b) { return ggml_backend_buft_is_host(b->buft); }

static ggml_backend_buffer_t tensor_buffer(const ggml_tensor* t) { return t->view_src ? t->view_src->buffer : t->buffer; }

void ggml_backend_tensor_set(ggml_tensor* tensor, const void* data, size_t offset, size_t size) {
    if (size == 0) return;
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    GGML_ASSERT(buf != nullptr && "tensor buffer not set");
    GGML_ASSERT(tensor->data != nullptr && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor write out of bounds");
    buf->iface.set_tensor(buf, tensor, data, offset, size);
}
void ggml_backend_tensor_get(const ggml_tensor* tensor, void* data, size_t offset, size_t size) {
    if (size == 0) return;
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    GGML_ASSERT(buf != nullptr && "tensor buffer not set");
    GGML_ASSERT(tensor->data != nullptr && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor read out of bounds");
    buf->iface.get_tensor(buf, tensor, data, offset, size);
}
void ggml_backend_tensor_set_async(ggml_backend_t backend, ggml_tensor* tensor, const void* data, size_t offset, size_t size) {
    if (backend->iface.set_tensor_async)
        backend->iface.set_tensor_async(backend, tensor, data, offset, size);
    else
        ggml_backend_tensor_set(tensor, data, offset, size);
}
void ggml_backend_tensor_get_async(ggml_backend_t backend, const ggml_tensor* tensor, void* data, size_t offset, size_t size) {
    if (backend->iface.get_tensor_async)
        backend->iface.get_tensor_async(backend, tensor, data, offset, size);
    else
        ggml_backend_tensor_get(tensor, data, offset, size);
}
void ggml_backend_tensor_memset(ggml_tensor* tensor, uint8_t value, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    GGML_ASSERT(buf && buf->iface.memset_tensor);
    buf->iface.memset_tensor(buf, tensor, value, offset, size);
}
void ggml_backend_tensor_copy(ggml_tensor* src, ggml_tensor* dst) {
    GGML_ASSERT(ggml_nbytes(src) == ggml_nbytes(dst));
    if (src == dst) return;
    ggml_backend_buffer_t sb = tensor_buffer(src), db = tensor_buffer(dst);
    if (db->iface.cpy_tensor && db->iface.cpy_tensor(db, src, dst)) return;
    if (ggml_backend_buffer_is_host(sb)) {
        ggml_backend_tensor_set(dst, src->data, 0, ggml_nbytes(src));
    } else if (ggml_backend_buffer_is_host(db)) {
        ggml_backend_tensor_get(src, dst->data, 0, ggml_nbytes(src));
    } else {
        std::vector<char> tmp(ggml_nbytes(src));
        ggml_backend_tensor_get(src, tmp.data(), 0, tmp.size());
        ggml_backend_tensor_set(dst, tmp.data(), 0, tmp.size());
    }
}

enum ggml_status ggml_backend_tensor_alloc(ggml_backend_buffer_t buffer, ggml_tensor* tensor, void* addr) {
    GGML_ASSERT(tensor->buffer == nullptr && tensor->data == nullptr && tensor->view_src == nullptr);
    GGML_ASSERT((char*)addr >= (char*)ggml_backend_buffer_get_base(buffer) &&
                (char*)addr + ggml_backend_buft_get_alloc_size(buffer->buft, tensor) <=
                    (char*)ggml_backend_buffer_get_base(buffer) + buffer->size);
    tensor->buffer = buffer;
    tensor->data   = addr;
    return buffer->iface.init_tensor ? buffer->iface.init_tensor(buffer, tensor) : GGML_STATUS_SUCCESS;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// allocate every unallocated tensor of a (no_alloc) context in ONE buffer — the weights path
// (model_manager.cpp:470-477)
ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors_from_buft(ggml_context* ctx, ggml_backend_buffer_type_t buft) {
    GGML_ASSERT(ggml_get_no_alloc(ctx));
    const size_t alignment = ggml_backend_buft_get_alignment(buft);
    size_t total           = 0;
    for (ggml_tensor* t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t)) {
        if (t->data == nullptr && t->view_src == nullptr) total += align_up(ggml_backend_buft_get_alloc_size(buft, t), alignment);
    }
    if (total == 0) return nullptr;
    ggml_backend_buffer_t buf = ggml_backend_buft_alloc_buffer(buft, total);
    if (!buf) return nullptr;
    char* base = (char*)ggml_backend_buffer_get_base(buf);
    size_t off = 0;
    for (ggml_tensor* t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t)) {
        if (t->data == nullptr && t->view_src == nullptr) {
            ggml_backend_tensor_alloc(buf, t, base + off);
            off += align_up(ggml_backend_buft_get_alloc_size(buft, t), alignment);
        }
    }
    for (ggml_tensor* t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t)) {
        if (t->view_src != nullptr && t->data == nullptr && t->view_src->data != nullptr) {
            t->buffer = t->view_src->buffer;
            t->data   = (char*)t->view_src->data + t->view_offs;
        }
    }
    return buf;
}
ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors(ggml_context* ctx, ggml_backend_t backend) {
    return ggml_backend_alloc_ctx_tensors_from_buft(ctx, ggml_backend_get_default_buffer_type(backend));
}

}  // extern "C"

// =================================================================================================
// gallocr — graph allocator: liveness-based offset assignment inside one compute buffer, with the
// same in-place reuse rule as upstream (an op that can run in place takes over a same-layout parent
// that has no other pending reader).  Deterministic: identical topology -> identical addresses, which
// is what lets the MI355X backend cache its execution plan across the per-step graph rebuilds
// (SURVEY.md F7).
// =================================================================================================
struct ggml_gallocr {
    ggml_backend_buffer_type_t buft;
    ggml_backend_buffer_t buffer = nullptr;
    size_t planned_size          = 0;
};

namespace {
bool op_can_inplace(enum ggml_op op) {
    switch (op) {
        case GGML_OP_SCALE:
        case GGML_OP_ADD:
        case GGML_OP_SUB:
        case GGML_OP_MUL:
        case GGML_OP_DIV:
        case GGML_OP_UNARY:
        case GGML_OP_RMS_NORM:
        case GGML_OP_SOFT_MAX:
            return true;
        default:
            return false;
    }
}
bool same_layout(const ggml_tensor* a, const ggml_tensor* b) {
    if (a->type != b->type) return false;
    for (int i = 0; i < GGML_MAX_DIMS; ++i)
        if (a->ne[i] != b->ne[i] || a->nb[i] != b->nb[i]) return false;
    return true;
}
struct free_block {
    size_t off, size;
};
struct planner {
    size_t alignment;
    std::vector<free_block> free_list;
    size_t high = 0;
    size_t alloc(size_t size) {
        size = align_up(std::max<size_t>(size, 1), alignment);
        // best fit
        int best = -1;
        for (int i = 0; i < (int)free_list.size(); ++i)
            if (free_list[i].size >= size && (best < 0 || free_list[i].size < free_list[best].size)) best = i;
        if (best >= 0) {
            size_t off = free_list[best].off;
            free_list[best].off += size;
            free_list[best].size -= size;
            if (free_list[best].size == 0) free_list.erase(free_list.begin() + best);
            return off;
        }
        // grow: extend a trailing free block if it touches the end
        for (int i = 0; i < (int)free_list.size(); ++i) {
            if (free_list[i].off + free_list[i].size == high) {
                size_t off = free_list[i].off;
                high       = off + size;
                free_list.erase(free_list.begin() + i);
                return off;
            }
        }
        size_t off = high;
        high += size;
        return off;
    }
    void release(size_t off, size_t size) {
        size = align_up(std::max<size_t>(size, 1), alignment);
        free_block b{off, size};
        auto it = std::lower_bound(free_list.begin(), free_list.end(), b, [](const free_block& x, const free_block& y) { return x.off < y.off; });
        it      = free_list.insert(it, b);
        // merge with next / prev
        if (it + 1 != free_list.end() && it->off + it->size == (it + 1)->off) {
            it->size += (it + 1)->size;
            free_list.erase(it + 1);
        }
        if (it != free_list.begin() && (it - 1)->off + (it - 1)->size == it->off) {
            (it - 1)->size += it->size;
            free_list.erase(it);
        }
    }
};
struct tinfo {
    int last_use  = -1;  // index of the last node reading this tensor (or a view of it)
    int n_views   = 0;
    bool own      = false;  // allocated by us in this pass
    bool pinned   = false;  // never freed (inputs / outputs / leafs)
    size_t off    = 0;
    size_t size   = 0;
    bool released = false;
};
}  // namespace

static bool gallocr_plan(ggml_gallocr* ga, ggml_cgraph* g, bool assign) {
    const size_t alignment = ggml_backend_buft_get_alignment(ga->buft);
    std::unordered_map<ggml_tensor*, tinfo> info;
    auto root = [](ggml_tensor* t) { return t->view_src ? t->view_src : t; };
    const int INF = 1 << 30;

    // pass 1: liveness
    auto touch = [&](ggml_tensor* t, int i) {
        tinfo& r = info[root(t)];
        r.last_use = std::max(r.last_use, i);
    };
    for (int i = 0; i < g->n_leafs; ++i) {
        info[root(g->leafs[i])].pinned = true;
    }
    for (int i = 0; i < g->n_nodes; ++i) {
        ggml_tensor* n = g->nodes[i];
        if (n->view_src) info[n->view_src].n_views++;
        for (int j = 0; j < GGML_MAX_SRC; ++j)
            if (n->src[j]) touch(n->src[j], i);
        if (n->flags & (GGML_TENSOR_FLAG_OUTPUT | GGML_TENSOR_FLAG_INPUT)) info[root(n)].pinned = true;
        touch(n, i);  // a view node keeps its root alive at least until the view node itself
    }
    if (g->n_nodes > 0) info[root(g->nodes[g->n_nodes - 1])].pinned = true;  // graph result

    planner pl;
    pl.alignment = alignment;
    std::vector<std::vector<ggml_tensor*>> dying(g->n_nodes);

    auto needs_alloc = [](ggml_tensor* t) { return t->view_src == nullptr && (t->data == nullptr || t->buffer == nullptr); };
    std::vector<ggml_tensor*> ours;

    auto give = [&](ggml_tensor* t) {
        tinfo& ti = info[t];
        ti.size   = ggml_backend_buft_get_alloc_size(ga->buft, t);
        ti.off    = pl.alloc(ti.size);
        ti.own    = true;
        ours.push_back(t);
    };

    // leafs first (inputs written by the host before compute; never recycled)
    for (int i = 0; i < g->n_leafs; ++i) {
        ggml_tensor* t = g->leafs[i];
        if (t->view_src == nullptr && t->buffer == nullptr) {
            t->data = nullptr;
            give(t);
        }
    }
    for (int i = 0; i < g->n_nodes; ++i) {
        ggml_tensor* n = g->nodes[i];
        if (n->view_src == nullptr && (n->buffer == nullptr || info[n].own)) {
            n->data   = nullptr;
            tinfo& ni = info[n];
            bool done = false;
            if (op_can_inplace(n->op)) {
                for (int j = 0; j < GGML_MAX_SRC && !done; ++j) {
                    ggml_tensor* p = n->src[j];
                    if (!p || p->view_src) continue;
                    auto pit = info.find(p);
                    if (pit == info.end()) continue;
                    tinfo& pi = pit->second;
                    if (pi.own && !pi.pinned && !pi.released && pi.n_views == 0 && pi.last_use == i && same_layout(n, p) &&
                        !(p->flags & GGML_TENSOR_FLAG_OUTPUT)) {
                        ni.off      = pi.off;
                        ni.size     = pi.size;
                        ni.own      = true;
                        pi.released = true;  // ownership moves to the child
                        ours.push_back(n);
                        done = true;
                    }
                }
            }
            if (!done) give(n);
        }
        // free everything whose last reader was this node
        for (int j = 0; j < GGML_MAX_SRC; ++j) {
            ggml_tensor* s = g->nodes[i]->src[j];
            if (!s) continue;
            ggml_tensor* r = root(s);
            tinfo& ri      = info[r];
            if (ri.own && !ri.pinned && !ri.released && ri.last_use == i && r != root(n)) {
                pl.release(ri.off, ri.size);
                ri.released = true;
            }
        }
        // a node nobody reads (and not pinned) can be recycled immediately
        {
            ggml_tensor* r = root(n);
            tinfo& ri      = info[r];
            if (ri.own && !ri.pinned && !ri.released && ri.last_use <= i) {
                pl.release(ri.off, ri.size);
                ri.released = true;
            }
        }
    }
    (void)INF;
    (void)needs_alloc;

    ga->planned_size = pl.high;
    if (!assign) return true;

    if (ga->buffer == nullptr || ga->buffer->size < pl.high) {
        if (ga->buffer) ggml_backend_buffer_free(ga->buffer);
        ga->buffer = ggml_backend_buft_alloc_buffer(ga->buft, std::max<size_t>(pl.high, alignment));
        if (!ga->buffer) return false;
        ggml_backend_buffer_set_usage(ga->buffer, GGML_BACKEND_BUFFER_USAGE_COMPUTE);
    }
    char* base = (char*)ggml_backend_buffer_get_base(ga->buffer);
    for (ggml_tensor* t : ours) {
        t->buffer = ga->buffer;
        t->data   = base + info[t].off;
        if (ga->buffer->iface.init_tensor) ga->buffer->iface.init_tensor(ga->buffer, t);
    }
    auto fix_view = [&](ggml_tensor* t) {
        if (t->view_src && t->view_src->data) {
            t->buffer = t->view_src->buffer;
            t->data   = (char*)t->view_src->data + t->view_offs;
        }
    };
    for (int i = 0; i < g->n_leafs; ++i) fix_view(g->leafs[i]);
    for (int i = 0; i < g->n_nodes; ++i) fix_view(g->nodes[i]);
    return true;
}

extern "C" {
ggml_gallocr_t ggml_gallocr_new(ggml_backend_buffer_type_t buft) {
    ggml_gallocr* ga = new ggml_gallocr();
    ga->buft         = buft;
    return ga;
}
void ggml_gallocr_free(ggml_gallocr_t ga) {
    if (!ga) return;
    if (ga->buffer) ggml_backend_buffer_free(ga->buffer);
    delete ga;
}
bool ggml_gallocr_reserve(ggml_gallocr_t ga, ggml_cgraph* graph) {
    if (!gallocr_plan(ga, graph, false)) return false;
    if (ga->buffer == nullptr || ga->buffer->size < ga->planned_size) {
        if (ga->buffer) ggml_backend_buffer_free(ga->buffer);
        ga->buffer = ggml_backend_buft_alloc_buffer(ga->buft, std::max<size_t>(ga->planned_size, 256));
        if (!ga->buffer) return false;
        ggml_backend_buffer_set_usage(ga->buffer, GGML_BACKEND_BUFFER_USAGE_COMPUTE);
    }
    return true;
}
bool ggml_gallocr_alloc_graph(ggml_gallocr_t ga, ggml_cgraph* graph) { return gallocr_plan(ga, graph, true); }
size_t ggml_gallocr_get_buffer_size(ggml_gallocr_t ga, int) { return ga->buffer ? ga->buffer->size : ga->planned_size; }
}
