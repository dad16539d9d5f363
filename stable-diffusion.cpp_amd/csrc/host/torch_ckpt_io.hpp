// torch_ckpt_io.hpp — PyTorch checkpoint (.ckpt / .pt / .pth / .bin) -> tensor directory (the reference reads these with
// src/model_io/torch_zip_io.cpp + pickle_io.cpp and, for files written before PyTorch 1.6 or with _use_new_zipfile_serialization=False,
// src/model_io/torch_legacy_io.cpp).  Only the directory is built here: where every tensor's bytes lie in the file; sd_load_weights then reads
// and converts them exactly like safetensors / GGUF tensors (model_io.hpp).
//
// A checkpoint is a pickled object graph whose tensors are "persistent ids": (storage type, key, device, element count) tuples handed to
// torch._utils._rebuild_tensor_v2 together with (element offset, shape, strides).  NOTHING is executed: the pickle program runs on a small value
// machine that only knows containers, strings, integers, globals-by-name and the REDUCE of the two tensor rebuilders and of collections.OrderedDict;
// any other REDUCE / BUILD / NEWOBJ yields an opaque value (Lightning checkpoints carry optimizer states, callbacks, hyper-parameter objects).
// Like the reference, tensors are collected from the root dictionary and every dictionary nested in it under their OWN key (no dotted prefix:
// {"state_dict": {...}} and a bare state dict give the same names), and only contiguous tensors are accepted.
//
//   zip container (torch >= 1.6):  <archive>/data.pkl, <archive>/data/<key> (one STORED entry per storage), <archive>/version, ...
//   legacy container:              pickle(magic 0x1950a86a20f9469cfc6c) pickle(protocol 1001) pickle(sys_info) pickle(object) pickle([storage keys])
//                                  then, per key in that order: int64 element count + raw bytes
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "model_io.hpp"

namespace sdmi {

// ---- value machine -----------------------------------------------------------------------------------------------------------------------------
struct PklValue;
using PklRef = std::shared_ptr<PklValue>;
struct PklValue {
    enum Kind { NONE, BOOL, INT, FLOAT, STR, BYTES, TUPLE, LIST, DICT, GLOBAL, STORAGE, TENSOR, OPAQUE, MARK } kind = NONE;
    int64_t i = 0;
    double f  = 0;
    std::string s;                              // STR / BYTES / GLOBAL ("module.name")
    std::vector<PklRef> items;                  // TUPLE / LIST
    std::vector<std::pair<PklRef, PklRef>> kv;  // DICT (insertion order)
    // STORAGE: s = key, elem = bytes per element, type / src as below, i = element count.  TENSOR: the same + offset (elements), ne (torch order)
    ggml_type type = GGML_TYPE_F32;
    SrcKind src    = SrcKind::NATIVE;
    bool decodable = true;
    int elem       = 0;
    int64_t offset = 0;
    std::vector<int64_t> shape;
};

inline PklRef pkl_new(PklValue::Kind k) {
    auto v  = std::make_shared<PklValue>();
    v->kind = k;
    return v;
}

struct PklResult {
    PklRef root;
    size_t consumed = 0;  // bytes of the buffer the program occupied (legacy container: pickles follow each other)
    std::map<std::string, std::pair<int, int64_t>> storages;  // every storage the program named: key -> (bytes per element, element count)
    std::string error;
};

inline bool pkl_storage_type(const std::string& g, PklValue& st) {
    struct T {
        const char* name;
        ggml_type type;
        SrcKind src;
        int elem;
        bool ok;
    };
    static const T table[] = {
        {"torch.FloatStorage", GGML_TYPE_F32, SrcKind::NATIVE, 4, true},   {"torch.HalfStorage", GGML_TYPE_F16, SrcKind::NATIVE, 2, true},
        {"torch.BFloat16Storage", GGML_TYPE_BF16, SrcKind::NATIVE, 2, true}, {"torch.DoubleStorage", GGML_TYPE_F32, SrcKind::F64, 8, true},
        {"torch.LongStorage", GGML_TYPE_F32, SrcKind::I64, 8, true},
        // integer / bool storages (step counters, position ids, masks): no parameter of the hot-path models has them — kept as "undecodable"
        {"torch.IntStorage", GGML_TYPE_F32, SrcKind::NATIVE, 4, false},    {"torch.ShortStorage", GGML_TYPE_F32, SrcKind::NATIVE, 2, false},
        {"torch.CharStorage", GGML_TYPE_F32, SrcKind::NATIVE, 1, false},   {"torch.ByteStorage", GGML_TYPE_F32, SrcKind::NATIVE, 1, false},
        {"torch.BoolStorage", GGML_TYPE_F32, SrcKind::NATIVE, 1, false},
    };
    for (const T& t : table)
        if (g == t.name) {
            st.type      = t.type;
            st.src       = t.src;
            st.elem      = t.elem;
            st.decodable = t.ok;
            return true;
        }
    return false;
}

// Runs one pickle program (protocols 2..5, the opcodes torch.save and pytorch-lightning emit).
inline PklResult pkl_run(const uint8_t* buf, size_t n) {
    PklResult R;
    std::vector<PklRef> st;
    std::map<int64_t, PklRef> memo;
    size_t p = 0;
    auto fail = [&](const std::string& m) {
        R.error = m;
        return R;
    };
    auto need = [&](size_t k) { return n - p >= k; };
    auto rd_u = [&](int bytes) {
        uint64_t v = 0;
        for (int b = 0; b < bytes; ++b) v |= (uint64_t)buf[p + b] << (8 * b);
        p += bytes;
        return v;
    };
    auto pop = [&]() {
        PklRef v = st.back();
        st.pop_back();
        return v;
    };
    auto to_mark = [&](std::vector<PklRef>& out) -> bool {
        size_t m = st.size();
        while (m > 0 && st[m - 1]->kind != PklValue::MARK) --m;
        if (m == 0) return false;
        out.assign(st.begin() + m, st.end());
        st.resize(m - 1);
        return true;
    };
    auto push_str = [&](size_t len, PklValue::Kind k) -> bool {
        if (!need(len)) return false;
        auto v = pkl_new(k);
        v->s.assign((const char*)buf + p, len);
        p += len;
        st.push_back(v);
        return true;
    };
    auto set_items = [&](PklRef d, const std::vector<PklRef>& flat) {
        if (d->kind != PklValue::DICT) return;  // SETITEMS on an opaque object: ignored
        for (size_t k = 0; k + 1 < flat.size(); k += 2) d->kv.emplace_back(flat[k], flat[k + 1]);
    };
    // REDUCE(callable, args): the only callables with a meaning here
    auto reduce = [&](const PklRef& fn, const PklRef& args) -> PklRef {
        if (fn->kind == PklValue::GLOBAL && args->kind == PklValue::TUPLE) {
            if (fn->s == "collections.OrderedDict") return pkl_new(PklValue::DICT);
            if ((fn->s == "torch._utils._rebuild_tensor_v2" || fn->s == "torch._utils._rebuild_tensor") && args->items.size() >= 4 &&
                args->items[0]->kind == PklValue::STORAGE && args->items[1]->kind == PklValue::INT && args->items[2]->kind == PklValue::TUPLE &&
                args->items[3]->kind == PklValue::TUPLE) {
                auto t     = std::make_shared<PklValue>(*args->items[0]);
                t->kind    = PklValue::TENSOR;
                t->offset  = args->items[1]->i;
                bool ok    = t->offset >= 0 && args->items[2]->items.size() == args->items[3]->items.size() && args->items[2]->items.size() <= 8;
                for (const auto& d : args->items[2]->items) {
                    ok = ok && d->kind == PklValue::INT && d->i >= 0;
                    t->shape.push_back(d->i);
                }
                // contiguous layouts only (like the reference): stride[k] = prod(shape[k+1:]) wherever shape[k] > 1
                int64_t expect = 1;
                for (size_t k = t->shape.size(); ok && k-- > 0;) {
                    const PklRef& sv = args->items[3]->items[k];
                    ok = sv->kind == PklValue::INT && (t->shape[k] <= 1 || sv->i == expect);
                    if (t->shape[k] > 0 && expect > (int64_t)1 << 46) ok = false;
                    expect *= t->shape[k] > 0 ? t->shape[k] : 1;
                }
                if (!ok) return pkl_new(PklValue::OPAQUE);  // non-contiguous or malformed: not offered to the loader
                return t;
            }
            if (fn->s == "torch._utils._rebuild_parameter" && !args->items.empty() && args->items[0]->kind == PklValue::TENSOR) return args->items[0];
        }
        return pkl_new(PklValue::OPAQUE);
    };
    while (p < n) {
        const uint8_t op = buf[p++];
        switch (op) {
            case 0x80: if (!need(1)) return fail("truncated pickle"); if (buf[p] < 2 || buf[p] > 5) return fail("unsupported pickle protocol"); ++p; break;  // PROTO
            case 0x95: if (!need(8)) return fail("truncated pickle"); p += 8; break;  // FRAME
            case '.':
                if (st.empty()) return fail("empty pickle stack at STOP");
                R.root     = st.back();
                R.consumed = p;
                return R;
            case '(': st.push_back(pkl_new(PklValue::MARK)); break;
            case 'N': st.push_back(pkl_new(PklValue::NONE)); break;
            case 0x88: case 0x89: { auto v = pkl_new(PklValue::BOOL); v->i = op == 0x88; st.push_back(v); break; }
            case 'K': case 'M': case 'J': {  // BININT1 / BININT2 / BININT
                const int b = op == 'K' ? 1 : (op == 'M' ? 2 : 4);
                if (!need(b)) return fail("truncated pickle");
                auto v = pkl_new(PklValue::INT);
                const uint64_t u = rd_u(b);
                v->i = op == 'J' ? (int64_t)(int32_t)(uint32_t)u : (int64_t)u;
                st.push_back(v);
                break;
            }
            case 0x8a: case 0x8b: {  // LONG1 / LONG4: little-endian two's complement
                const int lb = op == 0x8a ? 1 : 4;
                if (!need(lb)) return fail("truncated pickle");
                const uint64_t len = rd_u(lb);
                if (len > 64 || !need(len)) return fail("truncated pickle integer");
                // wider than 64 bits (the legacy container's 80-bit magic number): the low 8 bytes are kept
                uint64_t u = len ? rd_u((int)(len > 8 ? 8 : len)) : 0;
                if (len > 8) p += len - 8;
                if (len > 0 && len < 8 && (u >> (8 * len - 1)) & 1) u |= ~0ull << (8 * len);
                auto v = pkl_new(PklValue::INT);
                v->i   = (int64_t)u;
                st.push_back(v);
                break;
            }
            case 'G': {  // BINFLOAT: big-endian double
                if (!need(8)) return fail("truncated pickle");
                uint64_t u = 0;
                for (int b = 0; b < 8; ++b) u = (u << 8) | buf[p + b];
                p += 8;
                auto v = pkl_new(PklValue::FLOAT);
                memcpy(&v->f, &u, 8);
                st.push_back(v);
                break;
            }
            case 'X': case 'T': case 'B': {  // BINUNICODE / BINSTRING / BINBYTES: u32 length
                if (!need(4)) return fail("truncated pickle");
                const uint64_t len = rd_u(4);
                if (!push_str(len, op == 'B' ? PklValue::BYTES : PklValue::STR)) return fail("truncated pickle string");
                break;
            }
            case 0x8c: case 'U': case 'C': {  // SHORT_BINUNICODE / SHORT_BINSTRING / SHORT_BINBYTES: u8 length
                if (!need(1)) return fail("truncated pickle");
                const uint64_t len = rd_u(1);
                if (!push_str(len, op == 'C' ? PklValue::BYTES : PklValue::STR)) return fail("truncated pickle string");
                break;
            }
            case 0x8d: case 0x8e: {  // BINUNICODE8 / BINBYTES8
                if (!need(8)) return fail("truncated pickle");
                const uint64_t len = rd_u(8);
                if (len > n || !push_str(len, op == 0x8e ? PklValue::BYTES : PklValue::STR)) return fail("truncated pickle string");
                break;
            }
            case 'c': {  // GLOBAL: "module\nname\n"
                std::string parts[2];
                for (int k = 0; k < 2; ++k) {
                    size_t e = p;
                    while (e < n && buf[e] != '\n') ++e;
                    if (e >= n) return fail("truncated pickle GLOBAL");
                    parts[k].assign((const char*)buf + p, e - p);
                    p = e + 1;
                }
                auto v = pkl_new(PklValue::GLOBAL);
                v->s   = parts[0] + "." + parts[1];
                st.push_back(v);
                break;
            }
            case 0x93: {  // STACK_GLOBAL
                if (st.size() < 2) return fail("pickle stack underflow");
                PklRef name = pop(), mod = pop();
                auto v = pkl_new(PklValue::GLOBAL);
                v->s   = mod->s + "." + name->s;
                st.push_back(v);
                break;
            }
            case '}': st.push_back(pkl_new(PklValue::DICT)); break;
            case ']': st.push_back(pkl_new(PklValue::LIST)); break;
            case ')': st.push_back(pkl_new(PklValue::TUPLE)); break;
            case 0x8f: st.push_back(pkl_new(PklValue::OPAQUE)); break;  // EMPTY_SET
            case 't': case 'l': case 'd': {  // TUPLE / LIST / DICT from mark
                std::vector<PklRef> it;
                if (!to_mark(it)) return fail("pickle container without MARK");
                auto v = pkl_new(op == 't' ? PklValue::TUPLE : (op == 'l' ? PklValue::LIST : PklValue::DICT));
                if (op == 'd') set_items(v, it); else v->items = it;
                st.push_back(v);
                break;
            }
            case 0x85: case 0x86: case 0x87: {  // TUPLE1..3
                const size_t k = op - 0x84;
                if (st.size() < k) return fail("pickle stack underflow");
                auto v = pkl_new(PklValue::TUPLE);
                v->items.assign(st.end() - k, st.end());
                st.resize(st.size() - k);
                st.push_back(v);
                break;
            }
            case 'a': {  // APPEND
                if (st.size() < 2) return fail("pickle stack underflow");
                PklRef x = pop();
                if (st.back()->kind == PklValue::LIST) st.back()->items.push_back(x);
                break;
            }
            case 'e': case 0x90: {  // APPENDS / ADDITEMS
                std::vector<PklRef> it;
                if (!to_mark(it) || st.empty()) return fail("pickle APPENDS without MARK");
                if (st.back()->kind == PklValue::LIST) st.back()->items.insert(st.back()->items.end(), it.begin(), it.end());
                break;
            }
            case 's': {  // SETITEM
                if (st.size() < 3) return fail("pickle stack underflow");
                PklRef v = pop(), k = pop();
                set_items(st.back(), {k, v});
                break;
            }
            case 'u': {  // SETITEMS
                std::vector<PklRef> it;
                if (!to_mark(it) || st.empty()) return fail("pickle SETITEMS without MARK");
                set_items(st.back(), it);
                break;
            }
            case 'q': case 'r': {  // BINPUT / LONG_BINPUT
                const int b = op == 'q' ? 1 : 4;
                if (!need(b) || st.empty()) return fail("truncated pickle");
                memo[(int64_t)rd_u(b)] = st.back();
                break;
            }
            case 0x94: if (st.empty()) return fail("pickle stack underflow"); memo[(int64_t)memo.size()] = st.back(); break;  // MEMOIZE
            case 'h': case 'j': {  // BINGET / LONG_BINGET
                const int b = op == 'h' ? 1 : 4;
                if (!need(b)) return fail("truncated pickle");
                auto it = memo.find((int64_t)rd_u(b));
                if (it == memo.end()) return fail("pickle memo key not set");
                st.push_back(it->second);
                break;
            }
            case 'Q': {  // BINPERSID: ('storage', <storage type global>, key, device, element count [, view metadata])
                if (st.empty()) return fail("pickle stack underflow");
                PklRef id = pop();
                auto v    = pkl_new(PklValue::OPAQUE);
                if (id->kind == PklValue::TUPLE && id->items.size() >= 5 && id->items[0]->kind == PklValue::STR && id->items[0]->s == "storage" &&
                    id->items[1]->kind == PklValue::GLOBAL && id->items[2]->kind == PklValue::STR && id->items[4]->kind == PklValue::INT) {
                    PklValue sv;
                    if (pkl_storage_type(id->items[1]->s, sv)) {
                        *v      = sv;
                        v->kind = PklValue::STORAGE;
                        v->s    = id->items[2]->s;
                        v->i    = id->items[4]->i;
                        R.storages[v->s] = {v->elem, v->i};
                    }
                }
                st.push_back(v);
                break;
            }
            case 'R': {  // REDUCE
                if (st.size() < 2) return fail("pickle stack underflow");
                PklRef args = pop(), fn = pop();
                st.push_back(reduce(fn, args));
                break;
            }
            case 0x81: {  // NEWOBJ: cls.__new__(cls, *args) — an OrderedDict subclass instance or any other object
                if (st.size() < 2) return fail("pickle stack underflow");
                PklRef args = pop(), cls = pop();
                st.push_back(cls->kind == PklValue::GLOBAL && cls->s == "collections.OrderedDict" ? pkl_new(PklValue::DICT) : pkl_new(PklValue::OPAQUE));
                break;
            }
            case 0x92: {  // NEWOBJ_EX
                if (st.size() < 3) return fail("pickle stack underflow");
                pop(); pop(); pop();
                st.push_back(pkl_new(PklValue::OPAQUE));
                break;
            }
            case 'b': {  // BUILD: obj.__setstate__(state) — state is dropped (OrderedDict._metadata, module attributes)
                if (st.size() < 2) return fail("pickle stack underflow");
                pop();
                break;
            }
            case '0': if (st.empty()) return fail("pickle stack underflow"); st.pop_back(); break;  // POP
            case '2': if (st.empty()) return fail("pickle stack underflow"); st.push_back(st.back()); break;  // DUP
            case '1': { std::vector<PklRef> it; if (!to_mark(it)) return fail("pickle POP_MARK without MARK"); break; }
            default: {
                char hex[8];
                snprintf(hex, sizeof hex, "0x%02x", op);
                return fail(std::string("unsupported pickle opcode ") + hex);
            }
        }
    }
    return fail("unterminated pickle");
}

// tensors of the root dictionary and of every dictionary nested in it, under their own key (the reference's collect_tensors_from_pickle_value)
inline void pkl_collect(const PklRef& v, std::vector<std::pair<std::string, PklRef>>& out, int depth = 0) {
    if (!v || v->kind != PklValue::DICT || depth > 8) return;
    for (const auto& kv : v->kv) {
        if (kv.first->kind == PklValue::STR && kv.second->kind == PklValue::TENSOR)
            out.emplace_back(kv.first->s, kv.second);
        else if (kv.second->kind == PklValue::DICT)
            pkl_collect(kv.second, out, depth + 1);
    }
}

// pickled tensor -> directory entry; `base` = file offset of the first byte of its storage, `avail` = bytes the storage holds in the file
inline bool pkl_tensor_entry(const std::string& name, const PklValue& t, uint64_t base, uint64_t avail, ModelFile& mf) {
    if (!t.decodable) {
        mf.undecodable[name] = "an integer / bool torch storage";
        return true;
    }
    FileTensor ft;
    ft.name   = name;
    ft.type   = t.type;
    ft.kind   = t.src;
    uint64_t nelem = 1;
    for (int64_t d : t.shape)
        if (d == 0) {  // a zero-element tensor (torch.zeros(0, 4)): the reference accepts the file and carries no data for it (pickle_io.cpp: has_zero_dimension)
            mf.undecodable[name] = "a zero-element tensor";
            return true;
        }
    for (int64_t d : t.shape) {
        if (d < 0 || nelem > (1ull << 46) / (uint64_t)d) {
            mf.error = "tensor '" + name + "' has a negative or overflowing shape";
            return false;
        }
        nelem *= (uint64_t)d;
    }
    // ggml order = reversed torch order; more than four dims fold their outer ones like the safetensors reader does
    std::vector<int64_t> ne(t.shape.rbegin(), t.shape.rend());
    while (ne.size() > 4) {
        ne[3] *= ne.back();
        ne.pop_back();
    }
    ft.n_dims = (int)ne.size();
    for (size_t d = 0; d < ne.size(); ++d) ft.ne[d] = ne[d];
    const uint64_t bytes = nelem * (uint64_t)t.elem, off = (uint64_t)t.offset * (uint64_t)t.elem;
    if (off > avail || bytes > avail - off) {
        mf.error = "tensor '" + name + "' exceeds its storage '" + t.s + "'";
        return false;
    }
    ft.offset = base + off;
    ft.nbytes = bytes;
    mf.tensors.push_back(ft);
    return true;
}

// ---- zip container (STORED entries; ZIP64 sizes / offsets) -------------------------------------------------------------------------------------
struct ZipEntry {
    std::string name;
    uint64_t data_offset = 0, size = 0;
    int method = 0;
};

inline bool zip_directory(FILE* f, uint64_t fsize, std::vector<ZipEntry>& out, std::string& err) {
    auto rd = [&](uint64_t off, void* dst, size_t nb) { return off <= fsize && nb <= fsize - off && fseeko(f, (off_t)off, SEEK_SET) == 0 && fread(dst, 1, nb, f) == nb; };
    auto u16 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); };
    auto u32 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    auto u64 = [&](const uint8_t* p) { return (uint64_t)u32(p) | ((uint64_t)u32(p + 4) << 32); };
    // end-of-central-directory record: within the last 64 KB + 22 bytes
    const uint64_t tail = fsize < 65557 ? fsize : 65557;
    std::vector<uint8_t> tb(tail);
    if (tail < 22 || !rd(fsize - tail, tb.data(), tail)) {
        err = "not a zip archive";
        return false;
    }
    int64_t e = -1;
    for (int64_t k = (int64_t)tail - 22; k >= 0; --k)
        if (u32(&tb[k]) == 0x06054b50u) {
            e = k;
            break;
        }
    if (e < 0) {
        err = "zip end-of-central-directory record not found";
        return false;
    }
    uint64_t count = u16(&tb[e + 10]), cd_size = u32(&tb[e + 12]), cd_off = u32(&tb[e + 16]);
    if (e >= 20 && u32(&tb[e - 20]) == 0x07064b50u) {  // zip64 locator -> zip64 end-of-central-directory record
        const uint64_t z = u64(&tb[e - 20 + 8]);
        uint8_t zb[56];
        if (!rd(z, zb, 56) || u32(zb) != 0x06064b50u) {
            err = "malformed zip64 end-of-central-directory record";
            return false;
        }
        count   = u64(zb + 32);
        cd_size = u64(zb + 40);
        cd_off  = u64(zb + 48);
    }
    if (cd_off > fsize || cd_size > fsize - cd_off || count > cd_size / 46 + 1) {
        err = "zip central directory lies outside the file";
        return false;
    }
    std::vector<uint8_t> cd(cd_size);
    if (cd_size && !rd(cd_off, cd.data(), cd_size)) {
        err = "cannot read the zip central directory";
        return false;
    }
    size_t p = 0;
    for (uint64_t k = 0; k < count; ++k) {
        if (cd_size - p < 46 || u32(&cd[p]) != 0x02014b50u) {
            err = "malformed zip central directory entry";
            return false;
        }
        ZipEntry z;
        z.method            = (int)u16(&cd[p + 10]);
        uint64_t csize      = u32(&cd[p + 20]), usize = u32(&cd[p + 24]), lho = u32(&cd[p + 42]);
        const size_t nl = u16(&cd[p + 28]), xl = u16(&cd[p + 30]), cl = u16(&cd[p + 32]);
        if (cd_size - p - 46 < nl + xl + cl) {
            err = "malformed zip central directory entry";
            return false;
        }
        z.name.assign((const char*)&cd[p + 46], nl);
        // zip64 extended information (header id 1): the fields whose 32-bit value is 0xffffffff, in the order usize, csize, local header offset
        for (size_t x = p + 46 + nl; x + 4 <= p + 46 + nl + xl;) {
            const uint32_t id = u16(&cd[x]), sz = u16(&cd[x + 2]);
            if (x + 4 + sz > p + 46 + nl + xl) break;
            if (id == 1) {
                size_t q = x + 4;
                if (usize == 0xffffffffu && q + 8 <= x + 4 + sz) usize = u64(&cd[q]), q += 8;
                if (csize == 0xffffffffu && q + 8 <= x + 4 + sz) csize = u64(&cd[q]), q += 8;
                if (lho == 0xffffffffu && q + 8 <= x + 4 + sz) lho = u64(&cd[q]), q += 8;
            }
            x += 4 + sz;
        }
        uint8_t lh[30];
        if (!rd(lho, lh, 30) || u32(lh) != 0x04034b50u) {
            err = "zip local header of '" + z.name + "' lies outside the file";
            return false;
        }
        z.data_offset = lho + 30 + u16(lh + 26) + u16(lh + 28);
        z.size        = usize;
        if (z.method == 0 && (csize != usize || z.data_offset > fsize || usize > fsize - z.data_offset)) {
            err = "zip entry '" + z.name + "' lies outside the file";
            return false;
        }
        out.push_back(z);
        p += 46 + nl + xl + cl;
    }
    return true;
}

inline bool read_torch_zip(const std::string& path, ModelFile& mf) {
    mf.path = path;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        mf.error = "cannot open " + path;
        return false;
    }
    fseeko(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftello(f);
    std::vector<ZipEntry> dir;
    if (!zip_directory(f, fsize, dir, mf.error)) {
        fclose(f);
        return false;
    }
    bool found = false, ok = true;
    for (const ZipEntry& z : dir) {
        const size_t pos = z.name.find("data.pkl");
        if (pos == std::string::npos || pos + 8 != z.name.size()) continue;
        found = true;
        if (z.method != 0 || z.size > (1ull << 31)) {
            mf.error = "'" + z.name + "' is compressed or too large (torch.save stores its entries uncompressed)";
            ok       = false;
            break;
        }
        std::vector<uint8_t> pkl(z.size);
        if (fseeko(f, (off_t)z.data_offset, SEEK_SET) != 0 || fread(pkl.data(), 1, pkl.size(), f) != pkl.size()) {
            mf.error = "cannot read '" + z.name + "'";
            ok       = false;
            break;
        }
        PklResult r = pkl_run(pkl.data(), pkl.size());
        if (!r.root) {
            mf.error = "torch checkpoint pickle: " + r.error;
            ok       = false;
            break;
        }
        std::vector<std::pair<std::string, PklRef>> ts;
        pkl_collect(r.root, ts);
        const std::string prefix = z.name.substr(0, pos) + "data/";
        std::map<std::string, const ZipEntry*> by_name;
        for (const ZipEntry& d : dir) by_name[d.name] = &d;
        for (const auto& nt : ts) {
            const auto it = by_name.find(prefix + nt.second->s);
            if (it == by_name.end()) {
                mf.error = "storage entry '" + prefix + nt.second->s + "' was not found";
                ok       = false;
                break;
            }
            if (it->second->method != 0) {
                mf.error = "storage entry '" + it->first + "' is compressed";
                ok       = false;
                break;
            }
            if (it->second->size < (uint64_t)nt.second->i * (uint64_t)nt.second->elem) {
                mf.error = "storage entry '" + it->first + "' is smaller than the pickle says";
                ok       = false;
                break;
            }
            if (!pkl_tensor_entry(nt.first, *nt.second, it->second->data_offset, it->second->size, mf)) {
                ok = false;
                break;
            }
        }
        if (!ok) break;
    }
    fclose(f);
    if (ok && !found) {
        mf.error = "data.pkl was not found in '" + path + "'";
        ok       = false;
    }
    if (ok && mf.tensors.empty() && mf.undecodable.empty()) {
        mf.error = "torch pickle does not contain a supported state_dict";
        ok       = false;
    }
    return ok;
}

// ---- legacy container --------------------------------------------------------------------------------------------------------------------------
inline bool read_torch_legacy(const std::string& path, ModelFile& mf) {
    mf.path = path;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        mf.error = "cannot open " + path;
        return false;
    }
    fseeko(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftello(f);
    // the five pickles sit at the head of the file; the object's pickle of a multi-GB checkpoint stays in the low megabytes
    const uint64_t head = fsize < (256ull << 20) ? fsize : (256ull << 20);
    std::vector<uint8_t> buf(head);
    fseeko(f, 0, SEEK_SET);
    if (fread(buf.data(), 1, head, f) != head) {
        fclose(f);
        mf.error = "cannot read " + path;
        return false;
    }
    fclose(f);
    size_t p = 0;
    PklResult pk[5];
    for (int k = 0; k < 5; ++k) {
        pk[k] = pkl_run(buf.data() + p, buf.size() - p);
        if (!pk[k].root) {
            mf.error = "legacy torch checkpoint, pickle " + std::to_string(k) + ": " + pk[k].error;
            return false;
        }
        p += pk[k].consumed;
    }
    if (pk[1].root->kind != PklValue::INT || pk[1].root->i != 1001) {
        mf.error = "legacy torch checkpoint: unexpected protocol version";
        return false;
    }
    if (pk[4].root->kind != PklValue::LIST) {
        mf.error = "legacy torch checkpoint: the storage key list is missing";
        return false;
    }
    std::vector<std::pair<std::string, PklRef>> ts;
    pkl_collect(pk[3].root, ts);
    // element size of every storage the object's pickle named (also those of values this reader does not collect) -> where each storage's bytes start
    const std::map<std::string, std::pair<int, int64_t>>& sinfo = pk[3].storages;
    std::map<std::string, std::pair<uint64_t, uint64_t>> where;  // key -> (offset of the raw bytes, byte count)
    uint64_t off = p;
    for (const PklRef& k : pk[4].root->items) {
        if (k->kind != PklValue::STR) {
            mf.error = "legacy torch checkpoint: malformed storage key list";
            return false;
        }
        // int64 element count, then the raw bytes; a storage no collected tensor uses (optimizer state of an opaque object) cannot be sized
        const auto si = sinfo.find(k->s);
        if (off > fsize || fsize - off < 8) {
            mf.error = "legacy torch checkpoint: storage '" + k->s + "' lies outside the file";
            return false;
        }
        uint64_t numel = 0;
        if (off + 8 <= head) {
            memcpy(&numel, buf.data() + off, 8);
        } else {
            FILE* g = fopen(path.c_str(), "rb");
            const bool okr = g && fseeko(g, (off_t)off, SEEK_SET) == 0 && fread(&numel, 8, 1, g) == 1;
            if (g) fclose(g);
            if (!okr) {
                mf.error = "cannot read " + path;
                return false;
            }
        }
        if (si == sinfo.end()) {
            mf.error = "legacy torch checkpoint: storage '" + k->s + "' is of a type this reader does not know (its element size is unknown)";
            return false;
        }
        const uint64_t bytes = numel * (uint64_t)si->second.first;
        if (numel > (1ull << 46) || bytes > fsize - off - 8 || (int64_t)numel < si->second.second) {
            mf.error = "legacy torch checkpoint: storage '" + k->s + "' lies outside the file";
            return false;
        }
        where[k->s] = {off + 8, bytes};
        off += 8 + bytes;
    }
    for (const auto& nt : ts) {
        const auto w = where.find(nt.second->s);
        if (w == where.end()) {
            mf.error = "legacy torch checkpoint: storage '" + nt.second->s + "' is not in the key list";
            return false;
        }
        if (!pkl_tensor_entry(nt.first, *nt.second, w->second.first, w->second.second, mf)) return false;
    }
    if (mf.tensors.empty() && mf.undecodable.empty()) {
        mf.error = "torch pickle does not contain a supported state_dict";
        return false;
    }
    return true;
}

}  // namespace sdmi
