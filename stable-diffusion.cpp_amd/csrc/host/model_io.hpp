// model_io.hpp — on-disk checkpoint formats -> tensor directory: safetensors and GGUF.
//
// What the reference does (src/model_io/safetensors_io.cpp, gguf_io.cpp, src/model_loader.cpp:160-205, 1252): build a name -> TensorStorage
// map (dtype, shape, file offset) per file, then stream every tensor the model declares through convert_tensor() (file dtype -> f32 ->
// the parameter's ggml type) into the backend buffer with ggml_backend_tensor_set.  This header restates the two container formats
// from their public specifications (no reference code): the engine (engine.cpp sd_load_weights) does the convert-and-upload part with
// the same rule.  Name conversion between checkpoint dialects (src/name_conversion.cpp) is NOT done here: names must already be the
// original-LDM / sd.cpp GGUF names the graph builders register ("model.diffusion_model.…", "first_stage_model.…").
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "ggml.h"

namespace sdmi {

// dtypes the safetensors format names that ggml has no element type for: decoded to f32 at load time (the reference widens them the
// same way before its own convert step — F8 -> F16, F64 -> F32, I64 -> I32: src/model_io/safetensors_io.cpp:79-99, src/model_loader.cpp:81-153)
// Q4_1 / Q5_0 / Q5_1: GGUF block quantisations the graph never computes in (the engine's parameter types are f32 / f16 / bf16 / q8_0 / q4_0); they are
// decoded to f32 at load time and re-encoded in the parameter's type, like convert_tensor does for every source type (src/model_loader.cpp:155-205)
// K-quants (Q2_K .. Q6_K: super-blocks of 256 weights with 4- / 6-bit sub-block scales — the q2_k / q3_k / q4_k FLUX and SD3.5 files docs/flux.md:36-38
// points its users at) and IQ4_NL (32 weights, non-linear 4-bit code book) take the same route: decoded to f32 at load, re-encoded in the parameter's type.
enum class SrcKind { NATIVE, F64, I64, F8_E4M3, F8_E5M2, Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, IQ4_NL };

// ggml type id -> (kind, weights per block, bytes per block) of the block formats decoded at load time; false for every other type
inline bool gguf_decoded_kind(int ggml_type_id, SrcKind& kind, int& blck, int& bytes) {
    switch (ggml_type_id) {
        case 3: kind = SrcKind::Q4_1, blck = 32, bytes = 20; return true;
        case 6: kind = SrcKind::Q5_0, blck = 32, bytes = 22; return true;
        case 7: kind = SrcKind::Q5_1, blck = 32, bytes = 24; return true;
        case 10: kind = SrcKind::Q2_K, blck = 256, bytes = 84; return true;
        case 11: kind = SrcKind::Q3_K, blck = 256, bytes = 110; return true;
        case 12: kind = SrcKind::Q4_K, blck = 256, bytes = 144; return true;
        case 13: kind = SrcKind::Q5_K, blck = 256, bytes = 176; return true;
        case 14: kind = SrcKind::Q6_K, blck = 256, bytes = 210; return true;
        case 20: kind = SrcKind::IQ4_NL, blck = 32, bytes = 18; return true;
        default: return false;
    }
}

struct FileTensor {
    std::string name;
    ggml_type type = GGML_TYPE_F32;
    SrcKind kind   = SrcKind::NATIVE;
    int64_t ne[4]  = {1, 1, 1, 1};  // ggml order (ne0 fastest)
    int n_dims     = 0;
    uint64_t offset = 0;            // absolute file offset of the data
    uint64_t nbytes = 0;
};

struct ModelFile {
    std::string path, error;
    std::vector<FileTensor> tensors;
    std::map<std::string, std::string> undecodable;  // tensor name -> dtype this build cannot decode (K-quants, I8, BOOL ...): an error if a declared parameter needs one
    std::map<std::string, std::string> metadata;  // safetensors __metadata__ / GGUF string KVs
};

// ---- minimal JSON (the safetensors header is a flat object of objects) --------------------------------------
struct JsonCursor {
    const char* p;
    const char* end;
    bool ok = true;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    std::string str() {
        ws();
        std::string s;
        if (p >= end || *p != '"') {
            ok = false;
            return s;
        }
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'u':  // keep \uXXXX escapes verbatim (tensor names are ASCII)
                        s += "\\u";
                        break;
                    default: s += *p;
                }
                ++p;
            } else {
                s += *p++;
            }
        }
        if (p < end) ++p;
        return s;
    }
    int64_t integer() {
        ws();
        int64_t v = 0;
        bool neg  = false;
        if (p < end && *p == '-') {
            neg = true;
            ++p;
        }
        if (p >= end || *p < '0' || *p > '9') ok = false;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        return neg ? -v : v;
    }
    // skip any value (used for unknown keys)
    void skip() {
        ws();
        if (p >= end) {
            ok = false;
            return;
        }
        if (*p == '"') {
            (void)str();
        } else if (*p == '{' || *p == '[') {
            const char open = *p, close = (*p == '{') ? '}' : ']';
            int depth = 0;
            while (p < end) {
                if (*p == '"') {
                    (void)str();
                    continue;
                }
                if (*p == open) ++depth;
                if (*p == close && --depth == 0) {
                    ++p;
                    return;
                }
                ++p;
            }
            ok = false;
        } else {
            while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
        }
    }
};

inline bool st_dtype(const std::string& s, ggml_type& t, SrcKind& kind, size_t& elem_bytes) {
    kind = SrcKind::NATIVE;
    if (s == "F32") t = GGML_TYPE_F32, elem_bytes = 4;
    else if (s == "F16") t = GGML_TYPE_F16, elem_bytes = 2;
    else if (s == "BF16") t = GGML_TYPE_BF16, elem_bytes = 2;
    else if (s == "F64") t = GGML_TYPE_F32, kind = SrcKind::F64, elem_bytes = 8;
    else if (s == "I64") t = GGML_TYPE_F32, kind = SrcKind::I64, elem_bytes = 8;
    else if (s == "F8_E4M3") t = GGML_TYPE_F32, kind = SrcKind::F8_E4M3, elem_bytes = 1;
    else if (s == "F8_E5M2") t = GGML_TYPE_F32, kind = SrcKind::F8_E5M2, elem_bytes = 1;
    else return false;  // I8 / I16 / I32 / U8 / BOOL: no parameter of the hot-path models has these
    return true;
}

// OCP 8-bit floats -> f32.  E4M3 ("fn": bias 7, no infinities, S.1111.111 = NaN, subnormals m/8 * 2^-6); E5M2 is the high byte of an IEEE half.
inline float f8_e4m3_to_f32(uint8_t v) {
    const int sign = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = std::ldexp((float)m, -9);
    else r = std::ldexp(1.0f + (float)m / 8.0f, e - 7);
    return sign ? -r : r;
}
inline float f8_e5m2_to_f32(uint8_t v) { return ggml_fp16_to_fp32((ggml_fp16_t)((uint16_t)v << 8)); }

// widen a non-native safetensors payload to f32 (n elements)
inline void decode_src_kind(SrcKind kind, const uint8_t* raw, int64_t n, float* dst) {
    switch (kind) {
        case SrcKind::F64:
            for (int64_t i = 0; i < n; ++i) {
                double d;
                memcpy(&d, raw + 8 * i, 8);
                dst[i] = (float)d;
            }
            break;
        case SrcKind::I64:
            for (int64_t i = 0; i < n; ++i) {
                int64_t d;
                memcpy(&d, raw + 8 * i, 8);
                dst[i] = (float)(int32_t)d;  // i64 -> i32 like the reference, then the parameter's float type
            }
            break;
        case SrcKind::F8_E4M3:
            for (int64_t i = 0; i < n; ++i) dst[i] = f8_e4m3_to_f32(raw[i]);
            break;
        case SrcKind::F8_E5M2:
            for (int64_t i = 0; i < n; ++i) dst[i] = f8_e5m2_to_f32(raw[i]);
            break;
        // ggml block layouts (32 weights per block; byte j of qs holds element j in its low and element j + 16 in its high nibble):
        //   q4_1 {f16 d, f16 m, u8 qs[16]}            x = q * d + m
        //   q5_0 {f16 d, u8 qh[4], u8 qs[16]}         x = ((q | bit5 << 4) - 16) * d, bit5 of element j = bit j of the little-endian u32 qh
        //   q5_1 {f16 d, f16 m, u8 qh[4], u8 qs[16]}  x = (q | bit5 << 4) * d + m
        case SrcKind::Q4_1:
        case SrcKind::Q5_0:
        case SrcKind::Q5_1: {
            const int bs = kind == SrcKind::Q4_1 ? 20 : (kind == SrcKind::Q5_0 ? 22 : 24);
            for (int64_t b = 0; b < n / 32; ++b) {
                const uint8_t* p = raw + b * bs;
                ggml_fp16_t dh, mh = 0;
                memcpy(&dh, p, 2);
                p += 2;
                if (kind != SrcKind::Q5_0) {
                    memcpy(&mh, p, 2);
                    p += 2;
                }
                const float d = ggml_fp16_to_fp32(dh), m = kind != SrcKind::Q5_0 ? ggml_fp16_to_fp32(mh) : 0.f;
                uint32_t qh = 0;
                if (kind != SrcKind::Q4_1) {
                    memcpy(&qh, p, 4);
                    p += 4;
                }
                float* y = dst + b * 32;
                for (int j = 0; j < 16; ++j) {
                    int x0 = p[j] & 0x0F, x1 = p[j] >> 4;
                    if (kind != SrcKind::Q4_1) {
                        x0 |= (int)((qh >> j) & 1u) << 4;
                        x1 |= (int)((qh >> (j + 16)) & 1u) << 4;
                    }
                    if (kind == SrcKind::Q5_0) {
                        y[j]      = (float)(x0 - 16) * d;
                        y[j + 16] = (float)(x1 - 16) * d;
                    } else {
                        y[j]      = (float)x0 * d + m;
                        y[j + 16] = (float)x1 * d + m;
                    }
                }
            }
            break;
        }
        // ---- K-quants (ggml-common.h block_q{2,3,4,5,6}_K; QK_K = 256; all fields little-endian).  Values follow upstream's dequantize_row_q*_K.
        //   q2_K {u8 scales[16] (low nibble: scale, high nibble: min, per 16 weights); u8 qs[64] (2 bits per weight); f16 d; f16 dmin}
        //        x = d * sc * q - dmin * mn.  Weight order: per 128-weight half, bit pairs 0..3 of qs[0..31]: pair s covers weights 32 s .. 32 s + 31
        case SrcKind::Q2_K:
            for (int64_t blk = 0; blk < n / 256; ++blk) {
                const uint8_t* p  = raw + blk * 84;
                const uint8_t* sc = p;
                const uint8_t* q  = p + 16;
                ggml_fp16_t dh, mh;
                memcpy(&dh, p + 80, 2);
                memcpy(&mh, p + 82, 2);
                const float d = ggml_fp16_to_fp32(dh), dmin = ggml_fp16_to_fp32(mh);
                float* y = dst + blk * 256;
                int is   = 0;
                for (int half = 0; half < 2; ++half, q += 32)
                    for (int shift = 0; shift < 8; shift += 2)
                        for (int part = 0; part < 2; ++part, ++is) {
                            const float dl = d * (float)(sc[is] & 0xF), ml = dmin * (float)(sc[is] >> 4);
                            for (int l = 0; l < 16; ++l) *y++ = dl * (float)((q[l + 16 * part] >> shift) & 3) - ml;
                        }
            }
            break;
        //   q3_K {u8 hmask[32] (bit b of hmask[l]: high bit of weight 32 b + l); u8 qs[64] (low 2 bits, order as q2_K); u8 scales[12] (16 six-bit scales:
        //        low 4 bits in bytes 0..7 (nibbles), high 2 bits in bytes 8..11); f16 d}     x = d * (scale - 32) * (q2 - (hbit ? 0 : 4))
        case SrcKind::Q3_K:
            for (int64_t blk = 0; blk < n / 256; ++blk) {
                const uint8_t* p  = raw + blk * 110;
                const uint8_t* hm = p;
                const uint8_t* q  = p + 32;
                const uint8_t* s8 = p + 96;
                ggml_fp16_t dh;
                memcpy(&dh, p + 108, 2);
                const float d = ggml_fp16_to_fp32(dh);
                int scales[16];
                for (int j = 0; j < 16; ++j) {
                    const int lo = j < 8 ? (s8[j] & 0xF) : (s8[j - 8] >> 4);
                    const int hi = (s8[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
                    scales[j]    = (lo | (hi << 4)) - 32;
                }
                float* y  = dst + blk * 256;
                int is    = 0;
                uint8_t m = 1;
                for (int half = 0; half < 2; ++half, q += 32)
                    for (int shift = 0; shift < 8; shift += 2, m = (uint8_t)(m << 1))
                        for (int part = 0; part < 2; ++part, ++is) {
                            const float dl = d * (float)scales[is];
                            for (int l = 0; l < 16; ++l) {
                                const int k = l + 16 * part;
                                *y++        = dl * (float)((int)((q[k] >> shift) & 3) - ((hm[k] & m) ? 0 : 4));
                            }
                        }
            }
            break;
        //   q4_K {f16 d; f16 dmin; u8 scales[12] (8 six-bit scales + 8 six-bit mins, get_scale_min_k4); u8 qs[128]}: per 64 weights, 32 low nibbles then 32 high
        //   q5_K {f16 d; f16 dmin; u8 scales[12]; u8 qh[32] (bit 2 g + 0 / 1 of qh[l]: fifth bit of the low / high nibble weight l of group g); u8 qs[128]}
        //        x = d * sc * q - dmin * mn
        case SrcKind::Q4_K:
        case SrcKind::Q5_K: {
            const bool five = kind == SrcKind::Q5_K;
            const int bs    = five ? 176 : 144;
            for (int64_t blk = 0; blk < n / 256; ++blk) {
                const uint8_t* p  = raw + blk * bs;
                ggml_fp16_t dh, mh;
                memcpy(&dh, p, 2);
                memcpy(&mh, p + 2, 2);
                const float d = ggml_fp16_to_fp32(dh), dmin = ggml_fp16_to_fp32(mh);
                const uint8_t* s12 = p + 4;
                const uint8_t* qh  = p + 16;
                const uint8_t* q   = p + (five ? 48 : 16);
                float* y = dst + blk * 256;
                for (int g = 0; g < 4; ++g, q += 32)
                    for (int part = 0; part < 2; ++part) {
                        const int j = 2 * g + part;
                        int sc, mn;
                        if (j < 4) {
                            sc = s12[j] & 63;
                            mn = s12[j + 4] & 63;
                        } else {
                            sc = (s12[j + 4] & 0xF) | ((s12[j - 4] >> 6) << 4);
                            mn = (s12[j + 4] >> 4) | ((s12[j] >> 6) << 4);
                        }
                        const float dl = d * (float)sc, ml = dmin * (float)mn;
                        for (int l = 0; l < 32; ++l) {
                            int v = part ? (q[l] >> 4) : (q[l] & 0xF);
                            if (five && ((qh[l] >> j) & 1)) v += 16;
                            *y++ = dl * (float)v - ml;
                        }
                    }
            }
            break;
        }
        //   q6_K {u8 ql[128]; u8 qh[64]; i8 scales[16] (per 16 weights); f16 d}: per 128-weight half, weight l + 32 c (c = 0..3) has its low 4 bits in
        //        ql[l + 32 (c & 1)] (low nibble for c < 2, high nibble otherwise) and its high 2 bits at bits 2 c of qh[l];  x = d * scale * (q - 32)
        case SrcKind::Q6_K:
            for (int64_t blk = 0; blk < n / 256; ++blk) {
                const uint8_t* p  = raw + blk * 210;
                const uint8_t* ql = p;
                const uint8_t* qh = p + 128;
                const int8_t* sc  = (const int8_t*)(p + 192);
                ggml_fp16_t dh;
                memcpy(&dh, p + 208, 2);
                const float d = ggml_fp16_to_fp32(dh);
                float* y = dst + blk * 256;
                for (int half = 0; half < 2; ++half, y += 128, ql += 64, qh += 32, sc += 8)
                    for (int l = 0; l < 32; ++l)
                        for (int c = 0; c < 4; ++c) {
                            const int lo = (c < 2) ? (ql[l + 32 * (c & 1)] & 0xF) : (ql[l + 32 * (c & 1)] >> 4);
                            const int q6 = (lo | (((qh[l] >> (2 * c)) & 3) << 4)) - 32;
                            y[l + 32 * c] = d * (float)sc[l / 16 + 2 * c] * (float)q6;
                        }
            }
            break;
        //   iq4_nl {f16 d; u8 qs[16]}: x = d * kvalues[q], element j in the low nibble of byte j, element j + 16 in the high nibble
        case SrcKind::IQ4_NL: {
            static const int8_t kv[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
            for (int64_t blk = 0; blk < n / 32; ++blk) {
                const uint8_t* p = raw + blk * 18;
                ggml_fp16_t dh;
                memcpy(&dh, p, 2);
                const float d = ggml_fp16_to_fp32(dh);
                float* y = dst + blk * 32;
                for (int j = 0; j < 16; ++j) {
                    y[j]      = d * (float)kv[p[2 + j] & 0xF];
                    y[j + 16] = d * (float)kv[p[2 + j] >> 4];
                }
            }
            break;
        }
        default: break;
    }
}

// safetensors: u64 LE header length N, N bytes of JSON {"name": {"dtype": "F16", "shape": [..torch order..], "data_offsets": [b, e]}, ...,
// "__metadata__": {...}}, then the byte buffer the offsets index.
inline bool read_safetensors(const std::string& path, ModelFile& mf) {
    mf.path  = path;
    FILE* f  = fopen(path.c_str(), "rb");
    if (!f) {
        mf.error = "cannot open " + path;
        return false;
    }
    uint64_t hlen = 0;
    if (fread(&hlen, 8, 1, f) != 1 || hlen < 2 || hlen > (1ull << 31)) {
        mf.error = "not a safetensors file (bad header length)";
        fclose(f);
        return false;
    }
    std::string hdr(hlen, '\0');
    if (fread(&hdr[0], 1, hlen, f) != hlen) {
        mf.error = "truncated safetensors header";
        fclose(f);
        return false;
    }
    fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(f);
    fclose(f);
    const uint64_t base = 8 + hlen;
    JsonCursor c{hdr.data(), hdr.data() + hdr.size()};
    if (!c.eat('{')) {
        mf.error = "safetensors header is not a JSON object";
        return false;
    }
    while (c.ok) {
        c.ws();
        if (c.eat('}')) break;
        const std::string key = c.str();
        if (!c.eat(':')) break;
        if (key == "__metadata__") {
            if (c.eat('{')) {
                while (c.ok && !c.eat('}')) {
                    const std::string k = c.str();
                    (void)c.eat(':');
                    c.ws();
                    if (c.p < c.end && *c.p == '"')
                        mf.metadata[k] = c.str();
                    else
                        c.skip();
                    (void)c.eat(',');
                }
            } else {
                c.skip();
            }
        } else {
            FileTensor t;
            t.name = key;
            std::vector<int64_t> shape;
            int64_t b = -1, e = -1;
            bool known = true;
            size_t elem_bytes = 0;
            std::string dtype;
            if (!c.eat('{')) break;
            while (c.ok && !c.eat('}')) {
                const std::string k = c.str();
                (void)c.eat(':');
                if (k == "dtype") {
                    dtype = c.str();
                    known = st_dtype(dtype, t.type, t.kind, elem_bytes);
                } else if (k == "shape") {
                    (void)c.eat('[');
                    while (c.ok && !c.eat(']')) {
                        shape.push_back(c.integer());
                        (void)c.eat(',');
                    }
                } else if (k == "data_offsets") {
                    (void)c.eat('[');
                    b = c.integer();
                    (void)c.eat(',');
                    e = c.integer();
                    (void)c.eat(']');
                } else {
                    c.skip();
                }
                (void)c.eat(',');
            }
            if (shape.size() == 5) {  // the reference folds the two outermost (torch-order) dims of a 5-D tensor: safetensors_io.cpp:277-283
                shape[1] *= shape[0];
                shape.erase(shape.begin());
            }
            if (!known) {
                mf.undecodable[key] = dtype;
            } else if (shape.size() <= 4 && b >= 0 && e >= b && base + (uint64_t)e <= fsize) {
                t.n_dims = (int)shape.size();
                // the byte range must hold exactly prod(shape) elements ("size mismatch for tensor", safetensors_io.cpp:326-351): the loader
                // later reads nelements * type_size bytes out of a buffer of (e - b) bytes
                uint64_t nel = 1;
                bool shape_ok = true;
                for (size_t i = 0; i < shape.size(); ++i) {
                    const int64_t d = shape[i];
                    if (d < 0 || (d > 0 && nel > (1ull << 40) / (uint64_t)d)) shape_ok = false;
                    nel *= (uint64_t)std::max<int64_t>(d, 0);
                    t.ne[shape.size() - 1 - i] = d;  // torch order -> ggml order
                }
                if (!shape_ok || nel * elem_bytes != (uint64_t)(e - b)) {
                    mf.error = "size mismatch for tensor '" + key + "' (" + dtype + ")";
                    return false;
                }
                t.offset = base + (uint64_t)b;
                t.nbytes = (uint64_t)(e - b);
                if (nel > 0) mf.tensors.push_back(t);
            } else {
                mf.error = (shape.size() > 4 ? "invalid tensor '" : "bad shape / data_offsets for tensor '") + key + "'";  // > 5 dims: safetensors_io.cpp:265-268
                return false;
            }
        }
        (void)c.eat(',');
    }
    if (!c.ok) {
        mf.error = "malformed safetensors header";
        return false;
    }
    return true;
}

// GGUF v2/v3: "GGUF", u32 version, u64 n_tensors, u64 n_kv, KV pairs (string key, u32 type, value), tensor infos (string name, u32 n_dims,
// u64 dims[], u32 ggml_type, u64 offset relative to the data section), padding to general.alignment (default 32), data.
inline bool read_gguf(const std::string& path, ModelFile& mf) {
    mf.path = path;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        mf.error = "cannot open " + path;
        return false;
    }
    bool ok    = true;
    auto rd    = [&](void* dst, size_t n) { ok = ok && fread(dst, 1, n, f) == n; };
    auto rstr  = [&]() {
        uint64_t n = 0;
        rd(&n, 8);
        std::string s;
        if (ok && n < (1u << 20)) {
            s.resize(n);
            if (n) rd(&s[0], n);
        } else {
            ok = false;
        }
        return s;
    };
    char magic[4];
    uint32_t version = 0;
    uint64_t nt = 0, nkv = 0;
    rd(magic, 4);
    rd(&version, 4);
    rd(&nt, 8);
    rd(&nkv, 8);
    if (!ok || memcmp(magic, "GGUF", 4) != 0 || version < 2 || version > 3 || nt > (1u << 24) || nkv > (1u << 20)) {
        mf.error = "not a GGUF v2/v3 file";
        fclose(f);
        return false;
    }
    uint64_t alignment = 32;
    static const size_t scalar_size[] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};  // gguf_type 0..12 (8 = string, 9 = array)
    std::function<void(uint32_t, const std::string&)> skip_value = [&](uint32_t ty, const std::string& key) {
        if (ty == 8) {
            const std::string v = rstr();
            mf.metadata[key]    = v;
        } else if (ty == 9) {
            uint32_t et = 0;
            uint64_t n  = 0;
            rd(&et, 4);
            rd(&n, 8);
            for (uint64_t i = 0; ok && i < n; ++i) skip_value(et, std::string());
        } else if (ty < 13 && scalar_size[ty]) {
            uint64_t v = 0;
            rd(&v, scalar_size[ty]);
            if (key == "general.alignment" && ty == 4) alignment = (uint32_t)v;
        } else {
            ok = false;
        }
    };
    for (uint64_t i = 0; ok && i < nkv; ++i) {
        const std::string key = rstr();
        uint32_t ty           = 0;
        rd(&ty, 4);
        skip_value(ty, key);
    }
    for (uint64_t i = 0; ok && i < nt; ++i) {
        FileTensor t;
        t.name     = rstr();
        uint32_t nd = 0, ty = 0;
        rd(&nd, 4);
        if (nd > 4) ok = false;
        for (uint32_t d = 0; ok && d < nd; ++d) {
            uint64_t v = 0;
            rd(&v, 8);
            t.ne[d] = (int64_t)v;
        }
        rd(&ty, 4);
        rd(&t.offset, 8);
        t.n_dims = (int)nd;
        t.type   = (ggml_type)ty;
        mf.tensors.push_back(t);
    }
    if (!ok) {
        mf.error = "truncated or malformed GGUF header";
        fclose(f);
        return false;
    }
    uint64_t data0 = (uint64_t)ftell(f);
    data0          = (data0 + alignment - 1) / alignment * alignment;
    fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(f);
    fclose(f);
    std::vector<FileTensor> keep;
    for (auto& t : mf.tensors) {
        const int ty = (int)t.type;
        SrcKind dk;
        int dblck = 0, dbytes = 0;
        if (gguf_decoded_kind(ty, dk, dblck, dbytes)) {  // q4_1 / q5_0 / q5_1 / K-quants / iq4_nl: decoded to f32 when the tensor is read
            const uint64_t bs = (uint64_t)dbytes;
            uint64_t n        = 1;
            bool dims_ok      = t.ne[0] > 0 && t.ne[0] % dblck == 0 && t.ne[0] < (1ll << 40);
            for (int d = 0; d < 4 && dims_ok; ++d) {
                dims_ok = t.ne[d] > 0 && n <= (1ull << 46) / (uint64_t)t.ne[d];
                n *= (uint64_t)t.ne[d];
            }
            const uint64_t nb = dims_ok ? n / (uint64_t)dblck * bs : 0;
            if (!dims_ok || t.offset > fsize || data0 > fsize - t.offset || nb > fsize - t.offset - data0) {
                mf.error = "tensor '" + t.name + "' has invalid dimensions or lies outside the file";
                return false;
            }
            t.kind   = dk;
            t.type   = GGML_TYPE_F32;  // what decode_src_kind delivers
            t.nbytes = nb;
            t.offset += data0;
            keep.push_back(t);
            continue;
        }
        const bool known = t.type == GGML_TYPE_F32 || t.type == GGML_TYPE_F16 || t.type == GGML_TYPE_BF16 || t.type == GGML_TYPE_Q8_0 || t.type == GGML_TYPE_Q4_0;
        if (!known) {  // IQ types other than iq4_nl, ternary / MX types, integer tensors: not decodable by this build — an ERROR if a declared parameter needs one
            mf.undecodable[t.name] = "ggml type " + std::to_string((int)t.type);
            continue;
        }
        // dims come from the file: positive, ne0 a whole number of blocks, and every product / offset checked for overflow before it is trusted
        const int64_t blck = ggml_blck_size(t.type);
        uint64_t rows      = 1;
        bool dims_ok       = t.ne[0] > 0 && t.ne[0] % blck == 0 && t.ne[0] < (1ll << 40);
        for (int d = 1; d < 4 && dims_ok; ++d) {
            dims_ok = t.ne[d] > 0 && rows <= (1ull << 40) / (uint64_t)t.ne[d];
            rows *= (uint64_t)t.ne[d];
        }
        const uint64_t rs = dims_ok ? (uint64_t)ggml_row_size(t.type, t.ne[0]) : 0;
        if (!dims_ok || rs == 0 || rows > (1ull << 46) / rs || t.offset > fsize || data0 > fsize - t.offset || rs * rows > fsize - t.offset - data0) {
            mf.error = "tensor '" + t.name + "' has invalid dimensions or lies outside the file";
            return false;
        }
        t.nbytes = rs * rows;
        t.offset += data0;
        keep.push_back(t);
    }
    mf.tensors.swap(keep);
    return true;
}

inline bool read_torch_zip(const std::string& path, ModelFile& mf);     // torch_ckpt_io.hpp
inline bool read_torch_legacy(const std::string& path, ModelFile& mf);  // torch_ckpt_io.hpp
inline bool read_model_file(const std::string& path, ModelFile& mf) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        mf.error = "cannot open " + path;
        return false;
    }
    char magic[4] = {0, 0, 0, 0};
    const size_t n = fread(magic, 1, 4, f);
    fclose(f);
    if (n == 4 && memcmp(magic, "GGUF", 4) == 0) return read_gguf(path, mf);
    if (n == 4 && memcmp(magic, "PK\x03\x04", 4) == 0) return read_torch_zip(path, mf);             // torch.save since PyTorch 1.6 (.ckpt / .pt / .pth / .bin)
    if (n == 4 && (unsigned char)magic[0] == 0x80 && magic[1] == 2 && (unsigned char)magic[2] == 0x8a) return read_torch_legacy(path, mf);  // older torch.save: PROTO 2, LONG1 magic
    return read_safetensors(path, mf);
}

}  // namespace sdmi
