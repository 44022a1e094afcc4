// ply.hip — splats <-> PLY at the edges of the hot path.
//
// Reference: brush-serde/src/export.rs:86-204 (read_splat_data + splat_to_ply: bake the 3D-filter
// floor, pull three tensors to the host, permute SH to the INRIA [n, channel, coeff] layout,
// normalise quaternions, build one struct per splat, serialise with serde_ply) and
// brush-serde/src/import.rs:172-400 (parse_ply: row visitor -> SplatData -> Splats).
// serde_ply (un-vendored, Cargo.lock) fixes only the container: an ASCII header and a
// little-endian body of `property float` rows — restated here.
//
// MI355X shape: the per-splat row (x y z | scale_0..2 | opacity | rot_0..3 | f_dc_0..2 | f_rest_*)
// is assembled ON THE DEVICE by one kernel that writes the PLY body exactly as it will sit in the
// file (thread per output float -> fully coalesced 4-byte stores; the reads of a row fall in the
// same few cache lines), so the host side of an export is one D2H copy of the finished body behind
// a ~1 KB header — no host-side per-splat loop, no permuted SH temporary.  Import is the mirror:
// header parse on the host (property name -> column), one H2D copy of the body, one kernel that
// scatters columns into transforms [N,10] / sh_coeffs [N,C,3] / raw_opacities [N] with the
// reference's defaults for absent properties.
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "context.h"

namespace bh {

constexpr int PLY_WG = 256;
constexpr int PLY_MAX_REST = 72;  // f_rest_0..71 (SH degree 4), brush-serde-macros sh_field_names

// ---- export ------------------------------------------------------------------------------------
// export.rs:117-176.  row = 14 + 3*(C-1) floats.
__global__ __launch_bounds__(PLY_WG) void ply_pack_rows_kernel(const float* __restrict__ transforms, const float* __restrict__ sh,
                                                              const float* __restrict__ raw_opac, uint64_t n, uint32_t coeffs,
                                                              float* __restrict__ rows) {
    const uint32_t row_len = 14u + 3u * (coeffs - 1u);
    const uint64_t e = (uint64_t)blockIdx.x * PLY_WG + threadIdx.x;
    if (e >= n * row_len) return;
    const uint64_t i = e / row_len;
    const uint32_t j = (uint32_t)(e - i * row_len);
    const float* t = transforms + i * 10;
    float v;
    if (j < 3) {
        v = t[j];                       // x y z
    } else if (j < 6) {
        v = t[7 + (j - 3)];             // scale_0..2 (log-scales)
    } else if (j == 6) {
        v = raw_opac[i];                // opacity (logit)
    } else if (j < 11) {
        // rot_0..3: normalised on export (export.rs:151-158)
        const float r0 = t[3], r1 = t[4], r2 = t[5], r3 = t[6];
        const float rn = __builtin_fmaxf(__builtin_sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3), 1e-12f);
        v = t[3 + (j - 7)] / rn;
    } else if (j < 14) {
        v = sh[i * coeffs * 3 + (j - 11)];  // f_dc_0..2 = coefficient 0, channels r g b
    } else {
        // f_rest: INRIA order [channel][coeff 1..C-1]  (the permute([0,2,1]) of export.rs:91)
        const uint32_t r = j - 14u, per = coeffs - 1u;
        const uint32_t ch = r / per, k = 1u + (r - ch * per);
        v = sh[(i * coeffs + k) * 3 + ch];
    }
    rows[e] = v;
}

// ---- import ------------------------------------------------------------------------------------
// Which file rows a load keeps: output row i = file row first + i * step (subsample_points keeps every s-th row starting at
// s - 1, import.rs:346-349 / 501-506; SplatData::subsample keeps rows 0, step, 2 step, ..., import.rs:49-74).
struct RowPick { uint64_t first, step; };

struct PlyColumns {
    int16_t xyz[3], scale[3], opacity, rot[4], dc[3];
    int16_t rest[PLY_MAX_REST];
};

// import.rs:279-316 (row visitor) + :57-75 (into_splats defaults) + :128-143 (interleave_coeffs)
__global__ __launch_bounds__(PLY_WG) void ply_unpack_rows_kernel(const float* __restrict__ rows, uint64_t n, uint32_t row_len, uint32_t coeffs,
                                                                PlyColumns c, RowPick pick, float* __restrict__ transforms, float* __restrict__ sh,
                                                                float* __restrict__ raw_opac) {
    const uint64_t i = (uint64_t)blockIdx.x * PLY_WG + threadIdx.x;
    if (i >= n) return;
    const float* r = rows + (pick.first + i * pick.step) * row_len;
    float* t = transforms + i * 10;
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = r[c.xyz[k]];
    const bool has_rot = c.rot[0] >= 0;
    t[3] = has_rot ? r[c.rot[0]] : 1.0f;
#pragma unroll
    for (int k = 1; k < 4; ++k) t[3 + k] = has_rot ? (c.rot[k] >= 0 ? r[c.rot[k]] : 0.0f) : 0.0f;
    const bool has_scale = c.scale[0] >= 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) t[7 + k] = has_scale ? (c.scale[k] >= 0 ? r[c.scale[k]] : 0.0f) : -4.0f;
    raw_opac[i] = c.opacity >= 0 ? r[c.opacity] : 0.0f;  // inverse_sigmoid(0.5) = ln(1) = 0
    float* s = sh + i * coeffs * 3;
    const bool has_sh = c.dc[0] >= 0 || c.dc[1] >= 0 || c.dc[2] >= 0 || c.rest[0] >= 0;
    for (int ch = 0; ch < 3; ++ch) s[ch] = has_sh ? (c.dc[ch] >= 0 ? r[c.dc[ch]] : 0.0f) : 0.5f;
    const uint32_t per = coeffs - 1u;
    for (uint32_t k = 0; k < per; ++k)
        for (uint32_t ch = 0; ch < 3; ++ch) {
            const int col = c.rest[ch * per + k];
            s[(1 + k) * 3 + ch] = col >= 0 ? r[col] : 0.0f;
        }
}

// ---- import, SuperSplat / PlayCanvas "compressed.ply" ------------------------------------------------------
// brush-serde/src/import.rs:407-600 (parse_compressed_ply), quant.rs:1-75 (bit layouts), ply_gaussian.rs:24-33,105-119.
// Three elements: `chunk` (one row per 256 splats: min / max of position, log-scale and colour, floats found by NAME),
// `vertex` (four packed u32 per splat) and optionally `sh` (uchar f_rest_k, [channel][coeff] order).  The reference walks
// the rows on the host through serde visitors; here the body crosses PCIe as it is (16 B + K per splat) and one thread per
// splat decodes it into the layouts of the hot path.
struct CompressedLayout {
    uint64_t chunk_off, vert_off, sh_off;      // byte offsets of the three element blocks from the start of the body
    uint32_t chunk_stride, vert_stride, sh_stride;
    uint32_t sh_props;                          // K = number of f_rest_ properties (0: no `sh` element)
    int16_t chunk_col[18];                      // byte offset of min_x .. max_b inside a chunk row (order of kChunkNames)
    int16_t vert_col[4];                        // byte offset of packed_position / _rotation / _scale / _color
    int16_t rest_col[PLY_MAX_REST];             // byte offset of f_rest_k inside an sh row
};
static const char* const kChunkNames[18] = {"min_x", "min_y", "min_z", "max_x", "max_y", "max_z", "min_scale_x", "min_scale_y", "min_scale_z",
                                            "max_scale_x", "max_scale_y", "max_scale_z", "min_r", "min_g", "min_b", "max_r", "max_g", "max_b"};
static const char* const kVertexNames[4] = {"packed_position", "packed_rotation", "packed_scale", "packed_color"};

BH_DEV uint32_t ld_u32(const uint8_t* p) {   // rows may hold properties of other widths in front: no alignment assumed
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
BH_DEV float unpack_unorm(uint32_t packed, uint32_t bits) { return (float)packed / (float)((1u << bits) - 1u); }   // quant.rs:4-7

__global__ __launch_bounds__(PLY_WG) void ply_decode_compressed_kernel(const uint8_t* __restrict__ body, CompressedLayout L, uint64_t n, uint32_t coeffs,
                                                                      RowPick pick, float* __restrict__ transforms, float* __restrict__ sh,
                                                                      float* __restrict__ raw_opac) {
    const uint64_t i = (uint64_t)blockIdx.x * PLY_WG + threadIdx.x;
    if (i >= n) return;
    const uint64_t row = pick.first + i * pick.step;
    const uint8_t* ch = body + L.chunk_off + (row / 256u) * L.chunk_stride;   // import.rs:503: the chunk of the FILE row
    const uint8_t* vr = body + L.vert_off + row * L.vert_stride;
    float cm[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) cm[k] = u2f(ld_u32(ch + L.chunk_col[k]));
    const uint32_t ppos = ld_u32(vr + L.vert_col[0]), prot = ld_u32(vr + L.vert_col[1]);
    const uint32_t pscl = ld_u32(vr + L.vert_col[2]), pcol = ld_u32(vr + L.vert_col[3]);
    float* t = transforms + i * 10;
    // decode_vec_11_10_11 (quant.rs:9-18), then raw * (max - min) + min (import.rs:435-445)
    const float p3[3] = {unpack_unorm((ppos >> 21) & 0x7FFu, 11), unpack_unorm((ppos >> 11) & 0x3FFu, 10), unpack_unorm(ppos & 0x7FFu, 11)};
    const float s3[3] = {unpack_unorm((pscl >> 21) & 0x7FFu, 11), unpack_unorm((pscl >> 11) & 0x3FFu, 10), unpack_unorm(pscl & 0x7FFu, 11)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        t[k] = p3[k] * (cm[3 + k] - cm[k]) + cm[k];
        t[7 + k] = s3[k] * (cm[9 + k] - cm[6 + k]) + cm[6 + k];
    }
    // decode_quat (quat.rs:38-70): the largest component was dropped; scalar order (w, x, y, z) (import.rs:511-517)
    {
        const uint32_t largest = (prot >> 30) & 3u;
        const float norm = 0.5f * 1.41421356237309504880f;
        const float a = (unpack_unorm((prot >> 20) & 0x3FFu, 10) - 0.5f) / norm;
        const float b = (unpack_unorm((prot >> 10) & 0x3FFu, 10) - 0.5f) / norm;
        const float c = (unpack_unorm(prot & 0x3FFu, 10) - 0.5f) / norm;
        const float m = __builtin_sqrtf(1.0f - ((a * a + b * b) + c * c));
        const float vals[3] = {a, b, c};
        int ind = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (k == largest) t[3 + k] = m;
            else { t[3 + k] = vals[ind]; ++ind; }
        }
    }
    // decode_vec_8_8_8_8 (quant.rs:20-36): post-activation opacity -> logit, RGB -> SH DC (import.rs:518-522)
    const float cr = unpack_unorm((pcol >> 24) & 0xFFu, 8), cg = unpack_unorm((pcol >> 16) & 0xFFu, 8), cb = unpack_unorm((pcol >> 8) & 0xFFu, 8);
    const float al = unpack_unorm(pcol & 0xFFu, 8);
    raw_opac[i] = bh_logf(al / (1.0f - al));                                  // inverse_sigmoid (gaussian_splats.rs:76-78)
    const float rgb[3] = {cr, cg, cb};
    float* so = sh + i * coeffs * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) so[k] = ((rgb[k] * (cm[15 + k] - cm[12 + k]) + cm[12 + k]) - 0.5f) / 0.2820948f;   // rgb_to_sh (sh.rs:21-31)
    if (L.sh_props) {   // de_quant_sh (ply_gaussian.rs:105-111) + interleave_coeffs (import.rs:133-144)
        const uint8_t* sr = body + L.sh_off + row * L.sh_stride;
        const uint32_t per = L.sh_props / 3u;
        for (uint32_t c3 = 0; c3 < 3u; ++c3)
            for (uint32_t k = 0; k < per; ++k) {
                const float q = (float)sr[L.rest_col[c3 * per + k]] / 254.0f;
                so[(1u + k) * 3u + c3] = (q - 0.5f) * 8.0f;
            }
    }
}

// ---- import, vertex rows that are not all float ---------------------------------------------------------------
// The reference's row struct holds f32 fields and serde casts whatever scalar type the file carries (`as f32`); the colour
// override red / green / blue (aliases r g b) goes through de_quant (ply_gaussian.rs:36-58: f32 as is, u8 / 254, u16 / 65534)
// and, when all three are there, replaces the DC term by rgb_to_sh (import.rs:349-358) — point clouds with uchar colours.
enum PlyScalar : uint8_t { PS_F32 = 0, PS_F64, PS_I8, PS_U8, PS_I16, PS_U16, PS_I32, PS_U32, PS_NONE = 0xFF };
constexpr int PLY_SLOT_XYZ = 0, PLY_SLOT_SCALE = 3, PLY_SLOT_OPACITY = 6, PLY_SLOT_ROT = 7, PLY_SLOT_DC = 11, PLY_SLOT_REST = 14;
constexpr int PLY_SLOT_RGB = PLY_SLOT_REST + PLY_MAX_REST, PLY_SLOTS = PLY_SLOT_RGB + 3;
struct PlyByteCols {
    int32_t off[PLY_SLOTS];    // byte offset inside a row, -1 = absent
    uint8_t type[PLY_SLOTS];
    uint32_t stride;
};
BH_DEV float ld_scalar(const uint8_t* p, uint32_t type) {
    switch (type) {
        case PS_F32: return u2f(ld_u32(p));
        case PS_F64: return (float)__longlong_as_double((long long)((uint64_t)ld_u32(p) | ((uint64_t)ld_u32(p + 4) << 32)));
        case PS_I8: return (float)(int8_t)p[0];
        case PS_U8: return (float)p[0];
        case PS_I16: return (float)(int16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
        case PS_U16: return (float)(uint16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
        case PS_I32: return (float)(int32_t)ld_u32(p);
        default: return (float)ld_u32(p);
    }
}
__global__ __launch_bounds__(PLY_WG) void ply_unpack_mixed_rows_kernel(const uint8_t* __restrict__ rows, uint64_t n, uint32_t coeffs, PlyByteCols c,
                                                                      RowPick pick, float* __restrict__ transforms, float* __restrict__ sh,
                                                                      float* __restrict__ raw_opac) {
    const uint64_t i = (uint64_t)blockIdx.x * PLY_WG + threadIdx.x;
    if (i >= n) return;
    const uint8_t* r = rows + (pick.first + i * pick.step) * c.stride;
    auto get = [&](int slot, float dflt) { return c.off[slot] >= 0 ? ld_scalar(r + c.off[slot], c.type[slot]) : dflt; };
    float* t = transforms + i * 10;
    for (int k = 0; k < 3; ++k) t[k] = get(PLY_SLOT_XYZ + k, 0.0f);
    const bool has_rot = c.off[PLY_SLOT_ROT] >= 0, has_scale = c.off[PLY_SLOT_SCALE] >= 0;
    for (int k = 0; k < 4; ++k) t[3 + k] = has_rot ? get(PLY_SLOT_ROT + k, 0.0f) : (k == 0 ? 1.0f : 0.0f);
    for (int k = 0; k < 3; ++k) t[7 + k] = has_scale ? get(PLY_SLOT_SCALE + k, 0.0f) : -4.0f;
    raw_opac[i] = get(PLY_SLOT_OPACITY, 0.0f);
    float* s = sh + i * coeffs * 3;
    const bool has_rgb = c.off[PLY_SLOT_RGB] >= 0;
    const bool has_sh = has_rgb || c.off[PLY_SLOT_DC] >= 0 || c.off[PLY_SLOT_DC + 1] >= 0 || c.off[PLY_SLOT_DC + 2] >= 0 || c.off[PLY_SLOT_REST] >= 0;
    for (int ch = 0; ch < 3; ++ch) {
        if (has_rgb) {
            const uint32_t ty = c.type[PLY_SLOT_RGB + ch];
            const float raw = ld_scalar(r + c.off[PLY_SLOT_RGB + ch], ty);
            const float v = ty == PS_U8 ? raw / 254.0f : (ty == PS_U16 ? raw / 65534.0f : raw);
            s[ch] = (v - 0.5f) / 0.2820948f;
        } else {
            s[ch] = has_sh ? get(PLY_SLOT_DC + ch, 0.0f) : 0.5f;
        }
    }
    const uint32_t per = coeffs - 1u;
    for (uint32_t ch = 0; ch < 3u; ++ch)
        for (uint32_t k = 0; k < per; ++k) s[(1u + k) * 3u + ch] = get(PLY_SLOT_REST + (int)(ch * per + k), 0.0f);
}

// ---- host: header text ---------------------------------------------------------------------------
static std::string f32_display(float v) {  // Rust's `{}` for f32: shortest round-trip, never exponent form
    char buf[96];
    if (v != v) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

static std::string ply_header(uint64_t n, uint32_t sh_degree, bool render_mip, const float* up_axis) {
    std::string h = "ply\nformat binary_little_endian 1.0\n";
    h += "comment Exported from Brush\n";                       // export.rs:188
    if (up_axis) h += "comment Vertical axis: " + f32_display(up_axis[0]) + " " + f32_display(up_axis[1]) + " " + f32_display(up_axis[2]) + "\n";
    else h += "comment Vertical axis: y\n";                      // export.rs:189-193
    h += "comment SH degree: " + std::to_string(sh_degree) + "\n";
    h += std::string("comment SplatRenderMode: ") + (render_mip ? "mip" : "default") + "\n";
    h += "element vertex " + std::to_string(n) + "\n";
    static const char* core[] = {"x", "y", "z", "scale_0", "scale_1", "scale_2", "opacity", "rot_0", "rot_1", "rot_2", "rot_3", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (const char* p : core) h += std::string("property float ") + p + "\n";
    const uint32_t rest = 3u * ((sh_degree + 1u) * (sh_degree + 1u) - 1u);
    for (uint32_t k = 0; k < rest; ++k) h += "property float f_rest_" + std::to_string(k) + "\n";
    h += "end_header\n";
    return h;
}

// ---- host: header parse ----------------------------------------------------------------------------
static std::string lower(std::string s) {
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}
static std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

struct ParsedHeader {
    BhPlyInfo info{};
    PlyColumns cols{};
    CompressedLayout comp{};
    PlyByteCols bytes{};
    bool mixed = false;   // some vertex property is not a float, or a colour override is present: the byte-offset path
    std::string error;
};

struct PlyProp { std::string type, name; };
struct PlyElem { std::string name; uint64_t count = 0; std::vector<PlyProp> props; };

static uint32_t ply_type_size(const std::string& t) {
    if (t == "float" || t == "float32" || t == "uint" || t == "uint32" || t == "int" || t == "int32") return 4;
    if (t == "uchar" || t == "uint8" || t == "char" || t == "int8") return 1;
    if (t == "ushort" || t == "uint16" || t == "short" || t == "int16") return 2;
    if (t == "double" || t == "float64") return 8;
    return 0;
}

// `chunk` first: a SuperSplat-compressed file (import.rs:244-250).  Fills out.comp / out.info; body = offset of the first row.
static bool parse_compressed(const std::vector<PlyElem>& elems, uint64_t body, uint64_t len, ParsedHeader& out) {
    if (elems.size() < 2 || elems[1].name != "vertex") { out.error = "Unknown format"; return false; }   // import.rs:474-476
    CompressedLayout& L = out.comp;
    std::memset(L.chunk_col, 0xFF, sizeof L.chunk_col);
    std::memset(L.vert_col, 0xFF, sizeof L.vert_col);
    std::memset(L.rest_col, 0xFF, sizeof L.rest_col);
    auto stride_of = [&](const PlyElem& e, uint32_t& stride) {
        stride = 0;
        for (const PlyProp& p : e.props) {
            const uint32_t sz = ply_type_size(p.type);
            if (!sz) { out.error = "unsupported PLY: property '" + p.name + "' has type " + p.type; return false; }
            stride += sz;
        }
        if (stride == 0 || stride > 32000) { out.error = "bad element row size"; return false; }
        return true;
    };
    if (!stride_of(elems[0], L.chunk_stride) || !stride_of(elems[1], L.vert_stride)) return false;
    uint32_t off = 0;
    for (const PlyProp& p : elems[0].props) {
        for (int k = 0; k < 18; ++k)
            if (p.name == kChunkNames[k]) {
                if (ply_type_size(p.type) != 4 || p.type[0] != 'f') { out.error = "unsupported PLY: chunk property '" + p.name + "' is not a float"; return false; }
                L.chunk_col[k] = (int16_t)off;
            }
        off += ply_type_size(p.type);
    }
    for (int k = 0; k < 18; ++k)   // QuantMeta's fields are not optional (import.rs:417-436)
        if (L.chunk_col[k] < 0) { out.error = std::string("compressed PLY: chunk property '") + kChunkNames[k] + "' is missing"; return false; }
    off = 0;
    for (const PlyProp& p : elems[1].props) {
        for (int k = 0; k < 4; ++k)
            if (p.name == kVertexNames[k]) {
                if (p.type != "uint" && p.type != "uint32") { out.error = "unsupported PLY: vertex property '" + p.name + "' is not a uint"; return false; }
                L.vert_col[k] = (int16_t)off;
            }
        off += ply_type_size(p.type);
    }
    for (int k = 0; k < 4; ++k)
        if (L.vert_col[k] < 0) { out.error = std::string("compressed PLY: vertex property '") + kVertexNames[k] + "' is missing"; return false; }
    const uint64_t n = elems[1].count;
    if (elems[0].count * 256ull < n) { out.error = "compressed PLY: fewer chunks than ceil(vertices / 256)"; return false; }
    L.chunk_off = 0;
    L.vert_off = elems[0].count * (uint64_t)L.chunk_stride;
    L.sh_off = L.vert_off + n * (uint64_t)L.vert_stride;
    uint64_t total = L.sh_off;
    int coeffs = 1;
    L.sh_props = 0;
    if (elems.size() > 2 && elems[2].name == "sh") {   // import.rs:492-497: the third element holds the higher SH bands
        const PlyElem& e = elems[2];
        if (e.count != n) { out.error = "compressed PLY: sh rows != vertex rows"; return false; }
        if (!stride_of(e, L.sh_stride)) return false;
        off = 0;
        int found = 0;
        for (const PlyProp& p : e.props) {
            if (p.name.rfind("f_rest_", 0) == 0) {
                const int k = std::atoi(p.name.c_str() + 7);
                if (ply_type_size(p.type) != 1) { out.error = "unsupported PLY: sh property '" + p.name + "' is not a uchar"; return false; }
                if (k >= 0 && k < PLY_MAX_REST && L.rest_col[k] < 0) { L.rest_col[k] = (int16_t)off; ++found; }
            }
            off += ply_type_size(p.type);
        }
        if (found == 0 || found % 3 != 0) { out.error = "SH property count is not 3*((d+1)^2 - 1)"; return false; }
        coeffs = found / 3 + 1;
        for (int k = 0; k < found; ++k)
            if (L.rest_col[k] < 0) { out.error = "f_rest_ properties are not contiguous"; return false; }
        L.sh_props = (uint32_t)found;
        total += n * (uint64_t)L.sh_stride;
    }
    int deg = 0;
    while ((deg + 1) * (deg + 1) < coeffs) ++deg;
    if ((deg + 1) * (deg + 1) != coeffs || deg > 4) { out.error = "SH property count is not 3*((d+1)^2 - 1)"; return false; }
    out.info.num_splats = n;
    out.info.sh_degree = (uint32_t)deg;
    out.info.row_floats = 0;
    out.info.body_offset = body;
    out.info.compressed = 1;
    if (body + total > len) { out.error = "Unexpected EOF"; return false; }
    return true;
}

static bool parse_header(const uint8_t* bytes, uint64_t len, ParsedHeader& out) {
    std::memset(&out.cols, 0xFF, sizeof out.cols);  // every column = -1
    out.info.render_mode = -1;
    // find "end_header\n"
    const char* key = "end_header";
    uint64_t end = UINT64_MAX;
    const uint64_t scan = len < (1u << 20) ? len : (1u << 20);
    for (uint64_t i = 0; i + 10 <= scan; ++i)
        if (std::memcmp(bytes + i, key, 10) == 0 && (i == 0 || bytes[i - 1] == '\n')) { end = i; break; }
    if (end == UINT64_MAX) { out.error = "missing PLY header"; return false; }
    uint64_t body = end + 10;
    if (body < len && bytes[body] == '\r') ++body;
    if (body < len && bytes[body] == '\n') ++body;
    const std::string text((const char*)bytes, end);
    std::vector<std::string> lines;
    for (size_t a = 0; a < text.size();) {
        size_t b = text.find('\n', a);
        if (b == std::string::npos) b = text.size();
        lines.push_back(trim(text.substr(a, b - a)));
        a = b + 1;
    }
    if (lines.empty() || lines[0] != "ply") { out.error = "not a PLY file"; return false; }
    bool binary_le = false, in_vertex = false, seen_vertex = false, first_element = true;
    int col = 0;
    int sh_props = 0, rgb_props = 0;
    uint32_t byte_off = 0;
    std::memset(out.bytes.off, 0xFF, sizeof out.bytes.off);
    std::vector<PlyElem> elems;
    bool compressed = false;
    for (size_t li = 1; li < lines.size(); ++li) {
        const std::string& l = lines[li];
        if (l.rfind("format ", 0) == 0) {
            binary_le = l.find("binary_little_endian") != std::string::npos;
        } else if (l.rfind("comment", 0) == 0) {
            const std::string c = lower(trim(l.substr(7)));
            if (c.rfind("vertical axis: ", 0) == 0) {  // import.rs:195-222 (last one wins)
                const std::string s = trim(c.substr(15));
                float v[3];
                bool ok = true;
                if (s == "x") { v[0] = 1; v[1] = 0; v[2] = 0; }
                else if (s == "y") { v[0] = 0; v[1] = -1; v[2] = 0; }
                else if (s == "z") { v[0] = 0; v[1] = 0; v[2] = -1; }
                else {
                    int cnt = 0;
                    std::string tok;
                    auto flush = [&]() {
                        if (tok.empty()) return;
                        char* e = nullptr;
                        const float f = std::strtof(tok.c_str(), &e);
                        if (e && *e == 0) { if (cnt < 3) v[cnt] = f; ++cnt; }
                        tok.clear();
                    };
                    for (char ch : s) {
                        if (ch == ',' || std::isspace((unsigned char)ch) || ch == '[' || ch == ']') flush();
                        else tok += ch;
                    }
                    flush();
                    ok = cnt == 3;
                }
                if (ok) { out.info.has_up_axis = 1; out.info.up_axis[0] = v[0]; out.info.up_axis[1] = v[1]; out.info.up_axis[2] = v[2]; }
            } else if (c.rfind("splatrendermode: ", 0) == 0) {  // import.rs:224-238
                const std::string s = trim(c.substr(17));
                if (s == "mip") out.info.render_mode = 1;
                else if (s == "default") out.info.render_mode = 0;
            }
        } else if (l.rfind("element ", 0) == 0) {
            char name[64] = {0};
            unsigned long long cnt = 0;
            if (std::sscanf(l.c_str(), "element %63s %llu", name, &cnt) != 2) { out.error = "bad element line"; return false; }
            elems.push_back(PlyElem{name, cnt, {}});
            if (first_element && std::strcmp(name, "chunk") == 0) compressed = true;   // import.rs:244-250
            in_vertex = !compressed && std::strcmp(name, "vertex") == 0;
            if (in_vertex) {
                if (!first_element) { out.error = "unsupported PLY: the vertex element must come first"; return false; }
                seen_vertex = true;
                out.info.num_splats = cnt;
            }
            first_element = false;
        } else if (l.rfind("property ", 0) == 0 && compressed) {
            char type[32] = {0}, name[64] = {0};
            if (std::sscanf(l.c_str(), "property %31s %63s", type, name) != 2 || elems.empty()) { out.error = "bad property line"; return false; }
            elems.back().props.push_back(PlyProp{type, name});
        } else if (l.rfind("property ", 0) == 0 && in_vertex) {
            char type[32] = {0}, name[64] = {0};
            if (std::sscanf(l.c_str(), "property %31s %63s", type, name) != 2) { out.error = "bad property line"; return false; }
            const std::string ty = type, nm = name;
            uint8_t code = PS_NONE;
            if (ty == "float" || ty == "float32") code = PS_F32;
            else if (ty == "double" || ty == "float64") code = PS_F64;
            else if (ty == "char" || ty == "int8") code = PS_I8;
            else if (ty == "uchar" || ty == "uint8") code = PS_U8;
            else if (ty == "short" || ty == "int16") code = PS_I16;
            else if (ty == "ushort" || ty == "uint16") code = PS_U16;
            else if (ty == "int" || ty == "int32") code = PS_I32;
            else if (ty == "uint" || ty == "uint32") code = PS_U32;
            if (code == PS_NONE) {
                out.error = std::string("unsupported PLY: vertex property '") + name + "' has type " + type + " (scalar properties only)";
                return false;
            }
            if (code != PS_F32) out.mixed = true;
            int slot = -1;
            if (nm == "x") slot = PLY_SLOT_XYZ; else if (nm == "y") slot = PLY_SLOT_XYZ + 1; else if (nm == "z") slot = PLY_SLOT_XYZ + 2;
            else if (nm == "scale_0") slot = PLY_SLOT_SCALE; else if (nm == "scale_1") slot = PLY_SLOT_SCALE + 1; else if (nm == "scale_2") slot = PLY_SLOT_SCALE + 2;
            else if (nm == "opacity") slot = PLY_SLOT_OPACITY;
            else if (nm == "rot_0") slot = PLY_SLOT_ROT; else if (nm == "rot_1") slot = PLY_SLOT_ROT + 1; else if (nm == "rot_2") slot = PLY_SLOT_ROT + 2;
            else if (nm == "rot_3") slot = PLY_SLOT_ROT + 3;
            else if (nm == "f_dc_0") slot = PLY_SLOT_DC; else if (nm == "f_dc_1") slot = PLY_SLOT_DC + 1; else if (nm == "f_dc_2") slot = PLY_SLOT_DC + 2;
            else if (nm.rfind("f_rest_", 0) == 0) {
                const int k = std::atoi(nm.c_str() + 7);
                if (k >= 0 && k < PLY_MAX_REST) slot = PLY_SLOT_REST + k;
            } else if (nm == "red" || nm == "r" || nm == "green" || nm == "g" || nm == "blue" || nm == "b") {
                slot = PLY_SLOT_RGB + (nm[0] == 'r' ? 0 : nm[0] == 'g' ? 1 : 2);
                if (out.bytes.off[slot] >= 0) { out.error = "duplicate field " + nm; return false; }
                if (code != PS_F32 && code != PS_U8 && code != PS_U16) { out.error = "invalid type: a quantized value or a float expected for " + nm; return false; }
                ++rgb_props;
                out.mixed = true;
            }
            if (slot >= 0) { out.bytes.off[slot] = (int32_t)byte_off; out.bytes.type[slot] = code; }
            byte_off += code == PS_F64 ? 8u : (code == PS_I8 || code == PS_U8) ? 1u : (code == PS_I16 || code == PS_U16) ? 2u : 4u;
            PlyColumns& c = out.cols;
            if (nm == "x") c.xyz[0] = col; else if (nm == "y") c.xyz[1] = col; else if (nm == "z") c.xyz[2] = col;
            else if (nm == "scale_0") c.scale[0] = col; else if (nm == "scale_1") c.scale[1] = col; else if (nm == "scale_2") c.scale[2] = col;
            else if (nm == "opacity") c.opacity = col;
            else if (nm == "rot_0") c.rot[0] = col; else if (nm == "rot_1") c.rot[1] = col; else if (nm == "rot_2") c.rot[2] = col; else if (nm == "rot_3") c.rot[3] = col;
            else if (nm == "f_dc_0") { c.dc[0] = col; ++sh_props; } else if (nm == "f_dc_1") { c.dc[1] = col; ++sh_props; } else if (nm == "f_dc_2") { c.dc[2] = col; ++sh_props; }
            else if (nm.rfind("f_rest_", 0) == 0) {
                const int k = std::atoi(nm.c_str() + 7);
                if (k >= 0 && k < PLY_MAX_REST) c.rest[k] = col;
                ++sh_props;
            }
            ++col;
            if (col > 32000) { out.error = "too many vertex properties"; return false; }
        }
    }
    if (!binary_le) { out.error = "unsupported PLY: only binary_little_endian is handled"; return false; }
    if (compressed) return parse_compressed(elems, body, len, out);
    if (!seen_vertex || out.cols.xyz[0] < 0 || out.cols.xyz[1] < 0 || out.cols.xyz[2] < 0) { out.error = "Unknown format"; return false; }  // import.rs:252
    // sh_count = number of f_dc_ / f_rest_ / colour properties (import.rs:317-325); degree from count / 3.  A colour override
    // next to SH properties makes the reference's own coefficient count (1 + (sh_count - 3) / 3) a non-square: an error here.
    if (rgb_props != 0 && (rgb_props != 3 || sh_props != 0)) { out.error = "colour override next to SH properties / incomplete colour override"; return false; }
    sh_props += rgb_props;
    const int coeffs = sh_props > 0 ? sh_props / 3 : 1;
    int deg = 0;
    while ((deg + 1) * (deg + 1) < coeffs) ++deg;
    if ((deg + 1) * (deg + 1) != coeffs || deg > 4 || (sh_props % 3) != 0) { out.error = "SH property count is not 3*(d+1)^2"; return false; }
    // the rest columns must be exactly f_rest_0 .. f_rest_{3(C-1)-1}
    for (int k = 0; k < 3 * (coeffs - 1); ++k)
        if (out.cols.rest[k] < 0) { out.error = "f_rest_ properties are not contiguous"; return false; }
    out.info.sh_degree = (uint32_t)deg;
    out.info.row_floats = out.mixed ? 0u : (uint32_t)col;
    out.info.body_offset = body;
    out.bytes.stride = byte_off;
    const uint64_t need = body + out.info.num_splats * (uint64_t)byte_off;
    if (need > len) { out.error = "Unexpected EOF"; return false; }
    return true;
}

}  // namespace bh

using namespace bh;

extern "C" {

int bh_splat_to_ply(bh_ctx* ctx, const float* transforms, const float* sh_coeffs, const float* raw_opacities, const float* min_scale,
                    uint32_t n, uint32_t sh_degree, int render_mip, const float* up_axis, void* out, uint64_t cap, uint64_t* written) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!written) return set_error(ctx, BH_ERR_INVALID_ARG, "splat_to_ply: null size pointer");
    if (sh_degree > 4) return set_error(ctx, BH_ERR_INVALID_ARG, "sh_degree must be 0..4");
    const uint32_t coeffs = (sh_degree + 1) * (sh_degree + 1);
    const uint32_t row_len = 14u + 3u * (coeffs - 1u);
    const std::string header = ply_header(n, sh_degree, render_mip != 0, up_axis);
    const uint64_t body = (uint64_t)n * row_len * 4u;
    *written = header.size() + body;
    if (!out) return 0;  // size query
    if (cap < *written) return set_error(ctx, BH_ERR_INVALID_ARG, "splat_to_ply: output buffer too small");
    if (n > 0 && (!transforms || !sh_coeffs || !raw_opacities)) return set_error(ctx, BH_ERR_INVALID_ARG, "splat_to_ply: null splat tensor");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    std::memcpy(out, header.data(), header.size());
    if (n == 0) return 0;
    // export.rs:183: bake the 3D-filter floor so the file holds ordinary derived scales / opacity
    const float* t = transforms;
    const float* o = raw_opacities;
    if (min_scale) {
        auto* ft = (float*)ensure(ctx, SLOT_FOLDED_TRANSFORMS, (size_t)n * 10 * 4);
        auto* fo = (float*)ensure(ctx, SLOT_FOLDED_RAW_OPAC, (size_t)n * 4);
        if (!ft || !fo) return BH_ERR_OOM;
        BH_TRY(launch_fold_min_scale(ctx, transforms, raw_opacities, min_scale, n, ft, fo));
        t = ft;
        o = fo;
    }
    auto* rows = (float*)ensure(ctx, SLOT_PLY_ROWS, body);
    if (!rows) return BH_ERR_OOM;
    const uint64_t total = (uint64_t)n * row_len;
    hipLaunchKernelGGL(ply_pack_rows_kernel, dim3((unsigned)((total + PLY_WG - 1) / PLY_WG)), dim3(PLY_WG), 0, ctx->stream, t, sh_coeffs, o,
                       (uint64_t)n, coeffs, rows);
    BH_LAUNCH_CHECK(ctx, "ply_pack_rows_kernel");
    BH_HIP(ctx, hipMemcpyAsync((char*)out + header.size(), rows, body, hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int bh_ply_parse_header(const void* bytes, uint64_t len, BhPlyInfo* info) {
    if (!bytes || !info) return BH_ERR_INVALID_ARG;
    ParsedHeader ph;
    if (!parse_header((const uint8_t*)bytes, len, ph)) return ph.error.rfind("unsupported", 0) == 0 ? BH_ERR_UNSUPPORTED : BH_ERR_INVALID_ARG;
    *info = ph.info;
    return 0;
}

int bh_splats_from_ply(bh_ctx* ctx, const void* bytes, uint64_t len, float* transforms, float* sh_coeffs, float* raw_opacities) {
    return bh_splats_from_ply_strided(ctx, bytes, len, 0, 1, UINT64_MAX, transforms, sh_coeffs, raw_opacities);
}

int bh_splats_from_ply_strided(bh_ctx* ctx, const void* bytes, uint64_t len, uint64_t first, uint64_t step, uint64_t count, float* transforms,
                               float* sh_coeffs, float* raw_opacities) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!bytes) return set_error(ctx, BH_ERR_INVALID_ARG, "splats_from_ply: null buffer");
    ParsedHeader ph;
    if (!parse_header((const uint8_t*)bytes, len, ph))
        return set_error(ctx, ph.error.rfind("unsupported", 0) == 0 ? BH_ERR_UNSUPPORTED : BH_ERR_INVALID_ARG, ph.error);
    const uint64_t rows_in_file = ph.info.num_splats;
    if (rows_in_file > 0xFFFFFFFFull) return set_error(ctx, BH_ERR_UNSUPPORTED, "more than 2^32-1 splats");
    if (step == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "splats_from_ply: step must be >= 1");
    const uint64_t avail = first < rows_in_file ? (rows_in_file - first + step - 1) / step : 0;
    if (count == UINT64_MAX) count = avail;
    if (count > avail) return set_error(ctx, BH_ERR_INVALID_ARG, "splats_from_ply: first + (count - 1) * step is past the last row");
    const RowPick pick{first, step};
    const uint64_t n = count;             // rows decoded (the H2D copy below still carries the whole body: one DMA, no host-side gather)
    if (n == 0) return 0;
    if (!transforms || !sh_coeffs || !raw_opacities) return set_error(ctx, BH_ERR_INVALID_ARG, "splats_from_ply: null output tensor");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ph.info.compressed) {
        const CompressedLayout& L = ph.comp;
        const uint64_t bytes_body = L.sh_off + (L.sh_props ? rows_in_file * (uint64_t)L.sh_stride : 0ull);
        auto* dev_body = (uint8_t*)ensure(ctx, SLOT_PLY_ROWS, bytes_body);
        if (!dev_body) return BH_ERR_OOM;
        BH_HIP(ctx, hipMemcpyAsync(dev_body, (const char*)bytes + ph.info.body_offset, bytes_body, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t coeffs = (ph.info.sh_degree + 1) * (ph.info.sh_degree + 1);
        hipLaunchKernelGGL(ply_decode_compressed_kernel, dim3((unsigned)((n + PLY_WG - 1) / PLY_WG)), dim3(PLY_WG), 0, ctx->stream, dev_body, L, n, coeffs,
                           pick, transforms, sh_coeffs, raw_opacities);
        BH_LAUNCH_CHECK(ctx, "ply_decode_compressed_kernel");
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    if (ph.mixed) {
        const uint64_t bytes_body = rows_in_file * (uint64_t)ph.bytes.stride;
        auto* dev_body = (uint8_t*)ensure(ctx, SLOT_PLY_ROWS, bytes_body);
        if (!dev_body) return BH_ERR_OOM;
        BH_HIP(ctx, hipMemcpyAsync(dev_body, (const char*)bytes + ph.info.body_offset, bytes_body, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t coeffs = (ph.info.sh_degree + 1) * (ph.info.sh_degree + 1);
        hipLaunchKernelGGL(ply_unpack_mixed_rows_kernel, dim3((unsigned)((n + PLY_WG - 1) / PLY_WG)), dim3(PLY_WG), 0, ctx->stream, dev_body, n, coeffs,
                           ph.bytes, pick, transforms, sh_coeffs, raw_opacities);
        BH_LAUNCH_CHECK(ctx, "ply_unpack_mixed_rows_kernel");
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    const uint64_t body = rows_in_file * ph.info.row_floats * 4u;
    auto* rows = (float*)ensure(ctx, SLOT_PLY_ROWS, body);
    if (!rows) return BH_ERR_OOM;
    BH_HIP(ctx, hipMemcpyAsync(rows, (const char*)bytes + ph.info.body_offset, body, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t coeffs = (ph.info.sh_degree + 1) * (ph.info.sh_degree + 1);
    hipLaunchKernelGGL(ply_unpack_rows_kernel, dim3((unsigned)((n + PLY_WG - 1) / PLY_WG)), dim3(PLY_WG), 0, ctx->stream, rows, n, ph.info.row_floats,
                       coeffs, ph.cols, pick, transforms, sh_coeffs, raw_opacities);
    BH_LAUNCH_CHECK(ctx, "ply_unpack_rows_kernel");
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `bytes` may be freed by the caller on return
    return 0;
}

}  // extern "C"
