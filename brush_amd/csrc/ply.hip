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
struct PlyColumns {
    int16_t xyz[3], scale[3], opacity, rot[4], dc[3];
    int16_t rest[PLY_MAX_REST];
};

// import.rs:279-316 (row visitor) + :57-75 (into_splats defaults) + :128-143 (interleave_coeffs)
__global__ __launch_bounds__(PLY_WG) void ply_unpack_rows_kernel(const float* __restrict__ rows, uint64_t n, uint32_t row_len, uint32_t coeffs,
                                                                PlyColumns c, float* __restrict__ transforms, float* __restrict__ sh,
                                                                float* __restrict__ raw_opac) {
    const uint64_t i = (uint64_t)blockIdx.x * PLY_WG + threadIdx.x;
    if (i >= n) return;
    const float* r = rows + i * row_len;
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
    std::string error;
};

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
    int sh_props = 0;
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
            in_vertex = std::strcmp(name, "vertex") == 0;
            if (in_vertex) {
                if (!first_element) { out.error = "unsupported PLY: the vertex element must come first (SuperSplat-compressed files are not handled)"; return false; }
                seen_vertex = true;
                out.info.num_splats = cnt;
            }
            first_element = false;
        } else if (l.rfind("property ", 0) == 0 && in_vertex) {
            char type[32] = {0}, name[64] = {0};
            if (std::sscanf(l.c_str(), "property %31s %63s", type, name) != 2) { out.error = "bad property line"; return false; }
            if (std::strcmp(type, "float") != 0 && std::strcmp(type, "float32") != 0) {
                out.error = std::string("unsupported PLY: vertex property '") + name + "' has type " + type + " (only float rows are handled)";
                return false;
            }
            const std::string nm = name;
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
    if (!seen_vertex || out.cols.xyz[0] < 0 || out.cols.xyz[1] < 0 || out.cols.xyz[2] < 0) { out.error = "Unknown format"; return false; }  // import.rs:252
    // sh_count = number of f_dc_/f_rest_ properties (import.rs:256-265); degree from count / 3
    const int coeffs = sh_props > 0 ? sh_props / 3 : 1;
    int deg = 0;
    while ((deg + 1) * (deg + 1) < coeffs) ++deg;
    if ((deg + 1) * (deg + 1) != coeffs || deg > 4 || (sh_props % 3) != 0) { out.error = "SH property count is not 3*(d+1)^2"; return false; }
    // the rest columns must be exactly f_rest_0 .. f_rest_{3(C-1)-1}
    for (int k = 0; k < 3 * (coeffs - 1); ++k)
        if (out.cols.rest[k] < 0) { out.error = "f_rest_ properties are not contiguous"; return false; }
    out.info.sh_degree = (uint32_t)deg;
    out.info.row_floats = (uint32_t)col;
    out.info.body_offset = body;
    const uint64_t need = body + out.info.num_splats * (uint64_t)col * 4u;
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
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!bytes) return set_error(ctx, BH_ERR_INVALID_ARG, "splats_from_ply: null buffer");
    ParsedHeader ph;
    if (!parse_header((const uint8_t*)bytes, len, ph))
        return set_error(ctx, ph.error.rfind("unsupported", 0) == 0 ? BH_ERR_UNSUPPORTED : BH_ERR_INVALID_ARG, ph.error);
    const uint64_t n = ph.info.num_splats;
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return set_error(ctx, BH_ERR_UNSUPPORTED, "more than 2^32-1 splats");
    if (!transforms || !sh_coeffs || !raw_opacities) return set_error(ctx, BH_ERR_INVALID_ARG, "splats_from_ply: null output tensor");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t body = n * ph.info.row_floats * 4u;
    auto* rows = (float*)ensure(ctx, SLOT_PLY_ROWS, body);
    if (!rows) return BH_ERR_OOM;
    BH_HIP(ctx, hipMemcpyAsync(rows, (const char*)bytes + ph.info.body_offset, body, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t coeffs = (ph.info.sh_degree + 1) * (ph.info.sh_degree + 1);
    hipLaunchKernelGGL(ply_unpack_rows_kernel, dim3((unsigned)((n + PLY_WG - 1) / PLY_WG)), dim3(PLY_WG), 0, ctx->stream, rows, n, ph.info.row_floats,
                       coeffs, ph.cols, transforms, sh_coeffs, raw_opacities);
    BH_LAUNCH_CHECK(ctx, "ply_unpack_rows_kernel");
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `bytes` may be freed by the caller on return
    return 0;
}

}  // extern "C"
