// brush_hip.hpp — C++17 host-side mirror of Brush's operator surface over the C ABI of libbrush_hip.so.
//
// The reference's host language is Rust (not available in this image); where the reference is compiled code
// the host side above the C ABI is C++: this header restates, with the reference's names and argument
// meaning, what a Brush host sees (paths relative to /root/reference/crates/):
//
//   brush_hip::Camera            brush-render/src/camera.rs:12-58          (+ CameraModel, kernels/camera_model/mod.rs:31-38)
//   brush_hip::Splats            brush-render/src/gaussian_splats.rs:62-256 (transforms / sh_coeffs / raw_opacities / min_scale)
//   brush_hip::RasterPass        brush-render/src/gaussian_splats.rs:28-48
//   brush_hip::render_splats     brush-render/src/gaussian_splats.rs:365-446 -> (image, RenderAux)
//   brush_hip::render_splats_bwd brush-render/src/bwd/burn_glue.rs:223-311   (+ RenderBackwards::backward :121-182)
//   brush_hip::radix_argsort     brush-sort/src/lib.rs:16,  brush_hip::prefix_sum  brush-prefix-sum/src/lib.rs:11
//   brush_hip::tile_sort_offsets render.rs:228-243 + kernels/get_tile_offset.rs:11-58
//   brush_hip::SplatTrainer      brush-train/src/train.rs:140-893           (step, refine, set_view_cams)
//   brush_hip::splat_to_ply / load_splat_from_ply   brush-serde/src/export.rs:179-204, import.rs:166-170
//   brush_hip::image_loss / image_loss_backward      brush-loss/src/lib.rs:718-733, 1075-1104 (LossOps; [H,W,C] in, [H,W,C] out)
//   brush_hip::image_loss_value_and_grad             the composition SplatTrainer::step makes of it (train.rs:227-260)
//   brush_hip::adam_step / gather_stats              brush-train/src/adam_scaled.rs:75-147, stats.rs:40-50
//   brush_hip::BatchUploader / SceneLoader           brush-dataset/src/scene.rs:97-136, scene_loader.rs:59-174
//   brush_hip::sample_background / normal_samples    train.rs:896-908, 389-416 (the library's counter-based generator)
//   Context::comm_* / allreduce_* / exchange_strip_halos   not in the reference: RCCL behind the C ABI
//
// Errors are exceptions (brush_hip::Error carrying bh_last_error) where the reference panics.  Device memory is
// owned by DeviceBuffer<T> (hipMalloc/hipFree); nothing here computes — every operation is one C-ABI call.
// tests/cpp/test_host.cpp runs the reference-style checks through this header on the MI355X.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "brush_hip.h"

namespace brush_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("brush_hip error " + std::to_string(c) + ": " + m), code(c) {}
};

inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(BH_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// ---- device memory -------------------------------------------------------------------------------------------
template <class T>
class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { resize(n); }
    explicit DeviceBuffer(const std::vector<T>& host) { upload(host); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void resize(size_t n) {
        release();
        if (n) hip_check(hipMalloc((void**)&p_, n * sizeof(T)), "hipMalloc");
        n_ = n;
    }
    void upload(const std::vector<T>& host) {
        if (host.size() != n_) resize(host.size());
        if (n_) hip_check(hipMemcpy(p_, host.data(), n_ * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
    }
    std::vector<T> download() const {
        std::vector<T> out(n_);
        if (n_) hip_check(hipMemcpy(out.data(), p_, n_ * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        return out;
    }
    void zero() { if (n_) hip_check(hipMemset(p_, 0, n_ * sizeof(T)), "hipMemset"); }
    T* data() { return p_; }
    const T* data() const { return p_; }
    size_t size() const { return n_; }

  private:
    void release() { if (p_) (void)hipFree(p_); p_ = nullptr; n_ = 0; }
    T* p_ = nullptr;
    size_t n_ = 0;
};

template <class T>
std::vector<T> download(const T* dev, size_t n) {
    std::vector<T> out(n);
    if (n) hip_check(hipMemcpy(out.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return out;
}

// ---- context: one per thread / per GPU (brush-async/src/lib.rs:1-17) ---------------------------------------
class Context {
  public:
    explicit Context(int device = 0) : h_(nullptr) {
        check_abi();
        h_ = bh_create(device, nullptr, /*own_stream=*/1);
        if (!h_) throw Error(BH_ERR_HIP, "bh_create failed: no HIP device " + std::to_string(device));
    }
    // the library fills the header's structs with the layout IT was built with: refuse a library of another revision
    static void check_abi() {
        if (bh_abi_version() != BH_ABI_VERSION)
            throw Error(BH_ERR_UNSUPPORTED, "libbrush_hip.so speaks ABI " + std::to_string(bh_abi_version()) + ", this header ABI " + std::to_string(BH_ABI_VERSION));
        const size_t mine[BH_STRUCT_COUNT] = {sizeof(BhCamera), sizeof(BhRenderOut), sizeof(BhLossConfig), sizeof(BhTrainConfig), sizeof(BhTrainState),
                                              sizeof(BhTrainBatch), sizeof(BhTrainStats), sizeof(BhRefineConfig), sizeof(BhRefineStats), sizeof(BhPlyInfo)};
        for (uint32_t i = 0; i < BH_STRUCT_COUNT; ++i)
            if (bh_struct_size(i) != mine[i])
                throw Error(BH_ERR_UNSUPPORTED, "libbrush_hip.so: struct " + std::to_string(i) + " is " + std::to_string(bh_struct_size(i)) + " bytes in the library, " +
                                                    std::to_string(mine[i]) + " in this header");
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ~Context() { bh_destroy(h_); }
    bh_ctx* get() const { return h_; }
    void check(int rc) const { if (rc != 0) throw Error(rc, bh_last_error(h_)); }
    void sync() const { check(bh_sync(h_)); }

    // ---- per-tile list controls (include/brush_hip.h "depth-sliced lists"): only the TIME of a render depends on them
    void set_list_slicing(float near_share) const { check(bh_set_list_slicing(h_, near_share)); }       // <= 0: automatic (per-tile cuts)
    void set_list_cut_threshold(uint32_t min_pairs) const { check(bh_set_list_cut_threshold(h_, min_pairs)); }
    void set_view_id(uint32_t view_id) const { check(bh_set_view_id(h_, view_id)); }                     // sticky; 0 = keyed by the camera
    void forget_views() const { check(bh_forget_views(h_)); }                                            // after loading another scene
    // bh_set_option: select one of the library's alternative paths (same results; the keys: bh_option_name / include/brush_hip.h)
    void set_option(const std::string& key, const std::string& value) const { check(bh_set_option(h_, key.c_str(), value.c_str())); }
    static std::vector<std::pair<std::string, std::string>> options() {   // (key, one line of documentation)
        std::vector<std::pair<std::string, std::string>> out;
        for (int i = 0; i < bh_option_count(); ++i) out.emplace_back(bh_option_name(i), bh_option_help(i));
        return out;
    }
    float last_list_share() const { return bh_last_list_share(h_); }
    uint32_t far_slices_queued() const { return bh_far_slices_queued(h_); }
    uint32_t view_table_count() const { return bh_view_table_count(h_); }
    std::pair<uint32_t, uint32_t> last_list_counts() const {   // (near pairs, far pairs) of the last forward
        uint32_t a = 0, b = 0;
        check(bh_last_list_counts(h_, &a, &b));
        return {a, b};
    }
    BhRenderOut last_render_out() const {
        BhRenderOut o{};
        check(bh_last_render_out(h_, &o));
        return o;
    }

    // ---- per-stage profile (HIP events around every pipeline stage; 2 = only the dominant kernel)
    void profile(int level = 1) const { check(bh_profile_enable(h_, level)); }
    struct StageTime { std::string name; float ms; uint32_t calls; };
    std::vector<StageTime> profile_fetch() const {
        const char* names[64]; float ms[64]; uint32_t calls[64];
        const int n = bh_profile_fetch(h_, names, ms, calls, 64);
        if (n < 0) check(n);
        std::vector<StageTime> out;
        for (int i = 0; i < n; ++i) out.push_back({names[i], ms[i], calls[i]});
        return out;
    }

    // ---- the library's communicator (RCCL, resolved at first use; SURVEY §8e — not in the reference, which is single-GPU).
    // Rank 0 calls comm_unique_id() and hands the 128 bytes to every rank by its own means; every rank calls comm_init.  With a
    // communicator of world > 1 and no hook, SplatTrainer::step sums its gradients (and, for a tile-row window, exchanges the strips'
    // halos) through it.
    static std::array<uint8_t, 128> comm_unique_id() {
        std::array<uint8_t, 128> id{};
        const int rc = bh_comm_unique_id(id.data());
        if (rc != 0) throw Error(rc, "bh_comm_unique_id failed (is librccl.so loadable?)");
        return id;
    }
    void comm_init(int rank, int world, const std::array<uint8_t, 128>& id) const { check(bh_comm_init(h_, rank, world, id.data())); }
    void comm_destroy() const { check(bh_comm_destroy(h_)); }
    int comm_world() const { return bh_comm_world(h_); }
    int comm_rank() const { return bh_comm_rank(h_); }
    void comm_selftest() const { check(bh_comm_selftest(h_)); }   // collective: every rank calls it once after comm_init
    void allreduce_sum(float* dev, uint64_t count) const { check(bh_allreduce_sum_f32(h_, dev, count)); }
    void allreduce_max(float* dev, uint64_t count) const { check(bh_allreduce_max_f32(h_, dev, count)); }   // RefineRecord maxima before refine
    void allgather_bytes(const void* send_dev, void* recv_dev, uint64_t bytes_per_rank) const { check(bh_allgather_bytes(h_, send_dev, recv_dev, bytes_per_rank)); }
    // one frame split into strips of tile rows: fetch the 21 pixel rows above and below this rank's strip from its neighbours
    void exchange_strip_halos(float* img_hwc4, uint32_t h, uint32_t w, uint32_t row_begin_px, uint32_t row_end_px) const {
        check(bh_exchange_strip_halos(h_, img_hwc4, h, w, row_begin_px, row_end_px));
    }
    static std::vector<BhHaloOp> strip_halo_plan(uint32_t img_h, uint32_t row_begin_px, uint32_t row_end_px, int rank, int world) {
        BhHaloOp ops[4];
        const int n = bh_strip_halo_plan(img_h, row_begin_px, row_end_px, rank, world, ops);
        if (n < 0) throw Error(n, "strip_halo_plan: bad strip");
        return std::vector<BhHaloOp>(ops, ops + n);
    }

  private:
    bh_ctx* h_;
};

// ---- Camera (camera.rs:12-58) ------------------------------------------------------------------------------------
enum class CameraModel : uint32_t {
    Pinhole = BH_CAMERA_PINHOLE,
    KannalaBrandt4 = BH_CAMERA_KANNALA_BRANDT_4,
    RadialTangential8 = BH_CAMERA_RADIAL_TANGENTIAL_8,
    ThinPrismFisheye = BH_CAMERA_THIN_PRISM_FISHEYE
};

struct Camera {
    double fov_x = 1.0, fov_y = 1.0;
    float center_uv[2] = {0.5f, 0.5f};
    float position[3] = {0.0f, 0.0f, 0.0f};
    float rotation[4] = {0.0f, 0.0f, 0.0f, 1.0f};  // glam order x, y, z, w
    CameraModel camera_model = CameraModel::Pinhole;
    float dist[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // the model's parameter struct in field order (brush_hip.h)

    bool is_valid() const {  // camera.rs:41-47
        bool ok = std::isfinite(fov_x) && std::isfinite(fov_y) && std::isfinite(center_uv[0]) && std::isfinite(center_uv[1]);
        for (float v : position) ok = ok && std::isfinite(v);
        for (float v : rotation) ok = ok && std::isfinite(v);
        return ok;
    }
    // kernel uniforms for an (img_w, img_h) render: focal from fov through the lens law, view matrix, clamp limits
    BhCamera uniforms(uint32_t img_w, uint32_t img_h) const {
        BhCamera cam{};
        const int rc = bh_camera_setup_model(position, rotation, fov_x, fov_y, center_uv[0], center_uv[1], img_w, img_h,
                                             (uint32_t)camera_model, dist, &cam);
        if (rc != 0) throw Error(rc, "Can't render images with 0 size.");  // render.rs:50-53
        return cam;
    }
};
inline double fov_to_focal(double fov, uint32_t pixels, CameraModel m = CameraModel::Pinhole, const float* dist = nullptr) {
    return bh_fov_to_focal(fov, pixels, (uint32_t)m, dist);
}
inline double focal_to_fov(double focal, uint32_t pixels, CameraModel m = CameraModel::Pinhole, const float* dist = nullptr) {
    return bh_focal_to_fov(focal, pixels, (uint32_t)m, dist);
}

// ---- Splats (gaussian_splats.rs:62-256) --------------------------------------------------------------------
inline uint32_t sh_degree_from_coeffs(uint32_t coeffs) {  // sh.rs
    uint32_t d = 0;
    while ((d + 1) * (d + 1) < coeffs) ++d;
    if ((d + 1) * (d + 1) != coeffs || d > 4) throw Error(BH_ERR_INVALID_ARG, "sh_coeffs must have (d+1)^2 coefficients, d <= 4");
    return d;
}

struct Splats {
    DeviceBuffer<float> transforms;     // [N,10] means(3) quat wxyz(4) log-scales(3)
    DeviceBuffer<float> sh_coeffs;      // [N,C,3]
    DeviceBuffer<float> raw_opacities;  // [N] logits
    std::optional<DeviceBuffer<float>> min_scale;  // [N] Mip-Splatting 3D-filter floor
    bool render_mip = false;

    static Splats from_host(const std::vector<float>& transforms, const std::vector<float>& sh, const std::vector<float>& raw_opac,
                            bool render_mip = false) {
        const size_t n = raw_opac.size();
        if (transforms.size() != n * 10 || (n && sh.size() % (3 * n) != 0)) throw Error(BH_ERR_INVALID_ARG, "Splats: transforms [N,10], sh [N,C,3], raw_opacities [N]");
        Splats s;
        s.transforms.upload(transforms);
        s.sh_coeffs.upload(sh);
        s.raw_opacities.upload(raw_opac);
        s.render_mip = render_mip;
        (void)s.sh_degree();
        return s;
    }
    uint32_t num_splats() const { return (uint32_t)raw_opacities.size(); }
    uint32_t num_coeffs() const { return num_splats() ? (uint32_t)(sh_coeffs.size() / (3 * (size_t)num_splats())) : 1u; }
    uint32_t sh_degree() const { return sh_degree_from_coeffs(num_coeffs()); }
    Splats& with_min_scale(DeviceBuffer<float> f) {  // gaussian_splats.rs:188-194
        if (f.size() != num_splats()) throw Error(BH_ERR_INVALID_ARG, "min_scale must have one entry per splat");
        min_scale = std::move(f);
        return *this;
    }
    // Splats::bake_min_scale (gaussian_splats.rs:245-256): fold the floor into the raw parameters, in place
    Splats& bake_min_scale(const Context& ctx) {
        if (min_scale) {
            ctx.check(bh_fold_min_scale(ctx.get(), transforms.data(), raw_opacities.data(), min_scale->data(), num_splats(), transforms.data(),
                                        raw_opacities.data()));
            ctx.sync();  // the floor buffer is freed next
            min_scale.reset();
        }
        return *this;
    }
};

// ---- RasterPass / render (gaussian_splats.rs:28-48, 365-446) ---------------------------------------------------
enum class RasterPass { Forward, Backward, BackwardSmoothCutoff };
inline bool bwd_info(RasterPass p) { return p != RasterPass::Forward; }

struct RenderAux {  // render_aux.rs:17-68: host scalars + ctx-owned device pointers (valid until the next render on the ctx)
    BhRenderOut raw{};
    uint32_t img_w = 0, img_h = 0, num_splats = 0;
    uint32_t num_visible() const { return raw.num_visible; }
    uint32_t num_listed_splats() const { return raw.num_listed_splats; }   // entries of the compact (depth-ordered) arrays
    uint32_t num_intersections() const { return raw.num_intersections; }
    std::vector<float> image() const { return download(raw.out_img, (size_t)img_w * img_h * 4); }             // [H,W,4] f32
    std::vector<uint32_t> image_packed() const { return download(raw.out_img_packed, (size_t)img_w * img_h); }  // [H,W] rgba8
    void validate() const {  // render_aux.rs:30-45
        if (raw.num_visible > num_splats) throw Error(BH_ERR_STATE, "num_visible exceeds the splat count");
    }
};

namespace detail {
struct Folded {  // what the renderer sees: fold_min_scale(params) when a floor is set (gaussian_splats.rs:379-386)
    const float* t;
    const float* o;
    DeviceBuffer<float> ft, fo;
};
inline Folded fold(const Context& ctx, const Splats& s) {
    Folded f{s.transforms.data(), s.raw_opacities.data(), {}, {}};
    if (s.min_scale) {
        f.ft.resize(s.transforms.size());
        f.fo.resize(s.raw_opacities.size());
        ctx.check(bh_fold_min_scale(ctx.get(), s.transforms.data(), s.raw_opacities.data(), s.min_scale->data(), s.num_splats(), f.ft.data(), f.fo.data()));
        f.t = f.ft.data();
        f.o = f.fo.data();
    }
    return f;
}
inline uint32_t flags_of(const Splats& s, RasterPass pass) {
    return (s.render_mip ? BH_FLAG_MIP : 0u) | (bwd_info(pass) ? BH_FLAG_BWD_INFO : 0u) |
           (pass == RasterPass::BackwardSmoothCutoff ? BH_FLAG_SMOOTH_CUTOFF : 0u);
}
}  // namespace detail

inline RenderAux render_splats(const Context& ctx, const Splats& splats, const Camera& camera, uint32_t img_w, uint32_t img_h,
                               const float background[3], RasterPass pass = RasterPass::Forward) {
    const BhCamera cam = camera.uniforms(img_w, img_h);
    const detail::Folded f = detail::fold(ctx, splats);
    RenderAux aux;
    aux.img_w = img_w; aux.img_h = img_h; aux.num_splats = splats.num_splats();
    ctx.check(bh_render_forward(ctx.get(), &cam, splats.num_splats(), splats.sh_degree(), f.t, splats.sh_coeffs.data(), f.o, background,
                                detail::flags_of(splats, pass), &aux.raw));
    ctx.sync();  // the folded temporaries die with this scope
    return aux;
}

struct SplatGrads {  // SplatGrads + the refine weight (bwd/burn_glue.rs:184-193, 165-180)
    DeviceBuffer<float> v_transforms, v_sh_coeffs, v_raw_opacities, v_refine_weight;
};

// forward (Backward pass flags) + backward for a given dL/d(out_img) [H,W,4] on the device
inline std::pair<RenderAux, SplatGrads> render_splats_bwd(const Context& ctx, const Splats& splats, const Camera& camera, uint32_t img_w,
                                                          uint32_t img_h, const float background[3], const float* v_output,
                                                          RasterPass pass = RasterPass::Backward) {
    if (!bwd_info(pass)) throw Error(BH_ERR_INVALID_ARG, "render_splats_bwd requires a Backward variant");  // bwd/burn_glue.rs:281-284
    const BhCamera cam = camera.uniforms(img_w, img_h);
    const detail::Folded f = detail::fold(ctx, splats);
    RenderAux aux;
    aux.img_w = img_w; aux.img_h = img_h; aux.num_splats = splats.num_splats();
    const uint32_t n = splats.num_splats();
    ctx.check(bh_render_forward(ctx.get(), &cam, n, splats.sh_degree(), f.t, splats.sh_coeffs.data(), f.o, background,
                                detail::flags_of(splats, pass), &aux.raw));
    SplatGrads g;
    g.v_transforms.resize((size_t)n * 10);
    g.v_sh_coeffs.resize(splats.sh_coeffs.size());
    g.v_raw_opacities.resize(n);
    g.v_refine_weight.resize(n);
    // the saved state goes in explicitly (SplatBwdOps::{rasterize_bwd, project_bwd}, bwd/burn_glue.rs:62-92)
    ctx.check(bh_render_backward_saved(ctx.get(), &aux.raw, v_output, f.t, splats.sh_coeffs.data(), f.o, g.v_transforms.data(), g.v_sh_coeffs.data(),
                                       g.v_raw_opacities.data(), g.v_refine_weight.data()));
    if (splats.min_scale)  // chain through the fold (the autodiff of bwd/burn_glue.rs:260-270)
        ctx.check(bh_fold_min_scale_backward(ctx.get(), splats.transforms.data(), splats.raw_opacities.data(), splats.min_scale->data(), n,
                                             g.v_transforms.data(), g.v_raw_opacities.data()));
    ctx.sync();
    return {std::move(aux), std::move(g)};
}

// One differentiable render as the autodiff node the reference registers (bwd/burn_glue.rs:223-311): the forward now, the backward
// later from the node's SAVED state (:336-371).  retain = true keeps the node replayable while other renders run on the ctx
// (bh_render_retain); without it a later forward makes the node stale and backward() throws (BH_ERR_STATE) instead of returning
// another frame's gradients.
class RenderNode {
  public:
    RenderNode(const Context& ctx, const Splats& splats, const Camera& camera, uint32_t img_w, uint32_t img_h, const float background[3],
               bool retain = false, RasterPass pass = RasterPass::Backward)
        : ctx_(ctx), splats_(splats), folded_(detail::fold(ctx, splats)) {
        if (!bwd_info(pass)) throw Error(BH_ERR_INVALID_ARG, "RenderNode requires a Backward variant");
        const BhCamera cam = camera.uniforms(img_w, img_h);
        aux.img_w = img_w; aux.img_h = img_h; aux.num_splats = splats.num_splats();
        ctx.check(bh_render_forward(ctx.get(), &cam, splats.num_splats(), splats.sh_degree(), folded_.t, splats.sh_coeffs.data(), folded_.o, background,
                                    detail::flags_of(splats, pass), &aux.raw));
        if (retain) {
            ctx.check(bh_render_retain(ctx.get(), &aux.raw));
            retained_ = true;
        }
    }
    RenderNode(const RenderNode&) = delete;
    RenderNode& operator=(const RenderNode&) = delete;
    ~RenderNode() { if (retained_) (void)bh_render_release(ctx_.get(), &aux.raw); }
    SplatGrads backward(const float* v_output) const {
        const uint32_t n = splats_.num_splats();
        SplatGrads g;
        g.v_transforms.resize((size_t)n * 10);
        g.v_sh_coeffs.resize(splats_.sh_coeffs.size());
        g.v_raw_opacities.resize(n);
        g.v_refine_weight.resize(n);
        ctx_.check(bh_render_backward_saved(ctx_.get(), &aux.raw, v_output, folded_.t, splats_.sh_coeffs.data(), folded_.o, g.v_transforms.data(),
                                            g.v_sh_coeffs.data(), g.v_raw_opacities.data(), g.v_refine_weight.data()));
        if (splats_.min_scale)
            ctx_.check(bh_fold_min_scale_backward(ctx_.get(), splats_.transforms.data(), splats_.raw_opacities.data(), splats_.min_scale->data(), n,
                                                  g.v_transforms.data(), g.v_raw_opacities.data()));
        ctx_.sync();
        return g;
    }
    RenderAux aux;

  private:
    const Context& ctx_;
    const Splats& splats_;
    detail::Folded folded_;
    bool retained_ = false;
};

// ---- primitives --------------------------------------------------------------------------------------------------
inline void radix_argsort(const Context& ctx, const DeviceBuffer<uint32_t>& keys, const DeviceBuffer<uint32_t>& vals, uint32_t bits,
                          DeviceBuffer<uint32_t>& out_keys, DeviceBuffer<uint32_t>& out_vals) {
    if (keys.size() != vals.size()) throw Error(BH_ERR_INVALID_ARG, "Input keys and values must have the same number of elements");  // brush-sort/src/lib.rs:21-33
    if (bits > 32) throw Error(BH_ERR_INVALID_ARG, "Can only sort up to 32 bits");
    out_keys.resize(keys.size());
    out_vals.resize(keys.size());
    ctx.check(bh_radix_argsort(ctx.get(), keys.data(), vals.data(), (uint32_t)keys.size(), bits, out_keys.data(), out_vals.data()));
    ctx.sync();  // the ctx owns a non-blocking stream: DeviceBuffer::download (default stream) must not overtake it
}
// the forward's tile sort and its offsets table as one operator (render.rs:228-243 + kernels/get_tile_offset.rs:11-58)
inline void tile_sort_offsets(const Context& ctx, const DeviceBuffer<uint32_t>& tile_ids, const DeviceBuffer<uint32_t>& compact_gids, uint32_t num_tiles,
                              DeviceBuffer<uint32_t>& tile_ids_sorted, DeviceBuffer<uint32_t>& compact_gids_sorted, DeviceBuffer<uint32_t>& tile_offsets) {
    if (tile_ids.size() != compact_gids.size()) throw Error(BH_ERR_INVALID_ARG, "tile ids and splat ids must have the same number of elements");
    tile_ids_sorted.resize(tile_ids.size());
    compact_gids_sorted.resize(tile_ids.size());
    tile_offsets.resize((size_t)num_tiles * 2);
    ctx.check(bh_tile_sort_offsets(ctx.get(), tile_ids.data(), compact_gids.data(), (uint32_t)tile_ids.size(), num_tiles, tile_ids_sorted.data(),
                                   compact_gids_sorted.data(), tile_offsets.data()));
    ctx.sync();
}
inline void prefix_sum(const Context& ctx, const DeviceBuffer<uint32_t>& in, DeviceBuffer<uint32_t>& out) {
    out.resize(in.size());
    ctx.check(bh_prefix_sum(ctx.get(), in.data(), (uint32_t)in.size(), out.data()));
    ctx.sync();
}

// ---- image loss (brush-loss) ---------------------------------------------------------------------------------------
struct LossConfig {  // image_loss's arguments (brush-loss/src/lib.rs:1075-1104)
    float l1_weight = 0.8f, ssim_weight = -0.2f;
    std::optional<std::array<float, 3>> composite_bg;  // Some(bg): the GT is composited over bg first (gt + (1 - gt.a) * bg)
    bool mask = false;                                  // loss *= gt.a
    BhLossConfig c() const {
        BhLossConfig k{};
        k.l1_weight = l1_weight; k.ssim_weight = ssim_weight;
        if (composite_bg) { k.bg[0] = (*composite_bg)[0]; k.bg[1] = (*composite_bg)[1]; k.bg[2] = (*composite_bg)[2]; }
        k.composite_bg = composite_bg ? 1 : 0;
        k.mask = mask ? 1 : 0;
        return k;
    }
};
// LossOps::image_loss_forward (lib.rs:718-725): pred [C,H,W] (C = 3, or 4 with the alpha-match plane), gt [H,W] rgba8 -> loss map [C,H,W]
inline DeviceBuffer<float> image_loss(const Context& ctx, const float* pred_chw, const uint32_t* gt_packed, uint32_t channels, uint32_t h, uint32_t w,
                                      const LossConfig& cfg = {}) {
    if (channels != 3 && channels != 4) throw Error(BH_ERR_INVALID_ARG, "image_loss: 3 or 4 channels");
    DeviceBuffer<float> out((size_t)channels * h * w);
    const BhLossConfig k = cfg.c();
    ctx.check(bh_image_loss_forward(ctx.get(), pred_chw, gt_packed, channels, h, w, &k, out.data()));
    ctx.sync();
    return out;
}
// LossOps::image_loss_backward (lib.rs:726-733): dL/d(loss map) [C,H,W] -> dL/d(pred) [C,H,W]
inline DeviceBuffer<float> image_loss_backward(const Context& ctx, const float* pred_chw, const uint32_t* gt_packed, const float* dl_dmap, uint32_t channels,
                                               uint32_t h, uint32_t w, const LossConfig& cfg = {}) {
    if (channels != 3 && channels != 4) throw Error(BH_ERR_INVALID_ARG, "image_loss_backward: 3 or 4 channels");
    DeviceBuffer<float> out((size_t)channels * h * w);
    const BhLossConfig k = cfg.c();
    ctx.check(bh_image_loss_backward(ctx.get(), pred_chw, gt_packed, dl_dmap, channels, h, w, &k, out.data()));
    ctx.sync();
    return out;
}
// What SplatTrainer::step composes from image_loss + mean (+ the alpha term) + autodiff (train.rs:227-260), fused: the rasterizer's
// [H,W,4] image in, (loss, dloss/dimg [H,W,4]) out.
inline std::pair<float, DeviceBuffer<float>> image_loss_value_and_grad(const Context& ctx, const float* img_hwc4, const uint32_t* gt_packed, uint32_t h, uint32_t w,
                                                                       const LossConfig& cfg = {}, float alpha_weight = 0.0f) {
    DeviceBuffer<float> v_out((size_t)h * w * 4), loss(1);
    const BhLossConfig k = cfg.c();
    ctx.check(bh_image_loss_value_and_grad(ctx.get(), img_hwc4, gt_packed, h, w, &k, alpha_weight, loss.data(), v_out.data()));
    ctx.sync();
    return {loss.download()[0], std::move(v_out)};
}

// ---- optimizer / statistics (brush-train) -------------------------------------------------------------------------
// AdamScaled::step on one [rows, row_len] parameter, in place (adam_scaled.rs:75-147).  t = state.time AFTER this step (1 on the
// first call); col_scale: per-column learning-rate scale or null; reduce_m2: one second moment per row (m2 has `rows` entries).
// Queued on the ctx stream like every call that returns no host value: Context::sync() before touching the buffers from elsewhere.
inline void adam_step(const Context& ctx, float* param, const float* grad, float* m1, float* m2, uint64_t rows, uint32_t row_len, float lr, uint32_t t,
                      const float* col_scale = nullptr, bool reduce_m2 = false, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-15f) {
    ctx.check(bh_adam_step(ctx.get(), param, grad, m1, m2, rows, row_len, col_scale, lr, t, reduce_m2 ? 1 : 0, beta1, beta2, eps));
}
// RefineRecord::gather_stats (stats.rs:40-50): running maxima of the refine weight and the screen radius, running sum of visibility
inline void gather_stats(const Context& ctx, float* refine_weight_norm, float* vis_weight, float* max_screen_size, const float* refine_weight, const float* visible,
                         const float* screen_radius, uint64_t n) {
    ctx.check(bh_gather_stats(ctx.get(), refine_weight_norm, vis_weight, max_screen_size, refine_weight, visible, screen_radius, n));
}

// ---- the stochastic terms of step() (train.rs:389-416, 896-908) as pure functions of (seed, step) -------------------------
inline std::array<float, 3> sample_background(uint64_t seed, uint32_t step, const float base[3], float strength) {
    std::array<float, 3> out{};
    bh_sample_background(seed, step, base, strength, out.data());
    return out;
}
inline DeviceBuffer<float> normal_samples(const Context& ctx, uint64_t seed, uint32_t step, uint64_t n) {   // [n,3] N(0,1): what a device_noise step draws
    DeviceBuffer<float> out((size_t)n * 3);
    ctx.check(bh_normal_samples(ctx.get(), seed, step, n, out.data()));
    ctx.sync();
    return out;
}
inline std::array<uint32_t, 4> philox4x32_10(const std::array<uint32_t, 4>& ctr, const std::array<uint32_t, 2>& key) {   // Random123's generator, host
    std::array<uint32_t, 4> out{};
    bh_philox4x32_10(ctr.data(), key.data(), out.data());
    return out;
}

// ---- PLY (brush-serde) -------------------------------------------------------------------------------------------
inline std::vector<uint8_t> splat_to_ply(const Context& ctx, const Splats& s, const float* up_axis = nullptr) {
    uint64_t need = 0;
    const float* f = s.min_scale ? s.min_scale->data() : nullptr;
    ctx.check(bh_splat_to_ply(ctx.get(), s.transforms.data(), s.sh_coeffs.data(), s.raw_opacities.data(), f, s.num_splats(), s.sh_degree(),
                              s.render_mip ? 1 : 0, up_axis, nullptr, 0, &need));
    std::vector<uint8_t> out(need);
    ctx.check(bh_splat_to_ply(ctx.get(), s.transforms.data(), s.sh_coeffs.data(), s.raw_opacities.data(), f, s.num_splats(), s.sh_degree(),
                              s.render_mip ? 1 : 0, up_axis, out.data(), need, &need));
    return out;
}
// subsample_points = s keeps every s-th file row (s-1, 2s-1, ...: import.rs:346-349), max_splats then caps the count the way
// SplatData::subsample does (rows 0, step, 2 step, ...; import.rs:49-74).  info.num_splats is the number of rows kept.
inline std::pair<Splats, BhPlyInfo> load_splat_from_ply(const Context& ctx, const std::vector<uint8_t>& bytes, uint32_t subsample_points = 1,
                                                        size_t max_splats = 0) {
    BhPlyInfo info{};
    const int rc = bh_ply_parse_header(bytes.data(), bytes.size(), &info);
    if (rc != 0) throw Error(rc, rc == BH_ERR_UNSUPPORTED ? "unsupported PLY variant" : "malformed PLY");
    if (subsample_points < 1) throw Error(BH_ERR_INVALID_ARG, "subsample_points must be >= 1");
    uint64_t first = subsample_points - 1, step = subsample_points, n = info.num_splats / subsample_points;
    if (max_splats != 0 && n > max_splats) {
        const uint64_t step2 = (n + max_splats - 1) / max_splats;
        n = (n + step2 - 1) / step2;
        step *= step2;
    }
    Splats s;
    const size_t c = (size_t)(info.sh_degree + 1) * (info.sh_degree + 1);
    s.transforms.resize(n * 10);
    s.sh_coeffs.resize(n * c * 3);
    s.raw_opacities.resize(n);
    s.render_mip = info.render_mode == 1;
    ctx.check(bh_splats_from_ply_strided(ctx.get(), bytes.data(), bytes.size(), first, step, n, s.transforms.data(), s.sh_coeffs.data(),
                                         s.raw_opacities.data()));
    info.num_splats = n;
    return {std::move(s), info};
}

// ---- training (brush-train) ---------------------------------------------------------------------------------------
struct TrainConfig {  // config.rs:7-132 subset, same defaults
    uint32_t total_train_iters = 30000;
    double lr_mean = 2e-5, lr_mean_end = 2e-7, lr_coeffs_dc = 2e-3, lr_opac = 0.012, lr_scale = 5e-3, lr_rotation = 2e-3;
    float lr_coeffs_sh_scale = 10.0f, ssim_weight = 0.2f, match_alpha_weight = 0.1f, mean_noise_weight = 50.0f;
    float background_color[3] = {0.0f, 0.0f, 0.0f};
    float background_noise_strength = 0.1f;
    bool render_mip = false;
    bool exact_lists = false;  // not in the reference: false = depth-sliced per-tile lists in step() (BH_FLAG_SLICED_LISTS, same results)
    uint32_t max_splats = 10000000, refine_every = 200, growth_stop_iter = 15000;
    float growth_grad_threshold = 0.0025f, growth_select_fraction = 0.25f, split_at_screen_size = 0.5f, opac_decay = 0.004f;
};

struct SceneBatch {  // brush-dataset/src/scene.rs:139-147
    const uint32_t* img_packed = nullptr;  // device [H,W] rgba8
    uint32_t img_w = 0, img_h = 0;
    bool has_alpha = false, alpha_is_mask = false;
    Camera camera;
    uint32_t view_id = 0;  // which view of the dataset this is (its index + 1; 0 = unknown): keys the forward's per-tile depth cuts (time only)
};

struct TrainStepStats { uint32_t num_visible, num_intersections; double lr_mean; float loss; };

// ---- host image -> packed device batch (brush-dataset) ---------------------------------------------------------------
// view_to_packed_data (scene.rs:97-136) on the device behind a ring of pinned staging slots and a copy stream: the decoded RGB8 / RGBA8
// bytes cross PCIe unpacked, widening (a = 255), the byte-space premultiply of AlphaMode::Transparent and the packing run in a kernel
// on the copy stream while the previous batch trains.  Life of a slot: map -> (decode into the pinned bytes) -> commit -> acquire
// (the ctx stream waits for the upload on the device: no host block) -> queue the train step -> release.
class BatchUploader {
  public:
    BatchUploader(const Context& ctx, uint64_t max_pixels, uint32_t slots = 3) : slots_(slots) {
        up_ = bh_uploader_create(ctx.get(), max_pixels, slots);
        if (!up_) throw Error(BH_ERR_INVALID_ARG, "bh_uploader_create failed (max_pixels > 0, 2 <= slots <= 16, enough pinned memory)");
    }
    BatchUploader(const BatchUploader&) = delete;
    BatchUploader& operator=(const BatchUploader&) = delete;
    ~BatchUploader() { bh_uploader_destroy(up_); }
    uint32_t slots() const { return slots_; }
    // the next slot's pinned buffer to decode straight into (blocks only when the ring wraps onto a slot whose step is still running)
    std::pair<int, uint8_t*> map(uint64_t bytes) {
        void* p = nullptr;
        const int slot = check(bh_uploader_begin(up_, bytes, &p));
        return {slot, (uint8_t*)p};
    }
    void commit(int slot, uint32_t w, uint32_t h, uint32_t channels, bool premultiply) { check(bh_uploader_commit(up_, slot, w, h, channels, premultiply ? 1 : 0)); }
    // map + memcpy + commit for pixels that already live elsewhere: tightly packed [H,W,3|4] bytes; returns the slot
    int submit(const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t channels, bool premultiply = true) {
        if (channels != 3 && channels != 4) throw Error(BH_ERR_INVALID_ARG, "image must be [H,W,3] or [H,W,4] uint8");
        return check(bh_uploader_submit(up_, pixels, w, h, channels, (premultiply && channels == 4) ? 1 : 0));
    }
    struct Packed { const uint32_t* img; uint32_t w, h; bool has_alpha; };   // device [H,W] rgba8, aliasing the slot
    Packed acquire(int slot) {
        Packed r{};
        int ha = 0;
        check(bh_uploader_acquire(up_, slot, &r.img, &r.w, &r.h, &ha));
        r.has_alpha = ha != 0;
        return r;
    }
    void release(int slot) { check(bh_uploader_release(up_, slot)); }   // once the step that reads the slot is queued

  private:
    int check(int rc) const {
        if (rc < 0) throw Error(rc, std::string("uploader: ") + bh_uploader_last_error(up_));
        return rc;
    }
    bh_uploader* up_ = nullptr;
    uint32_t slots_;
};

// SceneLoader (scene_loader.rs:59-174): an endless shuffled stream of SceneBatch over a list of views, the upload of the NEXT views
// in flight while the current one trains.  Every epoch is a seeded Fisher-Yates permutation (SplitMix64) of this rank's views — the
// reference's order comes from rand::StdRng inside racing loader tasks and is not reproducible, so only "every view once per epoch"
// is kept.  rank / world shard the view list for data-parallel training (view i belongs to rank i % world).  No loader thread here:
// next_batch() submits the views that follow before it hands the current one out (the copies and the packing kernel run on the
// uploader's stream; what the host does per view is one memcpy into pinned memory, or the caller's decode straight into it).
struct LoaderView {
    uint32_t w = 0, h = 0, channels = 3;
    std::function<void(uint8_t* dst)> decode;   // writes w * h * channels tightly packed bytes
    Camera camera;
    bool alpha_is_mask = false;
};

class SceneLoader {
  public:
    SceneLoader(const Context& ctx, std::vector<LoaderView> views, uint64_t seed = 0, uint32_t slots = 3, uint32_t rank = 0, uint32_t world = 1)
        : seed_(seed) {
        for (size_t i = 0; i < views.size(); ++i)
            if (world == 0 || i % world == rank) { views_.push_back(std::move(views[i])); ids_.push_back((uint32_t)i + 1u); }
        if (views_.empty()) throw Error(BH_ERR_INVALID_ARG, "Need at least one view in dataset");  // scene_loader.rs:130
        for (const LoaderView& v : views_)
            if (!v.decode || v.w == 0 || v.h == 0 || (v.channels != 3 && v.channels != 4))
                throw Error(BH_ERR_INVALID_ARG, "LoaderView: needs a decode function, a size and 3 or 4 channels");
        uint64_t mp = 0;
        for (const LoaderView& v : views_) mp = std::max<uint64_t>(mp, (uint64_t)v.w * v.h);
        up_.emplace(ctx, mp, slots);
        depth_ = slots > 1 ? slots - 1 : 1;   // slots in flight = submitted + the one being trained on
    }
    // the view order of `epoch` (deterministic in seed, epoch and the shard)
    std::vector<uint32_t> epoch_order(uint64_t epoch) const {
        std::vector<uint32_t> order(views_.size());
        std::iota(order.begin(), order.end(), 0u);
        uint64_t st = seed_ ^ (0xD1B54A32D192ED03ull * (epoch + 1));
        for (size_t i = order.size(); i-- > 1;) {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            std::swap(order[i], order[(size_t)(z % (i + 1))]);
        }
        return order;
    }
    // -> the next batch (its img_packed aliases an uploader slot that stays valid until the NEXT next_batch call).  Call after queuing
    // the train step of the previous batch: that is what releases its slot.
    SceneBatch next_batch() {
        if (held_ >= 0) { up_->release(held_); held_ = -1; }
        // slots - 1 views submitted ahead: the slot mapped here was released one call ago, i.e. behind a step that has finished by
        // the time the NEXT step is running — map() does not wait for the GPU in steady state
        while (inflight_.size() < depth_) submit_next();
        const InFlight f = inflight_.front();
        inflight_.erase(inflight_.begin());
        const BatchUploader::Packed p = up_->acquire(f.slot);
        held_ = f.slot;
        const LoaderView& v = views_[f.idx];
        SceneBatch b;
        b.img_packed = p.img; b.img_w = p.w; b.img_h = p.h;
        b.has_alpha = p.has_alpha; b.alpha_is_mask = v.alpha_is_mask;
        b.camera = v.camera;
        b.view_id = ids_[f.idx];
        last_index_ = f.idx;
        return b;
    }
    uint32_t last_view_index() const { return last_index_; }   // index (within this rank's shard) of the batch just handed out
    size_t num_views() const { return views_.size(); }

  private:
    struct InFlight { int slot; uint32_t idx; };
    void submit_next() {
        if (cursor_ >= order_.size()) { order_ = epoch_order(epoch_++); cursor_ = 0; }
        const uint32_t idx = order_[cursor_++];
        const LoaderView& v = views_[idx];
        auto [slot, dst] = up_->map((uint64_t)v.w * v.h * v.channels);
        v.decode(dst);
        up_->commit(slot, v.w, v.h, v.channels, v.channels == 4 && !v.alpha_is_mask);
        inflight_.push_back({slot, idx});
    }
    std::vector<LoaderView> views_;
    std::vector<uint32_t> ids_;
    std::optional<BatchUploader> up_;
    uint64_t seed_;
    uint64_t epoch_ = 0;
    std::vector<uint32_t> order_;
    size_t cursor_ = 0;
    std::vector<InFlight> inflight_;
    size_t depth_ = 2;
    int held_ = -1;
    uint32_t last_index_ = 0;
};

class SplatTrainer {
  public:
    SplatTrainer(const Context& ctx, TrainConfig config, float median_scene_scale = 1.0f) : ctx_(ctx), cfg_(config), median_(median_scene_scale) {}

    void set_view_cams(std::vector<float> centre_xyz_focal /*[K,4]*/) { view_cams_ = std::move(centre_xyz_focal); }  // train.rs:170-174
    uint32_t step_count() const { return step_count_; }

    // SplatTrainer::step (train.rs:176-429): forward, L1+SSIM (+alpha) loss, backward, statistics, Adam, optional noise.
    // The two stochastic terms (mean noise train.rs:389-416, background jitter :896-908): with a seed (set_seed) they are
    // drawn by the library's counter-based generator as functions of (seed, step) — the reference's default behaviour;
    // without one they only appear when injected: `noise_samples` device [N,3] N(0,1) or null, `background` the colour
    // actually used this step (parity tests).
    void set_seed(uint64_t seed) { seed_ = seed; have_seed_ = true; }
    TrainStepStats step(const SceneBatch& batch, Splats& splats, const float* noise_samples = nullptr, const float* background = nullptr) {
        ensure_state(splats);
        BhTrainConfig c = c_config(splats);
        BhTrainState st = c_state(splats);
        BhTrainBatch b{};
        b.camera = batch.camera.uniforms(batch.img_w, batch.img_h);
        b.gt_packed = batch.img_packed;
        b.has_alpha = batch.has_alpha;
        b.alpha_is_mask = batch.alpha_is_mask;
        b.view_id = batch.view_id;
        for (int k = 0; k < 3; ++k) b.background[k] = background ? background[k] : cfg_.background_color[k];
        if (!background && have_seed_) bh_sample_background(seed_, step_count_ + 1, cfg_.background_color, cfg_.background_noise_strength, b.background);
        b.noise_samples = noise_samples;
        b.device_noise = (!noise_samples && have_seed_) ? 1 : 0;
        b.noise_seed = seed_;
        BhTrainStats stats{};
        ctx_.check(bh_train_step(ctx_.get(), &c, &st, &b, nullptr, nullptr, 1.0f, &stats));
        step_count_ = st.step_count;
        ctx_.sync();  // delivers stats.loss
        return {stats.num_visible, stats.num_intersections, stats.lr_mean, stats.loss};
    }

    // SplatTrainer::refine (train.rs:431-663); `seed` replaces rand::rng().  Replaces `splats` and the optimizer state.
    BhRefineStats refine(uint32_t iter, Splats& splats, uint64_t seed) {
        if (!have_state_) throw Error(BH_ERR_STATE, "Can only refine if refine stats are initialized");  // train.rs:445
        splats.bake_min_scale(ctx_);
        if (!have_bounds_) update_bounds(splats);
        BhRefineConfig rc{};
        rc.iter = iter;
        rc.total_train_iters = cfg_.total_train_iters ? cfg_.total_train_iters : 1;
        rc.growth_stop_iter = cfg_.growth_stop_iter < cfg_.total_train_iters ? cfg_.growth_stop_iter : cfg_.total_train_iters;
        rc.max_splats = cfg_.max_splats;
        rc.growth_grad_threshold = cfg_.growth_grad_threshold;
        rc.growth_select_fraction = cfg_.growth_select_fraction;
        rc.split_at_screen_size = cfg_.split_at_screen_size;
        rc.opac_decay = cfg_.opac_decay;
        for (int k = 0; k < 3; ++k) { rc.bounds_center[k] = center_[k]; rc.bounds_extent[k] = extent_[k]; }
        rc.seed = seed;
        BhTrainState in = c_state(splats);
        BhRefineStats rs{};
        ctx_.check(bh_refine_plan(ctx_.get(), &rc, &in, &rs));
        const size_t n2 = rs.total_splats, c3 = (size_t)splats.num_coeffs() * 3;
        Splats out;
        out.render_mip = splats.render_mip;
        out.transforms.resize(n2 * 10);
        out.sh_coeffs.resize(n2 * c3);
        out.raw_opacities.resize(n2);
        State ns;
        ns.alloc(n2, c3);
        BhTrainState o = in;
        o.n = (uint32_t)n2;
        o.transforms = out.transforms.data(); o.sh_coeffs = out.sh_coeffs.data(); o.raw_opacities = out.raw_opacities.data();
        ns.fill(o);
        o.min_scale = nullptr;
        ctx_.check(bh_refine_apply(ctx_.get(), &rc, &in, &o));
        ctx_.sync();
        splats = std::move(out);
        state_ = std::move(ns);
        update_bounds(splats);  // train.rs:634
        if ((float)iter / (float)rc.total_train_iters < 0.9f && !view_cams_.empty()) {  // train.rs:636-648, MIN_SCALE_FREEZE_FRAC
            DeviceBuffer<float> f(splats.num_splats());
            ctx_.check(bh_compute_min_scale(ctx_.get(), splats.transforms.data(), splats.num_splats(), view_cams_.data(),
                                            (uint32_t)(view_cams_.size() / 4), 0.1f, f.data()));
            splats.with_min_scale(std::move(f));
        }
        return rs;
    }

  private:
    struct State {
        DeviceBuffer<float> m1_t, m2_t, m1_sh, m2_sh, m1_o, m2_o, refine_weight_norm, vis_weight, max_screen_size;
        void alloc(size_t n, size_t c3) {
            m1_t.resize(n * 10); m2_t.resize(n * 10); m1_sh.resize(n * c3); m2_sh.resize(n); m1_o.resize(n); m2_o.resize(n);
            refine_weight_norm.resize(n); vis_weight.resize(n); max_screen_size.resize(n);
        }
        void zero() { for (auto* b : {&m1_t, &m2_t, &m1_sh, &m2_sh, &m1_o, &m2_o, &refine_weight_norm, &vis_weight, &max_screen_size}) b->zero(); }
        void fill(BhTrainState& s) {
            s.m1_transforms = m1_t.data(); s.m2_transforms = m2_t.data(); s.m1_sh = m1_sh.data(); s.m2_sh = m2_sh.data();
            s.m1_opac = m1_o.data(); s.m2_opac = m2_o.data();
            s.refine_weight_norm = refine_weight_norm.data(); s.vis_weight = vis_weight.data(); s.max_screen_size = max_screen_size.data();
        }
    };
    void ensure_state(const Splats& s) {
        if (have_state_ && state_.m2_o.size() == s.num_splats()) return;
        state_.alloc(s.num_splats(), (size_t)s.num_coeffs() * 3);
        state_.zero();
        have_state_ = true;
    }
    void update_bounds(const Splats& s) {  // get_splat_bounds (train.rs:124-133), BOUND_PERCENTILE 0.8; median_size = 2 * median extent
        ctx_.check(bh_splat_bounds(ctx_.get(), s.transforms.data(), s.num_splats(), 0.8f, center_, extent_));
        float e[3] = {extent_[0], extent_[1], extent_[2]};
        if (e[0] > e[1]) std::swap(e[0], e[1]);
        if (e[1] > e[2]) std::swap(e[1], e[2]);
        if (e[0] > e[1]) std::swap(e[0], e[1]);
        median_ = e[1] * 2.0f;
        have_bounds_ = true;
    }
    BhTrainConfig c_config(const Splats& s) const {
        BhTrainConfig c{};
        c.lr_mean = cfg_.lr_mean; c.lr_mean_end = cfg_.lr_mean_end; c.total_train_iters = cfg_.total_train_iters;
        c.lr_coeffs_dc = cfg_.lr_coeffs_dc; c.lr_coeffs_sh_scale = cfg_.lr_coeffs_sh_scale; c.lr_opac = cfg_.lr_opac;
        c.lr_scale = cfg_.lr_scale; c.lr_rotation = cfg_.lr_rotation; c.ssim_weight = cfg_.ssim_weight;
        c.match_alpha_weight = cfg_.match_alpha_weight; c.mean_noise_weight = cfg_.mean_noise_weight;
        for (int k = 0; k < 3; ++k) c.background[k] = cfg_.background_color[k];
        c.median_scene_scale = median_;
        c.render_mip = (cfg_.render_mip || s.render_mip) ? 1 : 0;
        c.exact_lists = cfg_.exact_lists ? 1 : 0;
        c.growth_stop_iter = cfg_.growth_stop_iter;   // from that step on nobody reads the refine weight (train.rs:589-614)
        return c;
    }
    BhTrainState c_state(Splats& s) {
        BhTrainState st{};
        st.n = s.num_splats(); st.sh_degree = s.sh_degree();
        st.transforms = s.transforms.data(); st.sh_coeffs = s.sh_coeffs.data(); st.raw_opacities = s.raw_opacities.data();
        state_.fill(st);
        st.step_count = step_count_;
        st.min_scale = s.min_scale ? s.min_scale->data() : nullptr;
        return st;
    }
    const Context& ctx_;
    TrainConfig cfg_;
    float median_;
    State state_;
    bool have_state_ = false, have_bounds_ = false;
    float center_[3] = {0, 0, 0}, extent_[3] = {1, 1, 1};
    uint32_t step_count_ = 0;
    uint64_t seed_ = 0;
    bool have_seed_ = false;
    std::vector<float> view_cams_;
};

}  // namespace brush_hip
