// test_host.cpp — the C++ host mirror (include/brush_hip.hpp) exercised like the reference's own tests, on the GPU:
//   * forward / backward of a seeded scene vs the CPU oracle (oracle/libbrush_oracle.so, dlopen'd: the oracle is
//     test infrastructure and only tests may touch it): counts and per-tile sort order exact, image <= 1e-6,
//     gradients <= 1e-4 * max|g|   (crates/brush-bench-test/src/reference.rs, tests/finite_diff.rs roles)
//   * radix_argsort / prefix_sum vs std::stable_sort / std::partial_sum on SplitMix64 inputs
//     (brush-sort/src/lib.rs:154-339, brush-prefix-sum/src/lib.rs:105-196)
//   * SplatTrainer::step x N + refine + PLY export / import round trip (tests/integration.rs:186-235, export.rs:305-349)
//   * image_loss / image_loss_backward / the fused value-and-gradient, adam_step, gather_stats vs the oracle (brush-loss, adam_scaled.rs, stats.rs)
//   * SceneLoader + BatchUploader (every view once per epoch, packed bytes exact, 15 steps through the ring), list controls, stage
//     profile, the counter-based generator, the strip-halo plan and a one-rank RCCL self-test
//   * error behaviour: what the reference asserts, this host throws.
// Build + run: tests/test_gpu_cpp_host.py.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

#include "brush_hip.hpp"

namespace bh = brush_hip;

static int g_failed = 0;
#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) { std::printf("FAIL %s:%d  %s  ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); ++g_failed; } \
    } while (0)

struct Sm64 {  // crates/brush-render/src/tests/mod.rs:168-186
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float unit() { return (float)((double)next() / 18446744073709551615.0); }
    float uni(float lo, float hi) { return lo + unit() * (hi - lo); }
};

struct HostScene { std::vector<float> transforms, sh, raw_opac; };
static HostScene make_scene(uint32_t n, uint64_t seed, uint32_t coeffs) {
    Sm64 r{seed};
    HostScene s;
    s.transforms.resize((size_t)n * 10);
    s.sh.resize((size_t)n * coeffs * 3);
    s.raw_opac.resize(n);
    const float t = std::tan(0.5235988f);
    for (uint32_t i = 0; i < n; ++i) {
        float* tr = &s.transforms[(size_t)i * 10];
        const float z = r.uni(2.0f, 12.0f);
        tr[0] = r.uni(-1.1f, 1.1f) * z * t;
        tr[1] = r.uni(-1.1f, 1.1f) * z * t * 0.75f;
        tr[2] = z;
        for (int k = 3; k < 7; ++k) tr[k] = r.uni(-1.0f, 1.0f);
        for (int k = 7; k < 10; ++k) tr[k] = r.uni(std::log(0.02f), std::log(0.2f));
        const float p = r.uni(0.05f, 0.95f);
        s.raw_opac[i] = std::log(p / (1.0f - p));
        for (uint32_t c = 0; c < coeffs * 3; ++c) s.sh[(size_t)i * coeffs * 3 + c] = c < 3 ? r.uni(-1.0f, 1.7f) : r.uni(-0.25f, 0.25f);
    }
    return s;
}

// ---- the oracle, bound at run time ---------------------------------------------------------------------------------
struct BoCamera { float vm[12], fx, fy, cx, cy, lim[4], cam_pos[3]; uint32_t img_w, img_h, model; float dist[8], half_fov; };
struct Oracle {
    void* lib = nullptr;
    void (*camera_setup_model)(const float*, const float*, double, double, float, float, uint32_t, uint32_t, uint32_t, const float*, BoCamera*);
    void* (*render_create)();
    void (*render_free)(void*);
    int (*render_forward)(void*, const BoCamera*, uint32_t, uint32_t, const float*, const float*, const float*, const float*, uint32_t);
    int (*render_backward)(void*, const float*, const float*, const float*, const float*);
    uint32_t (*num_visible)(void*);
    uint32_t (*num_intersections)(void*);
    const float* (*get_out_img)(void*, uint64_t*);
    const uint32_t* (*get_cgfi)(void*, uint64_t*);
    const float* (*get_v_transforms)(void*, uint64_t*);
    const float* (*get_v_raw_opac)(void*, uint64_t*);
    void (*image_loss_forward)(const float*, const uint32_t*, uint32_t, uint32_t, uint32_t, float, float, const float*, int, int, float*);
    void (*image_loss_backward)(const float*, const uint32_t*, const float*, uint32_t, uint32_t, uint32_t, float, float, const float*, int, int, float*);
    void (*adam_step)(float*, const float*, float*, float*, uint64_t, uint32_t, const float*, float, uint32_t, int, float, float, float);
    void (*gather_stats)(float*, float*, float*, const float*, const float*, const float*, uint64_t);
    bool load(const char* path) {
        lib = dlopen(path, RTLD_NOW);
        if (!lib) { std::printf("cannot load oracle %s: %s\n", path, dlerror()); return false; }
#define SYM(field, name) field = (decltype(field))dlsym(lib, name); if (!field) { std::printf("oracle symbol %s missing\n", name); return false; }
        SYM(camera_setup_model, "bo_camera_setup_model") SYM(render_create, "bo_render_create") SYM(render_free, "bo_render_free")
        SYM(render_forward, "bo_render_forward") SYM(render_backward, "bo_render_backward") SYM(num_visible, "bo_num_visible")
        SYM(num_intersections, "bo_num_intersections") SYM(get_out_img, "bo_get_out_img") SYM(get_cgfi, "bo_get_compact_gid_from_isect")
        SYM(get_v_transforms, "bo_get_v_transforms") SYM(get_v_raw_opac, "bo_get_v_raw_opac")
        SYM(image_loss_forward, "bo_image_loss_forward") SYM(image_loss_backward, "bo_image_loss_backward") SYM(adam_step, "bo_adam_step")
        SYM(gather_stats, "bo_gather_stats")
#undef SYM
        return true;
    }
};

static float rel_linf(const std::vector<float>& a, const float* b, size_t n) {
    double d = 0, m = 0;
    for (size_t i = 0; i < n; ++i) { d = std::max(d, (double)std::fabs(a[i] - b[i])); m = std::max(m, (double)std::fabs(b[i])); }
    return (float)(d / std::max(m, 1e-30));
}

static void test_render_vs_oracle(const bh::Context& ctx, Oracle& bo, bh::CameraModel model, const char* name) {
    const uint32_t n = 5000, w = 160, h = 120, coeffs = 4;
    const HostScene sc = make_scene(n, 0xC0FFEE, coeffs);
    bh::Splats splats = bh::Splats::from_host(sc.transforms, sc.sh, sc.raw_opac);
    bh::Camera cam;
    cam.fov_x = 1.0471976; cam.fov_y = 2.0 * std::atan(0.75 * std::tan(cam.fov_x / 2));
    cam.camera_model = model;
    if (model == bh::CameraModel::KannalaBrandt4) { const float d[4] = {-0.05f, 0.01f, -0.001f, 5e-5f}; std::memcpy(cam.dist, d, sizeof d); }
    if (model == bh::CameraModel::RadialTangential8) { const float d[8] = {-0.2f, 0.05f, -0.001f, 0, 0, 0, 1e-3f, -1e-3f}; std::memcpy(cam.dist, d, sizeof d); }
    const float bg[3] = {0.1f, 0.2f, 0.3f};
    std::vector<float> v_host((size_t)w * h * 4, 1.0f / (w * h * 4));
    bh::DeviceBuffer<float> v_out(v_host);
    auto [aux, grads] = bh::render_splats_bwd(ctx, splats, cam, w, h, bg, v_out.data());
    aux.validate();
    // oracle
    BoCamera oc{};
    bo.camera_setup_model(cam.position, cam.rotation, cam.fov_x, cam.fov_y, 0.5f, 0.5f, w, h, (uint32_t)model, cam.dist, &oc);
    void* r = bo.render_create();
    CHECK(bo.render_forward(r, &oc, n, 1, sc.transforms.data(), sc.sh.data(), sc.raw_opac.data(), bg, 2 /*BWD_INFO*/) == 0, "oracle forward");
    CHECK(aux.num_visible() == bo.num_visible(r) && aux.num_intersections() == bo.num_intersections(r), "%s: counts %u/%u vs %u/%u", name,
          aux.num_visible(), aux.num_intersections(), bo.num_visible(r), bo.num_intersections(r));
    CHECK(aux.num_visible() > n / 2, "%s: scene mostly visible", name);
    uint64_t cnt = 0;
    const uint32_t* ref_order = bo.get_cgfi(r, &cnt);
    const auto order = bh::download(aux.raw.compact_gid_from_isect, aux.num_intersections());
    CHECK(cnt == order.size() && std::memcmp(order.data(), ref_order, cnt * 4) == 0, "%s: per-tile sort order is bit-exact", name);
    const float* ref_img = bo.get_out_img(r, &cnt);
    const auto img = aux.image();
    float dmax = 0;
    for (size_t i = 0; i < img.size(); ++i) dmax = std::max(dmax, std::fabs(img[i] - ref_img[i]));
    CHECK(cnt == img.size() && dmax <= 1e-6f, "%s: image max |d| = %g", name, dmax);
    CHECK(bo.render_backward(r, v_host.data(), sc.transforms.data(), sc.sh.data(), sc.raw_opac.data()) == 0, "oracle backward");
    const float* ref_vt = bo.get_v_transforms(r, &cnt);
    const float e1 = rel_linf(grads.v_transforms.download(), ref_vt, cnt);
    const float* ref_vo = bo.get_v_raw_opac(r, &cnt);
    const float e2 = rel_linf(grads.v_raw_opacities.download(), ref_vo, cnt);
    CHECK(e1 <= 1e-4f && e2 <= 1e-4f, "%s: gradient rel. error %g / %g", name, e1, e2);
    bo.render_free(r);
    std::printf("ok render_vs_oracle[%s]  nv=%u isect=%u img_err=%.1e grad_err=%.1e\n", name, aux.num_visible(), aux.num_intersections(), dmax, e1);
}

// forward A, forward B, backward A.  With the saved state passed explicitly (bh_render_backward_saved) a
// retained A still yields A's gradients (== the oracle's for A), and an A that was NOT retained fails loudly instead of returning
// B's gradients (bwd/burn_glue.rs:62-92, 336-371: RenderBackwards owns its saved tensors).
static void test_two_forwards_alive(const bh::Context& ctx, Oracle& bo) {
    const uint32_t n = 4000, w = 144, h = 96, coeffs = 1;
    const HostScene sc = make_scene(n, 0xABCD, coeffs);
    bh::Splats splats = bh::Splats::from_host(sc.transforms, sc.sh, sc.raw_opac);
    bh::Camera camA, camB;
    camA.fov_x = camB.fov_x = 1.0471976; camA.fov_y = camB.fov_y = 2.0 * std::atan((double)h / w * std::tan(camA.fov_x / 2));
    camB.position[0] = 1.5f;   // another view: other lists, other gradients
    const float bg[3] = {0.2f, 0.1f, 0.0f};
    std::vector<float> v_host((size_t)w * h * 4);
    for (size_t i = 0; i < v_host.size(); ++i) v_host[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
    bh::DeviceBuffer<float> v_out(v_host);
    auto oracle_grads = [&](const bh::Camera& cam, std::vector<float>& vt) {
        BoCamera oc{};
        bo.camera_setup_model(cam.position, cam.rotation, cam.fov_x, cam.fov_y, 0.5f, 0.5f, w, h, 0u, cam.dist, &oc);
        void* r = bo.render_create();
        CHECK(bo.render_forward(r, &oc, n, 0, sc.transforms.data(), sc.sh.data(), sc.raw_opac.data(), bg, 2) == 0, "oracle forward");
        CHECK(bo.render_backward(r, v_host.data(), sc.transforms.data(), sc.sh.data(), sc.raw_opac.data()) == 0, "oracle backward");
        uint64_t cnt = 0;
        const float* p = bo.get_v_transforms(r, &cnt);
        vt.assign(p, p + cnt);
        bo.render_free(r);
    };
    std::vector<float> refA, refB;
    oracle_grads(camA, refA);
    oracle_grads(camB, refB);
    CHECK(rel_linf(refA, refB.data(), refB.size()) > 1e-2f, "the two views must have different gradients for this test to mean anything");
    {
        bh::RenderNode A(ctx, splats, camA, w, h, bg, /*retain=*/true);
        bh::RenderNode B(ctx, splats, camB, w, h, bg);
        CHECK(B.aux.raw.generation == A.aux.raw.generation + 1, "forwards are numbered");
        const float eA = rel_linf(A.backward(v_out.data()).v_transforms.download(), refA.data(), refA.size());
        const float eB = rel_linf(B.backward(v_out.data()).v_transforms.download(), refB.data(), refB.size());
        CHECK(eA <= 1e-4f && eB <= 1e-4f, "retained A after B: gradient rel. error %g (A) / %g (B)", eA, eB);
        const float eA2 = rel_linf(A.backward(v_out.data()).v_transforms.download(), refA.data(), refA.size());
        CHECK(eA2 <= 1e-4f, "a retained node can be replayed again: %g", eA2);
    }
    {
        bh::RenderNode A(ctx, splats, camA, w, h, bg);   // not retained
        bh::RenderNode B(ctx, splats, camB, w, h, bg);
        bool threw = false;
        try { (void)A.backward(v_out.data()); } catch (const bh::Error& e) { threw = e.code == BH_ERR_STATE; }
        CHECK(threw, "backward of a forward that a later forward overwrote must fail with BH_ERR_STATE");
        const float eB = rel_linf(B.backward(v_out.data()).v_transforms.download(), refB.data(), refB.size());
        CHECK(eB <= 1e-4f, "... and the live forward is unharmed: %g", eB);
    }
    std::printf("ok two_forwards_alive\n");
}

static void test_primitives(const bh::Context& ctx) {
    for (uint32_t n : {1u, 1000u, 4097u, 300000u}) {
        Sm64 r{n};
        std::vector<uint32_t> keys(n), vals(n);
        for (uint32_t i = 0; i < n; ++i) { keys[i] = (uint32_t)r.next(); vals[i] = i; }
        for (uint32_t bits : {13u, 32u}) {
            bh::DeviceBuffer<uint32_t> dk(keys), dv(vals), ok, ov;
            bh::radix_argsort(ctx, dk, dv, bits, ok, ov);
            std::vector<uint32_t> idx(n);
            std::iota(idx.begin(), idx.end(), 0u);
            const uint32_t mask = bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (keys[a] & mask) < (keys[b] & mask); });
            const auto gv = ov.download();
            const auto gk = ok.download();
            bool same = true;
            for (uint32_t i = 0; i < n; ++i) same = same && gv[i] == idx[i] && gk[i] == keys[idx[i]];
            CHECK(same, "radix_argsort n=%u bits=%u", n, bits);
        }
        std::vector<uint32_t> small(n);
        for (uint32_t i = 0; i < n; ++i) small[i] = (uint32_t)(r.next() % 90);
        bh::DeviceBuffer<uint32_t> din(small), dout;
        bh::prefix_sum(ctx, din, dout);
        std::vector<uint32_t> ref(n);
        std::partial_sum(small.begin(), small.end(), ref.begin());
        CHECK(dout.download() == ref, "prefix_sum n=%u", n);
        // the forward's tile sort + offsets table as one operator (render.rs:228-243 + get_tile_offset.rs:11-58)
        const uint32_t num_tiles = 8160;
        std::vector<uint32_t> tiles(n);
        for (uint32_t i = 0; i < n; ++i) tiles[i] = (uint32_t)(r.next() % num_tiles);
        bh::DeviceBuffer<uint32_t> dt(tiles), dg(vals), st, sg, offs;
        bh::tile_sort_offsets(ctx, dt, dg, num_tiles, st, sg, offs);
        std::vector<uint32_t> tidx(n);
        std::iota(tidx.begin(), tidx.end(), 0u);
        std::stable_sort(tidx.begin(), tidx.end(), [&](uint32_t a, uint32_t b) { return tiles[a] < tiles[b]; });
        std::vector<uint32_t> roffs((size_t)num_tiles * 2, 0u);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t t = tiles[tidx[i]];
            if (i == 0 || tiles[tidx[i - 1]] != t) roffs[2 * t] = i;
            roffs[2 * t + 1] = i + 1;
        }
        const auto gt = st.download();
        const auto gg = sg.download();
        bool tsame = offs.download() == roffs;
        for (uint32_t i = 0; i < n; ++i) tsame = tsame && gg[i] == tidx[i] && gt[i] == tiles[tidx[i]];
        CHECK(tsame, "tile_sort_offsets n=%u", n);
    }
    std::printf("ok primitives\n");
}

static void test_training_refine_ply(const bh::Context& ctx) {
    const uint32_t n = 3000, w = 128, h = 96;
    const HostScene sc = make_scene(n, 0xBEEF, 1);
    bh::Splats splats = bh::Splats::from_host(sc.transforms, sc.sh, sc.raw_opac);
    std::vector<uint32_t> gt((size_t)w * h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x)
            gt[(size_t)y * w + x] = (uint32_t)(127 + 120 * std::sin(x * 0.05)) | ((uint32_t)(127 + 120 * std::cos(y * 0.07)) << 8) | (128u << 16) | (255u << 24);
    bh::DeviceBuffer<uint32_t> dgt(gt);
    bh::TrainConfig cfg;
    cfg.total_train_iters = 1000;
    bh::SplatTrainer trainer(ctx, cfg, 3.0f);
    trainer.set_view_cams({0.0f, 0.0f, 0.0f, 110.0f});
    bh::SceneBatch batch;
    batch.img_packed = dgt.data(); batch.img_w = w; batch.img_h = h;
    batch.camera.fov_x = 1.0471976; batch.camera.fov_y = 2.0 * std::atan(0.75 * std::tan(1.0471976 / 2));
    float first = 0, last = 0;
    for (int i = 0; i < 10; ++i) {
        const bh::TrainStepStats st = trainer.step(batch, splats);
        if (i == 0) first = st.loss;
        last = st.loss;
        CHECK(std::isfinite(st.loss) && st.num_visible > 0, "step %d", i);
    }
    CHECK(last < first && trainer.step_count() == 10, "loss %g -> %g", first, last);
    const BhRefineStats rs = trainer.refine(200, splats, 42);
    CHECK(rs.total_splats == splats.num_splats() && rs.total_splats > 0, "refine: %u splats", rs.total_splats);
    CHECK(splats.min_scale.has_value(), "refine attaches the recomputed 3D-filter floor (train.rs:636-648)");
    const bh::TrainStepStats st = trainer.step(batch, splats);
    CHECK(std::isfinite(st.loss), "step after refine");
    // PLY round trip (export.rs:305-349): the floor is baked on export
    const float up[3] = {0.0f, 0.0f, 1.0f};
    const std::vector<uint8_t> ply = bh::splat_to_ply(ctx, splats, up);
    auto [back, info] = bh::load_splat_from_ply(ctx, ply);
    CHECK(info.num_splats == splats.num_splats() && info.sh_degree == 0 && info.has_up_axis && info.up_axis[2] == 1.0f, "ply header");
    CHECK(back.sh_coeffs.download() == splats.sh_coeffs.download(), "ply SH round trip");
    bh::Splats baked = bh::Splats::from_host(splats.transforms.download(), splats.sh_coeffs.download(), splats.raw_opacities.download());
    baked.with_min_scale(bh::DeviceBuffer<float>(splats.min_scale->download()));
    baked.bake_min_scale(ctx);
    const auto a = back.transforms.download(), b = baked.transforms.download();
    bool scales_same = true;
    for (size_t i = 0; i < a.size() / 10; ++i)
        for (int k = 7; k < 10; ++k) scales_same = scales_same && a[i * 10 + k] == b[i * 10 + k];
    CHECK(scales_same && back.raw_opacities.download() == baked.raw_opacities.download(), "exported scales / opacities carry the baked floor");
    {   // import.rs:651-670 (subsample_points) and :49-74 (SplatData::subsample)
        auto [half, hinfo] = bh::load_splat_from_ply(ctx, ply, 2);
        const auto full_sh = back.sh_coeffs.download(), half_sh = half.sh_coeffs.download();
        bool same = hinfo.num_splats == info.num_splats / 2 && half_sh.size() == (size_t)hinfo.num_splats * 3;
        for (size_t i = 0; same && i < hinfo.num_splats; ++i)
            for (int k = 0; k < 3; ++k) same = same && half_sh[i * 3 + k] == full_sh[(2 * i + 1) * 3 + k];
        CHECK(same, "subsample_points = 2 keeps rows 1, 3, 5, ...");
        auto [capped, cinfo] = bh::load_splat_from_ply(ctx, ply, 1, 100);
        const uint64_t step = (info.num_splats + 99) / 100;
        CHECK(cinfo.num_splats <= 100 && cinfo.num_splats == (info.num_splats + step - 1) / step, "max_splats caps the count with a ceil step");
        CHECK(capped.sh_coeffs.download()[3] == full_sh[step * 3], "max_splats keeps rows 0, step, 2 step, ...");
    }
    std::printf("ok training_refine_ply  loss %.4f -> %.4f, %u splats after refine, ply %zu bytes\n", first, last, rs.total_splats, ply.size());
}

// LossOps / AdamScaled / RefineRecord through the header vs the oracle (brush-loss/src/lib.rs:718-733, adam_scaled.rs:75-147, stats.rs:40-50)
static void test_loss_optimizer(const bh::Context& ctx, Oracle& bo) {
    const uint32_t h = 40, w = 56;
    Sm64 r{0x10555};
    for (uint32_t ch : {3u, 4u}) {
        std::vector<float> pred((size_t)ch * h * w), dl((size_t)ch * h * w);
        std::vector<uint32_t> gt((size_t)h * w);
        for (float& v : pred) v = r.uni(-0.1f, 1.1f);
        for (float& v : dl) v = r.uni(-1.0f, 1.0f);
        for (uint32_t& v : gt) v = (uint32_t)r.next();
        bh::LossConfig cfg;
        if (ch == 4) { cfg.composite_bg = std::array<float, 3>{0.2f, 0.4f, 0.6f}; }
        const float* bgp = cfg.composite_bg ? cfg.composite_bg->data() : nullptr;
        bh::DeviceBuffer<float> dpred(pred), ddl(dl);
        bh::DeviceBuffer<uint32_t> dgt(gt);
        const auto lm = bh::image_loss(ctx, dpred.data(), dgt.data(), ch, h, w, cfg).download();
        std::vector<float> ref(lm.size()), refg(lm.size());
        bo.image_loss_forward(pred.data(), gt.data(), ch, h, w, cfg.l1_weight, cfg.ssim_weight, bgp, cfg.composite_bg ? 1 : 0, 0, ref.data());
        float d = 0;
        for (size_t i = 0; i < lm.size(); ++i) d = std::max(d, std::fabs(lm[i] - ref[i]));
        CHECK(d <= 2e-6f, "image_loss [%u ch]: loss map max |d| = %g", ch, d);
        const auto g = bh::image_loss_backward(ctx, dpred.data(), dgt.data(), ddl.data(), ch, h, w, cfg).download();
        bo.image_loss_backward(pred.data(), gt.data(), dl.data(), ch, h, w, cfg.l1_weight, cfg.ssim_weight, bgp, cfg.composite_bg ? 1 : 0, 0, refg.data());
        float dg = 0, gm = 1.0f;
        for (size_t i = 0; i < g.size(); ++i) { dg = std::max(dg, std::fabs(g[i] - refg[i])); gm = std::max(gm, std::fabs(refg[i])); }
        CHECK(dg <= 2e-6f * gm, "image_loss_backward [%u ch]: max |d| %g (max |g| %g)", ch, dg, gm);
    }
    {   // the fused value-and-gradient == mean(loss map) and its gradient (train.rs:227-260)
        std::vector<float> img((size_t)h * w * 4), chw((size_t)3 * h * w);
        std::vector<uint32_t> gt((size_t)h * w);
        for (float& v : img) v = r.uni(0.0f, 1.0f);
        for (uint32_t& v : gt) v = (uint32_t)r.next() | 0xFF000000u;
        for (uint32_t c = 0; c < 3; ++c)
            for (size_t p = 0; p < (size_t)h * w; ++p) chw[c * h * w + p] = img[p * 4 + c];
        bh::DeviceBuffer<float> dimg(img);
        bh::DeviceBuffer<uint32_t> dgt(gt);
        auto [loss, v_out] = bh::image_loss_value_and_grad(ctx, dimg.data(), dgt.data(), h, w);
        std::vector<float> ref(chw.size()), dl(chw.size(), 1.0f / (float)chw.size()), refg(chw.size());
        bo.image_loss_forward(chw.data(), gt.data(), 3, h, w, 0.8f, -0.2f, nullptr, 0, 0, ref.data());
        bo.image_loss_backward(chw.data(), gt.data(), dl.data(), 3, h, w, 0.8f, -0.2f, nullptr, 0, 0, refg.data());
        double mean = 0;
        for (float v : ref) mean += v;
        mean /= (double)ref.size();
        CHECK(std::fabs(loss - mean) <= 2e-6 * std::max(1.0, std::fabs(mean)), "fused loss %g vs mean of the loss map %g", loss, mean);
        const auto v = v_out.download();
        double dmax = 0, gmax = 0;
        for (uint32_t c = 0; c < 3; ++c)
            for (size_t p = 0; p < (size_t)h * w; ++p) {
                dmax = std::max(dmax, (double)std::fabs(v[p * 4 + c] - refg[c * h * w + p]));
                gmax = std::max(gmax, (double)std::fabs(refg[c * h * w + p]));
            }
        bool alpha_zero = true;
        for (size_t p = 0; p < (size_t)h * w; ++p) alpha_zero = alpha_zero && v[p * 4 + 3] == 0.0f;
        CHECK(dmax <= 2e-6 * gmax && alpha_zero, "fused dloss/dimg: max |d| %g of %g", dmax, gmax);
    }
    {   // AdamScaled: three steps, bit for bit (column scale on; and the row-reduced second moment of the SH parameters)
        for (int reduce = 0; reduce < 2; ++reduce) {
            const uint64_t rows = 777; const uint32_t len = reduce ? 12 : 10;
            std::vector<float> p(rows * len), g(rows * len), m1(rows * len, 0.0f), m2(reduce ? rows : rows * len, 0.0f), cs(len);
            for (float& v : p) v = r.uni(-1.0f, 1.0f);
            for (uint32_t k = 0; k < len; ++k) cs[k] = k < 3 ? 1.0f : 0.1f;
            bh::DeviceBuffer<float> dp(p), dm1(m1), dm2(m2), dcs(cs), dg(g.size());
            for (uint32_t t = 1; t <= 3; ++t) {
                for (float& v : g) v = r.uni(-1e-3f, 1e-3f);
                dg.upload(g);
                bh::adam_step(ctx, dp.data(), dg.data(), dm1.data(), dm2.data(), rows, len, 2e-3f, t, dcs.data(), reduce != 0);
                bo.adam_step(p.data(), g.data(), m1.data(), m2.data(), rows, len, cs.data(), 2e-3f, t, reduce, 0.9f, 0.999f, 1e-15f);
            }
            ctx.sync();
            CHECK(dp.download() == p && dm1.download() == m1 && dm2.download() == m2, "adam_step (reduce_m2 = %d) is bit-exact after three steps", reduce);
        }
    }
    {   // RefineRecord::gather_stats
        const uint64_t n = 5000;
        std::vector<float> a(n), b(n), c(n), rw(n), vis(n), rad(n);
        for (uint64_t i = 0; i < n; ++i) { a[i] = r.unit(); b[i] = (float)(r.next() % 5); c[i] = r.uni(0, 50); rw[i] = r.unit(); vis[i] = (float)(r.next() & 1); rad[i] = r.uni(0, 60); }
        bh::DeviceBuffer<float> da(a), db(b), dc(c), drw(rw), dvis(vis), drad(rad);
        bh::gather_stats(ctx, da.data(), db.data(), dc.data(), drw.data(), dvis.data(), drad.data(), n);
        ctx.sync();
        bo.gather_stats(a.data(), b.data(), c.data(), rw.data(), vis.data(), rad.data(), n);
        CHECK(da.download() == a && db.download() == b && dc.download() == c, "gather_stats is exact");
    }
    std::printf("ok loss_optimizer\n");
}

// BatchUploader / SceneLoader (scene.rs:97-136, scene_loader.rs:59-174), the list controls, the profile, the generator, the communicator
static void test_loader_controls_comm(const bh::Context& ctx) {
    const uint32_t w = 64, h = 48, nviews = 5;
    auto pixel = [](uint32_t v, uint32_t x, uint32_t y, uint32_t c) { return (uint8_t)((x * 3 + y * 5 + v * 37 + c * 91) & 255u); };
    std::vector<bh::LoaderView> views;
    for (uint32_t v = 0; v < nviews; ++v) {
        bh::LoaderView lv;
        lv.w = w; lv.h = h; lv.channels = v == 3 ? 4 : 3;   // view 3 carries an alpha channel: premultiplied on the device
        const uint32_t ch = lv.channels;
        lv.decode = [=](uint8_t* dst) {
            for (uint32_t y = 0; y < h; ++y)
                for (uint32_t x = 0; x < w; ++x)
                    for (uint32_t c = 0; c < ch; ++c) dst[((size_t)y * w + x) * ch + c] = pixel(v, x, y, c);
        };
        lv.camera.fov_x = 1.0471976; lv.camera.fov_y = 2.0 * std::atan(0.75 * std::tan(1.0471976 / 2));
        lv.camera.position[0] = 0.05f * (float)v;
        views.push_back(lv);
    }
    bh::SceneLoader loader(ctx, views, /*seed=*/7);
    CHECK(loader.epoch_order(0) == loader.epoch_order(0) && loader.epoch_order(0) != loader.epoch_order(1), "epoch orders are seeded permutations");
    const HostScene sc = make_scene(3000, 0xFEED, 1);
    bh::Splats splats = bh::Splats::from_host(sc.transforms, sc.sh, sc.raw_opac);
    bh::TrainConfig cfg;
    cfg.total_train_iters = 1000;
    bh::SplatTrainer trainer(ctx, cfg, 3.0f);
    trainer.set_seed(0xB5EED);
    ctx.set_list_cut_threshold(0);   // (a small scene: let the per-tile cuts engage at all)
    // bh_set_option: the one configuration entry point (the library reads no environment variable)
    ctx.set_option("event_waits", "1");   // the host's mid-step waits through events: same results
    {
        bool threw = false;
        try { ctx.set_option("no_such_option", "1"); } catch (const bh::Error& e) { threw = e.code == BH_ERR_INVALID_ARG; }
        bool threw2 = false;
        try { ctx.set_option("k16_order", "7"); } catch (const bh::Error& e) { threw2 = e.code == BH_ERR_INVALID_ARG; }
        CHECK(threw && threw2 && bh::Context::options().size() >= 20, "set_option rejects unknown keys / bad values; %zu keys documented", bh::Context::options().size());
    }
    ctx.profile(1);
    bool pixels_ok = true, finite = true;
    for (uint32_t epoch = 0; epoch < 3; ++epoch) {
        std::vector<int> seen(nviews, 0);
        for (uint32_t k = 0; k < nviews; ++k) {
            const bh::SceneBatch b = loader.next_batch();
            CHECK(b.view_id >= 1 && b.view_id <= nviews && b.img_w == w && b.img_h == h, "batch of view %u", b.view_id);
            const uint32_t v = b.view_id - 1;
            seen[v]++;
            CHECK(b.has_alpha == (v == 3), "has_alpha follows the channel count");
            const bh::TrainStepStats st = trainer.step(b, splats);   // queued on the ctx stream behind the upload
            finite = finite && std::isfinite(st.loss);
            if (epoch == 0) {   // the packed image: widened (a = 255) or byte-space premultiplied ((c * a + 127) / 255)
                const auto got = bh::download(b.img_packed, (size_t)w * h);
                for (uint32_t y = 0; y < h && pixels_ok; ++y)
                    for (uint32_t x = 0; x < w; ++x) {
                        uint32_t c[4] = {pixel(v, x, y, 0), pixel(v, x, y, 1), pixel(v, x, y, 2), v == 3 ? pixel(v, x, y, 3) : 255u};
                        if (v == 3) for (int k2 = 0; k2 < 3; ++k2) c[k2] = (c[k2] * c[3] + 127u) / 255u;
                        if (got[(size_t)y * w + x] != (c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24))) { pixels_ok = false; break; }
                    }
            }
        }
        bool once = true;
        for (int c : seen) once = once && c == 1;
        CHECK(once, "epoch %u visits every view exactly once", epoch);
    }
    CHECK(pixels_ok, "uploaded batches equal view_to_packed_data of the host bytes");
    CHECK(finite && trainer.step_count() == 3 * nviews, "15 steps through the loader");
    const auto stages = ctx.profile_fetch();
    bool has_blend = false, has_update = false;
    for (const auto& st : stages) { has_blend = has_blend || st.name == "Rasterize"; has_update = has_update || st.name == "OptimizerStep"; }
    CHECK(has_blend && has_update && !stages.empty(), "the stage profile names the pipeline's stages (%zu of them)", stages.size());
    ctx.profile(0);
    const BhRenderOut lo = ctx.last_render_out();
    const auto counts = ctx.last_list_counts();
    CHECK(lo.generation >= 15 && lo.num_intersections > 0 && counts.first > 0 && counts.first <= lo.num_intersections, "last_render_out / last_list_counts");
    CHECK(ctx.last_list_share() > 0.0f && ctx.last_list_share() <= 1.0f, "last_list_share %g", ctx.last_list_share());
    ctx.forget_views();
    ctx.set_view_id(3); ctx.set_view_id(0);
    ctx.set_list_slicing(0.0f);
    // the generator
    const float base[3] = {0.2f, 0.5f, 0.8f};
    const auto bg1 = bh::sample_background(0xB5EED, 7, base, 0.1f), bg2 = bh::sample_background(0xB5EED, 7, base, 0.1f), bg3 = bh::sample_background(0xB5EED, 8, base, 0.1f);
    CHECK(bg1 == bg2 && bg1 != bg3 && bg1[0] >= 0.1f && bg1[0] <= 0.3f, "sample_background is a function of (seed, step)");
    const auto ns = bh::normal_samples(ctx, 0xB5EED, 3, 20000).download();
    double m = 0, v2 = 0;
    for (float x : ns) { m += x; v2 += (double)x * x; }
    m /= (double)ns.size(); v2 = v2 / (double)ns.size() - m * m;
    CHECK(std::fabs(m) < 0.02 && std::fabs(v2 - 1.0) < 0.03, "normal_samples: mean %g variance %g", m, v2);
    const auto kat = bh::philox4x32_10({0u, 0u, 0u, 0u}, {0u, 0u});
    CHECK(kat[0] == 0x6627e8d5u && kat[1] == 0xe169c58du && kat[2] == 0xbc57ac4cu && kat[3] == 0x9b00dbd8u, "Philox-4x32-10 known answer (Random123)");
    // the communicator: the strip plan is host arithmetic; a one-rank RCCL group exercises the binding when RCCL is loadable
    CHECK(ctx.comm_world() == 1 && ctx.comm_rank() == 0, "no communicator: world 1, rank 0");
    const auto plan = bh::Context::strip_halo_plan(1080, 360, 720, 1, 3);
    int sends = 0, recvs = 0;
    for (const BhHaloOp& op : plan) { (op.send ? sends : recvs)++; CHECK(op.rows == 21 && (op.peer == 0 || op.peer == 2), "halo op: 21 rows to / from a neighbour"); }
    CHECK(plan.size() == 4 && sends == 2 && recvs == 2, "a middle strip sends and receives two halos");
    try {
        const auto id = bh::Context::comm_unique_id();
        ctx.comm_init(0, 1, id);
        ctx.comm_selftest();
        std::vector<float> x(1000);
        for (size_t i = 0; i < x.size(); ++i) x[i] = (float)i * 0.5f;
        bh::DeviceBuffer<float> dx(x);
        ctx.allreduce_sum(dx.data(), x.size());
        ctx.allreduce_max(dx.data(), x.size());
        ctx.sync();
        CHECK(dx.download() == x && ctx.comm_world() == 1, "one-rank RCCL all-reduce is the identity");
        ctx.comm_destroy();
        std::printf("   (RCCL bound and self-tested on one rank)\n");
    } catch (const bh::Error& e) {
        std::printf("   (communicator part skipped: %s)\n", e.what());
    }
    std::printf("ok loader_controls_comm\n");
}

static void test_errors(const bh::Context& ctx) {
    const HostScene sc = make_scene(10, 1, 1);
    bh::Splats splats = bh::Splats::from_host(sc.transforms, sc.sh, sc.raw_opac);
    const float bg[3] = {0, 0, 0};
    bool threw = false;
    try { bh::render_splats(ctx, splats, bh::Camera{}, 0, 16, bg); } catch (const bh::Error&) { threw = true; }
    CHECK(threw, "zero-size image must fail (render.rs:50-53)");
    threw = false;
    try { bh::DeviceBuffer<uint32_t> k(std::vector<uint32_t>(8, 1u)), v(std::vector<uint32_t>(7, 1u)), a, b; bh::radix_argsort(ctx, k, v, 32, a, b); }
    catch (const bh::Error&) { threw = true; }
    CHECK(threw, "mismatched key / value counts must fail (brush-sort/src/lib.rs:21-33)");
    threw = false;
    try { bh::Splats::from_host(sc.transforms, std::vector<float>(10 * 2 * 3, 0.0f), sc.raw_opac); } catch (const bh::Error&) { threw = true; }
    CHECK(threw, "2 SH coefficients is not (d+1)^2");
    CHECK(std::fabs(bh::fov_to_focal(bh::focal_to_fov(800.0, 1920), 1920) - 800.0) < 1e-9, "fov round trip (tests/mod.rs:711-719)");
    std::printf("ok errors\n");
}

int main(int argc, char** argv) {
    Oracle bo;
    if (!bo.load(argc > 1 ? argv[1] : "oracle/libbrush_oracle.so")) return 2;
    try {
        bh::Context ctx(0);
        test_render_vs_oracle(ctx, bo, bh::CameraModel::Pinhole, "pinhole");
        test_render_vs_oracle(ctx, bo, bh::CameraModel::KannalaBrandt4, "kb4");
        test_render_vs_oracle(ctx, bo, bh::CameraModel::RadialTangential8, "rt8");
        test_two_forwards_alive(ctx, bo);
        test_primitives(ctx);
        test_training_refine_ply(ctx);
        test_loss_optimizer(ctx, bo);
        test_loader_controls_comm(ctx);
        test_errors(ctx);
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
    std::printf(g_failed ? "FAILED %d checks\n" : "all C++ host checks passed\n", g_failed);
    return g_failed ? 1 : 0;
}
