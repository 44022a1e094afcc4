// test_host.cpp — the C++ host mirror (include/brush_hip.hpp) exercised like the reference's own tests, on the GPU:
//   * forward / backward of a seeded scene vs the CPU oracle (oracle/libbrush_oracle.so, dlopen'd: the oracle is
//     test infrastructure and only tests may touch it): counts and per-tile sort order exact, image <= 1e-6,
//     gradients <= 1e-4 * max|g|   (crates/brush-bench-test/src/reference.rs, tests/finite_diff.rs roles)
//   * radix_argsort / prefix_sum vs std::stable_sort / std::partial_sum on SplitMix64 inputs
//     (brush-sort/src/lib.rs:154-339, brush-prefix-sum/src/lib.rs:105-196)
//   * SplatTrainer::step x N + refine + PLY export / import round trip (tests/integration.rs:186-235, export.rs:305-349)
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
    bool load(const char* path) {
        lib = dlopen(path, RTLD_NOW);
        if (!lib) { std::printf("cannot load oracle %s: %s\n", path, dlerror()); return false; }
#define SYM(field, name) field = (decltype(field))dlsym(lib, name); if (!field) { std::printf("oracle symbol %s missing\n", name); return false; }
        SYM(camera_setup_model, "bo_camera_setup_model") SYM(render_create, "bo_render_create") SYM(render_free, "bo_render_free")
        SYM(render_forward, "bo_render_forward") SYM(render_backward, "bo_render_backward") SYM(num_visible, "bo_num_visible")
        SYM(num_intersections, "bo_num_intersections") SYM(get_out_img, "bo_get_out_img") SYM(get_cgfi, "bo_get_compact_gid_from_isect")
        SYM(get_v_transforms, "bo_get_v_transforms") SYM(get_v_raw_opac, "bo_get_v_raw_opac")
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
        test_errors(ctx);
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
    std::printf(g_failed ? "FAILED %d checks\n" : "all C++ host checks passed\n", g_failed);
    return g_failed ? 1 : 0;
}
