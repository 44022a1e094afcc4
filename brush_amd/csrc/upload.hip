// upload.hip — host images -> packed rgba8 device batches, overlapped with training.
//
// Reference: brush-dataset/src/scene.rs:97-136 (view_to_packed_data / pack_rgba: widen RGB8 to RGBA8 with
// a = 255, premultiply transparent-alpha views in byte space, pack 4 bytes per pixel little-endian — one
// host walk over every image) and brush-dataset/src/scene_loader.rs:59-174 (a channel of depth 4 between
// the loader tasks and the trainer; the upload itself is wgpu's staging copy on the trainer's queue).
//
// MI355X shape: the widening / premultiply / pack walk moves to the device, so the host only ever touches
// the decoded bytes once — ideally the decoder writes them straight into a pinned slot (bh_uploader_begin)
// — and PCIe carries 3 B/pixel for RGB views instead of 4.  A ring of pinned slots + one copy stream:
// slot k+1's H2D copy and pack kernel run while the train step reads slot k; the hand-over to the ctx
// stream is an event wait on the device (no host block), the hand-back an event recorded on the ctx
// stream.  At 1080p an RGB view is 6.2 MB: ~0.12 ms of PCIe Gen5 against a 1.5 ms step.
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "context.h"

namespace bh {

constexpr int UP_WG = 256;

// scene.rs:111-116: RGB8 -> r | g << 8 | b << 16 | 255 << 24.  Four pixels per thread = three dwords in, four out.
__global__ __launch_bounds__(UP_WG) void pack_rgb8_kernel(const uint32_t* __restrict__ rgb, uint64_t pixels, uint32_t* __restrict__ out) {
    const uint64_t q = (uint64_t)blockIdx.x * UP_WG + threadIdx.x;  // group of 4 pixels
    const uint64_t p0 = q * 4;
    if (p0 >= pixels) return;
    if (p0 + 4 <= pixels) {
        const uint32_t a = rgb[q * 3], b = rgb[q * 3 + 1], c = rgb[q * 3 + 2];
        uint4 o;
        o.x = (a & 0x00FFFFFFu) | 0xFF000000u;
        o.y = (a >> 24) | ((b & 0x0000FFFFu) << 8) | 0xFF000000u;
        o.z = (b >> 16) | ((c & 0x000000FFu) << 16) | 0xFF000000u;
        o.w = (c >> 8) | 0xFF000000u;
        *reinterpret_cast<uint4*>(out + p0) = o;
    } else {  // ragged tail: byte reads (the staging buffer is padded to a dword, never read past it)
        const uint8_t* bytes = reinterpret_cast<const uint8_t*>(rgb);
        for (uint64_t p = p0; p < pixels; ++p)
            out[p] = (uint32_t)bytes[p * 3] | ((uint32_t)bytes[p * 3 + 1] << 8) | ((uint32_t)bytes[p * 3 + 2] << 16) | 0xFF000000u;
    }
}

// scene.rs:124-136 pack_rgba: mul(c) = (c * a + 127) / 255 in integer arithmetic when premultiplying.
__global__ __launch_bounds__(UP_WG) void pack_rgba8_kernel(const uint32_t* __restrict__ rgba, uint64_t pixels, int premultiply,
                                                          uint32_t* __restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * UP_WG + threadIdx.x;
    if (p >= pixels) return;
    const uint32_t v = rgba[p];
    if (!premultiply) { out[p] = v; return; }
    const uint32_t a = v >> 24;
    const uint32_t r = ((v & 0xFFu) * a + 127u) / 255u;
    const uint32_t g = (((v >> 8) & 0xFFu) * a + 127u) / 255u;
    const uint32_t b = (((v >> 16) & 0xFFu) * a + 127u) / 255u;
    out[p] = r | (g << 8) | (b << 16) | (a << 24);
}

}  // namespace bh

struct bh_uploader {
    bh_ctx* ctx = nullptr;
    hipStream_t copy_stream = nullptr;
    uint64_t max_pixels = 0;
    struct Slot {
        uint8_t* pinned = nullptr;   // max_pixels * 4 bytes (+ pad)
        uint8_t* raw_dev = nullptr;  // staging of the unpacked bytes
        uint32_t* packed = nullptr;  // [H,W] rgba8
        hipEvent_t ready = nullptr;     // recorded on copy_stream after the pack kernel
        hipEvent_t consumed = nullptr;  // recorded on the ctx stream by release
        uint32_t w = 0, h = 0, channels = 0;
        int state = 0;  // 0 free, 1 mapped (begin), 2 in flight / ready, 3 acquired
        bool consumed_pending = false;
    };
    std::vector<Slot> slots;
    uint32_t next = 0;
    std::mutex mu;  // submit side may live on a loader thread; acquire / release on the ctx thread
    std::string last_error;
};

using namespace bh;

static int up_fail(bh_uploader* up, int code, const char* msg) {
    if (up) up->last_error = msg;
    return code;
}

extern "C" {

bh_uploader* bh_uploader_create(bh_ctx* ctx, uint64_t max_pixels, uint32_t num_slots) {
    if (!ctx || max_pixels == 0 || num_slots < 2 || num_slots > 16) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    bh_uploader* up = new (std::nothrow) bh_uploader();
    if (!up) return nullptr;
    up->ctx = ctx;
    up->max_pixels = max_pixels;
    bool ok = hipStreamCreateWithFlags(&up->copy_stream, hipStreamNonBlocking) == hipSuccess;
    up->slots.resize(num_slots);
    const size_t bytes = (size_t)max_pixels * 4 + 16;
    for (auto& s : up->slots) {
        ok = ok && hipHostMalloc((void**)&s.pinned, bytes, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipMalloc((void**)&s.raw_dev, bytes) == hipSuccess;
        ok = ok && hipMalloc((void**)&s.packed, (size_t)max_pixels * 4) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        bh_uploader_destroy(up);
        return nullptr;
    }
    return up;
}

void bh_uploader_destroy(bh_uploader* up) {
    if (!up) return;
    (void)hipSetDevice(up->ctx->device);
    if (up->copy_stream) (void)hipStreamSynchronize(up->copy_stream);
    for (auto& s : up->slots) {
        if (s.consumed_pending && s.consumed) (void)hipEventSynchronize(s.consumed);
        if (s.pinned) (void)hipHostFree(s.pinned);
        if (s.raw_dev) (void)hipFree(s.raw_dev);
        if (s.packed) (void)hipFree(s.packed);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
    }
    if (up->copy_stream) (void)hipStreamDestroy(up->copy_stream);
    delete up;
}

const char* bh_uploader_last_error(bh_uploader* up) { return up ? up->last_error.c_str() : "null uploader"; }

int bh_uploader_begin(bh_uploader* up, uint64_t bytes, void** pinned) {
    if (!up || !pinned) return BH_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(up->mu);
    if (bytes > up->max_pixels * 4) return up_fail(up, BH_ERR_INVALID_ARG, "uploader_begin: image larger than the slot size");
    const uint32_t idx = up->next;
    bh_uploader::Slot& s = up->slots[idx];
    if (s.state == 1 || s.state == 3) return up_fail(up, BH_ERR_STATE, "uploader_begin: the next slot is still mapped or acquired (release it first)");
    if (s.state == 2) return up_fail(up, BH_ERR_STATE, "uploader_begin: ring full — the oldest batch was never acquired");
    if (hipSetDevice(up->ctx->device) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipSetDevice"); }
    if (s.consumed_pending) {  // the step that read this slot's packed image must have finished
        if (hipEventSynchronize(s.consumed) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipEventSynchronize(consumed)"); }
        s.consumed_pending = false;
    }
    s.state = 1;
    up->next = (idx + 1) % (uint32_t)up->slots.size();
    *pinned = s.pinned;
    return (int)idx;
}

int bh_uploader_commit(bh_uploader* up, int slot, uint32_t w, uint32_t h, uint32_t channels, int premultiply) {
    if (!up || slot < 0 || slot >= (int)up->slots.size()) return BH_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(up->mu);
    bh_uploader::Slot& s = up->slots[slot];
    if (s.state != 1) return up_fail(up, BH_ERR_STATE, "uploader_commit: slot was not mapped with uploader_begin");
    const uint64_t pixels = (uint64_t)w * h;
    if (pixels == 0 || pixels > up->max_pixels || (channels != 3 && channels != 4)) {
        s.state = 0;
        return up_fail(up, BH_ERR_INVALID_ARG, "uploader_commit: image must be RGB8 or RGBA8 and fit the slot");
    }
    if (hipSetDevice(up->ctx->device) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipSetDevice"); }
    const size_t bytes = ((size_t)pixels * channels + 3) & ~(size_t)3;
    hipError_t e = hipMemcpyAsync(s.raw_dev, s.pinned, bytes, hipMemcpyHostToDevice, up->copy_stream);
    if (e == hipSuccess) {
        if (channels == 3) {
            const uint64_t groups = (pixels + 3) / 4;
            hipLaunchKernelGGL(pack_rgb8_kernel, dim3((unsigned)((groups + UP_WG - 1) / UP_WG)), dim3(UP_WG), 0, up->copy_stream,
                               (const uint32_t*)s.raw_dev, pixels, s.packed);
        } else {
            hipLaunchKernelGGL(pack_rgba8_kernel, dim3((unsigned)((pixels + UP_WG - 1) / UP_WG)), dim3(UP_WG), 0, up->copy_stream,
                               (const uint32_t*)s.raw_dev, pixels, premultiply, s.packed);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(s.ready, up->copy_stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        s.state = 0;
        return up_fail(up, BH_ERR_HIP, hipGetErrorString(e));
    }
    s.w = w; s.h = h; s.channels = channels;
    s.state = 2;
    return 0;
}

int bh_uploader_submit(bh_uploader* up, const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t channels, int premultiply) {
    if (!up || !pixels) return BH_ERR_INVALID_ARG;
    void* dst = nullptr;
    const uint64_t bytes = (uint64_t)w * h * channels;
    const int slot = bh_uploader_begin(up, bytes, &dst);
    if (slot < 0) return slot;
    std::memcpy(dst, pixels, (size_t)bytes);
    const int rc = bh_uploader_commit(up, slot, w, h, channels, premultiply);
    return rc < 0 ? rc : slot;
}

int bh_uploader_acquire(bh_uploader* up, int slot, const uint32_t** packed, uint32_t* w, uint32_t* h, int* has_alpha) {
    if (!up || slot < 0 || slot >= (int)up->slots.size() || !packed) return BH_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(up->mu);
    bh_uploader::Slot& s = up->slots[slot];
    if (s.state != 2) return up_fail(up, BH_ERR_STATE, "uploader_acquire: slot holds no committed image");
    if (hipSetDevice(up->ctx->device) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipSetDevice"); }
    // device-side hand-over: work queued on the ctx stream from here on runs after the upload + pack
    if (hipStreamWaitEvent(up->ctx->stream, s.ready, 0) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipStreamWaitEvent"); }
    s.state = 3;
    *packed = s.packed;
    if (w) *w = s.w;
    if (h) *h = s.h;
    if (has_alpha) *has_alpha = s.channels == 4;  // image.color().has_alpha(), scene.rs:100
    return 0;
}

int bh_uploader_release(bh_uploader* up, int slot) {
    if (!up || slot < 0 || slot >= (int)up->slots.size()) return BH_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(up->mu);
    bh_uploader::Slot& s = up->slots[slot];
    if (s.state != 3) return up_fail(up, BH_ERR_STATE, "uploader_release: slot was not acquired");
    if (hipSetDevice(up->ctx->device) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipSetDevice"); }
    if (hipEventRecord(s.consumed, up->ctx->stream) != hipSuccess) { (void)hipGetLastError(); return up_fail(up, BH_ERR_HIP, "hipEventRecord"); }
    s.consumed_pending = true;
    s.state = 0;
    return 0;
}

}  // extern "C"
