// comm.hip — RCCL collectives inside the library (bh_allreduce_sum_f32 / bh_allreduce_max_f32),
// for a host that has no collective layer of its own (the Rust pipeline): one process per GPU, one communicator
// per bh_ctx, every collective queued on the ctx stream, in place.
//
// The reference is single-GPU; this is the exchange step of the data-parallel-over-cameras mode of SURVEY §8e:
// with a communicator attached and no bh_grad_hook, bh_train_step sums its ONE exchange buffer itself.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: a process that already carries an RCCL — a
// PyTorch host does, in torch/lib — keeps using that one copy, and a single-GPU user never loads it at all.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "context.h"

namespace bh {

// Types, enum values and signatures come from the toolchain's own <rccl/rccl.h> (compile time only: decltype of the declared
// functions), so a mismatch between this file and the library's ABI is a compile error, not a silent wrong collective.  The
// library itself is still resolved at run time.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
static_assert(sizeof(ncclUniqueId) == 128, "bh_comm_unique_id hands out 128 bytes");

static RcclApi& rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    // an RCCL already in the process (PyTorch's) first, then the ROCm one
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
        if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!api.lib)
        for (const char* n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!api.lib) {
        api.error = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed");
        return api;
    }
    auto sym = [&](const char* s) {
        void* p = dlsym(api.lib, s);
        if (!p && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + s;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    return api;
}

static int rccl_check(bh_ctx* ctx, ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return 0;
    RcclApi& a = rccl();
    return set_error(ctx, BH_ERR_HIP, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "RCCL error"));
}

#ifdef BH_TEST_HOOKS
// BH_BREAK_ALLREDUCE (test-hook build only, read once at bh_create): corrupt every all-reduce's first element — exists so that
// bench.py's pre-timing self-check of the exchange path can be shown to catch a broken collective (tests/test_gpu_bench_selfcheck.py)
__global__ void break_allreduce_kernel(float* buf) { buf[0] += 1.0f; }
#endif

// ---- direct all-reduce: reduce-scatter + all-gather over grouped send / recv ------------------------------------------------------
// The 8 GPUs of an MI355X node are FULLY connected (7 xGMI links per GPU): every rank can talk to every other at
// the same time.  The buffer is cut into `world` chunks, rank r owns chunk r:
//   1. every rank sends chunk p of its buffer to rank p and receives the other ranks' versions of its own chunk (one grouped
//      exchange: world - 1 sends and receives in flight at once, one per link);
//   2. one kernel adds the world versions up IN RANK ORDER (the result of a chunk is computed once, by its owner: replicas stay
//      bit-identical whatever the transport does);
//   3. every rank sends its finished chunk to everybody and receives theirs in place (the second grouped exchange).
// Each link carries count / world floats per phase.  Selectable beside ncclAllReduce (bh_set_option grad_allreduce = direct) for
// the dense gradient block; MAX reductions and short messages stay with ncclAllReduce.
__global__ __launch_bounds__(256) void reduce_versions_kernel(float* __restrict__ own, const float* __restrict__ others, uint32_t len, uint32_t stride,
                                                             int rank, int world) {
    // others: [world - 1][stride], the version of rank p at slot p - (p > rank)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < len; i += gridDim.x * 256u) {
        float acc = 0.0f;
        for (int p = 0; p < world; ++p) acc += p == rank ? own[i] : others[(size_t)(p - (p > rank ? 1 : 0)) * stride + i];
        own[i] = acc;
    }
}

// chunk c of `count` floats cut for `world` ranks: [begin, begin + len), chunk sizes a multiple of 4 floats (16-byte aligned pieces)
void direct_chunk(uint64_t count, int world, int c, uint64_t* begin, uint64_t* len) {
    const uint64_t per = ((count + (uint64_t)world - 1) / (uint64_t)world + 3ull) & ~3ull;
    const uint64_t b = (uint64_t)c * per;
    *begin = b < count ? b : count;
    *len = b < count ? ((count - b) < per ? (count - b) : per) : 0ull;
}

static int comm_allreduce_direct(bh_ctx* ctx, float* buf, uint64_t count) {
    RcclApi& a = rccl();
    const int W = ctx->comm_world, R = ctx->comm_rank;
    uint64_t my_b = 0, my_n = 0, b0 = 0, per = 0;
    direct_chunk(count, W, R, &my_b, &my_n);
    direct_chunk(count, W, 0, &b0, &per);
    auto* scratch = (float*)ensure(ctx, SLOT_COMM_SCRATCH, (size_t)(W - 1) * per * 4);
    if (!scratch) return BH_ERR_OOM;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    // 1. reduce-scatter
    BH_TRY(rccl_check(ctx, a.GroupStart(), "ncclGroupStart"));
    int rc = 0;
    for (int p = 0; p < W && rc == 0; ++p) {
        if (p == R) continue;
        uint64_t pb = 0, pn = 0;
        direct_chunk(count, W, p, &pb, &pn);
        if (pn) rc = rccl_check(ctx, a.Send(buf + pb, (size_t)pn, ncclFloat32, p, comm, ctx->stream), "ncclSend");
        if (rc == 0 && my_n) rc = rccl_check(ctx, a.Recv(scratch + (size_t)(p - (p > R ? 1 : 0)) * per, (size_t)my_n, ncclFloat32, p, comm, ctx->stream), "ncclRecv");
    }
    int rc2 = rccl_check(ctx, a.GroupEnd(), "ncclGroupEnd");
    if (rc || rc2) return rc ? rc : rc2;
    // 2. the owner's sum, in rank order
    if (my_n) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((my_n + 255) / 256, 2048);
        hipLaunchKernelGGL(reduce_versions_kernel, dim3(blocks), dim3(256), 0, ctx->stream, buf + my_b, scratch, (uint32_t)my_n, (uint32_t)per, R, W);
        BH_LAUNCH_CHECK(ctx, "reduce_versions_kernel");
    }
    // 3. all-gather, in place
    BH_TRY(rccl_check(ctx, a.GroupStart(), "ncclGroupStart"));
    for (int p = 0; p < W && rc == 0; ++p) {
        if (p == R) continue;
        uint64_t pb = 0, pn = 0;
        direct_chunk(count, W, p, &pb, &pn);
        if (my_n) rc = rccl_check(ctx, a.Send(buf + my_b, (size_t)my_n, ncclFloat32, p, comm, ctx->stream), "ncclSend");
        if (rc == 0 && pn) rc = rccl_check(ctx, a.Recv(buf + pb, (size_t)pn, ncclFloat32, p, comm, ctx->stream), "ncclRecv");
    }
    rc2 = rccl_check(ctx, a.GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc2;
}

int comm_allreduce(bh_ctx* ctx, float* buf, uint64_t count, bool max_op) {
    if (!ctx->comm) return set_error(ctx, BH_ERR_STATE, "no communicator: call bh_comm_init first");
    if (count == 0) return 0;
    if (ctx->knob_direct_allreduce && !max_op && ctx->comm_world > 1 && count >= DIRECT_ALLREDUCE_MIN_FLOATS && count / (uint64_t)ctx->comm_world < 0xFFFFFFF0ull) {
        int rc = comm_allreduce_direct(ctx, buf, count);
#ifdef BH_TEST_HOOKS
        if (rc == 0 && ctx->knob_break_allreduce) hipLaunchKernelGGL(break_allreduce_kernel, dim3(1), dim3(1), 0, ctx->stream, buf);
#endif
        return rc;
    }
    // (a one-rank communicator goes through RCCL like any other: the binding is the same code at every world size)
    int rc = rccl_check(ctx, rccl().AllReduce(buf, buf, (size_t)count, ncclFloat32, max_op ? ncclMax : ncclSum, (ncclComm_t)ctx->comm, ctx->stream),
                        "ncclAllReduce");
#ifdef BH_TEST_HOOKS
    if (rc == 0 && ctx->knob_break_allreduce) hipLaunchKernelGGL(break_allreduce_kernel, dim3(1), dim3(1), 0, ctx->stream, buf);
#endif
    return rc;
}

// ---- one frame split into strips of tile rows: the 21-px halos of the strip-wise loss -----------------------------
// Which rows go where: strips lie in rank order, rank r's directly above rank r + 1's; every strip is at least `halo` rows tall
// (the caller's precondition, identical on all ranks).  Rank r then sends its first rows up and its last rows down, and receives
// the rows just above / below its strip.  Pure host arithmetic, the same on both ends of every message.
int strip_halo_plan(uint32_t h, uint32_t row_begin_px, uint32_t row_end_px, int rank, int world, uint32_t halo, BhHaloOp out[4]) {
    int k = 0;
    if (row_begin_px >= row_end_px || row_end_px > h) return -1;
    // What goes to a neighbour is what the NEIGHBOUR expects from the geometry both sides know (where the strips meet, the image
    // height), never something that depends on this strip's own height: a rank whose strip is too short still posts messages of
    // the sizes its neighbours wait for (and reports its error afterwards) instead of leaving them blocked in ncclRecv.
    if (rank > 0 && row_begin_px > 0) {
        const uint32_t up = row_begin_px < halo ? row_begin_px : halo;                    // rows [b - up, b) come from above
        const uint32_t mine = (h - row_begin_px) < halo ? (h - row_begin_px) : halo;      // the rank above expects the rows [b, b + mine)
        out[k++] = BhHaloOp{/*send=*/1, rank - 1, row_begin_px, mine};
        out[k++] = BhHaloOp{/*send=*/0, rank - 1, row_begin_px - up, up};
    }
    if (rank < world - 1 && row_end_px < h) {
        const uint32_t down = (h - row_end_px) < halo ? (h - row_end_px) : halo;          // rows [e, e + down) come from below
        const uint32_t mine = row_end_px < halo ? row_end_px : halo;                       // the rank below expects the rows [e - mine, e)
        out[k++] = BhHaloOp{/*send=*/1, rank + 1, row_end_px - mine, mine};
        out[k++] = BhHaloOp{/*send=*/0, rank + 1, row_end_px, down};
    }
    return k;
}

int comm_exchange_strip_halos(bh_ctx* ctx, float* img, uint32_t h, uint32_t w, uint32_t row_begin_px, uint32_t row_end_px) {
    if (!ctx->comm) return set_error(ctx, BH_ERR_STATE, "no communicator: call bh_comm_init first");
    constexpr uint32_t HALO = 21;   // one 16-px tile row + the 5-px reach of the 11-tap SSIM window (loss_fused.hip)
    // (a strip shorter than the halo is this rank's error — but reported AFTER the grouped exchange below, whose message sizes do
    //  not depend on it: returning here would leave the neighbours blocked in their ncclRecv)
    const bool too_short = ctx->comm_world > 1 && row_end_px - row_begin_px < HALO;
    BhHaloOp ops[4];
    const int k = strip_halo_plan(h, row_begin_px, row_end_px, ctx->comm_rank, ctx->comm_world, HALO, ops);
    if (k < 0) return set_error(ctx, BH_ERR_INVALID_ARG, "strip halo exchange: bad strip");
    if (k == 0) return 0;
    RcclApi& a = rccl();
    const size_t row = (size_t)w * 4;
    BH_TRY(rccl_check(ctx, a.GroupStart(), "ncclGroupStart"));
    int rc = 0;
    for (int i = 0; i < k && rc == 0; ++i) {
        float* p = img + (size_t)ops[i].row_begin_px * row;
        rc = ops[i].send ? rccl_check(ctx, a.Send(p, ops[i].rows * row, ncclFloat32, ops[i].peer, (ncclComm_t)ctx->comm, ctx->stream), "ncclSend")
                         : rccl_check(ctx, a.Recv(p, ops[i].rows * row, ncclFloat32, ops[i].peer, (ncclComm_t)ctx->comm, ctx->stream), "ncclRecv");
    }
    const int rc2 = rccl_check(ctx, a.GroupEnd(), "ncclGroupEnd");
    if (rc || rc2) return rc ? rc : rc2;
    if (too_short) return set_error(ctx, BH_ERR_INVALID_ARG, "strip-wise loss: every rank's strip must be at least 21 pixel rows tall");
    return 0;
}

}  // namespace bh

using namespace bh;

extern "C" {

int bh_comm_unique_id(void* out_id) {
    if (!out_id) return BH_ERR_INVALID_ARG;
    RcclApi& a = rccl();
    if (!a.lib || !a.error.empty()) return BH_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess) return BH_ERR_HIP;
    std::memcpy(out_id, &id, sizeof id);
    return 0;
}

int bh_comm_init(bh_ctx* ctx, int rank, int world, const void* unique_id) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (world < 1 || rank < 0 || rank >= world || !unique_id) return set_error(ctx, BH_ERR_INVALID_ARG, "comm_init: need 0 <= rank < world and a unique id");
    if (ctx->comm) return set_error(ctx, BH_ERR_STATE, "comm_init: this context already has a communicator");
    RcclApi& a = rccl();
    if (!a.lib || !a.error.empty()) return set_error(ctx, BH_ERR_UNSUPPORTED, a.error.empty() ? "RCCL unavailable" : a.error);
    BH_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    BH_TRY(rccl_check(ctx, a.CommInitRank(&comm, world, id, rank), "ncclCommInitRank"));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    // side stream for the flag exchange (api.hip); without it the exchange simply stays on the ctx stream
    if (hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->comm_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
        ctx->comm_stream = nullptr;
        ctx->comm_ev = nullptr;
    }
    return 0;
}

int bh_comm_destroy(bh_ctx* ctx) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!ctx->comm) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) { (void)hipStreamSynchronize(ctx->comm_stream); (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
    if (ctx->comm_ev) { (void)hipEventDestroy(ctx->comm_ev); ctx->comm_ev = nullptr; }
    const ncclResult_t rc = rccl().CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return rccl_check(ctx, rc, "ncclCommDestroy");
}

int bh_comm_world(bh_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_world : 1; }

int bh_allreduce_sum_f32(bh_ctx* ctx, float* buf, uint64_t count) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (count > 0 && !buf) return set_error(ctx, BH_ERR_INVALID_ARG, "allreduce: null buffer");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return comm_allreduce(ctx, buf, count, false);
}

int bh_allreduce_max_f32(bh_ctx* ctx, float* buf, uint64_t count) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (count > 0 && !buf) return set_error(ctx, BH_ERR_INVALID_ARG, "allreduce: null buffer");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return comm_allreduce(ctx, buf, count, true);
}

int bh_allgather_bytes(bh_ctx* ctx, const void* send, void* recv, uint64_t bytes_per_rank) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!ctx->comm) return set_error(ctx, BH_ERR_STATE, "no communicator: call bh_comm_init first");
    if (bytes_per_rank == 0) return 0;
    if (!send || !recv) return set_error(ctx, BH_ERR_INVALID_ARG, "allgather: null buffer");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return rccl_check(ctx, rccl().AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream), "ncclAllGather");
}

int bh_comm_rank(bh_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_rank : 0; }

int bh_strip_halo_plan(uint32_t img_h, uint32_t row_begin_px, uint32_t row_end_px, int rank, int world, BhHaloOp* out /*[4]*/) {
    if (!out || world < 1 || rank < 0 || rank >= world) return BH_ERR_INVALID_ARG;
    const int k = strip_halo_plan(img_h, row_begin_px, row_end_px, rank, world, 21u, out);
    return k < 0 ? BH_ERR_INVALID_ARG : k;
}

int bh_exchange_strip_halos(bh_ctx* ctx, float* img_hwc4, uint32_t h, uint32_t w, uint32_t row_begin_px, uint32_t row_end_px) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!img_hwc4 || h == 0 || w == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "exchange_strip_halos: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return comm_exchange_strip_halos(ctx, img_hwc4, h, w, row_begin_px, row_end_px);
}

// Every collective this file binds, run once on small rank-dependent patterns and checked on the host: out-of-place all-reduce
// SUM and MAX, all-gather, and a grouped ring shift with send / recv.  What the first multi-GPU run of a build should call before
// it trusts a gradient exchange; on one rank it still drives every entry point through RCCL with non-trivial data.
int bh_comm_selftest(bh_ctx* ctx) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!ctx->comm) return set_error(ctx, BH_ERR_STATE, "no communicator: call bh_comm_init first");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    RcclApi& a = rccl();
    const int W = ctx->comm_world, R = ctx->comm_rank;
    constexpr size_t N = 1027;   // (odd on purpose)
    std::vector<float> host(N);
    for (size_t i = 0; i < N; ++i) host[i] = (float)(R + 1) * (float)((i % 97) + 1) - ((i & 1) ? 3.0f * (float)R : 0.0f);
    float* d = nullptr;   // [send N | sum N | max N | gather W*N | shifted N]
    const size_t total = N * (4 + (size_t)W);
    BH_HIP(ctx, hipMalloc((void**)&d, total * 4));
    int rc = check_hip(ctx, hipMemsetAsync(d, 0xFF, total * 4, ctx->stream), "selftest memset");
    if (rc == 0) rc = check_hip(ctx, hipMemcpyAsync(d, host.data(), N * 4, hipMemcpyHostToDevice, ctx->stream), "selftest upload");
    float *sum = d + N, *mx = d + 2 * N, *gat = d + 3 * N, *shf = d + (3 + (size_t)W) * N;
    if (rc == 0) rc = rccl_check(ctx, a.AllReduce(d, sum, N, ncclFloat32, ncclSum, (ncclComm_t)ctx->comm, ctx->stream), "ncclAllReduce(sum)");
    if (rc == 0) rc = rccl_check(ctx, a.AllReduce(d, mx, N, ncclFloat32, ncclMax, (ncclComm_t)ctx->comm, ctx->stream), "ncclAllReduce(max)");
    if (rc == 0) rc = rccl_check(ctx, a.AllGather(d, gat, N * 4, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream), "ncclAllGather");
    if (rc == 0) {
        rc = rccl_check(ctx, a.GroupStart(), "ncclGroupStart");
        if (rc == 0) rc = rccl_check(ctx, a.Send(d, N, ncclFloat32, (R + 1) % W, (ncclComm_t)ctx->comm, ctx->stream), "ncclSend");
        if (rc == 0) rc = rccl_check(ctx, a.Recv(shf, N, ncclFloat32, (R + W - 1) % W, (ncclComm_t)ctx->comm, ctx->stream), "ncclRecv");
        const int rc2 = rccl_check(ctx, a.GroupEnd(), "ncclGroupEnd");
        if (rc == 0) rc = rc2;
    }
    // the direct all-reduce (reduce-scatter + all-gather over grouped send / recv) against ncclAllReduce's sum, on a message long
    // enough to be cut into chunks with a ragged tail
    constexpr size_t ND = 70003;
    float* dd = nullptr;
    std::vector<float> dgot;
    if (rc == 0 && W > 1 && ctx->knob_direct_allreduce) {   // (only on a ctx that is going to use it: option grad_allreduce = direct)
        std::vector<float> dh(ND);
        for (size_t i = 0; i < ND; ++i) dh[i] = (float)(R + 1) * (float)((i % 89) + 1) - (float)((i * 7 + (size_t)R) % 13);
        rc = check_hip(ctx, hipMalloc((void**)&dd, ND * 4), "selftest hipMalloc");
        if (rc == 0) rc = check_hip(ctx, hipMemcpyAsync(dd, dh.data(), ND * 4, hipMemcpyHostToDevice, ctx->stream), "selftest upload");
        if (rc == 0) rc = comm_allreduce_direct(ctx, dd, ND);
        dgot.resize(ND);
        if (rc == 0) rc = check_hip(ctx, hipMemcpyAsync(dgot.data(), dd, ND * 4, hipMemcpyDeviceToHost, ctx->stream), "selftest download");
    }
    std::vector<float> got(total);
    if (rc == 0) rc = check_hip(ctx, hipMemcpyAsync(got.data(), d, total * 4, hipMemcpyDeviceToHost, ctx->stream), "selftest download");
    if (rc == 0) rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "selftest sync");
    (void)hipFree(d);
    if (dd) (void)hipFree(dd);
    if (rc != 0) return rc;
    for (size_t i = 0; i < dgot.size(); ++i) {
        double sref = 0.0;
        for (int r = 0; r < W; ++r) sref += (double)((float)(r + 1) * (float)((i % 89) + 1) - (float)((i * 7 + (size_t)r) % 13));
        if (std::fabs((double)dgot[i] - sref) > 1e-3 * (1.0 + std::fabs(sref))) {
            char msg[200];
            snprintf(msg, sizeof msg, "comm_selftest: direct all-reduce (reduce-scatter + all-gather) is wrong at element %zu on rank %d of %d", i, R, W);
            return set_error(ctx, BH_ERR_HIP, msg);
        }
    }
    auto pattern = [&](int r, size_t i) { return (float)(r + 1) * (float)((i % 97) + 1) - ((i & 1) ? 3.0f * (float)r : 0.0f); };
    for (size_t i = 0; i < N; ++i) {
        double s = 0.0;
        float m = pattern(0, i);
        for (int r = 0; r < W; ++r) { s += pattern(r, i); m = pattern(r, i) > m ? pattern(r, i) : m; }
        const char* what = nullptr;
        if (std::fabs((double)got[N + i] - s) > 1e-3 * (1.0 + std::fabs(s))) what = "all-reduce SUM";
        else if (got[2 * N + i] != m) what = "all-reduce MAX";
        else if (got[(3 + (size_t)W) * N + i] != pattern((R + W - 1) % W, i)) what = "send / recv ring shift";
        for (int r = 0; r < W && !what; ++r)
            if (got[(3 + (size_t)r) * N + i] != pattern(r, i)) what = "all-gather";
        if (what) {
            char msg[200];
            snprintf(msg, sizeof msg, "comm_selftest: %s is wrong at element %zu on rank %d of %d", what, i, R, W);
            return set_error(ctx, BH_ERR_HIP, msg);
        }
    }
    return 0;
}

}  // extern "C"
