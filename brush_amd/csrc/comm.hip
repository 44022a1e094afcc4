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

#include <cstring>

#include "context.h"

namespace bh {

// the slice of rccl.h this file needs (ABI of RCCL 2.x / NCCL 2.x)
typedef struct { char internal[128]; } RcclUniqueId;
typedef void* RcclComm;
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_FLOAT32 = 7, RCCL_UINT8 = 1 };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

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
    api.GetUniqueId = (int (*)(RcclUniqueId*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(RcclComm))sym("ncclCommDestroy");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))sym("ncclAllReduce");
    api.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))sym("ncclAllGather");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    return api;
}

static int rccl_check(bh_ctx* ctx, int rc, const char* what) {
    if (rc == 0) return 0;
    RcclApi& a = rccl();
    return set_error(ctx, BH_ERR_HIP, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "RCCL error"));
}

#ifdef BH_TEST_HOOKS
// BH_BREAK_ALLREDUCE (test-hook build only, read once at bh_create): corrupt every all-reduce's first element — exists so that
// bench.py's pre-timing self-check of the exchange path can be shown to catch a broken collective (tests/test_gpu_bench_selfcheck.py)
__global__ void break_allreduce_kernel(float* buf) { buf[0] += 1.0f; }
#endif

int comm_allreduce(bh_ctx* ctx, float* buf, uint64_t count, bool max_op) {
    if (!ctx->comm) return set_error(ctx, BH_ERR_STATE, "no communicator: call bh_comm_init first");
    if (count == 0) return 0;
    int rc = 0;
    if (ctx->comm_world > 1)   // (one rank: nothing to exchange)
        rc = rccl_check(ctx, rccl().AllReduce(buf, buf, (size_t)count, RCCL_FLOAT32, max_op ? RCCL_MAX : RCCL_SUM, (RcclComm)ctx->comm, ctx->stream),
                        "ncclAllReduce");
#ifdef BH_TEST_HOOKS
    if (rc == 0 && ctx->knob_break_allreduce) hipLaunchKernelGGL(break_allreduce_kernel, dim3(1), dim3(1), 0, ctx->stream, buf);
#endif
    return rc;
}

}  // namespace bh

using namespace bh;

extern "C" {

int bh_comm_unique_id(void* out_id) {
    if (!out_id) return BH_ERR_INVALID_ARG;
    RcclApi& a = rccl();
    if (!a.lib || !a.error.empty()) return BH_ERR_UNSUPPORTED;
    RcclUniqueId id;
    if (a.GetUniqueId(&id) != 0) return BH_ERR_HIP;
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
    RcclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    RcclComm comm = nullptr;
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
    const int rc = rccl().CommDestroy((RcclComm)ctx->comm);
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
    return rccl_check(ctx, rccl().AllGather(send, recv, (size_t)bytes_per_rank, RCCL_UINT8, (RcclComm)ctx->comm, ctx->stream), "ncclAllGather");
}

}  // extern "C"
