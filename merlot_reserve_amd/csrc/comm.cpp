// RCCL entry points of the C-ABI (include/mreserve_hip.h: mr_comm_*): the data-parallel collectives of the pretraining
// step -- jax.lax.all_gather of the contrastive targets (pretrain/pretrain_model.py:290), its transpose (a reduce-scatter
// in backward), pmean of the bf16 gradients (:329) and of the fp32 metrics (:336) -- on a library-owned communicator, one
// per process / GPU, over xGMI.  Every call is asynchronous on the caller's hipStream_t and capturable into a hipGraph
// (RCCL >= 2.9 records its kernels into the capturing stream), which is what lets the world > 1 step be ONE graph.
//
// RCCL is resolved at run time from the librccl.so.1 ALREADY in the process (torch brings one) or, failing that, from the
// loader path: no link-time dependency, and a second copy of RCCL can never be pulled in beside torch's.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>
#include "../../include/mreserve_hip.h"

void mr_set_error(const char* fmt, ...);

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi& api() {
    static RcclApi a;
    static bool tried = false;
    if (tried) return a;
    tried = true;
    a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);         // the copy torch (or the host program) already loaded
    if (!a.handle) a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle) a.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle) return a;
#define MR_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name))
    MR_SYM(GetUniqueId, "ncclGetUniqueId");
    MR_SYM(CommInitRank, "ncclCommInitRank");
    MR_SYM(CommDestroy, "ncclCommDestroy");
    MR_SYM(AllReduce, "ncclAllReduce");
    MR_SYM(AllGather, "ncclAllGather");
    MR_SYM(ReduceScatter, "ncclReduceScatter");
    MR_SYM(GetErrorString, "ncclGetErrorString");
#undef MR_SYM
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.ReduceScatter && a.GetErrorString;
    return a;
}

}  // namespace

struct mr_comm {
    ncclComm_t comm;
    int rank, world, device;
};

#define MR_RCCL_READY(what)                                                                            \
    RcclApi& A = api();                                                                                \
    if (!A.ok) {                                                                                       \
        mr_set_error("%s: librccl.so.1 not found / incomplete (%s)", what, A.handle ? "missing symbols" : dlerror()); \
        return MR_ELAUNCH;                                                                             \
    }
#define MR_RCCL_CALL(what, expr)                                                  \
    do {                                                                          \
        ncclResult_t r__ = (expr);                                                \
        if (r__ != ncclSuccess) {                                                 \
            mr_set_error("%s: RCCL error %d: %s", what, (int)r__, A.GetErrorString(r__)); \
            return MR_ELAUNCH;                                                    \
        }                                                                         \
    } while (0)
#define MR_ARG(cond, msg)          \
    do {                           \
        if (!(cond)) {             \
            mr_set_error(msg);     \
            return MR_EINVAL;      \
        }                          \
    } while (0)

extern "C" int mr_comm_unique_id(void* id_out) {
    MR_ARG(id_out, "mr_comm_unique_id: null pointer");
    MR_RCCL_READY("mr_comm_unique_id");
    ncclUniqueId id;
    MR_RCCL_CALL("mr_comm_unique_id", A.GetUniqueId(&id));
    static_assert(sizeof(id) == MR_COMM_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return MR_OK;
}

extern "C" int mr_comm_init(int32_t rank, int32_t world, const void* unique_id, mr_comm** out) {
    MR_ARG(out && unique_id && world >= 1 && rank >= 0 && rank < world, "mr_comm_init: bad rank / world / null pointer");
    MR_RCCL_READY("mr_comm_init");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    mr_comm* c = new mr_comm{nullptr, rank, world, 0};
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        mr_set_error("mr_comm_init: no current HIP device");
        return MR_ELAUNCH;
    }
    ncclResult_t r = A.CommInitRank(&c->comm, world, id, rank);      // collective: every rank calls it, on its own device
    if (r != ncclSuccess) {
        mr_set_error("mr_comm_init: ncclCommInitRank failed (%d): %s", (int)r, A.GetErrorString(r));
        delete c;
        return MR_ELAUNCH;
    }
    *out = c;
    return MR_OK;
}

extern "C" int mr_comm_destroy(mr_comm* c) {
    if (!c) return MR_OK;
    MR_RCCL_READY("mr_comm_destroy");
    A.CommDestroy(c->comm);
    delete c;
    return MR_OK;
}

extern "C" int32_t mr_comm_rank(const mr_comm* c) { return c ? c->rank : -1; }
extern "C" int32_t mr_comm_world(const mr_comm* c) { return c ? c->world : -1; }

extern "C" int mr_allreduce_mean_bf16(mr_comm* c, void* buf, int64_t n, void* stream) {
    MR_ARG(c && buf && n > 0, "mr_allreduce_mean_bf16: bad args");
    MR_RCCL_READY("mr_allreduce_mean_bf16");
    MR_RCCL_CALL("mr_allreduce_mean_bf16", A.AllReduce(buf, buf, (size_t)n, ncclBfloat16, ncclAvg, c->comm, static_cast<hipStream_t>(stream)));
    return MR_OK;
}

extern "C" int mr_allreduce_mean_f32(mr_comm* c, float* buf, int64_t n, void* stream) {
    MR_ARG(c && buf && n > 0, "mr_allreduce_mean_f32: bad args");
    MR_RCCL_READY("mr_allreduce_mean_f32");
    MR_RCCL_CALL("mr_allreduce_mean_f32", A.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, c->comm, static_cast<hipStream_t>(stream)));
    return MR_OK;
}

extern "C" int mr_allgather(mr_comm* c, const void* send, void* recv, int64_t n_per_rank, void* stream) {
    MR_ARG(c && send && recv && n_per_rank > 0, "mr_allgather: bad args");
    MR_RCCL_READY("mr_allgather");
    MR_RCCL_CALL("mr_allgather", A.AllGather(send, recv, (size_t)n_per_rank, ncclBfloat16, c->comm, static_cast<hipStream_t>(stream)));
    return MR_OK;
}

extern "C" int mr_reducescatter_sum(mr_comm* c, const void* send, void* recv, int64_t n_per_rank, void* stream) {
    MR_ARG(c && send && recv && n_per_rank > 0, "mr_reducescatter_sum: bad args");
    MR_RCCL_READY("mr_reducescatter_sum");
    MR_RCCL_CALL("mr_reducescatter_sum", A.ReduceScatter(send, recv, (size_t)n_per_rank, ncclBfloat16, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
    return MR_OK;
}

// fp32 form (the fp32 training step's dL/dE of the gathered contrastive embeddings); its all-gather is mr_allgather on twice the element count
extern "C" int mr_reducescatter_sum_f32(mr_comm* c, const float* send, float* recv, int64_t n_per_rank, void* stream) {
    MR_ARG(c && send && recv && n_per_rank > 0, "mr_reducescatter_sum_f32: bad args");
    MR_RCCL_READY("mr_reducescatter_sum_f32");
    MR_RCCL_CALL("mr_reducescatter_sum_f32", A.ReduceScatter(send, recv, (size_t)n_per_rank, ncclFloat32, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
    return MR_OK;
}
