// C1 / C2: the two collectives of the sharded path on a CALLER-OWNED RCCL communicator.
//
//   tt_allgather_topk    every rank's [vals | idx] block of per-shard top-k' lists -> all ranks, one ncclAllGather
//                        (the exchange step after backend/evaluators.py:185-186 once the corpus is row-sharded);
//                        the receive buffer is what tt_topk_merge_shards reads in place
//   tt_allreduce_grads   one summing ncclAllReduce of the flat fp32 gradient buffer, in place (between
//                        loss.backward() and clip_grad_norm_, backend/main.py:254 -> :257); the 1/world scale
//                        is applied by tt_clip_adam_step_f32 (grad_scale)
//
// The communicator is created by the host and passed as an opaque ncclComm_t, so libtt.so must call into the SAME
// RCCL instance the host used (a PyTorch process carries its own librccl.so next to libtorch, a C host links
// /opt/rocm/lib/librccl.so).  libtt.so therefore does not link RCCL: on first use it looks for an RCCL already
// mapped into the process (dl_iterate_phdr) and binds to that, and only otherwise opens librccl.so.1 itself.
// tt_comm_* are thin helpers for hosts without an RCCL binding of their own.
#include "tt_common.h"

#include <dlfcn.h>
#include <link.h>
#include <string.h>

#include <rccl/rccl.h>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char path[512] = "";
    bool ok = false;
};

// The RCCL library itself: a mapped object whose BASENAME is librccl.so or librccl.so.<version> -- not a plugin that merely
// carries the name (librccl-net.so, librccl_tuner.so ...: binding one of those leaves every symbol below unresolved).
int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data)
{
    const char *name = info->dlpi_name;
    if (!name)
        return 0;
    const char *base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "librccl.so", 10) == 0 && (base[10] == '\0' || base[10] == '.')) {
        strncpy((char *)data, name, 511);
        return 1;
    }
    return 0;
}

bool bind_rccl(Rccl &x)
{
#define TT_SYM(field, name) *(void **)(&x.field) = dlsym(x.handle, name)
    TT_SYM(GetUniqueId, "ncclGetUniqueId");
    TT_SYM(CommInitRank, "ncclCommInitRank");
    TT_SYM(CommDestroy, "ncclCommDestroy");
    TT_SYM(CommCount, "ncclCommCount");
    TT_SYM(CommUserRank, "ncclCommUserRank");
    TT_SYM(AllGather, "ncclAllGather");
    TT_SYM(AllReduce, "ncclAllReduce");
    TT_SYM(GetErrorString, "ncclGetErrorString");
#undef TT_SYM
    return x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.CommCount && x.CommUserRank && x.AllGather && x.AllReduce &&
           x.GetErrorString;
}

// Resolved once per process (function-local static: thread-safe initialisation, immutable afterwards).
const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl x;
        char loaded[512] = "";
        dl_iterate_phdr(find_loaded_rccl, loaded);
        if (loaded[0])
            x.handle = dlopen(loaded, RTLD_NOW | RTLD_NOLOAD);
        if (x.handle && loaded[0]) {
            strncpy(x.path, loaded, sizeof(x.path) - 1);
            x.ok = bind_rccl(x);
        }
        if (!x.ok) { // nothing mapped (a C host that has not touched RCCL yet), or the mapped object lacks the symbols
            for (const char *cand : {"librccl.so.1", "librccl.so"}) {
                void *h = dlopen(cand, RTLD_NOW);
                if (!h)
                    continue;
                x.handle = h;
                snprintf(x.path, sizeof(x.path), "%s (opened by libtt.so)", cand);
                x.ok = bind_rccl(x);
                if (x.ok)
                    break;
            }
        }
        return x;
    }();
    return r;
}

int need_rccl(const char *who)
{
    if (!rccl().ok)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: no usable RCCL (librccl.so.1) in this process: %s", who,
                       rccl().handle ? "symbols missing" : dlerror());
    return TT_OK;
}

#define TT_RCCL_CHECK(expr)                                                                                      \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess)                                                                                   \
            return tt_fail(TT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

} // namespace

TT_EXPORT const char *tt_comm_library(void)
{
    return rccl().ok ? rccl().path : "";
}

TT_EXPORT int tt_comm_unique_id(void *id_bytes)
{
    int rc = need_rccl("tt_comm_unique_id");
    if (rc != TT_OK)
        return rc;
    if (!id_bytes)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_comm_unique_id: null pointer");
    ncclUniqueId id;
    TT_RCCL_CHECK(rccl().GetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof(id));
    return TT_OK;
}

TT_EXPORT int tt_comm_init_rank(void **comm, int world, const void *id_bytes, int rank)
{
    int rc = need_rccl("tt_comm_init_rank");
    if (rc != TT_OK)
        return rc;
    if (!comm || !id_bytes || world < 1 || rank < 0 || rank >= world)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_comm_init_rank: world=%d rank=%d", world, rank);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    TT_RCCL_CHECK(rccl().CommInitRank(&c, world, id, rank));
    *comm = (void *)c;
    return TT_OK;
}

TT_EXPORT int tt_comm_destroy(void *comm)
{
    int rc = need_rccl("tt_comm_destroy");
    if (rc != TT_OK)
        return rc;
    if (comm)
        TT_RCCL_CHECK(rccl().CommDestroy((ncclComm_t)comm));
    return TT_OK;
}

TT_EXPORT int tt_comm_info(void *comm, int *world, int *rank)
{
    int rc = need_rccl("tt_comm_info");
    if (rc != TT_OK)
        return rc;
    if (!comm || !world || !rank)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_comm_info: null pointer");
    TT_RCCL_CHECK(rccl().CommCount((ncclComm_t)comm, world));
    TT_RCCL_CHECK(rccl().CommUserRank((ncclComm_t)comm, rank));
    return TT_OK;
}

TT_EXPORT int tt_allgather_topk(void *comm, const void *send_block, void *recv_blocks, size_t block_bytes,
                                tt_stream_t stream)
{
    int rc = need_rccl("tt_allgather_topk");
    if (rc != TT_OK)
        return rc;
    if (!comm || !send_block || !recv_blocks || block_bytes == 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_allgather_topk: null pointer or empty block");
    TT_RCCL_CHECK(rccl().AllGather(send_block, recv_blocks, block_bytes, ncclUint8, (ncclComm_t)comm, (hipStream_t)stream));
    return TT_OK;
}

TT_EXPORT int tt_allreduce_grads(void *comm, float *flat_grads, int64_t n, tt_stream_t stream)
{
    int rc = need_rccl("tt_allreduce_grads");
    if (rc != TT_OK)
        return rc;
    if (!comm || !flat_grads || n <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_allreduce_grads: null pointer or n=%lld", (long long)n);
    TT_RCCL_CHECK(rccl().AllReduce(flat_grads, flat_grads, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm,
                                   (hipStream_t)stream));
    return TT_OK;
}
