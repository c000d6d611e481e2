// comm.hip — the one collective of the hot path: an RCCL all-gather of packed right-operand rows over xGMI
// (SURVEY.md §8e).  The reference has no collective of any kind (symmer/process_handler.py is a CPU fork pool
// that never touches this path); this is new, MI355X-native plumbing: one process per GPU, the left-term axis is
// sharded across ranks, every rank contributes 1/G of the right operand and receives all of it.
//
// librccl is resolved with dlopen on first use so that single-GPU users never load it.
#include "common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <mutex>

namespace symgpu {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    // a bring-up that did not return in time was given up by the caller (symgpu_comm_abandon): if ncclCommInitRank does come back
    // later, its communicator is destroyed instead of installed — the ranks have agreed on another data plane by then
    int init_generation = 0, abandoned_generation = -1;
    std::mutex mu;
};
static Rccl g_rccl;

static int load_rccl() {
    if (g_rccl.handle) return SYMGPU_OK;
    if (getenv("SYMGPU_RCCL_DISABLE")) {                           // test knob: exercise the callers' no-RCCL fallback
        set_error("RCCL disabled by SYMGPU_RCCL_DISABLE");
        return SYMGPU_E_RCCL;
    }
    // absolute paths first: a dlopen by SONAME would hand back a *different* librccl that is already in the process
    // (PyTorch wheels bundle one, built against their own HIP runtime)
    std::string env_path = getenv("ROCM_PATH") ? std::string(getenv("ROCM_PATH")) + "/lib/librccl.so.1" : std::string("/opt/rocm/lib/librccl.so.1");
    const char *names[] = {env_path.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        // DEEPBIND: librccl must resolve HIP against ITS OWN dependency (the /opt/rocm libamdhip64 this library is linked
        // to), not against another HIP runtime that may sit in the global scope (PyTorch wheels bundle one).
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (h) break;
    }
    if (!h) {
        set_error("cannot load librccl: %s", dlerror());
        return SYMGPU_E_RCCL;
    }
#define SYM(field, name)                                                   \
    *(void **)(&g_rccl.field) = dlsym(h, name);                            \
    if (!g_rccl.field) { set_error("librccl lacks symbol %s", name); dlclose(h); return SYMGPU_E_RCCL; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.handle = h;
    return SYMGPU_OK;
}

static int rccl_fail(ncclResult_t r, const char *what) {
    set_error("RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
    return SYMGPU_E_RCCL;
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_comm_available(void) {
    // no device needed: only resolves librccl and its entry points, so that every rank can report BEFORE any of them enters the
    // collective ncclCommInitRank (a rank that cannot load the library would leave its peers blocked in there)
    return load_rccl();
}

int symgpu_comm_unique_id(uint8_t id[SYMGPU_UNIQUE_ID_BYTES]) {
    SG_REQUIRE(id, "comm_unique_id");
    SG_TRY(load_rccl());
    static_assert(sizeof(ncclUniqueId) == SYMGPU_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    memcpy(id, &u, SYMGPU_UNIQUE_ID_BYTES);
    return SYMGPU_OK;
}

int symgpu_comm_init(const uint8_t id[SYMGPU_UNIQUE_ID_BYTES], int rank, int nranks) {
    SG_TRY(require_ctx());
    SG_REQUIRE(id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init");
    SG_TRY(load_rccl());
    if (g_rccl.comm) { set_error("comm_init: communicator already exists"); return SYMGPU_E_INVALID; }
    ncclUniqueId u;
    memcpy(&u, id, SYMGPU_UNIQUE_ID_BYTES);
    int my_generation;
    { std::lock_guard<std::mutex> lk(g_rccl.mu); my_generation = ++g_rccl.init_generation; }
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, u, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    {
        std::lock_guard<std::mutex> lk(g_rccl.mu);
        if (g_rccl.abandoned_generation < my_generation) {
            g_rccl.comm = comm;
            g_rccl.rank = rank;
            g_rccl.nranks = nranks;
            return SYMGPU_OK;
        }
    }
    g_rccl.CommDestroy(comm);                                       // came back after the caller had given up on it
    set_error("comm_init: ncclCommInitRank returned after symgpu_comm_abandon; communicator destroyed");
    return SYMGPU_E_RCCL;
}

int symgpu_comm_abandon(void) {
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    g_rccl.abandoned_generation = g_rccl.init_generation;
    return SYMGPU_OK;
}

int symgpu_comm_destroy(void) {
    if (g_rccl.comm) {
        if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
        g_rccl.CommDestroy(g_rccl.comm);
        g_rccl.comm = nullptr;
    }
    return SYMGPU_OK;
}

// Every rank passes a shard with the SAME capacity Ts = ceil(M_total / nranks) (valid rows: its own T, the tail
// ranks may hold fewer); rank r's rows land at [r*Ts, ...) of `full`, i.e. at their global row index.
// full->T is set to nranks * Ts clipped by full capacity semantics: the caller passes M_total through full->capacity
// >= nranks*Ts and then trims with symgpu_op_info / its own bookkeeping (rows >= M_total are zero padding).
int symgpu_comm_allgather_op(symgpu_op_t shard, symgpu_op_t full) {
    SG_TRY(require_ctx());
    SG_REQUIRE(shard && full && shard->Wq == full->Wq, "comm_allgather_op: handles");
    if (!g_rccl.comm) { set_error("comm_allgather_op: symgpu_comm_init has not been called"); return SYMGPU_E_RCCL; }
    const i64 Ts = shard->capacity;
    const int W = 2 * shard->Wq;
    SG_REQUIRE(full->capacity >= Ts * g_rccl.nranks, "comm_allgather_op: full capacity < nranks * shard capacity");
    hipStream_t st = ctx().stream;
    // zero the unused tail of this rank's shard so that padding rows are identities
    if (shard->T < Ts) {
        HIP_TRY(hipMemsetAsync(shard->rows + shard->T * W, 0, (size_t)(Ts - shard->T) * W * 8, st));
        if (shard->coeff) HIP_TRY(hipMemsetAsync(shard->coeff + 2 * shard->T, 0, (size_t)(Ts - shard->T) * 16, st));
    }
    ncclResult_t r = g_rccl.AllGather(shard->rows, full->rows, (size_t)Ts * W, ncclUint64, g_rccl.comm, st);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather(rows)");
    if (shard->coeff && full->coeff) {
        r = g_rccl.AllGather(shard->coeff, full->coeff, (size_t)Ts * 2, ncclFloat64, g_rccl.comm, st);
        if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather(coeff)");
    }
    op_invalidate(full);
    full->T = Ts * g_rccl.nranks;
    return SYMGPU_OK;
}

int symgpu_comm_barrier(void) {
    SG_TRY(require_ctx());
    if (!g_rccl.comm) return SYMGPU_OK;
    Scratch one;
    SG_TRY(one.alloc(16));
    HIP_TRY(hipMemsetAsync(one.p, 0, 16, ctx().stream));
    ncclResult_t r = g_rccl.AllReduce(one.p, one.p, 1, ncclUint64, ncclSum, g_rccl.comm, ctx().stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce(barrier)");
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

}  // extern "C"
