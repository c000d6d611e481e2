// comm.hip — the one collective of the hot path: an RCCL all-gather of packed right-operand rows over xGMI
// (SURVEY.md §8e).  The reference has no collective of any kind (symmer/process_handler.py is a CPU fork pool
// that never touches this path); this is new, MI355X-native plumbing: one process per GPU, the left-term axis is
// sharded across ranks, every rank contributes 1/G of the right operand and receives all of it.
//
// Round 5: the same collective for ONE process that drives several devices (symgpu_init_all): ncclCommInitAll creates one communicator
// per device, and the per-device all-gathers of a step are issued from one thread inside ncclGroupStart / ncclGroupEnd, each on its
// device's stream (symgpu_comm_init_all / symgpu_comm_allgather_ops).
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
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    ncclComm_t all[SYMGPU_MAX_DEVICES] = {};   // single-process mode: one communicator per device 0 .. n_all-1
    int n_all = 0;
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
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
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
    for (int d = 0; d < g_rccl.n_all; ++d) {
        Context *c = ctx_of_device(d);
        if (c && c->ready) (void)hipStreamSynchronize(c->stream);
        if (g_rccl.all[d]) g_rccl.CommDestroy(g_rccl.all[d]);
        g_rccl.all[d] = nullptr;
    }
    g_rccl.n_all = 0;
    forget_bound_device();
    return SYMGPU_OK;
}

// ---- one process, several devices -------------------------------------------------------------------------------------------------
int symgpu_comm_init_all(int n_devices) {
    SG_REQUIRE(n_devices >= 1 && n_devices <= SYMGPU_MAX_DEVICES, "comm_init_all: device count");
    SG_TRY(load_rccl());
    if (g_rccl.n_all) { set_error("comm_init_all: communicators already exist"); return SYMGPU_E_INVALID; }
    int devs[SYMGPU_MAX_DEVICES];
    for (int d = 0; d < n_devices; ++d) {
        Context *c = ctx_of_device(d);
        if (!c || !c->ready) { set_error("comm_init_all: device %d has no context (symgpu_init_all first)", d); return SYMGPU_E_NODEVICE; }
        devs[d] = d;
    }
    ncclResult_t r = g_rccl.CommInitAll(g_rccl.all, n_devices, devs);
    forget_bound_device();                                           // RCCL walks the devices with hipSetDevice
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitAll");
    g_rccl.n_all = n_devices;
    return SYMGPU_OK;
}

// shards[d] / fulls[d] live on device d (d = 0 .. n-1, n = the communicator count); every shard has the SAME capacity Ts; device d's rows land
// at [d * Ts, (d + 1) * Ts) of every full operator.  All n all-gathers are enqueued from this thread as one RCCL group, each on its device's
// stream; nothing is waited for (the next call on a device is ordered behind its part by the stream).
int symgpu_comm_allgather_ops(const symgpu_op_t *shards, const symgpu_op_t *fulls, int n) {
    SG_REQUIRE(shards && fulls && n >= 1, "comm_allgather_ops: arguments");
    if (g_rccl.n_all != n) { set_error("comm_allgather_ops: %d operators for %d communicators (symgpu_comm_init_all)", n, g_rccl.n_all); return SYMGPU_E_RCCL; }
    const i64 Ts = shards[0] ? shards[0]->capacity : 0;
    for (int d = 0; d < n; ++d) {
        SG_REQUIRE(shards[d] && fulls[d] && shards[d]->Wq == fulls[d]->Wq && shards[d]->Wq == shards[0]->Wq, "comm_allgather_ops: handles");
        SG_REQUIRE(shards[d]->device == d && fulls[d]->device == d, "comm_allgather_ops: operator d must live on device d");
        SG_REQUIRE(shards[d]->capacity == Ts && fulls[d]->capacity >= Ts * n, "comm_allgather_ops: shard capacities must agree, full capacity >= n * Ts");
        SG_REQUIRE((shards[d]->coeff != nullptr) == (shards[0]->coeff != nullptr) && (fulls[d]->coeff != nullptr) == (fulls[0]->coeff != nullptr),
                   "comm_allgather_ops: coefficients on all operators or on none");
    }
    const int W = 2 * shards[0]->Wq;
    const bool with_coeff = shards[0]->coeff && fulls[0]->coeff;
    const int saved = selected_device();
    int rc = SYMGPU_OK;
    for (int d = 0; d < n && rc == SYMGPU_OK; ++d) {                   // padding rows of short shards are identities
        select_device(d);
        rc = require_ctx();
        if (rc != SYMGPU_OK) break;
        symgpu_op_s *sh = shards[d];
        if (sh->T < Ts) {
            if (hipMemsetAsync(sh->rows + sh->T * W, 0, (size_t)(Ts - sh->T) * W * 8, ctx().stream) != hipSuccess) rc = SYMGPU_E_HIP;
            if (sh->coeff && hipMemsetAsync(sh->coeff + 2 * sh->T, 0, (size_t)(Ts - sh->T) * 16, ctx().stream) != hipSuccess) rc = SYMGPU_E_HIP;
        }
    }
    if (rc == SYMGPU_OK) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int d = 0; d < n && r == ncclSuccess; ++d) {
            select_device(d);
            if (require_ctx() != SYMGPU_OK) { r = ncclInternalError; break; }
            r = g_rccl.AllGather(shards[d]->rows, fulls[d]->rows, (size_t)Ts * W, ncclUint64, g_rccl.all[d], ctx().stream);
            if (r == ncclSuccess && with_coeff) r = g_rccl.AllGather(shards[d]->coeff, fulls[d]->coeff, (size_t)Ts * 2, ncclFloat64, g_rccl.all[d], ctx().stream);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) rc = rccl_fail(r, "grouped ncclAllGather");
    }
    for (int d = 0; d < n; ++d) { op_invalidate(fulls[d]); if (rc == SYMGPU_OK) fulls[d]->T = Ts * n; }
    select_device(saved);
    forget_bound_device();
    return rc;
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
