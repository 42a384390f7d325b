// smc_comm.hip -- the only collective of the path: gathering the per-island
// log-evidence estimates of independent runs / SMC^2 islands that were sharded
// across the GPUs of a node (particles/core.py:431 multiSMC returns the list of
// per-run outputs; utils.py:178-186 collects them from the loky workers).
//
// One process per GPU; the communicator is RCCL over xGMI.  The payload is
// n_islands x 8 bytes per rank, i.e. latency-bound: one ncclAllGather per run.
// RCCL is dlopen'ed on first use so that single-GPU users never load it.
#include <dlfcn.h>

#include "smc_device.h"
#include "smc_internal.h"

// The binding below is compiled into the emulator build as well: with SMC_RCCL_LIBRARY naming a
// library that exports the nccl* entry points (tests/emu/fake_rccl.c: a multi-process test double on
// host memory) the CPU suite drives exactly the calls -- argument order, counts, datatypes, group
// nesting -- that run over xGMI on a node; without it the emulator has its one-rank stub.
namespace {
typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef ncclResult_t (*fn_GetUniqueId)(ncclUniqueId*);
typedef ncclResult_t (*fn_CommInitRank)(void**, int, ncclUniqueId, int);
typedef ncclResult_t (*fn_AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef ncclResult_t (*fn_CommDestroy)(void*);
typedef ncclResult_t (*fn_Send)(const void*, size_t, int, int, void*, hipStream_t);
typedef ncclResult_t (*fn_Recv)(void*, size_t, int, int, void*, hipStream_t);
typedef ncclResult_t (*fn_Group)(void);
const int kNcclInt8 = 0;      // ncclChar
typedef const char* (*fn_GetErrorString)(ncclResult_t);
const int kNcclFloat64 = 8;   // ncclDouble

struct Rccl {
    void* h = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_AllGather AllGather = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    fn_Send Send = nullptr;
    fn_Recv Recv = nullptr;
    fn_Group GroupStart = nullptr, GroupEnd = nullptr;
} g_rccl;

int rccl_load()
{
    if (g_rccl.h) return SMC_OK;
    // SMC_RCCL_LIBRARY: another build of RCCL (or, in the CPU tests, the test double)
    const char* names[] = {getenv("SMC_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        smc_set_error("RCCL not found: %s", dlerror());
        return SMC_ERR_HIP;
    }
    g_rccl.GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (fn_AllGather)dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(h, "ncclGetErrorString");
    g_rccl.Send = (fn_Send)dlsym(h, "ncclSend");
    g_rccl.Recv = (fn_Recv)dlsym(h, "ncclRecv");
    g_rccl.GroupStart = (fn_Group)dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (fn_Group)dlsym(h, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        smc_set_error("RCCL is missing expected symbols");
        dlclose(h);
        return SMC_ERR_HIP;
    }
    g_rccl.h = h;
    return SMC_OK;
}

int rccl_fail(const char* what, ncclResult_t r)
{
    smc_set_error("%s failed: %s", what,
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return SMC_ERR_HIP;
}
#ifdef SMC_EMULATE
bool use_rccl() { const char* e = getenv("SMC_RCCL_LIBRARY"); return e && *e; }
#else
constexpr bool use_rccl() { return true; }
#endif
}  // namespace

__global__ void __launch_bounds__(SMC_BLOCK)
k_copy_f64(const double* src, i64 n, double* dst)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" {

int smc_comm_unique_id(char* id_host)
{
    SMC_REQUIRE(id_host, "null argument");
    if (!use_rccl()) {
        memset(id_host, 0, SMC_COMM_ID_BYTES);
        return SMC_OK;
    }
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != 0) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_host, id.internal, SMC_COMM_ID_BYTES);
    return SMC_OK;
}

int smc_comm_create(smc_ctx* ctx, int nranks, int rank, const char* id_host, smc_comm** out)
{
    SMC_REQUIRE(ctx && id_host && out, "null argument");
    SMC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    smc_comm* c = new smc_comm();
    c->ctx = ctx;
    c->nccl = nullptr;
    c->nranks = nranks;
    c->rank = rank;
    if (!use_rccl()) {
        if (nranks != 1) {
            delete c;
            smc_set_error("the emulator build has no RCCL (nranks must be 1)");
            return SMC_ERR_INVALID;
        }
        *out = c;
        return SMC_OK;
    }
    int rc = rccl_load();
    if (rc) { delete c; return rc; }
    SMC_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, id_host, SMC_COMM_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, id, rank);
    if (r != 0) { delete c; return rccl_fail("ncclCommInitRank", r); }
    *out = c;
    return SMC_OK;
}

// enqueued on the context's stream, no synchronisation: the theta level of a sharded SMC^2 gathers the
// filters' evidence increments behind every time step without a host round trip (smc_filter_theta_enable_sharded)
int smc_comm_allgather_f64_async(smc_comm* c, const double* send, int64_t count, double* recv)
{
    SMC_REQUIRE(c && send && recv, "null argument");
    SMC_REQUIRE(count > 0, "count must be positive");
    hipStream_t st = c->ctx->stream;
    if (!c->nccl) {              // (emulator's one-rank stub)
        SMC_LAUNCH(k_copy_f64, dim3((unsigned)((count + SMC_BLOCK - 1) / SMC_BLOCK)), dim3(SMC_BLOCK),
                   st, send, (i64)count, recv);
    } else {
#ifdef SMC_EMULATE
        SMC_HIP_CHECK(hipStreamSynchronize(st));      // (the test double reads host memory: launches are done first)
#endif
        ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)count, kNcclFloat64, c->nccl, st);
        if (r != 0) return rccl_fail("ncclAllGather", r);
    }
    return SMC_OK;
}

int smc_comm_allgather_f64(smc_comm* c, const double* send, int64_t count, double* recv)
{
    const int rc = smc_comm_allgather_f64_async(c, send, count, recv);
    if (rc) return rc;
    SMC_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
    return SMC_OK;
}

int smc_comm_rank(smc_comm* c, int32_t* rank, int32_t* nranks)
{
    SMC_REQUIRE(c && rank && nranks, "null argument");
    *rank = c->rank;
    *nranks = c->nranks;
    return SMC_OK;
}

// All-to-all of byte blocks (island migration): rank p receives send[sdisp[p] .. + scount[p]) of
// every rank into recv[rdisp[.] ..]: grouped ncclSend / ncclRecv over xGMI (point-to-point links:
// one pair per peer, no ring), the self block included.  Counts and displacements in bytes (HOST).
int smc_comm_alltoallv(smc_comm* c, const void* send, const int64_t* scount, const int64_t* sdisp,
                       void* recv, const int64_t* rcount, const int64_t* rdisp)
{
    SMC_REQUIRE(c && scount && sdisp && rcount && rdisp, "null argument");
    hipStream_t st = c->ctx->stream;
    if (!c->nccl) {              // (emulator's one-rank stub)
        if (scount[0] != rcount[0]) { smc_set_error("alltoallv: self block sizes differ"); return SMC_ERR_INVALID; }
        if (scount[0])
            SMC_HIP_CHECK(hipMemcpyAsync((char*)recv + rdisp[0], (const char*)send + sdisp[0], (size_t)scount[0],
                                         hipMemcpyDeviceToDevice, st));
        SMC_HIP_CHECK(hipStreamSynchronize(st));
        return SMC_OK;
    }
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        smc_set_error("RCCL lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
        return SMC_ERR_HIP;
    }
    ncclResult_t r = g_rccl.GroupStart();
    if (r != 0) return rccl_fail("ncclGroupStart", r);
    for (int p = 0; p < c->nranks && r == 0; ++p) {
        if (scount[p]) r = g_rccl.Send((const char*)send + sdisp[p], (size_t)scount[p], kNcclInt8, p, c->nccl, st);
        if (r == 0 && rcount[p]) r = g_rccl.Recv((char*)recv + rdisp[p], (size_t)rcount[p], kNcclInt8, p, c->nccl, st);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r != 0) return rccl_fail("ncclSend / ncclRecv", r);
    if (r2 != 0) return rccl_fail("ncclGroupEnd", r2);
    SMC_HIP_CHECK(hipStreamSynchronize(st));
    return SMC_OK;
}

int smc_comm_destroy(smc_comm* c)
{
    if (!c) return SMC_OK;
    if (c->nccl) g_rccl.CommDestroy(c->nccl);
    delete c;
    return SMC_OK;
}

}  // extern "C"
