/* fake_rccl.c -- a TEST DOUBLE of the RCCL entry points libsmc_hip binds (smc_comm.hip), for the CPU
 * suite: several emulator processes on one host exchange their "device" buffers (host memory in the
 * emulator) through files in a directory named by the unique id.  It implements the semantics the
 * product relies on and CHECKS the protocol, so that the first multi-rank run on a node is not the
 * first time these calls are made:
 *   - datatypes: ncclChar = 0 (1 byte), ncclDouble = 8 (8 bytes); anything else is an error;
 *   - ncclAllGather(send, recv, count, type, comm, stream): count = elements PER RANK, recv holds
 *     nranks * count elements in rank order;
 *   - ncclSend / ncclRecv are only legal between ncclGroupStart and ncclGroupEnd here (an ungrouped
 *     pair of blocking calls between two ranks deadlocks on real hardware when both send first);
 *     nothing moves until the outermost ncclGroupEnd; byte counts of a matched pair must agree;
 *   - a communicator is usable only with the (nranks, rank) it was initialised with; every rank must
 *     present the same unique id.
 * Violations return ncclInvalidUsage (5) and ncclGetErrorString says what was wrong.
 * Built by tests/emu/build_emu.py into tests/emu/_build/libfake_rccl.so.  Not part of the product. */
#define _GNU_SOURCE
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

typedef struct {
    char dir[100];
    int nranks, rank;
    unsigned long seq_coll;            /* collectives done on this communicator */
    unsigned long seq_p2p[64][2];      /* per peer: sends, receives */
    int magic;
} Comm;

static __thread char g_err[256] = "no error";
static __thread int g_depth = 0;
enum { OP_SEND, OP_RECV };
typedef struct { int kind; void* buf; size_t bytes; int peer; Comm* c; } P2p;
static __thread P2p g_ops[256];
static __thread int g_nops = 0;

static ncclResult_t fail(int code, const char* fmt, const char* a, long b, long c)
{
    snprintf(g_err, sizeof g_err, fmt, a, b, c);
    return code;
}
static size_t type_bytes(int t) { return t == 0 ? 1 : (t == 8 ? 8 : 0); }

static int write_file(const char* path, const void* p, size_t n)
{
    char tmp[256];
    snprintf(tmp, sizeof tmp, "%s.tmp%d", path, (int)getpid());
    FILE* f = fopen(tmp, "wb");
    if (!f) return -1;
    if (n && fwrite(p, 1, n, f) != n) { fclose(f); return -1; }
    fclose(f);
    return rename(tmp, path);
}
/* waits (up to 120 s) for `path`, reads exactly n bytes; -2 if its size differs */
static int read_file(const char* path, void* p, size_t n)
{
    struct stat st;
    for (int i = 0; i < 120000; ++i) {
        if (stat(path, &st) == 0) {
            if ((size_t)st.st_size != n) return -2;
            FILE* f = fopen(path, "rb");
            if (!f) return -1;
            const size_t got = n ? fread(p, 1, n, f) : 0;
            fclose(f);
            return got == n ? 0 : -1;
        }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
    }
    return -3;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : g_err; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    const char* base = getenv("TMPDIR");
    snprintf(id->internal, 100, "%s/fake_rccl_%d_%ld", base && *base ? base : "/tmp", (int)getpid(), (long)time(NULL));
    if (mkdir(id->internal, 0700) != 0 && errno != EEXIST) return fail(ncclSystemError, "mkdir %s failed (%ld %ld)", id->internal, errno, 0);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks)
        return fail(ncclInvalidArgument, "ncclCommInitRank: bad nranks / rank%s (%ld, %ld)", "", nranks, rank);
    struct stat st;
    if (id.internal[0] != '/' || stat(id.internal, &st) != 0)
        return fail(ncclInvalidArgument, "ncclCommInitRank: unique id '%s' was not made by ncclGetUniqueId (%ld %ld)", id.internal, 0, 0);
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    snprintf(c->dir, sizeof c->dir, "%s", id.internal);
    c->nranks = nranks;
    c->rank = rank;
    c->magic = 0x52434c;
    /* every rank checks in with (nranks, rank); all must agree on nranks and be distinct */
    char path[256], buf[32];
    snprintf(path, sizeof path, "%s/init_%d", c->dir, rank);
    snprintf(buf, sizeof buf, "%d", nranks);
    if (stat(path, &st) == 0) { free(c); return fail(ncclInvalidUsage, "ncclCommInitRank: rank %s%ld joined twice (%ld)", "", rank, 0); }
    if (write_file(path, buf, strlen(buf))) { free(c); return fail(ncclSystemError, "cannot write %s (%ld %ld)", path, 0, 0); }
    for (int r = 0; r < nranks; ++r) {
        char other[32] = {0};
        snprintf(path, sizeof path, "%s/init_%d", c->dir, r);
        struct stat s2;
        int waited = 0;
        while (stat(path, &s2) != 0 && waited++ < 120000) { struct timespec ts = {0, 1000000}; nanosleep(&ts, NULL); }
        FILE* f = fopen(path, "rb");
        if (!f) { free(c); return fail(ncclSystemError, "rank %s%ld never joined (%ld)", "", r, 0); }
        if (!fgets(other, sizeof other, f)) other[0] = 0;
        fclose(f);
        if (atoi(other) != nranks) { free(c); return fail(ncclInvalidUsage, "ranks disagree on nranks%s (%ld vs %ld)", "", nranks, atoi(other)); }
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(void* comm)
{
    Comm* c = (Comm*)comm;
    if (!c || c->magic != 0x52434c) return fail(ncclInvalidArgument, "ncclCommDestroy: not a communicator%s (%ld %ld)", "", 0, 0);
    c->magic = 0;
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int type, void* comm, void* stream)
{
    (void)stream;
    Comm* c = (Comm*)comm;
    if (!c || c->magic != 0x52434c) return fail(ncclInvalidArgument, "ncclAllGather: not a communicator%s (%ld %ld)", "", 0, 0);
    const size_t tb = type_bytes(type);
    if (!tb) return fail(ncclInvalidArgument, "ncclAllGather: datatype %s%ld is neither ncclChar (0) nor ncclDouble (8) (%ld)", "", type, 0);
    if (!send || !recv || !count) return fail(ncclInvalidArgument, "ncclAllGather: null buffer or zero count%s (%ld %ld)", "", (long)count, 0);
    if (g_depth) return fail(ncclInvalidUsage, "ncclAllGather inside a group is not what smc_comm does%s (%ld %ld)", "", 0, 0);
    char path[256];
    const unsigned long s = c->seq_coll++;
    snprintf(path, sizeof path, "%s/ag_%lu_%d", c->dir, s, c->rank);
    if (write_file(path, send, count * tb)) return fail(ncclSystemError, "cannot write %s (%ld %ld)", path, 0, 0);
    for (int r = 0; r < c->nranks; ++r) {
        snprintf(path, sizeof path, "%s/ag_%lu_%d", c->dir, s, r);
        const int rc = read_file(path, (char*)recv + (size_t)r * count * tb, count * tb);
        if (rc == -2) return fail(ncclInvalidUsage, "ncclAllGather: rank %s%ld contributed another count than mine (%ld elements)", "", r, (long)count);
        if (rc) return fail(ncclSystemError, "ncclAllGather: rank %s%ld never arrived (%ld)", "", r, rc);
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++g_depth; return ncclSuccess; }

static ncclResult_t p2p(int kind, void* buf, size_t count, int type, int peer, void* comm)
{
    Comm* c = (Comm*)comm;
    if (!c || c->magic != 0x52434c) return fail(ncclInvalidArgument, "ncclSend/Recv: not a communicator%s (%ld %ld)", "", 0, 0);
    const size_t tb = type_bytes(type);
    if (!tb) return fail(ncclInvalidArgument, "ncclSend/Recv: datatype %s%ld unknown (%ld)", "", type, 0);
    if (peer < 0 || peer >= c->nranks) return fail(ncclInvalidArgument, "ncclSend/Recv: peer %s%ld out of range (nranks %ld)", "", peer, c->nranks);
    if (!g_depth) return fail(ncclInvalidUsage, "ncclSend/Recv outside ncclGroupStart/End: blocking pairs deadlock%s (%ld %ld)", "", 0, 0);
    if (!buf || !count) return fail(ncclInvalidArgument, "ncclSend/Recv: null buffer or zero count (peer %s%ld, %ld)", "", peer, (long)count);
    if (g_nops >= 256) return fail(ncclInvalidUsage, "too many operations in one group%s (%ld %ld)", "", 0, 0);
    g_ops[g_nops].kind = kind; g_ops[g_nops].buf = buf; g_ops[g_nops].bytes = count * tb;
    g_ops[g_nops].peer = peer; g_ops[g_nops].c = c;
    ++g_nops;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, int type, int peer, void* comm, void* stream)
{
    (void)stream;
    return p2p(OP_SEND, (void*)buf, count, type, peer, comm);
}
ncclResult_t ncclRecv(void* buf, size_t count, int type, int peer, void* comm, void* stream)
{
    (void)stream;
    return p2p(OP_RECV, buf, count, type, peer, comm);
}

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart%s (%ld %ld)", "", 0, 0);
    if (--g_depth) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    char path[256];
    /* all sends first (they never block), then the receives: the order real RCCL is free to choose */
    for (int i = 0; i < g_nops && rc == ncclSuccess; ++i) {
        P2p* o = &g_ops[i];
        if (o->kind != OP_SEND) continue;
        snprintf(path, sizeof path, "%s/p2p_%d_%d_%lu", o->c->dir, o->c->rank, o->peer, o->c->seq_p2p[o->peer][0]++);
        if (write_file(path, o->buf, o->bytes)) rc = fail(ncclSystemError, "cannot write %s (%ld %ld)", path, 0, 0);
    }
    for (int i = 0; i < g_nops && rc == ncclSuccess; ++i) {
        P2p* o = &g_ops[i];
        if (o->kind != OP_RECV) continue;
        snprintf(path, sizeof path, "%s/p2p_%d_%d_%lu", o->c->dir, o->peer, o->c->rank, o->c->seq_p2p[o->peer][1]++);
        const int r = read_file(path, o->buf, o->bytes);
        if (r == -2) rc = fail(ncclInvalidUsage, "ncclRecv from %s%ld: the sender's byte count differs from mine (%ld)", "", o->peer, (long)o->bytes);
        else if (r) rc = fail(ncclSystemError, "ncclRecv from %s%ld: nothing was sent (%ld)", "", o->peer, r);
        else unlink(path);
    }
    g_nops = 0;
    return rc;
}
