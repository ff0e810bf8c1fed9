// clfan.cpp -- the fan-out / fan-in of raw stream buffers over RCCL point-to-point (see include/cariboulite_fanout.h).
// Host code only: RCCL does the transfers (direct xGMI peer copies for ranks of one node).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "../../../include/cariboulite_fanout.h"

static_assert(sizeof(ncclUniqueId) == CLFAN_ID_BYTES, "ncclUniqueId is 128 bytes");

static thread_local char g_err[384] = "";
static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *clfan_last_error(void) { return g_err; }

#define CLFAN_NCCL(expr)                                                                   \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess) { set_err("%s: %s", #expr, ncclGetErrorString(r_)); return -1; } \
    } while (0)
#define CLFAN_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) { set_err("%s: %s", #expr, hipGetErrorString(e_)); return -1; } \
    } while (0)

struct clfan_comm {
    ncclComm_t comm;       // NULL when world == 1 (nothing to talk to)
    int world, rank;
};

extern "C" int clfan_unique_id(uint8_t id[CLFAN_ID_BYTES])
{
    ncclUniqueId u;
    CLFAN_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return 0;
}

extern "C" clfan_comm *clfan_create(const uint8_t id[CLFAN_ID_BYTES], int world, int rank)
{
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) { set_err("clfan_create: bad arguments"); return nullptr; }
    clfan_comm *c = new (std::nothrow) clfan_comm();
    if (!c) return nullptr;
    c->comm = nullptr; c->world = world; c->rank = rank;
    if (world > 1) {
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) { set_err("ncclCommInitRank: %s", ncclGetErrorString(r)); delete c; return nullptr; }
    }
    return c;
}

extern "C" void clfan_destroy(clfan_comm *c)
{
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

extern "C" int clfan_world(const clfan_comm *c) { return c ? c->world : 0; }
extern "C" int clfan_rank(const clfan_comm *c) { return c ? c->rank : -1; }

extern "C" int clfan_local_count(int n_streams, int world, int rank)
{
    if (n_streams <= rank || world < 1) return 0;
    return (n_streams - rank + world - 1) / world;
}

// One pass over the streams; `to_peers` = scatter (root sends), else gather (root receives).  Every transfer of the
// call sits in ONE group, so RCCL drives all peers' links concurrently.
static int exchange(clfan_comm *c, int root, const uint8_t *d_root, size_t root_stride, size_t bytes, int n_streams,
                    uint8_t *d_mine, size_t mine_stride, hipStream_t s, bool to_peers)
{
    if (!c || root < 0 || root >= c->world || n_streams < 0) { set_err("clfan: bad arguments"); return -1; }
    if (bytes == 0 || n_streams == 0) return 0;
    const int W = c->world, me = c->rank;
    if (me == root && !d_root) { set_err("clfan: the root needs its buffer"); return -1; }
    if (clfan_local_count(n_streams, W, me) > 0 && !d_mine) { set_err("clfan: this rank owns streams but has no buffer"); return -1; }
    if (W > 1) CLFAN_NCCL(ncclGroupStart());
    int rc = 0;
    for (int st = 0; st < n_streams && !rc; st++) {
        const int owner = st % W;
        const size_t j = (size_t)(st / W);                     // index among the owner's streams
        if (me == root && owner == root) {                      // stays on this GPU
            const void *src = to_peers ? (const void *)(d_root + (size_t)st * root_stride) : (const void *)(d_mine + j * mine_stride);
            void *dst = to_peers ? (void *)(d_mine + j * mine_stride) : (void *)(const_cast<uint8_t *>(d_root) + (size_t)st * root_stride);
            if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_err("clfan: local copy failed"); rc = -1; }
        } else if (me == root) {
            uint8_t *p = const_cast<uint8_t *>(d_root) + (size_t)st * root_stride;
            ncclResult_t r = to_peers ? ncclSend(p, bytes, ncclUint8, owner, c->comm, s) : ncclRecv(p, bytes, ncclUint8, owner, c->comm, s);
            if (r != ncclSuccess) { set_err("clfan: root transfer: %s", ncclGetErrorString(r)); rc = -1; }
        } else if (me == owner) {
            uint8_t *p = d_mine + j * mine_stride;
            ncclResult_t r = to_peers ? ncclRecv(p, bytes, ncclUint8, root, c->comm, s) : ncclSend(p, bytes, ncclUint8, root, c->comm, s);
            if (r != ncclSuccess) { set_err("clfan: peer transfer: %s", ncclGetErrorString(r)); rc = -1; }
        }
    }
    if (W > 1) {
        ncclResult_t r = ncclGroupEnd();
        if (r != ncclSuccess && !rc) { set_err("ncclGroupEnd: %s", ncclGetErrorString(r)); rc = -1; }
    }
    return rc;
}

extern "C" int clfan_scatter_streams(clfan_comm *c, int root, const void *d_root, size_t root_stride_bytes, size_t stream_bytes,
                                     int n_streams, void *d_mine, size_t mine_stride_bytes, void *stream)
{
    return exchange(c, root, (const uint8_t *)d_root, root_stride_bytes, stream_bytes, n_streams, (uint8_t *)d_mine,
                    mine_stride_bytes, (hipStream_t)stream, true);
}

extern "C" int clfan_gather_streams(clfan_comm *c, int root, const void *d_mine, size_t mine_stride_bytes, size_t stream_bytes,
                                    int n_streams, void *d_root, size_t root_stride_bytes, void *stream)
{
    return exchange(c, root, (const uint8_t *)d_root, root_stride_bytes, stream_bytes, n_streams,
                    const_cast<uint8_t *>((const uint8_t *)d_mine), mine_stride_bytes, (hipStream_t)stream, false);
}
