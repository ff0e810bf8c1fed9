// TEST-ONLY in-process model of the RCCL point-to-point calls clfan.cpp uses (see rccl/rccl.h in this directory).
// Ranks are threads of one process, "device" memory is host memory.  The model keeps the properties of the real library
// that a wrong schedule trips over:
//   * ncclCommInitRank is collective: it returns once all `nranks` ranks of the id have joined (or times out);
//   * sends and receives between one (source, destination) pair match IN ORDER, and a matched pair must agree on the
//     byte count -- a mismatch is an error on the receiving side;
//   * a send completes only when a receive has taken it and a receive only when a send has arrived: an operation with
//     no partner never completes.  The real library hangs there; the model gives up after a timeout and says which
//     operation was left (rccl_mock_last_error), so a deadlocking schedule FAILS the test instead of freezing it;
//   * operations issued between ncclGroupStart and ncclGroupEnd progress together (no order among them), operations
//     outside a group are a group of one -- so a schedule that only works because it is grouped shows up when the
//     group calls are taken out.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg { const void *src; size_t bytes; bool taken; };

struct World {
    int nranks = 0, joined = 0, left = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<Msg *>> box;          // [src * nranks + dst]: sends posted and not yet taken, in order
};

std::mutex g_mu;
std::map<std::string, World *> g_worlds;
unsigned long g_next_id = 1;
int g_timeout_ms = 5000;

thread_local char t_err[512] = "";
void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
}

struct Op { bool send; void *buf; size_t bytes; int peer; ncclComm *comm; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

}  // namespace

struct ncclComm {
    World *w;
    int rank;
};
// per THREAD (= per rank in the harness): what this rank has posted so far
static thread_local unsigned long t_groups = 0, t_sends = 0, t_recvs = 0, t_max_group_ops = 0;

static ncclResult_t run_ops(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    // (system_clock: libstdc++ then waits with pthread_cond_timedwait, which ThreadSanitizer understands; the steady-clock wait is
    // pthread_cond_clockwait, which gcc 11's TSan does not intercept and reports as a double lock)
    const auto deadline = std::chrono::system_clock::now() + std::chrono::milliseconds(g_timeout_ms);
    ncclComm *c0 = ops[0].comm;
    World *w = c0->w;
    t_groups++;
    if (ops.size() > t_max_group_ops) t_max_group_ops = (unsigned long)ops.size();
    std::vector<Msg> mine;
    mine.reserve(ops.size());
    std::unique_lock<std::mutex> lk(w->mu);
    for (Op &o : ops) {
        if (o.comm->w != w) { set_err("one group spans two communicators"); return ncclInvalidUsage; }
        if (!o.send) continue;
        mine.push_back(Msg{o.buf, o.bytes, false});
        w->box[(size_t)o.comm->rank * w->nranks + o.peer].push_back(&mine.back());
        t_sends++;
    }
    w->cv.notify_all();
    ncclResult_t rc = ncclSuccess;
    for (Op &o : ops) {
        if (o.send) continue;
        std::deque<Msg *> &q = w->box[(size_t)o.peer * w->nranks + o.comm->rank];
        if (!w->cv.wait_until(lk, deadline, [&] { return !q.empty(); })) {
            set_err("rank %d: ncclRecv of %zu bytes from rank %d found no matching send (the real library would hang)", o.comm->rank, o.bytes, o.peer);
            rc = ncclInternalError;
            break;
        }
        Msg *m = q.front();
        q.pop_front();
        if (m->bytes != o.bytes) {
            set_err("rank %d: ncclRecv of %zu bytes from rank %d met a send of %zu bytes", o.comm->rank, o.bytes, o.peer, m->bytes);
            rc = ncclInvalidArgument;
        } else if (o.bytes)
            memcpy(o.buf, m->src, o.bytes);
        m->taken = true;
        t_recvs++;
        w->cv.notify_all();
        if (rc != ncclSuccess) break;
    }
    // our sends have to be taken before the buffers (and the Msg records) may go away
    for (Msg &m : mine) {
        if (w->cv.wait_until(lk, deadline, [&] { return m.taken; })) continue;
        if (rc == ncclSuccess) {
            set_err("rank %d: an ncclSend of %zu bytes was never received (the real library would hang)", c0->rank, m.bytes);
            rc = ncclInternalError;
        }
        for (auto &q : w->box)                     // withdraw it: the record dies with this call
            for (auto it = q.begin(); it != q.end();)
                it = (*it == &m) ? q.erase(it) : it + 1;
    }
    return rc;
}

extern "C" {

const char *rccl_mock_last_error(void) { return t_err; }
void rccl_mock_set_timeout_ms(int ms) { g_timeout_ms = ms; }
void rccl_mock_counters(unsigned long *groups, unsigned long *sends, unsigned long *recvs, unsigned long *max_group_ops)
{
    if (groups) *groups = t_groups;
    if (sends) *sends = t_sends;
    if (recvs) *recvs = t_recvs;
    if (max_group_ops) *max_group_ops = t_max_group_ops;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> g(g_mu);
    snprintf(id->internal, sizeof id->internal, "rccl-mock-%lu", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) { set_err("ncclCommInitRank: bad arguments"); return ncclInvalidArgument; }
    World *w;
    {
        std::lock_guard<std::mutex> g(g_mu);
        const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
        World *&slot = g_worlds[key];
        if (!slot) { slot = new World(); slot->nranks = nranks; slot->box.resize((size_t)nranks * nranks); }
        w = slot;
    }
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->nranks != nranks) { set_err("ncclCommInitRank: ranks disagree about the world size"); return ncclInvalidArgument; }
    w->joined++;
    w->cv.notify_all();
    if (!w->cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(g_timeout_ms), [&] { return w->joined >= nranks; })) {
        set_err("ncclCommInitRank: only %d of %d ranks joined", w->joined, nranks);
        return ncclInternalError;
    }
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    World *w = comm->w;
    bool last;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        last = ++w->left == w->nranks;
    }
    delete comm;
    if (last) {
        std::lock_guard<std::mutex> g(g_mu);
        for (auto it = g_worlds.begin(); it != g_worlds.end(); ++it)
            if (it->second == w) { g_worlds.erase(it); break; }
        delete w;
    }
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument";
    case ncclInvalidUsage: return "invalid usage";
    case ncclInternalError: return t_err[0] ? t_err : "internal error";
    default: return "error";
    }
}

ncclResult_t ncclGroupStart(void) { t_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) { set_err("ncclGroupEnd without ncclGroupStart"); return ncclInvalidUsage; }
    if (--t_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

static ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm)
{
    if (!comm || peer < 0 || peer >= comm->w->nranks || (count && !buf)) { set_err("ncclSend/ncclRecv: bad arguments (peer %d)", peer); return ncclInvalidArgument; }
    const size_t el = (t == ncclInt8 || t == ncclUint8) ? 1 : 4;
    t_ops.push_back(Op{send, buf, count * el, peer, comm});
    if (t_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t) { return p2p(true, const_cast<void *>(buf), count, t, peer, comm); }
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t) { return p2p(false, buf, count, t, peer, comm); }

}  // extern "C"
