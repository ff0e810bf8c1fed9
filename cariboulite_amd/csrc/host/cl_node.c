/* cl_node.c -- ONE call over the stream groups of SEVERAL GPUs: the node-level form of cl_group_readStream / cl_group_writeStream.
 *
 * The reference's unit is one SoapySDR device per channel (soapy_api/SoapyCariboulite.cpp:46-69); independent channel streams shard over
 * the GPUs of a node with no data-path collective (stream s -> GPU s mod N).  bench.py does that with one process
 * per GPU; a Soapy client is ONE process with all its boards, so the same sharding has an in-process form: the devices are made with their
 * `gpu` kwarg, cl_node_make sorts them into one cl_group per GPU (a group lives on one GPU: its slab, streams, events and launches), and a
 * call runs every group's call at once -- shard 0 on the caller's thread, every other shard on a worker thread of its own that lives as
 * long as the node (a group's call sets its device first, so a worker belongs to its GPU).  Nothing is exchanged between the shards:
 * rets[i] and buffs[i] are member i's, exactly as in N cl_readStream / cl_writeStream calls; the node's return value is the groups' sum
 * (or -1 if a group failed: cl_node_last_error names it).
 *
 * kwarg SHARDS=<k> (default 1) cuts every GPU's members into k groups (contiguous blocks in the caller's order): a rehearsal of the
 * several-groups-at-once shape on a one-GPU box (tests) -- on one GPU it buys nothing (profiles/r04/group_ab_several_groups_per_gpu_no_effect.txt).
 * Every other kwarg goes to cl_group_make. */
#include "cl_internal.h"

typedef struct {
    cl_group *g;
    size_t n; size_t *member;             /* indices into the node's device table, in the caller's order */
    cl_device **devs;
    void **buffs; int *rets;              /* this shard's slice of a call's arguments */
    int result;
    pthread_t th; int has_thread;
    struct cl_node *node; int index;
} node_shard;

struct cl_node {
    size_t n; size_t n_shards; node_shard *sh;
    int dir;
    pthread_mutex_t mu; pthread_cond_t go, done;
    unsigned long gen; int pending, stop;
    int op; size_t num_elems; long timeout_us;      /* the call in flight: 0 read, 1 write, 2 flush */
    char err[256];
};

static char g_node_err[256];
const char *cl_node_last_error(const cl_node *nd) { return nd ? nd->err : g_node_err; }

static void shard_run(node_shard *s, int op, size_t num_elems, long timeout_us)
{
    if (op == 0) s->result = cl_group_readStream(s->g, (void *const *)s->buffs, num_elems, s->rets, timeout_us);
    else if (op == 1) s->result = cl_group_writeStream(s->g, (const void *const *)s->buffs, num_elems, s->rets, timeout_us);
    else s->result = cl_group_flush(s->g);
}

static void *shard_thread(void *arg)
{
    node_shard *s = (node_shard *)arg;
    cl_node *nd = s->node;
    unsigned long seen = 0;
    pthread_mutex_lock(&nd->mu);
    for (;;) {
        while (nd->gen == seen && !nd->stop) pthread_cond_wait(&nd->go, &nd->mu);
        if (nd->stop) break;
        seen = nd->gen;
        const int op = nd->op; const size_t num = nd->num_elems; const long to = nd->timeout_us;
        pthread_mutex_unlock(&nd->mu);
        shard_run(s, op, num, to);
        pthread_mutex_lock(&nd->mu);
        if (--nd->pending == 0) pthread_cond_broadcast(&nd->done);
    }
    pthread_mutex_unlock(&nd->mu);
    return NULL;
}

void cl_node_unmake(cl_node *nd)
{
    if (!nd) return;
    pthread_mutex_lock(&nd->mu);
    nd->stop = 1;
    pthread_cond_broadcast(&nd->go);
    pthread_mutex_unlock(&nd->mu);
    for (size_t s = 0; s < nd->n_shards; s++) {
        node_shard *sh = &nd->sh[s];
        if (sh->has_thread) pthread_join(sh->th, NULL);
        cl_group_unmake(sh->g);
        free(sh->member); free(sh->devs); free(sh->buffs); free(sh->rets);
    }
    free(nd->sh);
    pthread_mutex_destroy(&nd->mu); pthread_cond_destroy(&nd->go); pthread_cond_destroy(&nd->done);
    free(nd);
}

cl_node *cl_node_make(cl_device *const *devs, size_t n, const char *const *keys, const char *const *vals, size_t n_kwargs)
{
    g_node_err[0] = 0;
    if (!devs || !n) { cl_seterr(g_node_err, sizeof g_node_err, "cl_node_make: no devices"); return NULL; }
    for (size_t i = 0; i < n; i++)
        if (!devs[i] || !devs[i]->stream) { cl_seterr(g_node_err, sizeof g_node_err, "cl_node_make: device %zu is missing", i); return NULL; }
    for (size_t i = 1; i < n; i++)
        if (devs[i]->stream->native_dir != devs[0]->stream->native_dir) {
            cl_seterr(g_node_err, sizeof g_node_err, "cl_node_make: device %zu is set up for the other direction (a node reads or writes)", i);
            return NULL;
        }
    /* the group kwargs: everything but SHARDS */
    const char **gk = (const char **)calloc(n_kwargs + 1, sizeof *gk), **gv = (const char **)calloc(n_kwargs + 1, sizeof *gv);
    size_t n_gk = 0; int per_gpu = 1;
    if (!gk || !gv) { free(gk); free(gv); return NULL; }
    for (size_t i = 0; i < n_kwargs; i++) {
        if (!keys || !vals || !keys[i] || !vals[i]) continue;
        if (!strcmp(keys[i], "SHARDS")) { per_gpu = atoi(vals[i]); continue; }
        gk[n_gk] = keys[i]; gv[n_gk] = vals[i]; n_gk++;
    }
    if (per_gpu < 1) per_gpu = 1;
    if (per_gpu > 16) per_gpu = 16;
    cl_node *nd = (cl_node *)calloc(1, sizeof *nd);
    if (!nd) { free(gk); free(gv); return NULL; }
    pthread_mutex_init(&nd->mu, NULL); pthread_cond_init(&nd->go, NULL); pthread_cond_init(&nd->done, NULL);
    nd->n = n;
    nd->dir = devs[0]->stream->native_dir;
    nd->sh = (node_shard *)calloc(n, sizeof *nd->sh);          /* (at most one shard per member) */
    size_t *of_gpu = (size_t *)calloc(n, sizeof *of_gpu);
    uint8_t *taken = (uint8_t *)calloc(n, 1);
    int bad = !nd->sh || !of_gpu || !taken;
    for (size_t first = 0; first < n && !bad; first++) {
        if (taken[first]) continue;
        /* the members on this member's GPU, in the caller's order */
        const int gpu = devs[first]->smi->device;
        size_t cnt = 0;
        for (size_t i = first; i < n; i++)
            if (!taken[i] && devs[i]->smi->device == gpu) { of_gpu[cnt++] = i; taken[i] = 1; }
        const size_t k = (size_t)per_gpu < cnt ? (size_t)per_gpu : cnt, each = (cnt + k - 1) / k;
        for (size_t lo = 0; lo < cnt && !bad; lo += each) {
            const size_t m = cnt - lo < each ? cnt - lo : each;
            node_shard *sh = &nd->sh[nd->n_shards];
            sh->node = nd; sh->index = (int)nd->n_shards; sh->n = m;
            sh->member = (size_t *)calloc(m, sizeof *sh->member); sh->devs = (cl_device **)calloc(m, sizeof *sh->devs);
            sh->buffs = (void **)calloc(m, sizeof *sh->buffs); sh->rets = (int *)calloc(m, sizeof *sh->rets);
            nd->n_shards++;
            if (!sh->member || !sh->devs || !sh->buffs || !sh->rets) { bad = 1; break; }
            for (size_t j = 0; j < m; j++) { sh->member[j] = of_gpu[lo + j]; sh->devs[j] = devs[of_gpu[lo + j]]; }
            sh->g = cl_group_make(sh->devs, m, gk, gv, n_gk);
            if (!sh->g) { cl_seterr(g_node_err, sizeof g_node_err, "cl_node_make: the group of GPU %d: %s", gpu, cl_group_last_error(NULL)); bad = 1; }
        }
    }
    for (size_t s = 1; s < nd->n_shards && !bad; s++) {          /* (shard 0 runs on the caller's thread) */
        if (pthread_create(&nd->sh[s].th, NULL, shard_thread, &nd->sh[s])) { cl_seterr(g_node_err, sizeof g_node_err, "cl_node_make: threads"); bad = 1; break; }
        nd->sh[s].has_thread = 1;
    }
    free(of_gpu); free(taken); free(gk); free(gv);
    if (bad) { if (!g_node_err[0]) cl_seterr(g_node_err, sizeof g_node_err, "cl_node_make: out of memory"); cl_node_unmake(nd); return NULL; }
    return nd;
}

size_t cl_node_size(const cl_node *nd) { return nd ? nd->n : 0; }
size_t cl_node_shards(const cl_node *nd) { return nd ? nd->n_shards : 0; }
cl_group *cl_node_group(const cl_node *nd, size_t shard) { return nd && shard < nd->n_shards ? nd->sh[shard].g : NULL; }
int cl_node_shard_of(const cl_node *nd, size_t member)
{
    for (size_t s = 0; nd && s < nd->n_shards; s++)
        for (size_t j = 0; j < nd->sh[s].n; j++)
            if (nd->sh[s].member[j] == member) return (int)s;
    return -1;
}

static int node_call(cl_node *nd, int op, void *const *buffs, size_t num_elems, int *rets, long timeout_us)
{
    if (!nd || (op != 2 && (!buffs || !rets))) return -1;
    nd->err[0] = 0;
    for (size_t s = 0; s < nd->n_shards && op != 2; s++)
        for (size_t j = 0; j < nd->sh[s].n; j++) nd->sh[s].buffs[j] = buffs[nd->sh[s].member[j]];
    pthread_mutex_lock(&nd->mu);
    nd->op = op; nd->num_elems = num_elems; nd->timeout_us = timeout_us;
    nd->pending = (int)nd->n_shards - 1;
    nd->gen++;
    pthread_cond_broadcast(&nd->go);
    pthread_mutex_unlock(&nd->mu);
    shard_run(&nd->sh[0], op, num_elems, timeout_us);
    pthread_mutex_lock(&nd->mu);
    while (nd->pending) pthread_cond_wait(&nd->done, &nd->mu);
    pthread_mutex_unlock(&nd->mu);
    int total = 0, failed = 0;
    for (size_t s = 0; s < nd->n_shards; s++) {
        node_shard *sh = &nd->sh[s];
        for (size_t j = 0; j < sh->n && op != 2; j++) rets[sh->member[j]] = sh->rets[j];
        if (sh->result < 0) {
            if (!failed) cl_seterr(nd->err, sizeof nd->err, "shard %zu (GPU %d): %s", s, sh->devs[0]->smi->device, cl_group_last_error(sh->g));
            failed = 1;
        } else total += sh->result;
    }
    return failed ? -1 : total;
}

int cl_node_readStream(cl_node *nd, void *const *buffs, size_t numElems, int *rets, long timeoutUs)
{
    return node_call(nd, 0, buffs, numElems, rets, timeoutUs);
}

int cl_node_writeStream(cl_node *nd, const void *const *buffs, size_t numElems, int *rets, long timeoutUs)
{
    return node_call(nd, 1, (void *const *)buffs, numElems, rets, timeoutUs);
}

int cl_node_flush(cl_node *nd) { return node_call(nd, 2, NULL, 0, NULL, 0) < 0 ? -1 : 0; }

/* cl_group_register_buffers / _unregister_buffers of every group: buffs[i] is member i's buffer (bytes_each long), in the order the
 * devices were given.  0, or -1 with every registration made by this call taken back (cl_node_last_error). */
int cl_node_register_buffers(cl_node *nd, void *const *buffs, size_t bytes_each)
{
    if (!nd || !buffs || !bytes_each) return -1;
    nd->err[0] = 0;
    for (size_t s = 0; s < nd->n_shards; s++) {
        node_shard *sh = &nd->sh[s];
        for (size_t j = 0; j < sh->n; j++) sh->buffs[j] = buffs[sh->member[j]];
        if (cl_group_register_buffers(sh->g, (void *const *)sh->buffs, bytes_each)) {
            cl_seterr(nd->err, sizeof nd->err, "shard %zu (GPU %d): %s", s, sh->devs[0]->smi->device, cl_group_last_error(sh->g));
            for (size_t q = 0; q < s; q++) cl_group_unregister_buffers(nd->sh[q].g);
            return -1;
        }
    }
    return 0;
}

void cl_node_unregister_buffers(cl_node *nd)
{
    for (size_t s = 0; nd && s < nd->n_shards; s++) cl_group_unregister_buffers(nd->sh[s].g);
}
