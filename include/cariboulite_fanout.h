/*
 * cariboulite_fanout.h -- C-ABI of the one multi-GPU exchange step of the hot path (libcariboulite_fanout.so).
 *
 * Channel streams are independent (the reference models each channel as its own SoapySDR device,
 * soapy_api/SoapyCariboulite.cpp:46-69), so the path shards by stream -- stream s lives on rank s mod world -- and
 * needs no collective.  The exception is the fan-out case: ONE GPU receives the raw SMI buffers of all streams
 * (a capture box, a SoapyRemote fan-in) and hands every other GPU the streams it owns, and optionally collects
 * per-stream results again.  On MI355X xGMI is point-to-point (7 links per GPU): this is a set of direct
 * ncclSend / ncclRecv pairs inside ONE group, so that all links carry traffic at once -- never a ring broadcast,
 * which a single link would bound (SURVEY.md section 5 "Distributed comm backend", section 8e).
 *
 * One process per GPU.  Host code stays C: plain pointers and sizes; `void *stream` is a hipStream_t.
 * The 128-byte id is RCCL's ncclUniqueId; rank 0 makes it and the caller carries it to the other ranks by
 * whatever it has (torch.distributed, MPI, a file).
 */
#ifndef CARIBOULITE_FANOUT_H
#define CARIBOULITE_FANOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLFAN_ID_BYTES 128

typedef struct clfan_comm clfan_comm;

int         clfan_unique_id(uint8_t id[CLFAN_ID_BYTES]);                           /* rank 0 */
clfan_comm *clfan_create(const uint8_t id[CLFAN_ID_BYTES], int world, int rank);   /* on the CURRENT HIP device */
void        clfan_destroy(clfan_comm *c);
int         clfan_world(const clfan_comm *c);
int         clfan_rank(const clfan_comm *c);
const char *clfan_last_error(void);

/* How many of n_streams live on `rank` (stream s -> rank s mod world), and the s-th local stream's global index. */
int         clfan_local_count(int n_streams, int world, int rank);

/*
 * Scatter: on `root`, stream s is stream_bytes long at d_root + s * root_stride_bytes.  Afterwards every rank holds
 * its streams, in increasing s, at d_mine + j * mine_stride_bytes (j = 0 .. clfan_local_count()-1).  The root's own
 * streams are device-to-device copies.  Asynchronous on `stream`.  d_root is ignored on the other ranks.
 */
int clfan_scatter_streams(clfan_comm *c, int root, const void *d_root, size_t root_stride_bytes, size_t stream_bytes,
                          int n_streams, void *d_mine, size_t mine_stride_bytes, void *stream);
/* Gather: the inverse (per-stream results of out_bytes each back to `root`). */
int clfan_gather_streams(clfan_comm *c, int root, const void *d_mine, size_t mine_stride_bytes, size_t stream_bytes,
                         int n_streams, void *d_root, size_t root_stride_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
