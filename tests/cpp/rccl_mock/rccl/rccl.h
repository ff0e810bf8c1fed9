/* TEST-ONLY stand-in for <rccl/rccl.h>: the slice of the RCCL API that cariboulite_amd/csrc/fanout/clfan.cpp calls,
 * implemented in-process by tests/cpp/rccl_mock/rccl_mock.cpp (threads as ranks, host memory as "device" memory).
 * NOT RCCL and never part of the product: it exists so that the world > 1 branch of clfan_scatter_streams /
 * clfan_gather_streams -- the schedule, the peer / row arithmetic and the pairing of sends and receives inside one
 * group, i.e. what can deadlock on a real node -- runs on a box without GPUs (tests/test_clfan_mock.py). */
#ifndef RCCL_MOCK_RCCL_H
#define RCCL_MOCK_RCCL_H
#include <stddef.h>
#include <hip/hip_runtime.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclFloat32 = 7 } ncclDataType_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);   /* blocks until all ranks joined */
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char *ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);          /* runs the group's transfers to completion (the real one queues them on the stream) */
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s);
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s);
/* the mock's own: what went wrong in this thread's last failing call, and counters of the CALLING THREAD (= one rank) the harness checks the schedule with */
const char *rccl_mock_last_error(void);
void rccl_mock_counters(unsigned long *groups, unsigned long *sends, unsigned long *recvs, unsigned long *max_group_ops);
void rccl_mock_set_timeout_ms(int ms);
#ifdef __cplusplus
}
#endif
#endif
