/* rccl/rccl.h -- TEST DOUBLE of the slice of RCCL's API that csrc/mipt_multi.cpp uses.  Test infrastructure only.
 *
 * Why: the one-GPU test box cannot build an RCCL communicator with more than one rank (RCCL refuses a device listed twice), so the
 * n > 1 code of mipt_render_multi -- per-device host threads, share computation, root-only receive offsets, the zero-share branch,
 * the drain-on-failure path, stats aggregation -- would first run on the 8-GPU node.  `make multitest` compiles mipt_multi.cpp
 * against THIS header (-I tests/cpp/rccl_double ahead of /opt/rocm/include) and links rccl_double.hip instead of -lrccl, giving
 * libmipt_multitest.so: N logical ranks on ONE device, the collectives done with stream-ordered copies and a rank-ordered sum
 * kernel.  The product libmipt.so is never built from this directory (tests/test_abi.py checks it links the real librccl).
 *
 * Same names, types and enumerator values as /opt/rocm/include/rccl/rccl.h for everything declared here. */
#ifndef MIPT_RCCL_DOUBLE_H
#define MIPT_RCCL_DOUBLE_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_RCCL_DOUBLE 1

typedef struct ncclComm *ncclComm_t;

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7, ncclNumResults = 8 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);      /* the double accepts a device listed twice */
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *asyncError);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
/* ncclFloat only.  recvbuff is read on the root rank only. */
ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm,
                        hipStream_t stream);

/* ---- fault injection (exported by libmipt_multitest.so for tests/test_gpu_multirank.py) ----
 * kind 1: the next collective call queued by `rank` returns ncclInternalError (and the group it belongs to moves no data);
 * kind 2: the next ncclCommGetAsyncError on `rank`'s communicator reports ncclRemoteError once;
 * kind 0: clear.  Returns the number of collectives the double has completed so far (a liveness counter for the tests). */
__attribute__((visibility("default"))) long long rccl_double_inject(int rank, int kind);

#ifdef __cplusplus
}
#endif
#endif
