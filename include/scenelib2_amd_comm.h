/* scenelib2_amd_comm.h - sharding independent sequences over the GPUs of one node from HOST C / C++ (RCCL over xGMI).
 *
 * Library: scenelib2_amd/libscenelib2_amd_comm.so (links librccl and libscenelib2_amd.so; the engine library itself has no
 * communication dependency).  SURVEY.md 8(e): the reference is a single instance with no communication at all
 * (monoslam.h:158-218, framegrabber.h:69-70); sequences are independent, so a step has NO collective - every rank owns its
 * block of sequences and steps it with its own sl2_engine.  Communication exists only at the edges, and that is all this
 * header offers:
 *   sl2_scatter_frames   one step's frames from the rank that has them to the ranks that own the sequences
 *                        (grouped ncclSend / ncclRecv; bounded by the root's xGMI links - frames that can be loaded per rank,
 *                        sl2_ingest_*, should be)
 *   sl2_gather_states    per-sequence results of every rank's engine on every rank (ncclAllGather): the vehicle state,
 *                        optionally with its covariance block or with the map
 * One communicator per GPU: one process (or host thread) per GPU with sl2_comm_create, or all GPUs of a single process with
 * sl2_comm_create_all.  All calls return SL2_OK or an SL2_ERR_* code (sl2_comm_last_error() has the text); none throws.
 */
#ifndef SCENELIB2_AMD_COMM_H
#define SCENELIB2_AMD_COMM_H
#include "scenelib2_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sl2_comm sl2_comm;

#define SL2_COMM_ID_BYTES 128

/* What a rank contributes per sequence to sl2_gather_states, as doubles:
 *   SL2_GATHER_VEHICLE      xv_ (13)                                              motion_model.cpp:267-288
 *   SL2_GATHER_VEHICLE_PXX  xv_ (13) + Pxx_ row-major (169)                       monoslam.h:170-171
 *   SL2_GATHER_MAP          xv_ (13) + y of every feature SLOT (3 * max_features; slots not in use: zeros) */
#define SL2_GATHER_VEHICLE 0
#define SL2_GATHER_VEHICLE_PXX 1
#define SL2_GATHER_MAP 2

/* Contiguous block partition of `total` sequences over `nranks` (sizes differ by at most one): rank's first sequence and
 * count.  Pure arithmetic (no GPU): the one rule every scatter / gather below uses. */
int sl2_shard_range(int total, int nranks, int rank, int* first, int* count);
/* doubles per sequence of a gather kind for an engine of this capacity; < 0: unknown kind */
int sl2_gather_row_doubles(int what, int max_features);

/* ncclGetUniqueId: rank 0 fills SL2_COMM_ID_BYTES bytes and hands them to the other ranks by whatever means the job has. */
int sl2_comm_unique_id(void* id);
/* ncclCommInitRank on `device`; blocks until all nranks ranks have called it. */
int sl2_comm_create(const void* id, int nranks, int rank, int device, sl2_comm** out);
/* ncclCommInitAll: ndev communicators of ONE process, rank i on devices[i] (NULL: devices 0 .. ndev - 1). */
int sl2_comm_create_all(int ndev, const int* devices, sl2_comm** out);
void sl2_comm_destroy(sl2_comm* c);
int sl2_comm_rank(const sl2_comm* c);
int sl2_comm_nranks(const sl2_comm* c);
int sl2_comm_device(const sl2_comm* c);
const char* sl2_comm_last_error(void);

/* One step's frames, `total_sequences` x frame_bytes device bytes on rank `root` (frames_all; ignored elsewhere), to the owners:
 * rank r receives its sl2_shard_range block into recv (device memory of at least count * frame_bytes bytes).  Asynchronous on
 * `stream` (a hipStream_t of the rank's device; NULL: the default stream).  Every rank of the communicator calls it; within
 * one process over several communicators the calls must be bracketed by sl2_comm_group_begin / _end. */
int sl2_scatter_frames(sl2_comm* c, int root, const uint8_t* frames_all, size_t frame_bytes, int total_sequences, uint8_t* recv,
                       void* stream);
/* The per-sequence rows of engine e (all its sequences; every rank's engine must have the same batch and capacity) gathered on
 * every rank: out = device memory of nranks * batch * sl2_gather_row_doubles doubles, ordered by rank = by global sequence
 * index.  Ordered behind the engine's queued steps; asynchronous on `stream` - synchronise it before reading. */
int sl2_gather_states(sl2_comm* c, sl2_engine* e, int what, double* out, void* stream);
/* ncclGroupStart / ncclGroupEnd: needed around the calls of SEVERAL communicators issued by one host thread. */
int sl2_comm_group_begin(void);
int sl2_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* SCENELIB2_AMD_COMM_H */
