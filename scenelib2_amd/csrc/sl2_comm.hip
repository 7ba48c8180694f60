// Host-C++ sharding over RCCL (include/scenelib2_amd_comm.h): independent sequences partitioned over the GPUs of a node,
// communication only at the edges.  Built into libscenelib2_amd_comm.so, which links librccl and the engine
// library; the engine itself never communicates.
#include <rccl/rccl.h>

#include "../../include/scenelib2_amd_comm.h"
#include "sl2_common.hpp"

struct sl2_comm {
  ncclComm_t nccl = nullptr;
  int rank = 0, nranks = 1, device = 0;
  double* pack = nullptr;      // device staging of this rank's rows (grown on demand)
  size_t pack_doubles = 0;
  hipEvent_t ready = nullptr;  // orders the gather behind the engine's stream
};

namespace {

thread_local std::string g_comm_err;

int fail(int code, const std::string& s) { g_comm_err = s; return code; }

#define SL2C_HIP(call)                                                                           \
  do {                                                                                           \
    hipError_t _e = (call);                                                                      \
    if (_e != hipSuccess) return fail(SL2_ERR_HIP, std::string(#call) + " -> " + hipGetErrorString(_e)); \
  } while (0)
#define SL2C_NCCL(call)                                                                          \
  do {                                                                                           \
    ncclResult_t _r = (call);                                                                    \
    if (_r != ncclSuccess) return fail(SL2_ERR_HIP, std::string(#call) + " -> " + ncclGetErrorString(_r)); \
  } while (0)

// rows of one rank: xv, then (kind 1) the 13 x 13 vehicle block row-major, or (kind 2) the three coordinates of every slot
__global__ void __launch_bounds__(256) k_pack_states(const double* __restrict__ x, const double* __restrict__ P,
                                                     const int* __restrict__ f_flags, const int* __restrict__ n_slots, int what, int N,
                                                     int ld, int row, double* __restrict__ out) {
  const int b = blockIdx.x;
  const double* xb = x + (size_t)b * ld;
  double* o = out + (size_t)b * row;
  for (int i = threadIdx.x; i < row; i += blockDim.x) {
    double v;
    if (i < 13) v = xb[i];
    else if (what == SL2_GATHER_VEHICLE_PXX) { const int e = i - 13; v = P[(size_t)b * ld * ld + (size_t)(e / 13) * ld + (e % 13)]; }
    else {
      const int slot = (i - 13) / 3;
      const bool live = slot < n_slots[b] && (f_flags[(size_t)b * N + slot] & sl2::FF_ACTIVE);
      v = live ? xb[i] : 0.0;                 // (feature slot s lives at 13 + 3 s of the total state: the row IS that layout)
    }
    o[i] = v;
  }
}

}  // namespace

extern "C" {

const char* sl2_comm_last_error(void) { return g_comm_err.c_str(); }

int sl2_shard_range(int total, int nranks, int rank, int* first, int* count) {
  if (total < 0 || nranks <= 0 || rank < 0 || rank >= nranks || !first || !count) return fail(SL2_ERR_INVALID, "sl2_shard_range: bad argument");
  const int base = total / nranks, rem = total % nranks;
  *first = rank * base + (rank < rem ? rank : rem);
  *count = base + (rank < rem ? 1 : 0);
  return SL2_OK;
}

int sl2_gather_row_doubles(int what, int max_features) {
  if (what == SL2_GATHER_VEHICLE) return 13;
  if (what == SL2_GATHER_VEHICLE_PXX) return 13 + 169;
  if (what == SL2_GATHER_MAP) return 13 + 3 * max_features;
  return -1;
}

int sl2_comm_unique_id(void* id) {
  if (!id) return fail(SL2_ERR_INVALID, "sl2_comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) <= SL2_COMM_ID_BYTES, "ncclUniqueId larger than SL2_COMM_ID_BYTES");
  ncclUniqueId u;
  SL2C_NCCL(ncclGetUniqueId(&u));
  memset(id, 0, SL2_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return SL2_OK;
}

static int finish_create(sl2_comm* c) {
  SL2C_HIP(hipSetDevice(c->device));
  SL2C_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  return SL2_OK;
}

int sl2_comm_create(const void* id, int nranks, int rank, int device, sl2_comm** out) {
  if (!id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return fail(SL2_ERR_INVALID, "sl2_comm_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SL2_ERR_NO_DEVICE, "no HIP device visible");
  SL2C_HIP(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  sl2_comm* c = new sl2_comm();
  c->rank = rank; c->nranks = nranks; c->device = device;
  ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, u, rank);
  if (r != ncclSuccess) { delete c; return fail(SL2_ERR_HIP, std::string("ncclCommInitRank -> ") + ncclGetErrorString(r)); }
  int rc = finish_create(c);
  if (rc != SL2_OK) { sl2_comm_destroy(c); return rc; }
  *out = c;
  return SL2_OK;
}

int sl2_comm_create_all(int ndev, const int* devices, sl2_comm** out) {
  if (ndev <= 0 || !out) return fail(SL2_ERR_INVALID, "sl2_comm_create_all: bad argument");
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return fail(SL2_ERR_NO_DEVICE, "no HIP device visible");
  std::vector<int> devs(ndev);
  for (int i = 0; i < ndev; ++i) {
    devs[i] = devices ? devices[i] : i;
    if (devs[i] < 0 || devs[i] >= have) return fail(SL2_ERR_INVALID, "sl2_comm_create_all: fewer HIP devices than ranks");
  }
  std::vector<ncclComm_t> comms(ndev);
  SL2C_NCCL(ncclCommInitAll(comms.data(), ndev, devs.data()));
  for (int i = 0; i < ndev; ++i) {
    sl2_comm* c = new sl2_comm();
    c->nccl = comms[i]; c->rank = i; c->nranks = ndev; c->device = devs[i];
    out[i] = c;
    int rc = finish_create(c);
    if (rc != SL2_OK) return rc;
  }
  return SL2_OK;
}

void sl2_comm_destroy(sl2_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->pack) hipFree(c->pack);
  if (c->ready) hipEventDestroy(c->ready);
  if (c->nccl) ncclCommDestroy(c->nccl);
  delete c;
}

int sl2_comm_rank(const sl2_comm* c) { return c ? c->rank : -1; }
int sl2_comm_nranks(const sl2_comm* c) { return c ? c->nranks : 0; }
int sl2_comm_device(const sl2_comm* c) { return c ? c->device : -1; }

int sl2_comm_group_begin(void) { SL2C_NCCL(ncclGroupStart()); return SL2_OK; }
int sl2_comm_group_end(void) { SL2C_NCCL(ncclGroupEnd()); return SL2_OK; }

int sl2_scatter_frames(sl2_comm* c, int root, const uint8_t* frames_all, size_t frame_bytes, int total_sequences, uint8_t* recv,
                       void* stream) {
  if (!c || root < 0 || root >= c->nranks || frame_bytes == 0 || total_sequences < 0 || !recv || (c->rank == root && !frames_all))
    return fail(SL2_ERR_INVALID, "sl2_scatter_frames: bad argument");
  SL2C_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  int first = 0, count = 0;
  sl2_shard_range(total_sequences, c->nranks, c->rank, &first, &count);
  // one group: the root's sends to every other rank and every rank's receive; the root's own block is a device copy
  SL2C_NCCL(ncclGroupStart());
  if (c->rank == root) {
    for (int r = 0; r < c->nranks; ++r) {
      int f = 0, n = 0;
      sl2_shard_range(total_sequences, c->nranks, r, &f, &n);
      if (n == 0) continue;
      if (r == root) { if (hipMemcpyAsync(recv, frames_all + (size_t)f * frame_bytes, (size_t)n * frame_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { ncclGroupEnd(); return fail(SL2_ERR_HIP, "sl2_scatter_frames: device copy of the root's own block failed"); } }
      else { ncclResult_t q = ncclSend(frames_all + (size_t)f * frame_bytes, (size_t)n * frame_bytes, ncclUint8, r, c->nccl, st); if (q != ncclSuccess) { ncclGroupEnd(); return fail(SL2_ERR_HIP, std::string("ncclSend -> ") + ncclGetErrorString(q)); } }
    }
  } else if (count > 0) {
    ncclResult_t q = ncclRecv(recv, (size_t)count * frame_bytes, ncclUint8, root, c->nccl, st);
    if (q != ncclSuccess) { ncclGroupEnd(); return fail(SL2_ERR_HIP, std::string("ncclRecv -> ") + ncclGetErrorString(q)); }
  }
  SL2C_NCCL(ncclGroupEnd());
  return SL2_OK;
}

int sl2_gather_states(sl2_comm* c, sl2_engine* e, int what, double* out, void* stream) {
  if (!c || !e || !out) return fail(SL2_ERR_INVALID, "sl2_gather_states: null argument");
  const int row = sl2_gather_row_doubles(what, e->N);
  if (row < 0) return fail(SL2_ERR_INVALID, "sl2_gather_states: unknown kind");
  if (e->device != c->device) return fail(SL2_ERR_INVALID, "sl2_gather_states: the engine lives on another device than the communicator");
  SL2C_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const size_t need = (size_t)e->B * row;
  if (need > c->pack_doubles) {
    if (c->pack) { SL2C_HIP(hipFree(c->pack)); c->pack = nullptr; c->pack_doubles = 0; }
    SL2C_HIP(hipMalloc((void**)&c->pack, sizeof(double) * need));
    c->pack_doubles = need;
  }
  // behind everything the engine has queued (its root stream is joined with the sequence groups' streams after every call)
  SL2C_HIP(hipEventRecord(c->ready, e->stream));
  SL2C_HIP(hipStreamWaitEvent(st, c->ready, 0));
  hipLaunchKernelGGL(k_pack_states, dim3(e->B), dim3(256), 0, st, e->x, e->P, e->f_flags, e->n_slots, what, e->N, e->ld, row, c->pack);
  SL2C_HIP(hipGetLastError());
  SL2C_NCCL(ncclAllGather(c->pack, out, need, ncclDouble, c->nccl, st));
  return SL2_OK;
}

}  // extern "C"
