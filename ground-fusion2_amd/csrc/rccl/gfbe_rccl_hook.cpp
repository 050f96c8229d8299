// gfbe_rccl_hook.cpp — libgfbe_rccl.so: the gfbe_allreduce_fn of include/gfbe.h over RCCL (include/gfbe_rccl.h).
// In-place sum all-reduce of the solver's partial normal equations on the solver's own stream (xGMI between the GPUs of
// a node). Kept out of libgfbe.so so that the single-GPU product does not link librccl.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "../../../include/gfbe_rccl.h"

struct gfbe_rccl {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  int32_t err = 0;
  int64_t calls = 0, bytes = 0;
};

extern "C" {

int32_t gfbe_rccl_unique_id(char id[GFBE_RCCL_ID_BYTES]) {
  if (!id) return -1;
  static_assert(sizeof(ncclUniqueId) == GFBE_RCCL_ID_BYTES, "RCCL unique id size");
  ncclUniqueId u;
  const ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return (int32_t)r;
  std::memcpy(id, &u, sizeof u);
  return 0;
}

int32_t gfbe_rccl_create(gfbe_rccl **out, const char id[GFBE_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world || device < 0) return -1;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device >= n) { (void)hipGetLastError(); return -2; }   // no such GPU: nothing to reduce on
  if (hipSetDevice(device) != hipSuccess) return -3;
  gfbe_rccl *h = new (std::nothrow) gfbe_rccl();
  if (!h) return -4;
  h->rank = rank; h->world = world; h->device = device;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  const ncclResult_t r = ncclCommInitRank(&h->comm, world, u, rank);
  if (r != ncclSuccess) { delete h; return (int32_t)r; }
  *out = h;
  return 0;
}

void gfbe_rccl_destroy(gfbe_rccl *h) {
  if (!h) return;
  if (h->comm) (void)ncclCommDestroy(h->comm);
  delete h;
}

int32_t gfbe_rccl_allreduce(void *user, void *device_ptr, int64_t n_doubles, void *hip_stream) {
  gfbe_rccl *h = (gfbe_rccl *)user;
  if (!h || !h->comm || !device_ptr || n_doubles <= 0) { if (h && h->err == 0) h->err = -1; return -1; }
  const ncclResult_t r = ncclAllReduce(device_ptr, device_ptr, (size_t)n_doubles, ncclDouble, ncclSum, h->comm, (hipStream_t)hip_stream);
  if (r != ncclSuccess && h->err == 0) h->err = (int32_t)r;
  h->calls++;
  h->bytes += 8 * n_doubles;
  return r == ncclSuccess ? 0 : (int32_t)r;
}

int32_t gfbe_rccl_last_error(const gfbe_rccl *h) { return h ? h->err : -1; }
int64_t gfbe_rccl_calls(const gfbe_rccl *h) { return h ? h->calls : 0; }
int64_t gfbe_rccl_bytes(const gfbe_rccl *h) { return h ? h->bytes : 0; }
int32_t gfbe_rccl_comm_count(const gfbe_rccl *h) {
  int n = -1;
  if (!h || !h->comm || ncclCommCount(h->comm, &n) != ncclSuccess) return -1;
  return n;
}

}  // extern "C"
