/*
 * gfbe_rccl.h — native all-reduce hook for the landmark-sharded solve (BASELINE configs[2]).
 *
 * libgfbe itself contains no collective: gfbe_set_allreduce (gfbe.h) takes a callback. This small companion library
 * (libgfbe_rccl.so, links librccl) provides that callback in C: an in-place ncclAllReduce(sum, double) on the solver's
 * own HIP stream — no host synchronisation, no interpreter on the critical path of a ~20 us collective. One process per
 * GPU; the 128-byte RCCL unique id is created by rank 0 and handed to the other ranks by whatever the caller uses for
 * rendezvous (bench.py: one torch.distributed broadcast at start-up).
 *
 *   gfbe_rccl *h;  char id[GFBE_RCCL_ID_BYTES];
 *   if (rank == 0) gfbe_rccl_unique_id(id);   ... broadcast id ...
 *   gfbe_rccl_create(&h, id, rank, world, device);
 *   gfbe_set_allreduce(ctx, gfbe_rccl_allreduce, h, rank, world);
 *
 * Replaces nothing in the reference (Ground-Fusion++ is single-process); it is the multi-GPU leg north_star asks for.
 */
#ifndef GFBE_RCCL_H_
#define GFBE_RCCL_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define GFBE_RCCL_ID_BYTES 128
typedef struct gfbe_rccl gfbe_rccl;
/* 0 on success, otherwise the ncclResult_t / hipError_t value (negative: bad argument). */
int32_t gfbe_rccl_unique_id(char id[GFBE_RCCL_ID_BYTES]);
int32_t gfbe_rccl_create(gfbe_rccl **out, const char id[GFBE_RCCL_ID_BYTES], int32_t rank, int32_t world_size, int32_t device);
void gfbe_rccl_destroy(gfbe_rccl *h);
/* Signature of gfbe_allreduce_fn: user = the gfbe_rccl handle. Returns 0 or the ncclResult_t (negative: bad argument) — libgfbe turns
 * a non-zero return into GFBE_DEVICE_ERROR of the solve; the first error also stays readable with gfbe_rccl_last_error. */
int32_t gfbe_rccl_allreduce(void *user, void *device_ptr, int64_t n_doubles, void *hip_stream);
int32_t gfbe_rccl_last_error(const gfbe_rccl *h);
int64_t gfbe_rccl_calls(const gfbe_rccl *h);
/* Bytes handed to ncclAllReduce so far; ncclCommCount of the communicator (-1 without one): what a bench line reports beside its rate. */
int64_t gfbe_rccl_bytes(const gfbe_rccl *h);
int32_t gfbe_rccl_comm_count(const gfbe_rccl *h);
#ifdef __cplusplus
}
#endif
#endif
