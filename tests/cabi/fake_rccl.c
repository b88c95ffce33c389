/* fake_rccl.c -- TEST INFRASTRUCTURE ONLY (never shipped, never loaded by the product unless a test points ibs_comm_load at it).
 *
 * A stand-in for librccl that lets the library's own collective code (ibs_comm_init, ibs_comm_allgather_f64, the 16-slot
 * overlapped ibs_comm_allgather_start_f64 / ibs_comm_wait: csrc/ibs_api.hip) run with world > 1 on a box that has ONE GPU.
 * RCCL itself refuses two ranks on one device; the boxes this project is developed on have one GPU, so before this file that
 * code had only ever run with a one-rank communicator.
 *
 * It exports exactly the five symbols the library binds at run time (ibs_api.hip: ibs_comm_load):
 *   ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString
 * with RCCL's calling conventions (ncclUniqueId = 128 bytes passed BY VALUE to ncclCommInitRank).
 *
 * Transport: the ranks are processes that share the device; they meet in a POSIX shared-memory segment named by the unique id.
 * ncclAllGather is asynchronous and stream-ordered like the real one -- nothing happens at call time except enqueueing, on the
 * stream it is handed:
 *   host function   wait until every rank has drained this ring buffer's previous use
 *   D2H copy        send  -> ring[b][rank]                       (the segment is page-locked: the copies are truly asynchronous)
 *   host function   announce arrival, wait for all ranks; optional delay (FAKE_RCCL_DELAY_US) so that a consumer that fails
 *                   to order itself after the gather reads stale data instead of passing by luck
 *   H2D copy        ring[b][0 .. nranks) -> recv
 *   host function   announce "drained"
 * A wait that exceeds FAKE_RCCL_TIMEOUT_S (default 60) marks the segment failed: every later gather delivers NaNs and
 * ncclCommDestroy reports an error, so a test fails on its data instead of hanging the box.
 *
 * Build (tests/test_gpu_round6.py does it):  gcc -O1 -shared -fPIC -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -lrt -lpthread
 * (no link-time dependency on libamdhip64: like libibs_hip.so it binds to the ONE HIP runtime the process already holds.) */
#define __HIP_PLATFORM_AMD__ 1
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 };

#define FK_RING 4                  /* gathers in flight per communicator before a rank waits for the slowest reader */
#define FK_MAXRANKS 8
#define FK_SLOT_BYTES (256 * 1024) /* per rank per gather */
#define FK_HDR_BYTES 4096
#define FK_MAGIC 0x1b5fa4e

typedef struct {
  _Atomic int magic, joined, left, failed;
  _Atomic long arrived[FK_RING];   /* monotone: nranks per use of the buffer */
  _Atomic long drained[FK_RING];
  _Atomic long gathers;            /* completed by rank 0 (diagnostics) */
} fk_header;

typedef struct {
  fk_header* hdr;
  char* data;                      /* [FK_RING][FK_MAXRANKS][FK_SLOT_BYTES] */
  size_t map_bytes;
  int nranks, rank, registered;
  long next_q;                     /* sequence number of the next gather of this rank */
  char name[128];
  double timeout_s;
  long delay_us;
} fk_comm;

typedef struct { fk_comm* c; long q; size_t bytes; } fk_call;

static size_t fk_total_bytes(void) { return FK_HDR_BYTES + (size_t)FK_RING * FK_MAXRANKS * FK_SLOT_BYTES; }
static char* fk_slot(fk_comm* c, int b, int r) { return c->data + ((size_t)b * FK_MAXRANKS + r) * FK_SLOT_BYTES; }
static double fk_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* spin (politely) until *v >= target; false on time-out or when another rank has marked the segment failed */
static int fk_wait(fk_comm* c, _Atomic long* v, long target) {
  const double t0 = fk_now();
  unsigned spins = 0;
  while (atomic_load_explicit(v, memory_order_acquire) < target) {
    if (atomic_load(&c->hdr->failed)) return 0;
    if ((++spins & 63) == 0) {
      if (fk_now() - t0 > c->timeout_s) { atomic_store(&c->hdr->failed, 1); return 0; }
      usleep(20);
    }
  }
  return 1;
}

static void fk_poison(fk_comm* c, int b, size_t bytes) {
  for (int r = 0; r < c->nranks; ++r) {
    double* p = (double*)fk_slot(c, b, r);
    for (size_t i = 0; i < bytes / sizeof(double); ++i) p[i] = NAN;
  }
}

static void fk_pre(void* u) {      /* the buffer's previous use (q - FK_RING) must have been read by every rank */
  fk_call* k = (fk_call*)u;
  const int b = (int)(k->q % FK_RING);
  (void)fk_wait(k->c, &k->c->hdr->drained[b], (long)k->c->nranks * (k->q / FK_RING));
}
static void fk_arrive(void* u) {
  fk_call* k = (fk_call*)u;
  fk_comm* c = k->c;
  const int b = (int)(k->q % FK_RING);
  atomic_fetch_add_explicit(&c->hdr->arrived[b], 1, memory_order_acq_rel);
  if (!fk_wait(c, &c->hdr->arrived[b], (long)c->nranks * (k->q / FK_RING + 1))) fk_poison(c, b, k->bytes);
  if (c->delay_us > 0) usleep((useconds_t)c->delay_us);
}
static void fk_done(void* u) {
  fk_call* k = (fk_call*)u;
  const int b = (int)(k->q % FK_RING);
  atomic_fetch_add_explicit(&k->c->hdr->drained[b], 1, memory_order_acq_rel);
  if (k->c->rank == 0) atomic_fetch_add(&k->c->hdr->gathers, 1);
  free(k);
}

int ncclGetUniqueId(ncclUniqueId* id) {
  static _Atomic int counter;
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/ibs_fake_rccl_%d_%d", (int)getpid(), atomic_fetch_add(&counter, 1));
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)fk_total_bytes()) != 0) { close(fd); shm_unlink(id->internal); return ncclSystemError; }
  fk_header* h = (fk_header*)mmap(NULL, FK_HDR_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (h == MAP_FAILED) { shm_unlink(id->internal); return ncclSystemError; }
  memset(h, 0, FK_HDR_BYTES);
  atomic_store(&h->magic, FK_MAGIC);
  munmap(h, FK_HDR_BYTES);
  return ncclSuccess;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > FK_MAXRANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  fk_comm* c = (fk_comm*)calloc(1, sizeof(fk_comm));
  if (!c) return ncclSystemError;
  id.internal[sizeof(id.internal) - 1] = 0;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
  c->timeout_s = e ? atof(e) : 60.0;
  e = getenv("FAKE_RCCL_DELAY_US");
  c->delay_us = e ? atol(e) : 0;
  int fd = -1;
  for (double t0 = fk_now(); fd < 0 && fk_now() - t0 < c->timeout_s; ) {
    fd = shm_open(c->name, O_RDWR, 0600);
    if (fd < 0) usleep(1000);
  }
  if (fd < 0) { free(c); return ncclSystemError; }
  c->map_bytes = fk_total_bytes();
  void* base = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) { free(c); return ncclSystemError; }
  c->hdr = (fk_header*)base; c->data = (char*)base + FK_HDR_BYTES;
  c->nranks = nranks; c->rank = rank;
  if (atomic_load(&c->hdr->magic) != FK_MAGIC) { munmap(base, c->map_bytes); free(c); return ncclInvalidArgument; }
  /* page-lock the segment: an unpinned H2D copy would read the host buffer when it is ENQUEUED, i.e. before the other ranks
     have written their parts */
  if (hipHostRegister(base, c->map_bytes, hipHostRegisterDefault) != hipSuccess) { munmap(base, c->map_bytes); free(c); return ncclUnhandledCudaError; }
  c->registered = 1;
  /* collective: every rank joins before anyone returns (ncclCommInitRank blocks the same way) */
  atomic_fetch_add(&c->hdr->joined, 1);
  const double t0 = fk_now();
  while (atomic_load(&c->hdr->joined) < nranks) {
    if (fk_now() - t0 > c->timeout_s) { atomic_store(&c->hdr->failed, 1); break; }
    usleep(200);
  }
  *comm = c;
  return atomic_load(&c->hdr->failed) ? ncclSystemError : ncclSuccess;
}

static size_t fk_dtype_bytes(int dt) {
  switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  fk_comm* c = (fk_comm*)comm;
  const size_t bytes = count * fk_dtype_bytes(dtype);
  if (!c || !send || !recv || fk_dtype_bytes(dtype) == 0 || bytes > FK_SLOT_BYTES) return ncclInvalidArgument;
  fk_call* k = (fk_call*)malloc(sizeof(fk_call));
  if (!k) return ncclSystemError;
  k->c = c; k->q = c->next_q++; k->bytes = bytes;
  const int b = (int)(k->q % FK_RING);
  if (hipLaunchHostFunc(stream, fk_pre, k) != hipSuccess) return ncclUnhandledCudaError;
  if (bytes && hipMemcpyAsync(fk_slot(c, b, c->rank), send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, fk_arrive, k) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < c->nranks && bytes; ++r)
    if (hipMemcpyAsync((char*)recv + (size_t)r * bytes, fk_slot(c, b, r), bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, fk_done, k) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
  fk_comm* c = (fk_comm*)comm;
  if (!c) return ncclInvalidArgument;
  const int failed = atomic_load(&c->hdr->failed);
  const int left = atomic_fetch_add(&c->hdr->left, 1) + 1;
  if (c->registered) (void)hipHostUnregister(c->hdr);
  if (left >= c->nranks) shm_unlink(c->name);       /* the last rank out removes the name */
  munmap(c->hdr, c->map_bytes);
  free(c);
  return failed ? ncclSystemError : ncclSuccess;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
    case ncclSystemError: return "fake_rccl: shared segment / rendezvous failed or timed out";
    case ncclInvalidArgument: return "fake_rccl: invalid argument (payload above 256 KB per rank, more than 8 ranks, unknown type)";
    default: return "fake_rccl: internal error";
  }
}

/* diagnostics for the tests: gathers completed on this communicator's segment (rank 0's count) */
long fake_rccl_gathers(void* comm) { return comm ? atomic_load(&((fk_comm*)comm)->hdr->gathers) : -1; }
