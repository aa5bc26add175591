/* fake_rccl.c -- TEST INFRASTRUCTURE, never shipped: the eight RCCL entry points csrc/comm.hip binds
 * (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclBroadcast, ncclGroupStart,
 * ncclGroupEnd, ncclGetErrorString), implemented for HOST buffers between forked / spawned processes over
 * one POSIX shared-memory segment.  It exists so that the multi-rank code of comm.hip -- the in-place
 * all-gather of equal shards, the grouped per-rank broadcasts of ragged shards, the 64 grouped stereo
 * pieces -- executes with world > 1 on a machine without GPUs (tests/test_comm_multirank.py, loaded
 * through MDCT_RCCL_LIB).  Prototypes come from the real <rccl/rccl.h>, so an ABI drift is a compile
 * error here, not a silent mismatch.
 *
 * Semantics kept from RCCL: collectives are matched by issue order; inside ncclGroupStart/End they are
 * queued and run at the outermost ncclGroupEnd; in-place forms (sendbuff inside recvbuff) are legal.
 * Not kept: streams (calls complete synchronously, `stream` is ignored) and device memory.
 *
 * Exchange: every collective is cut into chunks of at most SLOT bytes per rank; per chunk each
 * contributing rank copies its bytes into its slot of the shared staging area, all ranks meet at a
 * barrier, every rank copies what it needs out, and a second barrier frees the slots. */
#define _GNU_SOURCE
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { FAKE_MAX_RANKS = 64, FAKE_SLOT = 1 << 20, FAKE_MAX_QUEUE = 4096 };
#define FAKE_TIMEOUT_S 120.0

typedef struct
{
  _Atomic int arrived;    /* ranks that have attached */
  _Atomic int count;      /* barrier: arrivals of the current generation */
  _Atomic int generation; /* barrier: generation number */
  _Atomic int failed;     /* a rank timed out or saw an argument mismatch: everybody bails out */
  int world;
  /* per-chunk argument cross-check: every rank publishes what it thinks the collective is */
  size_t op_bytes[FAKE_MAX_RANKS];
  int op_kind[FAKE_MAX_RANKS];
  int op_root[FAKE_MAX_RANKS];
  unsigned char staging[];
} shared_t;

struct ncclComm
{
  shared_t *sh;
  size_t map_bytes;
  int rank, world;
  char name[64];
};

typedef struct
{
  int kind; /* 0 all-gather, 1 broadcast */
  const void *send;
  void *recv;
  size_t bytes; /* per rank (all-gather) or total (broadcast) */
  int root;
  struct ncclComm *comm;
} op_t;

static __thread int g_depth;
static __thread int g_nq;
static __thread op_t g_q[FAKE_MAX_QUEUE];
static __thread long g_ops_run, g_allgathers, g_broadcasts, g_groups;

static double now_s(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static size_t type_bytes(ncclDataType_t t)
{
  switch (t)
  {
  case ncclInt8: case ncclUint8: return 1;
  case ncclFloat16: case ncclBfloat16: return 2;
  case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
  case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
  default: return 0;
  }
}

/* sense-free generation barrier with a deadline; returns 0, or -1 when the group has failed */
static int barrier(struct ncclComm *c)
{
  shared_t *sh = c->sh;
  const int gen = atomic_load(&sh->generation);
  if (atomic_fetch_add(&sh->count, 1) + 1 == sh->world)
  {
    atomic_store(&sh->count, 0);
    atomic_fetch_add(&sh->generation, 1);
    return atomic_load(&sh->failed) ? -1 : 0;
  }
  const double t0 = now_s();
  while (atomic_load(&sh->generation) == gen)
  {
    if (atomic_load(&sh->failed))
      return -1;
    if (now_s() - t0 > FAKE_TIMEOUT_S)
    {
      atomic_store(&sh->failed, 1);
      return -1;
    }
    usleep(50);
  }
  return atomic_load(&sh->failed) ? -1 : 0;
}

static ncclResult_t run_op(const op_t *o)
{
  struct ncclComm *c = o->comm;
  shared_t *sh = c->sh;
  const int W = c->world, me = c->rank;
  /* every rank must have issued the same collective: publish, meet, compare */
  sh->op_bytes[me] = o->bytes;
  sh->op_kind[me] = o->kind;
  sh->op_root[me] = o->root;
  if (barrier(c))
    return ncclSystemError;
  for (int r = 0; r < W; r++)
    if (sh->op_bytes[r] != o->bytes || sh->op_kind[r] != o->kind || sh->op_root[r] != o->root)
    {
      atomic_store(&sh->failed, 1);
      return ncclInvalidUsage;
    }
  for (size_t done = 0; done < o->bytes; done += FAKE_SLOT)
  {
    const size_t n = o->bytes - done < FAKE_SLOT ? o->bytes - done : FAKE_SLOT;
    if (o->kind == 0)
      memcpy(sh->staging + (size_t)me * FAKE_SLOT, (const unsigned char *)o->send + done, n);
    else if (me == o->root)
      memcpy(sh->staging, (const unsigned char *)o->send + done, n);
    if (barrier(c))
      return ncclSystemError;
    if (o->kind == 0)
    { /* recv = [rank 0's bytes][rank 1's bytes]...; the piece of rank r starts at r * bytes */
      for (int r = 0; r < W; r++)
        memcpy((unsigned char *)o->recv + (size_t)r * o->bytes + done, sh->staging + (size_t)r * FAKE_SLOT, n);
    }
    else if (me != o->root || o->recv != o->send)
      memcpy((unsigned char *)o->recv + done, sh->staging, n);
    if (barrier(c))
      return ncclSystemError;
  }
  g_ops_run++;
  if (o->kind == 0)
    g_allgathers++;
  else
    g_broadcasts++;
  return ncclSuccess;
}

static ncclResult_t submit(const op_t *o)
{
  if (g_depth > 0)
  {
    if (g_nq >= FAKE_MAX_QUEUE)
      return ncclInternalError;
    g_q[g_nq++] = *o;
    return ncclSuccess;
  }
  return run_op(o);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  static _Atomic int serial;
  if (!id)
    return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/mdct_fake_rccl_%ld_%d_%ld", (long)getpid(), atomic_fetch_add(&serial, 1), (long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
  if (!comm || nranks <= 0 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/')
    return ncclInvalidArgument;
  struct ncclComm *c = calloc(1, sizeof *c);
  if (!c)
    return ncclSystemError;
  c->rank = rank;
  c->world = nranks;
  memcpy(c->name, id.internal, sizeof c->name - 1);
  c->map_bytes = sizeof(shared_t) + (size_t)nranks * FAKE_SLOT;
  /* whoever comes first creates the segment (O_EXCL); everyone else opens it and waits for its size */
  int fd = shm_open(c->name, O_RDWR | O_CREAT | O_EXCL, 0600);
  const int creator = fd >= 0;
  if (creator)
  {
    if (ftruncate(fd, (off_t)c->map_bytes) != 0)
    {
      close(fd);
      shm_unlink(c->name);
      free(c);
      return ncclSystemError;
    }
  }
  else
  {
    const double t0 = now_s();
    for (;;)
    {
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes)
        break;
      if (fd >= 0)
        close(fd);
      if (now_s() - t0 > FAKE_TIMEOUT_S)
      {
        free(c);
        return ncclSystemError;
      }
      usleep(200);
    }
  }
  c->sh = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED)
  {
    free(c);
    return ncclSystemError;
  }
  if (creator)
    c->sh->world = nranks; /* the segment is zero-filled; `arrived` counts attachments */
  atomic_fetch_add(&c->sh->arrived, 1);
  const double t0 = now_s();
  while (atomic_load(&c->sh->arrived) < nranks || c->sh->world != nranks)
  { /* collective construction, like the real call */
    if (now_s() - t0 > FAKE_TIMEOUT_S)
    {
      munmap(c->sh, c->map_bytes);
      shm_unlink(c->name);
      free(c);
      return ncclSystemError;
    }
    usleep(200);
  }
  if (barrier(c))
    return ncclSystemError;
  if (rank == 0)
    shm_unlink(c->name); /* everybody holds a mapping now: the name can go */
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  if (!comm)
    return ncclSuccess;
  munmap(comm->sh, comm->map_bytes);
  free(comm);
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
  (void)stream;
  if (!comm || !type_bytes(datatype) || (sendcount && (!sendbuff || !recvbuff)))
    return ncclInvalidArgument;
  const op_t o = {0, sendbuff, recvbuff, sendcount * type_bytes(datatype), -1, comm};
  return submit(&o);
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
  (void)stream;
  if (!comm || !type_bytes(datatype) || root < 0 || root >= comm->world || (count && (!sendbuff || !recvbuff)))
    return ncclInvalidArgument;
  const op_t o = {1, sendbuff, recvbuff, count * type_bytes(datatype), root, comm};
  return submit(&o);
}

ncclResult_t ncclGroupStart(void)
{
  g_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
  if (g_depth <= 0)
    return ncclInvalidUsage;
  if (--g_depth > 0)
    return ncclSuccess;
  ncclResult_t e = ncclSuccess;
  for (int i = 0; i < g_nq && e == ncclSuccess; i++)
    e = run_op(&g_q[i]);
  g_nq = 0;
  g_groups++;
  return e;
}

const char *ncclGetErrorString(ncclResult_t result)
{
  switch (result)
  {
  case ncclSuccess: return "fake-rccl: success";
  case ncclSystemError: return "fake-rccl: a rank timed out or the group failed";
  case ncclInvalidArgument: return "fake-rccl: invalid argument";
  case ncclInvalidUsage: return "fake-rccl: ranks disagree on a collective (kind, bytes or root)";
  default: return "fake-rccl: internal error";
  }
}

/* what the calling thread has executed so far -- lets a test assert WHICH branch of comm.hip ran */
void fake_rccl_stats(long *collectives, long *allgathers, long *broadcasts, long *groups)
{
  *collectives = g_ops_run;
  *allgathers = g_allgathers;
  *broadcasts = g_broadcasts;
  *groups = g_groups;
}
