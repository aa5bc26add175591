// comm.hip -- the multi-GPU leg of the C-ABI: one process per GPU, RCCL over xGMI.
//
// The reference has no communication of any kind; its only parallelism hook is the row range
// startY/endY (simd_dct.cpp:2245-2255).  north_star shards images by block rows across GPUs and
// all-gathers the coefficients for the whole-node run: every rank transforms its block-row
// shard IN PLACE in a full-size output buffer (any entry point of mdct.h with [by0, by1)), then
// one call here makes every rank's buffer complete.
//
// RCCL is bound at run time (dlopen of librccl.so.1; a copy already loaded by the process,
// e.g. PyTorch's, is reused), so single-GPU users of libmdct_hip.so never load it.
// MDCT_RCCL_LIB=<path> names the library to bind instead: a site's own RCCL build, or the
// shared-memory stand-in tests/fake_rccl.c with which the multi-rank paths below (in-place
// all-gather, ragged broadcasts, the 64 grouped stereo pieces) run with world > 1 on host
// buffers on a machine without GPUs (tests/test_comm_multirank.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mdct.h"

extern "C" __attribute__((visibility("hidden"))) int mdct_set_error(int code, const char *fmt, ...); // mdct_api.hip

namespace
{

struct Rccl
{
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  char why[256] = "symbols missing"; // dlerror() text of the failed load, captured once inside the call_once
};

Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl *rccl()
{
  std::call_once(g_rccl_once, [] {
    Rccl &r = g_rccl;
    const char *override_lib = getenv("MDCT_RCCL_LIB");
    if (override_lib && *override_lib)
      r.handle = dlopen(override_lib, RTLD_NOW | RTLD_LOCAL);
    else
    {
      r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); // the copy the process already uses, if any
      if (!r.handle)
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!r.handle)
        r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle)
    {
      const char *e = dlerror(); // one call: a second one returns NULL
      snprintf(r.why, sizeof r.why, "%s", e ? e : "dlopen failed");
      return;
    }
#define MDCT_SYM(name) r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.handle, "nccl" #name))
    MDCT_SYM(GetUniqueId);
    MDCT_SYM(CommInitRank);
    MDCT_SYM(CommDestroy);
    MDCT_SYM(AllGather);
    MDCT_SYM(Broadcast);
    MDCT_SYM(GroupStart);
    MDCT_SYM(GroupEnd);
    MDCT_SYM(GetErrorString);
#undef MDCT_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Broadcast && r.GroupStart && r.GroupEnd && r.GetErrorString;
  });
  return g_rccl.ok ? &g_rccl : nullptr;
}

int no_rccl() { return mdct_set_error(MDCT_NOT_SUPPORTED, "RCCL (librccl.so.1 or MDCT_RCCL_LIB) could not be bound: %s", g_rccl.why); }

int nccl_fail(const Rccl *r, ncclResult_t e, const char *what) { return mdct_set_error(MDCT_NOT_SUPPORTED, "%s: %s", what, r->GetErrorString(e)); }

} // namespace

struct mdct_comm
{
  ncclComm_t comm;
  int rank, world;
};

extern "C" {

void mdct_shard_rows(size_t n_rows, int world, int rank, size_t *b0, size_t *b1)
{
  if (world <= 0 || rank < 0 || rank >= world)
  {
    *b0 = *b1 = 0;
    return;
  }
  const size_t base = n_rows / (size_t)world, extra = n_rows % (size_t)world, r = (size_t)rank;
  *b0 = r * base + (r < extra ? r : extra);
  *b1 = *b0 + base + (r < extra ? 1 : 0);
}

int mdct_stereo_shard_piece(size_t sizeX, size_t sizeY, int world, int rank, size_t *first_offset, size_t *plane_stride, size_t *piece_bytes)
{
  if (!first_offset || !plane_stride || !piece_bytes || world <= 0 || rank < 0 || rank >= world)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "bad rank %d of %d or null output", rank, world);
  if (sizeX % 16 != 0 || sizeY % 16 != 0)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "stereo planes are multiples of 16x16");
  size_t b0, b1;
  mdct_shard_rows(sizeY / 16, world, rank, &b0, &b1);
  const size_t bpr = sizeX / 8;
  *plane_stride = sizeX * sizeY / 64;  // simd_dct.cpp:259-264: 64 coefficient planes
  *first_offset = b0 * 2 * bpr;        // :284-294: blocks ordered (block row, eye, block x)
  *piece_bytes = (b1 - b0) * 2 * bpr;
  return MDCT_SUCCESS;
}

int mdct_comm_get_unique_id(void *id128)
{
  static_assert(sizeof(ncclUniqueId) == MDCT_UNIQUE_ID_BYTES, "mdct.h promises a 128-byte id");
  if (!id128)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null id buffer");
  const Rccl *r = rccl();
  if (!r)
    return no_rccl();
  const ncclResult_t e = r->GetUniqueId(static_cast<ncclUniqueId *>(id128));
  return e == ncclSuccess ? MDCT_SUCCESS : nccl_fail(r, e, "ncclGetUniqueId");
}

int mdct_comm_init(mdct_comm **comm, int rank, int world, const void *id128)
{
  if (!comm || !id128 || world <= 0 || rank < 0 || rank >= world)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "bad rank %d of %d or null argument", rank, world);
  const Rccl *r = rccl();
  if (!r)
    return no_rccl();
  ncclUniqueId id = *static_cast<const ncclUniqueId *>(id128);
  ncclComm_t c;
  const ncclResult_t e = r->CommInitRank(&c, world, id, rank); // collective; binds the calling thread's current HIP device
  if (e != ncclSuccess)
    return nccl_fail(r, e, "ncclCommInitRank");
  *comm = new mdct_comm{c, rank, world};
  return MDCT_SUCCESS;
}

int mdct_comm_destroy(mdct_comm *comm)
{
  if (!comm)
    return MDCT_SUCCESS;
  const Rccl *r = rccl();
  if (r)
    (void)r->CommDestroy(comm->comm);
  delete comm;
  return MDCT_SUCCESS;
}

int mdct_comm_rank(const mdct_comm *comm) { return comm ? comm->rank : -1; }
int mdct_comm_world(const mdct_comm *comm) { return comm ? comm->world : 0; }

// pieces[r] = (offset, bytes) of rank r inside every one of n_planes regions that lie plane_stride apart
static int gather_pieces(mdct_comm *comm, uint8_t *buf, size_t n_planes, size_t plane_stride, const size_t *off, const size_t *len, bool equal, void *stream)
{
  const Rccl *r = rccl();
  if (!r)
    return no_rccl();
  hipStream_t s = (hipStream_t)stream;
  ncclResult_t e = ncclSuccess;
  // MDCT_FORCE_RAGGED_GATHER=1 (diagnostics): take the grouped-broadcast form even for equal shards -- on a one-GPU box that is the only way
  // the form ragged shards use ever meets the real RCCL (tests/test_comm.py); same bytes either way
  const char *force = getenv("MDCT_FORCE_RAGGED_GATHER");
  if (force && force[0] == '1')
    equal = false;
  const bool grouped = n_planes > 1 || !equal;
  if (grouped && (e = r->GroupStart()) != ncclSuccess)
    return nccl_fail(r, e, "ncclGroupStart");
  for (size_t p = 0; p < n_planes && e == ncclSuccess; p++)
  {
    uint8_t *base = buf + p * plane_stride;
    if (equal) // shards in rank order and of equal size: one in-place all-gather (sendbuff = recvbuff + rank * count)
      e = r->AllGather(base + off[comm->rank], base + off[0], len[0], ncclUint8, comm->comm, s);
    else       // ragged shards: every rank broadcasts its own piece in place
      for (int q = 0; q < comm->world && e == ncclSuccess; q++)
        if (len[q])
          e = r->Broadcast(base + off[q], base + off[q], len[q], ncclUint8, q, comm->comm, s);
  }
  if (grouped)
  {
    const ncclResult_t e2 = r->GroupEnd();
    if (e == ncclSuccess)
      e = e2;
  }
  return e == ncclSuccess ? MDCT_SUCCESS : nccl_fail(r, e, "RCCL all-gather");
}

int mdct_allgather_rows(mdct_comm *comm, void *buf, size_t row_bytes, size_t n_rows, void *stream)
{
  if (!comm || !buf)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null communicator or buffer");
  if (comm->world > 1024)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "more than 1024 ranks");
  size_t off[1024], len[1024];
  bool equal = true;
  for (int q = 0; q < comm->world; q++)
  {
    size_t b0, b1;
    mdct_shard_rows(n_rows, comm->world, q, &b0, &b1);
    off[q] = b0 * row_bytes;
    len[q] = (b1 - b0) * row_bytes;
    equal = equal && len[q] == len[0];
  }
  if (row_bytes == 0 || n_rows == 0)
    return MDCT_SUCCESS;
  return gather_pieces(comm, static_cast<uint8_t *>(buf), 1, 0, off, len, equal, stream);
}

int mdct_allgather_stereo(mdct_comm *comm, uint8_t *buf, size_t sizeX, size_t sizeY, void *stream)
{
  if (!comm || !buf)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null communicator or buffer");
  if (comm->world > 1024)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "more than 1024 ranks");
  size_t off[1024], len[1024], stride = 0;
  bool equal = true;
  for (int q = 0; q < comm->world; q++)
  {
    const int rc = mdct_stereo_shard_piece(sizeX, sizeY, comm->world, q, &off[q], &stride, &len[q]);
    if (rc)
      return rc;
    equal = equal && len[q] == len[0];
  }
  if (stride == 0)
    return MDCT_SUCCESS;
  // 64 coefficient planes, one strided piece per rank in each: 64 collectives in ONE RCCL group (one launch)
  return gather_pieces(comm, buf, 64, stride, off, len, equal, stream);
}

} // extern "C"
