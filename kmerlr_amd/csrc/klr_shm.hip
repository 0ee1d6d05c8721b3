// klr_shm.hip -- host-staged collectives over ONE shared memory region, for ranks that are processes of one host:
// several ranks on one GPU (the two-rank tests of a one-GPU box) or a node without a device interconnect.  The same
// three hooks RCCL serves (klr_ctx_init_rccl) -- all-reduce, reduce-scatter, all-gather -- with the semantics of
// include/kmerlr.h; sums are taken in rank order by every rank, so all ranks hold bitwise the same result.
//
// The region is plain memory the launcher maps MAP_SHARED before it starts the ranks (kmerlr --ranks=N
// --transport=shm maps it anonymously before fork; unrelated processes may shm_open it):
//
//   [ header 4 KiB: magic | nranks | slot bytes | barrier {arrived, generation} | abort flag ] [ nranks slots ]
//
// A collective moves the payload in chunks of one slot: device -> own slot, barrier, every rank adds / copies what it
// needs from all slots, barrier, result -> device.  The barrier is a central counter with a generation word and gives
// up after KLR_SHM_TIMEOUT_S seconds (default 120) -- a rank that died must not hang the others.
#include "klr_internal.h"

#include <atomic>
#include <chrono>
#include <sched.h>

namespace {

const u64 SHM_MAGIC = 0x4b4c5253484d3031ull;      // "KLRSHM01"
const size_t SHM_HDR = 4096;
const size_t SHM_SLOT = (size_t)2 << 20;           // bytes per rank and chunk

struct ShmHdr {
  u64 magic;
  u32 nranks, pad;
  u64 slot_bytes;
  std::atomic<u32> arrived;
  std::atomic<u32> generation;
  std::atomic<u32> abort_flag;
};

struct ShmState {
  ShmHdr* hdr = nullptr;
  char*   slots = nullptr;
  int     rank = 0, nranks = 1;
  int     device = 0;
  double  timeout_s = 120.0;
  std::vector<char> tmp;          // host result of a chunk
};

bool shm_barrier(ShmState* st) {
  ShmHdr* h = st->hdr;
  if (h->abort_flag.load(std::memory_order_acquire)) return false;
  const u32 gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (u32)st->nranks) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  u32 spins = 0;
  while (h->generation.load(std::memory_order_acquire) == gen) {
    if (h->abort_flag.load(std::memory_order_acquire)) return false;
    if (++spins > 64) { sched_yield(); spins = 0;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > st->timeout_s) {
        h->abort_flag.store(1, std::memory_order_release);        // the others give up too
        return false;
      }
    }
  }
  return true;
}

inline char* slot_of(ShmState* st, int r) { return st->slots + (size_t)r * st->hdr->slot_bytes; }

template <typename T>
void add_slots(ShmState* st, T* dst, size_t off_elems, size_t count) {
  for (size_t i = 0; i < count; i++) dst[i] = ((const T*)slot_of(st, 0))[off_elems + i];
  for (int r = 1; r < st->nranks; r++) {
    const T* src = (const T*)slot_of(st, r) + off_elems;
    for (size_t i = 0; i < count; i++) dst[i] += src[i];        // rank order: every rank forms the same sum
  }
}

int shm_allreduce(void* user, void* dev, int64_t count, int dtype, void* stream) {
  ShmState* st = (ShmState*)user;
  if (hipSetDevice(st->device) != hipSuccess) return 1;
  const size_t per = st->hdr->slot_bytes / 8;
  st->tmp.resize(st->hdr->slot_bytes);
  for (int64_t o = 0; o < count; o += (int64_t)per) {
    const size_t c = (size_t)std::min<int64_t>((int64_t)per, count - o);
    if (hipMemcpyAsync(slot_of(st, st->rank), (char*)dev + o * 8, c * 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (!shm_barrier(st)) return 2;
    if (dtype == KLR_I64) add_slots<i64>(st, (i64*)st->tmp.data(), 0, c); else add_slots<double>(st, (double*)st->tmp.data(), 0, c);
    if (!shm_barrier(st)) return 2;                              // every rank has read the slots
    if (hipMemcpyAsync((char*)dev + o * 8, st->tmp.data(), c * 8, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;    // tmp is reused by the next chunk
  }
  return 0;
}

// recv[i] = sum over ranks of send[rank * recv_count + i]
int shm_reduce_scatter(void* user, const void* send, void* recv, int64_t recv_count, int dtype, void* stream) {
  ShmState* st = (ShmState*)user;
  if (hipSetDevice(st->device) != hipSuccess) return 1;
  const size_t per = st->hdr->slot_bytes / 8;
  st->tmp.resize(st->hdr->slot_bytes);
  // chunk c of EVERY rank's slice travels together: slot r holds the sender's elements [q * recv_count + o, + len) for
  // the one destination q being served -- R rounds per chunk keep the slot small and the code plain
  for (int64_t o = 0; o < recv_count; o += (int64_t)per) {
    const size_t c = (size_t)std::min<int64_t>((int64_t)per, recv_count - o);
    for (int q = 0; q < st->nranks; q++) {
      const char* src = (const char*)send + ((int64_t)q * recv_count + o) * 8;
      if (hipMemcpyAsync(slot_of(st, st->rank), src, c * 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
      if (!shm_barrier(st)) return 2;
      if (q == st->rank) {
        if (dtype == KLR_I64) add_slots<i64>(st, (i64*)st->tmp.data(), 0, c); else add_slots<double>(st, (double*)st->tmp.data(), 0, c);
      }
      if (!shm_barrier(st)) return 2;
      if (q == st->rank) {
        if (hipMemcpyAsync((char*)recv + o * 8, st->tmp.data(), c * 8, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
      }
    }
  }
  return 0;
}

// recv[r * send_bytes + b] = rank r's send[b]
int shm_allgather(void* user, const void* send, void* recv, int64_t send_bytes, void* stream) {
  ShmState* st = (ShmState*)user;
  if (hipSetDevice(st->device) != hipSuccess) return 1;
  const size_t per = st->hdr->slot_bytes;
  for (int64_t o = 0; o < send_bytes; o += (int64_t)per) {
    const size_t c = (size_t)std::min<int64_t>((int64_t)per, send_bytes - o);
    if (hipMemcpyAsync(slot_of(st, st->rank), (const char*)send + o, c, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (!shm_barrier(st)) return 2;
    for (int r = 0; r < st->nranks; r++)
      if (hipMemcpyAsync((char*)recv + (int64_t)r * send_bytes + o, slot_of(st, r), c, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (!shm_barrier(st)) return 2;
  }
  return 0;
}

}  // namespace

extern "C" int64_t klr_shm_region_bytes(int nranks) {
  if (nranks < 1) return 0;
  return (int64_t)(SHM_HDR + (size_t)nranks * SHM_SLOT);
}

extern "C" int klr_shm_region_init(void* region, int64_t bytes, int nranks) {
  if (!region || nranks < 1 || bytes < klr_shm_region_bytes(nranks)) return klr_fail(nullptr, KLR_EINVAL, "klr_shm_region_init: region too small for %d ranks", nranks);
  ShmHdr* h = (ShmHdr*)region;
  memset(region, 0, SHM_HDR);
  h->nranks = (u32)nranks;
  h->slot_bytes = SHM_SLOT;
  new (&h->arrived) std::atomic<u32>(0);
  new (&h->generation) std::atomic<u32>(0);
  new (&h->abort_flag) std::atomic<u32>(0);
  h->magic = SHM_MAGIC;
  return KLR_OK;
}

// a launcher that saw a rank die releases the ranks waiting at a barrier of the region
extern "C" void klr_shm_region_abort(void* region) {
  if (region && ((ShmHdr*)region)->magic == SHM_MAGIC) ((ShmHdr*)region)->abort_flag.store(1, std::memory_order_release);
}

extern "C" int klr_ctx_init_shm(klr_ctx* ctx, void* region, int64_t bytes, int rank, int nranks) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, region != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
  ShmHdr* h = (ShmHdr*)region;
  if (h->magic != SHM_MAGIC || (int)h->nranks != nranks || bytes < klr_shm_region_bytes(nranks))
    return klr_fail(ctx, KLR_ECOMM, "shared region was not initialised for %d ranks (klr_shm_region_init)", nranks);
  ShmState* st = new ShmState();
  st->hdr = h; st->slots = (char*)region + SHM_HDR; st->rank = rank; st->nranks = nranks; st->device = ctx->device;
  if (const char* t = ctx_opt(ctx, "KLR_SHM_TIMEOUT_S")) st->timeout_s = std::max(1.0, atof(t));
  if (ctx->shm) delete (ShmState*)ctx->shm;
  ctx->shm = st;
  ctx->ar_fn = shm_allreduce; ctx->ar_user = st;
  ctx->rs_fn = shm_reduce_scatter; ctx->ag_fn = shm_allgather; ctx->coll_user = st;
  ctx->rank = rank; ctx->nranks = nranks;
  return klr_ctx_comm_selfcheck(ctx);
}

void shm_release(klr_ctx* ctx) {
  if (ctx->shm) { delete (ShmState*)ctx->shm; ctx->shm = nullptr; }
}

// Sums a HOST buffer over the ranks through whatever hook is installed (staged through device scratch): what a
// launcher needs to collect per-row results of the shards -- each rank fills its own rows, zeros elsewhere.
extern "C" int klr_allreduce_host(klr_ctx* ctx, void* host_buf, int64_t count, int dtype) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, (host_buf != nullptr || count == 0) && count >= 0 && (dtype == KLR_F64 || dtype == KLR_I64));
  if (!ctx->ar_fn || count == 0) return KLR_OK;
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  void* d;
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)count * 8, &d));
  KLR_TRY(h2d(ctx, d, host_buf, (size_t)count * 8));
  KLR_TRY(klr_allreduce(ctx, d, count, dtype));
  return d2h_sync(ctx, host_buf, d, (size_t)count * 8);
}
