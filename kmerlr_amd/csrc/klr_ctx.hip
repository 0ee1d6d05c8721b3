// klr_ctx.hip -- context, error reporting, workspace, all-reduce hook (callback or RCCL),
// HIP-event profiling of the column sweep.
#include "klr_internal.h"

#include <dlfcn.h>

#include <cstdarg>
#include <mutex>
#include <set>

thread_local std::string g_klr_err;

int klr_fail(klr_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_klr_err = buf;
  if (ctx) ctx->err = buf;
  return code;
}

#include "klr_kernel_hash.inc"      // written by kmerlr_amd/build.py: hash of the sweep / ingest sources of this build
extern "C" const char* klr_version(void) { return "kmerlr-hip 0.3 (gfx950) kernels:" KLR_KERNEL_HASH; }

// ---- switches: read from the environment ONCE per context, changed through klr_ctx_set_option -------------------
extern char** environ;

const char* ctx_opt(const klr_ctx* ctx, const char* name) {
  if (!ctx) return nullptr;
  auto it = ctx->opts.find(name);
  return it == ctx->opts.end() ? nullptr : it->second.c_str();
}

void ctx_parse_switches(klr_ctx* ctx) {
  klr_ctx::Switches d;
  auto geti = [&](const char* k, long long def) { const char* v = ctx_opt(ctx, k); return v ? atoll(v) : def; };
  d.q64_lds_kib = (int)geti("KLR_Q64_LDS_KIB", d.q64_lds_kib);
  d.q64_slds_a = (int)geti("KLR_Q64_SLDS_A", d.q64_slds_a);
  d.stream_nt = (int)geti("KLR_STREAM_NT", d.stream_nt);
  d.finish_iter = (int)geti("KLR_FINISH_ITER", 0);
  d.finish_no_flat = ctx_opt(ctx, "KLR_FINISH_NO_FLAT") != nullptr;
  d.rs_min_dim = geti("KLR_RS_MIN_DIM", d.rs_min_dim);
  d.rs_force = ctx_opt(ctx, "KLR_RS_FORCE") != nullptr;
  d.rs = (int)geti("KLR_RS", -1);
  d.overlap = (int)geti("KLR_SWEEP_RANGES", -1);
  ctx->sw = d;
}

static void ctx_load_env(klr_ctx* ctx) {
  ctx->opts.clear();
  for (char** e = environ; e && *e; e++) {
    if (strncmp(*e, "KLR_", 4) != 0) continue;
    const char* eq = strchr(*e, '=');
    if (!eq) continue;
    ctx->opts[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
  }
  ctx_parse_switches(ctx);
}

extern "C" int klr_ctx_set_option(klr_ctx* ctx, const char* name, const char* value) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, name != nullptr && strncmp(name, "KLR_", 4) == 0);
  if (value) ctx->opts[name] = value; else ctx->opts.erase(name);
  ctx_parse_switches(ctx);
  return KLR_OK;
}

extern "C" int klr_ctx_reload_env(klr_ctx* ctx) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  ctx_load_env(ctx);
  return KLR_OK;
}

int ctx_func_lds(klr_ctx* ctx, const void* fn, int bytes) {
  if (ctx->lds_attr.count(fn)) return KLR_OK;
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  KLR_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  ctx->lds_attr.insert(fn);
  return KLR_OK;
}

extern "C" const char* klr_last_error(const klr_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_klr_err.c_str();
}

static void ctx_register(const klr_ctx* ctx);

extern "C" int klr_ctx_create(int device_id, klr_ctx** out) {
  if (!out) return klr_fail(nullptr, KLR_EINVAL, "klr_ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return klr_fail(nullptr, KLR_ENODEV, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev)
    return klr_fail(nullptr, KLR_EINVAL, "device_id %d out of range [0,%d)", device_id, ndev);
  hipDeviceProp_t prop;
  KLR_HIP(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return klr_fail(nullptr, KLR_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only",
                    device_id, prop.gcnArchName);
  KLR_HIP(nullptr, hipSetDevice(device_id));
  klr_ctx* ctx = new klr_ctx();
  ctx->device = device_id;
  ctx->n_cu = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ctx;
    return klr_fail(nullptr, KLR_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
  }
  ctx->own_stream = true;
  ctx_load_env(ctx);
  ctx_register(ctx);
  *out = ctx;
  return KLR_OK;
}

// A second context on the parent's device with the parent's switches: own stream, own workspaces.  For concurrent work
// on shared data -- a fold of a cross-validation per context (klr_mat_view).  A context with a communicator is not
// forked: the folds of all ranks would have to post their collectives in one agreed order.
extern "C" int klr_ctx_fork(klr_ctx* parent, klr_ctx** out) {
  if (!parent) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(parent, out != nullptr);
  *out = nullptr;
  if (parent->ar_fn || parent->nranks > 1) return klr_fail(parent, KLR_EINVAL, "klr_ctx_fork: a context with collectives is not forked");
  klr_ctx* c = nullptr;
  {
    std::lock_guard<std::mutex> lk(parent->fork_mu);
    if (!parent->fork_idle.empty()) { c = parent->fork_idle.back(); parent->fork_idle.pop_back(); }
  }
  if (!c) {
    KLR_TRY(klr_ctx_create(parent->device, &c));
    c->fork_parent = parent;
  }
  c->opts = parent->opts;
  ctx_parse_switches(c);
  *out = c;
  return KLR_OK;
}

// A forked context goes back to its parent (stream drained, workspaces kept) and ends with it; any other context, or a
// fork whose parent is gone, is destroyed.
extern "C" void klr_ctx_release(klr_ctx* ctx) {
  if (!ctx) return;
  klr_ctx* parent = ctx->fork_parent;
  if (parent && ctx_alive(parent)) {
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    std::lock_guard<std::mutex> lk(parent->fork_mu);
    parent->fork_idle.push_back(ctx);
    return;
  }
  klr_ctx_destroy(ctx);
}

// ---- RCCL, loaded lazily so that single-GPU use has no dependency on it -------------------
struct Id128 { char b[128]; };
struct RcclState {
  void* lib = nullptr;
  void* comm = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

static int rccl_load(RcclState* st) {
  if (st->lib) return KLR_OK;
  // 1. an RCCL this process has already mapped (a torch host bundles its own copy next to its HIP runtime: use that one,
  //    not a second library against another runtime); 2. the ROCm installation by absolute path (a Go / C host has
  //    nothing mapped and may have no library path set); 3. the loader's search path.
  std::string tried;
  const char* loaded[] = {"librccl.so.1", "librccl.so"};
  for (const char* nm : loaded) {
    st->lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (st->lib) break;
  }
  if (!st->lib) {
    std::vector<std::string> paths;
    if (const char* rp = getenv("ROCM_PATH")) { paths.push_back(std::string(rp) + "/lib/librccl.so.1"); paths.push_back(std::string(rp) + "/lib/librccl.so"); }
    paths.push_back("/opt/rocm/lib/librccl.so.1");
    paths.push_back("/opt/rocm/lib/librccl.so");
    paths.push_back("librccl.so.1");
    paths.push_back("librccl.so");
    for (const std::string& pth : paths) {
      st->lib = dlopen(pth.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (st->lib) break;
      const char* de = dlerror();
      tried += pth + ": " + (de ? de : "?") + "; ";
    }
  }
  if (!st->lib) return klr_fail(nullptr, KLR_ECOMM, "cannot load librccl (%s)", tried.c_str());
  st->GetUniqueId = (int (*)(void*))dlsym(st->lib, "ncclGetUniqueId");
  st->CommInitRank = (int (*)(void**, int, Id128, int))dlsym(st->lib, "ncclCommInitRank");
  st->AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(st->lib, "ncclAllReduce");
  st->ReduceScatter = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(st->lib, "ncclReduceScatter");
  st->AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(st->lib, "ncclAllGather");
  st->CommDestroy = (int (*)(void*))dlsym(st->lib, "ncclCommDestroy");
  st->CommCount = (int (*)(void*, int*))dlsym(st->lib, "ncclCommCount");
  st->GetErrorString = (const char* (*)(int))dlsym(st->lib, "ncclGetErrorString");
  if (!st->GetUniqueId || !st->CommInitRank || !st->AllReduce || !st->CommDestroy)
    return klr_fail(nullptr, KLR_ECOMM, "librccl lacks the expected nccl* symbols");
  return KLR_OK;
}

extern "C" int klr_rccl_unique_id(void* id128) {
  if (!id128) return klr_fail(nullptr, KLR_EINVAL, "id128 is NULL");
  static RcclState st;
  static std::mutex mu;                              // hosts with one context per thread may all ask for an id
  std::lock_guard<std::mutex> lk(mu);
  KLR_TRY(rccl_load(&st));
  int rc = st.GetUniqueId(id128);
  if (rc != 0) return klr_fail(nullptr, KLR_ECOMM, "ncclGetUniqueId failed (%d)", rc);
  return KLR_OK;
}

static int rccl_reduce_scatter_hook(void* user, const void* send, void* recv, int64_t recv_count, int dtype, void* stream) {
  RcclState* st = (RcclState*)user;
  const int ncclInt64 = 4, ncclFloat64 = 8, ncclSum = 0;
  if (!st->ReduceScatter) return 1;
  return st->ReduceScatter(send, recv, (size_t)recv_count, dtype == KLR_I64 ? ncclInt64 : ncclFloat64, ncclSum, st->comm, (hipStream_t)stream) == 0 ? 0 : 1;
}
static int rccl_allgather_hook(void* user, const void* send, void* recv, int64_t send_bytes, void* stream) {
  RcclState* st = (RcclState*)user;
  const int ncclInt8 = 0;
  if (!st->AllGather) return 1;
  return st->AllGather(send, recv, (size_t)send_bytes, ncclInt8, st->comm, (hipStream_t)stream) == 0 ? 0 : 1;
}

static int rccl_allreduce_hook(void* user, void* dev_buf, int64_t count, int dtype, void* stream) {
  RcclState* st = (RcclState*)user;
  const int ncclInt64 = 4, ncclFloat64 = 8, ncclSum = 0;
  int rc = st->AllReduce(dev_buf, dev_buf, (size_t)count, dtype == KLR_I64 ? ncclInt64 : ncclFloat64, ncclSum, st->comm,
                         (hipStream_t)stream);
  return rc == 0 ? 0 : 1;
}

extern "C" int klr_ctx_init_rccl(klr_ctx* ctx, const void* id128, int rank, int nranks) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
  RcclState* st = new RcclState();
  int rc = rccl_load(st);
  if (rc != KLR_OK) { ctx->err = g_klr_err; delete st; return rc; }
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  Id128 id;
  memcpy(id.b, id128, 128);
  int nrc = st->CommInitRank(&st->comm, nranks, id, rank);
  if (nrc != 0) {
    const char* es = st->GetErrorString ? st->GetErrorString(nrc) : "?";
    delete st;
    return klr_fail(ctx, KLR_ECOMM, "ncclCommInitRank failed: %s", es);
  }
  ctx->rccl = st;
  ctx->ar_fn = rccl_allreduce_hook;
  ctx->ar_user = st;
  if (st->ReduceScatter && st->AllGather) { ctx->rs_fn = rccl_reduce_scatter_hook; ctx->ag_fn = rccl_allgather_hook; ctx->coll_user = st; }
  ctx->rank = rank;
  ctx->nranks = nranks;
  // does the communicator really reduce over the ranks it claims?  [1, rank] as int64 must come back as [N, N(N-1)/2]
  // (and the same for f64); a failure here ends the run before any gradient is summed.  Reduce-scatter / all-gather that
  // fail their part of the check are dropped instead -- the sweep then keeps the all-reduce form, which passed (every
  // rank sees the same failure: the expected values are the same everywhere, so all ranks drop them together)
  rc = klr_ctx_comm_selfcheck(ctx);
  if (rc != KLR_OK && ctx->rs_fn && (ctx->err.find("reduce-scatter") != std::string::npos || ctx->err.find("all-gather") != std::string::npos)) {
    fprintf(stderr, "kmerlr: %s -- the sweep keeps the all-reduce form\n", ctx->err.c_str());
    ctx->rs_fn = nullptr; ctx->ag_fn = nullptr; ctx->coll_user = nullptr;
    rc = klr_ctx_comm_selfcheck(ctx);
  }
  return rc;
}

extern "C" int klr_ctx_set_allreduce(klr_ctx* ctx, klr_allreduce_fn fn, void* user, int rank, int nranks) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, nranks >= 1 && rank >= 0 && rank < nranks);
  ctx->ar_fn = fn;
  ctx->ar_user = user;
  ctx->rank = rank;
  ctx->nranks = nranks;
  ctx->rs_fn = nullptr; ctx->ag_fn = nullptr; ctx->coll_user = nullptr;     // belong to the previous hook
  ctx->comm_checked = false;
  return KLR_OK;
}

extern "C" int klr_ctx_comm_info(const klr_ctx* ctx, int* rank, int* nranks, int* rccl_ranks) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  if (rank) *rank = ctx->rank;
  if (nranks) *nranks = ctx->nranks;
  if (rccl_ranks) {
    *rccl_ranks = 0;                                  // 0: no RCCL communicator (no hook, or a caller-supplied hook)
    const RcclState* st = (const RcclState*)ctx->rccl;
    if (st && st->comm && st->CommCount) { int c = 0; if (st->CommCount(st->comm, &c) == 0) *rccl_ranks = c; }
  }
  return KLR_OK;
}

// average time of the all-reduce hook alone on `count` elements (a scratch buffer on the ctx stream, HIP events around
// `iters` back-to-back calls after one untimed call): what bench.py reports beside the step it is part of
extern "C" int klr_allreduce_time(klr_ctx* ctx, int64_t count, int dtype, int iters, double* ms_avg) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, count >= 1 && iters >= 1 && ms_avg != nullptr && (dtype == KLR_F64 || dtype == KLR_I64));
  *ms_avg = 0.0;
  if (!ctx->ar_fn) return KLR_OK;
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  void* buf;
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)count * 8, &buf));
  KLR_HIP(ctx, hipMemsetAsync(buf, 0, (size_t)count * 8, ctx->stream));
  KLR_TRY(klr_allreduce(ctx, buf, count, dtype));
  hipEvent_t e0, e1;
  KLR_HIP(ctx, hipEventCreate(&e0));
  KLR_HIP(ctx, hipEventCreate(&e1));
  KLR_HIP(ctx, hipEventRecord(e0, ctx->stream));
  int rc = KLR_OK;
  for (int i = 0; i < iters && rc == KLR_OK; i++) rc = klr_allreduce(ctx, buf, count, dtype);
  hipError_t he = hipEventRecord(e1, ctx->stream);
  if (he == hipSuccess) he = hipEventSynchronize(e1);
  float ms = 0.f;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc != KLR_OK) return rc;
  if (he != hipSuccess) return klr_fail(ctx, KLR_EHIP, "timing the all-reduce failed: %s", hipGetErrorString(he));
  *ms_avg = (double)ms / iters;
  return KLR_OK;
}

// Collective self-check: every installed hook is run once on small buffers whose sums every rank can predict --
// all-reduce as int64 and f64, and (when installed) reduce-scatter and all-gather -- and the results are compared on
// the host.  A communicator that does not really span `nranks` ranks, a hook that reduces the wrong element type or
// scatters the wrong slice fails here, before any gradient goes through it.  Collective: every rank must call it.
extern "C" int klr_ctx_comm_selfcheck(klr_ctx* ctx) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  ctx->comm_checked = false;
  if (!ctx->ar_fn) { ctx->comm_checked = true; return KLR_OK; }
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 R = ctx->nranks, me = ctx->rank;
  const i64 tri = R * (R + 1) / 2;                      // sum of (rank + 1)
  i64* buf;
  const i64 C3 = 3;
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)(64 + C3 * R + C3 + 16 * (R + 1)) * 8, (void**)&buf));
  // int64 all-reduce
  i64 hi[3] = {1, me, (me + 1) << 40};
  KLR_TRY(h2d(ctx, buf, hi, sizeof(hi)));
  KLR_TRY(klr_allreduce(ctx, buf, 3, KLR_I64));
  KLR_TRY(d2h_sync(ctx, hi, buf, sizeof(hi)));
  if (hi[0] != R || hi[1] != R * (R - 1) / 2 || hi[2] != (tri << 40))
    return klr_fail(ctx, KLR_ECOMM, "collective self-check: int64 all-reduce over %lld ranks returned [%lld, %lld, %lld], expected [%lld, %lld, %lld]",
                    (long long)R, (long long)hi[0], (long long)hi[1], (long long)hi[2], (long long)R, (long long)(R * (R - 1) / 2), (long long)(tri << 40));
  // f64 all-reduce
  double hd[2] = {0.5, (double)me};
  KLR_TRY(h2d(ctx, buf, hd, sizeof(hd)));
  KLR_TRY(klr_allreduce(ctx, buf, 2, KLR_F64));
  KLR_TRY(d2h_sync(ctx, hd, buf, sizeof(hd)));
  if (hd[0] != 0.5 * (double)R || hd[1] != (double)(R * (R - 1) / 2))
    return klr_fail(ctx, KLR_ECOMM, "collective self-check: f64 all-reduce over %lld ranks returned [%g, %g]", (long long)R, hd[0], hd[1]);
  if (ctx->rs_fn && ctx->ag_fn) {
    // every rank checks its own slice / the gathered blocks, then the verdicts are summed with the all-reduce that has
    // just passed: all ranks reach the SAME conclusion (a rank that dropped the hooks alone would leave the others
    // waiting in the next reduce-scatter)
    std::string why;
    // reduce-scatter: send[q * C3 + i] = (rank + 1) * (1000 q + i + 1)  ->  recv[i] = (1000 me + i + 1) * tri
    std::vector<i64> send((size_t)(C3 * R));
    for (i64 q = 0; q < R; q++) for (i64 i = 0; i < C3; i++) send[(size_t)(q * C3 + i)] = (me + 1) * (1000 * q + i + 1);
    i64* dsend = buf + 64; i64* drecv = dsend + C3 * R;
    KLR_TRY(h2d(ctx, dsend, send.data(), send.size() * 8));
    if (ctx->rs_fn(ctx->coll_user, dsend, drecv, C3, KLR_I64, (void*)ctx->stream) != 0) why = "reduce-scatter hook failed";
    i64 got[3] = {0, 0, 0};
    KLR_TRY(d2h_sync(ctx, got, drecv, sizeof(got)));
    for (i64 i = 0; i < C3 && why.empty(); i++)
      if (got[i] != (1000 * me + i + 1) * tri) {
        char t[256];
        snprintf(t, sizeof(t), "reduce-scatter gave %lld at element %lld of rank %lld's slice, expected %lld", (long long)got[i], (long long)i,
                 (long long)me, (long long)((1000 * me + i + 1) * tri));
        why = t;
      }
    // all-gather: 16 bytes per rank, byte b of rank r's block = (31 r + b) mod 251
    unsigned char blk[16];
    for (int b = 0; b < 16; b++) blk[b] = (unsigned char)((31 * me + b) % 251);
    char* dblk = (char*)(drecv + C3); char* dall = dblk + 16;
    KLR_TRY(h2d(ctx, dblk, blk, 16));
    if (ctx->ag_fn(ctx->coll_user, dblk, dall, 16, (void*)ctx->stream) != 0 && why.empty()) why = "all-gather hook failed";
    std::vector<unsigned char> all((size_t)(16 * R));
    KLR_TRY(d2h_sync(ctx, all.data(), dall, all.size()));
    for (i64 r = 0; r < R && why.empty(); r++) for (int b = 0; b < 16; b++)
      if (all[(size_t)(r * 16 + b)] != (unsigned char)((31 * r + b) % 251)) { why = "all-gather returned a wrong block"; break; }
    i64 bad[1] = {why.empty() ? 0 : 1};
    KLR_TRY(h2d(ctx, buf, bad, 8));
    KLR_TRY(klr_allreduce(ctx, buf, 1, KLR_I64));
    KLR_TRY(d2h_sync(ctx, bad, buf, 8));
    if (bad[0] != 0)
      return klr_fail(ctx, KLR_ECOMM, "collective self-check: %s (%lld of %lld ranks saw a reduce-scatter / all-gather error)",
                      why.empty() ? "another rank's reduce-scatter / all-gather check failed" : why.c_str(), (long long)bad[0], (long long)R);
  }
  ctx->comm_checked = true;
  return KLR_OK;
}

extern "C" int klr_ctx_comm_probe(const klr_ctx* ctx, int64_t* elements, double* reduce_scatter_ms, double* allgather_ms,
                                  double* allreduce_ms, int* scatter_form, int* ranges) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  if (elements) *elements = ctx->probe_dim ? ctx->probe_dim + IB_EXTRA : 0;
  if (reduce_scatter_ms) *reduce_scatter_ms = ctx->probe_ms;
  if (allgather_ms) *allgather_ms = ctx->probe_ag_ms;
  if (allreduce_ms) *allreduce_ms = ctx->probe_ar_ms;
  if (scatter_form) *scatter_form = ctx->probe_dim ? (ctx->probe_scatter ? 1 : 0) : -1;
  if (ranges) *ranges = ctx->probe_G;
  return KLR_OK;
}

extern "C" int klr_ctx_comm_stats(const klr_ctx* ctx, int64_t* sweeps_allreduce, int64_t* sweeps_scatter, int64_t* scatter_fallbacks, int* checked) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  if (sweeps_allreduce) *sweeps_allreduce = ctx->n_ar_sweeps;
  if (sweeps_scatter) *sweeps_scatter = ctx->n_rs_sweeps;
  if (scatter_fallbacks) *scatter_fallbacks = ctx->n_rs_fallbacks;
  if (checked) *checked = ctx->comm_checked ? 1 : 0;
  return KLR_OK;
}

extern "C" int klr_ctx_comm_ranged(const klr_ctx* ctx, int64_t* sweeps_ranged) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  if (sweeps_ranged) *sweeps_ranged = ctx->n_ranged_sweeps;
  return KLR_OK;
}

// klr_backend.solve_run_begin: a new run of reduced solves (one leapfrog estimate) starts on this context
extern "C" void klr_solve_run_begin(klr_ctx* ctx) {
  if (!ctx) return;
  ctx->cd_pred.clear();
  ctx->cd_pred_on = !ctx_opt(ctx, "KLR_CD_NO_PREDICT");
}

extern "C" void klr_solve_run_end(klr_ctx* ctx) {
  if (!ctx) return;
  ctx->cd_pred.clear();
  ctx->cd_pred_on = false;
}

extern "C" int klr_ctx_set_collectives(klr_ctx* ctx, klr_reduce_scatter_fn rs, klr_allgather_fn ag, void* user) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, (rs == nullptr) == (ag == nullptr));
  ctx->rs_fn = rs; ctx->ag_fn = ag; ctx->coll_user = user;
  ctx->comm_checked = false; ctx->probe_dim = 0;
  return KLR_OK;
}

int klr_allreduce(klr_ctx* ctx, void* dev, i64 count, int dtype) {
  if (!ctx->ar_fn) return KLR_OK;      // a hook installed for a single rank still runs (RCCL: in-place no-op)
  int rc = ctx->ar_fn(ctx->ar_user, dev, count, dtype, (void*)ctx->stream);
  if (rc != 0) return klr_fail(ctx, KLR_ECOMM, "all-reduce hook failed (%d) on %lld elements", rc, (long long)count);
  return KLR_OK;
}

// ---- live contexts: a matrix released after its context went away frees its block itself ------------------
static std::mutex g_live_mu;
static std::set<const klr_ctx*> g_live;
bool ctx_alive(const klr_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live.count(ctx) != 0;
}
static void ctx_register(const klr_ctx* ctx) { std::lock_guard<std::mutex> lk(g_live_mu); g_live.insert(ctx); }
static void ctx_unregister(const klr_ctx* ctx) { std::lock_guard<std::mutex> lk(g_live_mu); g_live.erase(ctx); }

// ---- pooled blocks of reduced matrices.  Everything that touches a block runs on ctx->stream, so a block put
// back by the host while kernels on it are still queued is safe to hand to the next sub-select (stream order);
// hipFree (eviction) synchronises by itself.
#define POOL_MAX 4
int pool_get(klr_ctx* ctx, size_t bytes, void** p, size_t* cap) {
  int best = -1;
  for (int i = 0; i < (int)ctx->pool.size(); i++)
    if (ctx->pool[i].second >= bytes && (best < 0 || ctx->pool[i].second < ctx->pool[best].second)) best = i;
  if (best >= 0) {
    *p = ctx->pool[best].first; *cap = ctx->pool[best].second;
    ctx->pool.erase(ctx->pool.begin() + best);
    return KLR_OK;
  }
  // head room: the next epoch's selection is a little larger or smaller than this one
  const size_t want = bytes + bytes / 2 + (1u << 16);
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(p, bytes); if (e == hipSuccess) { *cap = bytes; return KLR_OK; } }
  if (e != hipSuccess) return klr_fail(ctx, KLR_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  *cap = want;
  return KLR_OK;
}
void pool_put(klr_ctx* ctx, void* p, size_t cap) {
  if (!p) return;
  ctx->pool.emplace_back(p, cap);
  if ((int)ctx->pool.size() > POOL_MAX) {                         // drop the smallest
    int worst = 0;
    for (int i = 1; i < (int)ctx->pool.size(); i++) if (ctx->pool[i].second < ctx->pool[worst].second) worst = i;
    (void)hipFree(ctx->pool[worst].first);
    ctx->pool.erase(ctx->pool.begin() + worst);
  }
}

int stage_get(klr_ctx* ctx, size_t bytes, void** out) {
  if (!ctx->stage_ev) KLR_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev, hipEventDisableTiming));
  else KLR_HIP(ctx, hipEventSynchronize(ctx->stage_ev));
  if (ctx->stage_cap < bytes) {
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    ctx->stage = nullptr; ctx->stage_cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipHostMalloc(&ctx->stage, want, hipHostMallocDefault);
    if (e != hipSuccess) return klr_fail(ctx, KLR_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->stage_cap = want;
  }
  *out = ctx->stage;
  return KLR_OK;
}

extern "C" void klr_ctx_destroy(klr_ctx* ctx) {
  if (!ctx) return;
  {
    std::vector<klr_ctx*> idle;
    { std::lock_guard<std::mutex> lk(ctx->fork_mu); idle.swap(ctx->fork_idle); }
    for (klr_ctx* c : idle) { c->fork_parent = nullptr; klr_ctx_destroy(c); }
  }
  ctx_unregister(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : ctx->pool) (void)hipFree(b.first);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  if (ctx->stage_ev) (void)hipEventDestroy(ctx->stage_ev);
  for (auto& pr : ctx->prof_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& pr : ctx->prof_cd) { (void)hipEventDestroy(pr.e0); (void)hipEventDestroy(pr.e1); }
  for (hipEvent_t ev : ctx->prof_free) (void)hipEventDestroy(ev);
  if (ctx->evict_buf) (void)hipFree(ctx->evict_buf);
  for (int i = 0; i < WS_COUNT; i++) if (ctx->ws[i].p) (void)hipFree(ctx->ws[i].p);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  shm_release(ctx);
  for (hipEvent_t ev : ctx->range_ev) (void)hipEventDestroy(ev);
  if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
  if (ctx->rccl) {
    RcclState* st = (RcclState*)ctx->rccl;
    if (st->comm) st->CommDestroy(st->comm);
    delete st;
  }
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int klr_ctx_set_stream(klr_ctx* ctx, void* hip_stream) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hip_stream == nullptr) {
    if (!ctx->own_stream) {
      KLR_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = true;
    }
    return KLR_OK;
  }
  if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
  ctx->stream = (hipStream_t)hip_stream;
  return KLR_OK;
}

extern "C" int klr_ctx_synchronize(klr_ctx* ctx) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KLR_OK;
}

// ---- memory ---------------------------------------------------------------------------------
int dev_alloc(klr_ctx* ctx, void** p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess)
    return klr_fail(ctx, KLR_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return KLR_OK;
}

int ws_get(klr_ctx* ctx, int which, size_t bytes, void** out) {
  DevBuf& b = ctx->ws[which];
  if (bytes == 0) bytes = 16;
  if (b.cap < bytes) {
    // earlier work on the stream may still use the old block
    KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    KLR_TRY(dev_alloc(ctx, &b.p, want));
    b.cap = want;
  }
  *out = b.p;
  return KLR_OK;
}

int pin_get(klr_ctx* ctx, size_t bytes, void** out) {
  if (ctx->pin_cap < bytes) {
    KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    ctx->pin = nullptr; ctx->pin_cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&ctx->pin, want, hipHostMallocDefault);
    if (e != hipSuccess) return klr_fail(ctx, KLR_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->pin_cap = want;
  }
  *out = ctx->pin;
  return KLR_OK;
}

int h2d(klr_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return KLR_OK;
  // pageable source: hipMemcpyAsync stages synchronously, so the host buffer is reusable on return
  KLR_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return KLR_OK;
}

int d2h_sync(klr_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes) KLR_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KLR_OK;
}

// ---- profiling --------------------------------------------------------------------------------
bool prof_take(klr_ctx* ctx) {
  if (!ctx->prof_on) return false;
  return (ctx->prof_count++ % ctx->prof_stride) == 0;
}

int prof_event_get(klr_ctx* ctx, hipEvent_t* ev) {
  if (!ctx->prof_free.empty()) { *ev = ctx->prof_free.back(); ctx->prof_free.pop_back(); return KLR_OK; }
  KLR_HIP(ctx, hipEventCreate(ev));
  return KLR_OK;
}

extern "C" int klr_prof_reserve(klr_ctx* ctx, int64_t events) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  while ((int64_t)ctx->prof_free.size() < events) { hipEvent_t ev; KLR_HIP(ctx, hipEventCreate(&ev)); ctx->prof_free.push_back(ev); }
  return KLR_OK;
}

extern "C" int klr_prof_enable(klr_ctx* ctx, int on) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  ctx->prof_on = on != 0;
  ctx->prof_stride = on > 1 ? on : 1;
  ctx->prof_count = 0;
  return KLR_OK;
}

extern "C" int klr_prof_read(klr_ctx* ctx, int64_t* launches, double* total_ms, int reset) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  for (auto& pr : ctx->prof_ev) {
    float ms = 0.f;
    KLR_HIP(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += (double)ms;
  }
  if (launches) *launches = (int64_t)ctx->prof_ev.size();
  if (total_ms) *total_ms = tot;
  if (reset) {
    for (auto& pr : ctx->prof_ev) { ctx->prof_free.push_back(pr.first); ctx->prof_free.push_back(pr.second); }
    ctx->prof_ev.clear();
  }
  return KLR_OK;
}

extern "C" int klr_prof_read_cd(klr_ctx* ctx, int64_t* sweeps, int64_t* batches, double* total_ms, int reset) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0; int64_t sw = 0;
  for (auto& pr : ctx->prof_cd) {
    float ms = 0.f;
    KLR_HIP(ctx, hipEventElapsedTime(&ms, pr.e0, pr.e1));
    tot += (double)ms; sw += pr.sweeps;
  }
  if (sweeps) *sweeps = sw;
  if (batches) *batches = (int64_t)ctx->prof_cd.size();
  if (total_ms) *total_ms = tot;
  if (reset) {
    for (auto& pr : ctx->prof_cd) { ctx->prof_free.push_back(pr.e0); ctx->prof_free.push_back(pr.e1); }
    ctx->prof_cd.clear();
  }
  return KLR_OK;
}

// Pushes the resident data out of the 256 MiB Infinity Cache: `bytes` written and then copied (read + written) inside a
// scratch buffer on the ctx stream.  bench.py's cold-stream leg calls it between steps.
extern "C" int klr_prof_evict(klr_ctx* ctx, int64_t bytes) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, bytes >= 2);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->evict_cap < (size_t)bytes) {
    KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->evict_buf) (void)hipFree(ctx->evict_buf);
    ctx->evict_buf = nullptr; ctx->evict_cap = 0;
    KLR_TRY(dev_alloc(ctx, &ctx->evict_buf, (size_t)bytes));
    ctx->evict_cap = (size_t)bytes;
  }
  const size_t half = (size_t)bytes / 2;
  KLR_HIP(ctx, hipMemsetAsync(ctx->evict_buf, 0x5a, (size_t)bytes, ctx->stream));
  KLR_HIP(ctx, hipMemcpyAsync((char*)ctx->evict_buf + half, ctx->evict_buf, half, hipMemcpyDeviceToDevice, ctx->stream));
  return KLR_OK;
}

// ---- backend table of the host driver (include/kmerlr_driver.h) ---------------------------------
#include "../../include/kmerlr_driver.h"

extern "C" int klr_hip_backend(klr_ctx* ctx, klr_backend* out) {
  if (!ctx || !out) return klr_fail(ctx, KLR_EINVAL, "klr_hip_backend: NULL argument");
  out->ctx = ctx;
  out->last_error = klr_last_error;
  out->mat_shape = klr_mat_shape;
  out->mat_destroy = klr_mat_destroy;
  out->class_weights = klr_class_weights;
  out->gradient = klr_gradient;
  out->loss = klr_loss;
  out->select_columns = klr_mat_select_columns;
  out->sweep_select = klr_sweep_select;
  out->cd_solve = klr_cd_solve;
  out->set_active_rows = klr_mat_set_active_rows;
  out->log_pdf = klr_log_pdf;
  out->expand_cooccurrence = klr_mat_expand_cooccurrence;
  out->transform_fit = klr_transform_fit;
  out->ctx_fork = klr_ctx_fork;
  out->ctx_release = klr_ctx_release;
  out->mat_view = klr_mat_view;
  out->solve_run_begin = klr_solve_run_begin;
  out->solve_run_end = klr_solve_run_end;
  return KLR_OK;
}
