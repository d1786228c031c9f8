// nbp_api.hip -- host side of libnbp: the C ABI declared in include/nbp.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC nbp_api.hip -o libnbp.so
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unordered_map>

#include <dlfcn.h>
#include <algorithm>
#include <rccl/rccl.h>  // types and prototypes only: the entry points are resolved with dlsym (see rccl_api)

#ifndef NBP_TU
#define NBP_TU 0  // host code: the kernels live in the nbp_k_*.hip files (-DNBP_TU=0xFFFF: single-file build with every kernel)
#endif
#include "nbp_kernels.h"
#include "nbp_fused.h"

static thread_local std::string g_err;
static nbp_status fail(nbp_status code, const std::string &msg) {
  g_err = msg;
  return code;
}
// shared with nbp_host.cpp (same library): sets the message nbp_last_error() returns
// host-side loops over many beliefs (a tree level's worth of them packed into / unpacked from the staging buffer, ~1 us
// each): a few threads when there is enough work (profiles/r04_clique_seam_phases.txt)
template <class F>
static void host_parallel_for(int n, int grain, F &&f) {
  const unsigned hw = std::thread::hardware_concurrency();
  int nt = n / grain;
  if (nt > 8) nt = 8;
  if (hw && nt > (int)hw) nt = (int)hw;
  if (nt <= 1) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  auto part = [&](int t) {
    const int a = (int)((int64_t)n * t / nt), b = (int)((int64_t)n * (t + 1) / nt);
    for (int i = a; i < b; i++) f(i);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(part, t);
  part(0);
  for (std::thread &x : th) x.join();
}

extern "C" nbp_status nbp_internal_fail(nbp_status code, const char *msg) { return fail(code, msg ? msg : ""); }
#define HIPCHK(expr)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(NBP_ERR_HIP, std::string(#expr) + " (nbp_api.hip:" + std::to_string(__LINE__) + "): " + hipGetErrorString(e_)); \
  } while (0)

// Speculative fits (lcv_bandwidth_1d_spec): used when every workgroup of the launch is resident at once -- the launch,
// with 3 or 7 workgroups per fitted coordinate, stays below NBP_SPEC_MAXBLOCKS (the chip has 256 CUs and a fit's
// workgroup has a CU to itself: 1024 lanes, ~63 KB of LDS).  Counted are the workgroups that stay: those of a
// coordinate the manifold does not have leave at once (`coords` = sum of the manifold dimensions of the jobs).
#ifndef NBP_SPEC_MAXJOBS
#define NBP_SPEC_MAXJOBS 40
#endif
#ifndef NBP_SPEC_MAXBLOCKS
#define NBP_SPEC_MAXBLOCKS 224
#endif
struct nbp_program;
struct nbp_comm;
struct nbp_ctx {
  std::vector<nbp_program *> programs;  // live programs: detached (device blob freed, ctx = null) by nbp_ctx_destroy
  std::vector<nbp_comm *> comms;        // live communicators: shut down (and detached) by nbp_ctx_destroy
  void *attached = nullptr;             // an object of the layer above (nbp_ctx_attach: the native host's plan cache) ...
  void (*attached_destroy)(void *) = nullptr;  // ... destroyed at the top of nbp_ctx_destroy, while the context is whole
  uint64_t ws_gen = 0;  // bumped whenever a workspace a captured launch sequence holds by value (ws, gstats) is re-allocated
  int device = 0, N = 0, n_slots = 0, side_ints = 0, threads = 0, Npad = 0, P = 1;
  int64_t S = 0;
  double *arena = nullptr;
  bool own_arena = false;
  int32_t *side = nullptr;
  nbp_counters *counters = nullptr;
  hipStream_t stream = nullptr;
  nbp_levels T{};
  int32_t *lv_ints = nullptr;
  double *lv_dbls = nullptr;
  double *ws = nullptr;  // KD workspace: [products][kdF] x nbp_kd_ws_doubles(N), kdF = largest F of the batch
  size_t ws_doubles = 0;
  nbp_spec_area *spec = nullptr;  // rendezvous areas of the speculative fits (latency-mode launches), NBP_SPEC_MAXJOBS x 3
  bool spec_on = true;
  bool spec_depth3 = true;  // 7 workgroups per fit (three iterations per rendezvous) where the launch still fits the chip; NBP_SPEC_DEPTH3=0: 3 only
  double *gstats = nullptr;  // node statistics of products too large for the LDS
  size_t gstats_doubles = 0;
  // staging for immediate-mode calls
  void *stage = nullptr;
  size_t stage_bytes = 0;
  // pinned host staging of the batched belief transfers (nbp_belief_write_batch / _read_batch): one copy per run of
  // consecutive slots, asynchronous on the library stream
  double *pin = nullptr;
  size_t pin_doubles = 0;
  // device blobs of destroyed programs, kept for the next short-lived one (a clique call compiles, runs and drops a
  // program: hipMalloc / hipFree per call would synchronise the whole device each time)
  std::vector<std::pair<char *, size_t>> blob_cache;
  // ---- the asynchronous clique seam (nbp_clique_submit_batch / nbp_clique_wait, nbp_host.h) ----
  // pinned staging buffers of transfers that are queued WITHOUT waiting for the stream: a buffer is free again once the
  // event recorded behind its copy has completed (never waited for: a busy pool grows by a buffer)
  struct pin_buf { char *p = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool pending = false, held = false; };
  std::vector<pin_buf *> pin_pool;
  // programs handed to nbp_program_retire: destroyed once the event recorded behind their last launch has completed
  std::vector<std::pair<nbp_program *, hipEvent_t>> retired;
  int resident = 0;  // the LAST `resident` slots of the arena are resident slots (nbp_ctx_reserve_resident): handle h = slot n_slots - h
  // fused variable updates (nbp_fused.h) for the stages that fill the chip: NBP_NO_FUSED_UPDATE=1 turns them off,
  // NBP_FUSED_MIN = smallest stage (updates) that runs fused
  // Off unless NBP_FUSED_MIN is set: measured on config 2 and on the 10 000-variable chain the fused form moves a ninth of
  // the bytes and takes 10-35 % longer (one workgroup serialises the six fits of an update that the
  // three-launch form spreads over the chip).
  bool fused_on = true;
  int fused_min = 1 << 30;
  // smallest launch of simple Euclidean proposals that runs one WAVE per proposal (launch_proposals); -1: the default of the
  // class (NBP_PROPOSAL_WAVE_MIN overrides it for every class)
  int prop_wave_min = -1;
  int fused_p1_min = 2048;  // rounds with at least this many updates run one lane per particle (NBP_FUSED_P1_MIN); smaller
                            // ones two helper rows per update, so that they fill the chip with twice the lanes each
  // two-stream rounds (NBP_PIPELINE_MIN = smallest product batch; see plan_pipeline): the second stream, the fork / join
  // events, and the geometry of the WHOLE batch that both halves launch with (so that no result depends on the split)
  hipStream_t stream2 = nullptr;
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};
  int pipe_min = 1 << 30;
  int geom_n = 0, geom_blocks = 0;
  // timing: 0 proposal, 1 prep, 2 product, 3 plain bandwidth, 4 fused update kernel
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[5];
  double ms[5] = {0, 0, 0, 0, 0};
  int64_t nl[5] = {0, 0, 0, 0, 0};
};

static int manifold_dim_h(int m) { return m == NBP_SE2 ? 3 : (m == NBP_CIRCULAR ? 1 : m); }
static int manifold_P_h(int m) { return m == NBP_SE2 ? 6 : manifold_dim_h(m); }
static bool manifold_ok(int m) { return m >= NBP_EUCLID1 && m <= NBP_SE2; }
static double wrap_h(double a) { return nbpm_wrap_pi(a); }  // (include/nbp_math.h)

extern "C" {

int64_t nbp_slot_stride_doubles(int32_t N) { return 3 * (int64_t)N + 8; }
int64_t nbp_arena_bytes(int32_t N, int32_t n_slots) { return nbp_slot_stride_doubles(N) * 8 * (int64_t)n_slots; }
const char *nbp_last_error(void) { return g_err.c_str(); }

// data-independent level tables of the balanced KD-tree over N leaves
static nbp_status build_levels(nbp_ctx *c) {
  const int N = c->N;
  std::vector<std::vector<int>> lo(1), hi(1), child;
  lo[0] = {0};
  hi[0] = {N};
  int l = 0;
  for (;;) {
    bool all_leaf = true;
    for (size_t k = 0; k < lo[l].size(); k++)
      if (hi[l][k] - lo[l][k] > 1) all_leaf = false;
    if (all_leaf) break;
    lo.emplace_back();
    hi.emplace_back();
    child.emplace_back();
    for (size_t k = 0; k < lo[l].size(); k++) {
      int a = lo[l][k], b = hi[l][k];
      if (b - a == 1) {
        lo[l + 1].push_back(a);
        hi[l + 1].push_back(b);
      } else {
        int mid = a + (b - a + 1) / 2;
        lo[l + 1].push_back(a); hi[l + 1].push_back(mid);
        lo[l + 1].push_back(mid); hi[l + 1].push_back(b);
      }
      child[l].push_back((int)lo[l + 1].size() - 1);
    }
    l++;
    if (l >= NBP_MAXLEVELS - 1) return fail(NBP_ERR_RANGE, "tree too deep");
  }
  const int L = l;
  std::vector<int32_t> ints;
  std::vector<double> dbls;
  int total = 0;
  for (int i = 0; i <= L; i++) { c->T.cnt[i] = (int)lo[i].size(); c->T.off[i] = total; total += c->T.cnt[i]; }
  if ((size_t)total > nbp_kd_nodes_cap(N)) return fail(NBP_ERR_RANGE, "level tables exceed the node-sum workspace");
  std::vector<int32_t> nlo(total), nhi(total), nch(total, 0), pos((size_t)(L + 1) * N);
  dbls.resize(2 * (size_t)total);
  for (int i = 0; i <= L; i++)
    for (int k = 0; k < c->T.cnt[i]; k++) {
      nlo[c->T.off[i] + k] = lo[i][k];
      nhi[c->T.off[i] + k] = hi[i][k];
      if (i < L) nch[c->T.off[i] + k] = child[i][k];
      dbls[c->T.off[i] + k] = std::log((double)(hi[i][k] - lo[i][k]) / (double)N);
      dbls[total + c->T.off[i] + k] = (double)(hi[i][k] - lo[i][k]) / (double)N;
      for (int p = lo[i][k]; p < hi[i][k]; p++) pos[(size_t)i * N + p] = k;
    }
  ints.insert(ints.end(), nlo.begin(), nlo.end());
  ints.insert(ints.end(), nhi.begin(), nhi.end());
  ints.insert(ints.end(), nch.begin(), nch.end());
  ints.insert(ints.end(), pos.begin(), pos.end());
  HIPCHK(hipMalloc(&c->lv_ints, ints.size() * 4));
  HIPCHK(hipMalloc(&c->lv_dbls, dbls.size() * 8));
  HIPCHK(hipMemcpy(c->lv_ints, ints.data(), ints.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->lv_dbls, dbls.data(), dbls.size() * 8, hipMemcpyHostToDevice));
  c->T.N = N;
  c->T.L = L;
  c->T.node_lo = c->lv_ints;
  c->T.node_hi = c->lv_ints + total;
  c->T.node_child = c->lv_ints + 2 * total;
  c->T.pos_node = c->lv_ints + 3 * total;
  c->T.node_logw = c->lv_dbls;
  c->T.node_w = c->lv_dbls + total;
  return NBP_OK;
}

nbp_status nbp_ctx_create(int32_t device, int32_t N, int32_t n_slots, void *arena, int64_t arena_bytes,
                          int32_t side_ints, nbp_ctx **out) {
  if (!out) return fail(NBP_ERR_ARG, "out is null");
  if (N < 8 || N > NBP_MAXN) return fail(NBP_ERR_RANGE, "N must be in [8, 512]");
  if (n_slots < 1) return fail(NBP_ERR_ARG, "n_slots < 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NBP_ERR_NOGPU, "no HIP device visible: libnbp has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(NBP_ERR_ARG, "bad device index");
  HIPCHK(hipSetDevice(device));
  nbp_ctx *c = new nbp_ctx();
  // any failure below releases the half-built context (stream, side buffer, counters, arena)
  struct guard_t {
    nbp_ctx *c;
    ~guard_t() { if (c) nbp_ctx_destroy(c); }
  } guard{c};
  c->device = device;
  c->N = N;
  c->n_slots = n_slots;
  c->S = nbp_slot_stride_doubles(N);
  c->Npad = ((N + 63) / 64) * 64;
  c->P = 1024 / c->Npad;
  if (c->P > 4) c->P = 4;
  if (c->P < 1) c->P = 1;
  c->threads = c->P * c->Npad;
  c->side_ints = side_ints > 0 ? side_ints : 1;
  if (arena) {
    if (arena_bytes < nbp_arena_bytes(N, n_slots)) return fail(NBP_ERR_ARG, "arena too small");
    c->arena = (double *)arena;
  } else {
    HIPCHK(hipMalloc(&c->arena, nbp_arena_bytes(N, n_slots)));
    HIPCHK(hipMemset(c->arena, 0, nbp_arena_bytes(N, n_slots)));
    c->own_arena = true;
  }
  HIPCHK(hipMalloc(&c->side, (size_t)c->side_ints * 4));
  HIPCHK(hipMemset(c->side, 0, (size_t)c->side_ints * 4));
  HIPCHK(hipMalloc(&c->counters, sizeof(nbp_counters)));
  HIPCHK(hipMemset(c->counters, 0, sizeof(nbp_counters)));
  {  // NBP_FIT_F64=1: the bandwidth searches evaluate in double precision only (no single-precision bracketing, neg_loo_ll_f32)
    const char *e = getenv("NBP_FIT_F64");
    const unsigned long long fl = (e && *e && *e != '0') ? 1ull : 0ull;
    HIPCHK(hipMemcpy(&c->counters->flags, &fl, sizeof(fl), hipMemcpyHostToDevice));
  }
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc(&c->spec, sizeof(nbp_spec_area) * 3 * NBP_SPEC_MAXJOBS));
  c->spec_on = getenv("NBP_NO_SPECULATIVE_FITS") == nullptr;
  c->spec_depth3 = !(getenv("NBP_SPEC_DEPTH3") && atoi(getenv("NBP_SPEC_DEPTH3")) == 0);
  c->fused_on = getenv("NBP_NO_FUSED_UPDATE") == nullptr;
  if (getenv("NBP_FUSED_MIN")) c->fused_min = atoi(getenv("NBP_FUSED_MIN"));
  if (getenv("NBP_PROPOSAL_WAVE_MIN")) c->prop_wave_min = atoi(getenv("NBP_PROPOSAL_WAVE_MIN"));
  if (getenv("NBP_PIPELINE_MIN")) c->pipe_min = atoi(getenv("NBP_PIPELINE_MIN"));
  HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&c->pipe_ev[i], hipEventDisableTiming));
  if (getenv("NBP_FUSED_P1_MIN")) c->fused_p1_min = atoi(getenv("NBP_FUSED_P1_MIN"));
  nbp_status rc = build_levels(c);
  if (rc != NBP_OK) return rc;
  // allow the full 160 KiB LDS for the product kernel
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_x16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_y32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_l8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_x16_w1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_y32_w1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_l8_w1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_m4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_product_kernel_t2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (const void *k : {(const void *)nbp_product_kernel_t2_e1, (const void *)nbp_product_kernel_t2_e2, (const void *)nbp_product_kernel_t2_e3,
                        (const void *)nbp_product_kernel_t2_ci, (const void *)nbp_product_kernel_t2_se, (const void *)nbp_product_kernel_m4_e1,
                        (const void *)nbp_product_kernel_m4_e2, (const void *)nbp_product_kernel_m4_e3, (const void *)nbp_product_kernel_m4_ci,
                        (const void *)nbp_product_kernel_m4_se, (const void *)nbp_product_kernel_t2_e1_xs, (const void *)nbp_product_kernel_t2_e2_xs,
                        (const void *)nbp_product_kernel_t2_e3_xs, (const void *)nbp_product_kernel_t2_ci_xs, (const void *)nbp_product_kernel_t2_se_xs,
                        (const void *)nbp_product_kernel_m4_e1_xs, (const void *)nbp_product_kernel_m4_e2_xs, (const void *)nbp_product_kernel_m4_e3_xs,
                        (const void *)nbp_product_kernel_m4_ci_xs, (const void *)nbp_product_kernel_m4_se_xs})
    HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_bandwidth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_bandwidth_kernel_spec<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_bandwidth_kernel_spec<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_prep_kernel_spec<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_prep_kernel_spec<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // (the kernel has 608 B of static LDS, the hypothesis recipe: static + dynamic must stay within the 160 KiB)
  HIPCHK(hipFuncSetAttribute((const void *)nbp_update_kernel_lin2_p1, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)nbp_update_kernel_lin2_p2, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));

  guard.c = nullptr;
  *out = c;
  return NBP_OK;
}

static void program_detach(nbp_program *p);
static void program_delete(nbp_program *p);  // detach + delete (defined behind the type)
static void ctx_detach_comms(nbp_ctx *c);
static void reap_retired(nbp_ctx *c);

nbp_status nbp_ctx_attach(nbp_ctx *c, void *obj, void (*destroy)(void *)) {
  if (!c) return fail(NBP_ERR_ARG, "ctx is null");
  if (c->attached && c->attached_destroy && c->attached != obj) c->attached_destroy(c->attached);
  c->attached = obj;
  c->attached_destroy = destroy;
  return NBP_OK;
}
void *nbp_ctx_attached(const nbp_ctx *c) { return c ? c->attached : nullptr; }

nbp_status nbp_ctx_destroy(nbp_ctx *c) {
  if (!c) return NBP_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->attached && c->attached_destroy) c->attached_destroy(c->attached);  // (may destroy programs of this context: first)
  c->attached = nullptr;
  for (nbp_program *p : c->programs) program_detach(p);  // a program outliving its context must not touch it
  c->programs.clear();
  ctx_detach_comms(c);  // likewise a communicator: shut down now, its handle stays valid for nbp_comm_destroy
  for (auto &v : c->ev)
    for (auto &p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  if (c->own_arena) hipFree(c->arena);
  if (c->side) hipFree(c->side);
  if (c->counters) hipFree(c->counters);
  if (c->lv_ints) hipFree(c->lv_ints);
  if (c->lv_dbls) hipFree(c->lv_dbls);
  if (c->stage) hipFree(c->stage);
  if (c->pin) hipHostFree(c->pin);
  for (auto &b : c->blob_cache) hipFree(b.first);
  c->blob_cache.clear();
  // the programs handed to nbp_program_retire that are still pending: the caller gave up ownership there, so they end here
  // (the stream was synchronised above: everything they queued has run; detached like the live ones, then deleted)
  for (auto &r : c->retired) {
    hipEventDestroy(r.second);
    program_delete(r.first);
  }
  c->retired.clear();
  for (nbp_ctx::pin_buf *b : c->pin_pool) {
    if (b->p) hipHostFree(b->p);
    if (b->ev) hipEventDestroy(b->ev);
    delete b;
  }
  c->pin_pool.clear();
  if (c->ws) hipFree(c->ws);
  if (c->gstats) hipFree(c->gstats);
  if (c->spec) hipFree(c->spec);
  if (c->stream2) hipStreamDestroy(c->stream2);
  for (int i = 0; i < 2; i++) if (c->pipe_ev[i]) hipEventDestroy(c->pipe_ev[i]);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return NBP_OK;
}

nbp_status nbp_synchronize(nbp_ctx *c) {
  if (!c) return fail(NBP_ERR_ARG, "ctx is null");
  HIPCHK(hipStreamSynchronize(c->stream));
  reap_retired(c);
  return NBP_OK;
}
void *nbp_arena_ptr(nbp_ctx *c) { return c ? c->arena : nullptr; }
void *nbp_stream_ptr(nbp_ctx *c) { return c ? (void *)c->stream : nullptr; }
int32_t nbp_ctx_particles(const nbp_ctx *c) { return c ? c->N : 0; }
int32_t nbp_ctx_slots(const nbp_ctx *c) { return c ? c->n_slots : 0; }
int32_t nbp_ctx_device(const nbp_ctx *c) { return c ? c->device : -1; }
int32_t nbp_ctx_resident(const nbp_ctx *c) { return c ? c->resident : 0; }
nbp_status nbp_ctx_reserve_resident(nbp_ctx *c, int32_t n) {
  if (!c) return fail(NBP_ERR_ARG, "ctx is null");
  if (n < 0 || n > c->n_slots) return fail(NBP_ERR_RANGE, "resident slots: more than the context holds");
  c->resident = n;
  return NBP_OK;
}

// a pinned buffer nobody holds and no queued copy still reads (or a new one)
static nbp_ctx::pin_buf *pin_acquire(nbp_ctx *c, size_t bytes) {
  nbp_ctx::pin_buf *best = nullptr;
  for (nbp_ctx::pin_buf *b : c->pin_pool) {
    if (b->held) continue;
    if (b->pending) {
      if (hipEventQuery(b->ev) != hipSuccess) continue;
      b->pending = false;
    }
    if (b->bytes >= bytes && (!best || b->bytes < best->bytes)) best = b;
  }
  if (!best) {
    best = new nbp_ctx::pin_buf();
    const size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    if (hipHostMalloc((void **)&best->p, cap, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&best->ev, hipEventDisableTiming) != hipSuccess) {
      if (best->p) hipHostFree(best->p);
      delete best;
      return nullptr;
    }
    best->bytes = cap;
    c->pin_pool.push_back(best);
  }
  best->held = true;
  return best;
}
// free once everything queued on the library stream so far has run (the copies out of / into the buffer among it)
static void pin_release_behind_stream(nbp_ctx *c, nbp_ctx::pin_buf *b) {
  b->pending = hipEventRecord(b->ev, c->stream) == hipSuccess;
  b->held = false;
}

// ---- belief I/O --------------------------------------------------------------------------------
nbp_status nbp_slot_write(nbp_ctx *c, int32_t slot, int32_t manifold, const double *pts, const double *bw) {
  return nbp_belief_write(c, slot, manifold, pts, c ? c->N : 0, bw, nullptr);
}
nbp_status nbp_slot_read(nbp_ctx *c, int32_t slot, int32_t manifold, double *pts, double *bw) {
  return nbp_belief_read(c, slot, manifold, pts, nullptr, bw, nullptr);
}

// a belief as its slot holds it: coordinates SoA over N rows, then bandwidth (3), infoPerCoord (3), count
static void pack_belief(const nbp_ctx *c, int32_t manifold, const double *pts, int32_t n_pts, const double *bw, const double *ipc, double *s) {
  const int N = c->N, D = manifold_dim_h(manifold), P = manifold_P_h(manifold);
  memset(s, 0, sizeof(double) * (size_t)c->S);
  const int cnt = n_pts < N ? n_pts : N;  // more than N points: the first N (GraphProductOperations.jl:44, `_pts[1:N]`)
  s[3 * N + 6] = cnt < N ? (double)cnt : 0.0;
  for (int n = 0; n < cnt; n++) {
    const double *p = pts + (size_t)n * P;
    if (manifold == NBP_SE2) {
      s[n] = p[0];
      s[N + n] = p[1];
      s[2 * N + n] = nbpm_atan2(p[3], p[2]);  // (the shared atan2, include/nbp_math.h: the CPU checker converts with the same function)
    } else if (manifold == NBP_CIRCULAR) {
      s[n] = wrap_h(p[0]);
    } else {
      for (int d = 0; d < D; d++) s[d * N + n] = p[d];
    }
  }
  for (int d = 0; d < D; d++) s[3 * N + d] = bw ? bw[d] : 0.0;
  for (int d = 0; d < D; d++) s[3 * N + 3 + d] = ipc ? ipc[d] : 0.0;  // a fresh VariableNodeData carries infoPerCoord = 0
}
// with n_pts: the rows the belief holds (the caller sized `pts` by the count it expects back, at most N); without
// (nbp_slot_read): all N rows of the slot
static void unpack_belief(const nbp_ctx *c, int32_t manifold, const double *s, double *pts, int32_t *n_pts, double *bw, double *ipc) {
  const int N = c->N, D = manifold_dim_h(manifold), P = manifold_P_h(manifold);
  const int cnt_held = (s[3 * N + 6] > 0.0 && s[3 * N + 6] < (double)N) ? (int)s[3 * N + 6] : N;
  const int rows = n_pts ? cnt_held : N;
  for (int n = 0; n < rows; n++) {
    double *p = pts + (size_t)n * P;
    if (manifold == NBP_SE2) {
      double th = s[2 * N + n];
      p[0] = s[n]; p[1] = s[N + n];
      double sn, cs;
      nbpm_sincos(th, &sn, &cs);  // (the shared sincos: a libm's cos / sin and its fused sincos round differently now and then)
      p[2] = cs; p[3] = sn; p[4] = -sn; p[5] = cs;
    } else {
      for (int d = 0; d < D; d++) p[d] = s[d * N + n];
    }
  }
  if (bw)
    for (int d = 0; d < D; d++) bw[d] = s[3 * N + d];
  if (ipc)
    for (int d = 0; d < D; d++) ipc[d] = s[3 * N + 3 + d];
  if (n_pts) *n_pts = cnt_held;
}
static nbp_status belief_args(nbp_ctx *c, int32_t slot, int32_t manifold, const double *pts) {
  if (!c || !pts) return fail(NBP_ERR_ARG, "null argument");
  if (slot < 0 || slot >= c->n_slots) return fail(NBP_ERR_RANGE, "slot out of range");
  if (!manifold_ok(manifold)) return fail(NBP_ERR_ARG, "unknown manifold");
  return NBP_OK;
}

nbp_status nbp_belief_write(nbp_ctx *c, int32_t slot, int32_t manifold, const double *pts, int32_t n_pts, const double *bw,
                            const double *ipc) {
  nbp_status rc = belief_args(c, slot, manifold, pts);
  if (rc) return rc;
  if (n_pts < 1) return fail(NBP_ERR_RANGE, "belief: n_pts < 1");
  if (n_pts < c->N && !bw) return fail(NBP_ERR_ARG, "belief: a belief with fewer than N points needs its bandwidth (it is a density, not a point set)");
  std::vector<double> s(c->S);
  pack_belief(c, manifold, pts, n_pts, bw, ipc, s.data());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(c->arena + c->S * slot, s.data(), c->S * 8, hipMemcpyHostToDevice));
  return NBP_OK;
}

nbp_status nbp_belief_read(nbp_ctx *c, int32_t slot, int32_t manifold, double *pts, int32_t *n_pts, double *bw, double *ipc) {
  nbp_status rc = belief_args(c, slot, manifold, pts);
  if (rc) return rc;
  std::vector<double> s(c->S);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(s.data(), c->arena + c->S * slot, c->S * 8, hipMemcpyDeviceToHost));
  unpack_belief(c, manifold, s.data(), pts, n_pts, bw, ipc);
  return NBP_OK;
}

// ---- many beliefs in one call ---------------------------------------------------------------------------------------------
// The beliefs are packed into (unpacked from) a pinned staging buffer and moved with ONE asynchronous copy per run of
// consecutive slots on the library stream: a clique call moves its 3-10 beliefs in one or two copies instead of one
// synchronous copy each, a caller that loads a whole graph moves it in one.
static nbp_status ensure_pin(nbp_ctx *c, size_t doubles) {
  if (doubles <= c->pin_doubles) return NBP_OK;
  if (c->pin) HIPCHK(hipHostFree(c->pin));
  c->pin = nullptr;
  c->pin_doubles = 0;
  const size_t cap = doubles * 2;
  static const bool noncoh = getenv("NBP_PIN_NONCOHERENT") != nullptr;
  HIPCHK(hipHostMalloc((void **)&c->pin, cap * 8, noncoh ? hipHostMallocNonCoherent : hipHostMallocDefault));
  c->pin_doubles = cap;
  return NBP_OK;
}
// The beliefs of a batch are staged IN SLOT ORDER, so that every run of consecutive slots is one copy whatever order the caller
// names them in: resident handles count down from the end of the arena (handle h = slot n_slots - h), and a graph of a thousand
// beliefs written or read through its handles was a thousand copies of 4.9 KB (round 6: 1813 `copyBuffer` launches per queued walk
// of the clique seam, ~3 ms of device time and twice that in launch gaps; tools/exp/seam_walk_trace.sh).  ord[k] = the request
// staged at position k (stable: a slot named twice keeps the caller's order).
static std::vector<int> slot_order(int n, const int32_t *slots) {
  std::vector<int> ord((size_t)(n > 0 ? n : 0));
  for (int i = 0; i < n; i++) ord[(size_t)i] = i;
  bool asc = true;
  for (int i = 1; i < n && asc; i++) asc = slots[i] >= slots[i - 1];
  if (!asc) std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return slots[a] < slots[b]; });
  return ord;
}

nbp_status nbp_belief_write_batch(nbp_ctx *c, int32_t n, const int32_t *slots, const int32_t *manifolds, const double *const *pts,
                                  const int32_t *n_pts, const double *const *bw, const double *const *ipc) {
  if (!c || (n > 0 && (!slots || !manifolds || !pts))) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n; i++) {
    nbp_status rc = belief_args(c, slots[i], manifolds[i], pts[i]);
    if (rc) return rc;
    const int np = n_pts ? n_pts[i] : c->N;
    if (np < 1) return fail(NBP_ERR_RANGE, "belief: n_pts < 1");
    if (np < c->N && !(bw && bw[i])) return fail(NBP_ERR_ARG, "belief: a belief with fewer than N points needs its bandwidth (it is a density, not a point set)");
  }
  HIPCHK(hipStreamSynchronize(c->stream));  // the staging buffer is free again (and so is every slot about to be replaced)
  nbp_status rc = ensure_pin(c, (size_t)n * (size_t)c->S);
  if (rc) return rc;
  const std::vector<int> ord = slot_order(n, slots);
  host_parallel_for(n, 128, [&](int k) {
    const int i = ord[(size_t)k];
    pack_belief(c, manifolds[i], pts[i], n_pts ? n_pts[i] : c->N, bw ? bw[i] : nullptr, ipc ? ipc[i] : nullptr, c->pin + (size_t)k * c->S);
  });
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && slots[ord[(size_t)j]] == slots[ord[(size_t)j - 1]] + 1) j++;
    HIPCHK(hipMemcpyAsync(c->arena + c->S * slots[ord[(size_t)i]], c->pin + (size_t)i * c->S, (size_t)(j - i) * c->S * 8, hipMemcpyHostToDevice, c->stream));
    i = j;
  }
  return NBP_OK;  // stream-ordered: whatever is launched next sees the beliefs; the next use of the staging buffer waits for the copies
}
nbp_status nbp_belief_read_batch(nbp_ctx *c, int32_t n, const int32_t *slots, const int32_t *manifolds, double *const *pts,
                                 int32_t *n_pts, double *const *bw, double *const *ipc) {
  if (!c || (n > 0 && (!slots || !manifolds || !pts))) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n; i++) {
    nbp_status rc = belief_args(c, slots[i], manifolds[i], pts[i]);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  // ascending slots with small gaps (the updated variables of many cliques): reading the gaps along costs less than a copy
  // per run -- one copy of the whole span while it stays under four times the bytes asked for
  const std::vector<int> ord = slot_order(n, slots);  // (in slot order: see slot_order)
  const int32_t s_lo = slots[ord[0]], s_hi = slots[ord[(size_t)n - 1]];
  bool strictly = true;
  for (int k = 1; k < n; k++) strictly &= slots[ord[(size_t)k]] > slots[ord[(size_t)k - 1]];
  const size_t span = strictly ? (size_t)(s_hi - s_lo + 1) : 0;
  if (strictly && span > (size_t)n && span <= 4 * (size_t)n) {
    nbp_status rc = ensure_pin(c, span * (size_t)c->S);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->pin, c->arena + c->S * s_lo, span * c->S * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    host_parallel_for(n, 128, [&](int i) {
      unpack_belief(c, manifolds[i], c->pin + (size_t)(slots[i] - s_lo) * c->S, pts[i], n_pts ? &n_pts[i] : nullptr, bw ? bw[i] : nullptr,
                    ipc ? ipc[i] : nullptr);
    });
    return NBP_OK;
  }
  nbp_status rc = ensure_pin(c, (size_t)n * (size_t)c->S);
  if (rc) return rc;
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && slots[ord[(size_t)j]] == slots[ord[(size_t)j - 1]] + 1) j++;
    HIPCHK(hipMemcpyAsync(c->pin + (size_t)i * c->S, c->arena + c->S * slots[ord[(size_t)i]], (size_t)(j - i) * c->S * 8, hipMemcpyDeviceToHost, c->stream));
    i = j;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  host_parallel_for(n, 128, [&](int k) {
    const int i = ord[(size_t)k];
    unpack_belief(c, manifolds[i], c->pin + (size_t)k * c->S, pts[i], n_pts ? &n_pts[i] : nullptr, bw ? bw[i] : nullptr, ipc ? ipc[i] : nullptr);
  });
  return NBP_OK;
}

// The same transfers WITHOUT a host synchronisation (the asynchronous clique seam): staged in a buffer of the context's pool.
nbp_status nbp_belief_write_batch_async(nbp_ctx *c, int32_t n, const int32_t *slots, const int32_t *manifolds, const double *const *pts,
                                        const int32_t *n_pts, const double *const *bw, const double *const *ipc) {
  if (!c || (n > 0 && (!slots || !manifolds || !pts))) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n; i++) {
    nbp_status rc = belief_args(c, slots[i], manifolds[i], pts[i]);
    if (rc) return rc;
    const int np = n_pts ? n_pts[i] : c->N;
    if (np < 1) return fail(NBP_ERR_RANGE, "belief: n_pts < 1");
    if (np < c->N && !(bw && bw[i])) return fail(NBP_ERR_ARG, "belief: a belief with fewer than N points needs its bandwidth (it is a density, not a point set)");
  }
  nbp_ctx::pin_buf *b = pin_acquire(c, (size_t)n * (size_t)c->S * 8);
  if (!b) return fail(NBP_ERR_HIP, "pinned staging buffer");
  double *pin = (double *)b->p;
  const std::vector<int> ord = slot_order(n, slots);
  host_parallel_for(n, 128, [&](int k) {
    const int i = ord[(size_t)k];
    pack_belief(c, manifolds[i], pts[i], n_pts ? n_pts[i] : c->N, bw ? bw[i] : nullptr, ipc ? ipc[i] : nullptr, pin + (size_t)k * c->S);
  });
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && slots[ord[(size_t)j]] == slots[ord[(size_t)j - 1]] + 1) j++;
    if (hipMemcpyAsync(c->arena + c->S * slots[ord[(size_t)i]], pin + (size_t)i * c->S, (size_t)(j - i) * c->S * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      pin_release_behind_stream(c, b);
      return fail(NBP_ERR_HIP, "hipMemcpyAsync (beliefs in)");
    }
    i = j;
  }
  pin_release_behind_stream(c, b);
  return NBP_OK;
}
struct nbp_read_token {
  nbp_ctx *ctx;
  nbp_ctx::pin_buf *buf;
  hipEvent_t done;
  std::vector<int32_t> slots;
  std::vector<int> ord;  // staging position -> request (slot order: slot_order)
};
nbp_status nbp_belief_read_batch_begin(nbp_ctx *c, int32_t n, const int32_t *slots, nbp_read_token **out) {
  if (!c || !out || (n > 0 && !slots)) return fail(NBP_ERR_ARG, "null argument");
  *out = nullptr;
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n; i++)
    if (slots[i] < 0 || slots[i] >= c->n_slots) return fail(NBP_ERR_RANGE, "slot out of range");
  nbp_read_token *t = new nbp_read_token();
  t->ctx = c;
  t->buf = nullptr;
  t->done = nullptr;
  t->slots.assign(slots, slots + (n > 0 ? n : 0));
  if (hipEventCreateWithFlags(&t->done, hipEventDisableTiming) != hipSuccess) { delete t; return fail(NBP_ERR_HIP, "hipEventCreate"); }
  if (n > 0) {
    t->buf = pin_acquire(c, (size_t)n * (size_t)c->S * 8);
    if (!t->buf) { hipEventDestroy(t->done); delete t; return fail(NBP_ERR_HIP, "pinned staging buffer"); }
    double *pin = (double *)t->buf->p;
    t->ord = slot_order(n, slots);
    const std::vector<int> &ord = t->ord;
    for (int i = 0; i < n;) {
      int j = i + 1;
      while (j < n && slots[ord[(size_t)j]] == slots[ord[(size_t)j - 1]] + 1) j++;
      if (hipMemcpyAsync(pin + (size_t)i * c->S, c->arena + c->S * slots[ord[(size_t)i]], (size_t)(j - i) * c->S * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
        pin_release_behind_stream(c, t->buf);
        hipEventDestroy(t->done);
        delete t;
        return fail(NBP_ERR_HIP, "hipMemcpyAsync (beliefs out)");
      }
      i = j;
    }
  }
  if (hipEventRecord(t->done, c->stream) != hipSuccess) {
    if (t->buf) pin_release_behind_stream(c, t->buf);
    hipEventDestroy(t->done);
    delete t;
    return fail(NBP_ERR_HIP, "hipEventRecord");
  }
  *out = t;
  return NBP_OK;
}
// waits for the copies of `begin` (and everything queued before them), unpacks; the token is gone afterwards, whatever the status
nbp_status nbp_belief_read_batch_end(nbp_read_token *t, const int32_t *manifolds, double *const *pts, int32_t *n_pts, double *const *bw,
                                     double *const *ipc) {
  if (!t) return fail(NBP_ERR_ARG, "null argument");
  nbp_ctx *c = t->ctx;
  const int n = (int)t->slots.size();
  const hipError_t e = hipEventSynchronize(t->done);
  nbp_status rc = NBP_OK;
  if (e != hipSuccess) rc = fail(NBP_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
  else if (n > 0 && (!manifolds || !pts)) rc = fail(NBP_ERR_ARG, "null argument");
  else if (n > 0) {
    const double *pin = (const double *)t->buf->p;
    host_parallel_for(n, 128, [&](int k) {
      const int i = t->ord[(size_t)k];
      if (pts[i]) unpack_belief(c, manifolds[i], pin + (size_t)k * c->S, pts[i], n_pts ? &n_pts[i] : nullptr, bw ? bw[i] : nullptr, ipc ? ipc[i] : nullptr);
    });
  }
  if (t->buf) { t->buf->held = false; t->buf->pending = false; }
  hipEventDestroy(t->done);
  delete t;
  return rc;
}

nbp_status nbp_side_write(nbp_ctx *c, int32_t offset, const int32_t *src, int32_t n) {
  if (!c || !src) return fail(NBP_ERR_ARG, "null argument");
  if (offset < 0 || n < 0 || offset + n > c->side_ints) return fail(NBP_ERR_RANGE, "side buffer range");
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(c->side + offset, src, (size_t)n * 4, hipMemcpyHostToDevice));
  return NBP_OK;
}
nbp_status nbp_side_read(nbp_ctx *c, int32_t offset, int32_t *dst, int32_t n) {
  if (!c || !dst) return fail(NBP_ERR_ARG, "null argument");
  if (offset < 0 || n < 0 || offset + n > c->side_ints) return fail(NBP_ERR_RANGE, "side buffer range");
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(dst, c->side + offset, (size_t)n * 4, hipMemcpyDeviceToHost));
  return NBP_OK;
}

// ---- validation ----------------------------------------------------------------------------------
static nbp_status check_proposals(nbp_ctx *c, const nbp_proposal_desc *d, int n) {
  for (int i = 0; i < n; i++) {
    const nbp_proposal_desc &p = d[i];
    if (!manifold_ok(p.manifold)) return fail(NBP_ERR_ARG, "proposal: unknown manifold");
    if (p.factor_kind < NBP_F_PRIOR || p.factor_kind > NBP_F_PASSTHROUGH) return fail(NBP_ERR_ARG, "proposal: unknown factor kind");
    if (p.factor_kind == NBP_F_PASSTHROUGH) {  // the density is the proposal: a unary factor, two slots, an optional partial mask
      if (p.nvars != 1 || p.sfidx != 0) return fail(NBP_ERR_ARG, "proposal: a pass-through prior is unary");
      if (p.out_slot < 0 || p.out_slot >= c->n_slots) return fail(NBP_ERR_RANGE, "proposal: out_slot");
      for (int k = 0; k < 2; k++)
        if (p.var_slot[k] < 0 || p.var_slot[k] >= c->n_slots) return fail(NBP_ERR_RANGE, "proposal: var_slot");
      if (p.partial_mask < 0 || p.partial_mask >= (1 << manifold_dim_h(p.manifold))) return fail(NBP_ERR_RANGE, "proposal: partial_mask");
      if (p.has_multihypo || p.nullhypo != 0.0) return fail(NBP_ERR_ARG, "proposal: a pass-through prior takes no multihypo / nullhypo (evalFactor is bypassed)");
      if (p.keep_count < 0 || p.keep_count > 2) return fail(NBP_ERR_RANGE, "proposal: keep_count");
      continue;
    }
    if (p.nvars < 1 || p.nvars > NBP_MAXV) return fail(NBP_ERR_RANGE, "proposal: nvars");
    if (p.sfidx < 0 || p.sfidx >= p.nvars) return fail(NBP_ERR_RANGE, "proposal: sfidx");
    if (p.ncomp < 1 || p.ncomp > NBP_MAXC) return fail(NBP_ERR_RANGE, "proposal: ncomp");
    if (p.inflate_cycles < 0 || p.inflate_cycles > 8) return fail(NBP_ERR_RANGE, "proposal: inflate_cycles");
    if (p.out_slot < 0 || p.out_slot >= c->n_slots) return fail(NBP_ERR_RANGE, "proposal: out_slot");
    int nv = (p.factor_kind == NBP_F_MSGPRIOR) ? 2 : p.nvars;
    for (int k = 0; k < nv; k++)
      if (p.var_slot[k] < 0 || p.var_slot[k] >= c->n_slots) return fail(NBP_ERR_RANGE, "proposal: var_slot");
    if ((p.factor_kind == NBP_F_PRIOR || p.factor_kind == NBP_F_MSGPRIOR) && p.nvars != 1)
      return fail(NBP_ERR_ARG, "proposal: priors are unary");
    if (p.factor_kind >= NBP_F_LINREL && p.nvars < 2) return fail(NBP_ERR_ARG, "proposal: relative factor needs >= 2 variables");
    if (p.factor_kind >= NBP_F_LINREL && !p.has_multihypo && p.nvars != 2)
      return fail(NBP_ERR_ARG, "proposal: binary mechanics without multihypo needs nvars == 2");
    if (p.factor_kind == NBP_F_CIRCULAR && p.manifold != NBP_CIRCULAR) return fail(NBP_ERR_ARG, "CircularCircular needs Circular variables");
    if (p.factor_kind == NBP_F_SE2 && p.manifold != NBP_SE2) return fail(NBP_ERR_ARG, "SE2 factor needs SE2 variables");
    if (p.factor_kind == NBP_F_LINREL && p.manifold > NBP_EUCLID3) return fail(NBP_ERR_ARG, "LinearRelative needs Euclid variables");
    if (p.mhidx_in >= 0 && p.mhidx_in + c->N > c->side_ints) return fail(NBP_ERR_RANGE, "proposal: mhidx_in");
    if (p.mhidx_out >= 0 && p.mhidx_out + c->N > c->side_ints) return fail(NBP_ERR_RANGE, "proposal: mhidx_out");
    if (p.partial_mask) {
      const int D = manifold_dim_h(p.manifold);
      if (p.partial_mask < 0 || p.partial_mask >= (1 << D)) return fail(NBP_ERR_RANGE, "proposal: partial_mask");
      if (p.factor_kind == NBP_F_LINREL) {
        if (__builtin_popcount(p.partial_mask) > 2) return fail(NBP_ERR_ARG, "partial LinearRelative: one or two partial coordinates");
      } else if (p.factor_kind != NBP_F_PRIOR && p.factor_kind != NBP_F_SE2)
        return fail(NBP_ERR_ARG, "proposal: partial_mask is supported for Prior, LinearRelative and SE(2) ManifoldFactor factors");
    }
    if (p.meas_kde) {
      if (p.meas_kde < 0 || p.meas_kde > c->n_slots) return fail(NBP_ERR_RANGE, "proposal: meas_kde");
      if (p.factor_kind != NBP_F_LINREL && p.factor_kind != NBP_F_CIRCULAR && p.factor_kind != NBP_F_SE2)
        return fail(NBP_ERR_ARG, "proposal: a KDE measurement needs a relative factor whose measurement lives on the variable's manifold");
      if (p.partial_mask) return fail(NBP_ERR_ARG, "proposal: a KDE measurement cannot be partial");
    }
    if (p.has_multihypo) {
      int ncert = 0;
      for (int k = 0; k < p.nvars; k++) ncert += (p.multihypo[k] == 0.0);
      if (ncert != 1) return fail(NBP_ERR_ARG, "proposal: multihypo needs exactly one certain variable (binary mechanics)");
    }
  }
  return NBP_OK;
}
static nbp_status check_products(nbp_ctx *c, const nbp_product_desc *d, int n) {
  for (int i = 0; i < n; i++) {
    const nbp_product_desc &p = d[i];
    if (!manifold_ok(p.manifold)) return fail(NBP_ERR_ARG, "product: unknown manifold");
    if (p.nfactors < 1 || p.nfactors > NBP_MAXF) return fail(NBP_ERR_RANGE, "product: nfactors");
    if (p.niter < 1 || p.niter > 8) return fail(NBP_ERR_RANGE, "product: niter (1 .. 8)");
    if (p.out_slot < 0 || p.out_slot >= c->n_slots) return fail(NBP_ERR_RANGE, "product: out_slot");
    for (int k = 0; k < p.nfactors; k++)
      if (p.in_slot[k] < 0 || p.in_slot[k] >= c->n_slots) return fail(NBP_ERR_RANGE, "product: in_slot");
    if (p.labels_out >= 0 && p.labels_out + c->N * p.nfactors > c->side_ints) return fail(NBP_ERR_RANGE, "product: labels_out");
    {
      const int D = manifold_dim_h(p.manifold);
      bool any = false;
      for (int k = 0; k < p.nfactors; k++) {
        if (p.in_partial[k] >= (1 << D)) return fail(NBP_ERR_RANGE, "product: in_partial");
        any |= (p.in_partial[k] != 0);
      }
      if (any && D < 2) return fail(NBP_ERR_ARG, "product: partial densities need a variable of dimension >= 2");
      if (any && (p.old_slot < 0 || p.old_slot >= c->n_slots)) return fail(NBP_ERR_RANGE, "product: old_slot");
    }
  }
  return NBP_OK;
}
static nbp_status check_copies(nbp_ctx *c, const nbp_copy_desc *d, int n) {
  for (int i = 0; i < n; i++)
    if (d[i].src_slot < 0 || d[i].src_slot >= c->n_slots || d[i].dst_slot < 0 || d[i].dst_slot >= c->n_slots)
      return fail(NBP_ERR_RANGE, "copy: slot out of range");
  return NBP_OK;
}

// ---- launches --------------------------------------------------------------------------------------
static nbp_status tic(nbp_ctx *c, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
  if (!c->timing) return NBP_OK;
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  HIPCHK(hipEventRecord(a, c->stream));
  v.emplace_back(a, b);
  return NBP_OK;
}
static nbp_status toc(nbp_ctx *c, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
  if (!c->timing) return NBP_OK;
  HIPCHK(hipEventRecord(v.back().second, c->stream));
  return NBP_OK;
}

// 0, or the class of a batch whose relative factors are all full (non-partial) factors of one kind on one manifold, with
// everything else in it (priors, message priors, pass-through densities) on that manifold as well:
// 1 LinearRelative / Euclid(2), 3 LinearRelative / Euclid(3)
#define NBP_CLS_SIMPLE 256
static int proposals_uniform_class(const nbp_proposal_desc *d, int n) {
  if (n <= 0) return 0;
  const int M = d[0].manifold;
  // class = the manifold code: LinearRelative on Euclid(2) / Euclid(3), CircularCircular on the circle, ManifoldFactor on SE(2)
  const int cls = M == NBP_EUCLID2 ? 1 : (M == NBP_EUCLID3 ? 3 : (M == NBP_CIRCULAR ? 4 : (M == NBP_SE2 ? 5 : 0)));
  const int want = M == NBP_CIRCULAR ? NBP_F_CIRCULAR : (M == NBP_SE2 ? NBP_F_SE2 : NBP_F_LINREL);
  if (!cls) return 0;
  // "simple" (bit 8, NBP_CLS_SIMPLE): every particle of every proposal on the factor's one hypothesis -- no multihypo, no
  // nullhypo, no injected hypothesis indices, binary relatives: what the one-wave-per-proposal kernels are written for
  bool simple = true;
  for (int i = 0; i < n; i++) {
    if (d[i].manifold != M) return 0;
    const int k = d[i].factor_kind;
    if (d[i].partial_mask) return 0;  // partial priors and partial relatives run the generic kernel
    if (d[i].has_multihypo || d[i].nullhypo != 0.0 || d[i].mhidx_in >= 0) simple = false;
    if (k == NBP_F_PRIOR || k == NBP_F_MSGPRIOR || k == NBP_F_PASSTHROUGH) continue;
    if (k != want) return 0;
    if (d[i].nvars != 2) simple = false;
  }
  // (a batch of priors / message priors alone runs its manifold's instance as well)
  return cls | (simple ? NBP_CLS_SIMPLE : 0);
}
static nbp_status launch_proposals(nbp_ctx *c, const nbp_proposal_desc *dev, int n, int cls = 0) {
  if (n <= 0) return NBP_OK;
  nbp_status rc = tic(c, c->ev[0]);
  if (rc) return rc;
  (void)hipGetLastError();  // clear stale, unrelated errors
  const bool simple = (cls & NBP_CLS_SIMPLE) != 0;
  cls &= NBP_CLS_SIMPLE - 1;
  // chip-filling launches of simple Euclidean batches: one wave per proposal, no workgroup barrier (nbp_kernels.h,
  // proposal_wave_body); rows of up to five waves (N <= 320)
  const int wave_min = c->prop_wave_min >= 0 ? c->prop_wave_min : (cls == 1 ? 900 : (c->N > 256 ? 1500 : (1 << 30)));
  if (simple && n >= wave_min && (cls == 1 || cls == 3) && c->N <= (cls == 1 ? 256 : 320)) {
    auto *wk = cls == 1 ? nbp_proposal_wave_kernel_lin2 : (c->N <= 256 ? nbp_proposal_wave_kernel_lin3 : nbp_proposal_wave_kernel_lin3n5);
    hipLaunchKernelGGL(wk, dim3((n + NBP_PW_WAVES - 1) / NBP_PW_WAVES), dim3(64 * NBP_PW_WAVES), nbp_proposal_wave_lds_bytes(c->N, cls == 1 ? 2 : 3),
                       c->stream, dev, n, c->arena, c->N, c->S, c->side, c->counters);
    HIPCHK(hipGetLastError());
    return toc(c, c->ev[0]);
  }
  auto *kern = cls == 1 ? nbp_proposal_kernel_lin2 : (cls == 3 ? nbp_proposal_kernel_lin3 : (cls == 4 ? nbp_proposal_kernel_circ :
               (cls == 5 ? nbp_proposal_kernel_se2 : nbp_proposal_kernel)));
  hipLaunchKernelGGL(kern, dim3(n), dim3(c->Npad), nbp_proposal_lds_bytes(c->N), c->stream, dev, c->arena, c->N, c->Npad, c->S, c->side,
                     c->counters);
  HIPCHK(hipGetLastError());
  return toc(c, c->ev[0]);
}
static nbp_status ensure_ws(nbp_ctx *c, int nprod, int kdF) {
  const size_t need = (size_t)nprod * (size_t)kdF * nbp_kd_ws_doubles(c->N);
  if (need <= c->ws_doubles) return NBP_OK;
  // inside a two-stream round c->ws is an INTERIOR pointer of the workspace (the second half's share): it must never be
  // freed or re-allocated here -- the round is sized at finalize, a shortfall now is a planning error, not a reason to grow
  if (c->geom_n) return fail(NBP_ERR_RANGE, "KD workspace too small inside a two-stream round (sized at nbp_program_finalize)");
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->ws) HIPCHK(hipFree(c->ws));
  c->ws = nullptr;
  c->ws_gen++;  // captured graphs hold the old pointer as a kernel argument: they are re-captured before their next replay
  c->ws_doubles = need + need / 4 + 16 * nbp_kd_ws_doubles(c->N);
  HIPCHK(hipMalloc(&c->ws, c->ws_doubles * 8));
  return NBP_OK;
}
static nbp_status ensure_gstats(nbp_ctx *c, size_t need) {
  if (need <= c->gstats_doubles) return NBP_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->gstats) HIPCHK(hipFree(c->gstats));
  c->gstats = nullptr;
  c->ws_gen++;
  c->gstats_doubles = need + need / 4;
  HIPCHK(hipMalloc(&c->gstats, c->gstats_doubles * 8));
  return NBP_OK;
}

// sample groups per product: spread a product over G workgroups when the launch cannot fill the chip

// Helper lanes per particle for the LCV / KD launches: P = 4 (1024-lane workgroups) minimises the
// latency of a single fit when the launch cannot fill the chip; P = 2 (512 lanes) lets four
// workgroups share a CU so that the barrier/combine phases of one overlap the pair loop of the
// others when there are many fits (throughput mode).
// throughput-mode fits whose workgroup is 4k + 1 waves (N = 257 .. 320: config 5's N = 300) run the five-waves-per-SIMD
// instances of the fit kernels: four such workgroups per CU, five waves on every SIMD, instead of three (nbp_kernels.h)
static bool rows_of_4k_plus_1_waves(const nbp_ctx *c, int P) {
  static const bool off = getenv("NBP_NO_W5_FITS") != nullptr;
  return !off && P == 1 && ((c->Npad >> 6) & 3) == 1 && c->Npad > 64;
}
static int lcv_helpers(nbp_ctx *c, int nblocks) {
  static const int p2_min = getenv("NBP_LCV_P2_MIN") ? atoi(getenv("NBP_LCV_P2_MIN")) : 256;
  static const int p1_min = getenv("NBP_LCV_P1_MIN") ? atoi(getenv("NBP_LCV_P1_MIN")) : 4 * 256;
  int P = c->P;
  if (nblocks > p2_min && P > 2) P = 2;
  if (nblocks > p1_min) P = 1;
  return P;
}

// nbp_prep_kernel: pending bandwidth fits + KD builds of this product batch, one launch
static int coords_of(const int32_t *manis, size_t n) {
  int cds = 0;
  for (size_t i = 0; i < n; i++) cds += manifold_dim_h(manis[i]);
  return cds;
}
static void product_geometry(nbp_ctx *c, int n, int *HL, int *wpb, int *G, int mani = 0);
// the sin / cos rows of the node statistics are part of a product launch's LDS unless it runs a single-manifold throughput
// kernel of a manifold without a circular coordinate (product_kernel_uniform lays its LDS out by the same rule)
static inline bool product_lays_circ(int HL, int mani) { return HL >= 8 || !(mani == NBP_EUCLID1 || mani == NBP_EUCLID2 || mani == NBP_EUCLID3); }
// The product launch of a batch takes the node sums from the sorted coordinates itself (the _xs kernels: 4 KB instead of
// 33 KB of KD workspace per density through HBM) when the batch runs a single-manifold throughput kernel and every
// product has at most NBP_FUSED_MAXF densities; the prep launch in front then leaves the node sums out.
static bool products_use_xs(nbp_ctx *c, int n, int maxFD, int mani) {
  static const bool off = getenv("NBP_NO_XS_PRODUCTS") != nullptr;
  if (off || mani == 0 || n <= 0) return false;
  int HL, wpb, G;
  product_geometry(c, n, &HL, &wpb, &G, mani);
  const int F = maxFD / 4, D = maxFD % 4;
  if (HL > 4 || F > NBP_FUSED_MAXF) return false;
  return nbp_product_lds_bytes(F, D, c->N, wpb * 64 / HL, false, (size_t)2 * 2 * wpb * 64, product_lays_circ(HL, mani)) + 8 +
             nbp_product_xs_doubles(F, D, c->N) * 8 <= 150 * 1024;
}
// the rendezvous areas of the next launch of speculative fits, blanked on the library's stream (nbp_spec_blank_kernel)
static void blank_spec_areas(nbp_ctx *c, int jobs) {
#ifdef NBP_SPEC_BLANK_MEMSET  // experiment (tools/exp/concurrency_probe3.sh): the hipMemsetAsync this kernel replaced
  (void)hipMemsetAsync(c->spec, 0xFF, sizeof(nbp_spec_area) * 3 * (size_t)jobs, c->stream);
#else
  const int words = (int)(sizeof(nbp_spec_area) / 8) * 3 * jobs;
  hipLaunchKernelGGL(nbp_spec_blank_kernel, dim3((words + 255) / 256), dim3(256), 0, c->stream, (unsigned long long *)c->spec, words);
#endif
}

static nbp_status launch_prep(nbp_ctx *c, const int32_t *bw_slots, const int32_t *bw_manis, int nbw,
                              const nbp_product_desc *dev, int n, int maxFD, int coords = -1, int mani = 0) {
  if (coords < 0) coords = 3 * nbw;
  nbp_status rc = ensure_ws(c, n, maxFD / 4);
  if (rc) return rc;
  rc = tic(c, c->ev[1]);
  if (rc) return rc;
  const int P = lcv_helpers(c, c->geom_blocks ? c->geom_blocks : 2 * nbw + 2 * n);
  size_t lds = nbp_kd_lds_bytes(3, c->N, c->Npad, P);
  if (nbw > 0 && nbp_bandwidth_lds_bytes(c->N, c->Npad, P) > lds) lds = nbp_bandwidth_lds_bytes(c->N, c->Npad, P);
  (void)hipGetLastError();
  const int kdF = (maxFD / 4) | (products_use_xs(c, n, maxFD, mani) ? NBP_KD_NOSTATS : 0);
  const int nkd = n * (maxFD / 4);  // KD-build workgroups
  // latency mode: a handful of fits, the rest of the chip idle -> NBP_SPEC_K workgroups per fit
  // 3 workgroups per fit (two iterations per rendezvous) when the whole launch is resident at once (7 / three on request)
  int depth = 0;
  if (c->spec_on && nbw > 0 && nbw <= NBP_SPEC_MAXJOBS && !c->geom_n) {
    if (c->spec_depth3 && coords * 7 + nkd <= NBP_SPEC_MAXBLOCKS) depth = 3;
    else if (coords * 3 + nkd <= NBP_SPEC_MAXBLOCKS) depth = 2;
  }
  const bool spec = depth > 0;
  const int KS = spec ? (1 << depth) - 1 : 1;
  if (spec) blank_spec_areas(c, nbw);
  if (depth == 3)
    hipLaunchKernelGGL(nbp_prep_kernel_spec<3>, dim3(3 * nbw * KS + nkd), dim3(P * c->Npad), lds, c->stream, bw_slots, bw_manis, nbw,
                       dev, n, kdF, c->arena, c->ws, c->N, c->Npad, c->S, c->T, c->counters, c->spec);
  else if (depth == 2)
    hipLaunchKernelGGL(nbp_prep_kernel_spec<2>, dim3(3 * nbw * KS + nkd), dim3(P * c->Npad), lds, c->stream, bw_slots, bw_manis, nbw,
                       dev, n, kdF, c->arena, c->ws, c->N, c->Npad, c->S, c->T, c->counters, c->spec);
  else
    hipLaunchKernelGGL(rows_of_4k_plus_1_waves(c, P) ? nbp_prep_kernel_w5 : nbp_prep_kernel, dim3(3 * nbw + nkd), dim3(P * c->Npad), lds, c->stream,
                       bw_slots, bw_manis, nbw, dev, n, kdF, c->arena, c->ws, c->N, c->Npad, c->S, c->T, c->counters);
  HIPCHK(hipGetLastError());
  return toc(c, c->ev[1]);
}

// product launch geometry: HL helper lanes per sample (64/HL samples per wave), workgroups of `wpb` <= 8
// waves, grid.y = G workgroups per product.  Latency mode (the launch cannot fill the chip): HL = 32 (fewer than 16
// products; NBP_PRODUCT_HL32_MAX) or 8 and several small workgroups per product; throughput mode: HL = 2 so that one
// workgroup covers all samples and the node statistics of a product are computed once.
// `mani`: the manifold of a single-manifold batch (0: mixed; < 0: size for the widest workgroup any kernel takes)
static void product_geometry(nbp_ctx *c, int n, int *HL, int *wpb, int *G, int mani) {
  if (c->geom_n) n = c->geom_n;  // one half of a two-stream round: the geometry of the whole batch
  static const int hl2_min = getenv("NBP_PRODUCT_HL2_MIN") ? atoi(getenv("NBP_PRODUCT_HL2_MIN")) : 192;
  static const int hl32_max = getenv("NBP_PRODUCT_HL32_MAX") ? atoi(getenv("NBP_PRODUCT_HL32_MAX")) : 15;
  static const int hl4_min = getenv("NBP_PRODUCT_HL4_MIN") ? atoi(getenv("NBP_PRODUCT_HL4_MIN")) : 80;  // (48 until round 4: 66-product rounds of config 2 run 147 instead of 207 us with eight helpers)
  *HL = n >= hl2_min ? 2 : (n >= hl4_min ? 4 : (n >= 16 ? 8 : (n > hl32_max ? 16 : 32)));
  // latency geometries: workgroups of FOUR waves (one per SIMD of their CU; six until round 4: config 3's products 25.9 -> 24.4 ms,
  // config 2's 7.93 -> 7.77, config 4's 110.6 -> 107.8; two / three / five waves measured worse than four)
  static const int lat_cap = getenv("NBP_PRODUCT_LAT_CAP") ? atoi(getenv("NBP_PRODUCT_LAT_CAP")) : 4;
  const int SW = 64 / *HL, waves = (c->N + SW - 1) / SW, cap = (*HL >= 8) ? lat_cap : 8;
  int g = (waves + cap - 1) / cap;
  *wpb = (waves + g - 1) / g;
  *G = (waves + *wpb - 1) / *wpb;
  // Throughput geometries: workgroups of EIGHT waves when the even split is not a multiple of four (N = 300 at two helper
  // lanes: 10 waves of samples -> two workgroups of 8 instead of two of 5; N = 200: 7 -> 8).  A workgroup whose waves do not
  // divide over the four SIMDs leaves one of them a wave short and makes its own waves wait for each other at the level
  // barriers; the idle waves of the rounder workgroup have no samples and cost a few barriers.  Config 5: products 174.9 ->
  // 164.2 ms per solve; config 2: unchanged (7.91 / 7.92 ms).  Workgroups of four (more of them per product, every one staging
  // the node statistics again) measured worse: 215.7 ms.  NBP_PRODUCT_WPB8=0: the even split.
  static const bool wpb8 = !(getenv("NBP_PRODUCT_WPB8") && atoi(getenv("NBP_PRODUCT_WPB8")) == 0);
  if (wpb8 && *HL <= 4 && (*wpb & 3)) {
    *wpb = 8;
    *G = (waves + 7) / 8;
  }
  // The circle: workgroups of FOUR waves in the throughput geometries too.  Its instances hold 162 VGPRs = three waves per SIMD =
  // twelve wave slots per CU, of which one workgroup of eight leaves four empty and three workgroups of four none; the second
  // staging per product is cheap in one dimension.  Config 3: products 24.4 -> 22.8 ms, 52.8 -> 51.3 ms per solve.  (SE(2), two
  // waves per SIMD and a costly staging: 107 -> 126 ms -- it keeps eight.)  NBP_PRODUCT_THR_WPB overrides for every manifold.
  static const int thr_wpb = getenv("NBP_PRODUCT_THR_WPB") ? atoi(getenv("NBP_PRODUCT_THR_WPB")) : 0;
  const int tw = thr_wpb > 0 ? thr_wpb : (mani == NBP_CIRCULAR ? 4 : 0);
  if (tw > 0 && *HL <= 4) {
    *wpb = tw;
    *G = (waves + tw - 1) / tw;
  }
  // a product whose samples need more than eight waves: ONE workgroup of up to sixteen (a multiple of four) where the kernel takes
  // it -- the Euclidean instances, NBP_PROD_WIDE in nbp_kernels.h -- instead of a full workgroup and a nearly empty one, each
  // staging the node statistics (N = 300 at two helper lanes: ten waves of samples; config 5's products 162 -> 122 ms per solve)
  static const int wpb_max = getenv("NBP_PRODUCT_WPB_MAX") ? atoi(getenv("NBP_PRODUCT_WPB_MAX")) : 16;
  const bool wide = mani < 0 || NBP_PROD_WIDE(mani);
  // (two helper lanes only, i.e. launches that fill the chip: a launch of 80-191 products at four helper lanes leaves CUs idle, and
  //  there two workgroups per product on two CUs beat one on one -- config 2: 18.34 against 18.58 ms)
  if (wide && wpb_max > 8 && *HL == 2 && waves > 8) {
    const int g2 = (waves + wpb_max - 1) / wpb_max;
    *wpb = (((waves + g2 - 1) / g2) + 3) & ~3;
    *G = (waves + *wpb - 1) / *wpb;
  }
}
// LDS budget of a product workgroup: beyond it the node statistics live in global memory ("big")
static const size_t NBP_PRODUCT_LDS_CAP = 150 * 1024;
typedef void (*nbp_product_fn)(const nbp_product_desc *, double *, const double *, int, double *, int, int64_t, int32_t *, nbp_levels);
// the kernel of a product launch: HL helper lanes per sample; `mani` != 0: every multi-density product of the batch lives
// on that manifold and has only full inputs (the throughput variants then run the single-instantiation kernels)
// `w1`: the launch has at most one workgroup (of at most four waves) per CU -- the latency instances that own their SIMDs
static nbp_product_fn product_kernel_for(int HL, int mani, bool xs = false, bool w1 = false) {
  if (HL == 32) return w1 ? nbp_product_kernel_y32_w1 : nbp_product_kernel_y32;
  if (HL == 16) return w1 ? nbp_product_kernel_x16_w1 : nbp_product_kernel_x16;
  if (HL == 8) return w1 ? nbp_product_kernel_l8_w1 : nbp_product_kernel_l8;
  if (xs) {
    switch (mani * 8 + HL) {
    case NBP_EUCLID1 * 8 + 4: return nbp_product_kernel_m4_e1_xs;
    case NBP_EUCLID2 * 8 + 4: return nbp_product_kernel_m4_e2_xs;
    case NBP_EUCLID3 * 8 + 4: return nbp_product_kernel_m4_e3_xs;
    case NBP_CIRCULAR * 8 + 4: return nbp_product_kernel_m4_ci_xs;
    case NBP_SE2 * 8 + 4: return nbp_product_kernel_m4_se_xs;
    case NBP_EUCLID1 * 8 + 2: return nbp_product_kernel_t2_e1_xs;
    case NBP_EUCLID2 * 8 + 2: return nbp_product_kernel_t2_e2_xs;
    case NBP_EUCLID3 * 8 + 2: return nbp_product_kernel_t2_e3_xs;
    case NBP_CIRCULAR * 8 + 2: return nbp_product_kernel_t2_ci_xs;
    default: return nbp_product_kernel_t2_se_xs;
    }
  }
  if (HL == 4) {
    switch (mani) {
    case NBP_EUCLID1: return nbp_product_kernel_m4_e1;
    case NBP_EUCLID2: return nbp_product_kernel_m4_e2;
    case NBP_EUCLID3: return nbp_product_kernel_m4_e3;
    case NBP_CIRCULAR: return nbp_product_kernel_m4_ci;
    case NBP_SE2: return nbp_product_kernel_m4_se;
    default: return nbp_product_kernel_m4;
    }
  }
  switch (mani) {
  case NBP_EUCLID1: return nbp_product_kernel_t2_e1;
  case NBP_EUCLID2: return nbp_product_kernel_t2_e2;
  case NBP_EUCLID3: return nbp_product_kernel_t2_e3;
  case NBP_CIRCULAR: return nbp_product_kernel_t2_ci;
  case NBP_SE2: return nbp_product_kernel_t2_se;
  default: return nbp_product_kernel_t2;
  }
}
// 0 unless all products with more than one density share a manifold and none has a partial input
static int products_uniform_manifold(const nbp_product_desc *d, int n) {
  int mani = -1;
  // a batch of pass-through products only (single densities: AMP returns them) runs the copy path of whichever instance:
  // the first product's manifold picks a single-manifold kernel instead of the all-manifold one (308 B of scratch per lane)
  bool multi = false;
  for (int i = 0; i < n; i++) multi |= d[i].nfactors > 1;
  if (!multi && n > 0) return d[0].manifold;
  for (int i = 0; i < n; i++) {
    if (d[i].nfactors <= 1) continue;
    for (int j = 0; j < d[i].nfactors; j++)
      if (d[i].in_partial[j]) return 0;
    if (mani == -1) mani = d[i].manifold;
    else if (mani != d[i].manifold) return 0;
  }
  return mani > 0 ? mani : 0;
}
// ONE answer to "do the node statistics of this product launch fit the LDS" -- asked by the launch itself, by the presizing of
// its scratch area and by the two-stream guard of nbp_program_finalize, with the same terms: the launch's geometry, the chunk
// sums the throughput geometries keep in LDS (two chunks per lane at least) and the sin / cos rows of a circular coordinate.
// (Through round 5 the guard and the presizing left the chunk sums out: a round of Euclid(3) products at N = 300 with five
//  densities passed the guard, was pipelined, and its halves -- laid out by the geometry of the whole batch, whose kernels
//  have no scratch path -- ran past the LDS the launch had allocated.)
static bool product_is_big(nbp_ctx *c, int n, int maxFD, int mani) {
  int HL, wpb, G;
  product_geometry(c, n, &HL, &wpb, &G, mani);
  const int F = maxFD / 4, D = maxFD % 4;  // maxFD encodes the largest (F, D) of the batch as F*4 + D
  return nbp_product_lds_bytes(F, D, c->N, wpb * 64 / HL, false, HL <= 4 ? (size_t)2 * 2 * wpb * 64 : 0, product_lays_circ(HL, mani)) > NBP_PRODUCT_LDS_CAP;
}
static nbp_status launch_products(nbp_ctx *c, const nbp_product_desc *dev, int n, int maxFD, int mani = 0) {
  if (n <= 0) return NBP_OK;
  int HL, wpb, G;
  product_geometry(c, n, &HL, &wpb, &G, mani);
  const int F = maxFD / 4, D = maxFD % 4;
  bool big = product_is_big(c, n, maxFD, mani);
  // (the halves of a two-stream round take the geometry of the whole batch, whose throughput kernels have no scratch path, and
  //  share one scratch area: nbp_program_finalize pipelines no round whose products are big -- refused here should it ever)
  if (big && c->geom_n) return fail(NBP_ERR_RANGE, "product: node statistics beyond the LDS inside a two-stream round");
  if (big && HL != 8) {  // many densities: small sample groups keep the label table in LDS
    product_geometry(c, 16, &HL, &wpb, &G);
  }
  const int TB = wpb * 64, SPB = TB / HL;
  const bool circ = product_lays_circ(HL, mani);
  const bool xs = !big && products_use_xs(c, n, maxFD, mani);
  const size_t xsb = xs ? 8 + nbp_product_xs_doubles(F, D, c->N) * 8 : 0;
  // chunks per helper range of a throughput launch (pass 2 of a draw rescans ONE chunk): as many as the LDS takes while two
  // workgroups still share a CU (80 KB each), else as many as one workgroup's budget takes
  int nch = 0;
  if (HL <= 4 && !big) {
    static const int nch_env = getenv("NBP_PRODUCT_NCH") ? atoi(getenv("NBP_PRODUCT_NCH")) : 0;
    static const size_t half = getenv("NBP_PRODUCT_NCH_KB") ? (size_t)atoi(getenv("NBP_PRODUCT_NCH_KB")) * 1024 : 80 * 1024;
    auto lds_for = [&](int k) { return nbp_product_lds_bytes(F, D, c->N, SPB, false, (size_t)k * 2 * TB, circ) + xsb; };
    // (8 and 4 cost the same, 2 is ~5 % slower, 1 -- a range is its own chunk, pass 2 rescans all of it -- slower again; but ONE
    //  workgroup per CU costs a launch of 257 .. 512 workgroups a second generation: a mixed launch of two- and three-density
    //  products, sized by the three, took 736 us where two resident workgroups take ~600: profiles/r05_product_chunks_by_launch_size.txt)
    nch = 0;
    const int cand[4] = {8, 4, 2, 1};
    for (int k : cand)
      if (lds_for(k) <= half) { nch = k; break; }
    if (!nch) {
      nch = 2;
      for (int k : cand)
        if (lds_for(k) <= NBP_PRODUCT_LDS_CAP) { nch = k; break; }
    }
    if (nch_env >= 1 && nch_env <= 15 && lds_for(nch_env) <= 160 * 1024) nch = nch_env;
  }
  size_t lds = nbp_product_lds_bytes(F, D, c->N, SPB, big, (size_t)nch * 2 * TB, circ) + xsb;
  if (lds > 160 * 1024) return fail(NBP_ERR_RANGE, "product: too many densities for the LDS label table");
  // resident levels (NBP_PROD_ALL_LEVELS): the statistics of every tree level staged once, no barrier between the levels
  // of the Gibbs walk -- where two workgroups of the launch still fit a CU's 160 KB
  static const size_t all_cap = getenv("NBP_PRODUCT_ALL_LEVELS_KB") ? (size_t)atoi(getenv("NBP_PRODUCT_ALL_LEVELS_KB")) * 1024 : 76 * 1024;
  int flagsF = F | (nch << NBP_PROD_NCH_SHIFT);
  // (the latency geometries only: a lone product saves nine round trips to the KD workspace and eighteen barriers, 126 -> 116 us;
  //  a chip-filling launch gains nothing -- NBP_PRODUCT_ALL_LEVELS_HL = 2 switches it on there too)
  static const int all_hl = getenv("NBP_PRODUCT_ALL_LEVELS_HL") ? atoi(getenv("NBP_PRODUCT_ALL_LEVELS_HL")) : 8;
  if (!big && HL >= all_hl) {
    const int TOT = c->T.off[c->T.L] + c->T.cnt[c->T.L];
    // (with the chunk sums of the throughput geometries, which the kernel lays out behind the statistics: without them the
    //  experiment NBP_PRODUCT_ALL_LEVELS_HL <= 4 ran config 3's products past their LDS -- non-finite posteriors)
    const size_t lds_all = product_lds_layout(F, D, c->N, SPB, false, nullptr, nullptr, TOT, HL <= 4 ? (size_t)nch * 2 * TB : 0, circ) +
                           (xs ? 8 + nbp_product_xs_doubles(F, D, c->N) * 8 : 0);
    if (lds_all <= all_cap) { lds = lds_all; flagsF |= NBP_PROD_ALL_LEVELS; }
  }
  nbp_status rc = NBP_OK;
  double *gs = nullptr;
  if (big) {
    rc = ensure_gstats(c, (size_t)n * G * nbp_product_gstats_doubles(F, 3, c->N));  // (the kernel's stride: largest F of the launch, D = 3)
    if (rc) return rc;
    gs = c->gstats;
  }
  rc = tic(c, c->ev[2]);
  if (rc) return rc;
  (void)hipGetLastError();
  static const int w1_max = getenv("NBP_PRODUCT_W1_MAX") ? atoi(getenv("NBP_PRODUCT_W1_MAX")) : 256;  // workgroups (= CUs of the chip)
  const bool w1 = HL >= 8 && TB <= 256 && (long)n * G <= w1_max && !c->geom_n;
  hipLaunchKernelGGL(product_kernel_for(HL, mani, xs, w1), dim3(n, G), dim3(TB), lds, c->stream, dev, c->arena, c->ws, flagsF, gs, c->N, c->S, c->side, c->T);
  HIPCHK(hipGetLastError());
  return toc(c, c->ev[2]);
}

// the allocations launch_prep / launch_products would make on demand for a batch of n products
static nbp_status presize_products(nbp_ctx *c, int n, int maxFD, int mani = 0) {
  nbp_status rc = ensure_ws(c, n, maxFD / 4);
  if (rc) return rc;
  // the launch's own decision (product_is_big), or the widest workgroup any kernel may take (mani = -1: its larger label table
  // decides about the scratch) -- whichever asks for the scratch area gets it allocated here, outside the replayed region
  if (product_is_big(c, n, maxFD, mani) || product_is_big(c, n, maxFD, -1)) {
    int HL, wpb, G;
    product_geometry(c, n, &HL, &wpb, &G, mani);
    if (HL != 8) product_geometry(c, 16, &HL, &wpb, &G);
    const int F = maxFD / 4;
    rc = ensure_gstats(c, (size_t)n * G * nbp_product_gstats_doubles(F, 3, c->N));
  }
  return rc;
}

static nbp_status launch_bandwidth(nbp_ctx *c, const int32_t *dev_slots, const int32_t *dev_manis, int n, int coords = -1) {
  if (n <= 0) return NBP_OK;
  if (coords < 0) coords = 3 * n;
  nbp_status rc = tic(c, c->ev[3]);
  if (rc) return rc;
  (void)hipGetLastError();
  const int P = lcv_helpers(c, 2 * n);
  int depth = 0;
  if (c->spec_on && n <= NBP_SPEC_MAXJOBS) depth = (c->spec_depth3 && coords * 7 <= NBP_SPEC_MAXBLOCKS) ? 3 : ((coords * 3 <= NBP_SPEC_MAXBLOCKS) ? 2 : 0);
  const bool spec = depth > 0;
  if (spec) blank_spec_areas(c, n);
  if (depth == 3)
    hipLaunchKernelGGL(nbp_bandwidth_kernel_spec<3>, dim3(n, 3, 7), dim3(P * c->Npad), nbp_bandwidth_lds_bytes(c->N, c->Npad, P), c->stream,
                       dev_slots, dev_manis, c->arena, c->N, c->Npad, c->S, c->counters, c->spec);
  else if (depth == 2)
    hipLaunchKernelGGL(nbp_bandwidth_kernel_spec<2>, dim3(n, 3, 3), dim3(P * c->Npad), nbp_bandwidth_lds_bytes(c->N, c->Npad, P), c->stream,
                       dev_slots, dev_manis, c->arena, c->N, c->Npad, c->S, c->counters, c->spec);
  else
  {
    auto *kern = rows_of_4k_plus_1_waves(c, P) ? nbp_bandwidth_kernel_w5 : nbp_bandwidth_kernel;
    if (getenv("NBP_DEBUG_OCCUPANCY")) {  // what the runtime says a CU can hold of this launch
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, P * c->Npad, nbp_bandwidth_lds_bytes(c->N, c->Npad, P));
      fprintf(stderr, "[nbp] nbp_bandwidth_kernel%s: %d lanes, %zu B of LDS per workgroup: %d workgroups = %d waves per CU\n",
              kern == nbp_bandwidth_kernel_w5 ? "_w5" : "", P * c->Npad, nbp_bandwidth_lds_bytes(c->N, c->Npad, P), nb, nb * (P * c->Npad / 64));
    }
    hipLaunchKernelGGL(kern, dim3(n, 3), dim3(P * c->Npad), nbp_bandwidth_lds_bytes(c->N, c->Npad, P), c->stream,
                       dev_slots, dev_manis, c->arena, c->N, c->Npad, c->S, c->counters);
  }
  HIPCHK(hipGetLastError());
  return toc(c, c->ev[3]);
}

// bandwidth jobs of a batch of proposals / products (host side)
static void jobs_of_proposals(const nbp_proposal_desc *d, int n, std::vector<int32_t> &slots, std::vector<int32_t> &manis) {
  for (int i = 0; i < n; i++)
    if (!d[i].skip_bandwidth && d[i].factor_kind != NBP_F_PASSTHROUGH) { slots.push_back(d[i].out_slot); manis.push_back(d[i].manifold); }
}
static void jobs_of_products(const nbp_product_desc *d, int n, std::vector<int32_t> &slots, std::vector<int32_t> &manis) {
  for (int i = 0; i < n; i++)
    if (d[i].nfactors > 1) { slots.push_back(d[i].out_slot); manis.push_back(d[i].manifold); }  // pass-through keeps its bw
}

// largest (F, D) of a product batch, encoded F*4 + D (sizes the LDS of the launch)
static int products_maxfd(const nbp_product_desc *d, int n) {
  int F = 2, D = 1;
  for (int i = 0; i < n; i++)
    if (d[i].nfactors > 1) {
      if (d[i].nfactors > F) F = d[i].nfactors;
      if (manifold_dim_h(d[i].manifold) > D) D = manifold_dim_h(d[i].manifold);
    }
  return F * 4 + D;
}
static nbp_status launch_copies(nbp_ctx *c, const nbp_copy_desc *dev, int n) {
  if (n <= 0) return NBP_OK;
  (void)hipGetLastError();
  hipLaunchKernelGGL(nbp_copy_kernel, dim3(n), dim3(256), 0, c->stream, dev, c->arena, c->S);
  HIPCHK(hipGetLastError());
  return NBP_OK;
}

static nbp_status launch_copy_points(nbp_ctx *c, const nbp_copy_desc *dev, int n) {
  if (n <= 0) return NBP_OK;
  (void)hipGetLastError();
  hipLaunchKernelGGL(nbp_copy_points_kernel, dim3(n), dim3(256), 0, c->stream, dev, c->arena, c->S, c->N);
  HIPCHK(hipGetLastError());
  return NBP_OK;
}

static nbp_status stage_upload(nbp_ctx *c, const void *src, size_t bytes) {
  if (bytes > c->stage_bytes) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->stage) HIPCHK(hipFree(c->stage));
    c->stage_bytes = bytes * 2;
    HIPCHK(hipMalloc(&c->stage, c->stage_bytes));
  }
  HIPCHK(hipStreamSynchronize(c->stream));  // the previous batch may still read the staging buffer
  HIPCHK(hipMemcpy(c->stage, src, bytes, hipMemcpyHostToDevice));
  return NBP_OK;
}

// upload [descriptors | job slots | job manifolds] in one staging copy and return the device views
static nbp_status stage_with_jobs(nbp_ctx *c, const void *descs, size_t desc_bytes, const std::vector<int32_t> &slots,
                                  const std::vector<int32_t> &manis, const int32_t **dslots, const int32_t **dmanis) {
  size_t off = (desc_bytes + 63) & ~(size_t)63;
  std::vector<char> buf(off + (slots.size() + manis.size()) * 4 + 8);
  memcpy(buf.data(), descs, desc_bytes);
  if (!slots.empty()) {
    memcpy(buf.data() + off, slots.data(), slots.size() * 4);
    memcpy(buf.data() + off + slots.size() * 4, manis.data(), manis.size() * 4);
  }
  nbp_status rc = stage_upload(c, buf.data(), buf.size());
  if (rc) return rc;
  *dslots = (const int32_t *)((char *)c->stage + off);
  *dmanis = *dslots + slots.size();
  return NBP_OK;
}

nbp_status nbp_run_proposals(nbp_ctx *c, const nbp_proposal_desc *descs, int32_t n) {
  if (!c || (!descs && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  nbp_status rc = check_proposals(c, descs, n);
  if (rc) return rc;
  std::vector<int32_t> js, jm;
  jobs_of_proposals(descs, n, js, jm);
  const int32_t *ds, *dm;
  rc = stage_with_jobs(c, descs, sizeof(nbp_proposal_desc) * (size_t)n, js, jm, &ds, &dm);
  if (rc) return rc;
  rc = launch_proposals(c, (const nbp_proposal_desc *)c->stage, n, proposals_uniform_class(descs, n));
  if (rc) return rc;
  rc = launch_bandwidth(c, ds, dm, (int)js.size(), coords_of(jm.data(), jm.size()));  // manikde!(M, pts), ApproxConv.jl:36-42
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBP_OK;
}

nbp_status nbp_run_products(nbp_ctx *c, const nbp_product_desc *descs, int32_t n) {
  if (!c || (!descs && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  nbp_status rc = check_products(c, descs, n);
  if (rc) return rc;
  std::vector<int32_t> js, jm;
  jobs_of_products(descs, n, js, jm);
  const int32_t *ds, *dm;
  rc = stage_with_jobs(c, descs, sizeof(nbp_product_desc) * (size_t)n, js, jm, &ds, &dm);
  if (rc) return rc;
  rc = launch_prep(c, nullptr, nullptr, 0, (const nbp_product_desc *)c->stage, n, products_maxfd(descs, n), -1, products_uniform_manifold(descs, n));  // KD trees
  if (rc) return rc;
  rc = launch_products(c, (const nbp_product_desc *)c->stage, n, products_maxfd(descs, n), products_uniform_manifold(descs, n));
  if (rc) return rc;
  rc = launch_bandwidth(c, ds, dm, (int)js.size(), coords_of(jm.data(), jm.size()));  // rebandwidth of the product
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBP_OK;
}

static nbp_status check_deconv(nbp_ctx *c, const nbp_proposal_desc *descs, int n) {
  nbp_status rc = check_proposals(c, descs, n);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    const nbp_proposal_desc &p = descs[i];
    if (p.factor_kind < NBP_F_LINREL) return fail(NBP_ERR_ARG, "deconv: relative factors only (a prior's predicted measurement is the point itself)");
    if (p.has_multihypo || p.nvars != 2) return fail(NBP_ERR_ARG, "deconv: multihypo is not supported (reference issue #467/#927)");
    if (p.partial_mask) return fail(NBP_ERR_ARG, "deconv: partial factors are not supported");
  }
  return NBP_OK;
}

static nbp_status launch_deconv(nbp_ctx *c, const nbp_proposal_desc *dev_descs, const int32_t *dev_meas, int n) {
  if (n <= 0) return NBP_OK;
  (void)hipGetLastError();
  hipLaunchKernelGGL(nbp_deconv_kernel, dim3(n), dim3(c->Npad), 0, c->stream, dev_descs, dev_meas, c->arena, c->N, c->S, c->counters);
  HIPCHK(hipGetLastError());
  return NBP_OK;
}

nbp_status nbp_run_deconv(nbp_ctx *c, const nbp_proposal_desc *descs, const int32_t *meas_slots, int32_t n) {
  if (!c || (!descs && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  nbp_status rc = check_deconv(c, descs, n);
  if (rc) return rc;
  std::vector<int32_t> ms((size_t)n, -1), none;
  for (int i = 0; i < n; i++) {
    if (meas_slots) {
      if (meas_slots[i] >= c->n_slots) return fail(NBP_ERR_RANGE, "deconv: meas_slot");
      ms[i] = meas_slots[i];
    }
  }
  const int32_t *ds, *dm;
  rc = stage_with_jobs(c, descs, sizeof(nbp_proposal_desc) * (size_t)n, ms, none, &ds, &dm);
  if (rc) return rc;
  rc = tic(c, c->ev[0]);
  if (rc) return rc;
  rc = launch_deconv(c, (const nbp_proposal_desc *)c->stage, ds, n);
  if (rc) return rc;
  rc = toc(c, c->ev[0]);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBP_OK;
}

// ---- host-buffer entry points: one call per reference function, for callers that keep beliefs on the
// host (the Julia shim's factor / variable seams).  They stage through the first slots of the context.
nbp_status nbp_kde_bandwidth(nbp_ctx *c, int32_t manifold, const double *pts, double *bw_out) {
  if (!c || !pts || !bw_out) return fail(NBP_ERR_ARG, "null argument");
  nbp_status rc = nbp_slot_write(c, 0, manifold, pts, nullptr);
  if (rc) return rc;
  const int32_t slot = 0;
  rc = nbp_run_bandwidth(c, &slot, &manifold, 1);
  if (rc) return rc;
  std::vector<double> tmp((size_t)c->N * 6);
  return nbp_slot_read(c, 0, manifold, tmp.data(), bw_out);
}

nbp_status nbp_conv(nbp_ctx *c, const nbp_proposal_desc *tmpl, const double *const *var_pts, const double *const *var_bw,
                    const int32_t *mhidx_in, double *out_pts, double *out_bw, int32_t *out_mhidx) {
  if (!c || !tmpl || !var_pts || !out_pts) return fail(NBP_ERR_ARG, "null argument");
  nbp_proposal_desc d = *tmpl;
  const int nin = (d.factor_kind == NBP_F_MSGPRIOR || d.factor_kind == NBP_F_PASSTHROUGH) ? 2 : d.nvars;
  if (nin < 1 || nin > NBP_MAXV) return fail(NBP_ERR_RANGE, "conv: nvars");
  if (c->n_slots < nin + 1) return fail(NBP_ERR_RANGE, "conv: the context needs nvars + 1 slots");
  if ((mhidx_in || out_mhidx) && c->side_ints < 2 * c->N) return fail(NBP_ERR_RANGE, "conv: the context needs 2N side ints");
  for (int i = 0; i < nin; i++) {
    if (!var_pts[i]) return fail(NBP_ERR_ARG, "conv: null variable points");
    nbp_status rc = nbp_slot_write(c, i, d.manifold, var_pts[i], var_bw ? var_bw[i] : nullptr);
    if (rc) return rc;
    d.var_slot[i] = i;
  }
  d.out_slot = nin;
  d.mhidx_in = -1;
  d.mhidx_out = -1;
  if (mhidx_in) {
    nbp_status rc = nbp_side_write(c, 0, mhidx_in, c->N);
    if (rc) return rc;
    d.mhidx_in = 0;
  }
  if (out_mhidx) d.mhidx_out = c->N;
  d.skip_bandwidth = out_bw ? 0 : 1;
  nbp_status rc = nbp_run_proposals(c, &d, 1);
  if (rc) return rc;
  rc = nbp_slot_read(c, nin, d.manifold, out_pts, out_bw);
  if (rc) return rc;
  if (out_mhidx) rc = nbp_side_read(c, c->N, out_mhidx, c->N);
  return rc;
}

nbp_status nbp_manifold_product(nbp_ctx *c, int32_t manifold, int32_t F, const double *const *dens_pts, const double *const *dens_bw,
                                const uint8_t *partial_masks, const double *old_pts, int32_t niter, uint64_t seed,
                                double *out_pts, double *out_bw, int32_t *out_labels) {
  if (!c || !dens_pts || !dens_bw || !out_pts) return fail(NBP_ERR_ARG, "null argument");
  if (F < 1 || F > NBP_MAXF) return fail(NBP_ERR_RANGE, "product: nfactors");
  if (c->n_slots < F + 2) return fail(NBP_ERR_RANGE, "product: the context needs F + 2 slots");
  if (out_labels && c->side_ints < c->N * F) return fail(NBP_ERR_RANGE, "product: the context needs N*F side ints");
  nbp_product_desc d;
  memset(&d, 0, sizeof(d));
  d.manifold = manifold;
  d.nfactors = F;
  d.niter = niter;
  d.out_slot = F + 1;
  d.labels_out = out_labels ? 0 : -1;
  d.old_slot = -1;
  d.seed = seed;
  for (int j = 0; j < F; j++) {
    if (!dens_pts[j] || !dens_bw[j]) return fail(NBP_ERR_ARG, "product: null density");
    nbp_status rc = nbp_slot_write(c, j, manifold, dens_pts[j], dens_bw[j]);
    if (rc) return rc;
    d.in_slot[j] = j;
    d.in_partial[j] = partial_masks ? partial_masks[j] : 0;
  }
  if (old_pts) {
    nbp_status rc = nbp_slot_write(c, F, manifold, old_pts, nullptr);
    if (rc) return rc;
    d.old_slot = F;
  }
  nbp_status rc = nbp_run_products(c, &d, 1);
  if (rc) return rc;
  rc = nbp_slot_read(c, F + 1, manifold, out_pts, out_bw);
  if (rc) return rc;
  if (out_labels) rc = nbp_side_read(c, 0, out_labels, c->N * F);
  return rc;
}

nbp_status nbp_run_copies(nbp_ctx *c, const nbp_copy_desc *descs, int32_t n) {
  if (!c || (!descs && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  nbp_status rc = check_copies(c, descs, n);
  if (rc) return rc;
  rc = stage_upload(c, descs, sizeof(nbp_copy_desc) * (size_t)n);
  if (rc) return rc;
  rc = launch_copies(c, (const nbp_copy_desc *)c->stage, n);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBP_OK;
}

// slot copies queued on the library stream, nothing waited for (resident beliefs of the asynchronous clique seam:
// nbp_resident_copy).  The descriptors are read by the kernel from a pinned buffer of the context's pool (device-visible
// host memory: a few bytes per workgroup), freed behind the launch.  points_only: NBP_STAGE_COPY_POINTS semantics.
nbp_status nbp_run_copies_async(nbp_ctx *c, const nbp_copy_desc *descs, int32_t n, int32_t points_only) {
  if (!c || (!descs && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  nbp_status rc = check_copies(c, descs, n);
  if (rc) return rc;
  nbp_ctx::pin_buf *b = pin_acquire(c, sizeof(nbp_copy_desc) * (size_t)n);
  if (!b) return fail(NBP_ERR_HIP, "pinned staging buffer");
  memcpy(b->p, descs, sizeof(nbp_copy_desc) * (size_t)n);
  rc = points_only ? launch_copy_points(c, (const nbp_copy_desc *)b->p, n) : launch_copies(c, (const nbp_copy_desc *)b->p, n);
  pin_release_behind_stream(c, b);
  return rc;
}

nbp_status nbp_run_bandwidth(nbp_ctx *c, const int32_t *slots, const int32_t *manifolds, int32_t n) {
  if (!c || ((!slots || !manifolds) && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n; i++) {
    if (slots[i] < 0 || slots[i] >= c->n_slots) return fail(NBP_ERR_RANGE, "bandwidth: slot out of range");
    if (!manifold_ok(manifolds[i])) return fail(NBP_ERR_ARG, "bandwidth: unknown manifold");
  }
  std::vector<int32_t> both(slots, slots + n);
  both.insert(both.end(), manifolds, manifolds + n);
  nbp_status rc = stage_upload(c, both.data(), both.size() * 4);
  if (rc) return rc;
  const int32_t *ds = (const int32_t *)c->stage;
  rc = launch_bandwidth(c, ds, ds + n, n, coords_of(manifolds, (size_t)n));
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBP_OK;
}

nbp_status nbp_run_resample(nbp_ctx *c, const int32_t *slots, const int32_t *manifolds, int32_t n, uint64_t seed) {
  if (!c || ((!slots || !manifolds) && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n; i++) {
    if (slots[i] < 0 || slots[i] >= c->n_slots) return fail(NBP_ERR_RANGE, "resample: slot out of range");
    if (!manifold_ok(manifolds[i])) return fail(NBP_ERR_ARG, "resample: unknown manifold");
  }
  std::vector<int32_t> both(slots, slots + n);
  both.insert(both.end(), manifolds, manifolds + n);
  nbp_status rc = stage_upload(c, both.data(), both.size() * 4);
  if (rc) return rc;
  const int32_t *ds = (const int32_t *)c->stage;
  (void)hipGetLastError();
  hipLaunchKernelGGL(nbp_resample_kernel, dim3(n), dim3(256), 0, c->stream, ds, ds + n, c->arena, c->N, c->S, seed);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBP_OK;
}

// ---- resident programs ---------------------------------------------------------------------------------
// Scheduling of the bandwidth fits (manikde!).  Fits are *deferred*: a proposal stage only queues
// its fits; the next product stage runs them inside its nbp_prep_kernel launch, next to the KD-tree
// builds (which need points, not bandwidths); the product's own rebandwidth is queued in turn and
// rides with the prep launch of the NEXT update.  A queued fit is flushed early (plain
// nbp_bandwidth_kernel launch) when a later stage reads the bandwidth of that slot: a MsgPrior
// proposal sampling from it, or a copy stage moving whole slots.  Per update the critical path is
// proposal -> prep (LCV || KD) -> product.
struct nbp_stage {
  int kind = 0, n = 0, maxfd = 0, mani = 0;  // mani: products_uniform_manifold / proposals_uniform_class
  size_t offset = 0;            // byte offset of the descriptors in the program blob
  std::vector<int32_t> ent_s, ent_m;  // fits pending at ENTRY of the stage
  size_t ent_off = 0;
  bool flush_before = false;    // run the pending fits in a plain bandwidth launch before the stage
  bool need_prep = true;        // products: some product multiplies > 1 densities (KD builds; the entry fits run beside them)
  // fused variable updates: this PROPOSALS stage and the PRODUCTS stage behind it run as ONE nbp_update_kernel launch
  // (fused_second marks that PRODUCTS stage); upd = one nbp_update_desc per product
  bool fused = false, fused_second = false;
  int upd_F = 0, upd_cls = 0;
  std::vector<nbp_update_desc> upd;
  size_t upd_off = 0;
  // a range of nbp_program_run that ends or starts BETWEEN the two stages of a fused pair runs them in the three-launch
  // form (run_range): the fused_second stage keeps, as its entry fits, the proposals of the pair that carry a bandwidth
  // (ent_s / ent_m) and, here, the outputs the fused launch fits itself
  std::vector<int32_t> split_in_s, split_in_m, split_out_s, split_out_m;
  size_t split_out_off = 0;
  // two-stream round (plan_pipeline): descriptors [0, pipe_split) are the first half; on the PRODUCTS stage pipe_ent splits
  // the entry fits the same way.  `pipe` on the PROPOSALS stage = the pair runs that way.
  bool pipe = false;
  int pipe_split = -1, pipe_ent = 0;
};
struct nbp_program {
  nbp_ctx *ctx = nullptr;
  std::vector<nbp_stage> stages;  // + one trailing pseudo stage holding the fits pending at exit
  std::vector<char> blob;
  char *dev = nullptr;
  size_t dev_bytes = 0;
  bool finalized = false;
  bool lazy_bw = false;  // NBP_OPT_LAZY_BANDWIDTH
  bool use_graph = true; // NBP_OPT_GRAPH_REPLAY
  bool use_fused = true; // NBP_OPT_FUSED_UPDATES
  bool async_upload = false;  // NBP_OPT_ASYNC_UPLOAD: the descriptor blob travels stream-ordered from a pinned buffer, nothing waits
  int n_user_stages = 0;
  // captured launch sequences of nbp_program_run(first, last): key = first * 2^32 + last
  struct captured { hipGraphExec_t exec; uint64_t ws_gen; };  // ws_gen: the context's workspace generation at capture time
  std::unordered_map<uint64_t, captured> graphs;
  std::unordered_map<uint64_t, int> runs;
  size_t seed_off = 0;  // table of the blob offsets of every descriptor's seed field (one reseed launch)
  size_t seed_val_off = 0;  // n_seeds values behind the table: where nbp_program_set_seeds puts the new seeds before it scatters them
  int n_seeds = 0;
};

nbp_status nbp_program_create(nbp_ctx *c, nbp_program **out) {
  if (!c || !out) return fail(NBP_ERR_ARG, "null argument");
  nbp_program *p = new nbp_program();
  p->ctx = c;
  c->programs.push_back(p);
  *out = p;
  return NBP_OK;
}
// the context is going away: free the device blob while the context's device is still current
static void program_detach(nbp_program *p) {
  for (auto &kv : p->graphs) hipGraphExecDestroy(kv.second.exec);
  p->graphs.clear();
  if (p->dev) hipFree(p->dev);
  p->dev = nullptr;
  p->ctx = nullptr;
}
static void program_delete(nbp_program *p) {
  program_detach(p);
  delete p;
}
#define PROG_ALIVE(p) do { if (!(p)->ctx) return fail(NBP_ERR_ARG, "the program's context was destroyed"); } while (0)

nbp_status nbp_program_add_stage(nbp_program *p, int32_t kind, const void *descs, int32_t n) {
  if (!p || (!descs && n > 0)) return fail(NBP_ERR_ARG, "null argument");
  PROG_ALIVE(p);
  if (p->finalized) return fail(NBP_ERR_ARG, "program already finalized");
  if (n < 0) return fail(NBP_ERR_ARG, "n < 0");
  size_t esz;
  nbp_status rc = NBP_OK;
  nbp_stage st;
  switch (kind) {
  case NBP_STAGE_PROPOSALS:
    esz = sizeof(nbp_proposal_desc);
    rc = check_proposals(p->ctx, (const nbp_proposal_desc *)descs, n);
    st.mani = proposals_uniform_class((const nbp_proposal_desc *)descs, n);
    break;
  case NBP_STAGE_PRODUCTS:
    esz = sizeof(nbp_product_desc);
    rc = check_products(p->ctx, (const nbp_product_desc *)descs, n);
    st.maxfd = products_maxfd((const nbp_product_desc *)descs, n);
    st.mani = products_uniform_manifold((const nbp_product_desc *)descs, n);
    break;
  case NBP_STAGE_COPIES:
  case NBP_STAGE_COPY_POINTS: esz = sizeof(nbp_copy_desc); rc = check_copies(p->ctx, (const nbp_copy_desc *)descs, n); break;
  case NBP_STAGE_DECONV: esz = sizeof(nbp_proposal_desc); rc = check_deconv(p->ctx, (const nbp_proposal_desc *)descs, n); break;
  default: return fail(NBP_ERR_ARG, "unknown stage kind");
  }
  if (rc) return rc;
  size_t off = (p->blob.size() + 63) & ~(size_t)63;
  p->blob.resize(off + esz * (size_t)n);
  if (n) memcpy(p->blob.data() + off, descs, esz * (size_t)n);
  st.kind = kind; st.n = n; st.offset = off;
  p->stages.push_back(st);
  return NBP_OK;
}

nbp_status nbp_program_set_option(nbp_program *p, int32_t option, int32_t value) {
  if (!p) return fail(NBP_ERR_ARG, "null argument");
  // (graph replay is a property of the runs, not of the compiled stages: it may be switched after nbp_program_finalize --
  //  graphs already captured stay with the program and are used again when it is switched back on)
  if (option == NBP_OPT_GRAPH_REPLAY && p->finalized) { p->use_graph = value != 0; return NBP_OK; }
  if (p->finalized) return fail(NBP_ERR_ARG, "program already finalized");
  // NBP_NO_LAZY_BANDWIDTH=1 (environment): every fit the reference makes is made, whatever the program asks for -- the
  // measurement of what the option saves (bench.py "ms_per_step_every_fit")
  if (option == NBP_OPT_LAZY_BANDWIDTH) { p->lazy_bw = value != 0 && getenv("NBP_NO_LAZY_BANDWIDTH") == nullptr; return NBP_OK; }
  if (option == NBP_OPT_GRAPH_REPLAY) { p->use_graph = value != 0; return NBP_OK; }
  if (option == NBP_OPT_FUSED_UPDATES) { p->use_fused = value != 0; return NBP_OK; }
  if (option == NBP_OPT_ASYNC_UPLOAD) { p->async_upload = value != 0; return NBP_OK; }
  return fail(NBP_ERR_ARG, "unknown program option");
}

// Liveness of product rebandwidths (NBP_OPT_LAZY_BANDWIDTH): the bandwidth of a product's output is read
// only by a MsgPrior proposal sampling that slot, by a product taking it as input, and by slot copies
// (which carry it along).  dead[s][i] = the output of product i of stage s is overwritten by a later
// product / copy / proposal before any of those happens, so its fit can be skipped without changing any
// result.  An EMPTY copy stage is a barrier ("everything may be read now": slots leave the device).
struct nbp_liveness {
  std::vector<std::vector<char>> dead_product, dead_proposal;  // [stage][descriptor]
};
static nbp_liveness product_liveness(const nbp_program *p) {
  nbp_liveness L;
  L.dead_product.resize(p->n_user_stages);
  L.dead_proposal.resize(p->n_user_stages);
  struct Open { int stage, idx, pstage, pidx; };  // pstage >= 0: pass-through of that proposal's KDE
  std::unordered_map<int32_t, Open> open;                        // slot -> unresolved product output
  std::unordered_map<int32_t, std::pair<int, int>> last_prop;    // scratch slot -> proposal that wrote it
  // proposals whose own KDE (points + bandwidth) is read directly: an input of a real product, the message of a
  // MsgPrior, a measurement KDE or the source of a slot copy.  Such a fit is live whatever happens to the output of
  // a pass-through product that also carries it.
  std::vector<std::pair<int, int>> needed;
  auto read_kde = [&](int32_t slot) {
    open.erase(slot);
    auto lp = last_prop.find(slot);
    if (lp != last_prop.end()) needed.push_back(lp->second);
  };
  auto kill = [&](int32_t slot) {
    auto it = open.find(slot);
    if (it == open.end()) return;
    const Open &o = it->second;
    if (o.pstage >= 0) L.dead_proposal[o.pstage][o.pidx] = 1;  // a pass-through hands on the proposal's own fit
    else L.dead_product[o.stage][o.idx] = 1;
    open.erase(it);
  };
  for (int s = 0; s < p->n_user_stages; s++) {
    const nbp_stage &st = p->stages[s];
    const char *d = p->blob.data() + st.offset;
    if (st.kind == NBP_STAGE_PROPOSALS) {
      const nbp_proposal_desc *pd = (const nbp_proposal_desc *)d;
      L.dead_proposal[s].assign(st.n, 0);
      for (int i = 0; i < st.n; i++) {
        if (pd[i].factor_kind == NBP_F_MSGPRIOR || pd[i].factor_kind == NBP_F_PASSTHROUGH) read_kde(pd[i].var_slot[1]);  // read: live
        if (pd[i].meas_kde > 0) read_kde(pd[i].meas_kde - 1);
      }
      for (int i = 0; i < st.n; i++) {
        kill(pd[i].out_slot);  // overwritten
        last_prop[pd[i].out_slot] = {s, i};
      }
    } else if (st.kind == NBP_STAGE_DECONV) {  // reads points only; its outputs are always fitted
      const nbp_proposal_desc *pd = (const nbp_proposal_desc *)d;
      for (int i = 0; i < st.n; i++) { kill(pd[i].out_slot); last_prop.erase(pd[i].out_slot); }
    } else if (st.kind == NBP_STAGE_COPY_POINTS) {  // not a reader of the source's bandwidth; the destination is overwritten
      const nbp_copy_desc *cd = (const nbp_copy_desc *)d;
      for (int i = 0; i < st.n; i++) { kill(cd[i].dst_slot); last_prop.erase(cd[i].dst_slot); }
    } else if (st.kind == NBP_STAGE_COPIES) {
      const nbp_copy_desc *cd = (const nbp_copy_desc *)d;
      if (st.n == 0) {  // barrier: all live
        open.clear();
        for (auto &lp : last_prop) needed.push_back(lp.second);
      }
      for (int i = 0; i < st.n; i++) read_kde(cd[i].src_slot);
      for (int i = 0; i < st.n; i++) { kill(cd[i].dst_slot); last_prop.erase(cd[i].dst_slot); }
    } else if (st.kind == NBP_STAGE_PRODUCTS) {
      const nbp_product_desc *qd = (const nbp_product_desc *)d;
      L.dead_product[s].assign(st.n, 0);
      for (int i = 0; i < st.n; i++)
        for (int j = 0; j < qd[i].nfactors; j++) {  // input KDE: bandwidth read
          if (qd[i].nfactors > 1) read_kde(qd[i].in_slot[j]);
          else open.erase(qd[i].in_slot[j]);  // pass-through: the proposal's fit travels with the output (below)
        }
      for (int i = 0; i < st.n; i++) {
        kill(qd[i].out_slot);
        if (qd[i].nfactors > 1) {
          open[qd[i].out_slot] = {s, i, -1, -1};
        } else {  // pass-through: the output carries the bandwidth fitted for the proposal
          auto lp = last_prop.find(qd[i].in_slot[0]);
          if (lp != last_prop.end() && !((const nbp_proposal_desc *)(p->blob.data() + p->stages[lp->second.first].offset))[lp->second.second].skip_bandwidth)
            open[qd[i].out_slot] = {s, i, lp->second.first, lp->second.second};
        }
        last_prop.erase(qd[i].out_slot);
      }
    }
  }
  for (auto &nd : needed) L.dead_proposal[nd.first][nd.second] = 0;
  return L;  // whatever is still open is flushed at the end of the program: live
}

// ---- fused variable updates ---------------------------------------------------------------------------------------------
// A PROPOSALS stage followed by the PRODUCTS stage that multiplies exactly its proposals -- one round of Gibbs steps, as
// both hosts emit it -- runs as one launch of the fused update kernel when the round fills the chip and is of a class the
// kernel is built for.  What must hold (checked here, so that any program a caller assembles stays correct):
//   * every input of every product is the output of exactly one proposal of the stage in front, and every proposal feeds
//     exactly one product (the proposals never reach HBM; one that a later stage reads from its slot is written there too);
//   * no proposal reads a slot another update of the round writes (the three-launch form runs all proposals before any
//     product; fused, the updates of a round run in any order) -- rounds of commuting Gibbs steps satisfy this by construction;
//   * class: LinearRelative / priors / message priors on Euclid(2), full (non-partial) densities, no label output.
static bool fused_plan(const nbp_program *p, int s, std::vector<nbp_update_desc> &upd, int &Fmax, int &cls) {
  const nbp_ctx *c = p->ctx;
  if (!c->fused_on || !p->use_fused || s + 1 >= p->n_user_stages || c->Npad > 256) return false;
  const nbp_stage &A = p->stages[s], &B = p->stages[s + 1];
  if (A.kind != NBP_STAGE_PROPOSALS || B.kind != NBP_STAGE_PRODUCTS || B.n < c->fused_min || A.n < B.n) return false;
  if ((A.mani & (NBP_CLS_SIMPLE - 1)) != 1) return false;  // proposals_uniform_class: 1 = LinearRelative on Euclid(2)
  const int M = NBP_EUCLID2;
  const nbp_proposal_desc *pd = (const nbp_proposal_desc *)(p->blob.data() + A.offset);
  const nbp_product_desc *qd = (const nbp_product_desc *)(p->blob.data() + B.offset);
  std::unordered_map<int32_t, int> prop_of, prod_of;  // slot -> proposal / product writing it
  for (int i = 0; i < A.n; i++) {
    if (pd[i].manifold != M) return false;
    if (!prop_of.emplace(pd[i].out_slot, i).second) return false;
  }
  for (int i = 0; i < B.n; i++) {
    if (qd[i].manifold != M || qd[i].nfactors > NBP_FUSED_MAXF || qd[i].labels_out >= 0 || qd[i].old_slot >= 0) return false;
    if (!prod_of.emplace(qd[i].out_slot, i).second) return false;
  }
  upd.assign(B.n, nbp_update_desc{});
  std::vector<char> used(A.n, 0);
  int nused = 0;
  Fmax = 2;
  for (int i = 0; i < B.n; i++) {
    nbp_update_desc &u = upd[i];
    u.prod = i;
    if (qd[i].nfactors > Fmax) Fmax = qd[i].nfactors;
    for (int j = 0; j < qd[i].nfactors; j++) {
      if (qd[i].in_partial[j]) return false;
      auto it = prop_of.find(qd[i].in_slot[j]);
      if (it == prop_of.end() || used[it->second]) return false;
      const nbp_proposal_desc &q = pd[it->second];
      if (qd[i].nfactors > 1 && q.skip_bandwidth && q.factor_kind != NBP_F_PASSTHROUGH) return false;  // the product would read a stale bandwidth
      if (q.factor_kind == NBP_F_PASSTHROUGH && qd[i].nfactors == 1 && q.keep_count) return false;
      used[it->second] = 1;
      nused++;
      u.prop[j] = it->second;
      // what the proposal reads: never a slot that another update of this launch writes, nor another proposal's slot
      const int nv = (q.factor_kind == NBP_F_MSGPRIOR || q.factor_kind == NBP_F_PASSTHROUGH) ? 2 : q.nvars;
      for (int k = 0; k <= nv; k++) {
        const int32_t r = k < nv ? q.var_slot[k] : (q.meas_kde > 0 ? q.meas_kde - 1 : -1);
        if (r < 0) continue;
        auto w = prod_of.find(r);
        if (w != prod_of.end() && w->second != i) return false;
        if (prop_of.count(r)) return false;
      }
    }
  }
  if (nused != A.n) return false;
  // proposals a later stage reads from their arena slots before anything overwrites them (or that are still there when the
  // program ends): those are written as well
  std::unordered_map<int32_t, std::pair<int, int>> watch;  // slot -> (update, input)
  for (int i = 0; i < B.n; i++)
    for (int j = 0; j < qd[i].nfactors; j++) watch[qd[i].in_slot[j]] = {i, j};
  auto rd = [&](int32_t slot) {
    auto it = watch.find(slot);
    if (it == watch.end()) return;
    upd[it->second.first].flags |= 2 << it->second.second;
    watch.erase(it);
  };
  for (int i = 0; i < B.n; i++) watch.erase(qd[i].out_slot);  // (a product writing a proposal slot: overwritten at once)
  for (int t = s + 2; t < p->n_user_stages && !watch.empty(); t++) {
    const nbp_stage &st = p->stages[t];
    const char *d = p->blob.data() + st.offset;
    if (st.kind == NBP_STAGE_PROPOSALS || st.kind == NBP_STAGE_DECONV) {
      const nbp_proposal_desc *x = (const nbp_proposal_desc *)d;
      for (int i = 0; i < st.n; i++) {
        const int nv = (x[i].factor_kind == NBP_F_MSGPRIOR || x[i].factor_kind == NBP_F_PASSTHROUGH) ? 2 : x[i].nvars;
        for (int k = 0; k < nv; k++) rd(x[i].var_slot[k]);
        if (x[i].meas_kde > 0) rd(x[i].meas_kde - 1);
      }
      for (int i = 0; i < st.n; i++) watch.erase(x[i].out_slot);
    } else if (st.kind == NBP_STAGE_PRODUCTS) {
      const nbp_product_desc *x = (const nbp_product_desc *)d;
      for (int i = 0; i < st.n; i++) {
        for (int j = 0; j < x[i].nfactors; j++) rd(x[i].in_slot[j]);
        if (x[i].old_slot >= 0) rd(x[i].old_slot);
      }
      for (int i = 0; i < st.n; i++) watch.erase(x[i].out_slot);
    } else {
      const nbp_copy_desc *x = (const nbp_copy_desc *)d;
      if (st.n == 0) break;  // barrier: slots leave the device -- whatever is still watched is written (below)
      for (int i = 0; i < st.n; i++) rd(x[i].src_slot);
      for (int i = 0; i < st.n; i++) watch.erase(x[i].dst_slot);
    }
  }
  for (auto &w : watch) upd[w.second.first].flags |= 2 << w.second.second;
  cls = A.mani & (NBP_CLS_SIMPLE - 1);
  return true;
}

static nbp_status launch_update(nbp_ctx *c, const nbp_stage &st, const nbp_update_desc *uds, const nbp_proposal_desc *props,
                                const nbp_product_desc *prods, int n) {
  nbp_status rc = tic(c, c->ev[4]);
  if (rc) return rc;
  (void)hipGetLastError();
  const int P = n >= c->fused_p1_min ? 1 : 2;
  const size_t lds = nbp_update_lds_bytes(st.upd_F, 2, c->N, c->Npad, P, false);
  hipLaunchKernelGGL(P == 1 ? nbp_update_kernel_lin2_p1 : nbp_update_kernel_lin2_p2, dim3(n), dim3(P * c->Npad), lds, c->stream, uds, props, prods,
                     c->arena, c->N, c->Npad, c->S, c->side, c->T, c->counters, st.upd_F);
  HIPCHK(hipGetLastError());
  return toc(c, c->ev[4]);
}


// ---- two-stream rounds --------------------------------------------------------------------------------------------------
// A round of many variable updates is three chip-filling launches with different bottlenecks: the proposal launch waits
// (barriers of the inflation cycles, divergent per-particle searches), the fit launch issues FP64 back to back, and every
// launch ends in a tail of partly filled CUs.  The updates of a round are independent of each other, so the round is cut
// in two halves that run the same three launches on two streams, the second half one launch behind the first:
//     stream 1:  proposals(a)  fits + KD(a)   products(a)                 join
//     stream 2:                proposals(b)   fits + KD(b)   products(b)
// What must hold for that to compute what the single-stream order computes: a product of half a must not write a slot a
// proposal of half b still has to read.  plan_pipeline() puts such updates into the same half (union-find over the
// products of the stage), balances the halves, and reorders the descriptors of both stages so that each half is a
// contiguous range.  Every launch of a half uses the geometry of the whole batch (helper rows per fit, lanes per sample),
// so no particle depends on the split: tests/test_gpu_pipelined_rounds.py compares every slot, bit for bit.
static void plan_pipeline(nbp_program *p, int s) {
  nbp_stage &ps = p->stages[s], &qs = p->stages[s + 1];
  if (ps.kind != NBP_STAGE_PROPOSALS || qs.kind != NBP_STAGE_PRODUCTS || qs.n < p->ctx->pipe_min || qs.n < 128) return;
  nbp_proposal_desc *pd = (nbp_proposal_desc *)(p->blob.data() + ps.offset);
  nbp_product_desc *qd = (nbp_product_desc *)(p->blob.data() + qs.offset);
  std::unordered_map<int32_t, int> prop_of, prod_of, reader_of;
  for (int i = 0; i < ps.n; i++) prop_of[pd[i].out_slot] = i;
  for (int i = 0; i < qs.n; i++) prod_of[qd[i].out_slot] = i;
  std::vector<int> uf(qs.n), owner(ps.n, -1);
  for (int i = 0; i < qs.n; i++) uf[i] = i;
  auto find = [&](int a) { while (uf[a] != a) a = uf[a] = uf[uf[a]]; return a; };
  auto unite = [&](int a, int b) { a = find(a); b = find(b); if (a != b) uf[a] = b; };
  for (int i = 0; i < qs.n; i++) {
    if (qd[i].old_slot >= 0) return;  // partial products top their old points up inside the fit launch: left alone
    for (int j = 0; j < qd[i].nfactors; j++) {
      // two products reading one density (a proposal, or a message whose fit is still pending) stay together: the fit of
      // that density runs in one half only
      auto rd = reader_of.find(qd[i].in_slot[j]);
      if (rd == reader_of.end()) reader_of[qd[i].in_slot[j]] = i; else unite(rd->second, i);
      auto it = prop_of.find(qd[i].in_slot[j]);
      if (it != prop_of.end() && owner[it->second] < 0) owner[it->second] = i;
    }
  }
  for (int k = 0; k < ps.n; k++) {
    if (owner[k] < 0) continue;  // feeds no product of this round: first half, done before any product starts
    auto reads = [&](int32_t slot) { auto it = prod_of.find(slot); if (it != prod_of.end()) unite(owner[k], it->second); };
    for (int v = 0; v < pd[k].nvars && v < NBP_MAXV; v++) reads(pd[k].var_slot[v]);
    if (pd[k].factor_kind == NBP_F_MSGPRIOR || pd[k].factor_kind == NBP_F_PASSTHROUGH) reads(pd[k].var_slot[1]);
    if (pd[k].meas_kde > 0) reads(pd[k].meas_kde - 1);
  }
  // components, largest first, each to the lighter half (weight: densities, the unit of the fit and KD work)
  std::unordered_map<int, std::pair<int, std::vector<int>>> comp;
  for (int i = 0; i < qs.n; i++) { auto &c = comp[find(i)]; c.first += qd[i].nfactors; c.second.push_back(i); }
  std::vector<std::pair<int, std::vector<int>>> cs;
  for (auto &kv : comp) cs.push_back(std::move(kv.second));
  std::sort(cs.begin(), cs.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second[0] < b.second[0]; });
  std::vector<char> half(qs.n, 0);
  int w[2] = {0, 0}, cnt[2] = {0, 0};
  for (auto &c : cs) {
    const int h = w[1] < w[0] ? 1 : 0;
    w[h] += c.first;
    for (int i : c.second) { half[i] = (char)h; cnt[h]++; }
  }
  if (cnt[0] < qs.n / 3 || cnt[1] < qs.n / 3) return;  // one big component: nothing to run side by side
  std::vector<nbp_product_desc> q2;
  for (int h = 0; h < 2; h++)
    for (int i = 0; i < qs.n; i++) if (half[i] == h) q2.push_back(qd[i]);
  std::vector<nbp_proposal_desc> p2;
  int np0 = 0;
  for (int h = 0; h < 2; h++)
    for (int k = 0; k < ps.n; k++) {
      const int hk = owner[k] < 0 ? 0 : half[owner[k]];
      if (hk == h) { p2.push_back(pd[k]); np0 += h == 0; }
    }
  memcpy(qd, q2.data(), q2.size() * sizeof(nbp_product_desc));
  memcpy(pd, p2.data(), p2.size() * sizeof(nbp_proposal_desc));
  ps.pipe_split = np0;
  qs.pipe_split = cnt[0];
}

nbp_status nbp_program_finalize(nbp_program *p) {
  if (!p) return fail(NBP_ERR_ARG, "null argument");
  PROG_ALIVE(p);
  if (p->finalized) return NBP_OK;
  HIPCHK(hipSetDevice(p->ctx->device));
  p->n_user_stages = (int)p->stages.size();
  p->stages.emplace_back();  // trailing pseudo stage: fits pending at exit
  p->stages.back().kind = 0;
  std::vector<int32_t> pend_s, pend_m;
  int maxprod = 0;
  for (int s0 = 0; s0 + 1 < p->n_user_stages; s0++) plan_pipeline(p, s0);  // reorders descriptors: before anything indexes them
  nbp_liveness live;
  if (p->lazy_bw) live = product_liveness(p);
  int sidx = -1;
  for (nbp_stage &st : p->stages) {
    sidx++;
    const char *d = p->blob.data() + st.offset;
    st.ent_s = pend_s;
    st.ent_m = pend_m;
    if (st.kind == NBP_STAGE_PROPOSALS && sidx + 1 < p->n_user_stages && !st.fused) {
      std::vector<nbp_update_desc> upd;
      int Fm = 0, cls = 0;
      if (fused_plan(p, sidx, upd, Fm, cls) && nbp_update_lds_bytes(Fm, 2, p->ctx->N, p->ctx->Npad, 2, false) <= 158 * 1024) {
        // every fit of the round happens inside the launch: what is pending runs first, nothing is queued behind it
        nbp_stage &nx = p->stages[sidx + 1];
        const nbp_proposal_desc *pd = (const nbp_proposal_desc *)d;
        const nbp_product_desc *qd = (const nbp_product_desc *)(p->blob.data() + nx.offset);
        for (int i = 0; i < nx.n; i++) {
          bool fit;
          if (qd[i].nfactors > 1) fit = !p->lazy_bw || !live.dead_product[sidx + 1][i];
          else {
            const nbp_proposal_desc &q = pd[upd[i].prop[0]];
            fit = !q.skip_bandwidth && q.factor_kind != NBP_F_PASSTHROUGH && (!p->lazy_bw || !live.dead_proposal[sidx][upd[i].prop[0]]);
          }
          if (fit) upd[i].flags |= NBP_UPD_FIT_OUT;
        }
        nx.split_out_s.clear(); nx.split_out_m.clear();
        std::vector<int32_t> in_s, in_m;
        for (int i = 0; i < st.n; i++)
          if (!pd[i].skip_bandwidth && pd[i].factor_kind != NBP_F_PASSTHROUGH) { in_s.push_back(pd[i].out_slot); in_m.push_back(pd[i].manifold); }
        for (int i = 0; i < nx.n; i++)
          if (upd[i].flags & NBP_UPD_FIT_OUT) { nx.split_out_s.push_back(qd[i].out_slot); nx.split_out_m.push_back(qd[i].manifold); }
        nx.split_in_s = in_s;
        nx.split_in_m = in_m;
        st.fused = true;
        st.upd = std::move(upd);
        st.upd_F = Fm;
        st.upd_cls = cls;
        nx.fused_second = true;
        st.flush_before = !pend_s.empty();
        pend_s.clear(); pend_m.clear();
        continue;
      }
    }
    if (st.kind == NBP_STAGE_PRODUCTS && st.fused_second) {  // ran inside the launch of the stage in front
      // (its entry list: the fits of a range that splits the pair -- never read when the pair runs as one launch)
      st.ent_s = st.split_in_s;
      st.ent_m = st.split_in_m;
      st.need_prep = false;
      // an output slot whose old points still had a fit queued cannot be: the launch in front flushed everything
      continue;
    }
    if (st.kind == NBP_STAGE_PROPOSALS) {
      // a MsgPrior samples from the KDE in var_slot[1] (points AND bandwidth)
      for (int i = 0; i < st.n && !st.flush_before; i++) {
        const nbp_proposal_desc &pd = ((const nbp_proposal_desc *)d)[i];
        if (pd.factor_kind == NBP_F_MSGPRIOR || pd.factor_kind == NBP_F_PASSTHROUGH)
          for (int32_t ps : pend_s) st.flush_before |= (ps == pd.var_slot[1]);
        if (pd.meas_kde > 0)  // a measurement KDE: points and bandwidth
          for (int32_t ps : pend_s) st.flush_before |= (ps == pd.meas_kde - 1);
      }
      if (st.flush_before) { pend_s.clear(); pend_m.clear(); }
      if (p->lazy_bw) {
        const nbp_proposal_desc *pd = (const nbp_proposal_desc *)d;
        for (int i = 0; i < st.n; i++)
          if (!pd[i].skip_bandwidth && pd[i].factor_kind != NBP_F_PASSTHROUGH && !live.dead_proposal[sidx][i]) { pend_s.push_back(pd[i].out_slot); pend_m.push_back(pd[i].manifold); }
      } else
        jobs_of_proposals((const nbp_proposal_desc *)d, st.n, pend_s, pend_m);
    } else if (st.kind == NBP_STAGE_PRODUCTS) {
      const nbp_product_desc *qd0 = (const nbp_product_desc *)d;
      st.need_prep = false;
      for (int i = 0; i < st.n; i++) st.need_prep |= qd0[i].nfactors > 1;
      // the prep launch tops up the oldPoints of a partial product in place (topup_slot: reads the slot's bandwidth and
      // count, writes points and count) beside the fits of the same launch: a fit still pending for that very slot would
      // race with it, so such fits run first, in a launch of their own
      for (int i = 0; i < st.n && !st.flush_before; i++)
        if (qd0[i].nfactors > 1 && qd0[i].old_slot >= 0)
          for (int32_t ps : pend_s) st.flush_before |= (ps == qd0[i].old_slot);
      if (st.flush_before) { pend_s.clear(); pend_m.clear(); }
      if (st.need_prep) {
        pend_s.clear(); pend_m.clear();  // the entry fits run inside this stage's prep launch
      } else {
        // Only pass-through products (AMP returns the single density): nothing here reads a bandwidth, so nothing is
        // launched for the fits.  A pending fit of a density that is handed on moves with it -- same points, same
        // bandwidth, fitted in the output slot when the next launch with fits comes along (graph initialisation of a
        // chain: one proposal + one copy per variable on the critical path, all the fits in one launch at the end).
        for (int i = 0; i < st.n; i++)
          for (size_t q = 0; q < pend_s.size(); q++)
            if (pend_s[q] == qd0[i].in_slot[0]) pend_s[q] = qd0[i].out_slot;
        // (an output slot whose OLD points still had a fit queued: that fit would now see the new points -- which is what
        //  the reference's setBelief! does anyway: manikde! of the points it stores)
        for (size_t q = 0; q < pend_s.size(); q++)  // one fit per slot
          for (size_t r = q + 1; r < pend_s.size(); r++)
            if (pend_s[r] == pend_s[q]) { pend_s.erase(pend_s.begin() + r); pend_m.erase(pend_m.begin() + r); r--; }
      }
      if (p->lazy_bw) {
        const nbp_product_desc *qd = (const nbp_product_desc *)d;
        for (int i = 0; i < st.n; i++)
          if (qd[i].nfactors > 1 && !live.dead_product[sidx][i]) { pend_s.push_back(qd[i].out_slot); pend_m.push_back(qd[i].manifold); }
      } else
        jobs_of_products((const nbp_product_desc *)d, st.n, pend_s, pend_m);
      if (st.n > maxprod) maxprod = st.n;
    } else if (st.kind == NBP_STAGE_DECONV) {  // reads points only; queues the fits of its outputs
      const nbp_proposal_desc *pd = (const nbp_proposal_desc *)d;
      for (int i = 0; i < st.n; i++) {
        // an output slot that still has an older fit pending: that fit would see the new points, drop it
        for (size_t q = 0; q < pend_s.size(); q++)
          if (pend_s[q] == pd[i].out_slot) { pend_s.erase(pend_s.begin() + q); pend_m.erase(pend_m.begin() + q); q--; }
      }
      st.ent_s = pend_s;
      st.ent_m = pend_m;
      for (int i = 0; i < st.n; i++) { pend_s.push_back(pd[i].out_slot); pend_m.push_back(pd[i].manifold); }
    } else if (st.kind == NBP_STAGE_COPY_POINTS) {  // nothing is flushed; a fit still queued for a destination is void
      const nbp_copy_desc *cd = (const nbp_copy_desc *)d;
      for (int i = 0; i < st.n; i++)
        for (size_t q = 0; q < pend_s.size(); q++)
          if (pend_s[q] == cd[i].dst_slot) { pend_s.erase(pend_s.begin() + q); pend_m.erase(pend_m.begin() + q); q--; }
      st.ent_s = pend_s;
      st.ent_m = pend_m;
    } else {  // copies (move bandwidths too) and the trailing pseudo stage
      st.flush_before = true;
      pend_s.clear(); pend_m.clear();
    }
  }
  for (int s0 = 0; s0 + 1 < p->n_user_stages; s0++) {
    nbp_stage &ps = p->stages[s0], &qs = p->stages[s0 + 1];
    if (ps.kind != NBP_STAGE_PROPOSALS || qs.kind != NBP_STAGE_PRODUCTS || ps.pipe_split < 0 || qs.pipe_split < 0) continue;
    if (ps.fused || !qs.need_prep || qs.flush_before) continue;
    // products too large for the LDS share one node-statistics workspace: single stream (the launch's own decision)
    if (product_is_big(p->ctx, qs.n, qs.maxfd, qs.mani)) continue;
    const nbp_product_desc *qd = (const nbp_product_desc *)(p->blob.data() + qs.offset);
    std::unordered_map<int32_t, int> second;  // slots the second half's products read
    for (int i = qs.pipe_split; i < qs.n; i++)
      for (int j = 0; j < qd[i].nfactors; j++) second[qd[i].in_slot[j]] = 1;
    std::vector<int32_t> es, em;
    for (int h = 0; h < 2; h++)
      for (size_t q = 0; q < qs.ent_s.size(); q++)
        if ((int)second.count(qs.ent_s[q]) == h) { es.push_back(qs.ent_s[q]); em.push_back(qs.ent_m[q]); if (!h) qs.pipe_ent++; }
    qs.ent_s = es;
    qs.ent_m = em;
    ps.pipe = true;
  }
  for (nbp_stage &st : p->stages) {
    size_t off = (p->blob.size() + 63) & ~(size_t)63;
    p->blob.resize(off + (st.ent_s.size() + st.ent_m.size()) * 4);
    if (!st.ent_s.empty()) {
      memcpy(p->blob.data() + off, st.ent_s.data(), st.ent_s.size() * 4);
      memcpy(p->blob.data() + off + st.ent_s.size() * 4, st.ent_m.data(), st.ent_m.size() * 4);
    }
    st.ent_off = off;
  }
  for (nbp_stage &st : p->stages)
    if (st.fused_second) {
      size_t off = (p->blob.size() + 63) & ~(size_t)63;
      p->blob.resize(off + (st.split_out_s.size() + st.split_out_m.size()) * 4 + 4);
      if (!st.split_out_s.empty()) {
        memcpy(p->blob.data() + off, st.split_out_s.data(), st.split_out_s.size() * 4);
        memcpy(p->blob.data() + off + st.split_out_s.size() * 4, st.split_out_m.data(), st.split_out_m.size() * 4);
      }
      st.split_out_off = off;
    }
  for (nbp_stage &st : p->stages)
    if (st.fused) {
      size_t off = (p->blob.size() + 63) & ~(size_t)63;
      p->blob.resize(off + st.upd.size() * sizeof(nbp_update_desc));
      memcpy(p->blob.data() + off, st.upd.data(), st.upd.size() * sizeof(nbp_update_desc));
      st.upd_off = off;
    }
  {
    std::vector<int64_t> so;
    for (int s = 0; s < p->n_user_stages; s++) {
      const nbp_stage &st = p->stages[s];
      if (st.kind == NBP_STAGE_PROPOSALS || st.kind == NBP_STAGE_DECONV)
        for (int i = 0; i < st.n; i++) {
          const size_t o = st.offset + (size_t)i * sizeof(nbp_proposal_desc);
          so.push_back((int64_t)(o + offsetof(nbp_proposal_desc, seed)));
          // a reused measurement names another op's seed: re-keyed the same way, it keeps naming that op
          if (((const nbp_proposal_desc *)(p->blob.data() + o))->meas_seed) so.push_back((int64_t)(o + offsetof(nbp_proposal_desc, meas_seed)));
        }
      else if (st.kind == NBP_STAGE_PRODUCTS)
        for (int i = 0; i < st.n; i++) so.push_back((int64_t)(st.offset + (size_t)i * sizeof(nbp_product_desc) + offsetof(nbp_product_desc, seed)));
    }
    p->seed_off = (p->blob.size() + 63) & ~(size_t)63;
    p->n_seeds = (int)so.size();
    p->seed_val_off = p->seed_off + so.size() * 8;
    p->blob.resize(p->seed_val_off + so.size() * 8);
    if (!so.empty()) memcpy(p->blob.data() + p->seed_off, so.data(), so.size() * 8);
  }
  // size the workspaces now: nothing may allocate once a launch sequence is being captured
  nbp_status rc = NBP_OK;
  for (const nbp_stage &st : p->stages)
    if (st.kind == NBP_STAGE_PRODUCTS && st.n > 0 && !st.fused_second) {
      rc = presize_products(p->ctx, st.n, st.maxfd, st.mani);
      if (rc) return rc;
    }
  size_t bytes = p->blob.size() ? p->blob.size() : 64;
  {  // a blob a destroyed program left behind, if one is large enough (the smallest such)
    auto &bc = p->ctx->blob_cache;
    int best = -1;
    for (size_t i = 0; i < bc.size(); i++)
      if (bc[i].second >= bytes && (best < 0 || bc[i].second < bc[(size_t)best].second)) best = (int)i;
    if (best >= 0) {
      p->dev = bc[(size_t)best].first;
      p->dev_bytes = bc[(size_t)best].second;
      bc.erase(bc.begin() + best);
    }
  }
  if (!p->dev) {
    p->dev_bytes = bytes < 65536 ? 65536 : bytes;
    HIPCHK(hipMalloc(&p->dev, p->dev_bytes));
  }
  if (p->blob.size()) {
    if (p->async_upload) {
      // stream-ordered: behind whatever still reads a blob taken over from a retired program, in front of this program's launches
      nbp_ctx::pin_buf *b = pin_acquire(p->ctx, p->blob.size());
      if (!b) return fail(NBP_ERR_HIP, "pinned staging buffer");
      memcpy(b->p, p->blob.data(), p->blob.size());
      const hipError_t e = hipMemcpyAsync(p->dev, b->p, p->blob.size(), hipMemcpyHostToDevice, p->ctx->stream);
      pin_release_behind_stream(p->ctx, b);
      if (e != hipSuccess) return fail(NBP_ERR_HIP, std::string("hipMemcpyAsync (descriptors): ") + hipGetErrorString(e));
    } else {
      // (a blob from the cache may come from a RETIRED program whose launches are still queued: the library stream does not
      //  wait for the legacy stream this copy runs on)
      if (!p->ctx->retired.empty()) HIPCHK(hipStreamSynchronize(p->ctx->stream));
      HIPCHK(hipMemcpy(p->dev, p->blob.data(), p->blob.size(), hipMemcpyHostToDevice));
    }
  }
  p->finalized = true;
  return NBP_OK;
}

static nbp_status run_range(nbp_program *p, int first, int last) {
  nbp_ctx *c = p->ctx;
  auto ent_s = [&](const nbp_stage &st) { return (const int32_t *)(p->dev + st.ent_off); };
  for (int s = first; s < last; s++) {
    const nbp_stage &st = p->stages[s];
    const int nent = (int)st.ent_s.size();
    nbp_status rc = NBP_OK;
    if (st.flush_before) rc = launch_bandwidth(c, ent_s(st), ent_s(st) + nent, nent, coords_of(st.ent_m.data(), st.ent_m.size()));
    if (rc) return rc;
    if (st.kind == NBP_STAGE_PROPOSALS && st.fused && s + 1 >= last) {
      // the range ends between the two stages of a fused pair: the proposals alone, to their arena slots; their fits are the
      // entry list of the stage behind, which the end of the range runs (below)
      rc = launch_proposals(c, (const nbp_proposal_desc *)(p->dev + st.offset), st.n, st.mani);
    } else if (st.kind == NBP_STAGE_PROPOSALS && st.fused) {
      const nbp_stage &nx = p->stages[s + 1];
      rc = launch_update(c, st, (const nbp_update_desc *)(p->dev + st.upd_off), (const nbp_proposal_desc *)(p->dev + st.offset),
                         (const nbp_product_desc *)(p->dev + nx.offset), nx.n);
      s++;  // the products ran inside
    } else if (st.kind == NBP_STAGE_PROPOSALS && st.pipe && !c->timing && s + 1 < last) {
      // two-stream round (plan_pipeline): the second half runs one launch behind the first
      const nbp_stage &nx = p->stages[s + 1];
      const nbp_proposal_desc *pd = (const nbp_proposal_desc *)(p->dev + st.offset);
      const nbp_product_desc *dd = (const nbp_product_desc *)(p->dev + nx.offset);
      const int ne = (int)nx.ent_s.size(), ea = nx.pipe_ent, qa = nx.pipe_split, pa = st.pipe_split;
      const int32_t *es = ent_s(nx), *em = ent_s(nx) + ne;
      c->geom_n = nx.n;
      c->geom_blocks = 2 * ne + 2 * nx.n;
      hipStream_t s1 = c->stream;
      double *ws1 = c->ws;
      const size_t wsd1 = c->ws_doubles, wsa = (size_t)qa * (size_t)(nx.maxfd / 4) * nbp_kd_ws_doubles(c->N);
      auto half = [&](int h) -> nbp_status {
        const nbp_product_desc *dh = dd + (h ? qa : 0);
        const int nq = h ? nx.n - qa : qa, nf = h ? ne - ea : ea;
        nbp_status r = launch_prep(c, es + (h ? ea : 0), em + (h ? ea : 0), nf, dh, nq, nx.maxfd, coords_of(nx.ent_m.data() + (h ? ea : 0), nf), nx.mani);
        if (!r) r = launch_products(c, dh, nq, nx.maxfd, nx.mani);
        return r;
      };
      rc = launch_proposals(c, pd, pa, st.mani);
      hipError_t he = hipSuccess;
      if (!rc) he = hipEventRecord(c->pipe_ev[0], s1);
      if (!rc && he == hipSuccess) he = hipStreamWaitEvent(c->stream2, c->pipe_ev[0], 0);
      if (!rc && he == hipSuccess) {
        c->stream = c->stream2;
        c->ws = ws1 + wsa;
        c->ws_doubles = wsd1 - wsa;
        rc = launch_proposals(c, pd + pa, st.n - pa, st.mani);
        if (!rc) rc = half(1);
        c->stream = s1;
        c->ws = ws1;
        c->ws_doubles = wsd1;
        if (!rc) he = hipEventRecord(c->pipe_ev[1], c->stream2);
      }
      if (!rc && he == hipSuccess) rc = half(0);
      if (!rc && he == hipSuccess) he = hipStreamWaitEvent(s1, c->pipe_ev[1], 0);
      c->geom_n = c->geom_blocks = 0;
      if (he != hipSuccess) return fail(NBP_ERR_HIP, std::string("two-stream round: ") + hipGetErrorString(he));
      s++;  // the products ran here
    } else if (st.kind == NBP_STAGE_PROPOSALS) {
      rc = launch_proposals(c, (const nbp_proposal_desc *)(p->dev + st.offset), st.n, st.mani);
    } else if (st.kind == NBP_STAGE_PRODUCTS) {
      const nbp_product_desc *dd = (const nbp_product_desc *)(p->dev + st.offset);
      // (fused_second: only a range that starts between the two stages of a fused pair gets here -- three-launch form, the
      //  proposals' fits beside the KD builds, the outputs the fused launch would have fitted behind the products)
      if (st.need_prep || st.fused_second)
        rc = launch_prep(c, ent_s(st), ent_s(st) + nent, nent, dd, st.n, st.maxfd, coords_of(st.ent_m.data(), st.ent_m.size()), st.mani);
      if (!rc) rc = launch_products(c, dd, st.n, st.maxfd, st.mani);
      if (!rc && st.fused_second && !st.split_out_s.empty()) {
        const int32_t *os = (const int32_t *)(p->dev + st.split_out_off);
        const int no = (int)st.split_out_s.size();
        rc = launch_bandwidth(c, os, os + no, no, coords_of(st.split_out_m.data(), st.split_out_m.size()));
      }
    } else if (st.kind == NBP_STAGE_DECONV) {
      rc = launch_deconv(c, (const nbp_proposal_desc *)(p->dev + st.offset), nullptr, st.n);
    } else if (st.kind == NBP_STAGE_COPY_POINTS) {
      rc = launch_copy_points(c, (const nbp_copy_desc *)(p->dev + st.offset), st.n);
    } else {
      rc = launch_copies(c, (const nbp_copy_desc *)(p->dev + st.offset), st.n);
    }
    if (rc) return rc;
  }
  // leave every slot consistent: run whatever is still pending at the end of the range
  const nbp_stage &nx = p->stages[last];
  return launch_bandwidth(c, ent_s(nx), ent_s(nx) + nx.ent_s.size(), (int)nx.ent_s.size(), coords_of(nx.ent_m.data(), nx.ent_m.size()));
}

nbp_status nbp_program_run(nbp_program *p, int32_t first, int32_t last) {
  if (!p) return fail(NBP_ERR_ARG, "null argument");
  PROG_ALIVE(p);
  if (!p->finalized) return fail(NBP_ERR_ARG, "program not finalized");
  nbp_ctx *c = p->ctx;
  HIPCHK(hipSetDevice(c->device));
  const int nuser = p->n_user_stages;
  if (last < 0 || last > nuser) last = nuser;
  if (first < 0) first = 0;
  // (a range that splits a fused variable update -- a PROPOSALS stage and the PRODUCTS stage behind it -- runs that pair in
  //  the three-launch form: run_range)
  // Replay: the launch sequence of a range is captured into a hipGraph the second time it runs and launched as one
  // graph from then on (the descriptors live in device memory, so nbp_program_reseed still takes effect).  Per-kernel
  // event timing (nbp_timing_enable) and programs of a handful of launches take the plain path.
  if (c->timing || !p->use_graph || last - first < 4) return run_range(p, first, last);
  const uint64_t key = ((uint64_t)(uint32_t)first << 32) | (uint32_t)last;
  auto it = p->graphs.find(key);
  if (it != p->graphs.end() && it->second.ws_gen != c->ws_gen) {
    // a workspace was re-allocated since the capture (another program finalized on this context, an immediate-mode
    // call with a larger batch, a clique call): the graph's kernel nodes hold the freed pointer -- capture again
    HIPCHK(hipStreamSynchronize(c->stream));
    hipGraphExecDestroy(it->second.exec);
    p->graphs.erase(it);
    it = p->graphs.end();
  }
  if (it == p->graphs.end() && p->runs[key]++ == 0) return run_range(p, first, last);  // a program that runs once never pays for a capture
  if (it == p->graphs.end()) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    // (relaxed: nothing between begin and end is a call the capture could not take, and under the thread-local mode another
    //  host thread synchronising ITS context's stream meanwhile was refused with "operation not permitted when stream is
    //  capturing" -- seen in round 6 when concurrent callers' clique programs began to be replayed, tools/exp/plan_cache_repro.sh)
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    nbp_status rc = run_range(p, first, last);
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) {  // capture is not available: run directly from now on
      (void)hipGetLastError();
      p->use_graph = false;
      return run_range(p, first, last);
    }
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p->use_graph = false;
      return run_range(p, first, last);
    }
    it = p->graphs.emplace(key, nbp_program::captured{ge, c->ws_gen}).first;
  }
  HIPCHK(hipGraphLaunch(it->second.exec, c->stream));
  return NBP_OK;
  // asynchronous: nbp_synchronize / nbp_slot_read wait for completion
}

nbp_status nbp_program_reseed(nbp_program *p, uint64_t salt) {
  if (!p || !p->finalized) return fail(NBP_ERR_ARG, "program not finalized");
  PROG_ALIVE(p);
  nbp_ctx *c = p->ctx;
  HIPCHK(hipSetDevice(c->device));
  (void)hipGetLastError();
  if (p->n_seeds > 0)
    hipLaunchKernelGGL(nbp_reseed_kernel, dim3((p->n_seeds + 255) / 256), dim3(256), 0, c->stream, p->dev,
                       (const int64_t *)(p->dev + p->seed_off), p->n_seeds, salt);
  HIPCHK(hipGetLastError());
  return NBP_OK;
}

// New seeds for every op of a finalized program (the native host's plan cache: a batch of clique requests whose structure has not
// changed is the same program with other seeds).  `seeds` in stage order: per proposal / deconvolution descriptor its seed and,
// where the descriptor names a stored measurement (meas_seed != 0 when the program was finalized), that one behind it; per
// product descriptor its seed -- the order of the program's own seed table.  Stream-ordered: behind what the program has
// queued so far, in front of its next run.
__global__ void nbp_setseed_kernel(char *blob, const int64_t *seed_off, const uint64_t *vals, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *(uint64_t *)(blob + seed_off[i]) = vals[i];
}
nbp_status nbp_program_num_seeds(nbp_program *p, int32_t *out) {
  if (!p || !out) return fail(NBP_ERR_ARG, "null argument");
  if (!p->finalized) return fail(NBP_ERR_ARG, "program not finalized");
  *out = p->n_seeds;
  return NBP_OK;
}
nbp_status nbp_program_set_seeds(nbp_program *p, const uint64_t *seeds, int32_t n) {
  if (!p || !p->finalized) return fail(NBP_ERR_ARG, "program not finalized");
  PROG_ALIVE(p);
  if (n != p->n_seeds || (n > 0 && !seeds)) return fail(NBP_ERR_ARG, "set_seeds: the count is not the program's (nbp_program_num_seeds)");
  if (n == 0) return NBP_OK;
  nbp_ctx *c = p->ctx;
  HIPCHK(hipSetDevice(c->device));
  // (from a pinned buffer of the context's pool: a copy from pageable memory would make the caller wait for the stream)
  nbp_ctx::pin_buf *b = pin_acquire(c, (size_t)n * 8);
  if (!b) return fail(NBP_ERR_HIP, "pinned staging buffer");
  memcpy(b->p, seeds, (size_t)n * 8);
  const hipError_t e = hipMemcpyAsync(p->dev + p->seed_val_off, b->p, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
  pin_release_behind_stream(c, b);
  if (e != hipSuccess) return fail(NBP_ERR_HIP, std::string("hipMemcpyAsync (seeds): ") + hipGetErrorString(e));
  (void)hipGetLastError();
  hipLaunchKernelGGL(nbp_setseed_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, p->dev, (const int64_t *)(p->dev + p->seed_off),
                     (const uint64_t *)(p->dev + p->seed_val_off), n);
  HIPCHK(hipGetLastError());
  return NBP_OK;
}

nbp_status nbp_program_num_stages(nbp_program *p, int32_t *out) {
  if (!p || !out) return fail(NBP_ERR_ARG, "null argument");
  *out = p->finalized ? p->n_user_stages : (int32_t)p->stages.size();
  return NBP_OK;
}

nbp_status nbp_program_num_fused(nbp_program *p, int32_t *out) {
  if (!p || !out) return fail(NBP_ERR_ARG, "null argument");
  int n = 0;
  for (const nbp_stage &st : p->stages) n += st.fused ? 1 : 0;
  *out = n;
  return NBP_OK;
}

nbp_status nbp_program_num_two_stream(nbp_program *p, int32_t *out) {
  if (!p || !out) return fail(NBP_ERR_ARG, "null argument");
  int n = 0;
  for (const nbp_stage &st : p->stages) n += st.pipe ? 1 : 0;
  *out = n;
  return NBP_OK;
}

// the programs handed to nbp_program_retire whose last launch has run
static void program_free(nbp_program *p, bool sync);
static void reap_retired(nbp_ctx *c) {
  for (size_t i = 0; i < c->retired.size();) {
    if (hipEventQuery(c->retired[i].second) == hipSuccess) {
      hipEventDestroy(c->retired[i].second);
      nbp_program *p = c->retired[i].first;
      c->retired.erase(c->retired.begin() + (long)i);
      program_free(p, false);
    } else
      i++;
  }
}
// Destroy a program without waiting for it: it is dropped once everything queued on the library stream up to now has run
// (checked at the next retire / nbp_clique_submit_batch / nbp_synchronize; nbp_ctx_destroy takes what is left).  Its device
// blob goes to the next short-lived program only then.
nbp_status nbp_program_retire(nbp_program *p) {
  if (!p) return NBP_OK;
  if (!p->ctx) { delete p; return NBP_OK; }
  nbp_ctx *c = p->ctx;
  hipSetDevice(c->device);
  reap_retired(c);
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, c->stream) != hipSuccess) {
    if (ev) hipEventDestroy(ev);
    return nbp_program_destroy(p);  // (waits instead)
  }
  c->retired.emplace_back(p, ev);
  return NBP_OK;
}
nbp_status nbp_program_destroy(nbp_program *p) {
  if (!p) return NBP_OK;
  program_free(p, true);
  return NBP_OK;
}
static void program_free(nbp_program *p, bool sync) {
  if (p->ctx) {  // (a program whose context is gone was detached by nbp_ctx_destroy: nothing left on the device)
    hipSetDevice(p->ctx->device);
    if (sync) hipStreamSynchronize(p->ctx->stream);
    if (p->dev) {
      // small blobs are kept for the next program of this context (hipFree synchronises the whole device)
      auto &bc = p->ctx->blob_cache;
      // (a queued walk retires one program per tree level and direction before the first of them has run: room for all of them)
      if (p->dev_bytes <= (16u << 20) && bc.size() < 64) bc.emplace_back(p->dev, p->dev_bytes);
      else hipFree(p->dev);
    }
    for (auto &kv : p->graphs) hipGraphExecDestroy(kv.second.exec);
    auto &v = p->ctx->programs;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i] == p) { v.erase(v.begin() + i); break; }
  }
  delete p;
}

// ---- separator exchange between ranks: RCCL point-to-point over xGMI, from C -----------------------------------------
// The messages of a tree solve are single slots (4.9 KB at N = 200) on the few tree edges that cross a rank boundary:
// latency-bound traffic, so all messages of one exchange point go into ONE ncclGroupStart / ncclGroupEnd of ncclSend /
// ncclRecv on the library's own stream -- stream-ordered with the kernels that produce and consume the slots, no host
// synchronisation, no ring collective.  RCCL is bound at run time (dlopen + dlsym): a process that already has RCCL loaded
// (PyTorch brings its own) keeps exactly one copy, and libnbp.so has no link-time dependency on it.
struct rccl_api {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static rccl_api g_rccl;
static nbp_status rccl_load() {
  if (g_rccl.Send) return NBP_OK;
  const char *env = getenv("NBP_RCCL_LIB");
  const char *names[] = {env, "librccl.so", "librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)  // a copy that is already mapped first
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char *n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(NBP_ERR_HIP, std::string("RCCL not found (librccl.so): ") + (dlerror() ? dlerror() : ""));
  g_rccl.h = h;
#define NBP_RCCL_SYM(F) g_rccl.F = (decltype(g_rccl.F))dlsym(h, "nccl" #F); if (!g_rccl.F) return fail(NBP_ERR_HIP, "RCCL: missing symbol nccl" #F)
  NBP_RCCL_SYM(GetUniqueId);
  NBP_RCCL_SYM(CommInitRank);
  NBP_RCCL_SYM(CommDestroy);
  NBP_RCCL_SYM(CommCount);
  NBP_RCCL_SYM(CommUserRank);
  NBP_RCCL_SYM(GroupStart);
  NBP_RCCL_SYM(GroupEnd);
  NBP_RCCL_SYM(Recv);
  NBP_RCCL_SYM(GetErrorString);
  NBP_RCCL_SYM(Send);
#undef NBP_RCCL_SYM
  return NBP_OK;
}
#define RCCLCHK(expr)                                                                                  \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) return fail(NBP_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

struct nbp_comm {
  ncclComm_t comm = nullptr;
  nbp_ctx *ctx = nullptr;  // null once the context is gone (nbp_ctx_destroy shuts the communicator down first)
  int world = 0, rank = 0;
};
static void comm_shutdown(nbp_comm *m) {
  if (m->comm && m->ctx && g_rccl.CommDestroy) {
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    g_rccl.CommDestroy(m->comm);
  }
  m->comm = nullptr;
  m->ctx = nullptr;
}
// called by nbp_ctx_destroy while the stream still exists
static void ctx_detach_comms(nbp_ctx *c) {
  for (nbp_comm *m : c->comms) comm_shutdown(m);
  c->comms.clear();
}

nbp_status nbp_comm_unique_id(void *id_out) {
  if (!id_out) return fail(NBP_ERR_ARG, "null argument");
  nbp_status rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  RCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, NBP_COMM_ID_BYTES);
  return NBP_OK;
}

nbp_status nbp_comm_create(nbp_ctx *c, int32_t world, int32_t rank, const void *id, nbp_comm **out) {
  if (!c || !id || !out) return fail(NBP_ERR_ARG, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(NBP_ERR_RANGE, "comm: rank / world");
  nbp_status rc = rccl_load();
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId uid;
  static_assert(sizeof(uid) == NBP_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(&uid, id, sizeof(uid));
  nbp_comm *m = new nbp_comm();
  m->ctx = c; m->world = world; m->rank = rank;
  ncclResult_t r = g_rccl.CommInitRank(&m->comm, world, uid, rank);
  if (r != ncclSuccess) { delete m; return fail(NBP_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
  c->comms.push_back(m);
  *out = m;
  return NBP_OK;
}

nbp_status nbp_comm_destroy(nbp_comm *m) {
  if (!m) return NBP_OK;
  if (m->ctx) {
    auto &v = m->ctx->comms;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i] == m) { v.erase(v.begin() + i); break; }
  }
  comm_shutdown(m);  // nothing left to do when the context went first
  delete m;
  return NBP_OK;
}

nbp_status nbp_comm_info(nbp_comm *m, int32_t *nranks_out, int32_t *rank_out) {
  if (!m || !nranks_out || !rank_out) return fail(NBP_ERR_ARG, "null argument");
  if (!m->ctx || !m->comm) return fail(NBP_ERR_ARG, "comm_info: the communicator's context was destroyed");
  int n = 0, r = 0;
  RCCLCHK(g_rccl.CommCount(m->comm, &n));
  RCCLCHK(g_rccl.CommUserRank(m->comm, &r));
  *nranks_out = n;
  *rank_out = r;
  return NBP_OK;
}

// ---- self-test of the shared elementary functions (include/nbp_math.h) on the device -----------------------------------
__global__ void nbp_math_eval_kernel(int fn, const double *a, const double *b, double *o0, double *o1, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r0 = 0.0, r1 = 0.0;
  switch (fn) {
  case 0: r0 = nbpm_log(a[i]); break;
  case 1: nbpm_sincos(a[i], &r0, &r1); break;
  case 2: r0 = nbpm_atan2(a[i], b[i]); break;
  case 3: r0 = nbpm_wrap_pi(a[i]); break;
  default: nbpm_box_muller(a[i], b[i], &r0, &r1); break;
  }
  o0[i] = r0;
  if (o1) o1[i] = r1;
}
nbp_status nbp_math_eval(nbp_ctx *c, int32_t fn, const double *a, const double *b, double *out0, double *out1, int64_t n) {
  if (!c || !a || !out0) return fail(NBP_ERR_ARG, "null argument");
  if (fn < 0 || fn > 4) return fail(NBP_ERR_RANGE, "math_eval: fn");
  if ((fn == 2 || fn == 4) && !b) return fail(NBP_ERR_ARG, "math_eval: this function takes two arguments");
  if (n <= 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  double *d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(double) * 4 * (size_t)n));
  nbp_status rc = NBP_OK;
  do {
    if (hipMemcpyAsync(d, a, sizeof(double) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(NBP_ERR_HIP, "math_eval: copy in"); break; }
    if (b && hipMemcpyAsync(d + n, b, sizeof(double) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(NBP_ERR_HIP, "math_eval: copy in"); break; }
    hipLaunchKernelGGL(nbp_math_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (int)fn, d, b ? d + n : nullptr, d + 2 * n, d + 3 * n, (long long)n);
    if (hipMemcpyAsync(out0, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(NBP_ERR_HIP, "math_eval: copy out"); break; }
    if (out1 && hipMemcpyAsync(out1, d + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(NBP_ERR_HIP, "math_eval: copy out"); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(NBP_ERR_HIP, "math_eval: synchronize");
  } while (0);
  hipFree(d);
  return rc;
}

nbp_status nbp_exchange(nbp_ctx *c, nbp_comm *m, const nbp_xfer *sends, int32_t ns, const nbp_xfer *recvs, int32_t nr) {
  if (!c || !m || (ns > 0 && !sends) || (nr > 0 && !recvs)) return fail(NBP_ERR_ARG, "null argument");
  if (!m->ctx || !m->comm) return fail(NBP_ERR_ARG, "exchange: the communicator's context was destroyed");
  if (m->ctx != c) return fail(NBP_ERR_ARG, "exchange: the communicator belongs to another context");
  for (int i = 0; i < ns; i++)
    if (sends[i].peer < 0 || sends[i].peer >= m->world || sends[i].slot < 0 || sends[i].slot >= c->n_slots) return fail(NBP_ERR_RANGE, "exchange: send");
  for (int i = 0; i < nr; i++)
    if (recvs[i].peer < 0 || recvs[i].peer >= m->world || recvs[i].slot < 0 || recvs[i].slot >= c->n_slots) return fail(NBP_ERR_RANGE, "exchange: recv");
  if (ns + nr == 0) return NBP_OK;
  HIPCHK(hipSetDevice(c->device));
  RCCLCHK(g_rccl.GroupStart());
  // a failing Send / Recv must not leave the group open for the rest of the process: close it, then report the first error
  ncclResult_t bad = ncclSuccess;
  const char *what = "";
  for (int i = 0; i < ns && bad == ncclSuccess; i++) {
    bad = g_rccl.Send(c->arena + c->S * sends[i].slot, (size_t)c->S, ncclDouble, sends[i].peer, m->comm, c->stream);
    what = "ncclSend";
  }
  for (int i = 0; i < nr && bad == ncclSuccess; i++) {
    bad = g_rccl.Recv(c->arena + c->S * recvs[i].slot, (size_t)c->S, ncclDouble, recvs[i].peer, m->comm, c->stream);
    what = "ncclRecv";
  }
  const ncclResult_t ge = g_rccl.GroupEnd();
  if (bad != ncclSuccess) return fail(NBP_ERR_HIP, std::string(what) + ": " + g_rccl.GetErrorString(bad));
  if (ge != ncclSuccess) return fail(NBP_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
  return NBP_OK;  // stream-ordered: the next launch on the library stream sees the received slots
}

// ---- timing / diagnostics -------------------------------------------------------------------------------
nbp_status nbp_timing_enable(nbp_ctx *c, int32_t on) {
  if (!c) return fail(NBP_ERR_ARG, "null argument");
  c->timing = on != 0;
  return NBP_OK;
}
static nbp_status drain(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &ms, int64_t &cnt) {
  for (auto &p : v) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, p.first, p.second));
    ms += t;
    cnt++;
    hipEventDestroy(p.first);
    hipEventDestroy(p.second);
  }
  v.clear();
  return NBP_OK;
}
nbp_status nbp_timing_read(nbp_ctx *c, double *ms, int64_t *launches) {
  return nbp_timing_read_n(c, ms, launches, 4);
}
nbp_status nbp_timing_read_n(nbp_ctx *c, double *ms, int64_t *launches, int32_t n) {
  if (!c) return fail(NBP_ERR_ARG, "null argument");
  if (n < 0 || n > 5) return fail(NBP_ERR_RANGE, "timing: n in [0, 5]");
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 5; k++) {
    nbp_status rc = drain(c->ev[k], c->ms[k], c->nl[k]);
    if (rc) return rc;
    if (k < n && ms) ms[k] = c->ms[k];
    if (k < n && launches) launches[k] = c->nl[k];
    c->ms[k] = 0;
    c->nl[k] = 0;
  }
  return NBP_OK;
}
nbp_status nbp_diag_read(nbp_ctx *c, nbp_diag *out, int32_t reset) {
  if (!c || !out) return fail(NBP_ERR_ARG, "null argument");
  nbp_counters h;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
  out->solves = (int64_t)h.solves;
  out->nonconverged = (int64_t)h.nonconverged;
  out->nan_results = (int64_t)h.nan_results;
  out->residual_evals = (int64_t)h.residual_evals;
  out->lcv_evals = (int64_t)h.lcv_evals;
  out->lcv_evals_f32 = (int64_t)h.lcv_evals_f32;
  if (reset) HIPCHK(hipMemset(c->counters, 0, offsetof(nbp_counters, flags)));  // (the flags are the context's, not counters)
  return NBP_OK;
}

#ifdef NBP_PHASE_TIMING
nbp_status nbp_debug_phase_read(long long *out, int n, int reset) {
  long long h[64];
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(nbp_phase_clk), sizeof(h)));
  for (int i = 0; i < n && i < 64; i++) out[i] = h[i];
  if (reset) {
    memset(h, 0, sizeof(h));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nbp_phase_clk), h, sizeof(h)));
  }
  return NBP_OK;
}
nbp_status nbp_debug_block_read(long long *out, int nblocks) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nbp_block_clk), sizeof(long long) * 3 * (size_t)(nblocks < 8192 ? nblocks : 8192)));
  return NBP_OK;
}
nbp_status nbp_debug_block_hist(unsigned int *out, int reset) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nbp_block_hist), sizeof(unsigned int) * 4 * 64));
  if (reset) {
    unsigned int z[4 * 64] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nbp_block_hist), z, sizeof(z)));
  }
  return NBP_OK;
}
#endif

}  // extern "C"
