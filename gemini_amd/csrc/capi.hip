// extern "C" surface of libgemini_hip.so (declared in include/gemini_hip.h) + process context.
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "host_field.hpp"

namespace gm {

// ---- error strings -----------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return e == hipErrorOutOfMemory ? GM_ENOMEM : GM_EHIP;
}

// ---- tracked device allocation ---------------------------------------------------------------
MemStats& mem_stats() {
  static MemStats* m = new MemStats;  // never destroyed: frees may arrive from static destructors
  return *m;
}
void MemStats::note_cached(ptrdiff_t delta) {
  std::lock_guard<std::mutex> lk(mu);
  cached = (size_t)((ptrdiff_t)cached + delta);
  if (live - cached > peak_in_use) peak_in_use = live - cached;
}
hipError_t raw_malloc_v(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && *p) {
    MemStats& m = mem_stats();
    std::lock_guard<std::mutex> lk(m.mu);
    m.sizes[*p] = bytes;
    m.live += bytes;
    if (m.live > m.peak_live) m.peak_live = m.live;
    if (m.live - m.cached > m.peak_in_use) m.peak_in_use = m.live - m.cached;
  }
  return e;
}
hipError_t raw_free(void* p) {
  if (p) {
    MemStats& m = mem_stats();
    std::lock_guard<std::mutex> lk(m.mu);
    auto it = m.sizes.find(p);
    if (it != m.sizes.end()) {
      m.live -= it->second;
      m.sizes.erase(it);
    }
  }
  return hipFree(p);
}

// hipMalloc for the long-lived allocations outside the vector pool (bases, tables, matrices, index vectors): when the device
// is out of memory the pool may be holding up to 55 % of it in freed blocks -- give them back and retry once
// LOCK ORDER: this takes C->msm_mu (try-lock) and then C->mu, and it is reached from every allocation path (dev_malloc,
// DevBuf::ensure, DevPool::alloc) -- so no caller may hold C->mu across an allocation (none does: the handle tables are touched
// only around find_* / put_*).  What it frees is counted (gm_mem_stats[9]) and can be rebuilt (gm_g1_bases_precompute(h, -1)).
bool release_spare_tables(Context* C) {
  if (!C || C->msm_busy.load() != 0) return false;
  std::unique_lock<std::recursive_mutex> lk(C->msm_mu, std::try_to_lock);
  if (!lk.owns_lock() || C->msm_busy.load() != 0) return false;
  bool any = false;
  std::lock_guard<std::mutex> lk2(C->mu);
  for (auto& kv : C->bases) any = any || !kv.second->extra.empty();
  if (!any) return false;
  (void)hipDeviceSynchronize();
  for (auto& kv : C->bases) {
    for (auto& ts : kv.second->extra)
      if (ts.t) (void)gm::raw_free(ts.t);
    kv.second->extras_released = kv.second->extras_released || !kv.second->extra.empty();
    kv.second->extra.clear();
  }
  {
    MemStats& m = mem_stats();
    std::lock_guard<std::mutex> lkm(m.mu);
    m.spare_table_releases++;
  }
  return true;
}

hipError_t dev_malloc(void** p, size_t bytes) {
  hipError_t e = gm::raw_malloc(p, bytes);
  if (e == hipErrorOutOfMemory && context()) {
    (void)hipGetLastError();
    context()->pool.release_all();
    e = gm::raw_malloc(p, bytes);
    if (e == hipErrorOutOfMemory && release_spare_tables(context())) {
      (void)hipGetLastError();
      e = gm::raw_malloc(p, bytes);
    }
  }
  return e;
}

int DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return GM_OK;
  if (p) {
    (void)gm::raw_free(p);
    p = nullptr;
    cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = gm::raw_malloc(&p, want);
  if (e == hipErrorOutOfMemory && context()) {  // the vector pool may be hoarding freed blocks: give them back and retry once
    (void)hipGetLastError();
    context()->pool.release_all();
    e = gm::raw_malloc(&p, want);
    if (e == hipErrorOutOfMemory && release_spare_tables(context())) {  // then the prefix tables of the keys (not inside an MSM scope)
      (void)hipGetLastError();
      e = gm::raw_malloc(&p, want);
    }
  }
  if (e != hipSuccess) p = nullptr;
  GM_HIP(e);
  cap = want;
  return GM_OK;
}
void DevBuf::release() {
  if (p) (void)gm::raw_free(p);
  p = nullptr;
  cap = 0;
}

int DevPool::alloc(size_t bytes, void** p, size_t* cap) {
  if (bytes == 0) {
    *p = nullptr;
    *cap = 0;
    return GM_OK;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = free_list.lower_bound(bytes);
    if (it != free_list.end() && it->first <= bytes + bytes / 2 + 4096) {
      *p = it->second;
      *cap = it->first;
      pooled_bytes -= it->first;
      free_list.erase(it);
      mem_stats().note_cached(-(ptrdiff_t)*cap);
      return GM_OK;
    }
  }
  size_t want = (bytes + 4095) & ~(size_t)4095;
  hipError_t e = gm::raw_malloc(p, want);
  if (e == hipErrorOutOfMemory) {  // give cached blocks back and retry once
    (void)hipGetLastError();  // the failed attempt must not surface later as a stale hipGetLastError() of an unrelated launch
    release_all();
    e = gm::raw_malloc(p, want);
    if (e == hipErrorOutOfMemory && release_spare_tables(context())) {  // then the prefix tables of the keys
      (void)hipGetLastError();
      e = gm::raw_malloc(p, want);
    }
  }
  GM_HIP(e);
  *cap = want;
  return GM_OK;
}
void DevPool::free(void* p, size_t cap) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (pooled_bytes + cap <= max_pooled) {
      free_list.emplace(cap, p);
      pooled_bytes += cap;
      mem_stats().note_cached((ptrdiff_t)cap);
      return;
    }
  }
  (void)gm::raw_free(p);
}
void DevPool::release_all() {
  std::lock_guard<std::mutex> lk(mu);
  mem_stats().note_cached(-(ptrdiff_t)pooled_bytes);  // before the frees: live - cached must never go negative
  pooled_bytes = 0;
  for (auto& kv : free_list) (void)gm::raw_free(kv.second);
  free_list.clear();
}

void Profiler::begin(int part, int stage, hipStream_t st) {
  if (!on || (only_stage >= 0 && stage != only_stage)) return;
  if (!have_events) {
    for (auto& set : ev)
      for (auto& e : set) (void)hipEventCreate(&e);
    have_events = true;
  }
  (void)hipEventRecord(ev[part][2 * stage], st);
}
void Profiler::end(int part, int stage, hipStream_t st) {
  if (!on || (only_stage >= 0 && stage != only_stage)) return;
  (void)hipEventRecord(ev[part][2 * stage + 1], st);
  pending[part][stage] = true;
}
void Profiler::collect() {
  if (!on) return;
  for (int p = 0; p < 2; p++)
    for (int s = 0; s < PROF_NSTAGES; s++) {
      if (!pending[p][s]) continue;
      float t = 0;
      if (hipEventElapsedTime(&t, ev[p][2 * s], ev[p][2 * s + 1]) == hipSuccess) {
        ms[s] += t;
        count[s] += 1;
      }
      pending[p][s] = false;
    }
}

static Context* g_ctx = nullptr;
static std::mutex g_ctx_mu;
Context* context() { return g_ctx; }

Bases* find_bases(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  auto it = g_ctx->bases.find(h);
  return it == g_ctx->bases.end() ? nullptr : it->second.get();
}
FrVec* find_vec(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  auto it = g_ctx->vecs.find(h);
  return it == g_ctx->vecs.end() ? nullptr : it->second.get();
}
Sumcheck* find_prover(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  auto it = g_ctx->provers.find(h);
  return it == g_ctx->provers.end() ? nullptr : it->second.get();
}
uint64_t put_bases(std::unique_ptr<Bases> b) {
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  uint64_t h = g_ctx->next_handle++;
  g_ctx->bases[h] = std::move(b);
  return h;
}
uint64_t put_vec(std::unique_ptr<FrVec> v) {
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  uint64_t h = g_ctx->next_handle++;
  g_ctx->vecs[h] = std::move(v);
  return h;
}
uint64_t put_prover(std::unique_ptr<Sumcheck> p) {
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  uint64_t h = g_ctx->next_handle++;
  g_ctx->provers[h] = std::move(p);
  return h;
}

// implemented in msm.hip / fr.hip
int bases_from_host(Context* C, const void* bases, size_t stride, size_t n, std::unique_ptr<Bases>& out);
int fixed_base_generate(Context* C, const uint64_t base_affine[12], const void* d_scalars, int mont, size_t n,
                        std::unique_ptr<Bases>& out);
int bases_precompute(Context* C, Bases* b, int c);
void bases_free_tables(Bases* b);
int bases_export(Context* C, const Bases* b, size_t offset, size_t n, void* out96);
int bases_build_phi(Context* C, Bases* b);
int sc_set_herring(Sumcheck* S, int on);
int hg1_create(Context* C, const void* f_bases, size_t stride, size_t nf, const uint64_t* g_mont, size_t ng, const uint64_t twist[4],
               uint64_t* handle);
void hg1_destroy(Context* C, HerringG1* H);
int hg1_fold(Context* C, HerringG1* H, const uint64_t r[4]);
int hg1_round(Context* C, HerringG1* H, const uint64_t* challenge, uint64_t a_jac[18], uint64_t b_jac[18], int* has_msg);
int hg1_final(Context* C, HerringG1* H, uint64_t f0_jac[18], uint64_t g0[4], int* has);
int sc_create(Context* C, const void* f_src, size_t nf, const void* g_src, size_t ng, bool src_is_device,
              const uint64_t twist[4], uint64_t* handle, bool borrow = false);
void sc_destroy(Sumcheck* S);
int sc_round(Context* C, Sumcheck* S, const uint64_t* challenge, uint64_t a[4], uint64_t b[4], int* has_msg);
int sc_round_begin(Context* C, Sumcheck* S, const uint64_t* challenge, int* has_msg);
int sc_round_end(Context* C, Sumcheck* S, uint64_t a[4], uint64_t b[4]);
int sc_round_begin_many(Context* C, Sumcheck** S, size_t k, const uint64_t* challenge, int* has_msg);
int sc_fold(Context* C, Sumcheck* S, const uint64_t challenge[4]);
int sc_final(Context* C, Sumcheck* S, uint64_t f0[4], uint64_t g0[4], int* has);
int sp_create(Context* C, const void* f_stream, size_t nf, const void* g_stream, size_t ng, bool src_is_device,
              const uint64_t twist[4], uint64_t* handle, bool borrow = false);
void sp_destroy(Context* C, SpaceProver* S);
int sp_fold(Context* C, SpaceProver* S, const uint64_t challenge[4]);
int sp_round(Context* C, SpaceProver* S, const uint64_t* challenge, uint64_t a[4], uint64_t b[4], int* has_msg);
int sp_final(Context* C, SpaceProver* S, uint64_t f0[4], uint64_t g0[4], int* has);
int sp_to_time(Context* C, SpaceProver* S, uint64_t* time_handle);
int fr_stride_raw(Context* C, const uint8_t* in, size_t start, size_t stride, size_t count, uint8_t* out);
int msm_stream_create(Context* C, uint64_t bases_handle, size_t offset, int reversed, size_t chunk, size_t stride, int mont, uint64_t* handle);
void msm_stream_destroy(Context* C, MsmStream* S);
int msm_stream_add(Context* C, MsmStream* S, const void* bases_host, const void* scalars_host, size_t n);
int msm_stream_finalize(Context* C, MsmStream* S, uint64_t out_jac[18], size_t* pairs);
int fr_fold(Context* C, FrVec* f, const uint64_t r[4], FrVec* out);
int fr_powers(Context* C, const uint64_t x[4], size_t n, FrVec* out);
int fr_powers_at(Context* C, const uint64_t x[4], size_t start, size_t n, uint8_t* out);
int fr_tensor(Context* C, const uint64_t* rhos, size_t k, FrVec* out);
int fr_hadamard(Context* C, FrVec* a, FrVec* b, FrVec* out);
int fr_ip(Context* C, FrVec* a, FrVec* b, uint64_t result[4]);
int fr_eval_le(Context* C, FrVec* p, const uint64_t* xs, size_t npoints, uint64_t* results);
int fr_eval_le_batch(Context* C, FrVec* const* ps, size_t k, const uint64_t* xs, size_t npoints, uint64_t* results);
int fr_lincomb(Context* C, FrVec** polys, const uint64_t* coeffs, size_t k, FrVec* out);
int fr_scale_into(Context* C, FrVec* in, const uint64_t c[4], FrVec* out, size_t offset);
int fr_fold_chain(Context* C, FrVec* f, const uint64_t* challenges, size_t k, FrVec** outs);
int fr_scale_into_many(Context* C, FrVec** ins, const uint64_t* coeffs, size_t k, FrVec* out, const size_t* offsets);
int fr_add_at(Context* C, FrVec* v, const size_t* idx, const uint64_t* vals, size_t k);
int fr_fill(Context* C, FrVec* v, const uint64_t val[4]);
int fr_reverse(Context* C, FrVec* in, FrVec* out);
int spm_mul(Context* C, SparseMatrix* M, FrVec* x, FrVec* y);
int fr_div_linear_factors(Context* C, FrVec* f, const uint64_t* points, size_t k, FrVec* q, uint64_t* rem_out);
int fr_gather(Context* C, FrVec* src, const IdxVec* index, FrVec* out);
int fr_alg_hash(Context* C, FrVec* v, const IdxVec* index, const uint64_t zeta[4], FrVec* out, uint64_t base = 0);
int fr_tensor_gather(Context* C, const uint64_t* rhos, size_t k, const IdxVec* index, FrVec* out);
int fr_powers_gather(Context* C, const uint64_t x[4], size_t k, const IdxVec* index, FrVec* out);
int fr_plookup_set(Context* C, FrVec* v, const uint64_t y[4], const uint64_t z[4], FrVec* out);
int fr_add_scalar(Context* C, FrVec* v, const uint64_t y[4], FrVec* out);
int fr_shift_monic(Context* C, FrVec* v, FrVec* out);
int fr_acc_product(Context* C, FrVec* v, FrVec* out);
int fr_acc_product_ex(Context* C, FrVec* v, const uint64_t* carry_in, bool write_monic, FrVec* out, uint64_t* total);
int fr_tensor_range(Context* C, const uint64_t* rhos, size_t k, size_t start, size_t count, FrVec* out);
int fr_plookup_set_block(Context* C, FrVec* v, size_t v_off, size_t nv, const uint64_t* prev, size_t nout, const uint64_t y[4], const uint64_t z[4],
                         FrVec* out);
int fr_shift_block(Context* C, FrVec* v, const uint64_t first[4], size_t nout, FrVec* out);

}  // namespace gm

using namespace gm;

extern "C" {

int gm_abi_version(void) { return 1; }

const char* gm_last_error(void) { return g_err; }

int gm_init(int device) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (g_ctx) {
    GM_CHECK(g_ctx->device == device, GM_EINVAL, "gm_init: already bound to device %d (one process per GPU)", g_ctx->device);
    return GM_OK;
  }
  int count = 0;
  GM_HIP(hipGetDeviceCount(&count));
  GM_CHECK(count > 0, GM_EHIP, "gm_init: no HIP device visible");
  GM_CHECK(device >= 0 && device < count, GM_EINVAL, "gm_init: device %d out of range (%d visible)", device, count);
  GM_HIP(hipSetDevice(device));
  auto* C = new Context();
  C->device = device;
  hipDeviceProp_t prop;
  GM_HIP(hipGetDeviceProperties(&prop, device));
  C->cu_count = prop.multiProcessorCount;
  // the vector pool may keep up to 55 % of the device memory in freed blocks (158 GB of 288 on MI355X: the elastic
  // prover at 2^28 constraints recycles ~15 vectors of 8.6 GB -- 4.0 s with a 96 GB cap, 3.15 s with 160); an
  // allocation that fails gives them back and retries (DevPool::alloc, DevBuf::ensure)
  C->pool.max_pooled = prop.totalGlobalMem / 100 * 55;
  if (const char* e = getenv("GM_POOL_MAX_GB")) C->pool.max_pooled = (size_t)strtoull(e, nullptr, 10) << 30;
  // GM_STREAM_PRIO=1 (experiment): the side streams (small-call lanes, second big lane / tails of a split call) get the
  // highest priority, so their short latency-bound kernels take freed wave slots ahead of the next k_acc0 blocks
  int prio_lo = 0, prio_hi = 0;
  GM_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  const bool prio = getenv("GM_STREAM_PRIO") && atoi(getenv("GM_STREAM_PRIO")) != 0;
  GM_HIP(hipStreamCreateWithPriority(&C->stream, hipStreamNonBlocking, prio ? prio_lo : 0));
  for (int k = 0; k < MSM_SMALL_LANES; k++) GM_HIP(hipStreamCreateWithPriority(&C->small_stream[k], hipStreamNonBlocking, prio ? prio_hi : 0));
  GM_HIP(hipStreamCreateWithPriority(&C->stream_b, hipStreamNonBlocking, prio ? prio_hi : 0));
  {
    // XCD PARTITION of a batch: ONE of the eight XCDs (32 of 256 compute units, with its own L2) for the tails of every call, the
    // other seven for the accumulations.  Bit i of a CU mask is compute unit i / 8 of XCD i % 8 (consecutive bits rotate over the
    // XCDs), and a mask must cover WHOLE XCDs: the workgroups of a kernel are dealt round-robin to the XCDs its queue may use, so an
    // XCD left with a fraction of its CUs still gets its full share of workgroups and becomes the straggler -- measured, tail CUs =
    // 16 / 32 / 48 / 64 / 96 (stride 16 / 8 / 5 / 4 / 2 through the bits): psnark -i 22 485 / 334 / - / 337 / - ms against 351
    // without, the folding batch 2^20 .. 2: - / 8.2 / 21.6 / 8.2 / 15.5 ms against 8.1 (profiles/r5_cu_split_probe.txt).
    // GM_CU_SPLIT = 32 k: k XCDs for the tails (default 32); 0 switches the partition off.
    // Default ON only for the part this was measured on -- 256 CUs = 8 XCDs of 32 in SPX mode; a partitioned device (CPX: one XCD
    // per device) has no second XCD to set aside, and any other CU count has not been measured: there the partition is opt-in.
    const char* e = getenv("GM_CU_SPLIT");
    const int ncu = C->cu_count, per_xcd = ncu / 8;
    int T = e ? atoi(e) : (ncu == 256 ? per_xcd : 0);
    static const bool raw_env = getenv("GM_CU_SPLIT_RAW") != nullptr;  // experiment: any T, every (ncu / T)-th bit (how the table above was measured)
    if (!raw_env && per_xcd > 0) T = T / per_xcd * per_xcd;  // whole XCDs only
    if (ncu % 8 == 0 && T > 0 && T < ncu) {
      const int words = (ncu + 31) / 32, xcds = T / per_xcd, stride = ncu / T;
      std::vector<uint32_t> m_tail((size_t)words, 0u), m_acc((size_t)words, 0u);
      int taken = 0;
      for (int i = 0; i < ncu; i++) {
        const bool tail = raw_env ? (taken < T && i % stride == 0) : (i % 8 < xcds);
        (tail ? m_tail : m_acc)[(size_t)i >> 5] |= 1u << (i & 31);
        taken += tail ? 1 : 0;
      }
      bool ok = true;
      for (int k = 0; k < 2 + MSM_SMALL_LANES && ok; k++) {
        ok = hipExtStreamCreateWithCUMask(&C->part_acc[k], (uint32_t)words, m_acc.data()) == hipSuccess &&
             hipExtStreamCreateWithCUMask(&C->part_tail[k], (uint32_t)words, m_tail.data()) == hipSuccess;
      }
      if (ok) {
        C->cu_split = T;
        // queues created with a CU mask must be gone before the runtime tears itself down: left alive they crash the process at exit
        // under rocprofv3 (every profiled run of round 5's first measurement batch ended in a segmentation fault AFTER its output was
        // written).  Registered after the runtime's own handlers, so it runs before them.
        static bool registered = false;
        if (!registered) {
          registered = true;
          atexit([] {
            Context* c = g_ctx;
            if (!c || !c->cu_split) return;
            c->cu_split = 0;
            for (int k = 0; k < 2 + MSM_SMALL_LANES; k++) {
              if (c->part_acc[k]) (void)hipStreamSynchronize(c->part_acc[k]), (void)hipStreamDestroy(c->part_acc[k]);
              if (c->part_tail[k]) (void)hipStreamSynchronize(c->part_tail[k]), (void)hipStreamDestroy(c->part_tail[k]);
              c->part_acc[k] = c->part_tail[k] = nullptr;
            }
          });
        }
      } else {
        (void)hipGetLastError();
      }
    }
  }
  GM_HIP(hipHostMalloc((void**)&C->host_small, 1 << 16, hipHostMallocDefault));
  GM_HIP(hipHostMalloc((void**)&C->sc_desc_host, 8 << 15, hipHostMallocDefault));
  if (const char* e = getenv("GM_ZERO_COPY")) C->zero_copy = atoi(e);
  g_ctx = C;
  return GM_OK;
}

void gm_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (!g_ctx) return;
  Context* C = g_ctx;
  (void)hipStreamSynchronize(C->stream);
  for (auto& kv : C->bases) {
    bases_free_tables(kv.second.get());
    if (kv.second->phi) (void)gm::raw_free(kv.second->phi);
    if (kv.second->d) (void)gm::raw_free(kv.second->d);
  }
  for (auto& kv : C->vecs)
    if (kv.second->d) (void)gm::raw_free(kv.second->d);
  for (auto& kv : C->msm_streams) msm_stream_destroy(C, kv.second.get());
  C->pool.release_all();
  for (auto& kv : C->provers) sc_destroy(kv.second.get());
  for (auto& kv : C->space_provers) sp_destroy(C, kv.second.get());
  for (auto& kv : C->herring_g1) hg1_destroy(C, kv.second.get());
  C->partial_bufs.release_all();
  for (auto& kv : C->indices)
    if (kv.second->d) (void)gm::raw_free(kv.second->d);
  for (auto& kv : C->matrices) {
    if (kv.second->rowptr) (void)gm::raw_free(kv.second->rowptr);
    if (kv.second->cols) (void)gm::raw_free(kv.second->cols);
    if (kv.second->vals) (void)gm::raw_free(kv.second->vals);
  }
  auto release_ws = [](MsmWorkspace& w) {
    for (DevBuf* b : {&w.scalars, &w.counts, &w.offsets, &w.cursor, &w.entries, &w.tmp_entries, &w.sortmeta, &w.buckets, &w.pk[0], &w.pk[1], &w.pp[0],
                      &w.pp[1], &w.rows, &w.cols, &w.planes, &w.misc, &w.lvl_cnt, &w.lvl_pos, &w.lvl_pts[0], &w.lvl_pts[1], &w.lvl_keys[0],
                      &w.lvl_keys[1], &w.lvl_prefix, &w.lvl_lane, &w.lvl_entries, &w.lvl_n})
      b->release();
    for (int k = 0; k < MSM_SLOTS; k++)
      if (w.host_planes[k]) (void)hipHostFree(w.host_planes[k]);
    if (w.have_done_ev) {
      for (int e = 0; e < MSM_SLOTS; e++) (void)hipEventDestroy(w.done_ev[e]);
    }
  };
  release_ws(C->msm);
  release_ws(C->msm_b);
  if (C->stream_b) (void)hipStreamDestroy(C->stream_b);
  for (int k = 0; k < MSM_SMALL_LANES; k++) {
    release_ws(C->msm_small[k]);
    if (C->small_stream[k]) (void)hipStreamDestroy(C->small_stream[k]);
  }
  for (int k = 0; k < 2 + MSM_SMALL_LANES; k++) {
    if (C->part_acc[k]) (void)hipStreamDestroy(C->part_acc[k]);
    if (C->part_tail[k]) (void)hipStreamDestroy(C->part_tail[k]);
  }
  C->cu_split = 0;
  if (C->host_small) (void)hipHostFree(C->host_small);
  if (C->sc_desc_host) (void)hipHostFree(C->sc_desc_host);
  if (C->host_batch) (void)hipHostFree(C->host_batch);
  (void)hipStreamDestroy(C->stream);
  delete C;
  g_ctx = nullptr;
}

int gm_prof_enable(int on) {
  GM_CTX();
  C->prof.on = on != 0;
  C->prof.only_stage = on == 2 ? (int)PROF_ACC0 : -1;
  for (int s = 0; s < PROF_NSTAGES; s++) {
    C->prof.ms[s] = 0;
    C->prof.count[s] = 0;
    C->prof.pending[0][s] = C->prof.pending[1][s] = false;
  }
  C->prof.acc0_cycles = C->prof.acc0_ticks = 0;
  return GM_OK;
}
int gm_prof_read_clock(double* acc0_mhz) {
  GM_CTX();
  GM_CHECK(acc0_mhz != nullptr, GM_EINVAL, "prof_read_clock: null pointer");
  *acc0_mhz = C->prof.acc0_ticks > 0 ? C->prof.acc0_cycles / C->prof.acc0_ticks * 100.0 : 0.0;
  return GM_OK;
}
int gm_prof_read(double* ms_out, uint64_t* count_out, int n) {
  GM_CTX();
  GM_CHECK(n >= 0 && n <= PROF_NSTAGES, GM_EINVAL, "prof_read: n = %d, at most %d stages", n, (int)PROF_NSTAGES);
  for (int s = 0; s < n; s++) {
    if (ms_out) ms_out[s] = C->prof.ms[s];
    if (count_out) count_out[s] = C->prof.count[s];
  }
  return GM_OK;
}

int gm_set_msm_table_min(size_t n) {
  GM_CTX();
  C->msm_table_min = n;
  return GM_OK;
}

int gm_set_auto_tables(int on, size_t max_bytes) {
  GM_CTX();
  C->auto_tables = on != 0;
  C->auto_tables_max = max_bytes;
  return GM_OK;
}

int gm_pool_trim(void) {
  GM_CTX();
  C->pool.release_all();
  return GM_OK;
}

int gm_mem_stats(uint64_t out[12]) {
  GM_CTX();
  GM_CHECK(out != nullptr, GM_EINVAL, "gm_mem_stats: null output");
  size_t free_b = 0, total_b = 0;
  GM_HIP(hipMemGetInfo(&free_b, &total_b));
  size_t tables = 0, keys = 0;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    for (auto& kv : C->bases) {
      const Bases* b = kv.second.get();
      if (b->d) keys += b->n * 96;
      if (b->table) tables += (size_t)b->tab_W * b->n * 96;
      for (auto& ts : b->extra)
        if (ts.t) tables += (size_t)ts.W * ts.n * 96;
    }
  }
  size_t workspaces = msm_workspace_held(C->msm) + msm_workspace_held(C->msm_b);
  for (auto& w : C->msm_small) workspaces += msm_workspace_held(w);
  MemStats& m = mem_stats();
  std::lock_guard<std::mutex> lk(m.mu);
  out[0] = total_b;
  out[1] = free_b;
  out[2] = m.live;
  out[3] = m.peak_live;
  out[4] = m.cached;
  out[5] = m.live - m.cached;
  out[6] = m.peak_in_use;
  out[7] = tables;
  out[8] = keys;
  out[9] = m.spare_table_releases;
  out[10] = workspaces;
  out[11] = 0;
  return GM_OK;
}

int gm_runtime_info(int out[4]) {
  GM_CTX();
  GM_CHECK(out != nullptr, GM_EINVAL, "gm_runtime_info: null output");
  out[0] = C->cu_count;
  out[1] = C->cu_split;  // compute units set aside for the tails of a batch (whole XCDs; 0 = no partition)
  out[2] = C->zero_copy;
  out[3] = MSM_SMALL_LANES;
  return GM_OK;
}

int gm_mem_reset_peak(void) {
  GM_CTX();
  MemStats& m = mem_stats();
  std::lock_guard<std::mutex> lk(m.mu);
  m.peak_live = m.live;
  m.peak_in_use = m.live - m.cached;
  return GM_OK;
}

// ---- the footprint contract ----------------------------------------------------------------------------------------------------
// What a proof will allocate, BEFORE it starts: the high-water mark of its device vectors and prover buffers -- a walk of the
// prover's own alloc / release sequence (snark.cpp, psnark.cpp, psnark_elastic.cpp; every V.alloc there has its term here), in
// elements of 32 bytes -- and what the MSM workspaces still have to grow by for its largest calls.  The reference's memory story is
// its constants (README.md:38-46); a device prover that keeps everything resident owes its caller the number instead.
namespace {
struct Footprint {
  size_t vectors = 0, workspaces = 0;
};
size_t next_pow2(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
size_t fp_vectors_bytes(size_t elements) {
  // pool blocks are rounded to 4 KiB and may be reused up to 1.5 x oversized; measured peaks sit 2-4 % above the walk
  // (profiles/r5_footprint.txt): 6 % and a constant for the small buffers (partial sums, scratch, transcripts of a batch)
  const size_t b = elements * 32;
  return b + b / 16 + ((size_t)192 << 20);
}
size_t fp_workspaces(gm::Context* C, const gm::Bases* ck, size_t big0, size_t big1) {
  // the two largest calls of a batch alternate between the two full-size workspaces; the small lanes serve calls <= 2^17 pairs
  const size_t world = ck && ck->cyclic_n ? (size_t)ck->cyc_world : 1;  // a cyclic share commits to 1 / world of every polynomial
  big0 = (big0 + world - 1) / world;
  big1 = (big1 + world - 1) / world;
  size_t need = 0;
  const size_t b0 = gm::msm_workspace_bound(C, ck, big0), h0 = gm::msm_workspace_held(C->msm);
  const size_t b1 = gm::msm_workspace_bound(C, ck, big1), h1 = gm::msm_workspace_held(C->msm_b);
  need += b0 > h0 ? b0 - h0 : 0;
  need += b1 > h1 ? b1 - h1 : 0;
  const size_t bs = gm::msm_workspace_bound(C, ck, std::min<size_t>(big0, (size_t)1 << 17));
  for (auto& w : C->msm_small) {
    const size_t h = gm::msm_workspace_held(w);
    need += bs > h ? bs - h : 0;
  }
  return need;
}
// snark::Proof::{new_time, new_elastic}: n = |z|
Footprint fp_snark(gm::Context* C, const gm::Bases* ck, size_t n, int elastic) {
  const size_t nt = next_pow2(n);
  Footprint f;
  // peak = the abc phase: tensor(rho), powers(alpha), their product (3 nt), the three transposed products (3 n), abc (n); the
  // sumchecks (two or three vectors + 0.75 of each for the prover's folds) and the tensor check (levels n, the merged polynomial,
  // its quotient, one reversed copy in the elastic form: <= 5 n) stay below it.  z_a, z_b, z_c are released before it.
  size_t elems = 3 * nt + 4 * n;
  if (elastic) elems += n;  // lhs / z in both orders around the second sumcheck
  // elastic = 2, the LITERAL schedule (min_device_chunk = 1): the reversed copies the space provers and the stream MSMs read -- lhs and
  // the body in stream order (2 n) and the reversed levels of the folding tree (n / 2 + n / 4 + ... < n)
  if (elastic == 2) elems += 3 * n;
  f.vectors = fp_vectors_bytes(elems);
  f.workspaces = fp_workspaces(C, ck, n, (n + 1) / 2);
  return f;
}
// psnark::Proof::{new_time, new_elastic}: nz = |z|, nnz = joint support.  elastic: 1 = resident schedule, 2 = literal (space provers)
Footprint fp_psnark(gm::Context* C, const gm::Bases* ck, size_t nz, size_t nnz, int elastic) {
  const size_t nt = next_pow2(nz);
  // the nine lookup vectors (set, subset, sorted for r, alpha, z), each one longer as an accumulated product
  const size_t sum_l = 2 * ((nt + 2) + (nnz + 1) + (nt + nnz + 2)) + ((nz + 2) + (nnz + 1) + (nz + nnz + 2));
  const size_t sorted = 2 * (nt + nnz) + (nz + nnz);
  size_t elems;
  if (elastic == 2) {
    // literal: accumulated products and rotations in both orders (4 sum_l), the 8 reversed streams of the last four provers,
    // the time provers the space provers hand over to (<= 1.5 sum_l when the instance is below the threshold from the start)
    elems = nz + 4 * nnz + sorted + 4 * sum_l + 8 * nnz + sum_l + sum_l / 2 + 3 * nnz;
  } else {
    // the third sumcheck: r*, alpha*, ralpha*, z* (4 nnz), the sorted vectors, accumulated products and rotations (2 sum_l), the
    // three left-hand sides (3 nnz), and 13 time provers with 0.75 of each of their 26 vectors in fold buffers
    elems = 4 * nnz + sorted + 2 * sum_l + 3 * nnz + (3 * (2 * sum_l) + 3 * (8 * nnz)) / 4 + 16;
    if (elastic) elems += nz + (2 * nz + 2);  // w in little-endian order; one reversed copy of the longest committed vector
  }
  Footprint f;
  f.vectors = fp_vectors_bytes(elems);
  const size_t longest = nt + nnz + 2;
  f.workspaces = fp_workspaces(C, ck, longest, longest);
  return f;
}
// gm_psnark_new_time_sharded on `world` ranks with block size `block`: every vector of the resident schedule above in blocks (a family's block is
// block >> level >= its length / world, within 1 / 32 of it: the rounding of gm_psnark_shard_block); nothing is held WHOLE but z, the caller's
// (tensor(rho), powers(alpha) are table lookups) -- plus the temporaries of the cross-level combinations and the ranges of the hashed sets
// (per-level partial sums and their re-blocked copies: <= 6 coarsest blocks)
Footprint fp_psnark_shard(gm::Context* C, const gm::Bases* key, size_t nrows, size_t nz, size_t nnz, size_t block, size_t world) {
  const size_t nt = next_pow2(nrows);
  const size_t sum_l = 2 * ((nt + 2) + (nnz + 1) + (nt + nnz + 2)) + ((nz + 2) + (nnz + 1) + (nz + nnz + 2));
  const size_t sorted = 2 * (nt + nnz) + (nz + nnz);
  const size_t whole = 4 * nnz + sorted + 2 * sum_l + 3 * nnz + (3 * (2 * sum_l) + 3 * (8 * nnz)) / 4 + 5 * nnz + nz + 16;  // (+ the instance's own blocks: 5 nnz + w)
  const size_t g = world ? world : 1;
  size_t elems = (whole + g - 1) / g;
  (void)nt;  // (tensor(rho) / powers(alpha) are never built: table lookups; the ranges of the hashed sets are blocks)
  elems += elems / 32 + 6 * block;
  Footprint f;
  f.vectors = fp_vectors_bytes(elems);
  f.workspaces = fp_workspaces(C, key, block, block);
  return f;
}
// Is there room?  Freed pool blocks are given back on demand (dev_malloc); if that is not enough the PREFIX tables go now, before
// the proof starts, instead of by reflex after an allocation has failed half-way; if it still does not fit: GM_ENOMEM with the numbers.
int fp_fill_and_admit(gm::Context* C, const Footprint& f, uint64_t out[4], bool admit, const char* what) {
  size_t free_b = 0, total_b = 0;
  GM_HIP(hipMemGetInfo(&free_b, &total_b));
  size_t cached = 0, spare = 0;
  {
    std::lock_guard<std::mutex> lk(C->pool.mu);
    cached = C->pool.pooled_bytes;
  }
  {
    std::lock_guard<std::mutex> lk(C->mu);
    for (auto& kv : C->bases)
      for (auto& ts : kv.second->extra)
        if (ts.t) spare += (size_t)ts.W * ts.n * 96;
  }
  const size_t need = f.vectors + f.workspaces, reserve = (size_t)1 << 30;  // the runtime's own allocations
  if (out) {
    out[0] = f.vectors;
    out[1] = f.workspaces;
    out[2] = need;
    out[3] = free_b + cached + spare > reserve ? free_b + cached + spare - reserve : 0;
  }
  if (!admit) return GM_OK;
  if (need + reserve <= free_b + cached) return GM_OK;
  if (need + reserve <= free_b + cached + spare && gm::release_spare_tables(C)) return GM_OK;
  GM_CHECK(false, GM_ENOMEM,
           "%s: the proof needs %.1f GB of device memory (vectors and prover buffers %.1f GB, MSM workspaces still to grow %.1f GB) and %.1f GB can "
           "be had (free %.1f, pool cache %.1f, prefix tables %.1f); free vectors, register a key without tables (gm_set_auto_tables) or shard",
           what, need / 1e9, f.vectors / 1e9, f.workspaces / 1e9, (free_b + cached + spare) / 1e9, free_b / 1e9, cached / 1e9, spare / 1e9);
  return GM_OK;
}
}  // namespace

int gm_snark_footprint(uint64_t ck_bases, size_t num_constraints, int elastic, uint64_t out[4]) {
  GM_CTX();
  const Bases* ck = find_bases(ck_bases);
  GM_CHECK(ck != nullptr && out != nullptr, GM_EHANDLE, "gm_snark_footprint: unknown key handle or null output");
  return fp_fill_and_admit(C, fp_snark(C, ck, num_constraints, elastic), out, false, "gm_snark_footprint");
}
int gm_psnark_footprint(uint64_t ck_bases, size_t num_variables, size_t nnz, int elastic, uint64_t out[4]) {
  GM_CTX();
  const Bases* ck = find_bases(ck_bases);
  GM_CHECK(ck != nullptr && out != nullptr, GM_EHANDLE, "gm_psnark_footprint: unknown key handle or null output");
  return fp_fill_and_admit(C, fp_psnark(C, ck, num_variables, nnz, elastic), out, false, "gm_psnark_footprint");
}
int gm_psnark_shard_footprint(uint64_t key, size_t num_constraints, size_t num_variables, size_t nnz, size_t block, int world, int admit, uint64_t out[4]) {
  GM_CTX();
  const Bases* ck = find_bases(key);
  GM_CHECK(ck != nullptr && (out != nullptr || admit), GM_EHANDLE, "gm_psnark_shard_footprint: unknown key handle or null output");
  static const bool off = getenv("GM_FOOTPRINT_CHECK") && atoi(getenv("GM_FOOTPRINT_CHECK")) == 0;
  return fp_fill_and_admit(C, fp_psnark_shard(C, ck, num_constraints, num_variables, nnz, block, world > 0 ? (size_t)world : 1), out, admit && !off,
                           "block-sharded psnark prover");
}
// called by the provers compiled into the library before their first allocation
int gm_footprint_admit(int psnark, uint64_t ck_bases, size_t n, size_t nnz, int elastic) {
  GM_CTX();
  static const bool off = getenv("GM_FOOTPRINT_CHECK") && atoi(getenv("GM_FOOTPRINT_CHECK")) == 0;
  if (off) return GM_OK;
  const Bases* ck = find_bases(ck_bases);
  GM_CHECK(ck != nullptr, GM_EHANDLE, "footprint: unknown key handle");
  const Footprint f = psnark ? fp_psnark(C, ck, n, nnz, elastic) : fp_snark(C, ck, n, elastic);
  return fp_fill_and_admit(C, f, nullptr, true, psnark ? "psnark prover" : "snark prover");
}

int gm_g1_release_spare_tables(void) {
  GM_CTX();
  (void)release_spare_tables(C);
  return GM_OK;
}

int gm_set_msm_window(int c) {
  GM_CTX();
  GM_CHECK(c == 0 || (c >= 2 && c <= 22), GM_EINVAL, "gm_set_msm_window: c = %d not in {0} u [2, 22]", c);
  C->msm_c_override = c;
  return GM_OK;
}

int gm_set_msm_glv(int on) {
  GM_CTX();
#ifndef GM_EXPERIMENTS
  GM_CHECK(on == 0, GM_EINVAL, "gm_set_msm_glv: the GLV experiment is not in this build (make EXTRA=-DGM_EXPERIMENTS; DESIGN.md section 8)");
#endif
  C->msm_glv = on != 0;
  return GM_OK;
}

int gm_set_msm_split(int on) {
  GM_CTX();
#ifndef GM_EXPERIMENTS
  GM_CHECK(on == 0, GM_EINVAL, "gm_set_msm_split: the window-group experiment is not in this build (make EXTRA=-DGM_EXPERIMENTS; DESIGN.md section 8)");
#endif
  C->msm_split = on != 0;
  return GM_OK;
}

int gm_set_msm_affine_levels(int levels) {
  GM_CTX();
  GM_CHECK(levels >= -1 && levels <= 8, GM_EINVAL, "gm_set_msm_affine_levels: %d not in [-1, 8]", levels);
#ifndef GM_EXPERIMENTS
  if (levels == -1) levels = 0;  // automatic = no affine levels in a build without the experiment
  GM_CHECK(levels == 0, GM_EINVAL, "gm_set_msm_affine_levels: the affine-level experiment is not in this build (make EXTRA=-DGM_EXPERIMENTS; DESIGN.md section 8)");
#endif
  C->msm_affine_levels = levels;
  return GM_OK;
}

// ---- bases ---------------------------------------------------------------------------------
// Fixed-base window tables by default (gm_set_auto_tables): a registered key is resident for many MSMs -- the KZG key of a
// prover (src/kzg/time.rs:49-72 builds a window table of its own to GENERATE the key) -- so when W x n x 96 bytes fit the
// budget (default: 30 % of the device memory) the tables are built at registration, outside every prover span.
static int maybe_auto_tables(Context* C, Bases* b) {
  if (!C->auto_tables || b->n < C->msm_table_min || b->n < ((size_t)1 << 17)) return GM_OK;
  // 2^26: the pair-index field of a table entry (msm.hip: ENTRY_W_SHIFT).  Longer keys -- below 2^28 points: the key of `snark -i 26`; the provers at 2^27 / 2^28 constraints need the memory for their vectors --
  // get tables over their first points only (bases_precompute); under memory pressure they are given back (release_spare_tables)
  const bool prefix_only = b->n >= ((size_t)1 << 26);
  if (prefix_only && b->n >= ((size_t)1 << 28)) return GM_OK;
  const int c = b->n >= ((size_t)1 << 23) ? 22 : 20;
  const size_t W = (256 + c - 1) / c;
  const size_t bytes = (prefix_only ? (size_t)12 * 96 << 25 : W * b->n * 96) + (std::min<size_t>(b->n, (size_t)1 << 22) * 192) +
                       (c >= 22 ? (size_t)13 * 96 << 22 : 0) + ((size_t)16 * 96 << 17);  // + the prefix tables
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return GM_OK;
  const size_t budget = C->auto_tables_max ? C->auto_tables_max : total_b / 100 * 30;
  size_t pooled = 0;
  {
    std::lock_guard<std::mutex> lk(C->pool.mu);
    pooled = C->pool.pooled_bytes;
  }
  if (bytes > budget || bytes + ((size_t)2 << 30) > free_b + pooled) return GM_OK;  // no room: the plain path serves this key
  const int rc = bases_precompute(C, b, 0);
  return rc == GM_ENOMEM ? GM_OK : rc;
}

int gm_g1_bases_register(const void* bases, size_t base_stride, size_t n, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(handle != nullptr && (bases != nullptr || n == 0), GM_EINVAL, "bases_register: null pointer");
  std::unique_ptr<Bases> b;
  int rc = bases_from_host(C, bases, base_stride, n, b);
  if (rc) return rc;
  if ((rc = bases_build_phi(C, b.get()))) return rc;
  // no tables here: a plain registration may be one-shot (VariableBaseMSM::msm = register, one MSM, free) and a table build
  // costs ~240 doublings per point; the key constructors (gm_g1_srs_register*, gm_g1_fixed_base_register) build them,
  // and gm_g1_bases_precompute(handle, -1) does for a key uploaded from the host.
  *handle = put_bases(std::move(b));
  return GM_OK;
}

int gm_g1_bases_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<Bases> b;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->bases.find(handle);
    GM_CHECK(it != C->bases.end(), GM_EHANDLE, "bases_free: unknown handle %llu", (unsigned long long)handle);
    b = std::move(it->second);
    C->bases.erase(it);
  }
  bases_free_tables(b.get());
  if (b->phi) (void)gm::raw_free(b->phi);
  if (b->d) GM_HIP(gm::raw_free(b->d));
  return GM_OK;
}

int gm_g1_bases_precompute(uint64_t handle, int c) {
  GM_CTX();
  Bases* b = find_bases(handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "bases_precompute: unknown handle %llu", (unsigned long long)handle);
  if (c == -1 && b->extras_released) return bases_build_prefix_sets(C, b);  // given back under memory pressure: rebuilt on demand
  if (c == -1) return (b->table || !b->extra.empty()) ? GM_OK : maybe_auto_tables(C, b);  // automatic: the rule of the key constructors (size range, budget, free memory)
  return bases_precompute(C, b, c);
}

int gm_g1_bases_table_info(uint64_t handle, int* c, size_t* bytes) {
  GM_CTX();
  Bases* b = find_bases(handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "bases_table_info: unknown handle %llu", (unsigned long long)handle);
  if (c) *c = b->table ? b->tab_c : 0;
  if (bytes) {
    *bytes = b->table ? (size_t)b->tab_W * b->n * 96 : 0;
    for (const auto& ts : b->extra) *bytes += (size_t)ts.W * ts.n * 96;  // prefix tables for the calls below the main table's range
  }
  return GM_OK;
}

int gm_g1_bases_len(uint64_t handle, size_t* n) {
  GM_CTX();
  Bases* b = find_bases(handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "bases_len: unknown handle %llu", (unsigned long long)handle);
  *n = b->n;
  return GM_OK;
}

int gm_g1_bases_download(uint64_t handle, size_t offset, size_t n, void* out96) {
  GM_CTX();
  Bases* b = find_bases(handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "bases_download: unknown handle %llu", (unsigned long long)handle);
  GM_CHECK(offset + n <= b->n, GM_EINVAL, "bases_download: range [%zu, %zu) outside %zu bases", offset, offset + n, b->n);
  return bases_export(C, b, offset, n, out96);
}

static int msm_host_scalars(Context* C, Bases* b, size_t offset, int reversed, const uint64_t* scalars, size_t n,
                            uint64_t out_jac[18]) {
  int rc;
  GM_MSM_LOCK(C);  // held across the upload AND the MSM: another thread must not restage C->msm.scalars in between
  if ((rc = C->msm.scalars.ensure(n * 32 + 32))) return rc;
  if (n) GM_HIP(hipMemcpyAsync(C->msm.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, C->stream));
  return msm_run(C, b, (int64_t)offset, reversed ? -1 : 1, C->msm.scalars.p, 0, n, true, out_jac);
}

int gm_g1_msm(const void* bases, size_t base_stride, const uint64_t* scalars, size_t n, uint64_t out_jac[18]) {
  GM_CTX();
  GM_CHECK(out_jac != nullptr && ((bases != nullptr && scalars != nullptr) || n == 0), GM_EINVAL, "msm: null pointer");
  if (n >= ((size_t)1 << 22)) {
    // large one-shot calls stream: the copy of chunk i + 1 (and the conversion of its bases) under the kernels of
    // chunk i, O(chunk) device memory.  The sum does not depend on the cut (tests/test_gpu_msm_stream.py).
    uint64_t h = 0;
    int rc = msm_stream_create(C, 0, 0, 0, n >= ((size_t)1 << 24) ? (size_t)1 << 22 : (size_t)1 << 20, base_stride, 0, &h);
    if (rc) return rc;
    std::unique_ptr<MsmStream> S;
    {
      std::lock_guard<std::mutex> lk(C->mu);
      auto it = C->msm_streams.find(h);
      S = std::move(it->second);
      C->msm_streams.erase(it);
    }
    rc = msm_stream_add(C, S.get(), bases, scalars, n);
    if (!rc) rc = msm_stream_finalize(C, S.get(), out_jac, nullptr);
    GM_MSM_LOCK(C);
    msm_stream_destroy(C, S.get());
    return rc;
  }
  std::unique_ptr<Bases> b;
  int rc = bases_from_host(C, bases, base_stride, n, b);
  if (rc) return rc;
  rc = msm_host_scalars(C, b.get(), 0, 0, scalars, n, out_jac);
  if (b->d) (void)gm::raw_free(b->d);
  return rc;
}

int gm_g1_msm_h(uint64_t handle, size_t offset, int reversed, const uint64_t* scalars, size_t n, uint64_t out_jac[18]) {
  GM_CTX();
  Bases* b = find_bases(handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_h: unknown bases handle %llu", (unsigned long long)handle);
  GM_CHECK(out_jac != nullptr && (scalars != nullptr || n == 0), GM_EINVAL, "msm_h: null pointer");
  return msm_host_scalars(C, b, offset, reversed, scalars, n, out_jac);
}

int gm_g1_msm_v(uint64_t bases_handle, size_t offset, int reversed, uint64_t vec_handle, size_t voffset, size_t n,
                uint64_t out_jac[18]) {
  GM_CTX();
  Bases* b = find_bases(bases_handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_v: unknown bases handle %llu", (unsigned long long)bases_handle);
  FrVec* v = find_vec(vec_handle);
  GM_CHECK(v != nullptr, GM_EHANDLE, "msm_v: unknown vector handle %llu", (unsigned long long)vec_handle);
  GM_CHECK(voffset + n <= v->len, GM_EINVAL, "msm_v: range [%zu, %zu) outside vector of length %zu", voffset, voffset + n, v->len);
  return msm_run(C, b, (int64_t)offset, reversed ? -1 : 1, v->d + voffset * 32, 1, n, true, out_jac);
}

int gm_g1_msm_v_batch(uint64_t bases_handle, size_t offset, int reversed, const uint64_t* vec_handles, const size_t* ns, size_t k,
                      uint64_t* out_jac) {
  GM_CTX();
  Bases* b = find_bases(bases_handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_v_batch: unknown bases handle %llu", (unsigned long long)bases_handle);
  GM_CHECK(k == 0 || (vec_handles && ns && out_jac), GM_EINVAL, "msm_v_batch: null pointer");
  std::vector<const void*> ptrs(k);
  for (size_t j = 0; j < k; j++) {
    FrVec* v = find_vec(vec_handles[j]);
    GM_CHECK(v != nullptr, GM_EHANDLE, "msm_v_batch: unknown vector handle %llu", (unsigned long long)vec_handles[j]);
    GM_CHECK(ns[j] <= v->len, GM_EINVAL, "msm_v_batch: %zu pairs from a vector of length %zu", ns[j], v->len);
    ptrs[j] = v->d;
  }
  return msm_run_batch(C, b, (int64_t)offset, reversed ? -1 : 1, ptrs.data(), 1, ns, k, true, out_jac);
}

int gm_g1_msm_v_batch_partial(uint64_t bases_handle, size_t offset, int reversed, const uint64_t* vec_handles, const size_t* ns, size_t k,
                              uint64_t* out_jac) {
  GM_CTX();
  Bases* b = find_bases(bases_handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_v_batch_partial: unknown bases handle %llu", (unsigned long long)bases_handle);
  GM_CHECK(k == 0 || (vec_handles && ns && out_jac), GM_EINVAL, "msm_v_batch_partial: null pointer");
  std::vector<const void*> ptrs(k);
  for (size_t j = 0; j < k; j++) {
    FrVec* v = find_vec(vec_handles[j]);
    GM_CHECK(v != nullptr, GM_EHANDLE, "msm_v_batch_partial: unknown vector handle %llu", (unsigned long long)vec_handles[j]);
    GM_CHECK(ns[j] <= v->len, GM_EINVAL, "msm_v_batch_partial: %zu pairs from a vector of length %zu", ns[j], v->len);
    ptrs[j] = v->d;
  }
  return msm_run_batch(C, b, (int64_t)offset, reversed ? -1 : 1, ptrs.data(), 1, ns, k, false, out_jac);
}

int gm_g1_msm_v_batch_at(uint64_t bases_handle, const size_t* offsets, int reversed, const uint64_t* vec_handles, const size_t* ns, size_t k,
                         int partial, uint64_t* out_jac) {
  GM_CTX();
  Bases* b = find_bases(bases_handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_v_batch_at: unknown bases handle %llu", (unsigned long long)bases_handle);
  GM_CHECK(k == 0 || (offsets && vec_handles && ns && out_jac), GM_EINVAL, "msm_v_batch_at: null pointer");
  std::vector<const void*> ptrs(k);
  for (size_t j = 0; j < k; j++) {
    FrVec* v = find_vec(vec_handles[j]);
    GM_CHECK(v != nullptr, GM_EHANDLE, "msm_v_batch_at: unknown vector handle %llu", (unsigned long long)vec_handles[j]);
    GM_CHECK(ns[j] <= v->len, GM_EINVAL, "msm_v_batch_at: %zu pairs from a vector of length %zu", ns[j], v->len);
    GM_CHECK(offsets[j] <= ((size_t)1 << 62), GM_EINVAL, "msm_v_batch_at: offset %zu", offsets[j]);
    ptrs[j] = v->d;
  }
  return msm_run_batch_offsets(C, b, offsets, reversed ? -1 : 1, ptrs.data(), 1, ns, k, partial == 0, out_jac);
}

int gm_g1_msm_d(uint64_t bases_handle, size_t offset, int reversed, const void* d_scalars, int mont, size_t n,
                uint64_t out_jac[18]) {
  GM_CTX();
  Bases* b = find_bases(bases_handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_d: unknown bases handle %llu", (unsigned long long)bases_handle);
  GM_CHECK(out_jac != nullptr && (d_scalars != nullptr || n == 0), GM_EINVAL, "msm_d: null pointer");
  return msm_run(C, b, (int64_t)offset, reversed ? -1 : 1, d_scalars, mont, n, true, out_jac);
}

int gm_g1_msm_d_partial(uint64_t bases_handle, size_t offset, int reversed, const void* d_scalars, int mont, size_t n,
                        uint64_t out_jac[18]) {
  GM_CTX();
  Bases* b = find_bases(bases_handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "msm_d_partial: unknown bases handle %llu", (unsigned long long)bases_handle);
  GM_CHECK(out_jac != nullptr && (d_scalars != nullptr || n == 0), GM_EINVAL, "msm_d_partial: null pointer");
  return msm_run(C, b, (int64_t)offset, reversed ? -1 : 1, d_scalars, mont, n, false, out_jac);
}

// ---- streaming MSM over host-resident pairs (ChunkedPippenger / msm_chunks) ------------------------
static MsmStream* find_msm_stream(Context* C, uint64_t h) {
  std::lock_guard<std::mutex> lk(C->mu);
  auto it = C->msm_streams.find(h);
  return it == C->msm_streams.end() ? nullptr : it->second.get();
}
int gm_g1_msm_stream_new(size_t chunk_pairs, size_t base_stride, int scalars_mont, uint64_t* stream) {
  GM_CTX();
  GM_CHECK(stream != nullptr, GM_EINVAL, "msm_stream_new: null pointer");
  return msm_stream_create(C, 0, 0, 0, chunk_pairs, base_stride, scalars_mont, stream);
}
int gm_g1_msm_stream_new_h(uint64_t bases_handle, size_t offset, int reversed, size_t chunk_pairs, int scalars_mont, uint64_t* stream) {
  GM_CTX();
  GM_CHECK(stream != nullptr, GM_EINVAL, "msm_stream_new_h: null pointer");
  GM_CHECK(find_bases(bases_handle) != nullptr, GM_EHANDLE, "msm_stream_new_h: unknown bases handle %llu", (unsigned long long)bases_handle);
  return msm_stream_create(C, bases_handle, offset, reversed, chunk_pairs, 96, scalars_mont, stream);
}
int gm_g1_msm_stream_add(uint64_t stream, const void* bases_host, const void* scalars_host, size_t n) {
  GM_CTX();
  MsmStream* S = find_msm_stream(C, stream);
  GM_CHECK(S != nullptr, GM_EHANDLE, "msm_stream_add: unknown stream handle %llu", (unsigned long long)stream);
  GM_CHECK(n == 0 || (scalars_host != nullptr && (bases_host != nullptr || S->bases_handle != 0)), GM_EINVAL, "msm_stream_add: null pointer");
  return msm_stream_add(C, S, bases_host, scalars_host, n);
}
int gm_g1_msm_stream_finalize(uint64_t stream, uint64_t out_jac[18], size_t* pairs_or_null) {
  GM_CTX();
  MsmStream* S = find_msm_stream(C, stream);
  GM_CHECK(S != nullptr, GM_EHANDLE, "msm_stream_finalize: unknown stream handle %llu", (unsigned long long)stream);
  GM_CHECK(out_jac != nullptr, GM_EINVAL, "msm_stream_finalize: null pointer");
  return msm_stream_finalize(C, S, out_jac, pairs_or_null);
}
int gm_g1_msm_stream_free(uint64_t stream) {
  GM_CTX();
  std::unique_ptr<MsmStream> p;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->msm_streams.find(stream);
    GM_CHECK(it != C->msm_streams.end(), GM_EHANDLE, "msm_stream_free: unknown handle %llu", (unsigned long long)stream);
    p = std::move(it->second);
    C->msm_streams.erase(it);
  }
  GM_MSM_LOCK(C);
  msm_stream_destroy(C, p.get());
  return GM_OK;
}
// page-locked host memory for the streams: copied by DMA at the PCIe rate instead of through the runtime's staging buffer
int gm_host_alloc(size_t bytes, void** p) {
  GM_CTX();
  GM_CHECK(p != nullptr, GM_EINVAL, "host_alloc: null pointer");
  GM_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
  return GM_OK;
}
int gm_host_free(void* p) {
  GM_CTX();
  if (p) GM_HIP(hipHostFree(p));
  return GM_OK;
}

int gm_g1_sum(const uint64_t* points_jac, size_t k, uint64_t out_jac[18]) {
  GM_CHECK(out_jac != nullptr && (points_jac != nullptr || k == 0), GM_EINVAL, "g1_sum: null pointer");
  gmh::G1 acc = gmh::G1::identity();
  for (size_t i = 0; i < k; i++) acc = acc.add(gmh::G1::from_limbs(points_jac + 18 * i));
  acc.normalized().to_limbs(out_jac);
  return GM_OK;
}

int gm_g1_fixed_base_register(const uint64_t base_affine[12], const uint64_t* scalars, size_t n, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(base_affine && handle && (scalars || n == 0), GM_EINVAL, "fixed_base_register: null pointer");
  uint8_t* d_sc = nullptr;
  if (n) {
    GM_HIP(dev_malloc((void**)&d_sc, n * 32));
    GM_HIP(hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, C->stream));
  }
  std::unique_ptr<Bases> b;
  int rc = fixed_base_generate(C, base_affine, d_sc, 0, n, b);
  if (d_sc) (void)gm::raw_free(d_sc);
  if (rc) return rc;
  if ((rc = bases_build_phi(C, b.get()))) return rc;
  if ((rc = maybe_auto_tables(C, b.get()))) return rc;
  *handle = put_bases(std::move(b));
  return GM_OK;
}

int gm_g1_srs_register(const uint64_t base_affine[12], const uint64_t tau[4], size_t n, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(base_affine && tau && handle, GM_EINVAL, "srs_register: null pointer");
  // powers of tau on device (Montgomery), then fixed-base multiplication
  auto v = std::make_unique<FrVec>();
  v->cap = n;
  if (n) GM_HIP(dev_malloc((void**)&v->d, n * 32));
  gmh::Fr t = gmh::Fr::from_canonical(tau);
  uint64_t tm[4];
  t.to_limbs(tm);
  int rc = fr_powers(C, tm, n, v.get());
  std::unique_ptr<Bases> b;
  if (!rc) rc = fixed_base_generate(C, base_affine, v->d, 1, n, b);
  if (v->d) (void)gm::raw_free(v->d);
  if (rc) return rc;
  if ((rc = bases_build_phi(C, b.get()))) return rc;
  if ((rc = maybe_auto_tables(C, b.get()))) return rc;
  *handle = put_bases(std::move(b));
  return GM_OK;
}

int gm_g1_srs_register_segments(const uint64_t base_affine[12], const uint64_t tau[4], const size_t* starts, const size_t* counts, size_t nseg,
                                uint64_t* handle) {
  GM_CTX();
  GM_CHECK(base_affine && tau && handle && (nseg == 0 || (starts && counts)), GM_EINVAL, "srs_register_segments: null pointer");
  size_t n = 0;
  for (size_t s = 0; s < nseg; s++) {
    GM_CHECK(counts[s] <= ((size_t)1 << 40) && n + counts[s] >= n, GM_EINVAL, "srs_register_segments: segment %zu has %zu powers", s, counts[s]);
    n += counts[s];
  }
  // the exponents' images on device: segment s = tau^starts[s] * (1, tau, tau^2, ...), then one fixed-base pass over all of them
  auto v = std::make_unique<FrVec>();
  v->cap = n;
  if (n) GM_HIP(dev_malloc((void**)&v->d, n * 32));
  gmh::Fr t = gmh::Fr::from_canonical(tau);
  uint64_t tm[4];
  t.to_limbs(tm);
  int rc = GM_OK;
  size_t at = 0;
  for (size_t s = 0; s < nseg && !rc; s++) {
    rc = fr_powers_at(C, tm, starts[s], counts[s], v->d + at * 32);
    at += counts[s];
  }
  std::unique_ptr<Bases> b;
  if (!rc) rc = fixed_base_generate(C, base_affine, v->d, 1, n, b);
  if (v->d) (void)gm::raw_free(v->d);
  v->d = nullptr;
  if (rc) return rc;
  if ((rc = bases_build_phi(C, b.get()))) return rc;
  if ((rc = maybe_auto_tables(C, b.get()))) return rc;
  *handle = put_bases(std::move(b));
  return GM_OK;
}

// ---- Fr vectors ------------------------------------------------------------------------------
int gm_fr_vec_alloc(size_t n, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(handle != nullptr, GM_EINVAL, "vec_alloc: null handle pointer");
  auto v = std::make_unique<FrVec>();
  v->cap = n;
  v->len = n;
  size_t cap_bytes = 0;
  int rc = C->pool.alloc(n * 32, (void**)&v->d, &cap_bytes);
  if (rc) return rc;
  v->cap_bytes = cap_bytes;
  *handle = put_vec(std::move(v));
  return GM_OK;
}
int gm_fr_vec_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<FrVec> v;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->vecs.find(handle);
    GM_CHECK(it != C->vecs.end(), GM_EHANDLE, "vec_free: unknown handle %llu", (unsigned long long)handle);
    v = std::move(it->second);
    C->vecs.erase(it);
  }
  // stream-ordered reuse is safe: every kernel of this library runs on C->stream and entry points
  // return after synchronising it
  C->pool.free(v->d, v->cap_bytes);
  return GM_OK;
}
#define GM_VEC(var, h, who)                  \
  FrVec* var = find_vec(h);                  \
  GM_CHECK(var != nullptr, GM_EHANDLE, who ": unknown vector handle %llu", (unsigned long long)(h))

int gm_fr_vec_len(uint64_t handle, size_t* n) {
  GM_CTX();
  GM_VEC(v, handle, "vec_len");
  *n = v->len;
  return GM_OK;
}
int gm_fr_vec_set_len(uint64_t handle, size_t n) {
  GM_CTX();
  GM_VEC(v, handle, "vec_set_len");
  GM_CHECK(n <= v->cap, GM_EINVAL, "vec_set_len: %zu exceeds capacity %zu", n, v->cap);
  v->len = n;
  return GM_OK;
}
int gm_fr_vec_ptr(uint64_t handle, void** dptr) {
  GM_CTX();
  GM_VEC(v, handle, "vec_ptr");
  *dptr = v->d;
  return GM_OK;
}
int gm_fr_vec_upload(uint64_t handle, size_t offset, const uint64_t* mont, size_t n) {
  GM_CTX();
  GM_VEC(v, handle, "vec_upload");
  GM_CHECK(offset + n <= v->cap, GM_EINVAL, "vec_upload: range [%zu, %zu) outside capacity %zu", offset, offset + n, v->cap);
  if (n) {
    GM_HIP(hipMemcpyAsync(v->d + offset * 32, mont, n * 32, hipMemcpyHostToDevice, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
  }
  return GM_OK;
}
int gm_fr_vec_download(uint64_t handle, size_t offset, uint64_t* mont, size_t n) {
  GM_CTX();
  GM_VEC(v, handle, "vec_download");
  GM_CHECK(offset + n <= v->cap, GM_EINVAL, "vec_download: range [%zu, %zu) outside capacity %zu", offset, offset + n, v->cap);
  if (n) {
    GM_HIP(hipMemcpyAsync(mont, v->d + offset * 32, n * 32, hipMemcpyDeviceToHost, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
  }
  return GM_OK;
}
int gm_fr_vec_fill(uint64_t handle, const uint64_t value_mont[4]) {
  GM_CTX();
  GM_VEC(v, handle, "vec_fill");
  return fr_fill(C, v, value_mont);
}

int gm_fr_reverse(uint64_t in, uint64_t out) {
  GM_CTX();
  GM_VEC(vi, in, "fr_reverse");
  GM_VEC(vo, out, "fr_reverse");
  return fr_reverse(C, vi, vo);
}
int gm_fr_stride(uint64_t in, size_t start, size_t stride, size_t count, uint64_t out) {
  GM_CTX();
  GM_VEC(vi, in, "fr_stride");
  GM_VEC(vo, out, "fr_stride");
  GM_CHECK(vi != vo, GM_EINVAL, "fr_stride: output must not alias the input");
  // overflow-safe form of start + (count - 1) * stride < len
  GM_CHECK(stride >= 1 && (count == 0 || (start < vi->len && count - 1 <= (vi->len - 1 - start) / stride)), GM_EINVAL,
           "fr_stride: elements %zu + k * %zu, k < %zu, outside a vector of length %zu", start, stride, count, vi->len);
  GM_CHECK(vo->cap >= count, GM_EINVAL, "fr_stride: output capacity %zu < %zu", vo->cap, count);
  int rc = fr_stride_raw(C, vi->d, start, stride, count, vo->d);
  if (rc) return rc;
  vo->len = count;
  // entry points return with their work done: the input may be freed (and its pool block reused on another stream) right away
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}
int gm_fr_fold_chain(uint64_t f, const uint64_t* challenges_mont, size_t k, const uint64_t* outs) {
  GM_CTX();
  GM_VEC(vf, f, "fr_fold_chain");
  GM_CHECK(k == 0 || (challenges_mont && outs), GM_EINVAL, "fr_fold_chain: null pointer");
  std::vector<FrVec*> vo(k);
  for (size_t j = 0; j < k; j++) {
    vo[j] = find_vec(outs[j]);
    GM_CHECK(vo[j] != nullptr, GM_EHANDLE, "fr_fold_chain: unknown vector handle %llu", (unsigned long long)outs[j]);
    for (size_t i = 0; i < j; i++) GM_CHECK(vo[i] != vo[j], GM_EINVAL, "fr_fold_chain: repeated output vector");
  }
  return fr_fold_chain(C, vf, challenges_mont, k, vo.data());
}
int gm_fr_fold(uint64_t f, const uint64_t r_mont[4], uint64_t out) {
  GM_CTX();
  GM_VEC(vf, f, "fr_fold");
  GM_VEC(vo, out, "fr_fold");
  return fr_fold(C, vf, r_mont, vo);
}
int gm_fr_powers(const uint64_t x_mont[4], size_t n, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_powers");
  return fr_powers(C, x_mont, n, vo);
}
int gm_fr_tensor(const uint64_t* rhos_mont, size_t k, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_tensor");
  return fr_tensor(C, rhos_mont, k, vo);
}
int gm_fr_hadamard(uint64_t a, uint64_t b, uint64_t out) {
  GM_CTX();
  GM_VEC(va, a, "fr_hadamard");
  GM_VEC(vb, b, "fr_hadamard");
  GM_VEC(vo, out, "fr_hadamard");
  return fr_hadamard(C, va, vb, vo);
}
int gm_fr_ip(uint64_t a, uint64_t b, uint64_t result_mont[4]) {
  GM_CTX();
  GM_VEC(va, a, "fr_ip");
  GM_VEC(vb, b, "fr_ip");
  return fr_ip(C, va, vb, result_mont);
}
int gm_fr_eval_le(uint64_t poly, const uint64_t* xs_mont, size_t npoints, uint64_t* results_mont) {
  GM_CTX();
  GM_VEC(vp, poly, "fr_eval_le");
  return fr_eval_le(C, vp, xs_mont, npoints, results_mont);
}
int gm_fr_eval_le_batch(const uint64_t* polys, size_t k, const uint64_t* xs_mont, size_t npoints, uint64_t* results_mont) {
  GM_CTX();
  GM_CHECK((polys && results_mont && xs_mont) || k == 0, GM_EINVAL, "fr_eval_le_batch: null pointer");
  std::vector<FrVec*> ps(k);
  for (size_t j = 0; j < k; j++) {
    ps[j] = find_vec(polys[j]);
    GM_CHECK(ps[j] != nullptr, GM_EHANDLE, "fr_eval_le_batch: unknown vector handle %llu", (unsigned long long)polys[j]);
  }
  return fr_eval_le_batch(C, ps.data(), k, xs_mont, npoints, results_mont);
}
int gm_fr_scale_into(uint64_t in, const uint64_t c_mont[4], uint64_t out, size_t out_offset) {
  GM_CTX();
  GM_VEC(vi, in, "fr_scale_into");
  GM_VEC(vo, out, "fr_scale_into");
  GM_CHECK(c_mont != nullptr, GM_EINVAL, "fr_scale_into: null pointer");
  return fr_scale_into(C, vi, c_mont, vo, out_offset);
}
int gm_fr_scale_into_many(const uint64_t* ins, const uint64_t* coeffs_mont, size_t k, uint64_t out, const size_t* out_offsets) {
  GM_CTX();
  GM_VEC(vo, out, "fr_scale_into_many");
  GM_CHECK(k == 0 || (ins && coeffs_mont && out_offsets), GM_EINVAL, "fr_scale_into_many: null pointer");
  std::vector<FrVec*> ps(k);
  for (size_t j = 0; j < k; j++) {
    ps[j] = find_vec(ins[j]);
    GM_CHECK(ps[j] != nullptr, GM_EHANDLE, "fr_scale_into_many: unknown vector handle %llu", (unsigned long long)ins[j]);
  }
  return fr_scale_into_many(C, ps.data(), coeffs_mont, k, vo, out_offsets);
}
int gm_fr_add_at(uint64_t v, const size_t* positions, const uint64_t* values_mont, size_t k) {
  GM_CTX();
  GM_VEC(vv, v, "fr_add_at");
  GM_CHECK(k == 0 || (positions && values_mont), GM_EINVAL, "fr_add_at: null pointer");
  return fr_add_at(C, vv, positions, values_mont, k);
}
int gm_fr_lincomb(const uint64_t* polys, const uint64_t* coeffs_mont, size_t k, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_lincomb");
  std::vector<FrVec*> ps(k);
  for (size_t j = 0; j < k; j++) {
    ps[j] = find_vec(polys[j]);
    GM_CHECK(ps[j] != nullptr, GM_EHANDLE, "fr_lincomb: unknown vector handle %llu", (unsigned long long)polys[j]);
  }
  return fr_lincomb(C, ps.data(), coeffs_mont, k, vo);
}
int gm_fr_div_vanishing(uint64_t f, const uint64_t* points_mont, size_t k, uint64_t quotient, uint64_t* rem_mont) {
  GM_CTX();
  GM_VEC(vf, f, "fr_div_vanishing");
  GM_VEC(vq, quotient, "fr_div_vanishing");
  return fr_div_linear_factors(C, vf, points_mont, k, vq, rem_mont);
}

// ---- index vectors + entry-product / plookup builders (psnark) ----------------------------------
int gm_idx_register(const uint32_t* index, size_t n, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(handle && (index || n == 0), GM_EINVAL, "idx_register: null pointer");
  auto I = std::make_unique<IdxVec>();
  I->n = n;
  uint32_t mx = 0;
  for (size_t k = 0; k < n; k++) mx = index[k] > mx ? index[k] : mx;
  I->max_plus_1 = n ? (size_t)mx + 1 : 0;
  if (n) {
    GM_HIP(dev_malloc((void**)&I->d, n * 4));
    GM_HIP(hipMemcpyAsync(I->d, index, n * 4, hipMemcpyHostToDevice, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
  }
  std::lock_guard<std::mutex> lk(C->mu);
  *handle = C->next_handle++;
  C->indices[*handle] = std::move(I);
  return GM_OK;
}
int gm_idx_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<IdxVec> I;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->indices.find(handle);
    GM_CHECK(it != C->indices.end(), GM_EHANDLE, "idx_free: unknown handle %llu", (unsigned long long)handle);
    I = std::move(it->second);
    C->indices.erase(it);
  }
  if (I->d) (void)gm::raw_free(I->d);
  return GM_OK;
}
#define GM_IDX(var, h, who)                                                                             \
  IdxVec* var;                                                                                          \
  {                                                                                                     \
    std::lock_guard<std::mutex> lk(C->mu);                                                              \
    auto it = C->indices.find(h);                                                                       \
    GM_CHECK(it != C->indices.end(), GM_EHANDLE, who ": unknown index handle %llu", (unsigned long long)(h)); \
    var = it->second.get();                                                                             \
  }
int gm_fr_gather(uint64_t src, uint64_t index, uint64_t out) {
  GM_CTX();
  GM_VEC(vs, src, "fr_gather");
  GM_VEC(vo, out, "fr_gather");
  GM_IDX(ix, index, "fr_gather");
  return fr_gather(C, vs, ix, vo);
}
int gm_fr_alg_hash(uint64_t v, uint64_t index, const uint64_t zeta_mont[4], uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_alg_hash");
  GM_VEC(vo, out, "fr_alg_hash");
  if (index == 0) return fr_alg_hash(C, vv, nullptr, zeta_mont, vo);
  GM_IDX(ix, index, "fr_alg_hash");
  return fr_alg_hash(C, vv, ix, zeta_mont, vo);
}
int gm_fr_plookup_set(uint64_t v, const uint64_t y_mont[4], const uint64_t z_mont[4], uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_plookup_set");
  GM_VEC(vo, out, "fr_plookup_set");
  return fr_plookup_set(C, vv, y_mont, z_mont, vo);
}
int gm_fr_add_scalar(uint64_t v, const uint64_t y_mont[4], uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_add_scalar");
  GM_VEC(vo, out, "fr_add_scalar");
  return fr_add_scalar(C, vv, y_mont, vo);
}
int gm_fr_shift_monic(uint64_t v, uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_shift_monic");
  GM_VEC(vo, out, "fr_shift_monic");
  return fr_shift_monic(C, vv, vo);
}
int gm_fr_acc_product(uint64_t v, uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_acc_product");
  GM_VEC(vo, out, "fr_acc_product");
  return fr_acc_product(C, vv, vo);
}

// ---- the builders on one block of a block-sharded vector (gm_psnark_new_time_sharded) ----------
int gm_fr_tensor_range(const uint64_t* rhos_mont, size_t k, size_t start, size_t count, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_tensor_range");
  GM_CHECK(rhos_mont != nullptr, GM_EINVAL, "fr_tensor_range: null pointer");
  return fr_tensor_range(C, rhos_mont, k, start, count, vo);
}
int gm_fr_powers_range(const uint64_t x_mont[4], size_t start, size_t count, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_powers_range");
  GM_CHECK(x_mont != nullptr && vo->cap >= count, GM_EINVAL, "fr_powers_range: output capacity %zu < %zu", vo->cap, count);
  int rc = fr_powers_at(C, x_mont, start, count, vo->d);
  if (rc) return rc;
  vo->len = count;
  return GM_OK;
}
// lookups of vectors that are FUNCTIONS of the index, without the vectors (k_gather_prod2)
int gm_fr_tensor_gather(const uint64_t* rhos_mont, size_t k, uint64_t index, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_tensor_gather");
  GM_IDX(ix, index, "fr_tensor_gather");
  GM_CHECK(rhos_mont != nullptr, GM_EINVAL, "fr_tensor_gather: null pointer");
  return fr_tensor_gather(C, rhos_mont, k, ix, vo);
}
int gm_fr_powers_gather(const uint64_t x_mont[4], size_t log_len, uint64_t index, uint64_t out) {
  GM_CTX();
  GM_VEC(vo, out, "fr_powers_gather");
  GM_IDX(ix, index, "fr_powers_gather");
  GM_CHECK(x_mont != nullptr, GM_EINVAL, "fr_powers_gather: null pointer");
  return fr_powers_gather(C, x_mont, log_len, ix, vo);
}
// alg_hash of a RANGE of a vector: out[i] = v[i] + (first_index + i) zeta
int gm_fr_alg_hash_from(uint64_t v, size_t first_index, const uint64_t zeta_mont[4], uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_alg_hash_from");
  GM_VEC(vo, out, "fr_alg_hash_from");
  return fr_alg_hash(C, vv, nullptr, zeta_mont, vo, (uint64_t)first_index);
}
int gm_fr_plookup_set_block(uint64_t v, size_t v_offset, size_t v_count, const uint64_t* prev_or_null, size_t out_count, const uint64_t y_mont[4],
                            const uint64_t z_mont[4], uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_plookup_set_block");
  GM_VEC(vo, out, "fr_plookup_set_block");
  GM_CHECK(y_mont && z_mont, GM_EINVAL, "fr_plookup_set_block: null pointer");
  return fr_plookup_set_block(C, vv, v_offset, v_count, prev_or_null, out_count, y_mont, z_mont, vo);
}
int gm_fr_shift_block(uint64_t v, const uint64_t first_mont[4], size_t out_count, uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_shift_block");
  GM_VEC(vo, out, "fr_shift_block");
  GM_CHECK(first_mont != nullptr, GM_EINVAL, "fr_shift_block: null pointer");
  return fr_shift_block(C, vv, first_mont, out_count, vo);
}
int gm_fr_product(uint64_t v, uint64_t total_mont[4]) {
  GM_CTX();
  GM_VEC(vv, v, "fr_product");
  GM_CHECK(total_mont != nullptr, GM_EINVAL, "fr_product: null pointer");
  return fr_acc_product_ex(C, vv, nullptr, false, nullptr, total_mont);
}
int gm_fr_acc_product_block(uint64_t v, const uint64_t* carry_or_null, int write_monic, uint64_t out) {
  GM_CTX();
  GM_VEC(vv, v, "fr_acc_product_block");
  GM_VEC(vo, out, "fr_acc_product_block");
  return fr_acc_product_ex(C, vv, carry_or_null, write_monic != 0, vo, nullptr);
}

// ---- sparse matrices (R1CS) -------------------------------------------------------------------
int gm_spm_register(const uint64_t* rowptr, const uint32_t* cols, const uint64_t* vals_mont, size_t nrows, size_t ncols,
                    size_t nnz, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(rowptr && handle && ((cols && vals_mont) || nnz == 0), GM_EINVAL, "spm_register: null pointer");
  GM_CHECK(rowptr[0] == 0 && rowptr[nrows] == nnz, GM_EINVAL, "spm_register: rowptr must run from 0 to nnz");
  for (size_t k = 0; k < nnz; k++) GM_CHECK(cols[k] < ncols, GM_EINVAL, "spm_register: column %u >= %zu", cols[k], ncols);
  auto M = std::make_unique<SparseMatrix>();
  M->nrows = nrows;
  M->ncols = ncols;
  M->nnz = nnz;
  GM_HIP(dev_malloc((void**)&M->rowptr, (nrows + 1) * 8));
  GM_HIP(hipMemcpyAsync(M->rowptr, rowptr, (nrows + 1) * 8, hipMemcpyHostToDevice, C->stream));
  if (nnz) {
    GM_HIP(dev_malloc((void**)&M->cols, nnz * 4));
    GM_HIP(dev_malloc((void**)&M->vals, nnz * 32));
    GM_HIP(hipMemcpyAsync(M->cols, cols, nnz * 4, hipMemcpyHostToDevice, C->stream));
    GM_HIP(hipMemcpyAsync(M->vals, vals_mont, nnz * 32, hipMemcpyHostToDevice, C->stream));
  }
  GM_HIP(hipStreamSynchronize(C->stream));
  std::lock_guard<std::mutex> lk(C->mu);
  *handle = C->next_handle++;
  C->matrices[*handle] = std::move(M);
  return GM_OK;
}
int gm_spm_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<SparseMatrix> M;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->matrices.find(handle);
    GM_CHECK(it != C->matrices.end(), GM_EHANDLE, "spm_free: unknown handle %llu", (unsigned long long)handle);
    M = std::move(it->second);
    C->matrices.erase(it);
  }
  if (M->rowptr) (void)gm::raw_free(M->rowptr);
  if (M->cols) (void)gm::raw_free(M->cols);
  if (M->vals) (void)gm::raw_free(M->vals);
  return GM_OK;
}
int gm_spm_shape(uint64_t handle, size_t* nrows, size_t* ncols, size_t* nnz) {
  GM_CTX();
  SparseMatrix* M = nullptr;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->matrices.find(handle);
    if (it != C->matrices.end()) M = it->second.get();
  }
  GM_CHECK(M != nullptr, GM_EHANDLE, "spm_shape: unknown matrix handle %llu", (unsigned long long)handle);
  if (nrows) *nrows = M->nrows;
  if (ncols) *ncols = M->ncols;
  if (nnz) *nnz = M->nnz;
  return GM_OK;
}
// the CSR arrays back on the host (any pointer may be null): what the preprocessing of the indexed SNARK reads
// (gm_psnark_preprocess: joint support of A, B, C)
int gm_spm_download(uint64_t handle, uint64_t* rowptr, uint32_t* cols, uint64_t* vals_mont) {
  GM_CTX();
  SparseMatrix* M = nullptr;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->matrices.find(handle);
    if (it != C->matrices.end()) M = it->second.get();
  }
  GM_CHECK(M != nullptr, GM_EHANDLE, "spm_download: unknown matrix handle %llu", (unsigned long long)handle);
  GM_FR_LOCK(C);
  if (rowptr) GM_HIP(hipMemcpyAsync(rowptr, M->rowptr, (M->nrows + 1) * 8, hipMemcpyDeviceToHost, C->stream));
  if (cols && M->nnz) GM_HIP(hipMemcpyAsync(cols, M->cols, M->nnz * 4, hipMemcpyDeviceToHost, C->stream));
  if (vals_mont && M->nnz) GM_HIP(hipMemcpyAsync(vals_mont, M->vals, M->nnz * 32, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}
int gm_spm_mul(uint64_t matrix, uint64_t x, uint64_t y) {
  GM_CTX();
  SparseMatrix* M;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->matrices.find(matrix);
    GM_CHECK(it != C->matrices.end(), GM_EHANDLE, "spm_mul: unknown matrix handle %llu", (unsigned long long)matrix);
    M = it->second.get();
  }
  GM_VEC(vx, x, "spm_mul");
  GM_VEC(vy, y, "spm_mul");
  return spm_mul(C, M, vx, vy);
}

// ---- sumcheck ---------------------------------------------------------------------------------
#define GM_SC(var, h, who)                     \
  Sumcheck* var = find_prover(h);              \
  GM_CHECK(var != nullptr, GM_EHANDLE, who ": unknown prover handle %llu", (unsigned long long)(h))

int gm_sc_new(const uint64_t* f_mont, size_t nf, const uint64_t* g_mont, size_t ng, const uint64_t twist_mont[4],
              uint64_t* handle) {
  GM_CTX();
  GM_CHECK(f_mont && g_mont && twist_mont && handle, GM_EINVAL, "sc_new: null pointer");
  return sc_create(C, f_mont, nf, g_mont, ng, false, twist_mont, handle);
}
int gm_sc_new_v(uint64_t f_vec, uint64_t g_vec, const uint64_t twist_mont[4], uint64_t* handle) {
  GM_CTX();
  GM_VEC(vf, f_vec, "sc_new_v");
  GM_VEC(vg, g_vec, "sc_new_v");
  GM_CHECK(twist_mont && handle, GM_EINVAL, "sc_new_v: null pointer");
  return sc_create(C, vf->d, vf->len, vg->d, vg->len, true, twist_mont, handle);
}
int gm_sc_new_borrow(uint64_t f_vec, uint64_t g_vec, const uint64_t twist_mont[4], uint64_t* handle) {
  GM_CTX();
  GM_VEC(vf, f_vec, "sc_new_borrow");
  GM_VEC(vg, g_vec, "sc_new_borrow");
  GM_CHECK(twist_mont && handle, GM_EINVAL, "sc_new_borrow: null pointer");
  return sc_create(C, vf->d, vf->len, vg->d, vg->len, true, twist_mont, handle, true);
}
int gm_sc_round(uint64_t handle, const uint64_t* challenge_or_null, uint64_t a_mont[4], uint64_t b_mont[4], int* has_msg) {
  GM_CTX();
  GM_SC(S, handle, "sc_round");
  GM_CHECK(a_mont && b_mont && has_msg, GM_EINVAL, "sc_round: null pointer");
  return sc_round(C, S, challenge_or_null, a_mont, b_mont, has_msg);
}
int gm_sc_round_begin(uint64_t handle, const uint64_t* challenge_or_null, int* has_msg) {
  GM_CTX();
  Sumcheck* S = find_prover(handle);
  GM_CHECK(S != nullptr, GM_EHANDLE, "sc_round_begin: unknown prover handle %llu", (unsigned long long)handle);
  GM_CHECK(has_msg != nullptr, GM_EINVAL, "sc_round_begin: null pointer");
  return sc_round_begin(C, S, challenge_or_null, has_msg);
}
int gm_sc_round_begin_many(const uint64_t* handles, size_t k, const uint64_t* challenge_or_null, int* has_msg) {
  GM_CTX();
  GM_CHECK((handles && has_msg) || k == 0, GM_EINVAL, "sc_round_begin_many: null pointer");
  std::vector<Sumcheck*> S(k);
  for (size_t j = 0; j < k; j++) {
    S[j] = find_prover(handles[j]);
    GM_CHECK(S[j] != nullptr, GM_EHANDLE, "sc_round_begin_many: unknown prover handle %llu", (unsigned long long)handles[j]);
  }
  return k ? sc_round_begin_many(C, S.data(), k, challenge_or_null, has_msg) : GM_OK;
}
int gm_sc_round_end(uint64_t handle, uint64_t a_mont[4], uint64_t b_mont[4]) {
  GM_CTX();
  Sumcheck* S = find_prover(handle);
  GM_CHECK(S != nullptr, GM_EHANDLE, "sc_round_end: unknown prover handle %llu", (unsigned long long)handle);
  GM_CHECK(a_mont && b_mont, GM_EINVAL, "sc_round_end: null pointer");
  return sc_round_end(C, S, a_mont, b_mont);
}
int gm_sc_fold(uint64_t handle, const uint64_t challenge_mont[4]) {
  GM_CTX();
  GM_SC(S, handle, "sc_fold");
  return sc_fold(C, S, challenge_mont);
}
int gm_sc_rounds(uint64_t handle, size_t* tot_rounds, size_t* round) {
  GM_CTX();
  GM_SC(S, handle, "sc_rounds");
  if (tot_rounds) *tot_rounds = S->tot_rounds;
  if (round) *round = S->round;
  return GM_OK;
}
int gm_sc_final(uint64_t handle, uint64_t f0_mont[4], uint64_t g0_mont[4], int* has) {
  GM_CTX();
  GM_SC(S, handle, "sc_final");
  return sc_final(C, S, f0_mont, g0_mont, has);
}
// current vectors of the prover (after the folds so far): lengths, then the elements
int gm_sc_lens(uint64_t handle, size_t* nf, size_t* ng, uint64_t twist_mont[4]) {
  GM_CTX();
  GM_SC(S, handle, "sc_lens");
  std::lock_guard<std::mutex> lk(S->mu);
  if (nf) *nf = S->nf;
  if (ng) *ng = S->ng;
  if (twist_mont) memcpy(twist_mont, S->twist, 32);
  return GM_OK;
}
int gm_sc_download(uint64_t handle, uint64_t* f_mont, uint64_t* g_mont) {
  GM_CTX();
  GM_SC(S, handle, "sc_download");
  std::lock_guard<std::mutex> lk(S->mu);
  if (S->on_host) {  // the tail lives on the host (fr.hip: sc_host_step)
    if (f_mont && S->nf) memcpy(f_mont, S->hf.data(), S->nf * 32);
    if (g_mont && S->ng) memcpy(g_mont, S->hg.data(), S->ng * 32);
    return GM_OK;
  }
  if (f_mont && S->nf) GM_HIP(hipMemcpyAsync(f_mont, S->f[S->cur], S->nf * 32, hipMemcpyDeviceToHost, C->stream));
  if (g_mont && S->ng) GM_HIP(hipMemcpyAsync(g_mont, S->g[S->cur], S->ng * 32, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}
int gm_sc_set_shard(uint64_t handle, uint64_t pair_offset) {
  GM_CTX();
  GM_SC(S, handle, "sc_set_shard");
  S->pair_offset = pair_offset;
  return GM_OK;
}
// the same for a block whose length says nothing about the rounds of the WHOLE vectors (a short or partial block of a
// block-sharded prover): the round count is the caller's
int gm_sc_set_shard_rounds(uint64_t handle, uint64_t pair_offset, size_t tot_rounds) {
  GM_CTX();
  GM_SC(S, handle, "sc_set_shard_rounds");
  std::lock_guard<std::mutex> lk(S->mu);
  GM_CHECK(S->round == 0 && tot_rounds < 64, GM_ESTATE, "sc_set_shard_rounds: before the first round only (%zu rounds)", tot_rounds);
  S->pair_offset = pair_offset;
  S->tot_rounds = tot_rounds;
  return GM_OK;
}
// ---- herring provers -------------------------------------------------------------------------------
int gm_sc_set_herring(uint64_t handle, int on) {
  GM_CTX();
  GM_SC(S, handle, "sc_set_herring");
  return sc_set_herring(S, on);
}
static HerringG1* find_hg1(Context* C, uint64_t h) {
  std::lock_guard<std::mutex> lk(C->mu);
  auto it = C->herring_g1.find(h);
  return it == C->herring_g1.end() ? nullptr : it->second.get();
}
#define GM_HG1(var, h, who)                   \
  HerringG1* var = find_hg1(C, h);            \
  GM_CHECK(var != nullptr, GM_EHANDLE, who ": unknown herring G1 prover handle %llu", (unsigned long long)(h))
int gm_hg1_new(const void* f_bases, size_t base_stride, size_t nf, const uint64_t* g_mont, size_t ng, const uint64_t twist_mont[4],
               uint64_t* handle) {
  GM_CTX();
  GM_CHECK(f_bases && g_mont && twist_mont && handle, GM_EINVAL, "hg1_new: null pointer");
  return hg1_create(C, f_bases, base_stride, nf, g_mont, ng, twist_mont, handle);
}
int gm_hg1_round(uint64_t handle, const uint64_t* challenge_or_null, uint64_t a_jac[18], uint64_t b_jac[18], int* has_msg) {
  GM_CTX();
  GM_HG1(H, handle, "hg1_round");
  GM_CHECK(a_jac && b_jac && has_msg, GM_EINVAL, "hg1_round: null pointer");
  return hg1_round(C, H, challenge_or_null, a_jac, b_jac, has_msg);
}
int gm_hg1_fold(uint64_t handle, const uint64_t challenge_mont[4]) {
  GM_CTX();
  GM_HG1(H, handle, "hg1_fold");
  return hg1_fold(C, H, challenge_mont);
}
int gm_hg1_rounds(uint64_t handle, size_t* tot_rounds, size_t* round) {
  GM_CTX();
  GM_HG1(H, handle, "hg1_rounds");
  if (tot_rounds) *tot_rounds = H->tot_rounds;
  if (round) *round = H->round;
  return GM_OK;
}
int gm_hg1_final(uint64_t handle, uint64_t f0_jac[18], uint64_t g0_mont[4], int* has) {
  GM_CTX();
  GM_HG1(H, handle, "hg1_final");
  return hg1_final(C, H, f0_jac, g0_mont, has);
}
int gm_hg1_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<HerringG1> p;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->herring_g1.find(handle);
    GM_CHECK(it != C->herring_g1.end(), GM_EHANDLE, "hg1_free: unknown handle %llu", (unsigned long long)handle);
    p = std::move(it->second);
    C->herring_g1.erase(it);
  }
  hg1_destroy(C, p.get());
  return GM_OK;
}

// ---- space prover --------------------------------------------------------------------------------
static SpaceProver* find_sp(Context* C, uint64_t h) {
  std::lock_guard<std::mutex> lk(C->mu);
  auto it = C->space_provers.find(h);
  return it == C->space_provers.end() ? nullptr : it->second.get();
}
#define GM_SP(var, h, who)                     \
  SpaceProver* var = find_sp(C, h);            \
  GM_CHECK(var != nullptr, GM_EHANDLE, who ": unknown space prover handle %llu", (unsigned long long)(h))

int gm_sp_new(const uint64_t* f_stream_mont, size_t nf, const uint64_t* g_stream_mont, size_t ng, const uint64_t twist_mont[4],
              uint64_t* handle) {
  GM_CTX();
  GM_CHECK(f_stream_mont && g_stream_mont && twist_mont && handle, GM_EINVAL, "sp_new: null pointer");
  return sp_create(C, f_stream_mont, nf, g_stream_mont, ng, false, twist_mont, handle);
}
int gm_sp_new_v(uint64_t f_stream_vec, uint64_t g_stream_vec, const uint64_t twist_mont[4], uint64_t* handle) {
  GM_CTX();
  GM_CHECK(twist_mont && handle, GM_EINVAL, "sp_new_v: null pointer");
  GM_VEC(vf, f_stream_vec, "sp_new_v");
  GM_VEC(vg, g_stream_vec, "sp_new_v");
  return sp_create(C, vf->d, vf->len, vg->d, vg->len, true, twist_mont, handle);
}
int gm_sp_new_borrow(uint64_t f_stream_vec, uint64_t g_stream_vec, const uint64_t twist_mont[4], uint64_t* handle) {
  GM_CTX();
  GM_CHECK(twist_mont && handle, GM_EINVAL, "sp_new_borrow: null pointer");
  GM_VEC(vf, f_stream_vec, "sp_new_borrow");
  GM_VEC(vg, g_stream_vec, "sp_new_borrow");
  return sp_create(C, vf->d, vf->len, vg->d, vg->len, true, twist_mont, handle, true);
}
int gm_sp_round(uint64_t handle, const uint64_t* challenge_or_null, uint64_t a_mont[4], uint64_t b_mont[4], int* has_msg) {
  GM_CTX();
  GM_SP(S, handle, "sp_round");
  GM_CHECK(a_mont && b_mont && has_msg, GM_EINVAL, "sp_round: null pointer");
  return sp_round(C, S, challenge_or_null, a_mont, b_mont, has_msg);
}
int gm_sp_fold(uint64_t handle, const uint64_t challenge_mont[4]) {
  GM_CTX();
  GM_SP(S, handle, "sp_fold");
  return sp_fold(C, S, challenge_mont);
}
int gm_sp_rounds(uint64_t handle, size_t* tot_rounds, size_t* round) {
  GM_CTX();
  GM_SP(S, handle, "sp_rounds");
  if (tot_rounds) *tot_rounds = S->tot_rounds;
  if (round) *round = S->round;
  return GM_OK;
}
int gm_sp_final(uint64_t handle, uint64_t f0_mont[4], uint64_t g0_mont[4], int* has) {
  GM_CTX();
  GM_SP(S, handle, "sp_final");
  return sp_final(C, S, f0_mont, g0_mont, has);
}
int gm_sp_to_time(uint64_t handle, uint64_t* time_handle) {
  GM_CTX();
  GM_SP(S, handle, "sp_to_time");
  GM_CHECK(time_handle != nullptr, GM_EINVAL, "sp_to_time: null pointer");
  return sp_to_time(C, S, time_handle);
}
int gm_sp_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<SpaceProver> p;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->space_provers.find(handle);
    GM_CHECK(it != C->space_provers.end(), GM_EHANDLE, "sp_free: unknown handle %llu", (unsigned long long)handle);
    p = std::move(it->second);
    C->space_provers.erase(it);
  }
  sp_destroy(C, p.get());
  return GM_OK;
}

int gm_sc_free(uint64_t handle) {
  GM_CTX();
  std::unique_ptr<Sumcheck> p;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->provers.find(handle);
    GM_CHECK(it != C->provers.end(), GM_EHANDLE, "sc_free: unknown handle %llu", (unsigned long long)handle);
    p = std::move(it->second);
    C->provers.erase(it);
  }
  sc_destroy(p.get());
  return GM_OK;
}

}  // extern "C"
