// Process-wide state of libgemini_hip.so: one GPU per process, opaque u64 handles for
// device-resident bases / Fr vectors / sumcheck provers, grow-only workspaces, error strings.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gemini_hip.h"

namespace gm {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define GM_HIP(expr)                                                     \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) return ::gm::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define GM_CHECK(cond, code, ...)  \
  do {                             \
    if (!(cond)) {                 \
      ::gm::set_error(__VA_ARGS__); \
      return (code);               \
    }                              \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes);
  void release();
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct Bases {
  uint8_t* d = nullptr;  // n x 96 bytes: x, y Montgomery; identity = all zero
  // phi(P_i) = (beta x_i, y_i) = lambda P_i, the image under the curve's endomorphism, same layout; built at
  // registration (bases_build_phi) for the GLV split of the MSM, nullptr for one-shot / borrowed bases
  uint8_t* phi = nullptr;
  size_t n = 0;
  // optional fixed-base window tables: table[w * n + i] = 2^(tab_c * w) * base[i], w < tab_W
  // (table row 0 is a copy of d so one pointer serves every window)
  uint8_t* table = nullptr;
  int tab_c = 0, tab_W = 0;
  size_t tab_min = 0;  // smallest call the tables pay off for (depends on their window width)
  // PREFIX tables for the calls the main tables do not serve: a wide-window table (c = 22: 2^21 buckets to reduce) loses below
  // 2^22 pairs, but the short calls of a prover -- the low levels of a folding tree -- walk the FIRST powers of the key, so a second
  // table over that prefix with a narrower window costs a few GB (13 x 2^22 x 96 B).  A call [first, first + step * n) inside
  // the prefix with min_n <= n < max_n takes the set.
  struct TableSet {
    uint8_t* t = nullptr;
    int c = 0, W = 0;
    size_t n = 0, min_n = 0, max_n = 0;
  };
  std::vector<TableSet> extra;
  bool extras_released = false;  // the prefix sets were given back under memory pressure (release_spare_tables); gm_g1_bases_precompute(handle, -1) rebuilds them
  // a CYCLIC SHARE of a committer key (gm_g1_bases_set_cyclic): these n points are the powers i = cyc_rank (mod cyc_world)
  // of a key of cyclic_n powers; gm_ck_* and the provers commit through the all-gather of dist.cpp.  0 = a whole key
  size_t cyclic_n = 0;
  int cyc_rank = 0, cyc_world = 1;
};

struct FrVec {
  uint8_t* d = nullptr;  // cap x 32 bytes, Montgomery
  size_t cap = 0;
  size_t len = 0;
  size_t cap_bytes = 0;  // size of the pooled allocation behind d
};

// CSR sparse matrix resident in HBM (`Matrix<F> = Vec<Vec<(F, usize)>>`, src/circuit.rs:43)
struct SparseMatrix {
  uint64_t* rowptr = nullptr;  // nrows + 1
  uint32_t* cols = nullptr;    // nnz
  uint8_t* vals = nullptr;     // nnz x 32 bytes Montgomery
  size_t nrows = 0, ncols = 0, nnz = 0;
};

// index vector resident in HBM (`&[usize]` arguments of plookup / lookup: row_index, col_index, ...)
struct IdxVec {
  uint32_t* d = nullptr;
  size_t n = 0;
  size_t max_plus_1 = 0;  // 1 + the largest index (0 for an empty vector)
};

struct Sumcheck {
  // ping-pong state of TimeProver (src/subprotocols/sumcheck/time_prover.rs:42-52)
  uint8_t* f[2] = {nullptr, nullptr};
  uint8_t* g[2] = {nullptr, nullptr};
  size_t fcap[2] = {0, 0}, gcap[2] = {0, 0};  // pooled allocation sizes
  // gm_sc_new_borrow: f[0] / g[0] are the CALLER's vectors, read until the first fold has written the halves into f[1] / g[1]; the
  // slot gets a quarter-size buffer of the prover's own when the second fold needs somewhere to write
  bool borrowed = false;
  int cur = 0;
  size_t nf = 0, ng = 0;
  uint64_t twist[4];  // Montgomery
  size_t round = 0, tot_rounds = 0;
  uint64_t pair_offset = 0;  // shard origin (gm_sc_set_shard)
  bool herring = false;      // herring FModule prover: messages carry no twist (src/herring/time_prover.rs:91-123)
  uint8_t* partials = nullptr;   // per-block (a, b) partial sums
  uint64_t* host_partials = nullptr;  // pinned
  unsigned pending_blocks = 0;        // blocks of the round whose partial sums are on their way to host_partials
  // the TAIL on the host: once the vectors are down to SC_HOST_TAIL elements they are copied out once and the remaining rounds --
  // a launch, a copy and a wait each on the device, ~45 us for microseconds of arithmetic -- run on the host (fr.hip: sc_host_step)
  bool on_host = false;
  std::vector<uint64_t> hf, hg;  // 4 limbs per element, Montgomery
  uint64_t host_msg[8];          // the message of the round in flight (split-phase rounds)
  bool host_msg_pending = false;
  std::mutex mu;
};
constexpr size_t SC_HOST_TAIL = 256;

struct SpaceProver {
  // src/subprotocols/sumcheck/space_prover.rs:20-40: the witness streams (big-endian, never modified),
  // the challenges and the twisted challenges
  uint8_t* f = nullptr;
  uint8_t* g = nullptr;
  size_t nf = 0, ng = 0, fcap = 0, gcap = 0;
  bool borrowed = false;  // gm_sp_new_borrow: f / g are the caller's stream vectors (a space prover never writes them)
  std::vector<uint64_t> challenges, twisted;  // 4 limbs each, Montgomery
  uint64_t twist[4];
  size_t round = 0, tot_rounds = 0;
  uint8_t* tables = nullptr;  // weight tables + reduced ping / pong vectors of the current message: a block of the vector pool
  size_t tables_cap = 0;       // (hipMalloc / hipFree per prover cost a device-wide synchronisation each)
  uint8_t *wf_lo = nullptr, *wf_hi = nullptr, *wg_lo = nullptr, *wg_hi = nullptr;
  uint8_t* partials = nullptr;
  uint64_t* host_partials = nullptr;
  std::mutex mu;
};

// herring TimeProver over G1Module (Lhs = G1, Rhs = F, Target = G1): src/herring/module.rs:81-102
struct HerringG1 {
  uint8_t* f[2] = {nullptr, nullptr};  // affine points, 96 B each
  uint8_t* g[2] = {nullptr, nullptr};  // Fr
  size_t fcap[2] = {0, 0}, gcap[2] = {0, 0};
  int cur = 0;
  size_t nf = 0, ng = 0;
  uint64_t twist[4];
  size_t round = 0, tot_rounds = 0;
  uint8_t* tmp = nullptr;  // compacted scalars
  size_t tmpcap = 0;
  std::mutex mu;
};

// ChunkedPippenger / msm_chunks over HOST-resident pairs (src/kzg/msm/stream_pippenger.rs:209-272, src/kzg/space.rs:22-55):
// the device holds two chunks; chunk i + 1 is copied in while the MSM of chunk i runs (msm.hip: msm_stream_*)
struct MsmWorkspace;
// One MSM between its enqueue (all kernels + the async copy of the window bit-planes) and its finish (host Horner)
struct MsmPending {
  MsmWorkspace* ws = nullptr;  // the workspace (and stream) the call was enqueued on
  int slot = 0;
  bool empty = true;
  int Wb = 0, c = 0, m = 0;
  uint32_t nbits = 0, wf[3] = {0, 0, 0};
  size_t plane_off[3] = {0, 0, 0};
  size_t plane_count = 0;
  int multi_levels = 0;  // > 0: several small MSMs in ONE pass (msm.hip: MsmMulti); bucket sets [l W, (l + 1) W) belong to call l
  int multi_W = 0;
};
struct MsmStreamSlot {
  uint8_t *raw = nullptr, *packed = nullptr, *scalars = nullptr;  // staged records, device-form bases, scalars
  size_t raw_cap = 0, packed_cap = 0, scalars_cap = 0;
  hipEvent_t copied;
  bool have_ev = false;
  bool inflight = false;
  MsmPending P;
};
struct MsmStream {
  std::mutex mu;
  size_t chunk = 0, stride = 96;
  int mont = 0;
  uint64_t bases_handle = 0;  // 0: the pairs carry their bases; else scalars only, against registered bases
  int64_t base0 = 0, next_base = 0, step = 1;
  MsmStreamSlot s[2];
  int cur = 0;
  size_t fill = 0, total = 0;
  hipStream_t copy = nullptr;
  uint64_t acc[18];  // running sum (Jacobian, host)
  bool acc_set = false;
};

constexpr int MSM_SLOTS = 4;
constexpr int MSM_SMALL_LANES = 6;  // small calls of a batch in flight side by side (GM_MSM_SMALL_LANES uses fewer); a lane's workspace is ~150 MB.
// Measured on one box, snark -i 18 / -i 20 / sharded -i 21 / -i 22 (ms): 2 lanes 14.6 / 21.5 / 36.0 / 50.7, 4 lanes 13.1 / 19.8 / 33.5 / 48.1,
// 6 lanes 11.7 / 18.7 / 32.0 / 47.4, 8 lanes 17.0 / 21.1 / 33.6 / 48.5, 12 lanes 17.6 / 22.5 / 35.0 / 51.3: beyond six the chains only get in each other's way
struct MsmWorkspace {
  DevBuf scalars, counts, offsets, cursor, entries, tmp_entries, sortmeta, buckets, pk[2], pp[2], rows, cols, planes, misc;
  DevBuf clk;  // {shader cycles, 100 MHz ticks} of the first wave of the last k_acc0 (read only while profiling: gm_prof_read_clock)
  DevBuf lvl_cnt, lvl_pos, lvl_pts[2], lvl_keys[2], lvl_prefix, lvl_lane, lvl_entries, lvl_n;  // affine tree levels
  // pinned staging for the D2H of window bit-planes + the event behind it, per result slot: MSM_SLOTS calls of a lane can be enqueued
  // behind one another before the host finishes the first (big lanes use two, small lanes all of them)
  uint64_t* host_planes[MSM_SLOTS] = {};
  size_t host_planes_cap[MSM_SLOTS] = {};
  hipEvent_t done_ev[MSM_SLOTS];
  hipEvent_t sort_ev, acc_ev;  // phase hand-over between the streams of a split call
  bool have_done_ev = false;
};

// per-stage kernel timing with HIP events on the library's own stream (bench.py's roofline leg)
enum ProfStage { PROF_DIGITS = 0, PROF_SCAN, PROF_SCATTER, PROF_ACC0, PROF_MERGE, PROF_REDUCE, PROF_SC_ROUND, PROF_NSTAGES };
struct Profiler {
  bool on = false;
  int only_stage = -1;  // >= 0: events around that stage only (gm_prof_enable(2): the accumulation; every event record is a ~10 us bubble)
  hipEvent_t ev[2][2 * PROF_NSTAGES];  // [window group of a split MSM call][stage begin / end]
  bool have_events = false;
  bool pending[2][PROF_NSTAGES] = {};
  double ms[PROF_NSTAGES] = {};     // summed over calls AND over the window groups of a call
  uint64_t count[PROF_NSTAGES] = {};  // launches of the stage (a split call counts two)
  double acc0_cycles = 0, acc0_ticks = 0;  // clock64() / wall_clock64() spans of k_acc0's first wave, summed over calls
  void begin(int part, int stage, hipStream_t st);
  void end(int part, int stage, hipStream_t st);
  void begin(int stage, hipStream_t st) { begin(0, stage, st); }
  void end(int stage, hipStream_t st) { end(0, stage, st); }
  void collect();  // after the streams of the call have been waited for
};

// Caching allocator for device vectors: the prover allocates and drops O(100) multi-hundred-MB
// vectors per proof; hipMalloc/hipFree are synchronous and cost milliseconds at that size.
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_list;  // capacity -> pointer
  size_t pooled_bytes = 0;
  size_t max_pooled = (size_t)96 << 30;  // GM_POOL_MAX_GB overrides (gm_init)
  int alloc(size_t bytes, void** p, size_t* cap);
  void free(void* p, size_t cap);
  void release_all();
};

// the (device, pinned host) buffer pair a sumcheck / space prover collects its per-block partial sums in: 32 KiB each.  hipMalloc,
// hipHostMalloc and above all hipFree (a device-wide synchronisation) cost 0.1-0.3 ms apiece -- the preprocessing prover creates 15
// provers per proof -- so freed pairs are kept for the next prover
struct PartialBufs {
  std::mutex mu;
  std::vector<std::pair<uint8_t*, uint64_t*>> free_pairs;
  int take(uint8_t** dev, uint64_t** host);
  void give(uint8_t* dev, uint64_t* host);
  void release_all();
};
constexpr size_t PARTIAL_BUF_BYTES = 512 * 2 * 32;

struct Context {
  int device = -1;
  PartialBufs partial_bufs;
  DevPool pool;
  Profiler prof;
  hipStream_t stream = nullptr;
  std::mutex mu;       // guards handle tables
  // The MSM workspaces are single-flight (the reference's MSM calls are sequential): msm_mu is held from the
  // staging of host scalars to the end of the call (recursive: msm_host_scalars -> msm_run).
  std::recursive_mutex msm_mu;
  // Vector entry points (gm_fr_*) stage per-call parameters and partial results in fr_scratch / host_small and
  // share one stream: fr_mu serialises them, so calls from different threads are safe (and ordered).
  std::recursive_mutex fr_mu;
  uint64_t next_handle = 1;
  std::unordered_map<uint64_t, std::unique_ptr<Bases>> bases;
  std::unordered_map<uint64_t, std::unique_ptr<FrVec>> vecs;
  std::unordered_map<uint64_t, std::unique_ptr<Sumcheck>> provers;
  std::unordered_map<uint64_t, std::unique_ptr<SparseMatrix>> matrices;
  std::unordered_map<uint64_t, std::unique_ptr<SpaceProver>> space_provers;
  std::unordered_map<uint64_t, std::unique_ptr<HerringG1>> herring_g1;
  std::unordered_map<uint64_t, std::unique_ptr<MsmStream>> msm_streams;
  std::unordered_map<uint64_t, std::unique_ptr<IdxVec>> indices;
  MsmWorkspace msm;
  // extra workspaces + streams for the small calls of a batch (msm_run_batch)
  MsmWorkspace msm_small[MSM_SMALL_LANES];
  hipStream_t small_stream[MSM_SMALL_LANES] = {};
  // second full-size workspace + stream: the big calls of a batch alternate between the two so that the
  // sort / merge / reduce of one overlap the (ALU-bound) accumulation of the other
  MsmWorkspace msm_b;
  hipStream_t stream_b = nullptr;
  // XCD PARTITION of a batch (GM_CU_SPLIT, default one XCD; capi.hip: gm_init): the latency-bound tail of every call of a batch
  // (k_merge, the bucket reduction, the copy-out) runs on streams created with a CU mask that covers ONE whole XCD, the
  // accumulations (and sorts) on streams masked to the other seven.  Without it a small kernel does not progress beside an
  // accumulation: the SIMD arbiter serves the oldest waves first and a k_acc0 grid keeps every SIMD supplied with older ones
  // (HISTORY section 4.1), so the tails of a batch end up serialised behind the accumulations they were meant to hide under.
  // Index 0 / 1 / 2.. = main, second big, small lanes
  int cu_split = 0;
  hipStream_t part_acc[2 + MSM_SMALL_LANES] = {};
  hipStream_t part_tail[2 + MSM_SMALL_LANES] = {};
  hipEvent_t start_ev;  // recorded on `stream` when an MSM call starts: its other streams wait for the producer of the scalars
  bool have_start_ev = false;
  DevBuf fr_scratch;
  uint64_t* host_small = nullptr;  // pinned, 64 KiB, for small results
  // pinned staging of the descriptor arrays of k_sc_round_multi: a RING of 8 slots of 32 KiB, one per launch.  The copy to the device is
  // asynchronous and the round is split-phase, so the source must outlive the call (a local vector would not) and must not be rewritten by the
  // next launch of the same round (a sharded batch makes up to four per round); a slot comes round again after 8 launches, i.e. after at least
  // one collected round (sc_round_end waits for the stream)
  uint8_t* sc_desc_host = nullptr;
  unsigned sc_desc_next = 0;
  uint64_t* host_batch = nullptr;  // pinned, grow-only: partial sums of a batch of evaluations (fr_eval_le_batch)
  size_t host_batch_cap = 0;
  int msm_c_override = 0;
  int msm_glv = 0;            // build phi(P) at registration and split scalars by the GLV endomorphism (gm_set_msm_glv)
  int msm_split = 0;          // one-call MSMs as two window groups over three streams (gm_set_msm_split)
  int msm_affine_levels = 0;  // affine tree levels in front of the XYZZ accumulation; -1 = automatic
  size_t msm_table_min = (size_t)1 << 17;  // smallest MSM that uses fixed-base tables when present
  // small results (per-block partial sums, the bit-plane sums of an MSM) are written by their kernels straight into pinned host
  // memory instead of a device buffer that a blit kernel then copies (GM_ZERO_COPY=0 restores the copies; bit 0 field paths, bit 1 MSM planes.  Two A/B runs on one box: snark -i 20 14.9 -> 14.5 ms, -i 24 117.8 -> 117.0, psnark -i 20 129 -> 127, same proof bytes: profiles/r4_zero_copy_probe.txt)
  int zero_copy = 3;
  std::atomic<int> msm_busy{0};  // open GM_MSM_LOCK scopes
  bool auto_tables = true;    // build fixed-base tables when bases are registered, if they fit (gm_set_auto_tables)
  size_t auto_tables_max = 0;  // byte budget of one key's tables; 0 = 30 % of the device memory
  int cu_count = 256;
};

Context* context();  // nullptr before gm_init
// Every device allocation of the library goes through these two (hipMalloc / hipFree + bookkeeping): what the process holds from the
// driver, its high-water mark, and -- minus the vector pool's cached blocks -- what is IN USE and its peak (gm_mem_stats; the
// footprint contract of the provers is checked against these figures, tests/test_gpu_footprint.py)
struct MemStats {
  std::mutex mu;
  std::unordered_map<void*, size_t> sizes;
  size_t live = 0, peak_live = 0;  // bytes obtained from the driver and not given back
  size_t cached = 0;               // of those: freed blocks the vector pool keeps for reuse
  size_t peak_in_use = 0;          // high-water mark of live - cached
  uint64_t spare_table_releases = 0;  // release_spare_tables() calls that freed something (no silent degradation)
  void note_cached(ptrdiff_t delta);  // the pool took (+) or handed out (-) a cached block
};
MemStats& mem_stats();
hipError_t raw_malloc_v(void** p, size_t bytes);
template <class T>
inline hipError_t raw_malloc(T** p, size_t bytes) {
  return raw_malloc_v(reinterpret_cast<void**>(p), bytes);
}
hipError_t raw_free(void* p);
hipError_t dev_malloc(void** p, size_t bytes);  // raw_malloc that gives the vector pool's freed blocks back on OOM
// last resort of an allocation that failed twice: the PREFIX tables of every key (Bases::extra) go, the calls they served take the
// plain path from then on.  False when nothing was freed or an MSM may be reading them (any open GM_MSM_LOCK scope).
bool release_spare_tables(Context* C);

Bases* find_bases(uint64_t h);
FrVec* find_vec(uint64_t h);
Sumcheck* find_prover(uint64_t h);
uint64_t put_bases(std::unique_ptr<Bases> b);
uint64_t put_vec(std::unique_ptr<FrVec> v);
uint64_t put_prover(std::unique_ptr<Sumcheck> p);

#define GM_FR_LOCK(C) std::lock_guard<std::recursive_mutex> gm_fr_lock_((C)->fr_mu)
// (the guard also counts the MSM scopes that are open: release_spare_tables must not free a table an MSM in flight reads)
struct MsmBusyGuard {
  std::atomic<int>& n;
  explicit MsmBusyGuard(std::atomic<int>& c) : n(c) { n.fetch_add(1); }
  ~MsmBusyGuard() { n.fetch_sub(1); }
};
#define GM_MSM_LOCK(C)                                                \
  std::lock_guard<std::recursive_mutex> gm_msm_lock_((C)->msm_mu); \
  ::gm::MsmBusyGuard gm_msm_busy_((C)->msm_busy)

#define GM_CTX()                                            \
  ::gm::Context* C = ::gm::context();                       \
  GM_CHECK(C != nullptr, GM_ENOTINIT, "gm_init has not been called")

// MSM engine (msm.hip)
int bases_build_prefix_sets(Context* C, Bases* b);
size_t msm_workspace_held(const MsmWorkspace& ws);
size_t msm_workspace_bound(Context* C, const Bases* bases, size_t n);
int msm_run(Context* C, const Bases* bases, int64_t first, int64_t step, const void* d_scalars, int mont, size_t n,
            bool normalize, uint64_t out_jac[18]);
int msm_run_batch(Context* C, const Bases* bases, int64_t first, int64_t step, const void* const* d_scalars, int mont, const size_t* ns,
                  size_t k, bool normalize, uint64_t* out_jac);
// call j walks the bases from index pair_offsets[j] (step +1) or DOWN from it (step -1)
int msm_run_batch_offsets(Context* C, const Bases* bases, const size_t* pair_offsets, int64_t step, const void* const* d_scalars, int mont,
                          const size_t* ns, size_t k, bool normalize, uint64_t* out_jac);

}  // namespace gm
