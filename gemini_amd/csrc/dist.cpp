// The collective layer of libgemini_hip.so: one process per GPU, an all-gather among them INSIDE the library so that a
// Rust / C++ embedder can shard the prover without torch (BASELINE north_star: "independent MSM chunks and sumcheck rounds
// shard across the 8 GPUs of one node with a final RCCL reduce of partial G1 points over xGMI").
//
// What the path exchanges: 144-byte partial G1 points per commitment (EC addition is not an RCCL reduce
// op: every rank all-gathers and adds locally, gm_g1_sum), 64 bytes per sumcheck round, a few field elements per evaluation,
// and -- once per sumcheck / folding tree -- the short tails of block-sharded vectors.  All of it is ONE primitive, all-gather,
// in two flavours: host payloads (results the host produced: an MSM partial is finished by the host Horner) and device-resident
// Fr vectors.
//
// Three transports behind the same calls:
//   * RCCL   gm_dist_init_rccl: ncclAllGather on a communicator of the library's own (librccl is dlopen'ed: the library does
//            not drag RCCL into single-GPU processes; the function types come from <rccl/rccl.h>, so the binding is compile-
//            checked against the header of the image).  Host payloads are staged pinned host -> device -> ncclAllGather ->
//            pinned host on the library's stream, ONE wait per collective; device vectors go device to device over xGMI.
//   * hook   gm_dist_init_hook: the embedder's own all-gather (MPI, gloo, a torch.distributed call ...) for host buffers;
//            device vectors are staged through the host.  What the shared-GPU tests use (RCCL refuses two ranks on one device).
//   * shm    gm_dist_init_shm: ranks of one node over a POSIX shared-memory segment (sequence counters, no syscalls on the fast
//            path): the payloads of this path are host data of <= 1 KiB, for which a store + a load across processes (~1 us)
//            beats H2D + a GPU collective + D2H (~50 us).  Also what lets the C-ABI tests run N ranks with no Python at all.
// world = 1 (or no gm_dist_init_*): every all-gather is a copy.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <type_traits>

#include "ctx.hpp"

// The RCCL types this file needs.  With the development header present (and no -DGM_NO_RCCL) they are RCCL's own and the function
// pointer types below are compile-checked against its prototypes; without it -- a ROCm image that ships librccl.so but not
// <rccl/rccl.h>, or a single-GPU build -- the few ABI facts used here are declared locally (a 128-byte id, an opaque communicator,
// ncclSuccess = 0, ncclChar = 0: stable since NCCL 2.0) and the library still builds; librccl itself is only ever dlopen'ed.
#if !defined(GM_NO_RCCL) && defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define GM_HAVE_RCCL_HEADER 1
#endif
#endif
#ifndef GM_HAVE_RCCL_HEADER
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0 } ncclDataType_t;
}
#endif

namespace {

using Clock = std::chrono::steady_clock;

// T_FAILED: a collective of this communicator failed or timed out and the communicator was aborted.  rank / world keep their values, so
// every later collective returns GM_ESTATE (instead of "succeeding" as the copy of a one-rank world) until gm_dist_finalize / a re-init
enum Transport { T_NONE = 0, T_HOOK = 1, T_RCCL = 2, T_SHM = 3, T_FAILED = 4 };

using GetUniqueId_t = ncclResult_t (*)(ncclUniqueId*);
using CommInitRank_t = ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int);
using AllGather_t = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
using CommDestroy_t = ncclResult_t (*)(ncclComm_t);
using GetErrorString_t = const char* (*)(ncclResult_t);
using SendRecv_t = ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);  // ncclSend takes const void*: cast at the call
using Group_t = ncclResult_t (*)();
#ifdef GM_HAVE_RCCL_HEADER
static_assert(std::is_same<GetUniqueId_t, decltype(&ncclGetUniqueId)>::value, "ncclGetUniqueId prototype");
static_assert(std::is_same<CommInitRank_t, decltype(&ncclCommInitRank)>::value, "ncclCommInitRank prototype");
static_assert(std::is_same<AllGather_t, decltype(&ncclAllGather)>::value, "ncclAllGather prototype");
static_assert(std::is_same<CommDestroy_t, decltype(&ncclCommDestroy)>::value, "ncclCommDestroy prototype");
static_assert(std::is_same<GetErrorString_t, decltype(&ncclGetErrorString)>::value, "ncclGetErrorString prototype");
static_assert(std::is_same<SendRecv_t, decltype(&ncclRecv)>::value, "ncclRecv prototype (ncclSend: the same with a const buffer)");
static_assert(std::is_same<Group_t, decltype(&ncclGroupStart)>::value, "ncclGroupStart prototype");
#endif
struct Rccl {
  void* lib = nullptr;
  GetUniqueId_t GetUniqueId = nullptr;
  CommInitRank_t CommInitRank = nullptr;
  AllGather_t AllGather = nullptr;
  CommDestroy_t CommDestroy = nullptr;
  CommDestroy_t CommAbort = nullptr;  // optional: a rank that fails inside a collective aborts the communicator so that its peers do not hang
  SendRecv_t Send = nullptr, Recv = nullptr;  // optional (gm_dist_reblock_vecs falls back to all-gathers without them)
  Group_t GroupStart = nullptr, GroupEnd = nullptr;
  GetErrorString_t GetErrorString = nullptr;
};

// shared segment: header, then two banks (call parity) of world slots of slot_bytes
struct ShmHeader {
  std::atomic<uint64_t> magic;
  uint64_t world, slot_bytes;
  std::atomic<uint64_t> attached;
  std::atomic<uint64_t> seq[64];  // seq[r] = number of the last call rank r has published
  // liveness handshake of the attach: a peer stores a fresh random nonce in hello[r] and trusts the segment only once rank 0 OF
  // THIS RUN has echoed it in ack[r] -- the rank 0 of a crashed run never will, whatever its counters say
  std::atomic<uint64_t> hello[64], ack[64];
  // 1 + the rank that called gm_dist_abort (a prover that failed OUTSIDE a collective: its peers would otherwise wait out their timeout in the
  // next all-gather); every wait on this segment returns GM_ESTATE once it is set
  std::atomic<uint64_t> abort_by;
};
constexpr uint64_t SHM_MAGIC = 0x474d44495354ull;  // "GMDIST"
constexpr uint64_t SHM_DEAD = 0x44454144474dull;   // a newer run has replaced this segment (rank 0 poisons what it unlinks)

// routes a collective can take (gm_dist_stats_routes)
enum Route { R_COPY = 0, R_HOOK = 1, R_RCCL_HOST = 2, R_RCCL_VEC = 3, R_SHM = 4, R_NROUTES = 5 };

struct Dist {
  std::mutex mu;
  int rank = 0, world = 1;
  Transport tr = T_NONE;
  // hook
  gm_allgather_fn fn = nullptr;
  void* fn_ctx = nullptr;
  // rccl
  Rccl R;
  ncclComm_t comm = nullptr;
  void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;  // staging of host payloads
  size_t stage_in = 0, stage_out = 0;
  // shm (the transport of gm_dist_init_shm, and the SIDE CHANNEL of gm_dist_init_rccl_node for small host payloads)
  ShmHeader* shm = nullptr;
  size_t shm_bytes = 0, slot_bytes = 0;
  uint64_t shm_call = 0;
  std::string shm_name;
  ino_t shm_ino = 0;  // the inode the name resolved to when this rank attached: only that one is ever unlinked
  dev_t shm_dev = 0;
  // under RCCL: which payload classes go over the side segment (bit 0: field values, bit 1: partial G1 points)
  unsigned shm_classes = 0;
  // statistics
  uint64_t calls = 0, bytes = 0;
  double seconds = 0.0;
  uint64_t rcalls[R_NROUTES] = {}, rbytes[R_NROUTES] = {};
  double rseconds[R_NROUTES] = {};
  void note(Route r, size_t b, double sec) {
    calls++;
    bytes += b;
    seconds += sec;
    rcalls[r]++;
    rbytes[r] += b;
    rseconds[r] += sec;
  }
};

Dist& D() {
  static Dist d;
  return d;
}

int load_rccl(Rccl& R) {
  if (R.lib) return GM_OK;
  // RCCL must sit on the SAME HIP runtime as this library (its streams and device pointers are handed to ncclAllGather): a
  // process may hold a second runtime -- the PyTorch wheel ships its own libamdhip64 / librccl -- so the first candidate is the
  // librccl next to the libamdhip64 this library is bound to, found through the address of one of its functions.
  std::string beside;
  {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipStreamSynchronize), &info) && info.dli_fname) {
      beside = info.dli_fname;
      // (for a stand-in loaded through GM_RCCL_LIB that must call into THIS runtime: tests/fake_rccl)
      setenv("GM_HIP_RUNTIME", info.dli_fname, 1);
      const size_t slash = beside.rfind('/');
      beside = slash == std::string::npos ? std::string() : beside.substr(0, slash) + "/librccl.so.1";
    }
  }
  const char* override_path = getenv("GM_RCCL_LIB");
  const char* names[] = {override_path, beside.empty() ? nullptr : beside.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* nm : names) {
    if (!nm) continue;
    R.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (R.lib) break;
  }
  GM_CHECK(R.lib != nullptr, GM_ESTATE, "gm_dist: librccl.so not found (%s); set GM_RCCL_LIB", dlerror());
#define GM_SYM(field, name)                                                   \
  R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, name));          \
  GM_CHECK(R.field != nullptr, GM_ESTATE, "gm_dist: librccl lacks %s", name)
  GM_SYM(GetUniqueId, "ncclGetUniqueId");
  GM_SYM(CommInitRank, "ncclCommInitRank");
  GM_SYM(AllGather, "ncclAllGather");
  GM_SYM(CommDestroy, "ncclCommDestroy");
  GM_SYM(GetErrorString, "ncclGetErrorString");
#undef GM_SYM
  R.CommAbort = reinterpret_cast<CommDestroy_t>(dlsym(R.lib, "ncclCommAbort"));
  R.Send = reinterpret_cast<SendRecv_t>(dlsym(R.lib, "ncclSend"));
  R.Recv = reinterpret_cast<SendRecv_t>(dlsym(R.lib, "ncclRecv"));
  R.GroupStart = reinterpret_cast<Group_t>(dlsym(R.lib, "ncclGroupStart"));
  R.GroupEnd = reinterpret_cast<Group_t>(dlsym(R.lib, "ncclGroupEnd"));
  return GM_OK;
}

#define GM_NCCL(d, expr)                                                                   \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    GM_CHECK(r_ == ncclSuccess, GM_EHIP, "gm_dist: %s failed: %s", #expr, (d).R.GetErrorString(r_)); \
  } while (0)

int ensure_stage(Dist& d, size_t bytes) {
  const size_t out = bytes * (size_t)d.world;
  if (bytes > d.stage_in) {
    if (d.h_in) (void)hipHostFree(d.h_in);
    if (d.d_in) (void)gm::raw_free(d.d_in);
    const size_t cap = bytes < 4096 ? 4096 : bytes * 2;
    GM_HIP(hipHostMalloc(&d.h_in, cap, hipHostMallocDefault));
    GM_HIP(gm::raw_malloc(&d.d_in, cap));
    d.stage_in = cap;
  }
  if (out > d.stage_out) {
    if (d.h_out) (void)hipHostFree(d.h_out);
    if (d.d_out) (void)gm::raw_free(d.d_out);
    const size_t cap = out < 4096 * (size_t)d.world ? 4096 * (size_t)d.world : out * 2;
    GM_HIP(hipHostMalloc(&d.h_out, cap, hipHostMallocDefault));
    GM_HIP(gm::raw_malloc(&d.d_out, cap));
    d.stage_out = cap;
  }
  return GM_OK;
}

// ---- shm ------------------------------------------------------------------------------------------------------------
uint8_t* shm_slot(Dist& d, uint64_t call, int r) {
  uint8_t* base = reinterpret_cast<uint8_t*>(d.shm) + ((sizeof(ShmHeader) + 63) & ~(size_t)63);
  return base + ((call & 1) * (size_t)d.world + (size_t)r) * d.slot_bytes;
}

// A collective over RCCL whose peer never arrives would block hipStreamSynchronize for ever (the shm transport has its own
// timeout): poll the stream instead and give up after GM_DIST_RCCL_TIMEOUT_S (default 120 s) -- the caller then aborts the communicator.
hipError_t stream_wait_bounded(hipStream_t st, bool* timed_out, const std::atomic<uint64_t>* abort_flag = nullptr) {
  static const double limit = getenv("GM_DIST_RCCL_TIMEOUT_S") ? atof(getenv("GM_DIST_RCCL_TIMEOUT_S")) : 120.0;
  *timed_out = false;
  const auto t0 = Clock::now();
  for (unsigned spin = 0;; spin++) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (abort_flag && (spin & 63) == 0 && abort_flag->load(std::memory_order_acquire) != 0) {  // a peer called gm_dist_abort: it will never arrive
      *timed_out = true;
      return hipErrorNotReady;
    }
    if (spin > 20000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 1023) == 0 && std::chrono::duration<double>(Clock::now() - t0).count() > limit) {
      *timed_out = true;
      return hipErrorNotReady;
    }
  }
}

double shm_timeout() {
  static const double limit = getenv("GM_DIST_TIMEOUT_S") ? atof(getenv("GM_DIST_TIMEOUT_S")) : 300.0;
  return limit;
}

// Wait until a peer's counter reaches call k.  When this rank reads it the peer can only be AT k or ONE call ahead (it cannot
// publish k + 2 before this rank has published k + 1): any other value means the segment is not the one of this run -- a stale
// segment of a crashed run a rank attached to before rank 0 replaced it, or a corrupted counter -- and is an error, never data.
int shm_wait_seq(ShmHeader* h, int r, uint64_t k) {
  const auto t0 = Clock::now();
  std::atomic<uint64_t>& a = h->seq[r];
  for (unsigned spin = 0;; spin++) {
    const uint64_t v = a.load(std::memory_order_acquire);
    if (v == k || v == k + 1) return GM_OK;
    {
      const uint64_t ab = h->abort_by.load(std::memory_order_acquire);
      GM_CHECK(ab == 0, GM_ESTATE, "gm_dist: rank %llu aborted the run (gm_dist_abort: a prover failed on that rank); this rank stops waiting for rank %d",
               (unsigned long long)(ab - 1), r);
    }
    GM_CHECK(v < k, GM_ESTATE, "gm_dist(shm): rank %d is at call %llu while this rank is at call %llu: a stale or corrupted segment", r,
             (unsigned long long)v, (unsigned long long)k);
    if (spin < 2000) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    } else {
      // ranks may outnumber the cores the container may use (8 ranks + their helper threads on a 16-CPU quota)
      std::this_thread::yield();
      if ((spin & 1023) == 0) {
        GM_CHECK(h->magic.load(std::memory_order_acquire) == SHM_MAGIC, GM_ESTATE, "gm_dist(shm): the segment was replaced by a newer run");
        GM_CHECK(std::chrono::duration<double>(Clock::now() - t0).count() < shm_timeout(), GM_ESTATE,
                 "gm_dist(shm): all-gather: rank %d did not arrive within %.0f s", r, shm_timeout());
      }
    }
  }
}

// one all-gather of `bytes` <= slot_bytes per rank.  Call k: publish into bank k & 1, raise seq[rank] to k, read every peer's
// slot once its seq reaches k.  A rank can only start call k + 2 (the next writer of bank k & 1) after it has seen every
// seq >= k + 1, i.e. after every peer has finished READING call k: two banks suffice.
int shm_allgather_once(Dist& d, const void* send, size_t bytes, void* recv) {
  const uint64_t k = ++d.shm_call;
  memcpy(shm_slot(d, k, d.rank), send, bytes);
  d.shm->seq[d.rank].store(k, std::memory_order_release);
  for (int i = 0; i < d.world; i++) {
    const int r = (d.rank + i) % d.world;
    if (r != d.rank) {
      int rc = shm_wait_seq(d.shm, r, k);
      if (rc) return rc;
    }
    memcpy(static_cast<uint8_t*>(recv) + (size_t)r * bytes, shm_slot(d, k, r), bytes);
  }
  return GM_OK;
}

int shm_allgather(Dist& d, const void* send, size_t bytes, void* recv) {
  if (bytes <= d.slot_bytes) return shm_allgather_once(d, send, bytes, recv);
  // longer payloads: slot-sized pieces, each rank's pieces land at recv + r * bytes + offset
  std::vector<uint8_t> tmp((size_t)d.world * d.slot_bytes);
  for (size_t off = 0; off < bytes; off += d.slot_bytes) {
    const size_t m = bytes - off < d.slot_bytes ? bytes - off : d.slot_bytes;
    int rc = shm_allgather_once(d, static_cast<const uint8_t*>(send) + off, m, tmp.data());
    if (rc) return rc;
    for (int r = 0; r < d.world; r++) memcpy(static_cast<uint8_t*>(recv) + (size_t)r * bytes + off, tmp.data() + (size_t)r * m, m);
  }
  return GM_OK;
}

// does `name` still resolve to the segment this rank mapped?
bool shm_name_is(const char* name, dev_t dev, ino_t ino) {
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return false;
  struct stat st;
  const bool same = fstat(fd, &st) == 0 && st.st_dev == dev && st.st_ino == ino;
  close(fd);
  return same;
}

void shm_detach(Dist& d, bool count = true) {
  if (!d.shm) return;
  const uint64_t left = count ? d.shm->attached.fetch_sub(1) - 1 : 1;
  munmap(d.shm, d.shm_bytes);
  // the last one out removes the name -- if it still names THIS segment (a re-init under the same name may have replaced it)
  if (left == 0 && shm_name_is(d.shm_name.c_str(), d.shm_dev, d.shm_ino)) shm_unlink(d.shm_name.c_str());
  d.shm = nullptr;
  d.shm_classes = 0;
}

// a failed / timed-out RCCL collective: abort the communicator (the peers return instead of hanging) and refuse everything after it
void poison(Dist& d) {
  if (d.comm && d.R.CommAbort) (void)d.R.CommAbort(d.comm);
  else if (d.comm) (void)d.R.CommDestroy(d.comm);
  d.comm = nullptr;
  d.tr = T_FAILED;
}
#define GM_DIST_ALIVE(d)                                                                                                             \
  GM_CHECK((d).tr != T_FAILED, GM_ESTATE, "gm_dist: an earlier collective of this communicator failed or timed out (rank %d of %d); " \
                                          "gm_dist_finalize and initialise the transport again",                                        \
           (d).rank, (d).world);                                                                                                       \
  GM_CHECK((d).tr != T_NONE || (d).world == 1, GM_ESTATE, "gm_dist: %d ranks but no transport", (d).world)

void reset(Dist& d) {
  if (d.comm) {
    (void)d.R.CommDestroy(d.comm);
    d.comm = nullptr;
  }
  if (d.h_in) (void)hipHostFree(d.h_in);
  if (d.h_out) (void)hipHostFree(d.h_out);
  if (d.d_in) (void)gm::raw_free(d.d_in);
  if (d.d_out) (void)gm::raw_free(d.d_out);
  d.h_in = d.h_out = d.d_in = d.d_out = nullptr;
  d.stage_in = d.stage_out = 0;
  shm_detach(d);
  d.tr = T_NONE;
  d.rank = 0;
  d.world = 1;
  d.fn = nullptr;
  d.fn_ctx = nullptr;
  d.shm_call = 0;
}

// GM_DIST_CLASS_*: what a host payload is.  Under RCCL with the node's side segment open (gm_dist_init_rccl_node) the FIELD values
// of the path -- 64 bytes per sumcheck round, a few evaluations, the 2^10-element tails -- cross processes as a store and a load
// (a few us) instead of H2D + ncclAllGather + D2H + a stream wait (profiles/r5_collective_latency.txt); partial G1 points keep
// ncclAllGather (north_star: "a final RCCL reduce of partial G1 points over xGMI") unless GM_DIST_G1_ROUTE=shm.
int rccl_host_allgather(Dist& d, const void* send, size_t bytes, void* recv) {
  gm::Context* C = gm::context();
  GM_CHECK(C != nullptr, GM_ENOTINIT, "gm_init has not been called");
  int rc = GM_OK;
  if ((rc = ensure_stage(d, bytes))) return rc;
  memcpy(d.h_in, send, bytes);
  // a rank that fails here leaves its peers inside ncclAllGather: abort the communicator so that they return with an error
  // instead of hanging (the shm transport has a timeout of its own)
  auto fail = [&](int code) {
    poison(d);
    return code;
  };
  hipError_t e = hipMemcpyAsync(d.d_in, d.h_in, bytes, hipMemcpyHostToDevice, C->stream);
  if (e != hipSuccess) return fail(gm::hip_fail(e, "hipMemcpyAsync(H2D staging)", __FILE__, __LINE__));
  ncclResult_t r = d.R.AllGather(d.d_in, d.d_out, bytes, ncclChar, d.comm, C->stream);
  if (r != ncclSuccess) {
    gm::set_error("gm_dist: ncclAllGather failed: %s", d.R.GetErrorString(r));
    return fail(GM_EHIP);
  }
  e = hipMemcpyAsync(d.h_out, d.d_out, bytes * (size_t)d.world, hipMemcpyDeviceToHost, C->stream);
  bool late = false;
  if (e == hipSuccess) e = d.world > 1 ? stream_wait_bounded(C->stream, &late, d.shm ? &d.shm->abort_by : nullptr) : hipStreamSynchronize(C->stream);
  if (late) {
    gm::set_error("gm_dist: ncclAllGather did not complete in time: a peer never arrived");
    return fail(GM_ESTATE);
  }
  if (e != hipSuccess) return fail(gm::hip_fail(e, "D2H staging / wait", __FILE__, __LINE__));
  memcpy(recv, d.h_out, bytes * (size_t)d.world);
  return GM_OK;
}

int allgather_host_locked(Dist& d, const void* send, size_t bytes, void* recv, int cls = GM_DIST_CLASS_FIELD, int force_route = -1) {
  GM_DIST_ALIVE(d);
  if (bytes == 0) return GM_OK;
  const auto t0 = Clock::now();
  int rc = GM_OK;
  Route route = R_COPY;
  switch (d.tr) {
    case T_FAILED:  // (refused above)
    case T_NONE:
      memcpy(recv, send, bytes);
      break;
    case T_HOOK:
      route = R_HOOK;
      rc = d.fn(d.fn_ctx, send, bytes, recv);
      if (rc) gm::set_error("gm_dist: the all-gather hook returned %d", rc);
      break;
    case T_SHM:
      route = R_SHM;
      rc = shm_allgather(d, send, bytes, recv);
      break;
    case T_RCCL: {
      bool side = d.shm != nullptr && (d.shm_classes >> (cls == GM_DIST_CLASS_G1 ? 1 : 0) & 1u) != 0;
      if (force_route == R_SHM) side = d.shm != nullptr;
      if (force_route == R_RCCL_HOST) side = false;
      route = side ? R_SHM : R_RCCL_HOST;
      rc = side ? shm_allgather(d, send, bytes, recv) : rccl_host_allgather(d, send, bytes, recv);
      break;
    }
  }
  d.note(route, bytes * (size_t)d.world, std::chrono::duration<double>(Clock::now() - t0).count());
  return rc;
}

// Attach to (rank > 0) or create (rank 0) the segment `name`; leaves d.tr alone.  Rank 0 POISONS a segment of that name it finds
// (magic = SHM_DEAD) before unlinking it and creates the new one under O_EXCL; a peer accepts a segment only while its magic is
// SHM_MAGIC and the name still resolves to the inode it mapped, and re-attaches otherwise -- so a peer that opened the stale
// segment of a crashed run (bench.py reuses "/gm_bench_<port>") either never trusts it or notices within its first wait.
int shm_open_segment(Dist& d, int rank, int world, const char* name, size_t slot_bytes) {
  if (slot_bytes == 0) slot_bytes = (size_t)1 << 20;
  slot_bytes = (slot_bytes + 63) & ~(size_t)63;
  const size_t total = ((sizeof(ShmHeader) + 63) & ~(size_t)63) + 2 * (size_t)world * slot_bytes;
  const auto t0 = Clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(Clock::now() - t0).count(); };
  d.shm_name = name;
  d.slot_bytes = slot_bytes;
  d.shm_bytes = total;
  d.shm_call = 0;
  if (rank == 0) {
    int old = shm_open(name, O_RDWR, 0600);
    if (old >= 0) {  // a stale segment of a crashed run (or of a previous init under this name): poison, then unlink
      struct stat st;
      if (fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(ShmHeader)) {
        void* q = mmap(nullptr, sizeof(ShmHeader), PROT_READ | PROT_WRITE, MAP_SHARED, old, 0);
        if (q != MAP_FAILED) {
          static_cast<ShmHeader*>(q)->magic.store(SHM_DEAD, std::memory_order_release);
          munmap(q, sizeof(ShmHeader));
        }
      }
      close(old);
      (void)shm_unlink(name);
    }
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    GM_CHECK(fd >= 0, GM_ESTATE, "gm_dist_init_shm: shm_open(%s) failed: %s", name, strerror(errno));
    struct stat st;
    if (ftruncate(fd, (off_t)total) != 0 || fstat(fd, &st) != 0) {
      close(fd);
      (void)shm_unlink(name);
      GM_CHECK(false, GM_ENOMEM, "gm_dist_init_shm: ftruncate(%zu) failed: %s", total, strerror(errno));
    }
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    GM_CHECK(p != MAP_FAILED, GM_ENOMEM, "gm_dist_init_shm: mmap failed: %s", strerror(errno));
    d.shm = static_cast<ShmHeader*>(p);
    d.shm_ino = st.st_ino;
    d.shm_dev = st.st_dev;
    d.shm->world = (uint64_t)world;
    d.shm->slot_bytes = slot_bytes;
    for (auto& q : d.shm->seq) q.store(0, std::memory_order_relaxed);
    for (auto& q : d.shm->hello) q.store(0, std::memory_order_relaxed);
    for (auto& q : d.shm->ack) q.store(0, std::memory_order_relaxed);
    d.shm->abort_by.store(0, std::memory_order_relaxed);
    d.shm->attached.store(1, std::memory_order_relaxed);
    d.shm->magic.store(SHM_MAGIC, std::memory_order_release);
    // echo every peer's nonce (in whatever order they arrive)
    int waiting = world - 1;
    std::vector<char> seen((size_t)world, 0);
    for (unsigned spin = 0; waiting > 0; spin++) {
      for (int r = 1; r < world; r++) {
        if (seen[(size_t)r]) continue;
        const uint64_t v = d.shm->hello[r].load(std::memory_order_acquire);
        if (v != 0) {
          d.shm->ack[r].store(v, std::memory_order_release);
          seen[(size_t)r] = 1;
          waiting--;
        }
      }
      if (waiting > 0) {
        if (elapsed() > shm_timeout()) {
          shm_detach(d);
          GM_CHECK(false, GM_ESTATE, "gm_dist_init_shm: %d of %d peers never attached to %s", waiting, world - 1, name);
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
    }
    return GM_OK;
  }
  uint64_t nonce = 0;
  {
    const int rfd = open("/dev/urandom", O_RDONLY);
    if (rfd >= 0) {
      if (read(rfd, &nonce, sizeof nonce) != (ssize_t)sizeof nonce) nonce = 0;
      close(rfd);
    }
    if (nonce == 0) nonce = (uint64_t)Clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ull ^ ((uint64_t)getpid() << 32) ^ (uint64_t)rank ^ 1;
    if (nonce == 0) nonce = 1;
  }
  for (;;) {
    GM_CHECK(elapsed() < 120.0, GM_ESTATE, "gm_dist_init_shm: rank 0 never created %s", name);
    int fd = shm_open(name, O_RDWR, 0600);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < total) {
      if (fd >= 0) close(fd);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      continue;
    }
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    GM_CHECK(p != MAP_FAILED, GM_ENOMEM, "gm_dist_init_shm: mmap failed: %s", strerror(errno));
    ShmHeader* h = static_cast<ShmHeader*>(p);
    // wait for rank 0 to finish the header of THIS inode; give up on it as soon as it is poisoned or the name moves on
    bool good = false;
    for (unsigned spin = 0; elapsed() < 120.0; spin++) {
      const uint64_t m = h->magic.load(std::memory_order_acquire);
      if (m == SHM_MAGIC) {
        good = true;
        break;
      }
      if (m == SHM_DEAD) break;
      if ((spin & 255) == 255 && !shm_name_is(name, st.st_dev, st.st_ino)) break;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (good && (h->world != (uint64_t)world || h->slot_bytes != slot_bytes)) {
      // another world / slot size: the stale segment of a crashed run under a reused name -- unless a LIVE rank 0 really disagrees.  Only
      // a live one echoes this rank's nonce (the hello / ack slots do not depend on the geometry), so ask: an ack within two seconds is a
      // genuine mismatch (GM_EINVAL now); silence means stale -- keep re-attaching until the attach deadline, however late this run's rank 0 is
      // (gm_init skew across 8 GPUs exceeded the former 5 s window)
      if ((size_t)st.st_size >= sizeof(ShmHeader) && rank < 64) {
        h->hello[rank].store(nonce, std::memory_order_release);
        const double t_ask = elapsed();
        bool live = false;
        while (elapsed() - t_ask < 2.0 && h->magic.load(std::memory_order_acquire) == SHM_MAGIC) {
          if (h->ack[rank].load(std::memory_order_acquire) == nonce) {
            live = true;
            break;
          }
          std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (live) {
          const unsigned long long w2 = (unsigned long long)h->world, s2 = (unsigned long long)h->slot_bytes;
          munmap(p, total);
          GM_CHECK(false, GM_EINVAL, "gm_dist_init_shm: %s belongs to a live run of %llu ranks with %llu-byte slots (this rank: %d ranks, %zu bytes)", name, w2, s2, world,
                   slot_bytes);
        }
      }
      good = false;
    }
    if (good && !shm_name_is(name, st.st_dev, st.st_ino)) good = false;  // replaced while this rank was looking at it
    if (good) {  // is the rank 0 behind this segment alive, i.e. of this run?
      h->hello[rank].store(nonce, std::memory_order_release);
      good = false;
      for (unsigned spin = 0; elapsed() < 120.0; spin++) {
        if (h->ack[rank].load(std::memory_order_acquire) == nonce) {
          good = true;
          break;
        }
        if (h->magic.load(std::memory_order_acquire) != SHM_MAGIC) break;
        if ((spin & 255) == 255 && !shm_name_is(name, st.st_dev, st.st_ino)) break;
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
    }
    if (!good) {
      munmap(p, total);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      continue;
    }
    d.shm = h;
    d.shm_ino = st.st_ino;
    d.shm_dev = st.st_dev;
    d.shm->attached.fetch_add(1);
    return GM_OK;
  }
}

// rendezvous on a fresh segment: everybody attached before anybody may finalize (and unlink).  A peer that attached to a stale
// segment in the window before rank 0 poisoned it fails this all-gather (poisoned magic, or counters that are not 0 / 1 / 2) and
// attaches again.
int shm_rendezvous(Dist& d, int rank, int world, const char* name, size_t slot_bytes) {
  const auto t0 = Clock::now();
  for (;;) {
    int rc = shm_open_segment(d, rank, world, name, slot_bytes);
    if (rc) return rc;
    const int save_rank = d.rank, save_world = d.world;
    d.rank = rank;
    d.world = world;
    uint64_t one = 1;
    std::vector<uint64_t> all((size_t)world);
    rc = shm_allgather(d, &one, sizeof one, all.data());
    if (rc == GM_OK) return GM_OK;
    d.rank = save_rank;
    d.world = save_world;
    const bool replaced = rank != 0 && (d.shm->magic.load() != SHM_MAGIC || !shm_name_is(name, d.shm_dev, d.shm_ino));
    shm_detach(d, /*count=*/!replaced);
    if (!replaced || std::chrono::duration<double>(Clock::now() - t0).count() > 120.0) return rc;
  }
}

}  // namespace

extern "C" {

int gm_dist_init_hook(int rank, int world, gm_allgather_fn fn, void* ctx) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CHECK(world >= 1 && rank >= 0 && rank < world, GM_EINVAL, "gm_dist_init_hook: rank %d of %d", rank, world);
  GM_CHECK(fn != nullptr || world == 1, GM_EINVAL, "gm_dist_init_hook: no all-gather function for %d ranks", world);
  reset(d);
  d.rank = rank;
  d.world = world;
  d.fn = fn;
  d.fn_ctx = ctx;
  d.tr = world > 1 ? T_HOOK : T_NONE;
  return GM_OK;
}

int gm_dist_rccl_unique_id(uint8_t out[128]) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CHECK(out != nullptr, GM_EINVAL, "gm_dist_rccl_unique_id: null pointer");
  int rc = load_rccl(d.R);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the ABI of include/gemini_hip.h");
  ncclUniqueId id;
  GM_NCCL(d, d.R.GetUniqueId(&id));
  memcpy(out, &id, sizeof id);
  return GM_OK;
}

int gm_dist_init_rccl(int rank, int world, const uint8_t unique_id[128]) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  gm::Context* C = gm::context();
  GM_CHECK(C != nullptr, GM_ENOTINIT, "gm_init has not been called");  // the communicator binds to the device of gm_init
  GM_CHECK(world >= 1 && rank >= 0 && rank < world && unique_id != nullptr, GM_EINVAL, "gm_dist_init_rccl: rank %d of %d", rank, world);
  int rc = load_rccl(d.R);
  if (rc) return rc;
  reset(d);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  GM_HIP(hipSetDevice(C->device));
  GM_NCCL(d, d.R.CommInitRank(&d.comm, world, id, rank));
  d.rank = rank;
  d.world = world;
  d.tr = T_RCCL;
  return GM_OK;
}

int gm_dist_init_shm(int rank, int world, const char* name, size_t slot_bytes) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CHECK(world >= 1 && world <= 64 && rank >= 0 && rank < world && name && name[0] == '/', GM_EINVAL,
           "gm_dist_init_shm: rank %d of %d (<= 64), name must start with '/'", rank, world);
  reset(d);
  int rc = shm_rendezvous(d, rank, world, name, slot_bytes);
  if (rc) return rc;
  d.rank = rank;
  d.world = world;
  d.tr = T_SHM;
  return GM_OK;
}

// RCCL on ONE node with no out-of-band channel of the embedder's: the ranks meet in a shared-memory segment `name` (as for
// gm_dist_init_shm), rank 0 draws the unique id and the segment carries its 128 bytes to the peers; then the communicator is
// built.  The segment STAYS OPEN as the side channel of the small host payloads (allgather_host_locked).  One call per rank after
// gm_init(local_rank).
int gm_dist_init_rccl_node(int rank, int world, const char* name) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  gm::Context* C = gm::context();
  GM_CHECK(C != nullptr, GM_ENOTINIT, "gm_init has not been called");  // the communicator binds to the device of gm_init
  GM_CHECK(world >= 1 && world <= 64 && rank >= 0 && rank < world && name && name[0] == '/', GM_EINVAL,
           "gm_dist_init_rccl_node: rank %d of %d (<= 64), name must start with '/'", rank, world);
  int rc = load_rccl(d.R);
  if (rc) return rc;
  reset(d);
  if ((rc = shm_rendezvous(d, rank, world, name, (size_t)128 << 10))) return rc;
  d.rank = rank;
  d.world = world;
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the ABI of include/gemini_hip.h");
  if (rank == 0) {
    ncclResult_t r = d.R.GetUniqueId(&id);
    if (r != ncclSuccess) {
      gm::set_error("gm_dist: ncclGetUniqueId failed: %s", d.R.GetErrorString(r));
      memset(&id, 0xff, sizeof id);  // the peers must not wait for an id that will not come: all-ones = "rank 0 failed"
      rc = GM_EHIP;
    }
  }
  std::vector<uint8_t> all((size_t)world * 128);
  int rc2 = shm_allgather(d, &id, 128, all.data());
  if (rc == GM_OK) rc = rc2;
  if (rc == GM_OK) {
    bool bad = true;
    for (int i = 0; i < 128; i++) bad = bad && all[(size_t)i] == 0xff;
    if (bad) {
      gm::set_error("gm_dist_init_rccl_node: rank 0 could not draw a unique id");
      rc = GM_EHIP;
    }
  }
  if (rc) {
    reset(d);
    return rc;
  }
  memcpy(&id, all.data(), sizeof id);
  hipError_t e = hipSetDevice(C->device);
  ncclResult_t r = e == hipSuccess ? d.R.CommInitRank(&d.comm, world, id, rank) : ncclSuccess;
  if (e != hipSuccess || r != ncclSuccess) {
    if (e != hipSuccess) (void)gm::hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
    else gm::set_error("gm_dist: ncclCommInitRank failed: %s", d.R.GetErrorString(r));
    reset(d);
    return GM_EHIP;
  }
  d.tr = T_RCCL;
  // field values over the segment; partial G1 points over RCCL unless the embedder asks otherwise
  d.shm_classes = 1u;
  if (const char* g = getenv("GM_DIST_G1_ROUTE")) d.shm_classes = strcmp(g, "shm") == 0 ? 3u : 1u;
  if (const char* f = getenv("GM_DIST_FIELD_ROUTE")) d.shm_classes = (d.shm_classes & 2u) | (strcmp(f, "rccl") == 0 ? 0u : 1u);
  return GM_OK;
}

int gm_dist_finalize(void) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  reset(d);
  return GM_OK;
}

// A prover that FAILS on this rank outside a collective (an allocation, a bad input, a device error) would leave its peers waiting in their next
// all-gather until a timeout (300 s on the segment, 120 s under RCCL).  gm_dist_abort tells them: the flag in the node's segment makes every wait on it --
// and every bounded wait on an RCCL collective of a communicator that keeps the segment as its side channel -- return GM_ESTATE at once; the
// communicator is aborted and the transport of THIS rank refuses everything until it is initialised again.  The sharded provers call it on every
// failure; a hook transport has no channel for it (the embedder's collective must have its own failure detection).
int gm_dist_abort(void) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  if (d.world <= 1 || d.tr == T_NONE) return GM_OK;
  if (d.shm) d.shm->abort_by.store(1 + (uint64_t)d.rank, std::memory_order_release);
  poison(d);
  return GM_OK;
}

int gm_dist_info(int* rank, int* world, int* transport) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  if (rank) *rank = d.rank;
  if (world) *world = d.world;
  if (transport) *transport = (int)d.tr;
  return GM_OK;
}

int gm_dist_stats(uint64_t* calls, uint64_t* bytes, double* seconds, int reset_counters) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  if (calls) *calls = d.calls;
  if (bytes) *bytes = d.bytes;
  if (seconds) *seconds = d.seconds;
  if (reset_counters) {
    d.calls = d.bytes = 0;
    d.seconds = 0.0;
    for (int i = 0; i < R_NROUTES; i++) {
      d.rcalls[i] = d.rbytes[i] = 0;
      d.rseconds[i] = 0.0;
    }
  }
  return GM_OK;
}

// the same, split by the route each collective took: 0 copy (world 1), 1 hook, 2 RCCL with host staging, 3 RCCL device vectors,
// 4 shared memory (the transport of gm_dist_init_shm or the side segment of gm_dist_init_rccl_node)
int gm_dist_stats_routes(uint64_t calls[5], uint64_t bytes[5], double seconds[5]) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  for (int i = 0; i < R_NROUTES; i++) {
    if (calls) calls[i] = d.rcalls[i];
    if (bytes) bytes[i] = d.rbytes[i];
    if (seconds) seconds[i] = d.rseconds[i];
  }
  return GM_OK;
}

int gm_dist_allgather_host_class(const void* send, size_t bytes, void* recv, int payload_class) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CHECK(bytes == 0 || (send && recv), GM_EINVAL, "gm_dist_allgather_host_class: null pointer");
  GM_CHECK(payload_class == GM_DIST_CLASS_FIELD || payload_class == GM_DIST_CLASS_G1, GM_EINVAL, "gm_dist_allgather_host_class: class %d", payload_class);
  return allgather_host_locked(d, send, bytes, recv, payload_class);
}

// `iters` back-to-back all-gathers of `bytes` per rank over one route (-1: the route the class would take; 2: RCCL with host
// staging; 4: shared memory) -> microseconds per call, every rank.  What decided the routing (profiles/r5_collective_latency.txt).
int gm_dist_bench(size_t bytes, int iters, int payload_class, int route, double* usec_per_call) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CHECK(bytes > 0 && iters > 0 && usec_per_call, GM_EINVAL, "gm_dist_bench: bad arguments");
  GM_CHECK(route == -1 || route == R_RCCL_HOST || route == R_SHM, GM_EINVAL, "gm_dist_bench: route %d (want -1, 2 or 4)", route);
  GM_CHECK(route == -1 || d.tr == T_RCCL, GM_ESTATE, "gm_dist_bench: forcing a route needs the RCCL transport (gm_dist_init_rccl_node)");
  GM_CHECK(route != R_SHM || d.shm != nullptr, GM_ESTATE, "gm_dist_bench: no side segment (gm_dist_init_rccl opens none)");
  std::vector<uint8_t> send(bytes, (uint8_t)(d.rank + 1)), recv(bytes * (size_t)d.world);
  int rc = GM_OK;
  for (int i = 0; i < 3 && !rc; i++) rc = allgather_host_locked(d, send.data(), bytes, recv.data(), payload_class, route);
  const auto t0 = Clock::now();
  for (int i = 0; i < iters && !rc; i++) rc = allgather_host_locked(d, send.data(), bytes, recv.data(), payload_class, route);
  *usec_per_call = std::chrono::duration<double>(Clock::now() - t0).count() * 1e6 / iters;
  return rc;
}

int gm_dist_allgather_host(const void* send, size_t bytes, void* recv) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CHECK(bytes == 0 || (send && recv), GM_EINVAL, "gm_dist_allgather_host: null pointer");
  return allgather_host_locked(d, send, bytes, recv);
}

// out = the local vectors of ranks 0 .. world - 1 back to back (equal lengths on every rank; out is resized)
int gm_dist_allgather_vec(uint64_t local_vec, uint64_t out_vec) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CTX();
  GM_FR_LOCK(C);
  GM_DIST_ALIVE(d);
  gm::FrVec* in = gm::find_vec(local_vec);
  gm::FrVec* out = gm::find_vec(out_vec);
  GM_CHECK(in && out && in != out, GM_EHANDLE, "gm_dist_allgather_vec: unknown or aliased vector handle");
  const size_t n = in->len, bytes = n * 32;
  GM_CHECK(out->cap >= n * (size_t)d.world, GM_EINVAL, "gm_dist_allgather_vec: output capacity %zu < %zu x %d", out->cap, n, d.world);
  out->len = n * (size_t)d.world;
  if (n == 0) return GM_OK;
  const auto t0 = Clock::now();
  if (d.tr == T_NONE || (d.world == 1 && d.tr != T_RCCL)) {
    GM_HIP(hipMemcpyAsync(out->d, in->d, bytes, hipMemcpyDeviceToDevice, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
  } else if (d.tr == T_RCCL) {
    // (a failing rank aborts the communicator so that its peers return instead of hanging: see rccl_host_allgather)
    ncclResult_t r = d.R.AllGather(in->d, out->d, bytes, ncclChar, d.comm, C->stream);
    bool late = false;
    hipError_t e = r == ncclSuccess ? (d.world > 1 ? stream_wait_bounded(C->stream, &late, d.shm ? &d.shm->abort_by : nullptr) : hipStreamSynchronize(C->stream)) : hipSuccess;
    if (r != ncclSuccess || e != hipSuccess) {
      if (r != ncclSuccess) gm::set_error("gm_dist: ncclAllGather failed: %s", d.R.GetErrorString(r));
      if (late) gm::set_error("gm_dist: ncclAllGather of a vector did not complete in time: a peer never arrived");
      const int code = r != ncclSuccess ? GM_EHIP : (late ? GM_ESTATE : gm::hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
      poison(d);
      return code;
    }
  } else {
    std::vector<uint8_t> h_in(bytes), h_out(bytes * (size_t)d.world);
    GM_HIP(hipMemcpyAsync(h_in.data(), in->d, bytes, hipMemcpyDeviceToHost, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
    int rc = d.tr == T_HOOK ? d.fn(d.fn_ctx, h_in.data(), bytes, h_out.data()) : shm_allgather(d, h_in.data(), bytes, h_out.data());
    if (rc) {
      if (d.tr == T_HOOK) gm::set_error("gm_dist: the all-gather hook returned %d", rc);
      return rc;
    }
    GM_HIP(hipMemcpyAsync(out->d, h_out.data(), h_out.size(), hipMemcpyHostToDevice, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
  }
  d.note(d.tr == T_NONE || (d.world == 1 && d.tr != T_RCCL) ? R_COPY : d.tr == T_RCCL ? R_RCCL_VEC : d.tr == T_HOOK ? R_HOOK : R_SHM, bytes * (size_t)d.world,
         std::chrono::duration<double>(Clock::now() - t0).count());
  return GM_OK;
}

// RE-BLOCKING.  k device vectors, each block-distributed with equal blocks (rank p holds elements [p b, (p + 1) b) of a global
// vector of world x b elements, b = the local length); out j = the elements [rank B, (rank + 1) B) of global vector j that exist
// (length 0 when the range is past its end).  What the n / g opening of the block-sharded prover needs: level i of the folding
// tree is sharded in blocks of m / 2^i, the opened polynomial in blocks of m.  Over RCCL: ONE group of ncclSend / ncclRecv, every
// element crosses one xGMI link once (all-gathers of whole vectors when librccl lacks the point-to-point calls); over the host
// transports (shm, hook: the shared-GPU tests) the blocks are staged through host memory and all-gathered.
int gm_dist_reblock_vecs(const uint64_t* local_vecs, size_t k, size_t new_block, const uint64_t* out_vecs) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  GM_CTX();
  GM_FR_LOCK(C);
  GM_DIST_ALIVE(d);
  GM_CHECK((local_vecs && out_vecs) || k == 0, GM_EINVAL, "gm_dist_reblock_vecs: null pointer");
  GM_CHECK(new_block > 0, GM_EINVAL, "gm_dist_reblock_vecs: empty blocks");
  const size_t g = (size_t)d.world, r = (size_t)d.rank, B = new_block;
  std::vector<gm::FrVec*> in(k), out(k);
  for (size_t j = 0; j < k; j++) {
    in[j] = gm::find_vec(local_vecs[j]);
    out[j] = gm::find_vec(out_vecs[j]);
    GM_CHECK(in[j] && out[j] && in[j] != out[j], GM_EHANDLE, "gm_dist_reblock_vecs: unknown or aliased vector handle (%zu)", j);
    const size_t total = g * in[j]->len, lo = r * B, len = lo < total ? std::min(B, total - lo) : 0;
    GM_CHECK(out[j]->cap >= len, GM_EINVAL, "gm_dist_reblock_vecs: output %zu holds %zu elements, the block has %zu", j, out[j]->cap, len);
    out[j]->len = len;
  }
  if (k == 0) return GM_OK;
  if (g > 1) {
    // the send / recv sizes below assume rank p's block of vector j has MY length: a shorter block somewhere means mismatched sizes
    // (a hang or silent corruption).  One small all-gather of the k local lengths first
    std::vector<uint64_t> mine(k), all(k * g);
    for (size_t j = 0; j < k; j++) mine[j] = in[j]->len;
    int rcl = allgather_host_locked(d, mine.data(), 8 * k, all.data());
    if (rcl) return rcl;
    for (size_t p = 0; p < g; p++)
      for (size_t j = 0; j < k; j++)
        GM_CHECK(all[p * k + j] == mine[j], GM_EINVAL, "gm_dist_reblock_vecs: rank %zu holds %llu elements of vector %zu, this rank %llu (equal blocks are required)", p,
                 (unsigned long long)all[p * k + j], j, (unsigned long long)mine[j]);
  }
  const auto t0 = Clock::now();
  size_t moved = 0;
  const bool p2p = d.tr == T_RCCL && d.R.Send && d.R.Recv && d.R.GroupStart && d.R.GroupEnd;
  Route route = R_COPY;
  if (d.tr == T_NONE || g == 1 || p2p) {  // (one rank: device copies whatever the transport -- nothing to stage through the host)
    route = p2p ? R_RCCL_VEC : R_COPY;
    bool grouped = false;
    auto fail = [&](ncclResult_t res) {
      gm::set_error("gm_dist_reblock_vecs: RCCL point-to-point failed: %s", d.R.GetErrorString(res));
      poison(d);
      return GM_EHIP;
    };
    if (p2p && g > 1) {
      ncclResult_t res = d.R.GroupStart();
      if (res != ncclSuccess) return fail(res);
      grouped = true;
    }
    for (size_t j = 0; j < k; j++) {
      const size_t b = in[j]->len, total = g * b;
      for (size_t p = 0; p < g; p++) {
        // what rank p wants of my block
        size_t lo = std::max(p * B, r * b), hi = std::min(std::min((p + 1) * B, (r + 1) * b), total);
        if (lo < hi) {
          uint8_t* src = in[j]->d + (lo - r * b) * 32;
          if (p == r) {
            const hipError_t ce = hipMemcpyAsync(out[j]->d + (lo - r * B) * 32, src, (hi - lo) * 32, hipMemcpyDeviceToDevice, C->stream);
            if (ce != hipSuccess) {  // never leave a group open: the peers would wait out their timeout inside it
              if (grouped) (void)d.R.GroupEnd();
              if (g > 1) poison(d);
              return gm::hip_fail(ce, "hipMemcpyAsync(re-block, own part)", __FILE__, __LINE__);
            }
          } else {
            ncclResult_t res = d.R.Send(src, (hi - lo) * 32, ncclChar, (int)p, d.comm, C->stream);
            if (res != ncclSuccess) return fail(res);
          }
        }
        if (p == r) continue;
        // what I want of rank p's block
        lo = std::max(r * B, p * b);
        hi = std::min(std::min((r + 1) * B, (p + 1) * b), total);
        if (lo < hi) {
          ncclResult_t res = d.R.Recv(out[j]->d + (lo - r * B) * 32, (hi - lo) * 32, ncclChar, (int)p, d.comm, C->stream);
          if (res != ncclSuccess) return fail(res);
          moved += (hi - lo) * 32;
        }
      }
    }
    if (grouped) {
      ncclResult_t res = d.R.GroupEnd();
      if (res != ncclSuccess) return fail(res);
      bool late = false;
      const hipError_t e = stream_wait_bounded(C->stream, &late, d.shm ? &d.shm->abort_by : nullptr);
      if (late) {
        gm::set_error("gm_dist_reblock_vecs: the send / recv group did not complete in time: a peer never arrived");
        poison(d);
        return GM_ESTATE;
      }
      GM_HIP(e);
    } else {
      GM_HIP(hipStreamSynchronize(C->stream));
    }
  } else {
    // whole vectors: ncclAllGather into a temporary (RCCL without send / recv) or the host transports
    route = d.tr == T_RCCL ? R_RCCL_VEC : d.tr == T_HOOK ? R_HOOK : R_SHM;
    for (size_t j = 0; j < k; j++) {
      const size_t b = in[j]->len, bytes = b * 32, len = out[j]->len;
      if (b == 0) continue;
      if (d.tr == T_RCCL) {
        void* tmp = nullptr;
        GM_HIP(gm::dev_malloc(&tmp, bytes * g));
        ncclResult_t res = d.R.AllGather(in[j]->d, tmp, bytes, ncclChar, d.comm, C->stream);
        hipError_t e = res == ncclSuccess && len ? hipMemcpyAsync(out[j]->d, static_cast<uint8_t*>(tmp) + r * B * 32, len * 32, hipMemcpyDeviceToDevice, C->stream) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(C->stream);
        (void)gm::raw_free(tmp);
        GM_CHECK(res == ncclSuccess, GM_EHIP, "gm_dist_reblock_vecs: ncclAllGather failed: %s", d.R.GetErrorString(res));
        GM_HIP(e);
      } else {
        std::vector<uint8_t> h_in(bytes), h_all(bytes * g);
        GM_HIP(hipMemcpyAsync(h_in.data(), in[j]->d, bytes, hipMemcpyDeviceToHost, C->stream));
        GM_HIP(hipStreamSynchronize(C->stream));
        int rc = d.tr == T_HOOK ? d.fn(d.fn_ctx, h_in.data(), bytes, h_all.data()) : shm_allgather(d, h_in.data(), bytes, h_all.data());
        if (rc) {
          if (d.tr == T_HOOK) gm::set_error("gm_dist: the all-gather hook returned %d", rc);
          return rc;
        }
        if (len) {
          GM_HIP(hipMemcpyAsync(out[j]->d, h_all.data() + r * B * 32, len * 32, hipMemcpyHostToDevice, C->stream));
          GM_HIP(hipStreamSynchronize(C->stream));
        }
      }
      moved += bytes * g;
    }
  }
  d.note(route, moved, std::chrono::duration<double>(Clock::now() - t0).count());
  return GM_OK;
}

// Every rank sends patterns of several sizes and checks what it receives from every peer: run at start-up on a multi-GPU
// node.  With no transport initialised and a GPU context present it opens a ONE-rank RCCL communicator and pushes a 144-byte
// point through ncclAllGather (the binding, the staging and the stream ordering are then exercised on a single-GPU box too).
static int selftest_reblock();
static int selftest_host_payloads();
int gm_dist_selftest(void) {
  int rc = selftest_host_payloads();
  if (rc) return rc;
  int world = 1;
  (void)gm_dist_info(nullptr, &world, nullptr);
  return world > 1 ? selftest_reblock() : GM_OK;
}
static int selftest_host_payloads() {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  bool temp = false;
  if (d.tr == T_NONE && gm::context() != nullptr) {
    int rc = load_rccl(d.R);
    if (rc) return rc;
    ncclUniqueId id;
    GM_NCCL(d, d.R.GetUniqueId(&id));
    GM_HIP(hipSetDevice(gm::context()->device));
    GM_NCCL(d, d.R.CommInitRank(&d.comm, 1, id, 0));
    d.tr = T_RCCL;
    temp = true;
  }
  int rc = GM_OK;
  const size_t sizes[] = {8, 64, 144, 1000, 4096, 65536 + 24};
  for (size_t sz : sizes) {
    std::vector<uint8_t> send(sz), recv(sz * (size_t)d.world);
    for (size_t i = 0; i < sz; i++) send[i] = (uint8_t)(i * 131 + (size_t)d.rank * 17 + sz);
    if ((rc = allgather_host_locked(d, send.data(), sz, recv.data()))) break;
    for (int r = 0; r < d.world && !rc; r++)
      for (size_t i = 0; i < sz; i++)
        if (recv[(size_t)r * sz + i] != (uint8_t)(i * 131 + (size_t)r * 17 + sz)) {
          gm::set_error("gm_dist_selftest: rank %d received a wrong byte %zu of rank %d's %zu-byte payload", d.rank, i, r, sz);
          rc = GM_ESTATE;
          break;
        }
    if (rc) break;
  }
  if (temp) reset(d);
  return rc;
}

// the device half of the self-test (called by gm_dist_selftest after the host payloads, outside its lock): every rank holds a block
// of 8 known elements, re-blocked to blocks of 16 and of 4 -- the grouped ncclSend / ncclRecv of gm_dist_reblock_vecs with real peers
static int selftest_reblock() {
  if (gm::context() == nullptr) return GM_OK;
  int rank = 0, world = 1;
  int rc = gm_dist_info(&rank, &world, nullptr);
  if (rc) return rc;
  const size_t b = 8;
  auto value = [](size_t global, int limb) { return (uint64_t)global * 0x9e3779b97f4a7c15ull + (uint64_t)limb; };
  std::vector<uint64_t> host(4 * b);
  for (size_t i = 0; i < b; i++)
    for (int l = 0; l < 4; l++) host[4 * i + l] = value((size_t)rank * b + i, l) >> 3;  // below 2^61: any limb pattern is fine for a copy
  uint64_t in = 0, out = 0;
  if ((rc = gm_fr_vec_alloc(b, &in))) return rc;
  if ((rc = gm_fr_vec_alloc(16, &out))) {
    (void)gm_fr_vec_free(in);
    return rc;
  }
  rc = gm_fr_vec_upload(in, 0, host.data(), b);
  for (size_t B : {(size_t)16, (size_t)4}) {
    if (rc) break;
    if ((rc = gm_dist_reblock_vecs(&in, 1, B, &out))) break;
    size_t len = 0;
    if ((rc = gm_fr_vec_len(out, &len))) break;
    const size_t total = (size_t)world * b, lo = (size_t)rank * B, want = lo < total ? (B < total - lo ? B : total - lo) : 0;
    if (len != want) {
      gm::set_error("gm_dist_selftest: re-blocking to %zu gave rank %d %zu elements, expected %zu", B, rank, len, want);
      rc = GM_ESTATE;
      break;
    }
    std::vector<uint64_t> got(4 * (len ? len : 1));
    if (len && (rc = gm_fr_vec_download(out, 0, got.data(), len))) break;
    for (size_t i = 0; i < len && !rc; i++)
      for (int l = 0; l < 4; l++)
        if (got[4 * i + l] != value(lo + i, l) >> 3) {
          gm::set_error("gm_dist_selftest: rank %d received a wrong element %zu after re-blocking to %zu", rank, i, B);
          rc = GM_ESTATE;
          break;
        }
  }
  (void)gm_fr_vec_free(in);
  (void)gm_fr_vec_free(out);
  return rc;
}

}  // extern "C"
