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

#include <rccl/rccl.h>

#include "ctx.hpp"

namespace {

using Clock = std::chrono::steady_clock;

enum Transport { T_NONE = 0, T_HOOK = 1, T_RCCL = 2, T_SHM = 3 };

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// shared segment: header, then two banks (call parity) of world slots of slot_bytes
struct ShmHeader {
  std::atomic<uint64_t> magic;
  uint64_t world, slot_bytes;
  std::atomic<uint64_t> attached;
  std::atomic<uint64_t> seq[64];  // seq[r] = number of the last call rank r has published
};
constexpr uint64_t SHM_MAGIC = 0x474d44495354ull;  // "GMDIST"

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
  // shm
  ShmHeader* shm = nullptr;
  size_t shm_bytes = 0, slot_bytes = 0;
  uint64_t shm_call = 0;
  std::string shm_name;
  // statistics
  uint64_t calls = 0, bytes = 0;
  double seconds = 0.0;
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

int shm_wait(std::atomic<uint64_t>& a, uint64_t want, const char* what) {
  const auto t0 = Clock::now();
  static const double limit = getenv("GM_DIST_TIMEOUT_S") ? atof(getenv("GM_DIST_TIMEOUT_S")) : 300.0;
  for (unsigned spin = 0;; spin++) {
    if (a.load(std::memory_order_acquire) >= want) return GM_OK;
    if (spin < 2000) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    } else {
      // ranks may outnumber the cores the container may use (8 ranks + their helper threads on a 16-CPU quota)
      std::this_thread::yield();
      if ((spin & 1023) == 0) {
        GM_CHECK(std::chrono::duration<double>(Clock::now() - t0).count() < limit, GM_ESTATE, "gm_dist(shm): %s: a peer did not arrive within %.0f s",
                 what, limit);
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
      int rc = shm_wait(d.shm->seq[r], k, "all-gather");
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

void shm_detach(Dist& d) {
  if (!d.shm) return;
  const uint64_t left = d.shm->attached.fetch_sub(1) - 1;
  munmap(d.shm, d.shm_bytes);
  if (left == 0) shm_unlink(d.shm_name.c_str());
  d.shm = nullptr;
}

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

int allgather_host_locked(Dist& d, const void* send, size_t bytes, void* recv) {
  if (bytes == 0) return GM_OK;
  const auto t0 = Clock::now();
  int rc = GM_OK;
  switch (d.tr) {
    case T_NONE:
      memcpy(recv, send, bytes);
      break;
    case T_HOOK:
      rc = d.fn(d.fn_ctx, send, bytes, recv);
      if (rc) gm::set_error("gm_dist: the all-gather hook returned %d", rc);
      break;
    case T_SHM:
      rc = shm_allgather(d, send, bytes, recv);
      break;
    case T_RCCL: {
      gm::Context* C = gm::context();
      GM_CHECK(C != nullptr, GM_ENOTINIT, "gm_init has not been called");
      if ((rc = ensure_stage(d, bytes))) return rc;
      memcpy(d.h_in, send, bytes);
      GM_HIP(hipMemcpyAsync(d.d_in, d.h_in, bytes, hipMemcpyHostToDevice, C->stream));
      GM_NCCL(d, d.R.AllGather(d.d_in, d.d_out, bytes, ncclChar, d.comm, C->stream));
      GM_HIP(hipMemcpyAsync(d.h_out, d.d_out, bytes * (size_t)d.world, hipMemcpyDeviceToHost, C->stream));
      GM_HIP(hipStreamSynchronize(C->stream));
      memcpy(recv, d.h_out, bytes * (size_t)d.world);
      break;
    }
  }
  d.calls++;
  d.bytes += bytes * (size_t)d.world;
  d.seconds += std::chrono::duration<double>(Clock::now() - t0).count();
  return rc;
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
  if (slot_bytes == 0) slot_bytes = (size_t)1 << 20;
  slot_bytes = (slot_bytes + 63) & ~(size_t)63;
  const size_t total = ((sizeof(ShmHeader) + 63) & ~(size_t)63) + 2 * (size_t)world * slot_bytes;
  int fd = -1;
  if (rank == 0) {
    (void)shm_unlink(name);  // a stale segment of a crashed run
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    GM_CHECK(fd >= 0, GM_ESTATE, "gm_dist_init_shm: shm_open(%s) failed: %s", name, strerror(errno));
    if (ftruncate(fd, (off_t)total) != 0) {
      close(fd);
      GM_CHECK(false, GM_ENOMEM, "gm_dist_init_shm: ftruncate(%zu) failed: %s", total, strerror(errno));
    }
  } else {
    const auto t0 = Clock::now();
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= total) break;
      if (fd >= 0) close(fd);
      fd = -1;
      GM_CHECK(std::chrono::duration<double>(Clock::now() - t0).count() < 120.0, GM_ESTATE, "gm_dist_init_shm: rank 0 never created %s", name);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  GM_CHECK(p != MAP_FAILED, GM_ENOMEM, "gm_dist_init_shm: mmap failed: %s", strerror(errno));
  d.shm = static_cast<ShmHeader*>(p);
  d.shm_bytes = total;
  d.slot_bytes = slot_bytes;
  d.shm_name = name;
  if (rank == 0) {
    d.shm->world = (uint64_t)world;
    d.shm->slot_bytes = slot_bytes;
    for (auto& s : d.shm->seq) s.store(0, std::memory_order_relaxed);
    d.shm->attached.store(1, std::memory_order_relaxed);
    d.shm->magic.store(SHM_MAGIC, std::memory_order_release);
  } else {
    int rc = shm_wait(d.shm->magic, SHM_MAGIC, "attach");
    if (rc == GM_OK && (d.shm->magic.load() != SHM_MAGIC || d.shm->world != (uint64_t)world || d.shm->slot_bytes != slot_bytes)) {
      gm::set_error("gm_dist_init_shm: %s was created for another world / slot size", name);
      rc = GM_EINVAL;
    }
    if (rc) {
      munmap(p, total);
      d.shm = nullptr;
      return rc;
    }
    d.shm->attached.fetch_add(1);
  }
  d.rank = rank;
  d.world = world;
  d.tr = T_SHM;
  // everybody attached before anybody may finalize (and unlink)
  uint64_t one = 1;
  std::vector<uint64_t> all((size_t)world);
  return shm_allgather(d, &one, sizeof one, all.data());
}

// RCCL on ONE node with no out-of-band channel of the embedder's: the ranks meet in a shared-memory segment `name` (as for
// gm_dist_init_shm), rank 0 draws the unique id and the segment carries its 128 bytes to the peers; then the segment is left and the
// communicator is built.  One call per rank after gm_init(local_rank).
int gm_dist_init_rccl_node(int rank, int world, const char* name) {
  int rc = gm_dist_init_shm(rank, world, name, 4096);
  if (rc) return rc;
  uint8_t mine[128];
  memset(mine, 0, sizeof mine);
  if (rank == 0 && (rc = gm_dist_rccl_unique_id(mine))) {
    (void)gm_dist_finalize();
    return rc;
  }
  std::vector<uint8_t> all((size_t)world * 128);
  rc = gm_dist_allgather_host(mine, 128, all.data());
  if (rc) {
    (void)gm_dist_finalize();
    return rc;
  }
  // everybody has read rank 0's slot before anybody leaves the segment (the last one out unlinks it)
  uint64_t one = 1;
  std::vector<uint64_t> seen((size_t)world);
  rc = gm_dist_allgather_host(&one, 8, seen.data());
  (void)gm_dist_finalize();
  if (rc) return rc;
  return gm_dist_init_rccl(rank, world, all.data());
}

int gm_dist_finalize(void) {
  Dist& d = D();
  std::lock_guard<std::mutex> lk(d.mu);
  reset(d);
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
  }
  return GM_OK;
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
  gm::FrVec* in = gm::find_vec(local_vec);
  gm::FrVec* out = gm::find_vec(out_vec);
  GM_CHECK(in && out && in != out, GM_EHANDLE, "gm_dist_allgather_vec: unknown or aliased vector handle");
  const size_t n = in->len, bytes = n * 32;
  GM_CHECK(out->cap >= n * (size_t)d.world, GM_EINVAL, "gm_dist_allgather_vec: output capacity %zu < %zu x %d", out->cap, n, d.world);
  out->len = n * (size_t)d.world;
  if (n == 0) return GM_OK;
  const auto t0 = Clock::now();
  if (d.tr == T_NONE) {
    GM_HIP(hipMemcpyAsync(out->d, in->d, bytes, hipMemcpyDeviceToDevice, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
  } else if (d.tr == T_RCCL) {
    GM_NCCL(d, d.R.AllGather(in->d, out->d, bytes, ncclChar, d.comm, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
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
  d.calls++;
  d.bytes += bytes * (size_t)d.world;
  d.seconds += std::chrono::duration<double>(Clock::now() - t0).count();
  return GM_OK;
}

// Every rank sends patterns of several sizes and checks what it receives from every peer: run at start-up on a multi-GPU
// node.  With no transport initialised and a GPU context present it opens a ONE-rank RCCL communicator and pushes a 144-byte
// point through ncclAllGather (the binding, the staging and the stream ordering are then exercised on a single-GPU box too).
int gm_dist_selftest(void) {
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

}  // extern "C"
