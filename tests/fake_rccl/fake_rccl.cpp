// TEST INFRASTRUCTURE, not product: a stand-in for librccl.so for processes that SHARE ONE GPU.
//
// RCCL refuses two ranks on one device, and no multi-GPU node is available to this repository's tests, so the N-rank branches of
// gemini_amd/csrc/dist.cpp (ncclAllGather with world > 1 and its displacement arithmetic, the grouped ncclSend / ncclRecv of
// gm_dist_reblock_vecs, the abort path of a failing rank) would never execute.  This library implements the ten entry points dist.cpp
// binds -- selected with GM_RCCL_LIB, the library's own override -- over POSIX shared memory and host-staged copies:
//   ncclAllGather       sync the stream, D2H into this rank's outbox, barrier, H2D of every outbox into recvbuff, barrier
//   ncclGroupStart/End  sends and recvs are queued; at GroupEnd every rank publishes a directory + payloads, barrier, picks the messages
//                       addressed to it (matched in order per peer, sizes checked), barrier
//   ncclCommAbort       raises a flag in the control segment: every rank waiting in a barrier returns ncclSystemError
// Semantics are those a caller of RCCL may rely on: stream order (the call returns with the data in place, which is stronger),
// rank-order all-gather, point-to-point matching by order.  It is SLOW (host staging) and says nothing about performance.
// Fault injection: GM_FAKE_RCCL_FAIL_AT=<rank>:<n> makes that rank's n-th collective return ncclSystemError before it communicates.
// The HIP runtime must be the one libgemini_hip.so is bound to (its streams and device pointers come in): dist.cpp exports its path
// in GM_HIP_RUNTIME before it loads librccl; nothing is linked.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

using Clock = std::chrono::steady_clock;
enum { kSuccess = 0, kCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

// the three HIP calls, bound at run time to the runtime the product library uses
using StreamSync_t = int (*)(void*);
using Memcpy_t = int (*)(void*, const void*, size_t, int);
StreamSync_t hipStreamSynchronize_p = nullptr;
Memcpy_t hipMemcpy_p = nullptr;
constexpr int kH2D = 1, kD2H = 2;  // hipMemcpyHostToDevice / hipMemcpyDeviceToHost

bool bind_hip() {
  if (hipMemcpy_p) return true;
  const char* path = getenv("GM_HIP_RUNTIME");
  void* lib = path ? dlopen(path, RTLD_NOW | RTLD_NOLOAD) : nullptr;
  if (!lib) lib = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) return false;
  hipStreamSynchronize_p = reinterpret_cast<StreamSync_t>(dlsym(lib, "hipStreamSynchronize"));
  hipMemcpy_p = reinterpret_cast<Memcpy_t>(dlsym(lib, "hipMemcpy"));
  return hipStreamSynchronize_p && hipMemcpy_p;
}

struct Ctl {
  std::atomic<uint64_t> magic;
  std::atomic<uint64_t> world;
  std::atomic<uint64_t> count, gen;  // central barrier
  std::atomic<uint64_t> abort_flag;
  std::atomic<uint64_t> box_bytes[64];
};
constexpr uint64_t MAGIC = 0x46414b4552434cull;

struct Box {
  int fd = -1;
  uint8_t* p = nullptr;
  size_t mapped = 0;
};

struct FakeComm {
  int rank = 0, world = 1;
  std::string name;
  Ctl* ctl = nullptr;
  std::vector<Box> box;
  uint64_t calls = 0;
  double timeout = 60.0;
  int fail_rank = -1;
  uint64_t fail_at = 0;
};
FakeComm* g_comm = nullptr;  // the one live communicator of the process (a group with no operations still takes part in the exchange)

std::string box_name(const std::string& base, int r) { return base + "." + std::to_string(r); }

int barrier(FakeComm* c) {
  if (c->world == 1) return kSuccess;
  Ctl* h = c->ctl;
  const uint64_t g = h->gen.load(std::memory_order_acquire);
  if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint64_t)c->world) {
    h->count.store(0, std::memory_order_relaxed);
    h->gen.fetch_add(1, std::memory_order_acq_rel);
    return kSuccess;
  }
  const auto t0 = Clock::now();
  for (unsigned spin = 0; h->gen.load(std::memory_order_acquire) == g; spin++) {
    if (h->abort_flag.load(std::memory_order_acquire)) return kSystemError;
    if (spin > 2000) std::this_thread::yield();
    if ((spin & 4095) == 0 && std::chrono::duration<double>(Clock::now() - t0).count() > c->timeout) return kSystemError;
  }
  return h->abort_flag.load(std::memory_order_acquire) ? kSystemError : kSuccess;
}

// my outbox with room for `bytes`
uint8_t* own_box(FakeComm* c, size_t bytes) {
  Box& b = c->box[c->rank];
  if (bytes > b.mapped) {
    const size_t cap = std::max<size_t>((bytes * 2 + 4095) & ~(size_t)4095, 1 << 16);
    if (b.fd < 0) {
      b.fd = shm_open(box_name(c->name, c->rank).c_str(), O_CREAT | O_RDWR, 0600);
      if (b.fd < 0) return nullptr;
    }
    if (ftruncate(b.fd, (off_t)cap) != 0) return nullptr;
    if (b.p) munmap(b.p, b.mapped);
    b.p = static_cast<uint8_t*>(mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, b.fd, 0));
    if (b.p == MAP_FAILED) {
      b.p = nullptr;
      b.mapped = 0;
      return nullptr;
    }
    b.mapped = cap;
  }
  return b.p;
}
// rank p's outbox, at least the `box_bytes[p]` it published
const uint8_t* peer_box(FakeComm* c, int p) {
  const size_t need = (size_t)c->ctl->box_bytes[p].load(std::memory_order_acquire);
  Box& b = c->box[p];
  if (p == c->rank) return b.p;
  if (need > b.mapped || !b.p) {
    if (b.fd < 0) {
      b.fd = shm_open(box_name(c->name, p).c_str(), O_RDWR, 0600);
      if (b.fd < 0) return nullptr;
    }
    struct stat st;
    if (fstat(b.fd, &st) != 0 || (size_t)st.st_size < need) return nullptr;
    if (b.p) munmap(b.p, b.mapped);
    b.p = static_cast<uint8_t*>(mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, b.fd, 0));
    if (b.p == MAP_FAILED) {
      b.p = nullptr;
      b.mapped = 0;
      return nullptr;
    }
    b.mapped = (size_t)st.st_size;
  }
  return b.p;
}

bool injected_failure(FakeComm* c) {
  c->calls++;
  return c->rank == c->fail_rank && c->calls == c->fail_at;
}

void detach(FakeComm* c, bool unlink_ctl) {
  for (int p = 0; p < (int)c->box.size(); p++) {
    if (c->box[p].p) munmap(c->box[p].p, c->box[p].mapped);
    if (c->box[p].fd >= 0) close(c->box[p].fd);
  }
  shm_unlink(box_name(c->name, c->rank).c_str());
  if (c->ctl) munmap(c->ctl, sizeof(Ctl));
  if (unlink_ctl) shm_unlink(c->name.c_str());
  if (g_comm == c) g_comm = nullptr;
  delete c;
}

struct Op {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  void* stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

int run_group(FakeComm* c, std::vector<Op>& ops) {
  if (injected_failure(c)) return kSystemError;
  for (const Op& o : ops)
    if (hipStreamSynchronize_p(o.stream) != 0) return kCudaError;
  size_t nsend = 0, payload = 0;
  for (const Op& o : ops)
    if (o.send) {
      nsend++;
      payload += (o.bytes + 63) & ~(size_t)63;
    }
  const size_t dir = (8 + 24 * nsend + 63) & ~(size_t)63, total = dir + payload;
  uint8_t* mine = own_box(c, total);
  if (!mine) return kSystemError;
  {
    uint64_t* d = reinterpret_cast<uint64_t*>(mine);
    d[0] = nsend;
    size_t at = dir, i = 0;
    for (const Op& o : ops) {
      if (!o.send) continue;
      d[1 + 3 * i] = (uint64_t)o.peer;
      d[2 + 3 * i] = o.bytes;
      d[3 + 3 * i] = at;
      if (o.bytes && hipMemcpy_p(mine + at, o.ptr, o.bytes, kD2H) != 0) return kCudaError;
      at += (o.bytes + 63) & ~(size_t)63;
      i++;
    }
  }
  c->ctl->box_bytes[c->rank].store(total, std::memory_order_release);
  int rc = barrier(c);
  if (rc) return rc;
  for (int p = 0; p < c->world; p++) {
    if (p == c->rank) continue;
    std::vector<const Op*> want;
    for (const Op& o : ops)
      if (!o.send && o.peer == p) want.push_back(&o);
    const uint8_t* theirs = peer_box(c, p);
    if (!theirs) return kSystemError;
    const uint64_t* d = reinterpret_cast<const uint64_t*>(theirs);
    size_t k = 0;
    for (uint64_t i = 0; i < d[0]; i++) {
      if ((int)d[1 + 3 * i] != c->rank) continue;
      if (k >= want.size() || want[k]->bytes != d[2 + 3 * i]) return kInvalidArgument;  // unmatched or mis-sized: real RCCL would hang or corrupt
      if (want[k]->bytes && hipMemcpy_p(want[k]->ptr, theirs + d[3 + 3 * i], want[k]->bytes, kH2D) != 0) return kCudaError;
      k++;
    }
    if (k != want.size()) return kInvalidArgument;
  }
  return barrier(c);
}

}  // namespace

extern "C" {

typedef struct {
  char internal[128];
} ncclUniqueId;
typedef FakeComm* ncclComm_t;

int ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<unsigned> counter{0};
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/gmfakerccl_%d_%u_%llx", (int)getpid(), counter.fetch_add(1),
           (unsigned long long)Clock::now().time_since_epoch().count());
  return kSuccess;
}

int ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return kInvalidArgument;
  if (!bind_hip()) return kInternalError;
  FakeComm* c = new FakeComm;
  c->rank = rank;
  c->world = nranks;
  c->name.assign(id.internal, strnlen(id.internal, sizeof id.internal));
  c->box.resize((size_t)nranks);
  if (const char* t = getenv("GM_FAKE_RCCL_TIMEOUT_S")) c->timeout = atof(t);
  if (const char* f = getenv("GM_FAKE_RCCL_FAIL_AT")) {
    int r = -1;
    unsigned long long n = 0;
    if (sscanf(f, "%d:%llu", &r, &n) == 2) {
      c->fail_rank = r;
      c->fail_at = n;
    }
  }
  const auto t0 = Clock::now();
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Ctl)) != 0) {
      delete c;
      return kSystemError;
    }
  } else {
    while ((fd = shm_open(c->name.c_str(), O_RDWR, 0600)) < 0) {
      if (std::chrono::duration<double>(Clock::now() - t0).count() > c->timeout) {
        delete c;
        return kSystemError;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    struct stat st;
    while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(Ctl)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  c->ctl = static_cast<Ctl*>(mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->ctl == MAP_FAILED) {
    delete c;
    return kSystemError;
  }
  if (rank == 0) {
    c->ctl->world.store((uint64_t)nranks);
    c->ctl->magic.store(MAGIC, std::memory_order_release);
  } else {
    while (c->ctl->magic.load(std::memory_order_acquire) != MAGIC) {
      if (std::chrono::duration<double>(Clock::now() - t0).count() > c->timeout) return kSystemError;
      std::this_thread::yield();
    }
    if (c->ctl->world.load() != (uint64_t)nranks) return kInvalidArgument;
  }
  if (!own_box(c, 1 << 16)) return kSystemError;
  const int rc = barrier(c);  // everyone is attached and has an outbox
  if (rc) return rc;
  g_comm = c;
  *comm = c;
  return kSuccess;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, int /*datatype: ncclChar*/, ncclComm_t c, void* stream) {
  if (!c) return kInvalidArgument;
  if (injected_failure(c)) return kSystemError;
  if (hipStreamSynchronize_p(stream) != 0) return kCudaError;
  uint8_t* mine = own_box(c, count);
  if (!mine) return kSystemError;
  if (count && hipMemcpy_p(mine, sendbuff, count, kD2H) != 0) return kCudaError;
  c->ctl->box_bytes[c->rank].store(count, std::memory_order_release);
  int rc = barrier(c);
  if (rc) return rc;
  for (int p = 0; p < c->world; p++) {
    const uint8_t* src = peer_box(c, p);
    if (!src) return kSystemError;
    if (count && hipMemcpy_p(static_cast<uint8_t*>(recvbuff) + (size_t)p * count, src, count, kH2D) != 0) return kCudaError;
  }
  return barrier(c);
}

int ncclGroupStart() {
  g_depth++;
  return kSuccess;
}
int ncclGroupEnd() {
  if (g_depth <= 0) return kInvalidUsage;
  if (--g_depth > 0) return kSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (!g_comm) return ops.empty() ? kSuccess : kInvalidUsage;
  return run_group(g_comm, ops);
}
int ncclSend(const void* sendbuff, size_t count, int, int peer, ncclComm_t c, void* stream) {
  if (g_depth <= 0 || !c) return kInvalidUsage;  // (point-to-point outside a group is not something dist.cpp does)
  g_ops.push_back({true, const_cast<void*>(sendbuff), count, peer, stream});
  return kSuccess;
}
int ncclRecv(void* recvbuff, size_t count, int, int peer, ncclComm_t c, void* stream) {
  if (g_depth <= 0 || !c) return kInvalidUsage;
  g_ops.push_back({false, recvbuff, count, peer, stream});
  return kSuccess;
}

int ncclCommAbort(ncclComm_t c) {
  if (!c) return kInvalidArgument;
  c->ctl->abort_flag.store(1, std::memory_order_release);
  detach(c, c->rank == 0);
  return kSuccess;
}
int ncclCommDestroy(ncclComm_t c) {
  if (!c) return kInvalidArgument;
  detach(c, c->rank == 0);
  return kSuccess;
}
const char* ncclGetErrorString(int r) {
  static const char* names[] = {"no error", "unhandled device error (fake rccl)", "system error: a peer failed, aborted or never arrived (fake rccl)",
                                "internal error (fake rccl)", "invalid argument: unmatched or mis-sized point-to-point message (fake rccl)", "invalid usage (fake rccl)"};
  return r >= 0 && r <= 5 ? names[r] : "unknown (fake rccl)";
}

}  // extern "C"
