// The one collective of the inference path as a C-ABI entry point: the all-gather of per-rank arg-max coordinates that
// replaces tf.concat(hms_pred, axis=0) of the in-graph towers (main.py:573-574) when every GPU is its own process.
// RCCL (librccl.so.1, the ROCm build of the NCCL API) moves [B_local, 2, K] int32 per rank over xGMI.
//
// The library is resolved at run time, not at link time: inside a torch process dlopen("librccl.so.1") returns the copy
// torch already loaded (same SONAME), so there is one RCCL instance per process; a host that never calls these entry
// points does not need RCCL at all.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "ctx.h"

namespace {

typedef void* ncclComm_t_;
struct UniqueId { char internal[JCM_COMM_ID_BYTES]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(ncclComm_t_*, int, UniqueId, int);
typedef int (*CommDestroy_t)(ncclComm_t_);
typedef int (*AllGather_t)(const void*, void*, size_t, int /*ncclDataType_t*/, ncclComm_t_, hipStream_t);
typedef const char* (*GetErrorString_t)(int);
constexpr int kNcclInt32 = 2;      // ncclInt32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2)

struct Rccl {
  void* lib = nullptr;
  GetUniqueId_t get_id = nullptr;
  CommInitRank_t init_rank = nullptr;
  CommDestroy_t destroy = nullptr;
  AllGather_t all_gather = nullptr;
  GetErrorString_t err = nullptr;
  std::string why;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.why = std::string("RCCL is not loadable: ") + dlerror(); return; }
    r.get_id = reinterpret_cast<GetUniqueId_t>(dlsym(r.lib, "ncclGetUniqueId"));
    r.init_rank = reinterpret_cast<CommInitRank_t>(dlsym(r.lib, "ncclCommInitRank"));
    r.destroy = reinterpret_cast<CommDestroy_t>(dlsym(r.lib, "ncclCommDestroy"));
    r.all_gather = reinterpret_cast<AllGather_t>(dlsym(r.lib, "ncclAllGather"));
    r.err = reinterpret_cast<GetErrorString_t>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.get_id || !r.init_rank || !r.destroy || !r.all_gather) r.why = "librccl lacks the NCCL entry points";
  });
  return r;
}

int nccl_fail(const char* what, int code) {
  Rccl& r = rccl();
  return jcm::fail(JCM_ERR_HIP, std::string(what) + ": " + (r.err ? r.err(code) : "error ") + " (" + std::to_string(code) + ")");
}

}  // namespace

struct jcm_comm_s {
  ncclComm_t_ comm = nullptr;
  int world = 1, rank = 0, device = 0;
};

// ---- CRC-32C (Castagnoli) for the checkpoint files (tf_checkpoint.py: every tensor and every table block of a
// tf.train.Saver checkpoint carries one).  Host code; SSE4.2's crc32 instruction when the CPU has it.
namespace {
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const unsigned char* p, size_t n, uint32_t c) {
  uint64_t c64 = c;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
  for (; n >= 8; n -= 8, p += 8) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    c64 = __builtin_ia32_crc32di(c64, v);
  }
  while (n--) c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
  return (uint32_t)c64;
}
uint32_t crc32c_sw(const unsigned char* p, size_t n, uint32_t c) {
  static uint32_t table[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t v = i;
      for (int k = 0; k < 8; ++k) v = (v >> 1) ^ ((v & 1) ? 0x82F63B78u : 0u);
      table[i] = v;
    }
  });
  while (n--) c = table[(c ^ *p++) & 0xff] ^ (c >> 8);
  return c;
}
}  // namespace

extern "C" {

uint32_t jcm_crc32c(const void* data, size_t n, uint32_t crc) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  const uint32_t c = ~crc;
  static const bool hw = __builtin_cpu_supports("sse4.2");
  return ~(hw ? crc32c_hw(p, n, c) : crc32c_sw(p, n, c));
}

int jcm_comm_unique_id(unsigned char* id) {
  if (!id) return jcm::fail(JCM_ERR_ARG, "null id buffer");
  Rccl& r = rccl();
  if (!r.why.empty()) return jcm::fail(JCM_ERR_STATE, r.why);
  UniqueId u;
  if (int e = r.get_id(&u)) return nccl_fail("ncclGetUniqueId", e);
  std::memcpy(id, u.internal, JCM_COMM_ID_BYTES);
  return JCM_OK;
}

int jcm_comm_create(const unsigned char* id, int world, int rank, int device, jcm_comm* out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return jcm::fail(JCM_ERR_ARG, "bad comm_create arguments");
  Rccl& r = rccl();
  if (!r.why.empty()) return jcm::fail(JCM_ERR_STATE, r.why);
  jcm::DeviceGuard g(device);
  UniqueId u;
  std::memcpy(u.internal, id, JCM_COMM_ID_BYTES);
  jcm_comm_s* c = new jcm_comm_s();
  c->world = world; c->rank = rank; c->device = device;
  if (int e = r.init_rank(&c->comm, world, u, rank)) { delete c; return nccl_fail("ncclCommInitRank", e); }
  *out = c;
  return JCM_OK;
}

int jcm_comm_destroy(jcm_comm c) {
  if (!c) return JCM_OK;
  Rccl& r = rccl();
  int e = c->comm && r.destroy ? r.destroy(c->comm) : 0;
  delete c;
  return e ? nccl_fail("ncclCommDestroy", e) : JCM_OK;
}

int jcm_allgather_coords(jcm_handle h, jcm_comm c, const int32_t* local, int B_local, int32_t* all_out) {
  JCM_TRY(jcm::check(h, false));
  if (!c || !local || !all_out || B_local < 1) return jcm::fail(JCM_ERR_ARG, "bad allgather_coords arguments");
  if (c->device != h->device) return jcm::fail(JCM_ERR_ARG, "communicator and handle are on different devices");
  Rccl& r = rccl();
  jcm::DeviceGuard g(h->device);
  const size_t count = (size_t)B_local * 2 * h->K;
  {
    jcm::CallOrder order(h);      // the enqueue joins the device's call chain; the lock is released before the host waits for the peers
    if (int e = r.all_gather(local, all_out, count, kNcclInt32, c->comm, h->stream)) return nccl_fail("ncclAllGather", e);
  }
  HIP_TRY(hipStreamSynchronize(h->stream));       // the one call of the path that synchronises: the caller reads the result next
  return JCM_OK;
}

}  // extern "C"
