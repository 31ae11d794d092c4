// kmgpu.hip — libkmgpu.so: C-ABI (include/kmgpu.h) over the HIP kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <chrono>
#include <sys/mman.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/kmgpu.h"
#include "deliver_kernel.h"
#include "device_common.h"
#include "graph_kernel.h"
#include "jf_reader.h"
#include "table_kernels.h"
#include "walk_kernel.h"

using namespace kmd;

// ------------------------------------------------------------------ error plumbing
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(e_ == hipErrorOutOfMemory ? KM_E_NOMEM : KM_E_HIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                   \
  } while (0)

extern "C" const char* km_strerror(int code) {
  switch (code) {
    case KM_OK: return "ok";
    case KM_E_IO: return "I/O error";
    case KM_E_FORMAT: return "not a Jellyfish binary/sorted file";
    case KM_E_K: return "unsupported k (need 2 <= k <= 32)";
    case KM_E_ARG: return "bad argument";
    case KM_E_HIP: return "HIP runtime error";
    case KM_E_NOMEM: return "out of memory";
    case KM_E_STATE: return "call order violated";
    case KM_E_CAPACITY: return "output buffer too small";
  }
  return "unknown error";
}
extern "C" const char* km_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* km_version(void) { return "km_amd 0.1.0 (gfx950)"; }
extern "C" int km_device_count(int* n) {
  if (!n) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipGetDeviceCount(n));
  return KM_OK;
}

// ---- streams.  A pipelined consumer runs a few batches at a time, each on its own launch stream.  How
// those streams fall onto the GPU's hardware queues decides how well the batches overlap.  Measured on
// MI355X, four batches in flight (tools/pump_min.py): with the runtime's default of 4 hardware queues and
// k_graph_pure on a per-batch side stream (round 2's arrangement) 0.30 ms per step — every side stream
// shares a queue with ANOTHER batch's launch stream; 0.34 when launch streams themselves end up pairwise
// on one queue; 0.21 with 8 queues and the side streams on queues of their own; 0.19 with 8 queues and no
// side stream at all: a batch's kernels in ONE stream, every launch stream on a queue of its own.  So
// (1) there is no side stream any more, (2) the library asks for 8 hardware queues unless the environment
// says otherwise — when it is loaded, i.e. before the HIP runtime reads its settings — and (3) launch
// streams come from a per-device pool created once.
namespace {
__attribute__((constructor)) void km_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

constexpr int POOL_STREAMS = 7;                    // + the null stream: 8 hardware queues
struct StreamPool {
  std::vector<hipStream_t> launch;
  std::vector<char> in_use;                        // handed out by km_stream_create and not yet given back
};
std::mutex g_pool_mu;
std::map<int, StreamPool> g_pools;

// (device already current)  A pooled stream that nobody holds; once all are out, a fresh stream of the caller's
// own (two consumers never share a launch stream: a capture on it, or a wait for its last batch, would see the
// other's work).
int pool_get(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  StreamPool& p = g_pools[device];
  if (p.launch.empty()) {
    for (int i = 0; i < POOL_STREAMS; ++i) { hipStream_t st = nullptr; HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); p.launch.push_back(st); }
    p.in_use.assign(p.launch.size(), 0);
  }
  for (size_t i = 0; i < p.launch.size(); ++i)
    if (!p.in_use[i]) { p.in_use[i] = 1; *out = p.launch[i]; return KM_OK; }
  hipStream_t st = nullptr;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *out = st;
  return KM_OK;
}
// true: a pool stream (now free again); false: not ours to keep
bool pool_give_back(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& kv : g_pools)
    for (size_t i = 0; i < kv.second.launch.size(); ++i)
      if (kv.second.launch[i] == st) { kv.second.in_use[i] = 0; return true; }
  return false;
}
}  // namespace

extern "C" int km_stream_create(int device, void** stream) {
  if (!stream) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipSetDevice(device));
  hipStream_t st = nullptr;
  int rc = pool_get(device, &st);
  if (rc != KM_OK) return rc;
  *stream = st;
  return KM_OK;
}
// (pool streams live as long as the process: one handed back is free for the next km_stream_create; a stream made
// beyond the pool is destroyed)
extern "C" int km_stream_destroy(void* stream) {
  if (stream && !pool_give_back((hipStream_t)stream)) HIPCHK(hipStreamDestroy((hipStream_t)stream));
  return KM_OK;
}

// ------------------------------------------------------------------------ database
struct kmjf {
  int k = 0;
  int canonical = 0;
  std::vector<uint64_t> keys;
  std::vector<uint32_t> counts;
  uint64_t n_records = 0;
  // device table
  int device = -1;
  Slot* d_slots = nullptr;
  uint64_t n_slots = 0;
  uint32_t* d_dir = nullptr;     // [n_buckets + 1] (+ padding) exclusive prefix of bucket sizes
  uint32_t n_buckets = 0;
  uint32_t unit = 2;
  uint32_t max_probe = 2;
  OvfSlot* d_ovf = nullptr;
  uint64_t n_ovf = 0;
  uint64_t n_groups = 0;
};

static uint64_t mask_bits(int nbases) { return nbases >= 32 ? ~0ull : ((1ull << (2 * nbases)) - 1); }

static TableView view_of(const kmjf* h) {
  TableView t;
  t.slots = h->d_slots;
  t.dir = h->d_dir;
  t.n_slots = h->n_slots;
  t.ovf = h->d_ovf;
  t.n_ovf = h->n_ovf;
  t.kmask = mask_bits(h->k);
  t.pmask = mask_bits(h->k - 1);
  t.n_buckets = h->n_buckets;
  t.bshift = 31;                                    // (n_buckets: a power of two, 2^4 .. 2^30; 0 before the build)
  while (t.bshift > 1 && (1ull << (32 - t.bshift)) < h->n_buckets) --t.bshift;
  t.unit = h->unit;
  t.max_probe = h->max_probe;
  t.k = h->k;
  t.canonical = h->canonical;
  t.m = minimizer_len(h->k);
  t.w = h->k - t.m;
  t.mmask = (uint32_t)mask_bits(t.m);
  t.inv32 = (uint32_t)((1ull << 32) / ((uint64_t)2 * t.w * 256));
  t.cshift = 1;
  while ((1u << t.cshift) < 2u * (uint32_t)t.w) ++t.cshift;
  return t;
}

extern "C" int kmjf_open(const char* path, kmjf_t** out) {
  if (!path || !out) return fail(KM_E_ARG, "null argument");
  jfio::Records rec;
  std::string err;
  int rc = jfio::read_file(path, &rec, &err);
  if (rc == 1) return fail(KM_E_IO, "%s", err.c_str());
  if (rc == 2) return fail(KM_E_FORMAT, "%s", err.c_str());
  if (rc == 3) return fail(KM_E_K, "%s", err.c_str());
  if (rec.k < 2 || rec.k > 32) return fail(KM_E_K, "k=%d unsupported", rec.k);
  kmjf* h = new (std::nothrow) kmjf;
  if (!h) return fail(KM_E_NOMEM, "host allocation failed");
  h->k = rec.k;
  h->canonical = rec.canonical;
  h->keys.swap(rec.keys);
  h->counts.swap(rec.counts);
  h->n_records = h->keys.size();
  *out = h;
  return KM_OK;
}

extern "C" int kmjf_from_records(const uint64_t* keys, const uint32_t* counts, uint64_t n, int k,
                                 int canonical, kmjf_t** out) {
  if (!out || (n && (!keys || !counts))) return fail(KM_E_ARG, "null argument");
  if (k < 2 || k > 32) return fail(KM_E_K, "k=%d unsupported", k);
  kmjf* h = new (std::nothrow) kmjf;
  if (!h) return fail(KM_E_NOMEM, "host allocation failed");
  h->k = k;
  h->canonical = canonical ? 1 : 0;
  try {
    h->keys.assign(keys, keys + n);
    h->counts.assign(counts, counts + n);
  } catch (...) {
    delete h;
    return fail(KM_E_NOMEM, "host allocation failed");
  }
  h->n_records = n;
  *out = h;
  return KM_OK;
}

extern "C" int kmjf_create(int k, int canonical, kmjf_t** out) {
  return kmjf_from_records(nullptr, nullptr, 0, k, canonical, out);
}

static void free_table(kmjf* h) {
  if (h->d_slots) {
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_slots);
    h->d_slots = nullptr;
    if (h->d_ovf) (void)hipFree(h->d_ovf);
    h->d_ovf = nullptr;
    h->n_ovf = 0;
    if (h->d_dir) (void)hipFree(h->d_dir);
    h->d_dir = nullptr;
  }
  h->n_slots = h->n_groups = 0;
  h->n_buckets = 0;
  h->device = -1;
}

extern "C" int kmjf_close(kmjf_t* h) {
  if (!h) return KM_OK;
  free_table(h);
  delete h;
  return KM_OK;
}

extern "C" int kmjf_info(const kmjf_t* h, kmjf_info_t* info) {
  if (!h || !info) return fail(KM_E_ARG, "null argument");
  info->k = h->k;
  info->canonical = h->canonical;
  info->n_records = h->n_records;
  info->n_slots = h->n_slots;
  info->n_groups = h->n_groups;
  info->table_bytes = h->n_slots * sizeof(Slot) + h->n_ovf * sizeof(OvfSlot) +
                      (h->d_dir ? ((uint64_t)h->n_buckets + 1) * 4 : 0);
  info->device = h->device;
  info->max_probe = h->d_slots ? (int32_t)h->max_probe : 0;
  return KM_OK;
}

extern "C" int kmjf_records(const kmjf_t* h, const uint64_t** keys, const uint32_t** counts,
                            uint64_t* n) {
  if (!h || !keys || !counts || !n) return fail(KM_E_ARG, "null argument");
  *keys = h->keys.data();
  *counts = h->counts.data();
  *n = h->keys.size();
  return KM_OK;
}

static int grid_for(uint64_t n, int block) {
  uint64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 32) g = 256 * 32;      // grid-stride the rest
  return (int)g;
}

// Build, all on the device from device-resident records: count the entries of every minimizer
// bucket -> capacities -> exclusive scan (= the directory) -> insert every key into its home
// pair.  Buckets where some key found its pair taken are doubled and the table is rebuilt
// (a handful of rounds); the result is a table in which every lookup reads exactly one
// aligned 32-byte pair.
extern "C" int kmjf_upload_from_device(kmjf_t* h, int device, const uint64_t* d_keys,
                                       const uint32_t* d_counts, uint64_t n, void* stream) {
  if (!h || (n && (!d_keys || !d_counts))) return fail(KM_E_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  free_table(h);
  HIPCHK(hipSetDevice(device));
  // every record enters at most two groups
  const uint64_t max_entries = (h->canonical ? 2 : 1) * n;
  if (max_entries >= (1ull << 31)) return fail(KM_E_CAPACITY, "more than 2^31 table entries");
  // KM_TABLE_LOAD: initial load factor of every bucket (HBM capacity is plentiful): unit = 1/load
  uint32_t unit = 2;
  if (const char* lf = getenv("KM_TABLE_LOAD")) {
    double v = atof(lf);
    if (v >= 0.05 && v <= 0.5) unit = (uint32_t)(1.0 / v + 0.5);
  }
  // KM_DIR_LOG2: log2 of the bucket count (default: about one bucket per 2 entries; a
  // super-k-mer brings ~w entries of its own, so most buckets of real data are empty)
  uint32_t n_buckets = 1024;
  // (at most 1.5 entries per bucket: with 1.9 — a 500 M-k-mer sample under the old rule of 2 — half as many more buckets
  // double and the table takes 144 B per k-mer instead of ~105)
  while ((uint64_t)n_buckets * 3 < max_entries * 2 && n_buckets < (1u << 30)) n_buckets <<= 1;
  if (const char* dl = getenv("KM_DIR_LOG2")) { int v = atoi(dl); if (v >= 4 && v <= 30) n_buckets = 1u << v; }
  const uint32_t n_chunks = (uint32_t)(((uint64_t)n_buckets + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK);
  const uint64_t dir_words = (uint64_t)n_chunks * SCAN_CHUNK;
  uint32_t* dir = nullptr;
  uint32_t* caps = nullptr;
  uint32_t* sums = nullptr;
  unsigned long long* d_meta = nullptr;   // [0] occupied slots, [1] error, [2] flagged buckets, [3] max probe distance,
                                          // [4] big counts, [5] total capacity (pairs)
  Slot* slots = nullptr;
  uint64_t slots_cap = 0;
  OvfSlot* ovf = nullptr;
  uint32_t** ctr_ptr = nullptr;
  uint32_t* settle_bits = nullptr;      // one bit per bucket: on the list below
  uint32_t* settle_list = nullptr;      // buckets holding a key outside its home pair (k_table_settle)
  const uint32_t SETTLE_CAP = 1u << 22;
  const uint64_t settle_words = ((uint64_t)n_buckets + 31) / 32 + 1;
  auto bail = [&](int code, const char* what, hipError_t e) {
    if (dir) (void)hipFree(dir);
    if (caps) (void)hipFree(caps);
    if (settle_bits) (void)hipFree(settle_bits);
    if (settle_list) (void)hipFree(settle_list);
    if (sums) (void)hipFree(sums);
    if (d_meta) (void)hipFree(d_meta);
    if (slots) (void)hipFree(slots);
    if (ovf) (void)hipFree(ovf);
    if (ctr_ptr && *ctr_ptr) (void)hipFree(*ctr_ptr);
    return fail(code, "%s: %s", what, hipGetErrorString(e));
  };
  hipError_t e = hipMalloc((void**)&dir, dir_words * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&caps, dir_words * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&settle_bits, settle_words * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&settle_list, (uint64_t)SETTLE_CAP * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&sums, (uint64_t)n_chunks * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_meta, 64);
  if (e != hipSuccess) return bail(KM_E_NOMEM, "hipMalloc failed", e);
  (void)hipMemsetAsync(dir, 0, dir_words * 4, st);
  (void)hipMemsetAsync(caps, 0, dir_words * 4, st);
  (void)hipMemsetAsync(d_meta, 0, 64, st);

  kmjf shape;               // a view with the geometry only, for the build kernels
  shape.k = h->k; shape.canonical = h->canonical;
  shape.d_dir = dir; shape.n_buckets = n_buckets; shape.unit = unit;
  TableView tv = view_of(&shape);
  shape.d_dir = nullptr;    // not owned

  if (n) {
    hipLaunchKernelGGL(k_count_big, dim3(grid_for(n, 256)), dim3(256), 0, st, d_counts, n, d_meta + 4);
    hipLaunchKernelGGL(k_dir_count, dim3(grid_for(n, 256)), dim3(256), 0, st, tv, d_keys, d_counts, n, caps);
  }
  hipLaunchKernelGGL(k_dir_capacity, dim3(grid_for((uint64_t)n_buckets, 256)), dim3(256), 0, st, caps,
                     (uint64_t)n_buckets, unit, tv.cshift);
  // (KM_TABLE_LEAN_CROWDED=0: round 3's rule, a second doubling before a bucket becomes a two-choice table)
  const int lean_crowded = getenv("KM_TABLE_LEAN_CROWDED") ? atoi(getenv("KM_TABLE_LEAN_CROWDED")) : 1;
  const int MAX_ROUNDS = 5;             // CAP_MAX_GEN dry rounds, up to two more doublings found by the real
                                        // insert, then one final round that places every key wherever it fits
  unsigned long long meta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t n_slots = 0;
  uint32_t max_probe = 2;
  int rounds = 0, dry_rounds = 0;
  uint32_t* ctr = nullptr;          // dry rounds: entries per home pair, one byte each
  uint64_t ctr_cap = 0;
  ctr_ptr = &ctr;
  for (;; ++rounds) {
    const int final_round = rounds >= MAX_ROUNDS;
    (void)hipMemsetAsync(d_meta, 0, 32, st);          // [0..3]
    (void)hipMemsetAsync(d_meta + 5, 0, 8, st);
    hipLaunchKernelGGL(k_dir_copy, dim3(grid_for((uint64_t)n_buckets, 256)), dim3(256), 0, st, caps, dir,
                       (uint64_t)n_buckets, d_meta + 5);
    hipLaunchKernelGGL(k_scan_reduce, dim3(n_chunks), dim3(SCAN_THREADS), 0, st, dir, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, sums, n_chunks);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_chunks), dim3(SCAN_THREADS), 0, st, dir, sums);
    e = hipMemcpyAsync(meta, d_meta, 48, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(KM_E_HIP, "directory pass failed", e);
    if (meta[5] >= (1ull << 32))
      return bail(KM_E_CAPACITY, "table needs more than 2^33 slots (32-bit directory)", hipSuccess);
    n_slots = std::max<uint64_t>(64, 2ull * meta[5]);
    if (n && dry_rounds < (int)CAP_MAX_GEN) {
      // dry round (cheap: one byte per pair instead of the slots): find the buckets to double
      const uint64_t words = meta[5] / 4 + 2;
      if (words > ctr_cap) {
        if (ctr) (void)hipFree(ctr);
        ctr = nullptr;
        ctr_cap = words + words / 2;
        e = hipMalloc((void**)&ctr, ctr_cap * 4);
        if (e != hipSuccess) return bail(KM_E_NOMEM, "hipMalloc failed", e);
      }
      (void)hipMemsetAsync(ctr, 0, words * 4, st);
      hipLaunchKernelGGL(k_table_dry, dim3(grid_for(n, 256)), dim3(256), 0, st, tv, d_keys, d_counts, n, caps,
                         ctr, d_meta);
      e = hipMemcpyAsync(meta, d_meta, 32, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) return bail(KM_E_HIP, "table build failed", e);
      ++dry_rounds;
      if (getenv("KM_BUILD_VERBOSE"))
        fprintf(stderr, "libkmgpu: dry round %d: %llu slots, %llu buckets to grow\n", dry_rounds,
                (unsigned long long)n_slots, meta[2]);
      if (meta[2]) {
        hipLaunchKernelGGL(k_dir_grow, dim3(grid_for((uint64_t)n_buckets, 256)), dim3(256), 0, st, caps,
                           (uint64_t)n_buckets, tv.cshift, lean_crowded);
        continue;
      }
      dry_rounds = (int)CAP_MAX_GEN;               // nothing to grow: go straight to the insert
    }
    if (n_slots > slots_cap) {
      if (slots) (void)hipFree(slots);
      slots = nullptr;
      slots_cap = n_slots + n_slots / 4;              // head room for the following rounds
      e = hipMalloc((void**)&slots, (slots_cap + 16) * sizeof(Slot));
      if (e != hipSuccess) return bail(KM_E_NOMEM, "hipMalloc failed", e);
    }
    hipLaunchKernelGGL(k_table_init, dim3(grid_for(n_slots, 256)), dim3(256), 0, st, slots, n_slots);
    (void)hipMemsetAsync(settle_bits, 0, settle_words * 4, st);
    (void)hipMemsetAsync(d_meta + 6, 0, 16, st);
    if (n)
      hipLaunchKernelGGL(k_table_insert, dim3(grid_for(n, 256)), dim3(256), 0, st, tv, slots, d_keys,
                         d_counts, n, caps, final_round, d_meta, settle_bits, settle_list, SETTLE_CAP);
    e = hipMemcpyAsync(meta, d_meta, 64, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(KM_E_HIP, "table build failed", e);
    if (meta[1] & 0xFFFFFFFFull) return bail(KM_E_HIP, "table build overflowed", hipSuccess);
    max_probe = std::max<uint32_t>(2, (uint32_t)meta[3] + 1);
    // ---- settle: the buckets in which the race of the insert decided who sits where are laid out again as a
    // function of their keys alone (k_table_settle); that layout also decides which of them double once more
    if (n && meta[6] && meta[6] <= SETTLE_CAP && !getenv("KM_TABLE_NO_SETTLE")) {
      const uint32_t n_list = (uint32_t)meta[6];
      const uint32_t lds = 128u << 10;
      // (per device and cheap: set on every build rather than remembered per process)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_table_settle), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return bail(KM_E_HIP, "table settle pass: 128 KB of dynamic LDS refused", e);
      const uint32_t race_probe = max_probe;
      const unsigned long long n_slots_total = meta[5];
      (void)hipMemsetAsync(d_meta + 3, 0, 8, st);
      (void)hipMemsetAsync(d_meta + 5, 0, 8, st);
      hipLaunchKernelGGL(k_table_settle, dim3(n_list), dim3(256), lds, st, tv, slots, settle_list, n_list, lds, caps,
                         final_round, d_meta);
      // (a rejected launch would leave meta[3] = 0, i.e. max_probe 2 with keys further out: lookups would miss them)
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(meta, d_meta, 64, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) return bail(KM_E_HIP, "table settle pass failed", e);
      max_probe = std::max<uint32_t>(2, (uint32_t)meta[3] + 1);
      if (getenv("KM_BUILD_VERBOSE"))
        fprintf(stderr, "libkmgpu: settle pass: %u buckets laid out again by their keys alone (%llu too large: measured only); "
                "max_probe %u (the race had %u)\n", n_list, meta[5], max_probe, race_probe);
      meta[5] = n_slots_total;
    }
    if (getenv("KM_BUILD_VERBOSE"))
      fprintf(stderr, "libkmgpu: build round %d: %llu slots, %llu buckets to grow, max distance %llu; %llu buckets (%llu slots) hold a key outside its home pair\n", rounds,
              (unsigned long long)n_slots, meta[2], meta[3], meta[6], meta[7]);
    if (final_round) break;
    if (meta[2] == 0) break;
    hipLaunchKernelGGL(k_dir_grow, dim3(grid_for((uint64_t)n_buckets, 256)), dim3(256), 0, st, caps,
                       (uint64_t)n_buckets, tv.cshift, lean_crowded);
  }
  if (getenv("KM_BUILD_VERBOSE"))
    fprintf(stderr, "libkmgpu: table built in %d round(s): %llu slots for %llu groups, max_probe %u\n",
            rounds + 1, (unsigned long long)n_slots, meta[0], max_probe);
  // side table for the (rare) counts that do not fit 16 bits
  const uint64_t n_big = meta[4];
  const uint64_t n_ovf = n_big ? (n_big * 2 + 64) : 0;
  if (n_ovf) {
    e = hipMalloc((void**)&ovf, n_ovf * sizeof(OvfSlot));
    if (e != hipSuccess) return bail(KM_E_NOMEM, "hipMalloc failed", e);
    (void)hipMemsetAsync(ovf, 0, n_ovf * sizeof(OvfSlot), st);
    (void)hipMemsetAsync(d_meta + 1, 0, 8, st);
    hipLaunchKernelGGL(k_ovf_insert, dim3(grid_for(n, 256)), dim3(256), 0, st, d_keys, d_counts, n, h->k,
                       h->canonical, ovf, n_ovf, reinterpret_cast<unsigned int*>(d_meta + 1));
    unsigned long long err = 0;
    e = hipMemcpyAsync(&err, d_meta + 1, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(KM_E_HIP, "side table build failed", e);
    if (err & 0xFFFFFFFFull) return bail(KM_E_HIP, "side table overflowed", hipSuccess);
  }
  (void)hipFree(caps);
  (void)hipFree(settle_bits);
  (void)hipFree(settle_list);
  (void)hipFree(sums);
  (void)hipFree(d_meta);
  if (ctr) (void)hipFree(ctr);
  h->d_slots = slots;
  h->d_dir = dir;
  h->n_buckets = n_buckets;
  h->unit = unit;
  h->max_probe = max_probe;
  h->d_ovf = ovf;
  h->n_ovf = n_ovf;
  h->n_slots = n_slots;
  h->n_groups = meta[0];
  h->device = device;
  if (h->keys.empty()) h->n_records = n;
  return KM_OK;
}

extern "C" int kmjf_upload(kmjf_t* h, int device) {
  if (!h) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipSetDevice(device));
  const uint64_t n = h->keys.size();
  uint64_t* d_keys = nullptr;
  uint32_t* d_counts = nullptr;
  if (n) {
    HIPCHK(hipMalloc((void**)&d_keys, n * 8));
    hipError_t e = hipMalloc((void**)&d_counts, n * 4);
    if (e != hipSuccess) { (void)hipFree(d_keys); return fail(KM_E_NOMEM, "hipMalloc failed"); }
    e = hipMemcpy(d_keys, h->keys.data(), n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_counts, h->counts.data(), n * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d_keys); (void)hipFree(d_counts);
      return fail(KM_E_HIP, "record upload failed: %s", hipGetErrorString(e));
    }
  }
  int rc = kmjf_upload_from_device(h, device, d_keys, d_counts, n, nullptr);
  if (d_keys) (void)hipFree(d_keys);
  if (d_counts) (void)hipFree(d_counts);
  return rc;
}

// ---- kmjf_broadcast: one process, several GPUs.  RCCL is looked up at run time (dlopen) so that the library
// has no link-time dependency on it; only its types come from the header.
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
const RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(api.lib, "ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(api.lib, "ncclBroadcast"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
  });
  const bool ok = api.lib && api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Broadcast;
  return ok ? &api : nullptr;
}
}  // namespace

extern "C" int kmjf_broadcast(kmjf_t* h, const int* devices, int n, kmjf_t** replicas) {
  if (!h || !devices || !replicas || n < 1) return fail(KM_E_ARG, "bad argument");
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (devices[i] == devices[j]) return fail(KM_E_ARG, "device %d named twice", devices[i]);
  int n_dev = 0;
  HIPCHK(hipGetDeviceCount(&n_dev));
  for (int i = 0; i < n; ++i)
    if (devices[i] < 0 || devices[i] >= n_dev) return fail(KM_E_ARG, "no device %d (this process sees %d)", devices[i], n_dev);
  for (int i = 0; i < n; ++i) replicas[i] = nullptr;
  if (n == 1) {
    int rc = kmjf_upload(h, devices[0]);
    if (rc == KM_OK) replicas[0] = h;
    return rc;
  }
  const RcclApi* api = rccl_api();
  if (!api) return fail(KM_E_HIP, "RCCL (librccl.so.1) cannot be loaded: %s", dlerror() ? dlerror() : "symbols missing");
  const uint64_t cnt = h->keys.size();
  const uint64_t bytes = cnt * 12;                     // keys, then counts: one buffer, one broadcast
  std::vector<unsigned char*> buf(n, nullptr);
  std::vector<hipStream_t> st(n, nullptr);
  std::vector<ncclComm_t> comm(n, nullptr);
  std::vector<kmjf_t*> made;
  bool comms_up = false;
  auto cleanup = [&]() {
    for (int i = 0; i < n; ++i) {
      (void)hipSetDevice(devices[i]);
      if (buf[i]) (void)hipFree(buf[i]);
      if (st[i]) (void)hipStreamDestroy(st[i]);
      if (comms_up && comm[i]) (void)api->CommDestroy(comm[i]);
    }
  };
  auto bail = [&](int code, const char* what, const char* detail) {
    cleanup();
    for (kmjf_t* r : made) (void)kmjf_close(r);
    for (int i = 0; i < n; ++i) replicas[i] = nullptr;
    return fail(code, "%s: %s", what, detail);
  };
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipSetDevice(devices[i]);
    if (e == hipSuccess) e = hipMalloc((void**)&buf[i], bytes ? bytes : 16);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    if (e != hipSuccess) return bail(KM_E_NOMEM, "record buffer", hipGetErrorString(e));
  }
  {
    hipError_t e = hipSetDevice(devices[0]);
    if (e == hipSuccess && cnt) e = hipMemcpy(buf[0], h->keys.data(), cnt * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && cnt) e = hipMemcpy(buf[0] + cnt * 8, h->counts.data(), cnt * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(KM_E_HIP, "record upload", hipGetErrorString(e));
  }
  ncclResult_t nr = api->CommInitAll(comm.data(), n, devices);
  if (nr != ncclSuccess) return bail(KM_E_HIP, "ncclCommInitAll", api->GetErrorString ? api->GetErrorString(nr) : "failed");
  comms_up = true;
  if (bytes) {
    nr = api->GroupStart();
    for (int i = 0; i < n && nr == ncclSuccess; ++i) {
      (void)hipSetDevice(devices[i]);
      nr = api->Broadcast(buf[i], buf[i], bytes, ncclUint8, 0, comm[i], st[i]);
    }
    const ncclResult_t ne = api->GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return bail(KM_E_HIP, "ncclBroadcast", api->GetErrorString ? api->GetErrorString(nr) : "failed");
  }
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipSetDevice(devices[i]);
    if (e == hipSuccess) e = hipStreamSynchronize(st[i]);
    if (e != hipSuccess) return bail(KM_E_HIP, "broadcast did not complete", hipGetErrorString(e));
  }
  // every device builds its own table from its copy of the records
  for (int i = 0; i < n; ++i) {
    kmjf_t* r = h;
    if (i > 0) {
      int rc = kmjf_create(h->k, h->canonical, &r);
      if (rc != KM_OK) return bail(rc, "replica", km_last_error());
      made.push_back(r);
    }
    int rc = kmjf_upload_from_device(r, devices[i], reinterpret_cast<const uint64_t*>(buf[i]),
                                     reinterpret_cast<const uint32_t*>(buf[i] + cnt * 8), cnt, st[i]);
    if (rc != KM_OK) { const std::string why = km_last_error(); return bail(rc, "table build", why.c_str()); }
    replicas[i] = r;
  }
  cleanup();
  return KM_OK;
}

// Direct ingestion: header parsed on the host, the record area of the (memory-mapped) file is
// copied to HBM as it is, unpacked there (k_unpack_records) and the table is built from the
// device-resident records.  No host copy of the records is made or kept (kmjf_records()
// reports none).  Measured (bench.py `jf_ingestion`) against the host reader + upload.
extern "C" int kmjf_load(const char* path, int device, kmjf_t** out) {
  if (!path || !out) return fail(KM_E_ARG, "null argument");
  jfio::Layout lay;
  std::string err;
  void* file = nullptr;
  int rc = jfio::read_layout(path, &lay, &file, &err);
  if (rc == 1) return fail(KM_E_IO, "%s", err.c_str());
  if (rc == 2) return fail(KM_E_FORMAT, "%s", err.c_str());
  if (rc == 3) return fail(KM_E_K, "%s", err.c_str());
  FILE* f = static_cast<FILE*>(file);
  if (lay.k < 2 || lay.k > 32) { fclose(f); return fail(KM_E_K, "k=%d unsupported", lay.k); }
  const uint64_t n = lay.n_records;
  const uint64_t rec = (uint64_t)lay.key_bytes + lay.counter_bytes;
  const uint64_t body = n * rec;
  // map the whole file (the record area does not start on a page boundary)
  const uint64_t map_len = lay.body_offset + body;
  void* map = nullptr;
  if (body) {
    map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fileno(f), 0);   // populate: no per-page faults during the copy
    if (map == MAP_FAILED) { fclose(f); return fail(KM_E_IO, "cannot map %s", path); }
    (void)madvise(map, map_len, MADV_SEQUENTIAL);
  }
  fclose(f);                                   // the mapping stays valid
  kmjf* h = new (std::nothrow) kmjf;
  if (!h) { if (map) munmap(map, map_len); return fail(KM_E_NOMEM, "host allocation failed"); }
  h->k = lay.k;
  h->canonical = lay.canonical;

  unsigned char* d_raw = nullptr;
  uint64_t* d_keys = nullptr;
  uint32_t* d_counts = nullptr;
  unsigned long long* d_meta = nullptr;
  auto cleanup = [&]() {
    if (map) munmap(map, map_len);
    if (d_raw) (void)hipFree(d_raw);
    if (d_keys) (void)hipFree(d_keys);
    if (d_counts) (void)hipFree(d_counts);
    if (d_meta) (void)hipFree(d_meta);
  };
  auto bail = [&](int code, const char* what, hipError_t e) {
    cleanup();
    delete h;
    return fail(code, "%s: %s", what, hipGetErrorString(e));
  };
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return bail(KM_E_HIP, "device setup failed", e);
  // KM_LOAD_CHUNK_KB: copy granularity (default 256 MB; tests use small chunks)
  uint64_t chunk_target = 256ull << 20;
  if (const char* ck = getenv("KM_LOAD_CHUNK_KB")) { long v = atol(ck); if (v >= 1) chunk_target = (uint64_t)v << 10; }
  const uint64_t chunk_recs = std::max<uint64_t>(1, chunk_target / rec);
  if (n) {
    e = hipMalloc((void**)&d_raw, std::min(n, chunk_recs) * rec);
    if (e == hipSuccess) e = hipMalloc((void**)&d_keys, n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_counts, n * 4);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d_meta, 8);
  if (e != hipSuccess) return bail(KM_E_NOMEM, "allocation failed", e);
  (void)hipMemset(d_meta, 0, 8);
  const unsigned char* src = static_cast<const unsigned char*>(map) + lay.body_offset;
  for (uint64_t done = 0; done < n;) {
    const uint64_t m = std::min(chunk_recs, n - done);
    e = hipMemcpy(d_raw, src + done * rec, m * rec, hipMemcpyHostToDevice);   // pageable: staged by the runtime
    if (e != hipSuccess) return bail(KM_E_HIP, "ingestion failed", e);
    hipLaunchKernelGGL(k_unpack_records, dim3(grid_for(m, 256)), dim3(256), 0, nullptr, d_raw, m, lay.key_bytes,
                       lay.counter_bytes, d_keys + done, d_counts + done, d_meta);
    done += m;
  }
  unsigned long long nz = 0;
  e = hipMemcpy(&nz, d_meta, 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return bail(KM_E_HIP, "ingestion failed", e);
  if (d_raw) { (void)hipFree(d_raw); d_raw = nullptr; }
  rc = kmjf_upload_from_device(h, device, d_keys, d_counts, n, nullptr);
  cleanup();
  if (rc != KM_OK) { delete h; return rc; }
  h->n_records = nz;
  *out = h;
  return KM_OK;
}

// -------------------------------------------------------------------------- lookups
extern "C" int kmjf_query_batch_dev(kmjf_t* h, const uint64_t* d_kmers, uint64_t n,
                                    uint32_t* d_counts, void* stream) {
  if (!h || (n && (!d_kmers || !d_counts))) return fail(KM_E_ARG, "null argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
  if (!n) return KM_OK;
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_query, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     view_of(h), d_kmers, n, d_counts);
  HIPCHK(hipGetLastError());
  return KM_OK;
}

extern "C" int kmjf_children_batch_dev(kmjf_t* h, const uint64_t* d_kmers, uint64_t n, double ratio,
                                       int64_t n_cutoff, int forward, uint8_t* d_mask,
                                       uint32_t* d_counts4, void* stream) {
  if (!h || (n && !d_kmers)) return fail(KM_E_ARG, "null argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
  if (!n) return KM_OK;
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_children, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     view_of(h), d_kmers, n, ratio, n_cutoff, forward, d_mask, d_counts4);
  HIPCHK(hipGetLastError());
  return KM_OK;
}

extern "C" int kmjf_query_batch(kmjf_t* h, const uint64_t* kmers, uint64_t n, uint32_t* counts) {
  if (!h || (n && (!kmers || !counts))) return fail(KM_E_ARG, "null argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
  if (!n) return KM_OK;
  HIPCHK(hipSetDevice(h->device));
  uint64_t* dk = nullptr;
  uint32_t* dc = nullptr;
  HIPCHK(hipMalloc((void**)&dk, n * 8));
  hipError_t e = hipMalloc((void**)&dc, n * 4);
  if (e != hipSuccess) { (void)hipFree(dk); return fail(KM_E_NOMEM, "hipMalloc failed"); }
  int rc = KM_OK;
  e = hipMemcpy(dk, kmers, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = kmjf_query_batch_dev(h, dk, n, dc, nullptr);
    if (rc == KM_OK) e = hipMemcpy(counts, dc, n * 4, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dk); (void)hipFree(dc);
  if (rc != KM_OK) return rc;
  if (e != hipSuccess) return fail(KM_E_HIP, "query batch failed: %s", hipGetErrorString(e));
  return KM_OK;
}

extern "C" int kmjf_children_batch(kmjf_t* h, const uint64_t* kmers, uint64_t n, double ratio,
                                   int64_t n_cutoff, int forward, uint8_t* mask, uint32_t* counts4) {
  if (!h || (n && !kmers)) return fail(KM_E_ARG, "null argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
  if (!n) return KM_OK;
  HIPCHK(hipSetDevice(h->device));
  uint64_t* dk = nullptr;
  uint8_t* dm = nullptr;
  uint32_t* dc = nullptr;
  HIPCHK(hipMalloc((void**)&dk, n * 8));
  hipError_t e = hipMalloc((void**)&dm, n);
  if (e == hipSuccess) e = hipMalloc((void**)&dc, n * 16);
  int rc = KM_OK;
  if (e == hipSuccess) e = hipMemcpy(dk, kmers, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = kmjf_children_batch_dev(h, dk, n, ratio, n_cutoff, forward, dm, dc, nullptr);
    if (rc == KM_OK && mask) e = hipMemcpy(mask, dm, n, hipMemcpyDeviceToHost);
    if (rc == KM_OK && e == hipSuccess && counts4) e = hipMemcpy(counts4, dc, n * 16, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dk);
  if (dm) (void)hipFree(dm);
  if (dc) (void)hipFree(dc);
  if (rc != KM_OK) return rc;
  if (e != hipSuccess) return fail(KM_E_HIP, "children batch failed: %s", hipGetErrorString(e));
  return KM_OK;
}

// ---------------------------------------------------------------------------- environment knobs
// Read once per process.  The ones that make results INVALID (timing ablations) exist only in a
// diagnostics build of the library (-DKM_DIAGNOSTICS, what tools/_diag.py compiles): the product
// library cannot be talked into returning KM_OK over undelivered or partial data.
namespace {
struct Knobs {
  uint32_t debug_flags = 0;     // KM_DEBUG_FLAGS      (diagnostics build) stage cuts of k_dfs / k_graph
  int debug_deliver = 0;        // KM_DEBUG_DELIVER    (diagnostics build) skip delivery kernels / copy
  bool zero_copy = false;       // KM_DELIVER_ZEROCOPY (diagnostics build) pack straight into pinned memory
  int dfs_replay = 0;           // KM_DFS_REPLAY       (diagnostics build) k_dfs twice per step
  bool epilogue = true;         // KM_EPILOGUE=0: every flagged target through k_graph (results unchanged)
  bool seed_stamps = false;     // KM_SEED_STAMPS: in-kernel time stamps (results unchanged, slower)
  bool host_trace = false;      // KM_TRACE_HOST: host time of the sections of km_batch_run on stderr
  long spin_us = 0;             // KM_SPIN_US: poll the delivery event this long before sleeping on it
  uint32_t graph_grid = 0;      // KM_GRAPH_GRID: blocks of k_graph when the epilogue of k_dfs is on (tests: force the overflow path)
  bool speculate = true;        // KM_SPECULATE=0: k_dfs walks every chain one lookup after the other (results unchanged)
  bool dfs_grid_full = false;   // KM_DFS_GRID_FULL=1: one block of k_dfs per target of the batch, as before round 4
};
Knobs read_knobs() {
  Knobs q;
  auto num = [](const char* name, long dflt) { const char* v = getenv(name); return v ? strtol(v, nullptr, 0) : dflt; };
#ifdef KM_DIAGNOSTICS
  q.debug_flags = (uint32_t)num("KM_DEBUG_FLAGS", 0);
  q.debug_deliver = (int)num("KM_DEBUG_DELIVER", 0);
  q.zero_copy = num("KM_DELIVER_ZEROCOPY", 0) != 0;
  q.dfs_replay = (int)num("KM_DFS_REPLAY", 0);
#endif
  q.epilogue = num("KM_EPILOGUE", 1) != 0;
  q.seed_stamps = getenv("KM_SEED_STAMPS") != nullptr;
  q.host_trace = getenv("KM_TRACE_HOST") != nullptr;
  q.spin_us = num("KM_SPIN_US", 0);
  q.graph_grid = (uint32_t)std::max<long>(0, num("KM_GRAPH_GRID", 0));
  q.speculate = num("KM_SPECULATE", 1) != 0;
  q.dfs_grid_full = num("KM_DFS_GRID_FULL", 0) != 0;
  return q;
}
const Knobs& knobs() {
#ifdef KM_DIAGNOSTICS
  static thread_local Knobs k;      // the diagnostics tools change the ablation flags between runs
  k = read_knobs();
  return k;
#else
  static const Knobs k = read_knobs();
  return k;
#endif
}
}  // namespace

// ---------------------------------------------------------------------------- batch
namespace {

template <typename T>
struct DevBuf {
  T* p = nullptr;
  uint64_t n = 0;
  int alloc(uint64_t count) {
    if (count <= n && p) return KM_OK;
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) { p = nullptr; n = 0; return fail(KM_E_NOMEM, "hipMalloc of %llu bytes failed",
                                                          (unsigned long long)(count * sizeof(T))); }
    n = count;
    return KM_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

constexpr uint32_t FAST_EXTRA = 160;          // walk-discovered nodes a fast-tier target may add
constexpr uint32_t FAST_LDS_LIMIT = 64 * 1024;
constexpr uint32_t FAST_BCAP_MAX = 512;       // branch frames the fast tier keeps in LDS
constexpr uint32_t FAST_FCAP_MAX = 4096;      // stack frames per target in the fast tier's scratch
constexpr uint32_t LOOP_LOG_CAP = 4096;       // loop breaks of a batch kept for km_batch_graph_log (the rest is counted only)
constexpr uint32_t BIG_DEV_SLOTS = 32;        // targets per run the device's own large tier takes (the rest: the host's)
constexpr uint64_t BIG_DEV_MAX_BYTES = 1ull << 30;   // ... unless their node storage would exceed this (huge -n)

// Region A of the delivery buffer (deliver_kernel.h): offsets from n_targets alone.
struct OutLayout { uint64_t totals, status, n_ref, probes, node_off, extra_off, path_off, ref_max, esc_node, esc_value, a_bytes; };
OutLayout out_layout(uint32_t n) {
  auto al = [](uint64_t v) { return (v + 63) & ~63ull; };
  OutLayout L;
  uint64_t o = 0;
  L.totals = o;    o = al(o + 8ull * OT_WORDS);
  L.status = o;    o = al(o + 4ull * n);
  L.n_ref = o;     o = al(o + 4ull * n);
  L.probes = o;    o = al(o + 8ull * n);
  L.node_off = o;  o = al(o + 8ull * ((uint64_t)n + 1));
  L.extra_off = o; o = al(o + 8ull * ((uint64_t)n + 1));
  L.path_off = o;  o = al(o + 4ull * ((uint64_t)n + 1));
  L.ref_max = o;   o = al(o + 4ull * n);
  L.esc_node = o;  o = al(o + 8ull * OUT_ESC_CAP);
  L.esc_value = o; o = al(o + 4ull * OUT_ESC_CAP);
  L.a_bytes = o;
  return L;
}

}  // namespace

struct km_batch {
  kmjf* db = nullptr;
  km_params_t p{};
  uint32_t max_targets = 0;
  uint64_t max_bases = 0;
  int device = 0;
  uint32_t n_targets = 0;
  uint64_t total_bases = 0;
  uint64_t total_ref = 0;
  uint32_t max_len = 0;
  bool ran_walk = false, ran_graph = false, synced = true;
  hipStream_t last_stream = nullptr;

  // inputs
  DevBuf<uint8_t> d_bases;
  DevBuf<uint64_t> d_toff;
  std::vector<uint64_t> h_toff;
  DevBuf<uint64_t> d_woff, d_packed;   // 2-bit packed targets (k_pack)
  std::vector<uint64_t> h_woff;
  // k_seed work items and flag bitmaps
  DevBuf<uint32_t> d_item_off, d_flagbits, d_tflag, d_flagged, d_nflagged;
  DevBuf<uint4> d_flag_rec;            // k_seed -> k_dfs: one 32-byte record per flagged target
  DevBuf<uint32_t> d_left;             // k_dfs -> k_graph: flagged targets the epilogue did not answer
  DevBuf<EpiArgs> d_epi;               // where that epilogue writes (device copy of h_epi)
  EpiArgs h_epi{};
  bool epi_valid = false;
  DevBuf<unsigned long long> d_dfs_probes;
  DevBuf<uint64_t> d_items;
  DevBuf<uint64_t> d_fw_off;
  std::vector<uint32_t> h_item_off;
  std::vector<uint64_t> h_fw_off;
  uint32_t n_items = 0;
  int graph_mode = 0;                  // 1 = duplicate check only (walk stage run alone)
  // per-target
  DevBuf<uint64_t> d_node_base;
  DevBuf<uint32_t> d_node_cap;
  std::vector<uint64_t> h_node_base, h_node_base0;   // ...0: the fast-tier layout of layout_targets
  std::vector<uint32_t> h_node_cap, h_node_cap0;
  uint64_t node_pool0 = 0;
  bool layout_moved = false;           // the large tier re-homed some targets: restore before the next run
  DevBuf<uint32_t> d_n_nodes, d_n_ref, d_status, d_gstatus, d_npaths, d_pathbase, d_need_full, d_t_nruns, d_t_refmax;
  uint32_t pure_lds = 0;
  bool timed = false;                  // the last run recorded its timing events
  bool timed_fine = false;             // ... those between the kernels of the walk stage as well
  hipGraph_t graph = nullptr;          // captured step (KM_RUN_HIPGRAPH)
  hipGraphExec_t gexec = nullptr;
  int graph_stages = 0;
  hipStream_t graph_stream = nullptr;
  DevBuf<uint64_t> d_probes, d_fetches;
  // node pools
  DevBuf<uint64_t> d_node_kmer;
  DevBuf<uint32_t> d_node_cnt;
  uint64_t node_pool_used = 0;
  // path pools
  DevBuf<unsigned long long> d_counters;
  DevBuf<uint32_t> d_p_target, d_p_nruns, d_p_len, d_p_mincov, d_r_start, d_r_len;
  DevBuf<uint64_t> d_p_runbase;
  uint64_t path_pool = 0, run_pool = 0;
  // delivery (deliver_kernel.h): device buffer in its final host layout + its pinned host twin
  DevBuf<unsigned long long> d_loc, d_blk_tot, d_blk_base, d_psort;
  DevBuf<unsigned int> d_scan_ticket;
  DevBuf<uint32_t> d_cnt4;
  unsigned char* d_out = nullptr;
  unsigned char* h_out = nullptr;
  uint64_t out_cap = 0;
  hipEvent_t ev_out = nullptr;
  bool deliver_pending = false, result_ready = false;
  bool lean = false;                  // the pending / ready delivery omits bare-reference node counts
  bool count16 = false;               // ... and carries 16-bit counts + escape list (KM_DELIVER_COUNT16)
  bool count_fetches = false;         // the last run counted table slots read (KM_RUN_COUNT_FETCHES)
  uint64_t copied_tail = 0, tail_guess = 0;
  unsigned long long serial = 0;
  std::vector<uint64_t> h_packed;     // km_batch_fetch: packed targets, when node_kmer is asked for
  // large tier
  DevBuf<unsigned char> d_frames;     // fast-tier DFS stack frames, one slice per target
  DevBuf<unsigned long long> d_stamps; // diagnostics: k_seed time stamps (KM_SEED_STAMPS)
  DevBuf<float> d_tref;               // shared reference-chain distances
  DevBuf<uint32_t> d_big_ids;
  DevBuf<unsigned char> d_big_ws;
  // the device's own large tier (walk_kernel.h: WalkArgs::big_ctl)
  DevBuf<uint64_t> d_node_base0;
  DevBuf<uint32_t> d_big_ctl, d_big_walk, d_big_graph;
  // -v (km_batch_graph_log): reference edges stripped / edges kept per target, the walk's loop breaks
  DevBuf<uint32_t> d_t_eremoved, d_t_enonref, d_loop_list, d_loop_ctl;
  DevBuf<unsigned char> d_bigdev_walk_ws, d_bigdev_graph_ws;
  uint32_t big_entry = 0;              // nodes per slot of the region (0: tier off)
  uint64_t big_region = 0;             // its first node (the region sits in front of the fast-tier layout)
  uint32_t n_big_dev = 0;              // targets it took in the last synchronised run
  // Its two launches cost a step ~9 us when every kernel runs alone, needed or not.  They are launched once a
  // delivery of this batch has reported a target for the large tier (the first such target of a workspace's life
  // takes the host's path, as every one used to; KM_BIG_DEVICE=1 arms the tier from the first run)
  bool bigdev_armed = false;
  bool bigdev_ran = false;             // the last run launched it
  // k_graph's grid follows what the last delivery of this batch reported for its work list (4x + 64 blocks, at most
  // the default): with nothing left to it — the headline batch — its 1 250 blocks were a launch of 320 000 threads
  // that read two words each, 8.6 us alone and 31 us inside the pipeline.  More entries than blocks go to the
  // large tier (graph_kernel.h), and the next run's grid is larger.
  uint32_t graph_list_seen = 0xFFFFFFFFu;   // (nothing seen yet: the default grid)
  uint32_t flagged_seen = 0xFFFFFFFFu;      // flagged targets of the last delivered run: the grid of k_dfs
  // host mirrors after sync
  std::vector<uint32_t> h_status, h_gstatus, h_n_nodes, h_n_ref, h_npaths, h_pathbase;
  unsigned long long h_overflow = 0;
  uint32_t n_big = 0;
  // geometry of the last launch
  WalkArgs wa{};
  GraphArgs ga{};
  uint32_t walk_lds = 0, graph_lds = 0;
  // timing
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool timed_deliver = false;
};

static uint64_t default_tail_bytes(const km_batch* b, uint64_t nodes, uint64_t extra) {
  return out_align(4 * nodes) + out_align(8 * extra) + 2 * out_align(4 * b->path_pool) +
         out_align(8 * (b->path_pool + 1)) + 2 * out_align(4 * b->run_pool) + 256;
}

// Delivery buffers: region A for max_targets + `tail_need` bytes of tail.
static int ensure_out(km_batch* b, uint64_t tail_need) {
  const uint64_t need = out_layout(b->max_targets).a_bytes + tail_need;
  if (b->d_out && b->h_out && need <= b->out_cap) return KM_OK;
  if (b->d_out) (void)hipFree(b->d_out);
  if (b->h_out) (void)hipHostFree(b->h_out);
  b->d_out = nullptr; b->h_out = nullptr; b->out_cap = 0;
  const uint64_t cap = need + need / 8;
  if (hipMalloc((void**)&b->d_out, cap) != hipSuccess) { b->d_out = nullptr; return fail(KM_E_NOMEM, "hipMalloc of the delivery buffer failed"); }
  if (hipHostMalloc((void**)&b->h_out, cap, hipHostMallocDefault) != hipSuccess) {
    b->h_out = nullptr;
    return fail(KM_E_NOMEM, "pinned allocation of %llu bytes failed", (unsigned long long)cap);
  }
  b->out_cap = cap;
  return KM_OK;
}

extern "C" int km_batch_create(kmjf_t* h, const km_params_t* params, uint32_t max_targets,
                               uint64_t max_total_bases, km_batch_t** out) {
  if (!h || !params || !out || !max_targets) return fail(KM_E_ARG, "bad argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
  // word offsets of the packed targets and of the flag bitmaps travel as 32-bit halves of one record word
  if (max_total_bases / 32 + 2ull * max_targets + 2 >= (1ull << 32)) return fail(KM_E_ARG, "batch too large (more than 2^37 bases)");
  HIPCHK(hipSetDevice(h->device));
  km_batch* b = new (std::nothrow) km_batch;
  if (!b) return fail(KM_E_NOMEM, "host allocation failed");
  b->db = h;
  b->p = *params;
  b->max_targets = max_targets;
  b->max_bases = max_total_bases;
  b->device = h->device;
  int rc = KM_OK;
  auto A = [&](int r) { if (rc == KM_OK) rc = r; };
  A(b->d_bases.alloc(max_total_bases + 64));
  A(b->d_toff.alloc((uint64_t)max_targets + 1));
  A(b->d_woff.alloc((uint64_t)max_targets + 1));
  A(b->d_packed.alloc(max_total_bases / 32 + 2 * (uint64_t)max_targets + 2));
  A(b->d_items.alloc(16 * (max_total_bases / SEED_BLOCK + (uint64_t)max_targets + 1)));
  A(b->d_item_off.alloc((uint64_t)max_targets + 1));
  A(b->d_flagbits.alloc(max_total_bases / 32 + (uint64_t)max_targets + 1));
  A(b->d_fw_off.alloc((uint64_t)max_targets + 1));
  A(b->d_tflag.alloc(max_targets));
  A(b->d_flagged.alloc(max_targets));
  A(b->d_flag_rec.alloc(2ull * max_targets));
  A(b->d_left.alloc(max_targets));
  A(b->d_epi.alloc(1));
  A(b->d_nflagged.alloc(4));
  A(b->d_dfs_probes.alloc(max_targets));
  A(b->d_node_base.alloc(max_targets));
  A(b->d_node_cap.alloc(max_targets));
  A(b->d_n_nodes.alloc(max_targets));
  A(b->d_n_ref.alloc(max_targets));
  A(b->d_status.alloc(max_targets));
  A(b->d_gstatus.alloc(max_targets));
  A(b->d_npaths.alloc(max_targets));
  A(b->d_pathbase.alloc(max_targets));
  A(b->d_need_full.alloc(max_targets));
  A(b->d_t_nruns.alloc(max_targets));
  A(b->d_t_refmax.alloc(max_targets));
  A(b->d_loc.alloc(4ull * max_targets));
  A(b->d_cnt4.alloc(4ull * max_targets));
  A(b->d_blk_tot.alloc(8ull * (max_targets / OUT_SCAN_THREADS + 1)));
  A(b->d_blk_base.alloc(4ull * (max_targets / OUT_SCAN_THREADS + 1) + 8));
  A(b->d_scan_ticket.alloc(1));
  if (rc == KM_OK && hipMemset(b->d_scan_ticket.p, 0, 4) != hipSuccess) rc = fail(KM_E_HIP, "hipMemset failed");
  A(b->d_probes.alloc(max_targets));
  A(b->d_fetches.alloc(max_targets));
  {
    // the device's large tier: BIG_DEV_SLOTS slots of the reference's own bound on a walk (MutationFinder.py:140-156)
    const uint64_t entry = (uint64_t)params->max_node + params->max_stack + 1;
    if (!getenv("KM_BIG_DEVICE_OFF") && entry < 0x7FFFFFFFull && entry * BIG_DEV_SLOTS * 12 <= BIG_DEV_MAX_BYTES) b->big_entry = (uint32_t)entry;
    if (const char* e = getenv("KM_BIG_DEVICE")) b->bigdev_armed = atoi(e) != 0;
  }
  const uint64_t pool = max_total_bases + (uint64_t)max_targets * FAST_EXTRA + (uint64_t)b->big_entry * BIG_DEV_SLOTS;
  A(b->d_node_base0.alloc(max_targets));
  A(b->d_big_ctl.alloc(8));
  A(b->d_big_walk.alloc(BIG_DEV_SLOTS));
  A(b->d_big_graph.alloc(BIG_DEV_SLOTS));
  A(b->d_t_eremoved.alloc(max_targets));
  A(b->d_t_enonref.alloc(max_targets));
  A(b->d_loop_list.alloc(2ull * LOOP_LOG_CAP));
  A(b->d_loop_ctl.alloc(4));
  if (rc == KM_OK && hipMemset(b->d_loop_ctl.p, 0, 16) != hipSuccess) rc = fail(KM_E_HIP, "hipMemset failed");
  if (rc == KM_OK && hipMemset(b->d_big_ctl.p, 0, 32) != hipSuccess) rc = fail(KM_E_HIP, "hipMemset failed");
  A(b->d_node_kmer.alloc(pool));
  A(b->d_node_cnt.alloc(pool));
  A(b->d_counters.alloc(POOL_GROUPS * POOL_CTR_STRIDE + 16));
  b->path_pool = (((uint64_t)max_targets * 4 + 8192) / POOL_GROUPS + 1) * POOL_GROUPS;
  b->run_pool = (((uint64_t)max_targets * 16 + 32768) / POOL_GROUPS + 1) * POOL_GROUPS;
  if (getenv("KM_TEST_SMALL_POOLS")) {         // tests: force the pool-overflow path of km_batch_sync
    b->path_pool = 2 * POOL_GROUPS;
    b->run_pool = 4 * POOL_GROUPS;
  }
  A(b->d_p_target.alloc(b->path_pool));
  A(b->d_p_runbase.alloc(b->path_pool));
  A(b->d_p_nruns.alloc(b->path_pool));
  A(b->d_p_len.alloc(b->path_pool));
  A(b->d_p_mincov.alloc(b->path_pool));
  A(b->d_psort.alloc(b->path_pool));
  A(b->d_r_start.alloc(b->run_pool));
  A(b->d_r_len.alloc(b->run_pool));
  // the walk rarely adds more than a few nodes per target: the tail grows on demand
  if (rc == KM_OK) rc = ensure_out(b, default_tail_bytes(b, max_total_bases + 16ull * max_targets, 16ull * max_targets));
  if (rc == KM_OK) {
    if (hipEventCreateWithFlags(&b->ev_out, hipEventDisableTiming) != hipSuccess)
      rc = fail(KM_E_HIP, "stream/event creation failed");
  }
  if (rc == KM_OK) {
    for (int i = 0; i < 7; ++i)
      if (hipEventCreate(&b->ev[i]) != hipSuccess) rc = fail(KM_E_HIP, "hipEventCreate failed");
  }
  if (rc != KM_OK) { km_batch_destroy(b); return rc; }
  *out = b;
  return KM_OK;
}

static void drop_graph(km_batch* b) {
  if (b->gexec) (void)hipGraphExecDestroy(b->gexec);
  if (b->graph) (void)hipGraphDestroy(b->graph);
  b->gexec = nullptr;
  b->graph = nullptr;
}

extern "C" int km_batch_destroy(km_batch_t* b) {
  if (!b) return KM_OK;
  (void)hipSetDevice(b->device);
  (void)hipDeviceSynchronize();
  drop_graph(b);
  b->d_bases.release(); b->d_toff.release(); b->d_woff.release(); b->d_packed.release();
  b->d_items.release(); b->d_item_off.release(); b->d_flagbits.release(); b->d_fw_off.release();
  b->d_tflag.release(); b->d_flagged.release(); b->d_flag_rec.release(); b->d_left.release(); b->d_epi.release(); b->d_nflagged.release(); b->d_dfs_probes.release();
  b->d_node_base.release(); b->d_node_cap.release();
  b->d_n_nodes.release(); b->d_n_ref.release(); b->d_status.release(); b->d_gstatus.release();
  b->d_npaths.release(); b->d_pathbase.release(); b->d_need_full.release(); b->d_t_nruns.release(); b->d_t_refmax.release();
  b->d_loc.release(); b->d_cnt4.release(); b->d_blk_tot.release(); b->d_blk_base.release(); b->d_scan_ticket.release(); b->d_psort.release();
  if (b->ev_out) (void)hipEventDestroy(b->ev_out);
  b->d_probes.release(); b->d_fetches.release();
  b->d_node_kmer.release(); b->d_node_cnt.release(); b->d_counters.release();
  b->d_p_target.release(); b->d_p_runbase.release(); b->d_p_nruns.release(); b->d_p_len.release();
  b->d_p_mincov.release(); b->d_r_start.release(); b->d_r_len.release();
  b->d_big_ids.release(); b->d_big_ws.release(); b->d_tref.release(); b->d_frames.release(); b->d_stamps.release();
  b->d_node_base0.release(); b->d_big_ctl.release(); b->d_big_walk.release(); b->d_big_graph.release();
  b->d_bigdev_walk_ws.release(); b->d_bigdev_graph_ws.release();
  b->d_t_eremoved.release(); b->d_t_enonref.release(); b->d_loop_list.release(); b->d_loop_ctl.release();
  for (int i = 0; i < 7; ++i) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  if (b->d_out) (void)hipFree(b->d_out);
  if (b->h_out) (void)hipHostFree(b->h_out);
  delete b;
  return KM_OK;
}

static uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Per-batch geometry + per-target node storage layout.
static int layout_targets(km_batch* b, const uint64_t* offsets, uint32_t n) {
  if (n > b->max_targets) return fail(KM_E_ARG, "too many targets for this batch (%u > %u)", n, b->max_targets);
  const uint64_t total = offsets[n] - offsets[0];
  if (total > b->max_bases) return fail(KM_E_ARG, "too many bases for this batch");
  const int k = b->db->k;
  b->h_toff.assign(n + 1, 0);
  b->h_woff.assign(n + 1, 0);
  b->h_fw_off.assign(n + 1, 0);
  b->h_item_off.assign(n + 1, 0);
  b->h_node_base.assign(n, 0);
  b->h_node_cap.assign(n, 0);
  uint64_t pool = (uint64_t)b->big_entry * BIG_DEV_SLOTS, total_ref = 0;   // (the large tier's region comes first)
  b->big_region = 0;
  uint32_t max_len = 0;
  for (uint32_t t = 0; t < n; ++t) {
    if (offsets[t + 1] < offsets[t]) return fail(KM_E_ARG, "offsets must be non-decreasing");
    const uint64_t L = offsets[t + 1] - offsets[t];
    if (L > 0x7FFFFFFFull) return fail(KM_E_ARG, "target too long");
    b->h_toff[t] = offsets[t] - offsets[0];
    b->h_woff[t + 1] = b->h_woff[t] + (L + 31) / 32 + 1;
    const uint32_t n_ref = (L >= (uint64_t)k) ? (uint32_t)(L - k + 1) : 0;
    b->h_fw_off[t + 1] = b->h_fw_off[t] + (n_ref + 31) / 32;
    b->h_item_off[t + 1] = b->h_item_off[t] + (n_ref + SEED_BLOCK - 1) / SEED_BLOCK;
    b->h_node_base[t] = pool;
    b->h_node_cap[t] = n_ref + FAST_EXTRA;
    pool += (uint64_t)n_ref + FAST_EXTRA;
    total_ref += n_ref;
    max_len = std::max<uint32_t>(max_len, (uint32_t)L);
  }
  b->h_toff[n] = total;
  b->h_node_base0 = b->h_node_base;
  b->h_node_cap0 = b->h_node_cap;
  b->node_pool0 = b->node_pool_used = pool;
  b->layout_moved = false;
  b->n_items = b->h_item_off[n];
  b->n_targets = n;
  b->total_bases = total;
  b->total_ref = total_ref;
  b->max_len = max_len;
  b->tail_guess = 4 * total_ref + total_ref / 2 + (64u << 10);
  b->h_packed.clear();
  return KM_OK;
}

static int push_layout(km_batch* b, hipStream_t st) {
  const uint32_t n = b->n_targets;
  drop_graph(b);                       // geometry and pointers may change with the targets
  {
    // tref[j] = distance of reference node j from the source along the reference chain,
    // accumulated exactly as Graph.py does: float32 0 + 0.01f, then + 0.01f per hop
    const uint32_t need = b->max_len + 2;
    if (b->d_tref.n < need) {
      int rc = b->d_tref.alloc(std::max<uint64_t>(need, 4096));
      if (rc != KM_OK) return rc;
      std::vector<float> h(b->d_tref.n);
      volatile float acc = 0.0f;
      for (size_t j = 0; j < h.size(); ++j) { acc = acc + 0.01f; h[j] = acc; }
      HIPCHK(hipMemcpy(b->d_tref.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  HIPCHK(hipMemcpyAsync(b->d_toff.p, b->h_toff.data(), (uint64_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_woff.p, b->h_woff.data(), (uint64_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_fw_off.p, b->h_fw_off.data(), (uint64_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_item_off.p, b->h_item_off.data(), (uint64_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_node_base.p, b->h_node_base.data(), (uint64_t)n * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_node_base0.p, b->h_node_base.data(), (uint64_t)n * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_node_cap.p, b->h_node_cap.data(), (uint64_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  b->ran_walk = b->ran_graph = false;
  b->synced = true;
  b->deliver_pending = b->result_ready = false;
  return KM_OK;
}

extern "C" int km_batch_set_targets(km_batch_t* b, const uint8_t* bases, const uint64_t* offsets,
                                    uint32_t n_targets) {
  if (!b || !offsets || (!bases && n_targets)) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device));
  int rc = layout_targets(b, offsets, n_targets);
  if (rc != KM_OK) return rc;
  if (b->total_bases)
    HIPCHK(hipMemcpy(b->d_bases.p, bases + offsets[0], b->total_bases, hipMemcpyHostToDevice));
  return push_layout(b, nullptr);
}

extern "C" int km_batch_set_targets_dev(km_batch_t* b, const uint8_t* d_bases,
                                        const uint64_t* offsets_host, uint32_t n_targets, void* stream) {
  if (!b || !offsets_host || (!d_bases && n_targets)) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device));
  int rc = layout_targets(b, offsets_host, n_targets);
  if (rc != KM_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (b->total_bases)
    HIPCHK(hipMemcpyAsync(b->d_bases.p, d_bases + offsets_host[0], b->total_bases,
                          hipMemcpyDeviceToDevice, st));
  return push_layout(b, st);
}

static void fill_walk_args(km_batch* b, WalkArgs& a) {
  memset(&a, 0, sizeof a);
  a.tab = view_of(b->db);
  a.bases = b->d_bases.p;
  a.toff = b->d_toff.p;
  a.packed = b->d_packed.p;
  a.woff = b->d_woff.p;
  a.n_targets = b->n_targets;
  a.ratio = b->p.ratio;
  a.n_cutoff = b->p.count;
  a.nc = (double)b->p.count;
  threshold_shortcut(a.ratio, a.n_cutoff, &a.thr_below, &a.thr_T);
  a.max_stack = b->p.max_stack;
  a.max_break = b->p.max_break;
  a.max_node = b->p.max_node;
  a.items = b->d_items.p;
  a.item_off = b->d_item_off.p;
  a.n_items = b->n_items;
  a.flagbits = b->d_flagbits.p;
  a.fw_off = b->d_fw_off.p;
  a.tflag = b->d_tflag.p;
  a.flagged = b->d_flagged.p;
  a.flag_rec = b->d_flag_rec.p;
  a.fast_extra = FAST_EXTRA;
  a.epi = nullptr;
  a.n_flagged = b->d_nflagged.p;
  a.list = b->d_flagged.p;
  a.n_list_dev = b->d_nflagged.p;
  a.n_list_host = 0;
  a.node_kmer = b->d_node_kmer.p;
  a.node_cnt = b->d_node_cnt.p;
  a.node_base = b->d_node_base.p;
  a.node_cap = b->d_node_cap.p;
  a.node_base0 = b->d_node_base0.p;
  a.big_ctl = b->big_entry ? b->d_big_ctl.p : nullptr;
  a.big_walk = b->d_big_walk.p;
  a.big_slots = BIG_DEV_SLOTS;
  a.big_entry = b->big_entry;
  a.big_region = b->big_region;
  a.big_prep = 0;
  a.loop_list = b->d_loop_list.p;
  a.loop_ctl = b->d_loop_ctl.p;
  a.loop_cap = LOOP_LOG_CAP;
  a.t_eremoved = b->d_t_eremoved.p;
  a.t_enonref = b->d_t_enonref.p;
  a.n_nodes = b->d_n_nodes.p;
  a.n_ref = b->d_n_ref.p;
  a.status = b->d_status.p;
  a.probes = reinterpret_cast<unsigned long long*>(b->d_probes.p);
  a.dfs_probes = b->d_dfs_probes.p;
  a.fetches = reinterpret_cast<unsigned long long*>(b->d_fetches.p);
  a.g_ws = nullptr;
  a.g_stride = 0;
  a.dbg = knobs().debug_flags & 0xFFu;          // timing ablations (diagnostics build only); results are invalid
  a.spec = knobs().speculate ? 1u : 0u;
}

static void fill_graph_args(km_batch* b, GraphArgs& g) {
  g.k = b->db->k;
  g.kmask = mask_bits(b->db->k);
  g.pmask = mask_bits(b->db->k - 1);
  g.tids = nullptr;
  g.work_list = b->d_flagged.p;
  g.work_n = b->d_nflagged.p;
  g.left = b->d_left.p;
  g.dfs_answers = 0;
  g.big_ctl = b->big_entry ? b->d_big_ctl.p : nullptr;
  g.big_graph = b->d_big_graph.p;
  g.big_slots = BIG_DEV_SLOTS;
  g.tids_n = nullptr;
  g.n_targets = b->n_targets;
  g.node_kmer = b->d_node_kmer.p;
  g.node_cnt = b->d_node_cnt.p;
  g.node_base = b->d_node_base.p;
  g.packed = b->d_packed.p;
  g.woff = b->d_woff.p;
  g.words_cap = 0;
  g.n_nodes = b->d_n_nodes.p;
  g.n_ref = b->d_n_ref.p;
  g.status = b->d_status.p;
  g.tflag = b->d_tflag.p;
  g.need_full = b->d_need_full.p;
  g.use_need_full = 0;
  g.hcap_pure = 0;
  g.g_status = b->d_gstatus.p;
  g.t_npaths = b->d_npaths.p;
  g.t_pathbase = b->d_pathbase.p;
  g.t_nruns = b->d_t_nruns.p;
  g.t_refmax = b->d_t_refmax.p;
  g.t_eremoved = b->d_t_eremoved.p;
  g.t_enonref = b->d_t_enonref.p;
  g.counters = b->d_counters.p;
  g.path_pool = b->path_pool;
  g.run_pool = b->run_pool;
  g.p_target = b->d_p_target.p;
  g.p_runbase = b->d_p_runbase.p;
  g.p_nruns = b->d_p_nruns.p;
  g.p_len = b->d_p_len.p;
  g.p_mincov = b->d_p_mincov.p;
  g.r_start = b->d_r_start.p;
  g.r_len = b->d_r_len.p;
  g.tref = b->d_tref.p;
  g.tref_len = (uint32_t)std::min<uint64_t>(b->d_tref.n, 0xFFFFFFFFull);
  g.g_ws = nullptr;
  g.g_stride = 0;
  g.dbg = knobs().debug_flags >> 8;
  if (b->graph_mode == 1) g.dbg = 1;       // duplicate check only
  if (g.dbg && !(g.dbg & 0x80u)) g.work_list = nullptr;   // every target goes through k_graph: no list
}

// ---- fast-tier geometry.  The LDS-resident kernels are sized for the longest target of the
// batch that still fits FAST_LDS_LIMIT; longer targets (and walks that outgrow the extra-node,
// branch-frame or stack-frame allowance) are flagged T_NEEDS_BIG by the kernels themselves, one
// by one, and finished by the large tier in km_batch_sync.  One long target does not demote the
// rest of its batch.
static uint32_t words_cap_for(uint32_t len) { return round_up((len + 31) / 32 + 1, 2); }

// slots of k_dfs's node set in the fast tier.  It holds the walk's nodes, the stack, and the target k-mers that lost
// their slot of the position table (a fifth of them with the table at load 1/2): room for a quarter of the target's
// k-mers + every allowed extra node + 64 frames, at load <= 3/4 (what does not fit goes to the large tier)
static uint32_t walk_hs_cap(uint32_t nref) { return round_up((uint32_t)(((uint64_t)(nref / 4 + FAST_EXTRA + 64) * 4 + 2) / 3), 64); }
// slots of its position table: the power of two >= four times the target's k-mers (load <= 1/4: a tenth of the k-mers lose their slot)
static uint32_t walk_pcap(uint32_t nref) { uint32_t p = 64; while (p < 4 * nref) p <<= 1; return p; }

static bool fast_fits(const km_batch* b, uint32_t nref, uint32_t bcap) {
  const uint32_t len = nref + (uint32_t)b->db->k - 1;
  const uint32_t wc = words_cap_for(len);
  const uint32_t hs = walk_hs_cap(nref);
  const uint32_t ncap = nref + FAST_EXTRA + 2, hcap = round_up(ncap + ncap / 2 + 1, 64);
  return walk_lds_bytes(hs, wc, bcap, walk_pcap(nref), 2) <= FAST_LDS_LIMIT &&
         graph_ws_bytes<uint16_t>(ncap, hcap, wc) <= FAST_LDS_LIMIT && ncap < 0xFFFF &&
         (uint64_t)hcap * 4 + (uint64_t)wc * 8 <= FAST_LDS_LIMIT;   // (k_graph_pure hands over what its own table cannot hold)
}

static void fast_geometry(km_batch* b) {
  const int k = b->db->k;
  const uint32_t max_nref = b->max_len >= (uint32_t)k ? b->max_len - k + 1 : 1;
  const uint32_t bcap = std::min<uint32_t>(b->p.max_break, FAST_BCAP_MAX - 1) + 1;
  uint32_t nref = max_nref;
  if (!fast_fits(b, nref, bcap)) {
    uint32_t lo = 1, hi = max_nref;          // fits(lo) holds: a 1-k-mer target always fits
    while (lo + 1 < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (fast_fits(b, mid, bcap)) lo = mid; else hi = mid;
    }
    nref = lo;
  }
  const uint32_t len = nref + (uint32_t)k - 1;
  WalkArgs& wa = b->wa;
  fill_walk_args(b, wa);
  wa.hs_cap = walk_hs_cap(nref);
  wa.pcap = walk_pcap(nref);
  wa.words_cap = words_cap_for(len);
  wa.fcap = round_up(std::min<uint32_t>(b->p.max_stack, FAST_FCAP_MAX - 2) + 2, 2);
  wa.bcap = bcap;
  wa.f_stride = walk_frame_bytes(wa.fcap);
  b->walk_lds = (uint32_t)walk_lds_bytes(wa.hs_cap, wa.words_cap, wa.bcap, wa.pcap, 2);
  GraphArgs& ga = b->ga;
  fill_graph_args(b, ga);
  ga.ncap = nref + FAST_EXTRA + 2;
  ga.hcap = round_up(ga.ncap + ga.ncap / 2 + 1, 64);
  ga.words_cap = wa.words_cap;
  b->graph_lds = (uint32_t)graph_ws_bytes<uint16_t>(ga.ncap, ga.hcap, ga.words_cap);
  ga.hcap_pure = 64;                                             // position table at load <= 1/4 (graph_kernel.h: k_graph_pure)
  while (ga.hcap_pure < 4 * (nref + 2)) ga.hcap_pure <<= 1;
  b->pure_lds = (uint32_t)pure_lds_bytes(ga.hcap_pure, ga.words_cap);
  if (b->pure_lds > FAST_LDS_LIMIT) {                            // all -> need_full
    ga.hcap_pure = 64;
    b->pure_lds = (uint32_t)pure_lds_bytes(64, ga.words_cap);
  }
  // the epilogue of k_dfs answers the regular flagged targets when the graph stage is wanted in full
  // (KM_EPILOGUE=0: diagnostics, everything through k_graph as in round 2)
  if (knobs().epilogue && b->graph_mode == 0 && ga.dbg == 0 && ga.work_list != nullptr) {
    EpiArgs e;
    memset(&e, 0, sizeof e);
    e.counters = ga.counters; e.path_pool = ga.path_pool; e.run_pool = ga.run_pool;
    e.p_target = ga.p_target; e.p_runbase = ga.p_runbase; e.p_nruns = ga.p_nruns; e.p_len = ga.p_len;
    e.p_mincov = ga.p_mincov; e.r_start = ga.r_start; e.r_len = ga.r_len;
    e.g_status = ga.g_status; e.t_npaths = ga.t_npaths; e.t_pathbase = ga.t_pathbase; e.t_nruns = ga.t_nruns;
    e.t_refmax = ga.t_refmax;
    e.t_eremoved = ga.t_eremoved; e.t_enonref = ga.t_enonref;
    e.left = b->d_left.p; e.n_left = b->d_nflagged.p + 2;
    if (!b->epi_valid || memcmp(&e, &b->h_epi, sizeof e) != 0) {
      if (hipMemcpy(b->d_epi.p, &e, sizeof e, hipMemcpyHostToDevice) == hipSuccess) { b->h_epi = e; b->epi_valid = true; }
      else b->epi_valid = false;
    }
    if (b->epi_valid) { wa.epi = b->d_epi.p; ga.dfs_answers = 1; }
  }
}

// LDS-tier graph kernels, instantiated for k = 31 where that is the database's k
static void launch_pure(km_batch* b, hipStream_t st, const GraphArgs& ga) {
  if (ga.k == 31) hipLaunchKernelGGL((k_graph_pure<31>), dim3(b->n_targets), dim3(64), b->pure_lds, st, ga);
  else hipLaunchKernelGGL((k_graph_pure<0>), dim3(b->n_targets), dim3(64), b->pure_lds, st, ga);
}
static void launch_graph(km_batch* b, hipStream_t st, const GraphArgs& ga) {
  // One block per entry of the work list.  When the epilogue of k_dfs answers the regular targets the list is a
  // percent of the batch: a grid of an eighth of the batch (at least 1 024 blocks; KM_GRAPH_GRID sets it) instead
  // of one block per target, of which nearly all left at once; should more be left than that, the kernel hands
  // the rest to the large tier (graph_kernel.h).
  uint32_t grid = b->n_targets;
  if (ga.work_list && ga.dfs_answers) {
    uint32_t cap = knobs().graph_grid ? knobs().graph_grid : std::max<uint32_t>(1024u, b->n_targets / 8);
    if (!knobs().graph_grid && b->graph_list_seen != 0xFFFFFFFFu)
      cap = std::min<uint32_t>(cap, (uint32_t)std::min<uint64_t>(4ull * b->graph_list_seen + 64, 0x7FFFFFFFull));
    grid = std::min<uint32_t>(grid, cap);
  }
  if (ga.k == 31) hipLaunchKernelGGL((k_graph<false, 31>), dim3(grid), dim3(GRAPH_THREADS), b->graph_lds, st, ga);
  else hipLaunchKernelGGL((k_graph<false, 0>), dim3(grid), dim3(GRAPH_THREADS), b->graph_lds, st, ga);
}

// Graph stage on one stream: pure-chain pass, then the general kernel for the rest.
static int launch_graph_fast(km_batch* b, hipStream_t st) {
  HIPCHK(hipMemsetAsync(b->d_counters.p, 0, (POOL_GROUPS * POOL_CTR_STRIDE + 16) * sizeof(unsigned long long), st));
  b->ga.use_need_full = 1;
  b->ga.dfs_answers = 0;                  // no k_dfs in this pass: every flagged target goes through k_graph
  HIPCHK(hipMemsetAsync(b->d_nflagged.p + 1, 0, sizeof(uint32_t), st));   // k_graph_pure appends its hand-overs again
  launch_pure(b, st, b->ga);
  launch_graph(b, st, b->ga);
  HIPCHK(hipGetLastError());
  return KM_OK;
}

static void launch_seed(uint32_t n_items, hipStream_t st, const WalkArgs& wa, bool count_fetches) {
  if (wa.stamps) hipLaunchKernelGGL((k_seed<true, 0, true>), dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);   // KM_SEED_STAMPS diagnostics
  else if (wa.tab.k == 31) {
    if (count_fetches) hipLaunchKernelGGL((k_seed<false, 31, true>), dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);
    else hipLaunchKernelGGL((k_seed<false, 31, false>), dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);
  } else {
    if (count_fetches) hipLaunchKernelGGL((k_seed<false, 0, true>), dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);
    else hipLaunchKernelGGL((k_seed<false, 0, false>), dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);
  }
}

// Compaction kernels + ONE asynchronous copy of region A and the expected part of the tail into
// the pinned twin; km_batch_result() waits for ev_out and fetches what the guess left behind.
static int enqueue_deliver(km_batch* b, hipStream_t st, bool lean, bool count16 = false) {
  const double h_in = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const uint32_t n = b->n_targets;
  b->lean = lean;
  b->count16 = count16;
  const OutLayout L = out_layout(n);
  b->result_ready = false;
  if (n == 0) {
    memset(b->h_out, 0, L.a_bytes + 64);
    reinterpret_cast<unsigned long long*>(b->h_out)[OT_TAIL_BYTES] = 16;
    reinterpret_cast<unsigned long long*>(b->h_out)[OT_SERIAL] = ++b->serial;
    b->copied_tail = 16;
    b->deliver_pending = true;
    HIPCHK(hipEventRecord(b->ev_out, st));
    return KM_OK;
  }
  OutArgs oa;
  memset(&oa, 0, sizeof oa);
  oa.n_targets = n;
  oa.ran_graph = (b->ran_graph && b->graph_mode == 0) ? 1u : 0u;
  oa.lean = lean ? 1u : 0u;
  oa.count16 = count16 ? 1u : 0u;
  oa.count_fetches = b->count_fetches ? 1u : 0u;
  oa.serial = ++b->serial;
  oa.big_ctl = b->big_entry ? b->d_big_ctl.p : nullptr;
  oa.big_slots = BIG_DEV_SLOTS;
  oa.status = b->d_status.p; oa.g_status = b->d_gstatus.p; oa.n_nodes = b->d_n_nodes.p; oa.n_ref = b->d_n_ref.p;
  oa.t_npaths = b->d_npaths.p; oa.t_pathbase = b->d_pathbase.p; oa.t_nruns = b->d_t_nruns.p;
  oa.t_refmax = b->d_t_refmax.p;
  oa.probes = reinterpret_cast<unsigned long long*>(b->d_probes.p);
  oa.dfs_probes = b->d_dfs_probes.p;
  oa.fetches = reinterpret_cast<unsigned long long*>(b->d_fetches.p);
  oa.pool_overflow = b->d_counters.p + POOL_GROUPS * POOL_CTR_STRIDE;
  oa.n_flagged = b->d_nflagged.p;
  oa.node_base = b->d_node_base.p; oa.node_kmer = b->d_node_kmer.p; oa.node_cnt = b->d_node_cnt.p;
  oa.p_runbase = b->d_p_runbase.p; oa.p_nruns = b->d_p_nruns.p; oa.p_len = b->d_p_len.p;
  oa.p_mincov = b->d_p_mincov.p; oa.r_start = b->d_r_start.p; oa.r_len = b->d_r_len.p;
  oa.loc = b->d_loc.p; oa.cnt = b->d_cnt4.p; oa.blk_tot = b->d_blk_tot.p; oa.psort = b->d_psort.p;
  oa.blk_base = b->d_blk_base.p; oa.scan_ticket = b->d_scan_ticket.p;
  // KM_DELIVER_ZEROCOPY=1: the delivery kernels store straight into the pinned host buffer
  // (PCIe writes from the CUs, no copy command on the stream); default: device buffer + one DMA
  const bool zero_copy = knobs().zero_copy;
  unsigned char* dst = zero_copy ? b->h_out : b->d_out;
  oa.totals = reinterpret_cast<unsigned long long*>(dst + L.totals);
  oa.o_status = reinterpret_cast<uint32_t*>(dst + L.status);
  oa.o_nref = reinterpret_cast<uint32_t*>(dst + L.n_ref);
  oa.o_probes = reinterpret_cast<uint64_t*>(dst + L.probes);
  oa.o_node_off = reinterpret_cast<uint64_t*>(dst + L.node_off);
  oa.o_extra_off = reinterpret_cast<uint64_t*>(dst + L.extra_off);
  oa.o_path_off = reinterpret_cast<uint32_t*>(dst + L.path_off);
  oa.o_refmax = reinterpret_cast<uint32_t*>(dst + L.ref_max);
  oa.o_esc_node = reinterpret_cast<uint64_t*>(dst + L.esc_node);
  oa.o_esc_value = reinterpret_cast<uint32_t*>(dst + L.esc_value);
  oa.tail = dst + L.a_bytes;
  oa.tail_cap = b->out_cap - L.a_bytes;
  const int dbg_deliver = knobs().debug_deliver;   // timing ablations (diagnostics build only)
  const bool host_trace = knobs().host_trace;      // diagnostics: host time of the calls below
  auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double h0 = host_trace ? now_us() : 0;
  if (!(dbg_deliver & 2)) {
    hipLaunchKernelGGL(k_out_scan, dim3((n + OUT_SCAN_THREADS - 1) / OUT_SCAN_THREADS), dim3(OUT_SCAN_THREADS), 0, st, oa);
    hipLaunchKernelGGL(k_out_pack, dim3(n), dim3(64), 0, st, oa);
  }
  HIPCHK(hipGetLastError());
  const double h1 = host_trace ? now_us() : 0;
  if (b->timed) HIPCHK(hipEventRecord(b->ev[5], st));
  uint64_t guess = std::min<uint64_t>(oa.tail_cap, b->tail_guess);
  if (zero_copy) guess = oa.tail_cap;            // everything is already where it belongs
  else if (!(dbg_deliver & 1)) {
    HIPCHK(hipMemcpyAsync(b->h_out, b->d_out, L.a_bytes + guess, hipMemcpyDeviceToHost, st));
  }
  const double h2 = host_trace ? now_us() : 0;
  if (b->timed) HIPCHK(hipEventRecord(b->ev[6], st));
  b->timed_deliver = b->timed;
  HIPCHK(hipEventRecord(b->ev_out, st));
  if (host_trace) fprintf(stderr, "[km host] deliver: kernels %.1f us, memcpyAsync %.1f us, event %.1f us, whole %.1f\n", h1 - h0, h2 - h1, now_us() - h2, now_us() - h_in);
  b->copied_tail = guess;
  b->deliver_pending = true;
  return KM_OK;
}

// The large tier of an earlier run moved some targets to bigger node storage: back to the layout of
// layout_targets (a step replayed on the same targets starts from the same state).  The device arrays are reset
// by k_pack itself (node_base0); this is the host's mirror of them.
static int restore_layout(km_batch* b, hipStream_t st) {
  (void)st;
  if (!b->layout_moved) return KM_OK;
  drop_graph(b);
  b->h_node_base = b->h_node_base0;
  b->h_node_cap = b->h_node_cap0;
  b->node_pool_used = b->node_pool0;
  b->layout_moved = false;
  return KM_OK;
}

static int launch_big_walk_dev(km_batch* b, hipStream_t st);
static int launch_big_graph_dev(km_batch* b, hipStream_t st);
static int ensure_bigdev_ws(km_batch* b);
static double host_now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
extern "C" int km_batch_run(km_batch_t* b, int stages, void* stream) {
  if (!b) return fail(KM_E_ARG, "null argument");
  const bool host_trace = knobs().host_trace;      // diagnostics: host time of the sections
  double ht[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ht[0] = host_trace ? host_now_us() : 0;
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = (hipStream_t)stream;
  b->last_stream = st;
  const bool want_graph = (stages & KM_RUN_HIPGRAPH) != 0;
  const bool want_deliver = (stages & KM_RUN_DELIVER) != 0;
  const bool want_lean = (stages & KM_DELIVER_LEAN) != 0;
  const bool want_c16 = (stages & KM_DELIVER_COUNT16) != 0;
  if (stages & KM_STAGE_WALK) b->count_fetches = (stages & KM_RUN_COUNT_FETCHES) != 0;
  const bool want_timed = (stages & KM_RUN_TIMED) != 0;
  const bool timed_fine = want_timed && !(stages & KM_RUN_TIMED_STAGES);
  const bool serial = (stages & KM_RUN_SERIAL) != 0;
  stages &= (KM_STAGE_WALK | KM_STAGE_GRAPH);
  b->deliver_pending = b->result_ready = false;
  b->timed_deliver = false;
  b->n_big = 0;
  if (!b->n_targets) {
    b->ran_walk = true;
    b->ran_graph = (stages & KM_STAGE_GRAPH) != 0;
    b->graph_mode = b->ran_graph ? 0 : 1;
    b->synced = true;
    return want_deliver ? enqueue_deliver(b, st, want_lean, want_c16) : KM_OK;
  }
  if (stages & KM_STAGE_WALK) {
    int rc = restore_layout(b, st);
    if (rc != KM_OK) return rc;
  }
  ht[1] = host_trace ? host_now_us() : 0;
  if (want_graph && !serial && b->gexec && b->graph_stages == stages && b->graph_stream == st) {
    HIPCHK(hipGraphLaunch(b->gexec, st));
    b->bigdev_ran = b->big_entry && b->bigdev_armed;      // (a flip of that state drops the captured step)
    b->ran_walk = true;
    b->ran_graph = true;
    b->synced = false;
    b->timed = false;
    return want_deliver ? enqueue_deliver(b, st, want_lean, want_c16) : KM_OK;
  }

  b->graph_mode = (stages & KM_STAGE_GRAPH) ? 0 : 1;
  fast_geometry(b);
  WalkArgs& wa = b->wa;
  GraphArgs& ga = b->ga;
  {
    int rc = b->d_frames.alloc((uint64_t)b->n_targets * wa.f_stride);
    if (rc != KM_OK) return rc;
    rc = ensure_bigdev_ws(b);
    if (rc != KM_OK) return rc;
  }
  wa.f_ws = b->d_frames.p;
  wa.stamps = nullptr;
  if (knobs().seed_stamps) {
    int rc = b->d_stamps.alloc(16ull * (SEED_BLOCK / 64) * (b->n_items + 4));
    if (rc != KM_OK) return rc;
    wa.stamps = b->d_stamps.p;
  }

  bool graph_launched = false;
  // a captured step has no host round trips inside
  b->timed = want_timed;
  b->timed_fine = timed_fine;
  const bool capturing = want_graph && st != nullptr && (stages & KM_STAGE_WALK);   // the NULL stream cannot be captured
  if (capturing) {
    b->timed = false;
    drop_graph(b);
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  }
  ht[2] = host_trace ? host_now_us() : 0;
  if (stages & KM_STAGE_WALK) {
    // the path-pool counters of the graph kernels (zeroed here, outside the timed walk stage)
    HIPCHK(hipMemsetAsync(b->d_counters.p, 0, (POOL_GROUPS * POOL_CTR_STRIDE + 16) * sizeof(unsigned long long), st));
    ht[3] = host_trace ? host_now_us() : 0;
    if (b->timed) HIPCHK(hipEventRecord(b->ev[0], st));
    hipLaunchKernelGGL(k_pack, dim3((b->n_targets + PACK_WAVES - 1) / PACK_WAVES), dim3(64 * PACK_WAVES), 0, st, wa);
    if (b->timed && timed_fine) HIPCHK(hipEventRecord(b->ev[3], st));
    if (b->n_items)
      launch_seed(b->n_items, st, wa, b->count_fetches);
    if (b->timed && timed_fine) HIPCHK(hipEventRecord(b->ev[4], st));
    // a batch's kernels run in ONE stream, in order (k_graph_pure after k_dfs): batches overlap with each
    // other, every launch stream on a hardware queue of its own (see "streams" above).  (Round 2 ran
    // k_graph_pure beside k_dfs on a side stream per batch; KM_RUN_SERIAL selected today's order.)
    ga.use_need_full = 1;
    (void)serial;
    // one single-wave block per FLAGGED target: the grid follows what the batch's last delivery reported (x 1.25 + 64;
    // the whole batch until one has been seen) — 6 000 of the headline batch's 10 000 blocks used to leave after one
    // load, each having claimed its LDS first.  More flagged targets than blocks: the kernel hands the rest to the
    // large tier (walk_kernel.h), and the next run's grid is larger.
    uint32_t dfs_grid = b->n_targets;
    if (b->flagged_seen != 0xFFFFFFFFu && !knobs().dfs_grid_full)
      dfs_grid = (uint32_t)std::min<uint64_t>(dfs_grid, (uint64_t)b->flagged_seen + b->flagged_seen / 4 + 64);
    auto launch_dfs = [&]() {
      if (wa.tab.k == 31) hipLaunchKernelGGL((k_dfs<false, 31>), dim3(dfs_grid), dim3(64), b->walk_lds, st, wa);
      else hipLaunchKernelGGL((k_dfs<false, 0>), dim3(dfs_grid), dim3(64), b->walk_lds, st, wa);
    };
    // diagnostics (KM_DFS_REPLAY=1|2): k_dfs twice, the SECOND launch is the one timed — its instruction
    // cache is warm; with 2 a 1 GiB memset in between flushes L2 / Infinity Cache (data cold again)
    const int dfs_replay = knobs().dfs_replay;
    if (dfs_replay) {
      launch_dfs();
      if (dfs_replay == 2) {
        static void* scratch = nullptr;
        if (!scratch) HIPCHK(hipMalloc(&scratch, 1ull << 30));
        HIPCHK(hipMemsetAsync(scratch, 0, 1ull << 30, st));
      }
      if (b->timed) HIPCHK(hipEventRecord(b->ev[4], st));
    }
    launch_dfs();
    HIPCHK(hipGetLastError());
    if (b->timed) HIPCHK(hipEventRecord(b->ev[1], st));
    {
      b->bigdev_ran = b->big_entry && b->bigdev_armed;
      int rc = launch_big_walk_dev(b, st);
      if (rc != KM_OK) return rc;
    }
    launch_pure(b, st, ga);
    launch_graph(b, st, ga);
    {
      int rc = launch_big_graph_dev(b, st);
      if (rc != KM_OK) return rc;
    }
    HIPCHK(hipGetLastError());
    graph_launched = true;
    b->ran_walk = true;
    b->ran_graph = false;
  } else if (!b->ran_walk) {
    return fail(KM_E_STATE, "graph stage requested before the walk stage");
  } else {
    if (b->timed) HIPCHK(hipEventRecord(b->ev[0], st));
    if (b->timed) HIPCHK(hipEventRecord(b->ev[3], st));
    if (b->timed) HIPCHK(hipEventRecord(b->ev[4], st));
    if (b->timed) HIPCHK(hipEventRecord(b->ev[1], st));
  }
  // the graph kernels also host the duplicate-k-mer check, so they always run
  // (graph_mode 1 = stop after that check)
  if (!graph_launched) {
    int rc = launch_graph_fast(b, st);
    if (rc != KM_OK) return rc;
  }
  b->ran_graph = true;                      // graph_mode says how far it went
  if (b->timed) HIPCHK(hipEventRecord(b->ev[2], st));
  if (capturing) {
    HIPCHK(hipStreamEndCapture(st, &b->graph));
    HIPCHK(hipGraphInstantiate(&b->gexec, b->graph, nullptr, nullptr, 0));
    b->graph_stages = stages;
    b->graph_stream = st;
    HIPCHK(hipGraphLaunch(b->gexec, st));
  }
  b->synced = false;
  if (host_trace) {
    ht[4] = host_now_us();
  }
  const int rc_deliver = want_deliver ? enqueue_deliver(b, st, want_lean, want_c16) : KM_OK;
  if (host_trace)
    fprintf(stderr, "[km host] run: setdevice+layout %.1f us, geometry %.1f, memset %.1f, launches %.1f, delivery %.1f, whole call %.1f\n",
            ht[1] - ht[0], ht[2] - ht[1], ht[3] - ht[2], ht[4] - ht[3], host_now_us() - ht[4], host_now_us() - ht[0]);
  return rc_deliver;
}

static int pull_status(km_batch* b, hipStream_t st) {
  const uint32_t n = b->n_targets;
  b->h_status.resize(n); b->h_gstatus.assign(n, 0); b->h_n_nodes.resize(n); b->h_n_ref.resize(n);
  b->h_npaths.assign(n, 0); b->h_pathbase.assign(n, 0);
  HIPCHK(hipMemcpyAsync(b->h_status.data(), b->d_status.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b->h_n_nodes.data(), b->d_n_nodes.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b->h_n_ref.data(), b->d_n_ref.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  if (b->ran_graph) {
    HIPCHK(hipMemcpyAsync(b->h_gstatus.data(), b->d_gstatus.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_npaths.data(), b->d_npaths.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_pathbase.data(), b->d_pathbase.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&b->h_overflow, b->d_counters.p + POOL_GROUPS * POOL_CTR_STRIDE, 8, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return KM_OK;
}

// Geometry of the large-tier walk (global-memory workspaces sized for the reference's own bound on a walk)
static int big_walk_geometry(km_batch* b, WalkArgs& a) {
  const int k = b->db->k;
  const uint32_t max_nref = b->max_len >= (uint32_t)k ? b->max_len - k + 1 : 1;
  const uint64_t max_nodes = std::max<uint64_t>(max_nref, (uint64_t)b->p.max_node + b->p.max_stack) + 1;
  const uint64_t hs = 2 * (max_nodes + b->p.max_stack + 64);
  if (hs > 0x7FFFFF00ull) return fail(KM_E_ARG, "node limit too large");
  a.hs_cap = round_up((uint32_t)hs, 64);
  a.pcap = walk_pcap(max_nref);
  a.words_cap = words_cap_for(b->max_len);
  a.fcap = round_up(b->p.max_stack + 2, 2);
  a.bcap = b->p.max_break + 1;
  a.g_stride = walk_ws_bytes(a.hs_cap, a.words_cap, a.fcap, a.bcap, a.pcap);
  return KM_OK;
}

// Workspaces of the device's own large tier (allocated before a step is launched or captured; they grow with the
// longest target of the batch)
static int ensure_bigdev_ws(km_batch* b) {
  if (!b->big_entry || !b->bigdev_armed) return KM_OK;
  WalkArgs a;
  memset(&a, 0, sizeof a);
  int rc = big_walk_geometry(b, a);
  if (rc != KM_OK) return rc;
  rc = b->d_bigdev_walk_ws.alloc((uint64_t)BIG_DEV_SLOTS * a.g_stride);
  if (rc != KM_OK) return rc;
  const uint32_t ncap = b->big_entry + 2, hcap = round_up(ncap + ncap / 2 + 1, 64);
  return b->d_bigdev_graph_ws.alloc((uint64_t)BIG_DEV_SLOTS * graph_ws_bytes<uint32_t>(ncap, hcap, words_cap_for(b->max_len)));
}

// The device's own large tier, walk: one more launch behind the fast k_dfs, in its stream — BIG_DEV_SLOTS single-wave
// blocks that leave at once unless the fast kernel appended targets to the list (WalkArgs::big_ctl).
static int launch_big_walk_dev(km_batch* b, hipStream_t st) {
  if (!b->big_entry || !b->bigdev_armed) return KM_OK;
  WalkArgs a;
  fill_walk_args(b, a);
  int rc = big_walk_geometry(b, a);
  if (rc != KM_OK) return rc;
  if (b->d_bigdev_walk_ws.n < (uint64_t)BIG_DEV_SLOTS * a.g_stride) return fail(KM_E_STATE, "large-tier workspace missing");
  a.g_ws = b->d_bigdev_walk_ws.p;
  a.list = b->d_big_walk.p;
  a.n_list_dev = b->d_big_ctl.p;
  a.n_list_host = 0;
  a.big_prep = 1;
  a.stamps = nullptr;
  hipLaunchKernelGGL((k_dfs<true, 0>), dim3(BIG_DEV_SLOTS), dim3(64), 0, st, a);
  return KM_OK;
}
// ... and graph: behind the fast k_graph, over what it (or the large-tier walk's results) could not hold
static int launch_big_graph_dev(km_batch* b, hipStream_t st) {
  if (!b->big_entry || !b->bigdev_armed) return KM_OK;
  GraphArgs g;
  fill_graph_args(b, g);
  g.ncap = b->big_entry + 2;
  g.hcap = round_up(g.ncap + g.ncap / 2 + 1, 64);
  g.words_cap = words_cap_for(b->max_len);
  g.g_stride = graph_ws_bytes<uint32_t>(g.ncap, g.hcap, g.words_cap);
  if (b->d_bigdev_graph_ws.n < (uint64_t)BIG_DEV_SLOTS * g.g_stride) return fail(KM_E_STATE, "large-tier workspace missing");
  g.g_ws = b->d_bigdev_graph_ws.p;
  g.tids = b->d_big_graph.p;
  g.tids_n = b->d_big_ctl.p + 1;
  hipLaunchKernelGGL((k_graph<true, 0>), dim3(BIG_DEV_SLOTS), dim3(GRAPH_THREADS), 0, st, g);
  return KM_OK;
}

// Large tier: rerun the listed targets with global-memory workspaces.
static int run_big_walk(km_batch* b, const std::vector<uint32_t>& ids, hipStream_t st) {
  const uint32_t nb = (uint32_t)ids.size();
  const int k = b->db->k;
  drop_graph(b);                      // a captured step holds the addresses that change below
  b->layout_moved = true;
  if (b->big_entry) {                 // the device's own large tier may have re-homed targets of this run
    HIPCHK(hipMemcpyAsync(b->h_node_base.data(), b->d_node_base.p, (uint64_t)b->n_targets * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_node_cap.data(), b->d_node_cap.p, (uint64_t)b->n_targets * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  // per-target node storage big enough for the reference's own bound
  uint64_t extra = 0;
  std::vector<uint64_t> old_base;
  for (uint32_t t : ids) old_base.push_back(b->h_node_base[t]);
  for (uint32_t t : ids) {
    const uint64_t L = b->h_toff[t + 1] - b->h_toff[t];
    const uint32_t n_ref = (L >= (uint64_t)k) ? (uint32_t)(L - k + 1) : 0;
    const uint64_t cap = std::max<uint64_t>(n_ref, (uint64_t)b->p.max_node + b->p.max_stack) + 1;
    if (cap > 0x7FFFFFFFull) return fail(KM_E_ARG, "node limit too large");
    b->h_node_base[t] = b->node_pool_used + extra;
    b->h_node_cap[t] = (uint32_t)cap;
    extra += cap;
  }
  const uint64_t need = b->node_pool_used + extra;
  if (need > b->d_node_kmer.n) {
    // grow the pools, keeping the fast-tier results
    DevBuf<uint64_t> nk; DevBuf<uint32_t> nc;
    int rc = nk.alloc(need); if (rc != KM_OK) return rc;
    rc = nc.alloc(need); if (rc != KM_OK) { nk.release(); return rc; }
    HIPCHK(hipMemcpyAsync(nk.p, b->d_node_kmer.p, b->node_pool_used * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(nc.p, b->d_node_cnt.p, b->node_pool_used * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    b->d_node_kmer.release(); b->d_node_cnt.release();
    b->d_node_kmer = nk; b->d_node_cnt = nc;
  }
  b->node_pool_used = need;
  // the seed kernel's results (the counts of the target's own k-mers) move to the new storage
  for (size_t q = 0; q < ids.size(); ++q) {
    const uint32_t t = ids[q];
    const uint64_t nref = b->h_n_ref[t];
    if (!nref) continue;
    HIPCHK(hipMemcpyAsync(b->d_node_cnt.p + b->h_node_base[t], b->d_node_cnt.p + old_base[q], nref * 4,
                          hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(hipMemcpyAsync(b->d_node_base.p, b->h_node_base.data(), (uint64_t)b->n_targets * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_node_cap.p, b->h_node_cap.data(), (uint64_t)b->n_targets * 4, hipMemcpyHostToDevice, st));
  int rc = b->d_big_ids.alloc(nb); if (rc != KM_OK) return rc;
  HIPCHK(hipMemcpyAsync(b->d_big_ids.p, ids.data(), (uint64_t)nb * 4, hipMemcpyHostToDevice, st));

  WalkArgs a;
  fill_walk_args(b, a);
  a.n_list_dev = nullptr;
  a.big_ctl = nullptr;                // (this pass IS the fallback)
  rc = big_walk_geometry(b, a);
  if (rc != KM_OK) return rc;
  // run in slices so the workspace stays bounded
  const uint64_t budget = 8ull << 30;
  uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nb, budget / a.g_stride));
  rc = b->d_big_ws.alloc((uint64_t)per * a.g_stride); if (rc != KM_OK) return rc;
  a.g_ws = b->d_big_ws.p;
  for (uint32_t s = 0; s < nb; s += per) {
    const uint32_t cnt = std::min(per, nb - s);
    a.list = b->d_big_ids.p + s;
    a.n_list_host = cnt;
    hipLaunchKernelGGL((k_dfs<true, 0>), dim3(cnt), dim3(64), 0, st, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
  }
  return KM_OK;
}

static int run_big_graph(km_batch* b, const std::vector<uint32_t>& ids, hipStream_t st) {
  const uint32_t nb = (uint32_t)ids.size();
  int rc = b->d_big_ids.alloc(nb); if (rc != KM_OK) return rc;
  HIPCHK(hipMemcpyAsync(b->d_big_ids.p, ids.data(), (uint64_t)nb * 4, hipMemcpyHostToDevice, st));
  uint32_t max_nodes = 0;
  for (uint32_t t : ids) max_nodes = std::max(max_nodes, b->h_n_nodes[t]);
  GraphArgs g;
  fill_graph_args(b, g);
  g.ncap = max_nodes + 2;
  g.hcap = round_up(g.ncap + g.ncap / 2 + 1, 64);
  g.words_cap = words_cap_for(b->max_len);
  g.g_stride = graph_ws_bytes<uint32_t>(g.ncap, g.hcap, g.words_cap);
  const uint64_t budget = 8ull << 30;
  uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nb, budget / g.g_stride));
  rc = b->d_big_ws.alloc((uint64_t)per * g.g_stride); if (rc != KM_OK) return rc;
  g.g_ws = b->d_big_ws.p;
  for (uint32_t s = 0; s < nb; s += per) {
    const uint32_t cnt = std::min(per, nb - s);
    g.tids = b->d_big_ids.p + s;
    hipLaunchKernelGGL((k_graph<true, 0>), dim3(cnt), dim3(GRAPH_THREADS), 0, st, g);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
  }
  return KM_OK;
}

static int grow_path_pools(km_batch* b) {
  drop_graph(b);                      // a captured step holds the old pool addresses and sizes
  b->path_pool = (b->path_pool * 4 / POOL_GROUPS + 1) * POOL_GROUPS;
  b->run_pool = (b->run_pool * 4 / POOL_GROUPS + 1) * POOL_GROUPS;
  int rc = KM_OK;
  auto A = [&](int r) { if (rc == KM_OK) rc = r; };
  A(b->d_p_target.alloc(b->path_pool)); A(b->d_p_runbase.alloc(b->path_pool));
  A(b->d_p_nruns.alloc(b->path_pool)); A(b->d_p_len.alloc(b->path_pool));
  A(b->d_p_mincov.alloc(b->path_pool)); A(b->d_psort.alloc(b->path_pool));
  A(b->d_r_start.alloc(b->run_pool)); A(b->d_r_len.alloc(b->run_pool));
  return rc;
}

static int relaunch_fast_graph(km_batch* b, hipStream_t st) {
  // same geometry as the run, new pool sizes / addresses
  const GraphArgs old = b->ga;
  fill_graph_args(b, b->ga);
  b->ga.ncap = old.ncap; b->ga.hcap = old.hcap; b->ga.words_cap = old.words_cap; b->ga.hcap_pure = old.hcap_pure;
  int rc = launch_graph_fast(b, st);
  if (rc != KM_OK) return rc;
  HIPCHK(hipStreamSynchronize(st));
  return KM_OK;
}

static void read_timings(km_batch* b) {
  for (float& v : b->ms) v = 0.0f;
  if (!b->timed) return;
  (void)hipEventElapsedTime(&b->ms[0], b->ev[0], b->ev[1]);
  (void)hipEventElapsedTime(&b->ms[1], b->ev[1], b->ev[2]);
  (void)hipEventElapsedTime(&b->ms[2], b->ev[0], b->ev[2]);
  if (b->timed_fine) {
    (void)hipEventElapsedTime(&b->ms[3], b->ev[3], b->ev[4]);
    (void)hipEventElapsedTime(&b->ms[4], b->ev[0], b->ev[3]);
    (void)hipEventElapsedTime(&b->ms[5], b->ev[4], b->ev[1]);
  }
  if (b->timed_deliver) {
    (void)hipEventElapsedTime(&b->ms[6], b->ev[2], b->ev[5]);
    (void)hipEventElapsedTime(&b->ms[7], b->ev[5], b->ev[6]);
  }
  (void)hipGetLastError();
}

// Wait for the launched kernels, then finish the rare work that needs the host:
// targets that outgrew the LDS-resident tier are rerun with global workspaces,
// and the path pools are enlarged if they overflowed.
extern "C" int km_batch_sync(km_batch_t* b) {
  if (!b) return fail(KM_E_ARG, "null argument");
  if (b->synced) return KM_OK;
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = b->last_stream;
  HIPCHK(hipStreamSynchronize(st));
  read_timings(b);
  if (!b->n_targets) { b->synced = true; return KM_OK; }
  int rc = pull_status(b, st);
  if (rc != KM_OK) return rc;
  const uint32_t n = b->n_targets;

  std::vector<uint32_t> big;
  for (uint32_t t = 0; t < n; ++t) if (b->h_status[t] == T_NEEDS_BIG) big.push_back(t);
  b->n_big = (uint32_t)big.size();
  if (b->big_entry && !big.empty() && !b->bigdev_armed) { b->bigdev_armed = true; drop_graph(b); }
  std::vector<char> force_big(n, 0);
  bool changed = false;
  if (!big.empty()) {
    rc = run_big_walk(b, big, st);
    if (rc != KM_OK) return rc;
    for (uint32_t t : big) force_big[t] = 1;     // the fast graph pass skipped them
    rc = pull_status(b, st);
    if (rc != KM_OK) return rc;
    changed = true;
  }
  if (b->ran_graph) {
    for (int pass = 0;; ++pass) {
      if (pass > 0) {
        if (pass > 8) return fail(KM_E_NOMEM, "path pools keep overflowing");
        rc = grow_path_pools(b);
        if (rc != KM_OK) return rc;
        rc = relaunch_fast_graph(b, st);
        if (rc != KM_OK) return rc;
        rc = pull_status(b, st);
        if (rc != KM_OK) return rc;
        std::fill(force_big.begin(), force_big.end(), 0);   // the relaunch saw their final walk status
        changed = true;
      }
      std::vector<uint32_t> todo;
      for (uint32_t t = 0; t < n; ++t)
        if (b->h_status[t] == T_OK && (force_big[t] || b->h_gstatus[t] == T_NEEDS_BIG)) todo.push_back(t);
      if (!todo.empty()) {
        rc = run_big_graph(b, todo, st);
        if (rc != KM_OK) return rc;
        rc = pull_status(b, st);
        if (rc != KM_OK) return rc;
        changed = true;
      }
      if (!b->h_overflow) break;
    }
  }
  if (changed) b->deliver_pending = b->result_ready = false;   // any earlier delivery is stale
  b->synced = true;
  return KM_OK;
}

// Wait for an event; KM_SPIN_US=n polls it for the first n microseconds instead of putting the
// thread to sleep at once (default 0: on the boxes measured a polling consumer gained nothing,
// 0.315 against 0.312 ms per delivered step).
static hipError_t wait_event_hot(hipEvent_t ev) {
  const long spin_us = knobs().spin_us;
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipEventQuery(ev);
      if (e != hipErrorNotReady) return e;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
  }
  return hipEventSynchronize(ev);
}

// Results of the last run in the pinned delivery buffer (delivering now if the run did not).
// `need_full`: a lean delivery (pending or ready) is replaced by a full one.
static int finish_result(km_batch* b, bool need_full) {
  if (!b->ran_walk) return fail(KM_E_STATE, "nothing has run yet");
  const bool partial = b->lean || b->count16;       // the pending / ready delivery is not the full 32-bit form
  if (b->result_ready && !(need_full && partial)) return KM_OK;
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = b->last_stream;
  if (need_full && partial && (b->deliver_pending || b->result_ready)) {
    HIPCHK(hipEventSynchronize(b->ev_out));
    b->deliver_pending = b->result_ready = false;
  }
  if (!b->deliver_pending) {
    int rc = km_batch_sync(b);
    if (rc != KM_OK) return rc;
    rc = enqueue_deliver(b, st, false);
    if (rc != KM_OK) return rc;
  }
  const OutLayout L = out_layout(b->n_targets);
  const unsigned long long* T = reinterpret_cast<const unsigned long long*>(b->h_out + L.totals);
  for (int attempt = 0;; ++attempt) {
    HIPCHK(wait_event_hot(b->ev_out));
    if (knobs().debug_deliver) {                      // timing ablation (diagnostics build only): nothing valid arrived
      b->deliver_pending = false; b->result_ready = true;
      return KM_OK;
    }
    if (T[OT_SERIAL] != b->serial) return fail(KM_E_HIP, "delivery buffer out of step");
    const unsigned long long nh = T[OT_NEEDS_HOST];
    {
      const uint32_t fl = (uint32_t)std::min<unsigned long long>(T[OT_N_FLAGGED], 0x7FFFFFFFull);
      if (b->gexec && b->flagged_seen != 0xFFFFFFFFu && (fl > b->flagged_seen + b->flagged_seen / 8 + 32 || 4 * fl + 256 < b->flagged_seen)) drop_graph(b);
      b->flagged_seen = fl;
    }
    if (b->ran_graph && b->graph_mode == 0) {
      const uint32_t seen = (uint32_t)std::min<unsigned long long>(T[OT_N_GRAPH_LIST], 0x7FFFFFFFull);
      // (a captured step holds its grid: it is dropped when the list outgrows a quarter of it or shrinks to a 16th)
      if (b->gexec && b->graph_list_seen != 0xFFFFFFFFu && (seen > 2 * b->graph_list_seen + 16 || 16 * seen + 64 < b->graph_list_seen)) drop_graph(b);
      b->graph_list_seen = seen;
    }
    if (b->big_entry && !b->bigdev_armed && ((nh & 1ull) || T[OT_N_BIG_DEV])) {
      b->bigdev_armed = true;              // from the next run on, the device's own large tier is launched
      drop_graph(b);                       // (a captured step does not contain its launches)
    }
    if (!nh && b->count16 && T[OT_N_ESC] > OUT_ESC_CAP) {
      // more counts >= 65535 than the escape list holds: this batch is delivered with 32-bit counts
      if (attempt >= 4) return fail(KM_E_NOMEM, "result delivery keeps failing");
      int rc = enqueue_deliver(b, st, b->lean, false);
      if (rc != KM_OK) return rc;
      continue;
    }
    if (!nh) break;
    if (attempt >= 4) return fail(KM_E_NOMEM, "result delivery keeps failing");
    if (nh & 1ull) {
      b->synced = false;
      int rc = km_batch_sync(b);
      if (rc != KM_OK) return rc;
    }
    if (nh == 2ull) {
      // everything is final, only the tail is larger than the buffer
      HIPCHK(hipStreamSynchronize(st));
      int rc = ensure_out(b, T[OT_TAIL_BYTES] + 4096);
      if (rc != KM_OK) return rc;
    } else {
      HIPCHK(hipStreamSynchronize(st));
      int rc = ensure_out(b, default_tail_bytes(b, b->node_pool_used, b->node_pool_used));
      if (rc != KM_OK) return rc;
    }
    T = reinterpret_cast<const unsigned long long*>(b->h_out + L.totals);
    int rc = enqueue_deliver(b, st, b->lean, b->count16);
    if (rc != KM_OK) return rc;
  }
  const uint64_t tail = T[OT_TAIL_BYTES];
  if (tail > b->copied_tail)
    HIPCHK(hipMemcpy(b->h_out + L.a_bytes + b->copied_tail, b->d_out + L.a_bytes + b->copied_tail,
                     tail - b->copied_tail, hipMemcpyDeviceToHost));
  b->tail_guess = tail + tail / 16 + 4096;
  if (b->count16 && T[OT_N_ESC] > 1) {
    // the escape list in node order (the delivery kernel appends as its waves come)
    const uint32_t ne = (uint32_t)T[OT_N_ESC];
    uint64_t* en = reinterpret_cast<uint64_t*>(b->h_out + L.esc_node);
    uint32_t* ev = reinterpret_cast<uint32_t*>(b->h_out + L.esc_value);
    std::vector<std::pair<uint64_t, uint32_t>> tmp(ne);
    for (uint32_t i = 0; i < ne; ++i) tmp[i] = {en[i], ev[i]};
    std::sort(tmp.begin(), tmp.end());
    for (uint32_t i = 0; i < ne; ++i) { en[i] = tmp[i].first; ev[i] = tmp[i].second; }
  }
  b->deliver_pending = false;
  b->result_ready = true;
  return KM_OK;
}

static void view_of_result(const km_batch* b, km_batch_out_t* v) {
  const OutLayout L = out_layout(b->n_targets);
  unsigned char* h = b->h_out;
  const unsigned long long* T = reinterpret_cast<const unsigned long long*>(h + L.totals);
  unsigned char* tail = h + L.a_bytes;
  memset(v, 0, sizeof *v);
  v->status = reinterpret_cast<uint32_t*>(h + L.status);
  v->n_ref = reinterpret_cast<uint32_t*>(h + L.n_ref);
  v->probes = reinterpret_cast<uint64_t*>(h + L.probes);
  v->node_off = reinterpret_cast<uint64_t*>(h + L.node_off);
  v->extra_off = reinterpret_cast<uint64_t*>(h + L.extra_off);
  v->path_off = reinterpret_cast<uint32_t*>(h + L.path_off);
  v->ref_max_cov = reinterpret_cast<uint32_t*>(h + L.ref_max);
  if (b->count16) {
    v->node_count16 = reinterpret_cast<uint16_t*>(tail + T[OT_OFF_COUNT]);
    v->count_esc_node = reinterpret_cast<uint64_t*>(h + L.esc_node);
    v->count_esc_value = reinterpret_cast<uint32_t*>(h + L.esc_value);
  } else {
    v->node_count = reinterpret_cast<uint32_t*>(tail + T[OT_OFF_COUNT]);
  }
  v->extra_kmer = reinterpret_cast<uint64_t*>(tail + T[OT_OFF_EXTRA]);
  v->path_len = reinterpret_cast<uint32_t*>(tail + T[OT_OFF_PLEN]);
  v->path_min_cov = reinterpret_cast<uint32_t*>(tail + T[OT_OFF_PMIN]);
  v->run_off = reinterpret_cast<uint64_t*>(tail + T[OT_OFF_RUNOFF]);
  v->run_start = reinterpret_cast<uint32_t*>(tail + T[OT_OFF_RSTART]);
  v->run_len = reinterpret_cast<uint32_t*>(tail + T[OT_OFF_RLEN]);
}

static void sizes_of_result(const km_batch* b, km_batch_sizes_t* s) {
  const unsigned long long* T = reinterpret_cast<const unsigned long long*>(b->h_out + out_layout(b->n_targets).totals);
  memset(s, 0, sizeof *s);
  s->n_targets = b->n_targets;
  s->n_paths = (uint32_t)T[OT_N_PATHS];
  s->n_nodes = T[OT_N_NODES];
  s->n_runs = T[OT_N_RUNS];
  s->n_extra = T[OT_N_EXTRA];
  s->logical_probes = T[OT_PROBES];
  s->table_fetches = T[OT_FETCHES];
  s->n_big_tier = b->n_big + (b->bigdev_ran ? (uint32_t)T[OT_N_BIG_DEV] : 0u);
  s->n_flagged = (uint32_t)T[OT_N_FLAGGED];
  s->seed_probes = T[OT_SEED_PROBES];
  s->n_count_escapes = b->count16 ? (uint32_t)T[OT_N_ESC] : 0;
}

extern "C" int km_batch_result(km_batch_t* b, km_batch_out_t* view, km_batch_sizes_t* sizes) {
  if (!b) return fail(KM_E_ARG, "null argument");
  int rc = finish_result(b, false);
  if (rc != KM_OK) return rc;
  if (view) view_of_result(b, view);
  if (sizes) sizes_of_result(b, sizes);
  return KM_OK;
}

// `steps` runs over `n` batches in flight, round robin: before a batch is run again its last
// delivery is awaited (km_batch_result), at the end every batch's.  The loop a pipelined consumer
// writes, kept on the library's side of the ABI so that an interpreter between two launches does
// not sit in the timed region (diagnostics / bench; tools/launch_cost.py).
extern "C" int km_batch_pump(km_batch_t* const* bs, void* const* streams, int n, int steps, int stages) {
  if (!bs || n <= 0 || steps < 0) return fail(KM_E_ARG, "bad argument");
  const bool deliver = (stages & KM_RUN_DELIVER) != 0;
  for (int i = 0; i < steps; ++i) {
    km_batch_t* b = bs[i % n];
    if (i >= n && deliver) {
      int rc = finish_result(b, false);
      if (rc != KM_OK) return rc;
    }
    int rc = km_batch_run(b, stages, streams ? streams[i % n] : nullptr);
    if (rc != KM_OK) return rc;
  }
  for (int q = 0; q < std::min(n, steps); ++q) {
    int rc = deliver ? finish_result(bs[q], false) : km_batch_sync(bs[q]);
    if (rc != KM_OK) return rc;
  }
  return KM_OK;
}

extern "C" int km_batch_debug_stamps(km_batch_t* b, uint64_t* dst, uint64_t cap_words, uint64_t* n_words) {
  if (!b || !n_words) return fail(KM_E_ARG, "null argument");
  int rc = km_batch_sync(b);
  if (rc != KM_OK) return rc;
  const uint64_t n = b->d_stamps.p ? 16ull * (SEED_BLOCK / 64) * b->n_items : 0;
  *n_words = n;
  if (!dst || !n) return KM_OK;
  if (cap_words < n) return fail(KM_E_CAPACITY, "stamp buffer too small");
  HIPCHK(hipMemcpy(dst, b->d_stamps.p, n * 8, hipMemcpyDeviceToHost));
  return KM_OK;
}

// ---- measurement helpers for consumers that hold no device buffers of their own (bench.py at N = 1
// runs without PyTorch in the process: the library is then served by the ROCm installation's HIP runtime,
// as it is for a C consumer)
extern "C" int km_device_sync(int device) {
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  return KM_OK;
}

// Device-to-device copy of `bytes` bytes, `reps` times: read + write bandwidth in GB/s (the box's
// large-copy rate beside the 8 TB/s spec).
extern "C" int km_device_copy_GBs(int device, uint64_t bytes, int reps, double* gbs) {
  if (!gbs || !bytes || reps < 1) return fail(KM_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(device));
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess) return fail(KM_E_NOMEM, "hipMalloc failed");
  if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return fail(KM_E_NOMEM, "hipMalloc failed"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps && e == hipSuccess; ++i) e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (e != hipSuccess) return fail(KM_E_HIP, "copy bandwidth measurement failed: %s", hipGetErrorString(e));
  *gbs = 2.0 * (double)reps * (double)bytes / ((double)ms * 1e-3) / 1e9;
  return KM_OK;
}

// k_query and k_children alone over `n` k-mers given on the host: average kernel time over `reps`
// launches each (HIP events), and how many of the k-mers have count 0.
extern "C" int km_probe_bench(kmjf_t* h, const uint64_t* kmers, uint64_t n, int reps, double ratio, int64_t n_cutoff,
                              double* query_ms, double* children_ms, uint64_t* n_zero) {
  if (!h || !kmers || !n || reps < 1 || !query_ms || !children_ms) return fail(KM_E_ARG, "bad argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
  HIPCHK(hipSetDevice(h->device));
  uint64_t* dk = nullptr;
  uint32_t* dq = nullptr;
  uint8_t* dm = nullptr;
  uint32_t* dc = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  int rc = KM_OK;
  hipError_t e = hipMalloc((void**)&dk, n * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&dq, n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&dm, n);
  if (e == hipSuccess) e = hipMalloc((void**)&dc, n * 16);
  for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&ev[i]);
  if (e == hipSuccess) e = hipMemcpy(dk, kmers, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    for (int w = 0; w < 2 && rc == KM_OK; ++w) {
      rc = kmjf_query_batch_dev(h, dk, n, dq, nullptr);
      if (rc == KM_OK) rc = kmjf_children_batch_dev(h, dk, n, ratio, n_cutoff, 1, dm, dc, nullptr);
    }
    if (rc == KM_OK) e = hipDeviceSynchronize();
    if (rc == KM_OK && e == hipSuccess) e = hipEventRecord(ev[0], nullptr);
    for (int i = 0; i < reps && rc == KM_OK; ++i) rc = kmjf_query_batch_dev(h, dk, n, dq, nullptr);
    if (rc == KM_OK && e == hipSuccess) e = hipEventRecord(ev[1], nullptr);
    for (int i = 0; i < reps && rc == KM_OK; ++i) rc = kmjf_children_batch_dev(h, dk, n, ratio, n_cutoff, 1, dm, dc, nullptr);
    if (rc == KM_OK && e == hipSuccess) e = hipEventRecord(ev[2], nullptr);
    if (rc == KM_OK && e == hipSuccess) e = hipEventSynchronize(ev[2]);
    float q = 0, c = 0;
    if (rc == KM_OK && e == hipSuccess) e = hipEventElapsedTime(&q, ev[0], ev[1]);
    if (rc == KM_OK && e == hipSuccess) e = hipEventElapsedTime(&c, ev[1], ev[2]);
    *query_ms = q / reps;
    *children_ms = c / reps;
    if (rc == KM_OK && e == hipSuccess && n_zero) {
      std::vector<uint32_t> hq(n);
      e = hipMemcpy(hq.data(), dq, n * 4, hipMemcpyDeviceToHost);
      uint64_t z = 0;
      for (uint32_t v : hq) z += v == 0;
      *n_zero = z;
    }
  }
  for (int i = 0; i < 3; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
  if (dk) (void)hipFree(dk);
  if (dq) (void)hipFree(dq);
  if (dm) (void)hipFree(dm);
  if (dc) (void)hipFree(dc);
  if (rc != KM_OK) return rc;
  if (e != hipSuccess) return fail(KM_E_HIP, "probe benchmark failed: %s", hipGetErrorString(e));
  return KM_OK;
}

// Diagnostics: the device counters of the last run — [0] flagged targets (k_seed), [1] unflagged
// targets k_graph_pure handed to k_graph, [2] flagged targets the epilogue of k_dfs left to k_graph.
// What the reference logs with -v from inside the walk and the graph (km/utils/MutationFinder.py:160-161,
// km/utils/Graph.py:198, 231), for the last run: per target the reference edges stripped and the edges kept, and the
// walk's loop breaks as {target, node index} pairs in walk order.  Any output may be NULL.
extern "C" int km_batch_graph_log(km_batch_t* b, uint32_t* removed_ref_edges, uint32_t* nonref_edges,
                                  uint32_t* n_loop_breaks, uint32_t* loop_pairs, uint32_t loop_cap) {
  if (!b) return fail(KM_E_ARG, "null argument");
  if (!b->ran_walk) return fail(KM_E_STATE, "no run to report on");
  if (b->deliver_pending || b->result_ready) {          // a delivered run: finish it (large tier, pools) first
    int rc = finish_result(b, false);
    if (rc != KM_OK) return rc;
  } else {
    int rc = km_batch_sync(b);
    if (rc != KM_OK) return rc;
  }
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = b->last_stream;
  HIPCHK(hipStreamSynchronize(st));
  const uint32_t n = b->n_targets;
  if (removed_ref_edges && n) HIPCHK(hipMemcpy(removed_ref_edges, b->d_t_eremoved.p, 4ull * n, hipMemcpyDeviceToHost));
  if (nonref_edges && n) HIPCHK(hipMemcpy(nonref_edges, b->d_t_enonref.p, 4ull * n, hipMemcpyDeviceToHost));
  uint32_t n_loops = 0;
  if (n) HIPCHK(hipMemcpy(&n_loops, b->d_loop_ctl.p, 4, hipMemcpyDeviceToHost));
  if (n_loop_breaks) *n_loop_breaks = n_loops;
  const uint32_t have = std::min<uint32_t>(std::min<uint32_t>(n_loops, LOOP_LOG_CAP), loop_cap);
  if (loop_pairs && have) HIPCHK(hipMemcpy(loop_pairs, b->d_loop_list.p, 8ull * have, hipMemcpyDeviceToHost));
  return KM_OK;
}

extern "C" int km_batch_debug_counts(km_batch_t* b, uint32_t* out4) {
  if (!b || !out4) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->last_stream));
  HIPCHK(hipMemcpy(out4, b->d_nflagged.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return KM_OK;
}

extern "C" int km_batch_timings(km_batch_t* b, float* ms8) {
  float* ms3 = ms8;
  if (!b || !ms3) return fail(KM_E_ARG, "null argument");
  if (!b->synced) {
    // timing events only: no status pull, no large tier (finish_result does that when asked)
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->last_stream));
    read_timings(b);
  }
  for (int i = 0; i < 8; ++i) ms3[i] = b->ms[i];
  return KM_OK;
}

extern "C" int km_batch_sizes(km_batch_t* b, km_batch_sizes_t* s) {
  if (!b || !s) return fail(KM_E_ARG, "null argument");
  int rc = finish_result(b, true);          // the sizes km_batch_fetch fills: a full delivery
  if (rc != KM_OK) return rc;
  sizes_of_result(b, s);
  return KM_OK;
}

// Copying variant of km_batch_result: fills caller-allocated arrays (sizes from km_batch_sizes).
// node_kmer, when asked for, is rebuilt here: the target's own k-mers from the packed targets,
// the walk-discovered ones from extra_kmer.
extern "C" int km_batch_fetch(km_batch_t* b, const km_batch_out_t* out) {
  if (!b || !out) return fail(KM_E_ARG, "null argument");
  int rc = finish_result(b, true);          // the copying API always returns every node
  if (rc != KM_OK) return rc;
  km_batch_out_t v;
  km_batch_sizes_t s;
  view_of_result(b, &v);
  sizes_of_result(b, &s);
  const uint32_t n = b->n_targets;
  if (out->status) memcpy(out->status, v.status, 4ull * n);
  if (out->aux) memset(out->aux, 0, 4ull * n);
  if (out->n_ref) memcpy(out->n_ref, v.n_ref, 4ull * n);
  if (out->probes) memcpy(out->probes, v.probes, 8ull * n);
  if (out->node_off) memcpy(out->node_off, v.node_off, 8ull * (n + 1));
  if (out->extra_off) memcpy(out->extra_off, v.extra_off, 8ull * (n + 1));
  if (out->node_count) memcpy(out->node_count, v.node_count, 4 * s.n_nodes);
  if (out->extra_kmer) memcpy(out->extra_kmer, v.extra_kmer, 8 * s.n_extra);
  if (out->path_off) memcpy(out->path_off, v.path_off, 4ull * (n + 1));
  if (out->ref_max_cov) memcpy(out->ref_max_cov, v.ref_max_cov, 4ull * n);
  if (out->run_off) memcpy(out->run_off, v.run_off, 8ull * (s.n_paths + 1));
  if (out->run_start) memcpy(out->run_start, v.run_start, 4 * s.n_runs);
  if (out->run_len) memcpy(out->run_len, v.run_len, 4 * s.n_runs);
  if (out->path_len) memcpy(out->path_len, v.path_len, 4ull * s.n_paths);
  if (out->path_min_cov) memcpy(out->path_min_cov, v.path_min_cov, 4ull * s.n_paths);
  if (out->node_kmer && n) {
    HIPCHK(hipSetDevice(b->device));
    if (b->h_packed.empty()) {
      b->h_packed.resize(b->h_woff[n]);
      HIPCHK(hipMemcpy(b->h_packed.data(), b->d_packed.p, b->h_woff[n] * 8, hipMemcpyDeviceToHost));
    }
    const int k = b->db->k;
    for (uint32_t t = 0; t < n; ++t) {
      const uint64_t a0 = v.node_off[t], cnt = v.node_off[t + 1] - a0;
      if (!cnt) continue;
      const uint64_t ne = v.extra_off[t + 1] - v.extra_off[t], nr = cnt - ne;
      const uint64_t* words = b->h_packed.data() + b->h_woff[t];
      uint64_t* dst = out->node_kmer + a0;
      for (uint64_t i = 0; i < nr; ++i) {
        const uint64_t w = i >> 5, sh = (i & 31) * 2;
        const uint64_t x = sh ? ((words[w] << sh) | (words[w + 1] >> (64 - sh))) : words[w];
        dst[i] = x >> (64 - 2 * k);
      }
      memcpy(dst + nr, v.extra_kmer + v.extra_off[t], 8 * ne);
    }
  }
  return KM_OK;
}
