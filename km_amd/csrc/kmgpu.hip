// kmgpu.hip — libkmgpu.so: C-ABI (include/kmgpu.h) over the HIP kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/kmgpu.h"
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

extern "C" int km_stream_create(int device, void** stream) {
  if (!stream) return fail(KM_E_ARG, "null argument");
  HIPCHK(hipSetDevice(device));
  hipStream_t st = nullptr;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = st;
  return KM_OK;
}
extern "C" int km_stream_destroy(void* stream) {
  if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
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
  while ((uint64_t)n_buckets * 2 < max_entries && n_buckets < (1u << 30)) n_buckets <<= 1;
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
  auto bail = [&](int code, const char* what, hipError_t e) {
    if (dir) (void)hipFree(dir);
    if (caps) (void)hipFree(caps);
    if (sums) (void)hipFree(sums);
    if (d_meta) (void)hipFree(d_meta);
    if (slots) (void)hipFree(slots);
    if (ovf) (void)hipFree(ovf);
    if (ctr_ptr && *ctr_ptr) (void)hipFree(*ctr_ptr);
    return fail(code, "%s: %s", what, hipGetErrorString(e));
  };
  hipError_t e = hipMalloc((void**)&dir, dir_words * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&caps, dir_words * 4);
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
  const int MAX_ROUNDS = 4;             // CAP_MAX_GEN doublings + 1, then one final round just in case
  unsigned long long meta[6] = {0, 0, 0, 0, 0, 0};
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
                           (uint64_t)n_buckets, tv.cshift);
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
    if (n)
      hipLaunchKernelGGL(k_table_insert, dim3(grid_for(n, 256)), dim3(256), 0, st, tv, slots, d_keys,
                         d_counts, n, caps, final_round, d_meta);
    e = hipMemcpyAsync(meta, d_meta, 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(KM_E_HIP, "table build failed", e);
    if (meta[1] & 0xFFFFFFFFull) return bail(KM_E_HIP, "table build overflowed", hipSuccess);
    max_probe = std::max<uint32_t>(2, (uint32_t)meta[3] + 1);
    if (getenv("KM_BUILD_VERBOSE"))
      fprintf(stderr, "libkmgpu: build round %d: %llu slots, %llu buckets to grow, max distance %llu\n", rounds,
              (unsigned long long)n_slots, meta[2], meta[3]);
    if (final_round) break;
    if (meta[2] == 0) break;
    hipLaunchKernelGGL(k_dir_grow, dim3(grid_for((uint64_t)n_buckets, 256)), dim3(256), 0, st, caps,
                       (uint64_t)n_buckets, tv.cshift);
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

}  // namespace

struct km_batch {
  kmjf* db = nullptr;
  km_params_t p{};
  uint32_t max_targets = 0;
  uint64_t max_bases = 0;
  int device = 0;
  uint32_t n_targets = 0;
  uint64_t total_bases = 0;
  uint32_t max_len = 0;
  bool ran_walk = false, ran_graph = false, synced = true;
  hipStream_t last_stream = nullptr;
  // pinned staging for km_batch_fetch (node pools cross PCIe at DMA speed, no zero-filled vectors)
  unsigned char* pin = nullptr;
  uint64_t pin_cap = 0;

  // inputs
  DevBuf<uint8_t> d_bases;
  DevBuf<uint64_t> d_toff;
  std::vector<uint64_t> h_toff;
  DevBuf<uint64_t> d_woff, d_packed;   // 2-bit packed targets (k_pack)
  std::vector<uint64_t> h_woff;
  // k_seed work items and flag bitmaps
  DevBuf<uint32_t> d_item_off, d_flagbits, d_tflag, d_flagged, d_nflagged;
  DevBuf<unsigned long long> d_dfs_probes;
  DevBuf<uint64_t> d_items;
  DevBuf<uint64_t> d_fw_off;
  std::vector<uint32_t> h_item_off;
  std::vector<uint64_t> h_fw_off;
  uint32_t n_items = 0;
  bool big_walk_done = false;
  int graph_mode = 0;                  // 1 = duplicate check only (walk stage run alone)
  // per-target
  DevBuf<uint64_t> d_node_base;
  DevBuf<uint32_t> d_node_cap;
  std::vector<uint64_t> h_node_base;
  std::vector<uint32_t> h_node_cap;
  DevBuf<uint32_t> d_n_nodes, d_n_ref, d_status, d_gstatus, d_npaths, d_pathbase, d_need_full;
  hipStream_t side = nullptr;          // overlaps k_graph_pure with k_dfs
  hipEvent_t ev_seed_done = nullptr, ev_pure_done = nullptr;       // eager fork / join
  hipEvent_t ev_cap_seed = nullptr, ev_cap_pure = nullptr;         // fork / join inside a captured step
  uint32_t pure_lds = 0;
  bool timed = false;                  // the last run recorded its timing events
  hipGraph_t graph = nullptr;          // captured step (KM_RUN_HIPGRAPH)
  hipGraphExec_t gexec = nullptr;
  int graph_stages = 0;
  hipStream_t graph_stream = nullptr;
  DevBuf<uint64_t> d_probes, d_fetches;
  // node pools
  DevBuf<uint64_t> d_node_kmer;
  DevBuf<uint32_t> d_node_cnt;
  uint64_t node_pool_used = 0;       // fast-tier part
  // path pools
  DevBuf<unsigned long long> d_counters;
  DevBuf<uint32_t> d_p_target, d_p_nruns, d_p_len, d_p_mincov, d_r_start, d_r_len;
  DevBuf<uint64_t> d_p_runbase;
  uint64_t path_pool = 0, run_pool = 0;
  // big tier
  DevBuf<unsigned char> d_frames;     // fast-tier DFS stack frames, one slice per target
  DevBuf<unsigned long long> d_stamps; // diagnostics: k_seed time stamps (KM_SEED_STAMPS)
  DevBuf<float> d_tref;               // shared reference-chain distances
  DevBuf<uint32_t> d_big_ids;
  DevBuf<unsigned char> d_big_ws;
  // host mirrors after sync
  std::vector<uint32_t> h_status, h_gstatus, h_n_nodes, h_n_ref, h_npaths, h_pathbase;
  std::vector<uint64_t> h_probes, h_fetches;
  std::vector<unsigned long long> h_dfs_probes;
  unsigned long long h_overflow = 0;
  uint32_t n_big = 0;
  // geometry of the last launch
  bool fast_ok = true;
  WalkArgs wa{};
  GraphArgs ga{};
  uint32_t walk_lds = 0, graph_lds = 0;
  // timing
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float ms[4] = {0, 0, 0, 0};
  unsigned long long h_seed_probes = 0;
  uint32_t h_nflagged = 0;
};

extern "C" int km_batch_create(kmjf_t* h, const km_params_t* params, uint32_t max_targets,
                               uint64_t max_total_bases, km_batch_t** out) {
  if (!h || !params || !out || !max_targets) return fail(KM_E_ARG, "bad argument");
  if (!h->d_slots) return fail(KM_E_STATE, "table not uploaded");
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
  A(b->d_nflagged.alloc(1));
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
  A(b->d_probes.alloc(max_targets));
  A(b->d_fetches.alloc(max_targets));
  const uint64_t pool = max_total_bases + (uint64_t)max_targets * FAST_EXTRA;
  A(b->d_node_kmer.alloc(pool));
  A(b->d_node_cnt.alloc(pool));
  A(b->d_counters.alloc(POOL_GROUPS * POOL_CTR_STRIDE + 16));
  b->path_pool = (((uint64_t)max_targets * 4 + 8192) / POOL_GROUPS + 1) * POOL_GROUPS;
  b->run_pool = (((uint64_t)max_targets * 16 + 32768) / POOL_GROUPS + 1) * POOL_GROUPS;
  A(b->d_p_target.alloc(b->path_pool));
  A(b->d_p_runbase.alloc(b->path_pool));
  A(b->d_p_nruns.alloc(b->path_pool));
  A(b->d_p_len.alloc(b->path_pool));
  A(b->d_p_mincov.alloc(b->path_pool));
  A(b->d_r_start.alloc(b->run_pool));
  A(b->d_r_len.alloc(b->run_pool));
  if (rc == KM_OK) {
    if (hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_seed_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_pure_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_cap_seed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_cap_pure, hipEventDisableTiming) != hipSuccess)
      rc = fail(KM_E_HIP, "stream/event creation failed");
  }
  if (rc == KM_OK) {
    for (int i = 0; i < 5; ++i)
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
  b->d_tflag.release(); b->d_flagged.release(); b->d_nflagged.release(); b->d_dfs_probes.release(); b->d_node_base.release(); b->d_node_cap.release();
  b->d_n_nodes.release(); b->d_n_ref.release(); b->d_status.release(); b->d_gstatus.release();
  b->d_npaths.release(); b->d_pathbase.release(); b->d_need_full.release();
  if (b->side) (void)hipStreamDestroy(b->side);
  if (b->ev_seed_done) (void)hipEventDestroy(b->ev_seed_done);
  if (b->ev_pure_done) (void)hipEventDestroy(b->ev_pure_done);
  if (b->ev_cap_seed) (void)hipEventDestroy(b->ev_cap_seed);
  if (b->ev_cap_pure) (void)hipEventDestroy(b->ev_cap_pure); b->d_probes.release(); b->d_fetches.release();
  b->d_node_kmer.release(); b->d_node_cnt.release(); b->d_counters.release();
  b->d_p_target.release(); b->d_p_runbase.release(); b->d_p_nruns.release(); b->d_p_len.release();
  b->d_p_mincov.release(); b->d_r_start.release(); b->d_r_len.release();
  b->d_big_ids.release(); b->d_big_ws.release(); b->d_tref.release(); b->d_frames.release(); b->d_stamps.release();
  for (int i = 0; i < 5; ++i) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  if (b->pin) (void)hipHostFree(b->pin);
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
  uint64_t pool = 0;
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
    max_len = std::max<uint32_t>(max_len, (uint32_t)L);
  }
  b->h_toff[n] = total;
  b->node_pool_used = pool;
  b->n_items = b->h_item_off[n];
  b->n_targets = n;
  b->total_bases = total;
  b->max_len = max_len;
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
  HIPCHK(hipMemcpyAsync(b->d_node_cap.p, b->h_node_cap.data(), (uint64_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  b->ran_walk = b->ran_graph = false;
  b->synced = true;
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
  a.n_flagged = b->d_nflagged.p;
  a.list = b->d_flagged.p;
  a.n_list_dev = b->d_nflagged.p;
  a.n_list_host = 0;
  a.node_kmer = b->d_node_kmer.p;
  a.node_cnt = b->d_node_cnt.p;
  a.node_base = b->d_node_base.p;
  a.node_cap = b->d_node_cap.p;
  a.n_nodes = b->d_n_nodes.p;
  a.n_ref = b->d_n_ref.p;
  a.status = b->d_status.p;
  a.probes = reinterpret_cast<unsigned long long*>(b->d_probes.p);
  a.dfs_probes = b->d_dfs_probes.p;
  a.fetches = reinterpret_cast<unsigned long long*>(b->d_fetches.p);
  a.g_ws = nullptr;
  a.g_stride = 0;
  const char* dbg = getenv("KM_DEBUG_FLAGS");   // timing ablations only; results are invalid
  a.dbg = dbg ? ((uint32_t)strtoul(dbg, nullptr, 0) & 0xFFu) : 0;
}

static void fill_graph_args(km_batch* b, GraphArgs& g) {
  g.k = b->db->k;
  g.kmask = mask_bits(b->db->k);
  g.pmask = mask_bits(b->db->k - 1);
  g.tids = nullptr;
  g.n_targets = b->n_targets;
  g.node_kmer = b->d_node_kmer.p;
  g.node_cnt = b->d_node_cnt.p;
  g.node_base = b->d_node_base.p;
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
  const char* dbg = getenv("KM_DEBUG_FLAGS");
  g.dbg = dbg ? ((uint32_t)strtoul(dbg, nullptr, 0) >> 8) : 0;
  if (b->graph_mode == 1) g.dbg = 1;       // duplicate check only
}

static void pure_geometry(km_batch* b) {
  const uint32_t max_nref = b->max_len >= (uint32_t)b->db->k ? b->max_len - b->db->k + 1 : 1;
  b->ga.hcap_pure = round_up(4 * (max_nref + 2), 64);            // 32-bit fingerprints at load <= 1/4
  b->pure_lds = b->ga.hcap_pure * 4;
  if (b->pure_lds > FAST_LDS_LIMIT) { b->ga.hcap_pure = 64; b->pure_lds = 256; }   // all -> need_full
}

// Graph stage on one stream: pure-chain pass, then the general kernel for the rest.
static int launch_graph_fast(km_batch* b, hipStream_t st) {
  HIPCHK(hipMemsetAsync(b->d_counters.p, 0, (POOL_GROUPS * POOL_CTR_STRIDE + 16) * sizeof(unsigned long long), st));
  if (b->fast_ok) {
    pure_geometry(b);
    b->ga.use_need_full = 1;
    hipLaunchKernelGGL(k_graph_pure, dim3(b->n_targets), dim3(64), b->pure_lds, st, b->ga);
    hipLaunchKernelGGL(k_graph<false>, dim3(b->n_targets), dim3(GRAPH_THREADS), b->graph_lds, st, b->ga);
  } else {
    // no LDS-resident tier for these parameters: mark everything for the large tier
    HIPCHK(hipMemsetAsync(b->d_gstatus.p, 0, (uint64_t)b->n_targets * 4, st));
    HIPCHK(hipMemsetAsync(b->d_npaths.p, 0, (uint64_t)b->n_targets * 4, st));
  }
  HIPCHK(hipGetLastError());
  return KM_OK;
}

static void launch_seed(uint32_t n_items, hipStream_t st, const WalkArgs& wa) {
  if (wa.stamps) hipLaunchKernelGGL(k_seed<true>, dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);   // KM_SEED_STAMPS diagnostics
  else hipLaunchKernelGGL(k_seed<false>, dim3(n_items), dim3(SEED_BLOCK), 0, st, wa);
}

extern "C" int km_batch_run(km_batch_t* b, int stages, void* stream) {
  if (!b) return fail(KM_E_ARG, "null argument");
  if (!b->n_targets) { b->ran_walk = true; b->ran_graph = (stages & KM_STAGE_GRAPH) != 0; return KM_OK; }
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = (hipStream_t)stream;
  b->last_stream = st;
  const bool want_graph = (stages & KM_RUN_HIPGRAPH) != 0;
  stages &= (KM_STAGE_WALK | KM_STAGE_GRAPH);
  if (want_graph && b->gexec && b->graph_stages == stages && b->graph_stream == st) {
    HIPCHK(hipGraphLaunch(b->gexec, st));
    b->ran_walk = true;
    b->ran_graph = true;
    b->big_walk_done = false;
    b->synced = false;
    b->timed = false;
    return KM_OK;
  }
  const uint32_t max_nref = b->max_len >= (uint32_t)b->db->k ? b->max_len - b->db->k + 1 : 1;

  // ---- fast-tier geometry
  WalkArgs& wa = b->wa;
  fill_walk_args(b, wa);
  wa.hs_cap = round_up(2 * (max_nref + FAST_EXTRA), 64);
  wa.words_cap = round_up((b->max_len + 31) / 32 + 1, 2);
  wa.fcap = round_up(b->p.max_stack + 2, 2);
  wa.bcap = b->p.max_break + 1;
  const uint64_t wl = walk_lds_bytes(wa.hs_cap, wa.words_cap, wa.bcap);
  wa.f_stride = walk_frame_bytes(wa.fcap);
  {
    int rc = b->d_frames.alloc((uint64_t)b->n_targets * wa.f_stride);
    if (rc != KM_OK) return rc;
  }
  wa.f_ws = b->d_frames.p;
  wa.stamps = nullptr;
  if (getenv("KM_SEED_STAMPS")) {
    int rc = b->d_stamps.alloc(16ull * (SEED_BLOCK / 64) * (b->n_items + 4));
    if (rc != KM_OK) return rc;
    wa.stamps = b->d_stamps.p;
  }
  b->graph_mode = (stages & KM_STAGE_GRAPH) ? 0 : 1;
  GraphArgs& ga = b->ga;
  fill_graph_args(b, ga);
  ga.ncap = max_nref + FAST_EXTRA + 2;
  ga.hcap = round_up(ga.ncap + ga.ncap / 2 + 1, 64);
  const uint64_t gl = graph_ws_bytes<uint16_t>(ga.ncap, ga.hcap);
  b->fast_ok = wl <= FAST_LDS_LIMIT && gl <= FAST_LDS_LIMIT && ga.ncap < 0xFFFF &&
               b->p.max_break < 4096;
  b->walk_lds = (uint32_t)wl;
  b->graph_lds = (uint32_t)gl;

  bool graph_launched = false;
  // a captured step needs the LDS-resident tier on both stages (no host round trips inside)
  b->timed = true;
  const bool capturing = want_graph && st != nullptr && b->fast_ok && (stages & KM_STAGE_WALK);   // the NULL stream cannot be captured
  if (capturing) {
    b->timed = false;
    drop_graph(b);
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  }
  if (stages & KM_STAGE_WALK) {
    if (!capturing) HIPCHK(hipEventRecord(b->ev[0], st));
    hipLaunchKernelGGL(k_pack, dim3(b->n_targets), dim3(64), 0, st, wa);
    if (!capturing) HIPCHK(hipEventRecord(b->ev[3], st));
    if (b->n_items)
      launch_seed(b->n_items, st, wa);
    if (!capturing) HIPCHK(hipEventRecord(b->ev[4], st));
    if (b->fast_ok) {
      // unflagged targets are final after k_seed: their pure-chain check runs on the side
      // stream while k_dfs (latency-bound, few waves) walks the flagged ones
      HIPCHK(hipMemsetAsync(b->d_counters.p, 0, (POOL_GROUPS * POOL_CTR_STRIDE + 16) * sizeof(unsigned long long), st));
      hipEvent_t e_fork = capturing ? b->ev_cap_seed : b->ev_seed_done;
      hipEvent_t e_join = capturing ? b->ev_cap_pure : b->ev_pure_done;
      HIPCHK(hipEventRecord(e_fork, st));
      HIPCHK(hipStreamWaitEvent(b->side, e_fork, 0));
      pure_geometry(b);
      ga.use_need_full = 1;
      hipLaunchKernelGGL(k_graph_pure, dim3(b->n_targets), dim3(64), b->pure_lds, b->side, ga);
      HIPCHK(hipEventRecord(e_join, b->side));
      hipLaunchKernelGGL(k_dfs<false>, dim3(b->n_targets), dim3(64), b->walk_lds, st, wa);
      HIPCHK(hipGetLastError());
      if (!capturing) HIPCHK(hipEventRecord(b->ev[1], st));
      HIPCHK(hipStreamWaitEvent(st, e_join, 0));
      hipLaunchKernelGGL(k_graph<false>, dim3(b->n_targets), dim3(GRAPH_THREADS), b->graph_lds, st, ga);
      HIPCHK(hipGetLastError());
      graph_launched = true;
    } else {
      HIPCHK(hipGetLastError());
      if (!capturing) HIPCHK(hipEventRecord(b->ev[1], st));
    }
    b->ran_walk = true;
    b->ran_graph = false;
    b->big_walk_done = false;
  } else if (!b->ran_walk) {
    return fail(KM_E_STATE, "graph stage requested before the walk stage");
  } else {
    if (!capturing) HIPCHK(hipEventRecord(b->ev[0], st));
    if (!capturing) HIPCHK(hipEventRecord(b->ev[3], st));
    if (!capturing) HIPCHK(hipEventRecord(b->ev[4], st));
    if (!capturing) HIPCHK(hipEventRecord(b->ev[1], st));
  }
  // the graph kernels also host the duplicate-k-mer check, so they always run
  // (graph_mode 1 = stop after that check)
  if (!graph_launched) {
    int rc = launch_graph_fast(b, st);
    if (rc != KM_OK) return rc;
  }
  b->ran_graph = true;                      // graph_mode says how far it went
  if (!capturing) HIPCHK(hipEventRecord(b->ev[2], st));
  if (capturing) {
    HIPCHK(hipStreamEndCapture(st, &b->graph));
    HIPCHK(hipGraphInstantiate(&b->gexec, b->graph, nullptr, nullptr, 0));
    b->graph_stages = stages;
    b->graph_stream = st;
    HIPCHK(hipGraphLaunch(b->gexec, st));
  }
  b->synced = false;
  return KM_OK;
}

static int pull_status(km_batch* b, hipStream_t st) {
  const uint32_t n = b->n_targets;
  b->h_status.resize(n); b->h_gstatus.assign(n, 0); b->h_n_nodes.resize(n); b->h_n_ref.resize(n);
  b->h_npaths.assign(n, 0); b->h_pathbase.assign(n, 0); b->h_probes.resize(n); b->h_fetches.resize(n);
  HIPCHK(hipMemcpyAsync(b->h_status.data(), b->d_status.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b->h_n_nodes.data(), b->d_n_nodes.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b->h_n_ref.data(), b->d_n_ref.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b->h_probes.data(), b->d_probes.p, (uint64_t)n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b->h_fetches.data(), b->d_fetches.p, (uint64_t)n * 8, hipMemcpyDeviceToHost, st));
  b->h_dfs_probes.resize(n);
  HIPCHK(hipMemcpyAsync(b->h_dfs_probes.data(), b->d_dfs_probes.p, (uint64_t)n * 8, hipMemcpyDeviceToHost, st));
  if (b->ran_graph) {
    HIPCHK(hipMemcpyAsync(b->h_gstatus.data(), b->d_gstatus.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_npaths.data(), b->d_npaths.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_pathbase.data(), b->d_pathbase.p, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&b->h_overflow, b->d_counters.p + POOL_GROUPS * POOL_CTR_STRIDE, 8, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  b->h_seed_probes = 0;
  for (uint32_t t = 0; t < n; ++t) {
    b->h_seed_probes += b->h_probes[t];
    b->h_probes[t] += b->h_dfs_probes[t];
  }
  return KM_OK;
}

// Large tier: rerun the listed targets with global-memory workspaces.
static int run_big_walk(km_batch* b, const std::vector<uint32_t>& ids, hipStream_t st) {
  const uint32_t nb = (uint32_t)ids.size();
  const int k = b->db->k;
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
  // the seed kernel's results (target k-mers and their counts) move to the new storage
  for (size_t q = 0; q < ids.size(); ++q) {
    const uint32_t t = ids[q];
    const uint64_t nref = b->h_n_ref[t];
    if (!nref) continue;
    HIPCHK(hipMemcpyAsync(b->d_node_kmer.p + b->h_node_base[t], b->d_node_kmer.p + old_base[q], nref * 8,
                          hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(b->d_node_cnt.p + b->h_node_base[t], b->d_node_cnt.p + old_base[q], nref * 4,
                          hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(hipMemcpyAsync(b->d_node_base.p, b->h_node_base.data(), (uint64_t)b->n_targets * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_node_cap.p, b->h_node_cap.data(), (uint64_t)b->n_targets * 4, hipMemcpyHostToDevice, st));
  int rc = b->d_big_ids.alloc(nb); if (rc != KM_OK) return rc;
  HIPCHK(hipMemcpyAsync(b->d_big_ids.p, ids.data(), (uint64_t)nb * 4, hipMemcpyHostToDevice, st));

  const uint32_t max_nref = b->max_len >= (uint32_t)k ? b->max_len - k + 1 : 1;
  WalkArgs a;
  fill_walk_args(b, a);
  a.n_list_dev = nullptr;
  const uint64_t max_nodes = std::max<uint64_t>(max_nref, (uint64_t)b->p.max_node + b->p.max_stack) + 1;
  const uint64_t hs = 2 * (max_nodes + b->p.max_stack + 64);
  if (hs > 0x7FFFFF00ull) return fail(KM_E_ARG, "node limit too large");
  a.hs_cap = round_up((uint32_t)hs, 64);
  a.words_cap = round_up((b->max_len + 31) / 32 + 1, 2);
  a.fcap = round_up(b->p.max_stack + 2, 2);
  a.bcap = b->p.max_break + 1;
  a.g_stride = walk_ws_bytes(a.hs_cap, a.words_cap, a.fcap, a.bcap);
  // run in slices so the workspace stays bounded
  const uint64_t budget = 8ull << 30;
  uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nb, budget / a.g_stride));
  rc = b->d_big_ws.alloc((uint64_t)per * a.g_stride); if (rc != KM_OK) return rc;
  a.g_ws = b->d_big_ws.p;
  for (uint32_t s = 0; s < nb; s += per) {
    const uint32_t cnt = std::min(per, nb - s);
    a.list = b->d_big_ids.p + s;
    a.n_list_host = cnt;
    hipLaunchKernelGGL(k_dfs<true>, dim3(cnt), dim3(64), 0, st, a);
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
  g.g_stride = graph_ws_bytes<uint32_t>(g.ncap, g.hcap);
  const uint64_t budget = 8ull << 30;
  uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nb, budget / g.g_stride));
  rc = b->d_big_ws.alloc((uint64_t)per * g.g_stride); if (rc != KM_OK) return rc;
  g.g_ws = b->d_big_ws.p;
  for (uint32_t s = 0; s < nb; s += per) {
    const uint32_t cnt = std::min(per, nb - s);
    g.tids = b->d_big_ids.p + s;
    hipLaunchKernelGGL(k_graph<true>, dim3(cnt), dim3(GRAPH_THREADS), 0, st, g);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
  }
  return KM_OK;
}

static int grow_path_pools(km_batch* b) {
  b->path_pool = (b->path_pool * 4 / POOL_GROUPS + 1) * POOL_GROUPS;
  b->run_pool = (b->run_pool * 4 / POOL_GROUPS + 1) * POOL_GROUPS;
  int rc = KM_OK;
  auto A = [&](int r) { if (rc == KM_OK) rc = r; };
  A(b->d_p_target.alloc(b->path_pool)); A(b->d_p_runbase.alloc(b->path_pool));
  A(b->d_p_nruns.alloc(b->path_pool)); A(b->d_p_len.alloc(b->path_pool));
  A(b->d_p_mincov.alloc(b->path_pool)); A(b->d_r_start.alloc(b->run_pool));
  A(b->d_r_len.alloc(b->run_pool));
  return rc;
}

static int relaunch_fast_graph(km_batch* b, hipStream_t st) {
  fill_graph_args(b, b->ga);
  const uint32_t max_nref = b->max_len >= (uint32_t)b->db->k ? b->max_len - b->db->k + 1 : 1;
  b->ga.ncap = max_nref + FAST_EXTRA + 2;
  b->ga.hcap = round_up(b->ga.ncap + b->ga.ncap / 2 + 1, 64);
  int rc = launch_graph_fast(b, st);
  if (rc != KM_OK) return rc;
  HIPCHK(hipStreamSynchronize(st));
  return KM_OK;
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
  if (b->timed) {
    (void)hipEventElapsedTime(&b->ms[0], b->ev[0], b->ev[1]);
    (void)hipEventElapsedTime(&b->ms[1], b->ev[1], b->ev[2]);
    (void)hipEventElapsedTime(&b->ms[2], b->ev[0], b->ev[2]);
    (void)hipEventElapsedTime(&b->ms[3], b->ev[3], b->ev[4]);
    (void)hipGetLastError();
  } else {
    b->ms[0] = b->ms[1] = b->ms[2] = b->ms[3] = 0.0f;
  }
  HIPCHK(hipMemcpy(&b->h_nflagged, b->d_nflagged.p, 4, hipMemcpyDeviceToHost));
  int rc = pull_status(b, st);
  if (rc != KM_OK) return rc;
  const uint32_t n = b->n_targets;

  std::vector<uint32_t> big;
  if (b->fast_ok) {
    for (uint32_t t = 0; t < n; ++t) if (b->h_status[t] == T_NEEDS_BIG) big.push_back(t);
  } else if (!b->big_walk_done) {
    // no LDS-resident tier for these parameters: every flagged target takes the large tier
    uint32_t nf = 0;
    HIPCHK(hipMemcpy(&nf, b->d_nflagged.p, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> fl(nf);
    if (nf) HIPCHK(hipMemcpy(fl.data(), b->d_flagged.p, (uint64_t)nf * 4, hipMemcpyDeviceToHost));
    for (uint32_t t : fl) if (b->h_status[t] == T_OK) big.push_back(t);
    std::sort(big.begin(), big.end());
  }
  b->big_walk_done = true;
  b->n_big = (uint32_t)big.size();
  std::vector<char> force_big(n, 0);
  if (!big.empty()) {
    rc = run_big_walk(b, big, st);
    if (rc != KM_OK) return rc;
    for (uint32_t t : big) force_big[t] = 1;     // the fast graph pass skipped them
    rc = pull_status(b, st);
    if (rc != KM_OK) return rc;
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
      }
      std::vector<uint32_t> todo;
      for (uint32_t t = 0; t < n; ++t)
        if (b->h_status[t] == T_OK && (force_big[t] || b->h_gstatus[t] == T_NEEDS_BIG)) todo.push_back(t);
      if (!todo.empty()) {
        rc = run_big_graph(b, todo, st);
        if (rc != KM_OK) return rc;
        rc = pull_status(b, st);
        if (rc != KM_OK) return rc;
      }
      if (!b->h_overflow) break;
    }
  }
  b->synced = true;
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

extern "C" int km_batch_timings(km_batch_t* b, float* ms4) {
  float* ms3 = ms4;
  if (!b || !ms3) return fail(KM_E_ARG, "null argument");
  int rc = km_batch_sync(b);
  if (rc != KM_OK) return rc;
  ms3[0] = b->ms[0]; ms3[1] = b->ms[1]; ms3[2] = b->ms[2]; ms3[3] = b->ms[3];
  return KM_OK;
}

extern "C" int km_batch_sizes(km_batch_t* b, km_batch_sizes_t* s) {
  if (!b || !s) return fail(KM_E_ARG, "null argument");
  if (!b->ran_walk) return fail(KM_E_STATE, "nothing has run yet");
  int rc = km_batch_sync(b);
  if (rc != KM_OK) return rc;
  memset(s, 0, sizeof *s);
  s->n_targets = b->n_targets;
  s->n_big_tier = b->n_big;
  s->n_flagged = b->h_nflagged;
  s->seed_probes = b->h_seed_probes;
  uint64_t paths = 0;
  for (uint32_t t = 0; t < b->n_targets; ++t) {
    if (b->h_status[t] == T_OK || b->h_status[t] == T_NODE_LIMIT) s->n_nodes += b->h_n_nodes[t];
    s->logical_probes += b->h_probes[t];
    s->table_fetches += b->h_fetches[t];
    paths += b->h_npaths[t];
  }
  s->n_paths = (uint32_t)paths;
  // runs: sum over the path records actually referenced
  if (b->ran_graph && paths) {
    std::vector<uint32_t> nruns((size_t)b->path_pool);
    HIPCHK(hipMemcpy(nruns.data(), b->d_p_nruns.p, nruns.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t t = 0; t < b->n_targets; ++t)
      for (uint32_t i = 0; i < b->h_npaths[t]; ++i) s->n_runs += nruns[b->h_pathbase[t] + i];
  }
  return KM_OK;
}

namespace {
struct PathRec {
  const uint32_t* rs;
  const uint32_t* rl;
  uint32_t nruns, len, mincov;
};
// lexicographic order of the expanded index sequences
bool path_less(const PathRec& A, const PathRec& B) {
  uint32_t ia = 0, ib = 0, oa = 0, ob = 0;
  while (ia < A.nruns && ib < B.nruns) {
    const uint32_t va = A.rs[ia] + oa, vb = B.rs[ib] + ob;
    if (va != vb) return va < vb;
    const uint32_t step = std::min(A.rl[ia] - oa, B.rl[ib] - ob);
    oa += step; ob += step;
    if (oa == A.rl[ia]) { ++ia; oa = 0; }
    if (ob == B.rl[ib]) { ++ib; ob = 0; }
  }
  return ia == A.nruns && ib < B.nruns;
}
}  // namespace

extern "C" int km_batch_fetch(km_batch_t* b, const km_batch_out_t* out) {
  if (!b || !out) return fail(KM_E_ARG, "null argument");
  if (!b->ran_walk) return fail(KM_E_STATE, "nothing has run yet");
  int rc = km_batch_sync(b);
  if (rc != KM_OK) return rc;
  HIPCHK(hipSetDevice(b->device));
  const uint32_t n = b->n_targets;
  for (uint32_t t = 0; t < n; ++t) {
    if (out->status) out->status[t] = b->h_status[t] == T_OK && b->h_gstatus[t] != T_OK ? KM_T_INTERNAL
                                                                                     : b->h_status[t];
    if (out->aux) out->aux[t] = 0;
    if (out->n_ref) out->n_ref[t] = b->h_n_ref[t];
    if (out->probes) out->probes[t] = b->h_probes[t];
  }
  // ---- nodes (CSR in target order)
  if (out->node_off || out->node_kmer || out->node_count) {
    std::vector<uint64_t> noff(n + 1, 0);
    for (uint32_t t = 0; t < n; ++t) {
      const bool has = b->h_status[t] == T_OK || b->h_status[t] == T_NODE_LIMIT;
      noff[t + 1] = noff[t] + (has ? b->h_n_nodes[t] : 0);
    }
    if (out->node_off) memcpy(out->node_off, noff.data(), (n + 1) * 8);
    if (out->node_kmer || out->node_count) {
      // pull the used part of the pools through a pinned staging buffer, compact on the host
      const uint64_t used = b->node_pool_used;
      const uint64_t need = used * 12 + 64;
      if (need > b->pin_cap) {
        if (b->pin) (void)hipHostFree(b->pin);
        b->pin = nullptr;
        b->pin_cap = 0;
        HIPCHK(hipHostMalloc((void**)&b->pin, need + need / 4, hipHostMallocDefault));
        b->pin_cap = need + need / 4;
      }
      const uint64_t* hk = reinterpret_cast<const uint64_t*>(b->pin);
      const uint32_t* hc = reinterpret_cast<const uint32_t*>(b->pin + used * 8);
      if (out->node_kmer && used)
        HIPCHK(hipMemcpy(b->pin, b->d_node_kmer.p, used * 8, hipMemcpyDeviceToHost));
      if (out->node_count && used)
        HIPCHK(hipMemcpy(b->pin + used * 8, b->d_node_cnt.p, used * 4, hipMemcpyDeviceToHost));
      for (uint32_t t = 0; t < n; ++t) {
        const uint64_t cnt = noff[t + 1] - noff[t];
        if (!cnt) continue;
        if (out->node_kmer) memcpy(out->node_kmer + noff[t], hk + b->h_node_base[t], cnt * 8);
        if (out->node_count) memcpy(out->node_count + noff[t], hc + b->h_node_base[t], cnt * 4);
      }
    }
  }
  // ---- paths
  if (b->ran_graph && (out->path_off || out->run_off || out->run_start || out->run_len ||
                       out->path_len || out->path_min_cov)) {
    const uint64_t np = b->path_pool, nr = b->run_pool;
    std::vector<uint32_t> p_nruns(np), p_len(np), p_mincov(np), r_start(nr), r_len(nr);
    std::vector<uint64_t> p_runbase(np);
    if (np) {
      HIPCHK(hipMemcpy(p_nruns.data(), b->d_p_nruns.p, np * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(p_len.data(), b->d_p_len.p, np * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(p_mincov.data(), b->d_p_mincov.p, np * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(p_runbase.data(), b->d_p_runbase.p, np * 8, hipMemcpyDeviceToHost));
    }
    if (nr) {
      HIPCHK(hipMemcpy(r_start.data(), b->d_r_start.p, nr * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(r_len.data(), b->d_r_len.p, nr * 4, hipMemcpyDeviceToHost));
    }
    uint32_t pcur = 0;
    uint64_t rcur = 0;
    std::vector<PathRec> recs;
    for (uint32_t t = 0; t < n; ++t) {
      if (out->path_off) out->path_off[t] = pcur;
      recs.clear();
      for (uint32_t i = 0; i < b->h_npaths[t]; ++i) {
        const uint64_t pi = (uint64_t)b->h_pathbase[t] + i;
        PathRec r{r_start.data() + p_runbase[pi], r_len.data() + p_runbase[pi], p_nruns[pi], p_len[pi],
                  p_mincov[pi]};
        recs.push_back(r);
      }
      std::sort(recs.begin(), recs.end(), path_less);
      for (const PathRec& r : recs) {
        if (out->run_off) out->run_off[pcur] = rcur;
        if (out->path_len) out->path_len[pcur] = r.len;
        if (out->path_min_cov) out->path_min_cov[pcur] = r.mincov;
        for (uint32_t j = 0; j < r.nruns; ++j) {
          if (out->run_start) out->run_start[rcur + j] = r.rs[j];
          if (out->run_len) out->run_len[rcur + j] = r.rl[j];
        }
        rcur += r.nruns;
        ++pcur;
      }
    }
    if (out->path_off) out->path_off[n] = pcur;
    if (out->run_off) out->run_off[pcur] = rcur;
  } else if (out->path_off) {
    for (uint32_t t = 0; t <= n; ++t) out->path_off[t] = 0;
    if (out->run_off) out->run_off[0] = 0;
  }
  return KM_OK;
}
