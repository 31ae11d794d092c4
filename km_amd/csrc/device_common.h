// device_common.h — k-mer arithmetic, the HBM table layout and its probe.
//
// TABLE LAYOUT ("sibling buckets").  The reference's innermost operation is
// Jellyfish.get_child (km/utils/Jellyfish.py:55-72): four Jellyfish.query calls
// (km/utils/Jellyfish.py:47-53) for the k-mers S+A, S+C, S+G, S+T that share the
// (k-1)-mer S.  A plain k-mer -> count hash scatters those four over four HBM
// lines.  Here the open-addressing table is keyed by the *shared (k-1)-mer*
// instead, so one aligned 16-byte slot (a single dwordx4 load) answers a whole get_child:
//
//     slot = { u64 tag ; u16 count[4] }                    (16 B, 16-B aligned)
//     tag  = (G << 1) | side,   G = the canonical (k-1)-mer (min(S, revcomp S))
//     side 0: count[c] = count of the k-mer  G+c           (right extension)
//     side 1: count[c] = count of the k-mer  c+G           (left extension)
//
// A stored canonical k-mer K is entered twice, once for each orientation
// O in {K, revcomp K}: with P = O[:-1], c = O[-1]:  P <= revcomp(P) -> (P, side 0,
// slot c), else (revcomp P, side 1, slot 3-c)  [because revcomp(P+c) = comp(c)+revcomp(P)].
// A lookup of the forward children of X uses P = X[1:] with the same rule, a
// single query(X) uses P = X[:-1], c = X[-1].  Non-canonical databases store and
// look up P as is (side 0 only).  Empty slots have tag == ~0 (a valid tag is < 2^63).
// Counts are stored as u16; a count >= 65535 is stored as 0xFFFF and its exact value
// lives in a small side table keyed by the canonical k-mer (OvfSlot), consulted only then.
// HBM serves the table in 128-byte lines (8 slots), so linear probing stays in the line.
// Linear probing; slot index = mulhi64(mix64(tag), n_slots) (any capacity).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmd {

constexpr uint64_t EMPTY = ~0ull;

struct __attribute__((aligned(16))) Slot {
  uint64_t tag;
  uint16_t c[4];
};
static_assert(sizeof(Slot) == 16, "slot must be one dwordx4");
constexpr uint32_t COUNT_ESCAPE = 0xFFFFu;

// exact counts >= COUNT_ESCAPE; empty iff count == 0
struct __attribute__((aligned(16))) OvfSlot {
  uint64_t kmer;     // canonical k-mer (as stored in the database)
  uint32_t count;
  uint32_t pad;
};

struct TableView {
  const Slot* slots;
  uint64_t n_slots;
  const OvfSlot* ovf;
  uint64_t n_ovf;
  uint64_t kmask;   // 2k low bits set
  uint64_t pmask;   // 2(k-1) low bits set
  int k;
  int canonical;
};

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__host__ __device__ inline uint64_t revcomp(uint64_t x, int k) {
  x = ~x;
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * k);
}

struct Group {
  uint64_t tag;
  uint32_t flip;   // child base c lives in count[flip ? 3 - c : c]
};

// Group of the k-mers that have (k-1)-mer P as their prefix.
__host__ __device__ inline Group group_of_prefix(uint64_t P, int k, int canonical) {
  Group g;
  if (!canonical) {
    g.tag = P << 1;
    g.flip = 0;
    return g;
  }
  uint64_t R = revcomp(P, k - 1);
  if (P <= R) {
    g.tag = P << 1;
    g.flip = 0;
  } else {
    g.tag = (R << 1) | 1;
    g.flip = 1;
  }
  return g;
}

__host__ __device__ inline uint64_t slot_index(uint64_t tag, uint64_t n_slots) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(mix64(tag), n_slots);
#else
  return (uint64_t)(((unsigned __int128)mix64(tag) * n_slots) >> 64);
#endif
}

__device__ inline uint32_t pick4(uint4 v, uint32_t i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// Counts of the four members of the group `tag` in slot order (zeros if absent), as
// stored (u16, possibly COUNT_ESCAPE).  One 16-byte load per probe step; *fetches counts
// the slots read.
__device__ inline uint4 table_lookup4(const TableView& t, uint64_t tag, uint32_t* fetches) {
  uint64_t idx = slot_index(tag, t.n_slots);
  for (uint64_t step = 0; step < t.n_slots; ++step) {
    const uint4 a = *reinterpret_cast<const uint4*>(t.slots + idx);
    ++*fetches;
    const uint64_t tg = ((uint64_t)a.y << 32) | a.x;
    if (tg == tag) return make_uint4(a.z & 0xFFFFu, a.z >> 16, a.w & 0xFFFFu, a.w >> 16);
    if (tg == EMPTY) break;
    if (++idx == t.n_slots) idx = 0;
  }
  return make_uint4(0, 0, 0, 0);
}

// Exact count of a k-mer whose stored count is COUNT_ESCAPE.
__device__ inline uint32_t overflow_count(const TableView& t, uint64_t kmer) {
  if (t.canonical) {
    const uint64_t r = revcomp(kmer, t.k);
    if (r < kmer) kmer = r;
  }
  if (t.n_ovf == 0) return COUNT_ESCAPE;
  uint64_t idx = slot_index(kmer, t.n_ovf);
  for (uint64_t step = 0; step < t.n_ovf; ++step) {
    const OvfSlot o = t.ovf[idx];
    if (o.count == 0) break;
    if (o.kmer == kmer) return o.count;
    if (++idx == t.n_ovf) idx = 0;
  }
  return COUNT_ESCAPE;
}

// Counts of X[1:]+A, +C, +G, +T (child-base order).
__device__ inline uint4 forward_children(const TableView& t, uint64_t X, uint32_t* fetches) {
  Group g = group_of_prefix(X & t.pmask, t.k, t.canonical);
  uint4 c = table_lookup4(t, g.tag, fetches);
  if (g.flip) c = make_uint4(c.w, c.z, c.y, c.x);
  if (c.x == COUNT_ESCAPE || c.y == COUNT_ESCAPE || c.z == COUNT_ESCAPE || c.w == COUNT_ESCAPE) {
    const uint64_t base = (X << 2) & t.kmask;
    if (c.x == COUNT_ESCAPE) c.x = overflow_count(t, base | 0);
    if (c.y == COUNT_ESCAPE) c.y = overflow_count(t, base | 1);
    if (c.z == COUNT_ESCAPE) c.z = overflow_count(t, base | 2);
    if (c.w == COUNT_ESCAPE) c.w = overflow_count(t, base | 3);
  }
  return c;
}

// Jellyfish.query(X): km/utils/Jellyfish.py:47-53.
__device__ inline uint32_t query_one(const TableView& t, uint64_t X, uint32_t* fetches) {
  Group g = group_of_prefix(X >> 2, t.k, t.canonical);
  uint4 c = table_lookup4(t, g.tag, fetches);
  uint32_t s = (uint32_t)(X & 3);
  uint32_t v = pick4(c, g.flip ? 3 - s : s);
  if (v == COUNT_ESCAPE) v = overflow_count(t, X);
  return v;
}

// Children kept by Jellyfish.get_child: count >= max(sum * cutoff, n_cutoff)
// evaluated as Python does (float64 product, exact int/float comparison):
// km/utils/Jellyfish.py:69-72.  Returns a 4-bit mask, bit c = child base c.
__device__ inline uint32_t child_mask(uint4 c, double ratio, int64_t n_cutoff) {
  uint64_t sum = (uint64_t)c.x + c.y + c.z + c.w;
  double t = (double)sum * ratio;
  double nc = (double)n_cutoff;
  double thr = (nc > t) ? nc : t;          // Python max(t, nc)
  uint32_t m = 0;
  m |= ((double)c.x >= thr) ? 1u : 0u;
  m |= ((double)c.y >= thr) ? 2u : 0u;
  m |= ((double)c.z >= thr) ? 4u : 0u;
  m |= ((double)c.w >= thr) ? 8u : 0u;
  return m;
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace kmd
