// device_common.h — k-mer arithmetic, the HBM table layout and its probe.
//
// TABLE LAYOUT ("sibling slots in minimizer buckets").  The reference's innermost
// operation is Jellyfish.get_child (km/utils/Jellyfish.py:55-72): four Jellyfish.query
// calls (km/utils/Jellyfish.py:47-53) for the k-mers S+A, S+C, S+G, S+T that share the
// (k-1)-mer S.  A plain k-mer -> count hash scatters those four over four HBM lines, and
// the next step of a walk (S shifted by one base) over four more.  Here
//
//  (1) one aligned 16-byte slot (a single dwordx4 load) answers a whole get_child:
//        slot = { u64 tag ; u16 count[4] }
//        tag  = (G << 1) | side,   G = the canonical (k-1)-mer (min(S, revcomp S))
//        side 0: count[c] = count of the k-mer  G+c           (right extension)
//        side 1: count[c] = count of the k-mer  c+G           (left extension)
//      A stored canonical k-mer K is entered twice, once for each orientation
//      O in {K, revcomp K}: with P = O[:-1], c = O[-1]:  P <= revcomp(P) -> (P, side 0,
//      slot c), else (revcomp P, side 1, slot 3-c)  [revcomp(P+c) = comp(c)+revcomp(P)].
//      A lookup of the forward children of X uses P = X[1:] with the same rule, a single
//      query(X) uses P = X[:-1], c = X[-1].  Non-canonical databases store and look up
//      P as is (side 0 only).  Empty slots have tag == ~0 (a valid tag is < 2^63).
//
//  (2) slots are grouped into variable-size buckets by the MINIMIZER of P (the m-mer of
//      P, m = 15 for k = 31, whose canonical form has the smallest order hash):
//        bucket b owns the slots [2*dir[b], 2*dir[b+1])   (dir = exclusive prefix sum, u32)
//      Consecutive (k-1)-mers of a sequence share their minimizer for ~(w+1)/2 steps
//      (w = k-m windows), i.e. they live in the same bucket.  Every bucket is sized from
//      its own entry count, so a heavy minimizer cannot overfill its neighbourhood.
//      * a bucket with fewer than ~w entries is a small hash table (2 slots per entry) with
//        an order-preserving home position;
//      * a larger one ("class mode", S = q * NC slots, NC = the power of two >= 2w, q >= 2)
//        gives each of the 2w classes (strand of the minimizer, its offset in P) its own q
//        slots: the (k-1)-mers of one super-k-mer fall into consecutive classes and never
//        collide with each other, so a lookup finds its slot in the aligned pair it reads
//        first and the next (k-1)-mer of a walk sits q slots further in the same HBM lines.
//      * a bucket that still cannot keep its keys in their home pairs after two doublings
//        (real data: dozens of error variants of one super-k-mer crowd a few classes) becomes
//        a plain hash table over all its pairs (S = an ODD multiple of NC marks it) and probes
//        linearly; max_probe bounds every lookup.
//      A wave waits for its slowest lane: bounding the probe length matters more than its mean.
//
// Counts are stored as u16; a count >= 65535 is stored as 0xFFFF and its exact value
// lives in a small side table keyed by the canonical k-mer (OvfSlot), consulted only then.
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
  const uint32_t* dir;   // [n_buckets + 1] exclusive prefix sum of entries per bucket
  uint64_t n_slots;      // 2 * dir[n_buckets]
  const OvfSlot* ovf;
  uint64_t n_ovf;
  uint64_t kmask;        // 2k low bits set
  uint64_t pmask;        // 2(k-1) low bits set
  uint32_t n_buckets;    // a power of two
  uint32_t bshift;       // 32 - log2(n_buckets): bucket = hash >> bshift
  uint32_t unit;         // slots per entry when sizing a bucket (2 = load factor <= 0.5)
  uint32_t cshift;       // log2(NC), NC = power of two >= 2w: class-mode buckets hold q * NC slots
  uint32_t max_probe;    // every stored key sits within this many slots of its home (2: the home pair)
  uint32_t mmask;        // 2m low bits set
  uint32_t inv32;        // floor(2^32 / (2w * 256)): fine position -> 32-bit fraction
  int k;
  int canonical;
  int m;                 // minimizer length (odd, <= 15)
  int w;                 // minimizer windows in a (k-1)-mer: k - m
};

// minimizer length / window count for a given k (odd m: an m-mer never equals its own
// reverse complement, so the strand of a minimizer is always defined)
__host__ __device__ inline int minimizer_len(int k) {
  int m = k - 1 < 15 ? k - 1 : 15;
  if (!(m & 1)) --m;
  return m < 1 ? 1 : m;
}

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__host__ __device__ inline uint64_t revcomp(uint64_t x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  // bit reversal reverses the base order and swaps the two bits of every base; swap them back
  uint64_t y = __brevll(~x);
  y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
  return y >> (64 - 2 * k);
#else
  x = ~x;
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * k);
#endif
}
// the same for an m-mer held in 32 bits (m <= 16)
__host__ __device__ inline uint32_t revcomp32(uint32_t x, int m) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t y = __brev(~x);
  y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
  return y >> (32 - 2 * m);
#else
  return (uint32_t)revcomp((uint64_t)x, m);
#endif
}

__host__ __device__ inline uint64_t slot_index(uint64_t tag, uint64_t n_slots) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(mix64(tag), n_slots);
#else
  return (uint64_t)(((unsigned __int128)mix64(tag) * n_slots) >> 64);
#endif
}

// Where the group of the k-mers with (k-1)-mer prefix P lives.
struct Key {
  uint64_t tag;
  uint32_t flip;     // child base c lives in count[flip ? 3 - c : c]
  uint32_t bucket;
  uint32_t frac;     // small buckets: order-preserving home position as a 32-bit fraction
  uint32_t cls;      // class-mode buckets: class in [0, 2w)
  uint32_t hsub;     // class-mode buckets: hash choosing the slot pair inside the class
};

// order hash of a canonical m-mer (a bijection on 32 bits: distinct m-mers never tie)
__host__ __device__ inline uint32_t mm_order(uint32_t c) { return c * 0x9E3779B1u; }
// bucket hash, independent of the order hash (a minimizer has a small order hash by construction)
__host__ __device__ inline uint32_t mm_bucket(uint32_t c) {
  uint32_t h = c ^ 0x5bd1e995u;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// Window j of P (bases j .. j+m-1): selection key (23 bits of order hash above 9 bits of
// position, so the smallest key is the leftmost smallest minimizer), canonical m-mer, strand.
// 9 bits: k_seed tags the keys of a whole item record (256 + w positions) this way.
constexpr uint32_t SEL_POS = 511u;
__host__ __device__ inline uint32_t window_key(const TableView& t, uint64_t P, uint64_t R, int j,
                                               uint32_t* canon, uint32_t* strand) {
  const uint32_t f = (uint32_t)(P >> (2 * (t.w - 1 - j))) & t.mmask;
  const uint32_t r = (uint32_t)(R >> (2 * j)) & t.mmask;
  const uint32_t c = f < r ? f : r;
  *canon = c;
  *strand = f < r ? 0u : 1u;
  return (mm_order(c) & ~SEL_POS) | (uint32_t)j;
}

// floor(hash / 2^32 * n_buckets) — n_buckets is a power of two, so that is a shift (a 32-bit multiply-high is a
// quarter-rate instruction, and k_seed is where the pipelined step's issue slots go)
__host__ __device__ inline uint32_t bucket_of(const TableView& t, uint32_t canon_mmer) {
  return mm_bucket(canon_mmer) >> t.bshift;
}
__host__ __device__ inline void finish_key(const TableView& t, uint64_t P, uint64_t R, uint32_t c,
                                           uint32_t s, uint32_t u, Key* key) {
  if (!t.canonical || P <= R) {
    key->tag = P << 1;
    key->flip = 0;
  } else {
    key->tag = (R << 1) | 1;
    key->flip = 1;
  }
  key->bucket = bucket_of(t, c);
  // walking forward along a strand moves the minimizer one base to the left: u falls, cls rises
  const uint32_t cls = s * (uint32_t)t.w + ((uint32_t)t.w - 1u - u);
  // every bit of the tag takes part: (k-1)-mers that differ only in their first bases (left
  // siblings, junction k-mers of one variant, error variants of a super-k-mer) share their low
  // 32 bits, and would share their home pair in every table size
  const uint32_t h = ((uint32_t)key->tag * 0x9E3779B1u) ^ ((uint32_t)(key->tag >> 32) * 0x85EBCA6Bu);
  key->frac = ((cls << 8) | (h >> 24)) * t.inv32;
  key->cls = cls;
  key->hsub = h;
}

// One thread computes the whole key (k_seed, table build, batched lookups).
__host__ __device__ inline Key make_key(const TableView& t, uint64_t P) {
  const uint64_t R = revcomp(P, t.k - 1);
  uint32_t best = ~0u, bc = 0, bs = 0;
  for (int j = 0; j < t.w; ++j) {
    uint32_t c, s;
    const uint32_t key = window_key(t, P, R, j, &c, &s);
    if (key < best) { best = key; bc = c; bs = s; }
  }
  Key key;
  finish_key(t, P, R, bc, bs, best & SEL_POS, &key);
  return key;
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

// Kernels instantiated for one k (K = 31: what `jellyfish count -m 31` and every bundled file
// use) see the geometry of the table view as compile-time constants: shifts, masks and the window
// loops fold, and a dozen scalar registers stop being live across the kernel.  K = 0: as given.
template <int K>
__device__ inline TableView specialized_view(const TableView& v) {
  TableView t = v;
  if constexpr (K != 0) {
    constexpr int M = (K - 1 < 15 ? K - 1 : 15) - (((K - 1 < 15 ? K - 1 : 15) & 1) ? 0 : 1);
    constexpr int W = K - M;
    t.k = K;
    t.m = M;
    t.w = W;
    t.kmask = K >= 32 ? ~0ull : ((1ull << (2 * K)) - 1);
    t.pmask = (1ull << (2 * (K - 1))) - 1;
    t.mmask = (uint32_t)((1ull << (2 * M)) - 1);
    t.inv32 = (uint32_t)((1ull << 32) / ((uint64_t)2 * W * 256));
    uint32_t cs = 1;
    while ((1u << cs) < 2u * (uint32_t)W) ++cs;
    t.cshift = cs;
  }
  return t;
}

// k-mer i of a 2-bit packed target (32 bases per word, first base most significant; the
// packed form carries one extra zero word, so words[w + 1] is always readable)
__device__ inline uint64_t kmer_from_words(const uint64_t* words, uint32_t i, int k) {
  const uint32_t w = i >> 5, sh = (i & 31) * 2;
  const uint64_t hi = words[w], lo = words[w + 1];
  const uint64_t x = sh ? ((hi << sh) | (lo >> (64 - sh))) : hi;
  return x >> (64 - 2 * k);
}

// The same key for a wave-uniform P with the windows spread over the lanes (k_dfs: one
// lookup per walk step, latency matters).  Every lane returns the full key.
// best = selection key of the minimizer window (its position in the low bits), bc / bs = its
// canonical m-mer and strand.
__device__ inline void minimizer_wave(const TableView& t, uint64_t P, uint64_t R, uint32_t* best_out,
                                      uint32_t* bc_out, uint32_t* bs_out) {
  const int lane = lane_id();
  uint32_t c = 0, s = 0;
  uint32_t mine = ~0u;
  if (lane < t.w) mine = window_key(t, P, R, lane, &c, &s);
  // minimum over the lanes < w (w <= 17): a row_shr DPP tree leaves each 16-lane row's
  // minimum in its last lane
  uint32_t v = mine;
  {
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xF, 0xF, false); v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xF, 0xF, false); v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xF, 0xF, false); v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xF, 0xF, false); v = o < v ? o : v;
  }
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
  const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
  const uint32_t best = b0 < b1 ? b0 : b1;
  const int u = (int)(best & SEL_POS);
  *best_out = best;
  *bc_out = (uint32_t)__builtin_amdgcn_readlane((int)c, u);
  *bs_out = (uint32_t)__builtin_amdgcn_readlane((int)s, u);
}
__device__ inline Key make_key_wave(const TableView& t, uint64_t P) {
  const uint64_t R = revcomp(P, t.k - 1);
  uint32_t best, bc, bs;
  minimizer_wave(t, P, R, &best, &bc, &bs);
  Key key;
  finish_key(t, P, R, bc, bs, best & SEL_POS, &key);
  return key;
}

// Sliding-window form used by k_seed (one lane per base position q < 512 of a sequence): the
// selection key of the m-mer whose 64 leading bits are `bits`, tagged with its position.
// min over q .. q+w-1 of these keys picks the same window as make_key (smallest order hash,
// leftmost on ties).
__device__ inline uint32_t mmer_scan_key(const TableView& t, uint64_t bits, uint32_t q) {
  const uint32_t f = (uint32_t)(bits >> (64 - 2 * t.m));
  const uint32_t r = revcomp32(f, t.m);
  const uint32_t c = f < r ? f : r;
  return (mm_order(c) & ~SEL_POS) | q;
}
// Key of P given the window index u chosen by the scan.
__device__ inline Key key_from_window(const TableView& t, uint64_t P, uint32_t u) {
  const uint64_t R = revcomp(P, t.k - 1);
  uint32_t c, s;
  (void)window_key(t, P, R, (int)u, &c, &s);
  Key key;
  finish_key(t, P, R, c, s, u, &key);
  return key;
}

struct __attribute__((packed, aligned(4))) DirPair { uint32_t lo, hi; };

__device__ inline uint32_t pick4(uint4 v, uint32_t i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// Home slot of a key in a bucket of S slots (0 for an empty bucket).  Always even: the home
// PAIR (home, home + 1) is one aligned 32-byte piece of a line, and the build guarantees that
// every key sits inside its home pair (a bucket that cannot is doubled until it can).
__device__ inline uint64_t home_slot(const TableView& t, const Key& key, uint64_t S) {
  // Three layouts, one multiply: the lanes of a wave meet all three (as branches every lane paid for all of them)
  //  small bucket (S < NC): floor(frac / 2^32 * S), order preserving
  //  crowded bucket (an odd multiple of NC: many near-identical super-k-mers behind one minimizer pile up in a
  //                  few classes): plain hashing over all its pairs
  //  class mode (S = q * NC, q even): the class owns q slots
  const uint32_t q = (uint32_t)(S >> t.cshift);
  const bool small = q == 0, crowded = (q & 1u) != 0;
  const uint32_t h = small ? key.frac : key.hsub;
  const uint32_t n = small ? (uint32_t)S : (crowded ? (uint32_t)(S >> 1) : q);
  const uint32_t m = __umulhi(h, n);
  const uint64_t in_class = (uint64_t)key.cls * q + (m & ~1u);
  return small ? (uint64_t)(m & ~1u) : (crowded ? ((uint64_t)m << 1) : in_class);
}
// Two-choice probing: a key whose home pair is taken moves to a SECOND pair of its bucket, chosen
// by an independent hash, and only from there on probes linearly (insert and lookup follow the
// same sequence).  Only buckets that have used up their doublings ever place a key outside its
// home pair — crowded ones (an odd multiple of NC slots: plain hashing over all pairs) and small
// ones whose keys share class and leading hash bits; four slots then hold almost every such key,
// where linear probing from the home pair needed distances of 8-12.
__device__ inline bool bucket_is_crowded(const TableView& t, uint64_t S) { return ((S >> t.cshift) & 1ull) != 0; }
__device__ inline uint64_t second_pair(uint64_t tag, uint64_t S, uint64_t home) {
  uint32_t h = ((uint32_t)(tag >> 32) * 0xC2B2AE35u) ^ ((uint32_t)tag * 0x27D4EB2Fu);
  h ^= h >> 15;
  const uint32_t np = (uint32_t)(S >> 1);                  // pairs (>= 2); never the home pair again
  uint32_t p = (uint32_t)(home >> 1) + 1u + __umulhi(h * 0x165667B1u, np - 1u);
  if (p >= np) p -= np;
  return (uint64_t)p << 1;
}
// slots of a bucket from its two directory words
__device__ inline uint64_t bucket_slots(uint32_t lo, uint32_t hi) { return 2ull * (hi - lo); }

__device__ inline uint4 slot_counts(uint4 a) {
  return make_uint4(a.z & 0xFFFFu, a.z >> 16, a.w & 0xFFFFu, a.w >> 16);
}

// Counts of the four members of the group in slot order (zeros if absent), as stored (u16,
// possibly COUNT_ESCAPE), given the two slots a0, a1 of the home pair at base[idx], already
// loaded.  Normally (max_probe == 2) that is the whole search; a table whose build had to
// give up on the pair bound keeps probing up to max_probe slots.  *fetches counts slots read.
__device__ inline uint4 bucket_resolve2(const TableView& t, const Key& key, const Slot* base,
                                        uint64_t S, uint64_t idx, uint4 a0, uint4 a1,
                                        uint32_t* fetches) {
  ++*fetches;
  const uint64_t t0 = ((uint64_t)a0.y << 32) | a0.x;
  if (t0 == key.tag) return slot_counts(a0);
  if (t0 == EMPTY) return make_uint4(0, 0, 0, 0);
  ++*fetches;
  const uint64_t t1 = ((uint64_t)a1.y << 32) | a1.x;
  if (t1 == key.tag) return slot_counts(a1);
  if (t1 == EMPTY) return make_uint4(0, 0, 0, 0);
  idx = S >= 4 ? second_pair(key.tag, S, idx) : idx + 2;
  for (uint32_t step = 2; step < t.max_probe && step < S; ++step) {
    if (idx >= S) idx = 0;
    const uint4 a = *reinterpret_cast<const uint4*>(base + idx);
    ++*fetches;
    const uint64_t tg = ((uint64_t)a.y << 32) | a.x;
    if (tg == key.tag) return slot_counts(a);
    if (tg == EMPTY) break;
    ++idx;
  }
  return make_uint4(0, 0, 0, 0);
}

__device__ inline uint4 bucket_lookup4(const TableView& t, const Key& key, uint32_t lo, uint32_t hi,
                                       uint32_t* fetches) {
  const uint64_t S = bucket_slots(lo, hi);
  if (S == 0) return make_uint4(0, 0, 0, 0);
  const Slot* base = t.slots + 2ull * lo;
  const uint64_t idx = home_slot(t, key, S);
  const uint4 a0 = *reinterpret_cast<const uint4*>(base + idx);
  const uint4 a1 = *reinterpret_cast<const uint4*>(base + idx + 1);
  return bucket_resolve2(t, key, base, S, idx, a0, a1, fetches);
}

__device__ inline uint4 table_lookup4(const TableView& t, const Key& key, uint32_t* fetches) {
  const DirPair d = *reinterpret_cast<const DirPair*>(t.dir + key.bucket);
  return bucket_lookup4(t, key, d.lo, d.hi, fetches);
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

// Child-base order + exact values for escaped counts of the group fetched for X[1:].
__device__ inline uint4 finish_children(const TableView& t, uint64_t X, uint32_t flip, uint4 c) {
  if (flip) c = make_uint4(c.w, c.z, c.y, c.x);
  if (c.x == COUNT_ESCAPE || c.y == COUNT_ESCAPE || c.z == COUNT_ESCAPE || c.w == COUNT_ESCAPE) {
    const uint64_t base = (X << 2) & t.kmask;
    if (c.x == COUNT_ESCAPE) c.x = overflow_count(t, base | 0);
    if (c.y == COUNT_ESCAPE) c.y = overflow_count(t, base | 1);
    if (c.z == COUNT_ESCAPE) c.z = overflow_count(t, base | 2);
    if (c.w == COUNT_ESCAPE) c.w = overflow_count(t, base | 3);
  }
  return c;
}

// Counts of X[1:]+A, +C, +G, +T (child-base order); g = key of X[1:].
__device__ inline uint4 forward_children_keyed(const TableView& t, uint64_t X, const Key& g,
                                               uint32_t* fetches) {
  return finish_children(t, X, g.flip, table_lookup4(t, g, fetches));
}
__device__ inline uint4 forward_children(const TableView& t, uint64_t X, uint32_t* fetches) {
  return forward_children_keyed(t, X, make_key(t, X & t.pmask), fetches);
}

// The same for a wave-uniform X (k_dfs), split in two so that the loads of the NEXT walk step
// can be in flight while the current one is still being booked: issue (key computed across
// the lanes; the directory word of the last bucket is kept in registers — consecutive walk
// steps mostly stay in it; the home pair is requested) and finish (compare, child order).
struct DirCache { uint32_t bucket, lo, hi; };
struct PendingLookup {
  uint64_t X;        // k-mer whose forward children were requested
  Key g;
  const Slot* base;
  uint64_t S, idx;
  uint4 a0, a1;
  bool valid;
};
__device__ inline void children_issue_wave(const TableView& t, uint64_t X, DirCache* dc,
                                           PendingLookup* p) {
  p->X = X;
  p->g = make_key_wave(t, X & t.pmask);
  if (p->g.bucket != dc->bucket) {
    const DirPair d = *reinterpret_cast<const DirPair*>(t.dir + p->g.bucket);
    dc->bucket = p->g.bucket; dc->lo = d.lo; dc->hi = d.hi;
  }
  p->S = bucket_slots(dc->lo, dc->hi);
  p->base = t.slots + 2ull * dc->lo;
  p->idx = home_slot(t, p->g, p->S);
  const Slot* b0 = (p->S ? p->base : t.slots) + p->idx;
  // The address is wave-uniform, but the request must stay a VECTOR load whose result is not
  // looked at before children_finish_wave: given a uniform address the compiler moves the eight
  // loaded words to scalar registers at once, i.e. waits for the memory right here, and the
  // latency the early request was meant to hide is paid in full.  A lane offset it cannot see
  // through (always 0) keeps the words in vector registers until they are needed.
  uint32_t lane_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  const uint4* q = reinterpret_cast<const uint4*>(b0) + lane_zero;
  p->a0 = q[0];
  p->a1 = q[1];
  p->valid = true;
}
__device__ inline uint32_t lane_u32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ inline uint64_t lane_u64(uint64_t v, uint32_t l) {
  return ((uint64_t)lane_u32((uint32_t)(v >> 32), l) << 32) | lane_u32((uint32_t)v, l);
}
// bucket of the group of the (k-1)-mer P (wave-uniform; the windows are spread over the lanes)
__device__ inline uint32_t bucket_of_wave(const TableView& t, uint64_t P) {
  const uint64_t R = revcomp(P, t.k - 1);
  uint32_t best, bc, bs;
  minimizer_wave(t, P, R, &best, &bc, &bs);
  return bucket_of(t, bc);
}

// ---- A bucket held in the wave's registers, with every slot's SUCCESSOR worked out in advance
// (k_dfs chain runs).  The k-mers of a chain share their minimizer for ~w/2 steps, and with it the
// bucket: lane j keeps slot j of that bucket.  A slot {tag, four counts} is everything get_child
// (km/utils/Jellyfish.py:55-72) looks at for the k-mers x with x[1:] = the slot's (k-1)-mer, so the
// lane that loads a slot also evaluates it, once, for all 64 slots at a time and off the walk's
// critical path: the threshold, the kept children, and — when exactly one child is kept — that
// child's base and count, the TAG OF THE GROUP THE WALK LOOKS UP NEXT, and whether the child sits
// in its home slot of the node set (a likely rejoin).  A chain step is then: compare the tag against
// the lanes (one v_cmp + ballot), read the hit lane's three words.  No hashing, no arithmetic on
// counts, no memory access, until the tag is not among the lanes: then either the minimizer has
// changed (the next bucket is loaded and evaluated) or the group does not exist.  A key never
// leaves its bucket (probing wraps inside it), so "some slot of the bucket holds the tag" is exactly
// the table's own answer.  Buckets of more than 64 * BUCKET_LANES_SETS slots are not held.
constexpr uint32_t BUCKET_LANES_SETS = 7;          // slots per lane: buckets of up to 576 slots are held (a crowded
                                                   // bucket of the usual size, 4 * 128 + NC = 544 slots, fits)
constexpr uint32_t SLOT_SINGLE = 4u, SLOT_HINT = 8u;
struct BucketLanes {
  uint64_t tag[BUCKET_LANES_SETS];                           // lane j, set i: slot 64 i + j (EMPTY past the end)
  uint32_t info[BUCKET_LANES_SETS];                          // child base | SLOT_SINGLE | SLOT_HINT | count << 16
  // (round 3 also kept the tag of the group the walk looks up next: two more registers per set, i.e. buckets of 256
  // slots instead of 576 in the same registers; the walk works that tag out from the child, a dozen scalar instructions)
  uint32_t bucket, S;    // wave-uniform
  bool valid, resident;
};
struct ChildRule {       // what get_child needs beside the counts (wave-uniform)
  double ratio;
  double nc;             // (double)n_cutoff
  uint64_t thr_below;    // sums below it share the threshold thr_T (threshold_shortcut)
  uint32_t thr_T;
};
// One slot evaluated by its lane.  `hintf(k-mer)`: is it probably a node of the walk (the hint only).
// (returns the single kept child, 0 if there is none: the hint is worked out by the caller, for all sets of a bucket
// together — its LDS reads are then in flight side by side instead of set after set)
__device__ inline uint64_t slot_successor(const TableView& t, const ChildRule& r, uint64_t tag, uint64_t zw, uint32_t* info);
// lo / hi: the bucket's directory words (wave-uniform)
template <class HintF>
__device__ inline void bucket_load_wave(const TableView& t, const ChildRule& r, uint32_t bucket, uint32_t lo, uint32_t hi,
                                        HintF& hintf, BucketLanes* b, uint32_t* fetches) {
  const uint32_t S = 2u * (hi - lo);
  b->bucket = bucket; b->S = S; b->valid = true; b->resident = S <= 64u * BUCKET_LANES_SETS;
  if (b->resident) {
    const uint32_t lane = (uint32_t)lane_id();
    uint64_t zw[BUCKET_LANES_SETS];
    // all the loads first (independent), then the evaluation
#pragma unroll
    for (uint32_t i = 0; i < BUCKET_LANES_SETS; ++i) {
      b->tag[i] = EMPTY; b->info[i] = 0; zw[i] = 0;
      if (64u * i < S) {                             // wave-uniform
        if (64u * i + lane < S) {
          const uint4 v = *reinterpret_cast<const uint4*>(t.slots + 2ull * lo + 64u * i + lane);
          b->tag[i] = ((uint64_t)v.y << 32) | v.x;
          zw[i] = ((uint64_t)v.w << 32) | v.z;
        }
      }
    }
    uint64_t child[BUCKET_LANES_SETS];
#pragma unroll
    for (uint32_t i = 0; i < BUCKET_LANES_SETS; ++i) {
      child[i] = 0;
      if (64u * i < S) {                             // wave-uniform
        if (b->tag[i] != EMPTY) child[i] = slot_successor(t, r, b->tag[i], zw[i], &b->info[i]);
      }
    }
    // the hints of every set, branch-free (the reads of one set do not wait for those of the set before it)
    bool hint[BUCKET_LANES_SETS];
#pragma unroll
    for (uint32_t i = 0; i < BUCKET_LANES_SETS; ++i) hint[i] = (64u * i < S) ? hintf(child[i]) : false;
#pragma unroll
    for (uint32_t i = 0; i < BUCKET_LANES_SETS; ++i) if ((b->info[i] & SLOT_SINGLE) && hint[i]) b->info[i] |= SLOT_HINT;
    *fetches += S;
  }
}
// the slot of the group `tag` in a resident bucket
struct SlotHit { uint32_t info; bool hit; };
// (nested, so that a hit leaves through one branch: a flat loop over the sets makes the compiler chain
// an exit flag through every later set — sixteen taken branches behind a hit in the first one)
// (no test of the set against the bucket's size: the sets a bucket does not reach hold EMPTY tags, which no
// group has — nine scalar compares a step otherwise, and a miss, which looks at every set, is a bucket change)
template <uint32_t I>
__device__ inline void bucket_find_from(const BucketLanes& b, uint64_t tag, SlotHit& h) {
  if constexpr (I < BUCKET_LANES_SETS) {
    const unsigned long long hit_ = __ballot(b.tag[I] == tag);
    if (hit_) {
      const uint32_t l_ = (uint32_t)__ffsll((long long)hit_) - 1;
      h.info = lane_u32(b.info[I], l_);
      h.hit = true;
    } else {
      bucket_find_from<I + 1>(b, tag, h);
    }
  }
}
__device__ inline SlotHit bucket_find_wave(const BucketLanes& b, uint64_t tag) {
  SlotHit h;
  h.hit = false; h.info = 0;
  bucket_find_from<0>(b, tag, h);
  return h;
}
// canonical tag of the group of the (k-1)-mer P (R = its reverse complement)
__device__ inline uint64_t group_tag(const TableView& t, uint64_t P, uint64_t R, uint32_t* flip) {
  // (selects by arithmetic: as branches these cost the one wave that walks a chain more than the work)
  uint64_t diff;
  const bool rev = __builtin_usubl_overflow(R, P, &diff) && t.canonical;   // R < P
  const uint64_t m = 0ull - (uint64_t)rev;
  *flip = (uint32_t)rev;
  return (((R << 1) | 1ull) & m) | ((P << 1) & ~m);
}

__device__ inline uint4 uniform4(uint4 v) {
  return make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y),
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)v.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)v.w));
}
__device__ inline uint4 children_finish_wave(const TableView& t, const PendingLookup& p,
                                             uint32_t* fetches) {
  uint4 c = make_uint4(0, 0, 0, 0);
  if (p.S) c = bucket_resolve2(t, p.g, p.base, p.S, p.idx, uniform4(p.a0), uniform4(p.a1), fetches);
  return finish_children(t, p.X, p.g.flip, c);
}
__device__ inline uint4 forward_children_wave(const TableView& t, uint64_t X, DirCache* dc,
                                              uint32_t* fetches) {
  PendingLookup p;
  children_issue_wave(t, X, dc, &p);
  return children_finish_wave(t, p, fetches);
}

// g = key of X[:-1]
__device__ inline uint32_t query_one_keyed(const TableView& t, uint64_t X, const Key& g,
                                           uint32_t* fetches) {
  uint4 c = table_lookup4(t, g, fetches);
  uint32_t s = (uint32_t)(X & 3);
  uint32_t v = pick4(c, g.flip ? 3 - s : s);
  if (v == COUNT_ESCAPE) v = overflow_count(t, X);
  return v;
}
__device__ inline uint32_t query_one(const TableView& t, uint64_t X, uint32_t* fetches) {
  const Key g = make_key(t, X >> 2);
  uint4 c = table_lookup4(t, g, fetches);
  uint32_t s = (uint32_t)(X & 3);
  uint32_t v = pick4(c, g.flip ? 3 - s : s);
  if (v == COUNT_ESCAPE) v = overflow_count(t, X);
  return v;
}

// Children kept by Jellyfish.get_child: count >= max(sum * cutoff, n_cutoff)
// evaluated as Python does (float64 product, exact int/float comparison):
// km/utils/Jellyfish.py:69-72.  Returns a 4-bit mask, bit c = child base c.
// The threshold as an integer: a count is kept iff it is >= T, unless *none (the threshold lies
// above every 32-bit count, or is NaN).
// (nc = (double)n_cutoff, converted once by the caller: the kernels get it as a kernel argument — there is no scalar
// int64 -> double conversion, so every lane would do it)
__host__ __device__ inline uint32_t threshold_of(double t, double nc, bool* none) {
  const double thr = (nc > t) ? nc : t;     // Python max(t, nc)
  // an integer count is >= thr  <=>  it is >= ceil(thr): four integer compares
  const double ct = ceil(thr);
  *none = !(ct < 4294967296.0);
  return ct <= 0.0 ? 0u : (*none ? 0xFFFFFFFFu : (uint32_t)ct);
}
__host__ __device__ inline uint32_t child_threshold(uint64_t sum, double ratio, int64_t n_cutoff, bool* none) {
  return threshold_of((double)sum * ratio, (double)n_cutoff, none);
}
__device__ inline uint32_t child_mask(uint4 c, double ratio, double nc) {
  bool none;
  uint32_t T;
  // the sum of four counts fits 32 bits unless one of them came from the side table of large counts (>= 2^30,
  // conservatively): one conversion instead of two + ldexp + add, the same double either way
  if (!__any((int)(((c.x | c.y | c.z | c.w) >> 30) != 0))) T = threshold_of((double)(c.x + c.y + c.z + c.w) * ratio, nc, &none);
  else T = threshold_of((double)((uint64_t)c.x + c.y + c.z + c.w) * ratio, nc, &none);
  if (none) return 0u;
  return (c.x >= T ? 1u : 0u) | (c.y >= T ? 2u : 0u) | (c.z >= T ? 4u : 0u) | (c.w >= T ? 8u : 0u);
}
__device__ inline uint32_t child_mask(uint4 c, double ratio, int64_t n_cutoff) { return child_mask(c, ratio, (double)n_cutoff); }
// device_common.h: BucketLanes.  What Jellyfish.get_child (km/utils/Jellyfish.py:55-72) makes of one
// slot, for the k-mers x whose suffix x[1:] is the slot's (k-1)-mer in the orientation its side bit names.
__device__ inline uint64_t slot_successor(const TableView& t, const ChildRule& r, uint64_t tag, uint64_t zw, uint32_t* info) {
  const uint32_t z = (uint32_t)zw, w = (uint32_t)(zw >> 32);
  const uint32_t s0 = z & 0xFFFFu, s1 = z >> 16, s2 = w & 0xFFFFu, s3 = w >> 16;   // slot order
  const bool esc = s0 == COUNT_ESCAPE || s1 == COUNT_ESCAPE || s2 == COUNT_ESCAPE || s3 == COUNT_ESCAPE;
  const uint64_t sum = (uint64_t)(s0 + s1 + s2 + s3);
  uint32_t T = r.thr_T;
  bool none = false;
  if (sum >= r.thr_below) T = threshold_of((double)(uint32_t)sum * r.ratio, r.nc, &none);   // (four 16-bit counts)
  uint32_t m4 = (s0 >= T ? 1u : 0u) | (s1 >= T ? 2u : 0u) | (s2 >= T ? 4u : 0u) | (s3 >= T ? 8u : 0u);
  if (none) m4 = 0;
  const bool single = !esc && m4 != 0 && (m4 & (m4 - 1)) == 0;
  *info = 0;
  if (!single) return 0;
  const uint32_t si = (uint32_t)__ffs((int)m4) - 1;
  const uint32_t side = t.canonical ? (uint32_t)(tag & 1) : 0u;
  const uint32_t c = side ? 3u - si : si;                        // child base
  const uint32_t cnt = (uint32_t)(zw >> (16 * si)) & 0xFFFFu;
  const uint64_t G = tag >> 1;
  // P: the (k-1)-mer as the walk reads it (x[1:])
  const uint64_t P = (t.canonical && side) ? revcomp(G, t.k - 1) : G;
  const uint64_t child = (P << 2) | c;                           // x[1:] + c, 2k bits
  *info = c | SLOT_SINGLE | (cnt << 16);
  return child;
}

// Sums below *below* all have the threshold T = ceil(n_cutoff) (sum * ratio <= n_cutoff: the float64
// product is monotone in the sum for ratio >= 0), so that a walk step can skip the float64 arithmetic
// for them.  0 = no such shortcut (negative or NaN ratio).
__host__ inline void threshold_shortcut(double ratio, int64_t n_cutoff, uint64_t* below, uint32_t* T) {
  *below = 0; *T = 0;
  if (!(ratio >= 0.0) || ratio > 1.7e308) return;
  const double nc = (double)n_cutoff;
  auto same = [&](uint64_t sum) { return (double)sum * ratio <= nc; };
  if (!same(0)) return;
  uint64_t lo = 0, hi = 1ull << 36;                      // sums of four 32-bit counts stay below 2^34
  if (same(hi)) lo = hi;
  else while (hi - lo > 1) { const uint64_t mid = lo + (hi - lo) / 2; if (same(mid)) lo = mid; else hi = mid; }
  bool none;
  const uint32_t t = child_threshold(0, ratio, n_cutoff, &none);
  if (none) return;
  *below = lo + 1; *T = t;
}

}  // namespace kmd
