// table_kernels.h — build and batched probes of the HBM-resident table.
#pragma once
#include "device_common.h"

namespace kmd {

// Fill the table with empty slots.  16 B per lane, fully coalesced.
__global__ void k_table_init(Slot* slots, uint64_t n_slots) {
  uint4* p = reinterpret_cast<uint4*>(slots);
  const uint4 empty = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots;
       i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = empty;
}

// Number of records whose count does not fit the 16-bit slot field.
__global__ void k_count_big(const uint32_t* counts, uint64_t n, unsigned long long* out) {
  unsigned long long local = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    local += counts[i] >= COUNT_ESCAPE ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// Raw `binary/sorted` records as they sit in the file ([kb little-endian key bytes][cb count
// bytes], kb + cb <= 12) -> key / count arrays.  meta[0] += records with a non-zero count.
__global__ void k_unpack_records(const unsigned char* raw, uint64_t n, uint32_t kb, uint32_t cb,
                                 uint64_t* keys, uint32_t* counts, unsigned long long* meta) {
  unsigned long long nz = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t key = 0;
    uint32_t cnt = 0;
    if (kb == 8 && cb == 4) {                 // 12-byte records: three aligned dwords
      const uint32_t* w = reinterpret_cast<const uint32_t*>(raw) + 3 * i;
      key = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
      cnt = w[2];
    } else {
      const unsigned char* r = raw + i * (kb + cb);
      for (uint32_t b = 0; b < kb; ++b) key |= (uint64_t)r[b] << (8 * b);
      for (uint32_t b = 0; b < cb; ++b) cnt |= (uint32_t)r[kb + b] << (8 * b);
    }
    keys[i] = key;
    counts[i] = cnt;
    nz += cnt != 0;
  }
  for (int o = 32; o > 0; o >>= 1) nz += __shfl_xor(nz, o);
  if ((threadIdx.x & 63) == 0 && nz) atomicAdd(meta, nz);
}

// Orientations under which a stored record is entered: 0 (a non-canonical key in a canonical
// database is unreachable by query(), as in the reference), 1 (palindrome / non-canonical
// database) or 2.
__device__ inline int record_orientations(uint64_t K, int k, int canonical, uint64_t* R) {
  *R = K;
  if (!canonical) return 1;
  *R = revcomp(K, k);
  if (*R < K) return 0;
  return (*R == K) ? 1 : 2;
}

// Pass 1 of the build: entries per minimizer bucket.
__global__ void k_dir_count(TableView t, const uint64_t* keys, const uint32_t* counts, uint64_t n,
                            uint32_t* cnt) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t K = keys[i];
    if (counts[i] == 0) continue;
    uint64_t R;
    const int n_or = record_orientations(K, t.k, t.canonical, &R);
    for (int o = 0; o < n_or; ++o) {
      const Key key = make_key(t, (o ? R : K) >> 2);
      atomicAdd(&cnt[key.bucket], 1u);
    }
  }
}

// Capacity rules (slots; stored in units of 2 slots).  Small buckets: `unit` slots per entry,
// fewer than NC slots.  Class mode: a multiple of 2*NC slots (q even), at least 2*NC.
__device__ inline uint64_t shape_capacity(uint64_t cap, uint64_t NC) {
  cap += cap & 1;
  if (cap >= NC) cap = (cap + 2 * NC - 1) / (2 * NC) * (2 * NC);
  return cap;
}
// caps[] word: capacity in units of 2 slots (29 bits), how often the bucket was doubled
// (2 bits), and the flag "some key did not fit its home pair in this round".
constexpr uint32_t CAP_GROW = 0x80000000u;
constexpr uint32_t CAP_GEN_SHIFT = 29;
constexpr uint32_t CAP_SIZE = (1u << CAP_GEN_SHIFT) - 1;
constexpr uint32_t CAP_MAX_GEN = 2;          // doublings under the home-pair rule (load 1/2 -> 1/8)
constexpr uint32_t CAP_HARD_GEN = 3;         // one more for a bucket in which a key fits neither of its
                                             // two pairs (device_common.h: second_pair); then it probes on
__device__ inline uint32_t cap_gen(uint32_t c) { return (c >> CAP_GEN_SHIFT) & 3u; }

// Entry count -> initial capacity, in place.
__global__ void k_dir_capacity(uint32_t* caps, uint64_t n, uint32_t unit, uint32_t cshift) {
  const uint64_t NC = 1ull << cshift;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t pairs = shape_capacity((uint64_t)caps[i] * unit, NC) >> 1;
    // a bucket too large for the size field starts saturated (it could not double anyway)
    caps[i] = pairs > CAP_SIZE / 8 ? (uint32_t)(pairs > CAP_SIZE ? CAP_SIZE : pairs) | (CAP_HARD_GEN << CAP_GEN_SHIFT)
                                   : (uint32_t)pairs;
  }
}

// Double every flagged bucket that may still grow (and clear the flags).
// lean_crowded: a class-mode bucket that would double a second time is turned into a plain two-choice table at
// the size it has (load 1/4) instead of at twice that (load 1/8): half the slots for k_dfs to hold in its lanes
__global__ void k_dir_grow(uint32_t* caps, uint64_t n, uint32_t cshift, int lean_crowded) {
  const uint64_t NC = 1ull << cshift;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = caps[i];
    if (!(c & CAP_GROW)) continue;
    const uint32_t gen = cap_gen(c);
    // doubling; the extra step of a bucket in which a key fitted neither of its two pairs
    // (a few hundred buckets in 10^8) quadruples it, so that the rebuild settles it for good
    uint64_t mult = gen >= CAP_MAX_GEN ? 8ull : 4ull;
    if (lean_crowded && gen + 1 == CAP_MAX_GEN && 2ull * (c & CAP_SIZE) >= 2 * NC) mult = 2ull;
    uint64_t slots = shape_capacity(mult * (c & CAP_SIZE), NC);
    // the last doubling of a class-mode bucket turns it into a plain hash table over all its
    // pairs (an odd multiple of NC, see home_slot): its classes are too unevenly filled
    if (gen + 1 >= CAP_MAX_GEN && slots >= 2 * NC) slots += NC;
    caps[i] = (uint32_t)(slots >> 1) | ((gen + 1) << CAP_GEN_SHIFT);
  }
}

// caps -> dir (to be scanned in place) and the total, which must fit the 32-bit directory.
__global__ void k_dir_copy(const uint32_t* caps, uint32_t* dir, uint64_t n,
                           unsigned long long* total_pairs) {
  unsigned long long local = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = caps[i] & CAP_SIZE;
    dir[i] = c;
    local += c;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(total_pairs, local);
}

// Exclusive prefix sum of the bucket counts, in place (three passes over chunks of
// SCAN_CHUNK words; the array is zero-padded to a whole number of chunks).
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER_THREAD;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wave_sum[SCAN_THREADS / 64];
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  uint32_t inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t x = __shfl_up(inc, o);
    if (lane >= o) inc += x;
  }
  if (lane == 63) wave_sum[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (int q = 0; q < SCAN_THREADS / 64; ++q) {
    const uint32_t ws = wave_sum[q];
    if (q < wave) before += ws;
    all += ws;
  }
  __syncthreads();
  *total = all;
  return before + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t* data, uint32_t* sums) {
  const uint4* p = reinterpret_cast<const uint4*>(data + (uint64_t)blockIdx.x * SCAN_CHUNK);
  uint32_t v = 0;
  for (int q = 0; q < SCAN_PER_THREAD / 4; ++q) {
    const uint4 x = p[q * SCAN_THREADS + threadIdx.x];
    v += x.x + x.y + x.z + x.w;
  }
  uint32_t total;
  (void)block_exclusive_scan(v, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one block: sums[0..n) -> exclusive prefix, in place
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(uint32_t* sums, uint32_t n) {
  const uint32_t per = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  const uint32_t lo = threadIdx.x * per;
  const uint32_t hi = lo + per < n ? lo + per : n;
  uint32_t v = 0;
  for (uint32_t i = lo; i < hi; ++i) v += sums[i];
  uint32_t total;
  uint32_t run = block_exclusive_scan(v, &total);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t x = sums[i];
    sums[i] = run;
    run += x;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t* data, const uint32_t* sums) {
  uint4* p = reinterpret_cast<uint4*>(data + (uint64_t)blockIdx.x * SCAN_CHUNK) +
             threadIdx.x * (SCAN_PER_THREAD / 4);
  uint4 x[SCAN_PER_THREAD / 4];
  uint32_t v = 0;
  for (int q = 0; q < SCAN_PER_THREAD / 4; ++q) {
    x[q] = p[q];
    v += x[q].x + x[q].y + x[q].z + x[q].w;
  }
  uint32_t total;
  uint32_t run = sums[blockIdx.x] + block_exclusive_scan(v, &total);
  for (int q = 0; q < SCAN_PER_THREAD / 4; ++q) {
    uint4 o;
    o.x = run; run += x[q].x;
    o.y = run; run += x[q].y;
    o.z = run; run += x[q].z;
    o.w = run; run += x[q].w;
    p[q] = o;
  }
}

// Exact values of the counts that do not fit 16 bits, keyed by the k-mer as stored.
__global__ void k_ovf_insert(const uint64_t* keys, const uint32_t* counts, uint64_t n, int k,
                             int canonical, OvfSlot* ovf, uint64_t n_ovf, unsigned int* err) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t K = keys[i];
    const uint32_t v = counts[i];
    if (v < COUNT_ESCAPE) continue;
    uint64_t R;
    if (record_orientations(K, k, canonical, &R) == 0) continue;
    bool placed = false;
    uint64_t oi = n_ovf ? slot_index(K, n_ovf) : 0;
    for (uint64_t step = 0; step < n_ovf; ++step) {
      if (atomicCAS(&ovf[oi].count, 0u, v) == 0u) { ovf[oi].kmer = K; placed = true; break; }
      if (++oi == n_ovf) oi = 0;
    }
    if (!placed) atomicExch(err, 1u);
  }
}

// Dry round: which buckets would not keep every key in its home pair?  Same key / directory /
// home computation as the insert, but instead of the 10 GB of slots it touches one byte per
// pair (entries per home pair, four counters per word).  Conservative: siblings that will share
// a slot are counted separately.  A bucket whose pair receives a third entry is flagged in
// caps[] while it may still double (meta[2] counts flagged buckets).
__global__ void k_table_dry(TableView t, const uint64_t* keys, const uint32_t* counts, uint64_t n,
                            uint32_t* caps, uint32_t* pair_ctr, unsigned long long* meta) {
  unsigned long long flagged = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t K = keys[i];
    if (counts[i] == 0) continue;
    uint64_t R;
    const int n_or = record_orientations(K, t.k, t.canonical, &R);
    for (int o = 0; o < n_or; ++o) {
      const Key g = make_key(t, (o ? R : K) >> 2);
      const uint32_t lo = t.dir[g.bucket], hi = t.dir[g.bucket + 1];
      const uint64_t S = bucket_slots(lo, hi);
      if (S == 0) continue;
      const uint64_t pair = (uint64_t)lo + (home_slot(t, g, S) >> 1);
      const uint32_t sh = 8u * (uint32_t)(pair & 3);
      const uint32_t old = atomicAdd(&pair_ctr[pair >> 2], 1u << sh);
      if (((old >> sh) & 0xFFu) >= 2u && cap_gen(caps[g.bucket]) < CAP_MAX_GEN)
        flagged += !(atomicOr(&caps[g.bucket], CAP_GROW) & CAP_GROW);
    }
  }
  for (int o = 32; o > 0; o >>= 1) flagged += __shfl_xor(flagged, o);
  if ((threadIdx.x & 63) == 0 && flagged) atomicAdd(&meta[2], flagged);
}

// Insert pass: one record per thread, entered under the group of each orientation into the
// home pair of its bucket.  A key whose home pair is taken flags its bucket in caps[]
// (meta[2] counts flagged buckets) — the host doubles those buckets and rebuilds — unless the
// bucket has used up its doublings (a minimizer shared by very many similar k-mers: real
// data at high coverage) or this is the final round (final != 0): then the key moves on to its
// second pair and probes linearly from there; such a bucket goes on the list of k_table_settle
// (meta[6] buckets, meta[7] their slots).  meta[0] counts occupied slots, meta[1] != 0 reports a
// full bucket, meta[3] the largest probe distance the race produced.
__global__ void k_table_insert(TableView t, Slot* slots, const uint64_t* keys,
                               const uint32_t* counts, uint64_t n, uint32_t* caps, int final,
                               unsigned long long* meta, uint32_t* settle_bits, uint32_t* settle_list,
                               uint32_t settle_cap) {
  unsigned long long claimed = 0;          // slots this thread occupied (summed per wave at the end:
                                           // one same-address atomic per slot would serialise the kernel)
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t K = keys[i];
    const uint32_t v = counts[i];
    if (v == 0) continue;
    uint64_t R;
    const int n_or = record_orientations(K, t.k, t.canonical, &R);
    for (int o = 0; o < n_or; ++o) {
      const uint64_t O = o ? R : K;
      const Key g = make_key(t, O >> 2);
      uint32_t s = (uint32_t)(O & 3);
      if (g.flip) s = 3 - s;
      const uint32_t lo = t.dir[g.bucket], hi = t.dir[g.bucket + 1];
      const uint64_t S = bucket_slots(lo, hi);
      Slot* base = slots + 2ull * lo;
      uint64_t idx = home_slot(t, g, S);
      const uint64_t home = idx;
      const uint32_t gen = cap_gen(caps[g.bucket]);
      const bool may_grow = !final && gen < CAP_MAX_GEN;
      bool done = false;
      for (uint64_t step = 0; step < S; ++step) {
        if (step == 2 && may_grow) break;               // the pair is taken (three groups want it, whoever comes
                                                        // first): grow this bucket
        if (step == 2 && S >= 4) idx = second_pair(g.tag, S, home);             // two-choice (device_common.h)
        // (whether a key fits NEITHER of its pairs depends on who came first: k_table_settle decides the one
        // further doubling such a bucket may get, from the layout it makes of the bucket's keys)
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&base[idx].tag),
                                           (unsigned long long)EMPTY, (unsigned long long)g.tag);
        claimed += old == EMPTY;
        if (old == EMPTY || old == g.tag) {
          base[idx].c[s] = (uint16_t)(v >= COUNT_ESCAPE ? COUNT_ESCAPE : v);
          if (step >= 2) {
            atomicMax(&meta[3], (unsigned long long)step);
            // a key outside its home pair: where it ended up depended on who came first.  The bucket goes on
            // the list of k_table_settle, which lays it out again as a function of its keys alone.
            if (settle_bits && !((atomicOr(&settle_bits[g.bucket >> 5], 1u << (g.bucket & 31)) >> (g.bucket & 31)) & 1u)) {
              const unsigned long long at = atomicAdd(&meta[6], 1ull);
              if (at < settle_cap) settle_list[at] = g.bucket;
              atomicAdd(&meta[7], (unsigned long long)S);
            }
          }
          done = true;
          break;
        }
        if (++idx == S) idx = 0;
      }
      if (!done) {
        if (may_grow) {
          if (!(atomicOr(&caps[g.bucket], CAP_GROW) & CAP_GROW)) atomicAdd(&meta[2], 1ull);
        } else {
          atomicExch(reinterpret_cast<unsigned int*>(&meta[1]), 1u);
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) claimed += __shfl_xor(claimed, o);
  if ((threadIdx.x & 63) == 0 && claimed) atomicAdd(&meta[0], claimed);
}

// Settle pass.  The insert above is a race: in a bucket that has used up its doublings, which of the groups
// competing for a pair keeps it, and where the others end up along their probe sequences, depends on which thread
// came first — so the same records gave a table with max_probe 4 on one build and 5 on the next.  This pass lays
// every bucket that holds a key outside its home pair out AGAIN as a function of its keys alone: one block per
// bucket reads the occupied slots, sorts the groups by tag, clears the bucket and enters them one after the other
// in that order — first every group into its home pair while there is room, then the others along the rest of
// the insert's probe sequence (second pair, linear from there) — counts following their groups.  Lookups are unchanged (same sequence, no holes before a key).  meta[3] = largest probe
// distance over all settled buckets; a bucket too large for the block's LDS is left as the race built it and only
// measured (meta[5] counts those).
__global__ __launch_bounds__(256) void k_table_settle(TableView t, Slot* slots, const uint32_t* list, uint32_t n_list,
                                                      uint32_t lds_bytes, uint32_t* caps, int final,
                                                      unsigned long long* meta) {
  extern __shared__ __align__(16) unsigned char settle_lds[];
  __shared__ uint32_t n_e_s, n_gather_s, maxstep_s, unplaced_s;
  const uint32_t tid = threadIdx.x;
  const uint32_t bucket = list[blockIdx.x];
  const uint32_t lo = t.dir[bucket], hi = t.dir[bucket + 1];
  const uint32_t S = (uint32_t)bucket_slots(lo, hi);
  Slot* base = slots + 2ull * lo;
  if (tid == 0) { n_e_s = 0; n_gather_s = 0; maxstep_s = 0; unplaced_s = 0; }
  __syncthreads();
  // the probe sequence of a group, from its tag alone (the tag names the (k-1)-mer and its orientation)
  auto sequence_of = [&](uint64_t tag, uint32_t* home, uint32_t* second) {
    const uint64_t G = tag >> 1;
    const uint64_t P = (t.canonical && (tag & 1)) ? revcomp(G, t.k - 1) : G;
    const Key g = make_key(t, P);
    *home = (uint32_t)home_slot(t, g, S);
    *second = S >= 4 ? (uint32_t)second_pair(tag, S, *home) : (*home + 2 >= S ? 0u : *home + 2);
  };
  uint32_t occupied = 0;
  for (uint32_t i = tid; i < S; i += 256) occupied += base[i].tag != EMPTY;
  if (occupied) atomicAdd(&n_e_s, occupied);
  __syncthreads();
  const uint32_t n_e = n_e_s;
  uint32_t P2 = 1;
  while (P2 < n_e) P2 <<= 1;
  // LDS: the new layout (S slots), then per group: tag, old slot, home, second pair, new slot
  const uint64_t need = (uint64_t)S * sizeof(Slot) + (uint64_t)P2 * (8 + 4 * 4);
  // measure only (the bucket stays as the race built it): how far along its sequence does every group sit?
  auto measure_only = [&]() {
    uint32_t far = 0;
    for (uint32_t i = tid; i < S; i += 256) {
      const uint64_t tag = base[i].tag;
      if (tag == EMPTY) continue;
      uint32_t idx, second;
      sequence_of(tag, &idx, &second);
      for (uint32_t step = 0; step < S; ++step) {
        if (step == 2) idx = second;
        if (idx == i) { far = max(far, step); break; }
        if (++idx == S) idx = 0;
      }
    }
    if (far >= 2) atomicMax(&maxstep_s, far);
    __syncthreads();
    if (tid == 0) {
      if (maxstep_s >= 2) atomicMax(&meta[3], (unsigned long long)maxstep_s);
      atomicAdd(&meta[5], 1ull);
      if (!final && maxstep_s >= 4 && cap_gen(caps[bucket]) < CAP_HARD_GEN && !(atomicOr(&caps[bucket], CAP_GROW) & CAP_GROW))
        atomicAdd(&meta[2], 1ull);
    }
  };
  if (S > 0xFFFFu || need > lds_bytes) {
    measure_only();
    return;
  }
  Slot* A = reinterpret_cast<Slot*>(settle_lds);
  uint64_t* tags = reinterpret_cast<uint64_t*>(settle_lds + (uint64_t)S * sizeof(Slot));
  uint32_t* old_at = reinterpret_cast<uint32_t*>(tags + P2);
  uint32_t* home_at = old_at + P2;
  uint32_t* second_at = home_at + P2;
  uint32_t* new_at = second_at + P2;
  for (uint32_t i = tid; i < P2; i += 256) { tags[i] = EMPTY; old_at[i] = 0; }
  __syncthreads();                                       // (n_e_s is not touched again: every wave has the same n_e)
  for (uint32_t i = tid; i < S; i += 256) {
    const uint64_t tag = base[i].tag;
    A[i].tag = EMPTY;
    A[i].c[0] = A[i].c[1] = A[i].c[2] = A[i].c[3] = 0;
    if (tag != EMPTY) {
      const uint32_t at = atomicAdd(&n_gather_s, 1u);
      tags[at] = tag;
      old_at[at] = i;
    }
  }
  __syncthreads();
  // bitonic sort of (tag, old slot) by tag — tags are distinct, the padding (EMPTY) sorts last
  for (uint32_t kk = 2; kk <= P2; kk <<= 1)
    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < P2; i += 256) {
        const uint32_t x = i ^ j;
        if (x > i) {
          const bool up = (i & kk) == 0;
          const uint64_t a = tags[i], b = tags[x];
          if ((a > b) == up) {
            tags[i] = b; tags[x] = a;
            const uint32_t o = old_at[i]; old_at[i] = old_at[x]; old_at[x] = o;
          }
        }
      }
      __syncthreads();
    }
  for (uint32_t e = tid; e < n_e; e += 256) sequence_of(tags[e], &home_at[e], &second_at[e]);
  __syncthreads();
  if (tid == 0) {
    // home pairs first, for every group (as the race does, where nobody moves on before its pair is full — entering
    // the groups one by one to the end of their sequences lets the early ones take the home pairs of the later
    // ones: 6 probes where the race needed 4), then whoever found its pair taken, in the same order
    uint32_t far = 0;
    for (uint32_t e = 0; e < n_e; ++e) {
      const uint32_t h = home_at[e];
      new_at[e] = 0xFFFFFFFFu;
      if (A[h].tag == EMPTY) { A[h].tag = tags[e]; new_at[e] = h; }
      else if (A[h + 1].tag == EMPTY) { A[h + 1].tag = tags[e]; new_at[e] = h + 1; }
    }
    bool unplaced = false;
    for (uint32_t e = 0; e < n_e; ++e) {
      if (new_at[e] != 0xFFFFFFFFu) continue;
      uint32_t idx = second_at[e];
      for (uint32_t step = 2; step < S; ++step) {
        if (A[idx].tag == EMPTY) { A[idx].tag = tags[e]; new_at[e] = idx; far = max(far, step); break; }
        if (++idx == S) idx = 0;
      }
      // (the sequence visits S - 2 slots behind the home pair, not the whole ring: a group may find none of them
      // free where the race, in another order, placed every group)
      if (new_at[e] == 0xFFFFFFFFu) { unplaced = true; break; }
    }
    if (unplaced) {
      unplaced_s = 1;
    } else {
      if (far >= 2) atomicMax(&meta[3], (unsigned long long)far);
      // a group that fits neither of its pairs: the bucket gets its one further doubling (and the table another round)
      if (!final && far >= 4 && cap_gen(caps[bucket]) < CAP_HARD_GEN && !(atomicOr(&caps[bucket], CAP_GROW) & CAP_GROW))
        atomicAdd(&meta[2], 1ull);
    }
  }
  __syncthreads();
  if (unplaced_s) {                                      // nothing has been written back: the race's layout stands
    measure_only();
    return;
  }
  for (uint32_t e = tid; e < n_e; e += 256) {
    const Slot src = base[old_at[e]];
    Slot& dst = A[new_at[e]];
    dst.c[0] = src.c[0]; dst.c[1] = src.c[1]; dst.c[2] = src.c[2]; dst.c[3] = src.c[3];
  }
  __syncthreads();                                       // every old slot is read before the first is overwritten
  for (uint32_t i = tid; i < S; i += 256) base[i] = A[i];
}

// Jellyfish.query for a batch (km/utils/Jellyfish.py:47-53; the loop of
// common.get_cov, km/utils/common.py:73-92).
__global__ void k_query(TableView t, const uint64_t* kmers, uint64_t n, uint32_t* out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t f = 0;
    out[i] = query_one(t, kmers[i] & t.kmask, &f);
  }
}

// Jellyfish.get_child for a batch (km/utils/Jellyfish.py:55-72).
__global__ void k_children(TableView t, const uint64_t* kmers, uint64_t n, double ratio,
                           int64_t n_cutoff, int forward, uint8_t* mask, uint32_t* counts4) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t X = kmers[i] & t.kmask;
    uint32_t f = 0;
    uint4 c;
    if (forward) {
      c = forward_children(t, X, &f);
    } else {
      const uint64_t Q = X >> 2;                 // c + X[:-1]
      const int sh = 2 * (t.k - 1);
      c.x = query_one(t, Q | (0ull << sh), &f);
      c.y = query_one(t, Q | (1ull << sh), &f);
      c.z = query_one(t, Q | (2ull << sh), &f);
      c.w = query_one(t, Q | (3ull << sh), &f);
    }
    if (mask) mask[i] = (uint8_t)child_mask(c, ratio, n_cutoff);
    if (counts4) reinterpret_cast<uint4*>(counts4)[i] = c;
  }
}

}  // namespace kmd
