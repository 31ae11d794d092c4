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

// One record per thread: enter the k-mer under the group of each orientation.
// err[0] != 0 on return means the table was too small (never with our sizing).
__global__ void k_table_insert(Slot* slots, uint64_t n_slots, const uint64_t* keys,
                               const uint32_t* counts, uint64_t n, int k, int canonical,
                               OvfSlot* ovf, uint64_t n_ovf,
                               unsigned long long* n_groups, unsigned int* err) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t K = keys[i];
    const uint32_t v = counts[i];
    if (v == 0) continue;
    uint64_t R = K;
    int n_or = 1;
    if (canonical) {
      R = revcomp(K, k);
      if (R < K) continue;            // not canonical: unreachable by query(), as in the reference
      n_or = (R == K) ? 1 : 2;
    }
    if (v >= COUNT_ESCAPE) {
      // exact value to the side table (keyed by the k-mer as stored)
      bool placed = false;
      uint64_t oi = n_ovf ? slot_index(K, n_ovf) : 0;
      for (uint64_t step = 0; step < n_ovf; ++step) {
        if (atomicCAS(&ovf[oi].count, 0u, v) == 0u) { ovf[oi].kmer = K; placed = true; break; }
        if (++oi == n_ovf) oi = 0;
      }
      if (!placed) atomicExch(err, 1u);
    }
    for (int o = 0; o < n_or; ++o) {
      const uint64_t O = o ? R : K;
      Group g = group_of_prefix(O >> 2, k, canonical);
      uint32_t s = (uint32_t)(O & 3);
      if (g.flip) s = 3 - s;
      uint64_t idx = slot_index(g.tag, n_slots);
      bool done = false;
      for (uint64_t step = 0; step < n_slots; ++step) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slots[idx].tag),
                                           (unsigned long long)EMPTY, (unsigned long long)g.tag);
        if (old == EMPTY) atomicAdd(n_groups, 1ull);
        if (old == EMPTY || old == g.tag) {
          slots[idx].c[s] = (uint16_t)(v >= COUNT_ESCAPE ? COUNT_ESCAPE : v);
          done = true;
          break;
        }
        if (++idx == n_slots) idx = 0;
      }
      if (!done) atomicExch(err, 1u);
    }
  }
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
