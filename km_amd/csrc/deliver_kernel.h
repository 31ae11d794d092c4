// deliver_kernel.h — result delivery: the walk / path-search outputs of a whole batch are
// compacted ON THE DEVICE into one contiguous buffer in its final host layout, which then
// crosses PCIe with a single asynchronous copy into pinned memory (kmgpu.hip: enqueue_deliver).
//
// What the reference hands on per target (km/tools/find_mutation.py:49-58 -> MutationFinder's
// kmer / counts lists, km/utils/MutationFinder.py:122-124, and the paths of
// km/utils/Graph.py:220-240) becomes, for all targets of the batch:
//
//   region A (offsets known on the host from n_targets alone)
//     totals[OT_WORDS]                       sizes of the tail arrays + batch statistics
//     status[n] n_ref[n] probes[n] ref_max_cov[n]    per target
//     node_off[n+1] extra_off[n+1] path_off[n+1]     CSR offsets
//   tail (sub-offsets in totals[], every array 16-byte aligned)
//     node_count[n_nodes]     counts of every node, target k-mers first          (CSR node_off);
//                             with KM_DELIVER_COUNT16 two bytes each, a count >= 65535 as 0xFFFF and exactly
//                             in the escape list of region A (esc_node / esc_value, totals[OT_N_ESC] entries)
//     extra_kmer[n_extra]     packed k-mers of the walk-discovered nodes only    (CSR extra_off)
//                             — a target's own k-mers are not shipped: node i < n_ref is the
//                             k-mer at base i of the target the caller already holds
//     path_len[n_paths] path_min_cov[n_paths] run_off[n_paths+1]                 (CSR path_off)
//     run_start[n_runs] run_len[n_runs]      paths, run-length encoded            (CSR run_off)
//
// Lean delivery (KM_DELIVER_LEAN): a target whose result is the bare reference path — no
// walk-discovered node, one path 0..n_ref-1; 70 % of a typical batch — needs nothing but that
// path's min coverage and the max count of its k-mers (ref_max_cov) for its TSV row
// (km/utils/MutationFinder.py:575-648 with km/utils/PathQuant.py:144-154: rVAF nan, expression
// min(counts) of the -1 sentinels, or nan when every count is 0); its node_count rows are then
// omitted (node_off[t+1] == node_off[t]) and stay on the device.
//
// Paths of one target are sorted by their node-index sequence here (the canonical order of
// DESIGN.md §2), nodes carry no per-target slack, nothing is reorganised on the host.
//
//  k_out_scan  one thread per target: sizes -> offsets local to a 256-target block + block totals
//  k_out_pack  one wave per target: adds the totals of the blocks before its own (and writes the
//              batch totals), copies counts / extra k-mers, ranks and emits the paths
#pragma once
#include "device_common.h"
#include "graph_kernel.h"
#include "walk_kernel.h"

namespace kmd {

enum {
  OT_N_NODES = 0, OT_N_EXTRA, OT_N_PATHS, OT_N_RUNS, OT_TAIL_BYTES,
  OT_NEEDS_HOST,        // bit 0: some target needs the large tier / the path pools overflowed
                        // bit 1: the tail does not fit the delivery buffer (nothing was packed)
  OT_OFF_COUNT, OT_OFF_EXTRA, OT_OFF_PLEN, OT_OFF_PMIN, OT_OFF_RUNOFF, OT_OFF_RSTART, OT_OFF_RLEN,
  OT_PROBES, OT_FETCHES, OT_SEED_PROBES, OT_N_FLAGGED, OT_SERIAL,
  OT_N_ESC = 24,        // 16-bit counts: counts >= 65535 met so far (k_out_scan zeroes it, k_out_pack adds)
  OT_N_BIG_DEV = 25,    // targets the device's own large tier took in this run (walk_kernel.h: WalkArgs::big_ctl)
  OT_N_GRAPH_LIST = 26, // entries of k_graph's work list in this run (what the epilogue of k_dfs left + what k_graph_pure handed over)
  OT_WORDS = 32
};
constexpr uint32_t OUT_ESC_CAP = 2048;   // exact counts the escape list of one delivery holds

constexpr uint32_t OUT_SCAN_THREADS = 256;

struct OutArgs {
  uint32_t n_targets;
  uint32_t ran_graph;
  uint32_t lean;                       // omit the node counts of bare-reference targets
  uint32_t count16;                    // node counts as 16-bit values + escape list
  uint32_t count_fetches;              // the run counted table slots read (KM_RUN_COUNT_FETCHES); else reported as 0
  unsigned long long serial;           // stamped into totals[OT_SERIAL]: which run this delivery belongs to
  const uint32_t* big_ctl;             // [0] targets handed to the device's large-tier walk, [1] to its graph pass (null: tier off)
  uint32_t big_slots;
  // walk / graph results
  const uint32_t* status;
  const uint32_t* g_status;
  const uint32_t* n_nodes;
  const uint32_t* n_ref;
  const uint32_t* t_npaths;
  const uint32_t* t_pathbase;
  const uint32_t* t_nruns;
  const uint32_t* t_refmax;
  const unsigned long long* probes;
  const unsigned long long* dfs_probes;
  const unsigned long long* fetches;
  const unsigned long long* pool_overflow;   // path / run pools exhausted (graph kernels)
  const uint32_t* n_flagged;
  const uint64_t* node_base;
  const uint64_t* node_kmer;
  const uint32_t* node_cnt;
  const uint64_t* p_runbase;
  const uint32_t* p_nruns;
  const uint32_t* p_len;
  const uint32_t* p_mincov;
  const uint32_t* r_start;
  const uint32_t* r_len;
  // scratch
  unsigned long long* loc;             // [n][4] offsets (nodes, extra, paths, runs) local to the scan block
  uint32_t* cnt;                       // [n][4] the four sizes of each target
  unsigned long long* blk_tot;         // [scan blocks][8] block totals: 4 sizes, probes, fetches, seed probes, needs-host
  unsigned long long* blk_base;        // [scan blocks][4] sizes of all blocks before this one (written by the block
                                       // that finishes last); [nblk][8] behind them: the batch totals
  unsigned int* scan_ticket;           // blocks done so far (reset by the last one)
  unsigned long long* psort;           // [path pool] (source path << 32 | runs) in sorted order
  // delivery buffer
  unsigned long long* totals;
  uint32_t* o_status;
  uint32_t* o_nref;
  uint64_t* o_probes;
  uint64_t* o_node_off;
  uint64_t* o_extra_off;
  uint32_t* o_path_off;
  uint32_t* o_refmax;
  uint64_t* o_esc_node;                // [OUT_ESC_CAP]
  uint32_t* o_esc_value;               // [OUT_ESC_CAP]
  unsigned char* tail;
  uint64_t tail_cap;
};

__host__ __device__ inline uint64_t out_align(uint64_t v) { return (v + 15) & ~15ull; }

struct OutCounts { uint64_t nodes, extra, paths, runs; };

__device__ inline OutCounts out_counts_of(const OutArgs& a, uint32_t t, uint32_t* needs, uint32_t* st_out) {
  const uint32_t st = a.status[t];
  const uint32_t gs = a.ran_graph ? a.g_status[t] : T_OK;
  OutCounts c = {0, 0, 0, 0};
  if (st == T_NEEDS_BIG || (st == T_OK && gs == T_NEEDS_BIG)) *needs = 1;
  if (st == T_OK || st == T_NODE_LIMIT) {
    c.nodes = a.n_nodes[t];
    const uint32_t nr = a.n_ref[t];
    c.extra = c.nodes > nr ? c.nodes - nr : 0;
  }
  if (a.ran_graph && st == T_OK && gs == T_OK) {
    c.paths = a.t_npaths[t];
    c.runs = a.t_nruns[t];
    // lean delivery: a bare-reference target is fully described by its path record
    // (path_min_cov) and ref_max_cov; its counts stay on the device
    if (a.lean && c.paths == 1 && a.t_refmax[t] != NOT_BARE) c.nodes = 0;
  }
  *st_out = (st == T_OK && gs != T_OK) ? T_INTERNAL : st;
  return c;
}

// Per-target sizes -> offsets local to a block of OUT_SCAN_THREADS targets + the block's totals.
// One thread per target: all loads of a block are in flight together.
__global__ __launch_bounds__(OUT_SCAN_THREADS) void k_out_scan(OutArgs a) {
  __shared__ unsigned long long part[4][OUT_SCAN_THREADS / 64];
  __shared__ unsigned long long stat[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n = a.n_targets;
  const uint32_t t = blockIdx.x * OUT_SCAN_THREADS + tid;
  if (tid < 4) stat[tid] = 0;
  if (t == 0) a.totals[OT_N_ESC] = 0;                      // (k_out_pack counts into it)
  __syncthreads();
  unsigned long long s[4] = {0, 0, 0, 0};
  unsigned long long probes = 0, fetches = 0, seedp = 0;
  uint32_t needs = 0;
  if (t < n) {
    uint32_t st;
    const OutCounts c = out_counts_of(a, t, &needs, &st);
    s[0] = c.nodes; s[1] = c.extra; s[2] = c.paths; s[3] = c.runs;
    const unsigned long long sp = a.probes[t], dp = a.dfs_probes[t];
    seedp = sp; probes = sp + dp; fetches = a.count_fetches ? a.fetches[t] : 0ull;
    a.o_status[t] = st;
    a.o_nref[t] = a.n_ref[t];
    a.o_probes[t] = sp + dp;
    a.o_refmax[t] = a.ran_graph ? a.t_refmax[t] : NOT_BARE;
  }
  // exclusive scan of the four sizes over the block: within the wave, then across the waves
  unsigned long long ex[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned long long v = s[q];
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long u = __shfl_up(v, o);
      if ((int)lane >= o) v += u;
    }
    if (lane == 63) part[q][wave] = v;
    ex[q] = v - s[q];
  }
  for (int o = 32; o > 0; o >>= 1) {
    probes += __shfl_xor(probes, o); fetches += __shfl_xor(fetches, o); seedp += __shfl_xor(seedp, o);
    needs |= __shfl_xor(needs, o);
  }
  if (lane == 0) {
    atomicAdd(&stat[0], probes); atomicAdd(&stat[1], fetches); atomicAdd(&stat[2], seedp);
    if (needs) atomicOr(&stat[3], 1ull);
  }
  __syncthreads();
  unsigned long long tot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned long long before = 0, all = 0;
    for (uint32_t w = 0; w < OUT_SCAN_THREADS / 64; ++w) {
      const unsigned long long v = part[q][w];
      if (w < wave) before += v;
      all += v;
    }
    ex[q] += before;
    tot[q] = all;
  }
  if (t < n) {
    ulonglong4* L = reinterpret_cast<ulonglong4*>(a.loc) + t;
    *L = make_ulonglong4(ex[0], ex[1], ex[2], ex[3]);
    reinterpret_cast<uint4*>(a.cnt)[t] = make_uint4((uint32_t)s[0], (uint32_t)s[1], (uint32_t)s[2], (uint32_t)s[3]);
  }
  __shared__ unsigned int last_flag;
  if (tid == 0) {
    unsigned long long* B = a.blk_tot + 8ull * blockIdx.x;
    B[0] = tot[0]; B[1] = tot[1]; B[2] = tot[2]; B[3] = tot[3];
    B[4] = stat[0]; B[5] = stat[1]; B[6] = stat[2]; B[7] = stat[3];
    __threadfence();                                       // totals visible before the ticket
    last_flag = atomicAdd(a.scan_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last_flag || wave != 0) return;
  // The block that finishes last turns the block totals into exclusive prefixes (one wave, 64
  // blocks per round) so that k_out_pack finds its base with two loads instead of a reduction.
  const uint32_t nblk = gridDim.x;
  unsigned long long carry[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
    const uint32_t bq = b0 + lane;
    unsigned long long v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      v[q] = bq < nblk ? __hip_atomic_load(a.blk_tot + 8ull * bq + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned long long inc = v[q];
      if (q == 7) {                                        // needs-host flags: OR
        for (int o = 1; o < 64; o <<= 1) inc |= __shfl_xor(inc, o);
        carry[q] |= inc;
      } else {
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned long long u = __shfl_up(inc, o);
          if ((int)lane >= o) inc += u;
        }
        if (q < 4 && bq < nblk) a.blk_base[4ull * bq + q] = carry[q] + inc - v[q];
        carry[q] += __shfl(inc, 63);
      }
    }
  }
  if (lane == 0) {
    unsigned long long* G = a.blk_base + 4ull * nblk;
#pragma unroll
    for (int q = 0; q < 8; ++q) G[q] = carry[q];
    *a.scan_ticket = 0u;
  }
}

// lexicographic order of two run-length encoded index sequences
__device__ inline bool rle_less(const uint32_t* sa, const uint32_t* la, uint32_t na,
                                const uint32_t* sb, const uint32_t* lb, uint32_t nb) {
  uint32_t ia = 0, ib = 0, oa = 0, ob = 0;
  while (ia < na && ib < nb) {
    const uint32_t va = sa[ia] + oa, vb = sb[ib] + ob;
    if (va != vb) return va < vb;
    const uint32_t ra = la[ia] - oa, rb = lb[ib] - ob;
    const uint32_t step = ra < rb ? ra : rb;
    oa += step; ob += step;
    if (oa == la[ia]) { ++ia; oa = 0; }
    if (ob == lb[ib]) { ++ib; ob = 0; }
  }
  return ia == na && ib < nb;
}

__global__ __launch_bounds__(64) void k_out_pack(OutArgs a) {
  const uint32_t t = blockIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = a.n_targets;
  // ---- global offsets: sizes of the scan blocks before this target's + the batch totals
  // (prefix sums by the last block of k_out_scan)
  const uint32_t nblk = (n + OUT_SCAN_THREADS - 1) / OUT_SCAN_THREADS, myblk = t / OUT_SCAN_THREADS;
  // every word that depends on the target alone is requested here, with the block bases (they used to follow
  // one another behind the offsets: five dependent round trips per wave, most of a wave's 3.5 us)
  const ulonglong4 lc = reinterpret_cast<const ulonglong4*>(a.loc)[t];
  const uint4 ct = reinterpret_cast<const uint4*>(a.cnt)[t];
  const uint64_t nb = a.node_base[t];
  const uint32_t pb = a.t_pathbase[t];
  const unsigned long long pool_ovf = a.ran_graph ? *a.pool_overflow : 0ull;
  unsigned long long pre[4], tot[8];
  {
    const ulonglong4 bb = *reinterpret_cast<const ulonglong4*>(a.blk_base + 4ull * myblk);
    const ulonglong4* G = reinterpret_cast<const ulonglong4*>(a.blk_base + 4ull * nblk);
    const ulonglong4 g0 = G[0], g1 = G[1];
    pre[0] = bb.x; pre[1] = bb.y; pre[2] = bb.z; pre[3] = bb.w;
    tot[0] = g0.x; tot[1] = g0.y; tot[2] = g0.z; tot[3] = g0.w;
    tot[4] = g1.x; tot[5] = g1.y; tot[6] = g1.z; tot[7] = g1.w;
  }
  uint64_t o = 0;
  const uint64_t off_count = o;  o = out_align(o + (a.count16 ? 2 : 4) * tot[0]);
  const uint64_t off_extra = o;  o = out_align(o + 8 * tot[1]);
  const uint64_t off_plen = o;   o = out_align(o + 4 * tot[2]);
  const uint64_t off_pmin = o;   o = out_align(o + 4 * tot[2]);
  const uint64_t off_runoff = o; o = out_align(o + 8 * (tot[2] + 1));
  const uint64_t off_rstart = o; o = out_align(o + 4 * tot[3]);
  const uint64_t off_rlen = o;   o = out_align(o + 4 * tot[3]);
  unsigned long long nh = tot[7];
  if (pool_ovf) nh |= 1ull;
  if (tot[2] >= (1ull << 32)) nh |= 1ull;            // path_off is 32-bit: the host splits such batches
  if (o > a.tail_cap) nh |= 2ull;
  if (t == 0 && lane == 0) {
    unsigned long long* T = a.totals;
    T[OT_N_NODES] = tot[0]; T[OT_N_EXTRA] = tot[1]; T[OT_N_PATHS] = tot[2]; T[OT_N_RUNS] = tot[3];
    T[OT_TAIL_BYTES] = o; T[OT_NEEDS_HOST] = nh;
    T[OT_OFF_COUNT] = off_count; T[OT_OFF_EXTRA] = off_extra; T[OT_OFF_PLEN] = off_plen; T[OT_OFF_PMIN] = off_pmin;
    T[OT_OFF_RUNOFF] = off_runoff; T[OT_OFF_RSTART] = off_rstart; T[OT_OFF_RLEN] = off_rlen;
    T[OT_PROBES] = tot[4]; T[OT_FETCHES] = tot[5]; T[OT_SEED_PROBES] = tot[6];
    T[OT_N_FLAGGED] = *a.n_flagged;
    T[OT_SERIAL] = a.serial;
    for (int q = OT_SERIAL + 1; q < OT_WORDS; ++q) if (q != OT_N_ESC) T[q] = 0;
    T[OT_N_GRAPH_LIST] = (unsigned long long)a.n_flagged[1] + a.n_flagged[2];
    if (a.big_ctl) T[OT_N_BIG_DEV] = min(a.big_ctl[0], a.big_slots);      // (walks; a target only its graph pass took is not counted, as on the host's path)
    a.o_node_off[n] = tot[0]; a.o_extra_off[n] = tot[1]; a.o_path_off[n] = (uint32_t)tot[2];
  }
  if (nh) return;                        // the host finishes the batch and delivers again
  const uint64_t n0 = pre[0] + lc.x, e0 = pre[1] + lc.y;
  const uint32_t p0 = (uint32_t)(pre[2] + lc.z);
  uint64_t cur = pre[3] + lc.w;          // first output run of this target
  const uint32_t nn = ct.x, ne = ct.y, np = ct.z;
  if (lane == 0) { a.o_node_off[t] = n0; a.o_extra_off[t] = e0; a.o_path_off[t] = p0; }
  uint32_t* o_cnt = reinterpret_cast<uint32_t*>(a.tail + off_count);
  uint64_t* o_ext = reinterpret_cast<uint64_t*>(a.tail + off_extra);
  uint32_t* o_plen = reinterpret_cast<uint32_t*>(a.tail + off_plen);
  uint32_t* o_pmin = reinterpret_cast<uint32_t*>(a.tail + off_pmin);
  uint64_t* o_roff = reinterpret_cast<uint64_t*>(a.tail + off_runoff);
  uint32_t* o_rs = reinterpret_cast<uint32_t*>(a.tail + off_rstart);
  uint32_t* o_rl = reinterpret_cast<uint32_t*>(a.tail + off_rlen);
  // ---- nodes (four independent loads per lane in flight)
  if (a.count16) {
    const uint32_t* src = a.node_cnt + nb;
    uint16_t* dst = reinterpret_cast<uint16_t*>(a.tail + off_count) + n0;
    for (uint32_t i0 = lane; i0 < nn; i0 += 256) {
      uint32_t v[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) v[u] = (i0 + 64 * u < nn) ? src[i0 + 64 * u] : 0u;
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        if (i0 + 64 * u >= nn) continue;
        if (v[u] >= 0xFFFFu) {                              // the exact value goes on the escape list
          const unsigned long long at = atomicAdd(&a.totals[OT_N_ESC], 1ull);
          if (at < OUT_ESC_CAP) { a.o_esc_node[at] = n0 + i0 + 64 * u; a.o_esc_value[at] = v[u]; }
          v[u] = 0xFFFFu;
        }
        dst[i0 + 64 * u] = (uint16_t)v[u];
      }
    }
  } else {
    const uint32_t* src = a.node_cnt + nb;
    uint32_t* dst = o_cnt + n0;
    for (uint32_t i0 = lane; i0 < nn; i0 += 256) {
      uint32_t v[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) v[u] = (i0 + 64 * u < nn) ? src[i0 + 64 * u] : 0u;
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) if (i0 + 64 * u < nn) dst[i0 + 64 * u] = v[u];
    }
  }
  if (ne) {
    const uint64_t* src = a.node_kmer + nb + (nn - ne);
    uint64_t* dst = o_ext + e0;
    for (uint32_t i = lane; i < ne; i += 64) dst[i] = src[i];
  }
  if (t + 1 == n && lane == 0) o_roff[tot[2]] = tot[3];
  // ---- paths, sorted by index sequence
  if (np == 0) return;
  if (np == 1) {
    const uint32_t nr = a.p_nruns[pb];
    const uint64_t rb = a.p_runbase[pb];
    if (lane == 0) { o_plen[p0] = a.p_len[pb]; o_pmin[p0] = a.p_mincov[pb]; o_roff[p0] = cur; }
    for (uint32_t q = lane; q < nr; q += 64) { o_rs[cur + q] = a.r_start[rb + q]; o_rl[cur + q] = a.r_len[rb + q]; }
    return;
  }
  for (uint32_t i = lane; i < np; i += 64) {
    const uint32_t ni = a.p_nruns[pb + i];
    const uint32_t* si = a.r_start + a.p_runbase[pb + i];
    const uint32_t* li = a.r_len + a.p_runbase[pb + i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < np; ++j) {
      if (j == i) continue;
      const uint32_t nj = a.p_nruns[pb + j];
      const uint32_t* sj = a.r_start + a.p_runbase[pb + j];
      const uint32_t* lj = a.r_len + a.p_runbase[pb + j];
      if (rle_less(sj, lj, nj, si, li, ni) || (j < i && !rle_less(si, li, ni, sj, lj, nj))) ++rank;
    }
    a.psort[pb + rank] = ((unsigned long long)i << 32) | ni;
    o_plen[p0 + rank] = a.p_len[pb + i];
    o_pmin[p0 + rank] = a.p_mincov[pb + i];
  }
  __syncthreads();                       // psort written by other lanes (single-wave workgroup)
  for (uint32_t r = 0; r < np; ++r) {
    const unsigned long long e = a.psort[pb + r];
    const uint32_t i = (uint32_t)(e >> 32), nr = (uint32_t)e;
    const uint64_t rb = a.p_runbase[pb + i];
    if (lane == 0) o_roff[p0 + r] = cur;
    for (uint32_t q = lane; q < nr; q += 64) { o_rs[cur + q] = a.r_start[rb + q]; o_rl[cur + q] = a.r_len[rb + q]; }
    cur += nr;
  }
}

}  // namespace kmd
