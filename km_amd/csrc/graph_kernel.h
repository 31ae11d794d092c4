// graph_kernel.h — MutationFinder.graph_analysis + Graph for a batch of targets.
//
// Reference: km/utils/MutationFinder.py:496-572 (overlap edges weight 1, reference
// and cap edges weight 0.01), km/utils/Graph.py:63-119 (dense float32 Dijkstra,
// first-index argmin, strict '<' relaxation), :121-198 (two runs, strip reference
// edges), :200-240 (one source->sink path per remaining edge, unique set).
//
// One 256-thread workgroup per target (all four waves share the parallel phases, wave 0
// runs the short sequential ones); node numbering as written by the walk kernel,
// BigBang (source) = m, BigCrunch (sink) = m + 1 with m = n_nodes.
//
// How the dense O(n^2) algorithm is reproduced exactly on a sparse graph:
//  * edges   One LDS table keyed by the (k-1)-mer PREFIX of every node holds the
//    node index per last base; the suffix of node j looked up there yields its
//    <=4 successors in one probe (the reference's prefix_dct), predecessors are
//    the transposed scatter (a predecessor is identified by its first base).
//  * dist[]  With strictly positive weights and monotone float32 addition the
//    distances Dijkstra ends with are the unique solution of
//    dist[j] = min_i fl(dist[i] + w_ij); any label-setting order that relaxes
//    every edge with the final distance of its tail yields them, with the same
//    float32 additions hop by hop.  These graphs are long index-contiguous
//    chains (node j -> j+1 being j's only out-edge and j+1's only in-edge: the
//    "link" bitmap), so a frontier Dijkstra only runs over chain heads; a whole
//    chain is finalised by a register loop of dependent float adds with one LDS
//    write per 64 nodes.
//  * prev[]  Graph.py visits nodes in (dist, index) order and overwrites prev[j]
//    only on a strict improvement, hence prev[j] is the in-neighbour i minimising
//    (fl(dist[i] + w_ij), dist[i], i) lexicographically — a purely local rule
//    evaluated for all nodes in parallel once dist[] is known.
//  * unique paths  The path through edge (a,b) is B(a).A(b) (source-tree chain to
//    a, sink-tree chain from b).  Two edges give the same path iff both lie on it
//    and every edge between them is a tree edge of both trees; an edge is kept as
//    the representative iff no other candidate edge precedes it on its own path
//    under that condition — decided by walking back from the edge (O(1) in the
//    usual case), no path materialisation or hashing needed.
//
// Paths leave the kernel run-length encoded (consecutive node indices collapse
// to (start, len)); emission hops from chain to chain through the link bitmap.
// Pool space is reserved with one atomicAdd per target.
#pragma once
#include <type_traits>

#include "device_common.h"
#include "walk_kernel.h"

namespace kmd {

constexpr uint32_t GRAPH_THREADS = 256;

struct GraphArgs {
  int k;
  uint64_t kmask;
  uint64_t pmask;
  const uint32_t* tids;      // BIG: targets to run; nullptr = blockIdx.x
  uint32_t n_targets;
  const uint64_t* node_kmer;   // walk-discovered nodes only (index >= n_ref); the target's own
                               // k-mers are read from its packed words
  const uint32_t* node_cnt;
  const uint64_t* node_base;
  const uint64_t* packed;      // 2-bit packed targets (k_pack)
  const uint64_t* woff;        // word offsets into `packed`
  uint32_t words_cap;          // packed words staged per target (>= (max_len + 31) / 32 + 1)
  const uint32_t* n_nodes;
  const uint32_t* n_ref;
  uint32_t* status;          // T_REPEAT is raised here (duplicate k-mer in the target)
  const uint32_t* tflag;     // target has flagged seeds (its node list comes from k_dfs)
  uint32_t* need_full;       // k_graph_pure -> k_graph: target needs the general algorithm
  uint32_t use_need_full;    // k_graph: skip targets k_graph_pure already answered
  uint32_t hcap_pure;        // k_graph_pure: fingerprint slots (multiple of 64)
  // Work list of k_graph (LDS tier): the flagged targets (k_seed's list, work_n[0] of them) followed
  // by the unflagged ones k_graph_pure could not answer (appended there, work_n[1]; few).  Block b
  // of k_graph takes entry b.  Null when every target goes to k_graph anyway (duplicate check
  // only, ablations): block b then takes target b.
  uint32_t* work_list;
  uint32_t* work_n;
  // When the epilogue of k_dfs answers the regular flagged targets itself (walk_kernel.h), the first
  // part of the list is `left` (work_n[2] entries: what it could not answer) instead of all flagged ones.
  const uint32_t* left;
  uint32_t dfs_answers;      // 1: this run's k_dfs had its epilogue on
  // large tier on the device (walk_kernel.h: WalkArgs::big_ctl): the LDS tier appends what it cannot hold to
  // big_graph (count in big_ctl[1]); the large-tier launch that follows in the same stream takes tids = big_graph
  // and the count from tids_n
  uint32_t* big_ctl;
  uint32_t* big_graph;
  uint32_t big_slots;
  const uint32_t* tids_n;
  // outputs
  uint32_t* g_status;        // per target: T_OK / T_NEEDS_BIG / T_INTERNAL
  uint32_t* t_npaths;        // per target
  uint32_t* t_pathbase;      // per target: first path record
  uint32_t* t_nruns;         // per target: run records over all its paths
  uint32_t* t_refmax;        // per target: max count over its own k-mers when the result is the bare
                             // reference path (decided by the pure-chain tests), else NOT_BARE
  uint32_t* t_eremoved;      // per target: reference edges stripped (Graph.py:184-198) / edges left in the non-reference
  uint32_t* t_enonref;       // edge set (Graph.py:231): the two numbers the reference logs with -v, in our node order
  // Pools are split into POOL_GROUPS equal regions (group = target & 63) so that the
  // bump-allocation atomics of different targets rarely share an address.
  // counters[g*16+0] paths used in group g, [g*16+1] runs used, counters[64*16] overflow flag
  unsigned long long* counters;
  uint64_t path_pool, run_pool;  // total capacities (multiples of POOL_GROUPS)
  uint32_t* p_target;
  uint64_t* p_runbase;
  uint32_t* p_nruns;
  uint32_t* p_len;
  uint32_t* p_mincov;
  uint32_t* r_start;
  uint32_t* r_len;
  // geometry
  uint32_t ncap;   // max nodes incl. caps
  uint32_t hcap;   // prefix-table slots, multiple of 64, >= 1.5 * ncap
  const float* tref;  // tref[j] = fl(...fl(0.01f + 0.01f)... ) (j+1 terms): reference-chain distances
  uint32_t tref_len;
  uint32_t dbg;    // diagnostic (KM_DEBUG_FLAGS >> 8): stop after step N (timing ablation only)
  unsigned char* g_ws;
  uint64_t g_stride;
};

template <typename idx_t>
__host__ __device__ inline uint64_t graph_region_a(uint32_t ncap, uint32_t hcap) {
  const uint64_t tab = (uint64_t)hcap * (8 + 4 * sizeof(idx_t));   // prefix keys + 4 indices
  const uint64_t run = (uint64_t)ncap * (8 + 6 * sizeof(idx_t));   // dist_f/b, before, after, frontier x2, cand(2)
  return ((tab > run ? tab : run) + 15) & ~15ull;
}

template <typename idx_t>
__host__ __device__ inline uint64_t graph_ws_bytes(uint32_t ncap, uint32_t hcap, uint32_t words_cap) {
  uint64_t b = graph_region_a<idx_t>(ncap, hcap);

  b += (((uint64_t)ncap * 8 * sizeof(idx_t)) + 15) & ~15ull;  // succ, pred
  b += (uint64_t)((ncap + 31) / 32) * 4 * 3;                  // link, inq (forward / backward) bits
  b += (uint64_t)((4 * (uint64_t)ncap + 2 + 31) / 32) * 4;    // removed bits
  b += 32;                                                    // scalars
  b += 8 + (uint64_t)words_cap * 8;                           // packed target (8-byte aligned)
  return (b + 15) & ~15ull;
}


// LDS of k_graph_pure per target: position table (16-bit, hcap slots), its losers' table (32-bit, hcap / 8 slots), the
// packed target
__host__ __device__ inline uint64_t pure_lds_bytes(uint32_t hcap, uint32_t words_cap) {
  return (uint64_t)hcap * 2 + (uint64_t)(hcap / 8) * 4 + (uint64_t)words_cap * 8;
}

// ---------------------------------------------------------------------------- k_graph_pure
// Unflagged targets (node list == the target's own k-mers, final right after k_seed):
// decide whether the graph is the bare reference chain — all (k-1)-mer prefixes distinct
// and the last k-mer's suffix not among them — and if so emit its single path.  Small LDS
// (prefix keys only), so it runs at full occupancy and overlaps k_dfs on a second stream.
// Everything else is left to k_graph through need_full[].
template <int K>
__global__ __launch_bounds__(64) void k_graph_pure(GraphArgs a) {
  if constexpr (K != 0) {               // instantiated for one k: shifts and masks fold
    a.k = K;
    a.kmask = K >= 32 ? ~0ull : ((1ull << (2 * K)) - 1);
    a.pmask = (1ull << (2 * (K - 1))) - 1;
  }
  extern __shared__ __align__(16) unsigned char smem[];
  // one wave per target: this pass runs beside k_dfs, it should take few wave slots
  const uint32_t lane = threadIdx.x & 63u, NT = 64;
  const uint32_t tid = lane;
  const uint32_t t = blockIdx.x;
  const uint32_t h_status = a.status[t], h_tflag = a.tflag[t];   // header words requested together
  const uint32_t n_ref = a.n_ref[t];
  const uint64_t nb = a.node_base[t];
  const uint64_t h_woff = a.woff[t];
  // (a flagged target's t_refmax belongs to k_dfs when its epilogue is on — every exit of its fast tier writes it —
  // and to k_graph, which resets it before it decides anything)
  if (tid == 0 && !(a.dfs_answers && h_tflag)) a.t_refmax[t] = NOT_BARE;
  if (h_status != T_OK) {
    if (tid == 0) { a.need_full[t] = 0; a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; }
    return;
  }
  if (h_tflag) return;                         // k_graph handles it once k_dfs is done
  const uint32_t hcap = a.hcap_pure;                        // slots of the position table: a power of two >= 4 (n_ref + 1)
  if ((uint64_t)4 * (n_ref + 2) > (uint64_t)hcap || n_ref >= 0xFFFFu || (a.dbg != 0 && !(a.dbg & 0x80u))) {
    if (tid == 0) { a.need_full[t] = 1; if (a.work_list) a.work_list[a.work_n[0] + atomicAdd(&a.work_n[1], 1u)] = t; }
    return;
  }
  const uint32_t* ncnt = a.node_cnt + nb;
  // the packed target: n_ref + k - 1 bases, (L + 31) / 32 + 1 words (the last one zero)
  const uint32_t nwords = (n_ref + (uint32_t)a.k - 1 + 31) >> 5;
  uint16_t* pos = reinterpret_cast<uint16_t*>(smem);                              // [hcap]
  uint32_t* tab2 = reinterpret_cast<uint32_t*>(smem + (uint64_t)hcap * 2);        // [hcap / 8]: the losers of `pos`
  uint64_t* words = reinterpret_cast<uint64_t*>(smem + pure_lds_bytes(hcap, 0));
  if (nwords + 1 > a.words_cap) {
    if (tid == 0) { a.need_full[t] = 1; if (a.work_list) a.work_list[a.work_n[0] + atomicAdd(&a.work_n[1], 1u)] = t; }
    return;
  }
  {
    const uint64_t* src = a.packed + h_woff;
    for (uint32_t w = tid; w <= nwords; w += NT) words[w] = src[w];
  }
  // Are the n_ref + 1 (k-1)-mers of the target (the prefix of every k-mer and the suffix of the last) distinct?
  // As k_dfs does for its node set (walk_kernel.h, "position table"): every (k-1)-mer stores its INDEX at the slot
  // its low 32 bits hash to, with a plain store (the last one wins); whoever does not read its own index back lost
  // the slot — to an equal (k-1)-mer (the answer is no) or to another one — and the losers (a tenth at load 1/4) go
  // into a small exact table with compare-and-swap, where an equal one is met for sure: equal (k-1)-mers share
  // both slots.  Round 3 entered all 471 as 32-bit fingerprints with compare-and-swap.
  {
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), zero = make_uint4(0u, 0u, 0u, 0u);
    uint4* q = reinterpret_cast<uint4*>(smem);
    for (uint32_t x = tid; x < hcap / 8; x += NT) q[x] = ones;                    // 2 hcap bytes
    uint4* q2 = reinterpret_cast<uint4*>(tab2);
    for (uint32_t x = tid; x < hcap / 32; x += NT) q2[x] = zero;                  // hcap / 2 bytes
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // one wave: LDS runs its operations in order
  const uint32_t sh1 = 32u - (uint32_t)__ffs((int)hcap) + 1u, m2 = hcap / 8 - 1;
  auto part_of = [&](uint32_t j) -> uint64_t {           // (k-1)-mer j: prefix of k-mer j, suffix of the last one for j = n_ref
    return j < n_ref ? (kmer_from_words(words, j, a.k) >> 2) : (kmer_from_words(words, n_ref - 1, a.k) & a.pmask);
  };
  auto h1 = [&](uint64_t key) -> uint32_t { return ((uint32_t)key * 0x9E3779B1u) >> sh1; };
  uint32_t not_pure = 0;
  uint32_t mincov = 0xFFFFFFFFu, maxcov = 0;
  for (uint32_t j = tid; j <= n_ref; j += NT) pos[h1(part_of(j))] = (uint16_t)j;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (uint32_t j0 = 0; j0 <= n_ref; j0 += NT) {         // wave-uniform trip count
    const uint32_t j = j0 + tid;
    if (j <= n_ref) {
      const uint32_t cc = j < n_ref ? ncnt[j] : 0xFFFFFFFFu;
      const uint64_t key = part_of(j);
      const uint32_t w = pos[h1(key)];
      if (w != j) {
        if (part_of(w) == key) not_pure = 1;             // (w is an index some (k-1)-mer stored)
        else {
          uint32_t s2 = (((uint32_t)key * 0x85EBCA6Bu) ^ ((uint32_t)(key >> 32) * 0xC2B2AE35u)) >> 7 & m2;
          for (uint32_t step = 0; step <= m2; ++step) {
            const uint32_t old = atomicCAS(&tab2[s2], 0u, j + 1);
            if (old == 0u) break;
            if (part_of(old - 1) == key) { not_pure = 1; break; }
            s2 = (s2 + 1) & m2;
            if (step == m2) not_pure = 1;                // full (cannot happen at this load): k_graph decides
          }
        }
      }
      mincov = cc < mincov ? cc : mincov;
      if (j < n_ref) maxcov = cc > maxcov ? cc : maxcov;
    }
  }
  if (__any((int)not_pure)) {
    if (tid == 0) { a.need_full[t] = 1; if (a.work_list) a.work_list[a.work_n[0] + atomicAdd(&a.work_n[1], 1u)] = t; }
    return;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = __shfl_xor(mincov, o); mincov = other < mincov ? other : mincov;
    const uint32_t om = __shfl_xor(maxcov, o); maxcov = om > maxcov ? om : maxcov;
  }
  if (lane == 0) {
    const uint32_t pg = t % POOL_GROUPS;
    const uint64_t pg_paths = a.path_pool / POOL_GROUPS, pg_runs = a.run_pool / POOL_GROUPS;
    unsigned long long* ctr = a.counters + (uint64_t)pg * POOL_CTR_STRIDE;
    const unsigned long long pl = atomicAdd(&ctr[0], 1ull);
    const unsigned long long rl = atomicAdd(&ctr[1], 1ull);
    if (pl + 1 > pg_paths || rl + 1 > pg_runs) {
      atomicExch(a.counters + (uint64_t)POOL_GROUPS * POOL_CTR_STRIDE, 1ull);
      a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0;
    } else {
      const uint64_t pi = (uint64_t)pg * pg_paths + pl, ri = (uint64_t)pg * pg_runs + rl;
      a.r_start[ri] = 0; a.r_len[ri] = n_ref;
      a.p_target[pi] = t; a.p_runbase[pi] = ri; a.p_nruns[pi] = 1; a.p_len[pi] = n_ref; a.p_mincov[pi] = mincov;
      a.t_npaths[t] = 1; a.t_pathbase[t] = (uint32_t)pi; a.t_nruns[t] = 1;
      a.t_refmax[t] = maxcov;
    }
    a.t_eremoved[t] = n_ref >= 2 ? n_ref - 1 : 0;           // every chain edge but (0, 1); (source, 0) and it stay
    a.t_enonref[t] = 2;
    a.need_full[t] = 0;
    a.g_status[t] = T_OK;
  }
}

// Direct mode (large tier, a.tids given): block b works on target tids[b].  LDS tier: block b
// works on entry b of the work list (GraphArgs::work_list) — the blocks that have something to do
// are the first ones of the grid and are dispatched together; the others leave after one load
// instead of being scattered among them, each holding 26 KB of LDS while it reads its headers.
template <bool BIG, int K>
__global__ __launch_bounds__(GRAPH_THREADS) void k_graph(GraphArgs a0) {
  uint32_t t_;
  if (BIG || a0.tids || !a0.work_list) {
    if (a0.tids_n) {                                  // the device's own list: as many entries as were appended (at most big_slots)
      const uint32_t n_l = min(*a0.tids_n, a0.big_slots);
      if (blockIdx.x >= n_l) return;
    }
    t_ = a0.tids ? a0.tids[blockIdx.x] : blockIdx.x;
  } else {
    const uint32_t n_first = a0.dfs_answers ? a0.work_n[2] : a0.work_n[0];
    const uint32_t n_total = n_first + a0.work_n[1];
    auto entry = [&](uint32_t e) -> uint32_t {
      return e < n_first ? (a0.dfs_answers ? a0.left[e] : a0.work_list[e]) : a0.work_list[a0.work_n[0] + (e - n_first)];
    };
    // The grid may be smaller than the batch (kmgpu.hip: launch_graph — when the epilogue of k_dfs answers the
    // regular targets the list is a percent of the batch, and one block per TARGET meant 40 000 waves a step
    // launched to read two words): entries beyond the grid go to the large tier, as any target does that
    // outgrows this one — block 0 says so, the host reruns them (km_batch_sync).
    if (blockIdx.x == 0 && n_total > gridDim.x) {
      for (uint32_t e = gridDim.x + threadIdx.x; e < n_total; e += GRAPH_THREADS) {
        const uint32_t tt = entry(e);
        if (a0.status[tt] == T_OK) {
          a0.g_status[tt] = T_NEEDS_BIG; a0.t_npaths[tt] = 0; a0.t_pathbase[tt] = 0; a0.t_nruns[tt] = 0;
          if (a0.big_ctl) { const uint32_t at = atomicAdd(&a0.big_ctl[1], 1u); if (at < a0.big_slots) a0.big_graph[at] = tt; }
        }
      }
    }
    if (blockIdx.x >= n_total) return;
    t_ = entry(blockIdx.x);
  }
  const uint32_t t = t_;
  GraphArgs a = a0;
  // (LDS tier) this target goes to the large tier: the device's own, if its list has room
  auto hand_to_big = [&]() {
    if constexpr (!BIG) {
      if (a0.big_ctl) { const uint32_t at = atomicAdd(&a0.big_ctl[1], 1u); if (at < a0.big_slots) a0.big_graph[at] = t; }
    }
  };
  if constexpr (K != 0) {               // instantiated for one k: shifts and masks fold
    a.k = K;
    a.kmask = K >= 32 ? ~0ull : ((1ull << (2 * K)) - 1);
    a.pmask = (1ull << (2 * (K - 1))) - 1;
  }
  const bool keep_pure = (a.dbg & 0x80u) != 0;   // ablation: pure pass stays active
  a.dbg &= 0x7Fu;
  using idx_t = typename std::conditional<BIG, uint32_t, uint16_t>::type;
  constexpr idx_t NONE = (idx_t)~(idx_t)0;
  constexpr uint32_t NIL = 0xFFFFFFFFu;
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t tid = threadIdx.x, NT = GRAPH_THREADS;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  // wave-local ordering for the sections only wave 0 executes
  auto wsync = [&]() {
    if constexpr (BIG) __threadfence_block();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // bump allocation in this target's pool group
  const uint32_t pg = t % POOL_GROUPS;
  const uint64_t pg_paths = a.path_pool / POOL_GROUPS, pg_runs = a.run_pool / POOL_GROUPS;
  unsigned long long* ctr = a.counters + (uint64_t)pg * POOL_CTR_STRIDE;
  unsigned long long* ovf = a.counters + (uint64_t)POOL_GROUPS * POOL_CTR_STRIDE;

  // the per-target header words, requested together (the early exits below would otherwise
  // chain them into four dependent round trips)
  const uint32_t h_status = a.status[t], h_tflag = a.tflag[t], h_need = a.need_full[t];
  const uint32_t m = a.n_nodes[t];
  const uint32_t n_ref = a.n_ref[t];
  const uint64_t nb = a.node_base[t];
  const uint64_t h_woff = a.woff[t];
  if (h_status != T_OK) {
    if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; }
    return;
  }
  if (a.use_need_full && !h_tflag && !h_need) return;   // answered by k_graph_pure
  // whatever an earlier kernel or an earlier batch left there: only the bare-chain exit (1b, the same thread)
  // names a maximum
  if (tid == 0) a.t_refmax[t] = NOT_BARE;
  const uint32_t n = m + 2, src = m, snk = m + 1;
  const uint32_t ncap = a.ncap, hcap = a.hcap;
  if (n > ncap || (uint64_t)3 * n > (uint64_t)2 * hcap || (!BIG && n >= 0xFFFFu)) {
    if (tid == 0) { a.g_status[t] = (BIG && !a0.tids_n) ? T_INTERNAL : T_NEEDS_BIG; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; hand_to_big(); }
    return;
  }
  const uint64_t* nkx = a.node_kmer + nb;        // valid for indices >= n_ref only
  const uint32_t* ncnt = a.node_cnt + nb;
  const int k = a.k;
  const uint32_t nwords = (n_ref + (uint32_t)k - 1 + 31) >> 5;
  if (nwords + 1 > a.words_cap) {
    if (tid == 0) { a.g_status[t] = (BIG && !a0.tids_n) ? T_INTERNAL : T_NEEDS_BIG; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; hand_to_big(); }
    return;
  }

  if (a.dbg == 8) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  unsigned char* wsb;
  if constexpr (BIG) wsb = a.g_ws + (uint64_t)blockIdx.x * a.g_stride;
  else wsb = smem;
  // region A, first life: prefix table
  uint64_t* pkeys = reinterpret_cast<uint64_t*>(wsb);
  idx_t* pidx = reinterpret_cast<idx_t*>(pkeys + hcap);
  // region A, second life
  float* dist_f = reinterpret_cast<float*>(wsb);
  float* dist_b = dist_f + ncap;
  idx_t* before = reinterpret_cast<idx_t*>(dist_b + ncap);
  idx_t* after = before + ncap;
  idx_t* frontier = after + ncap;                  // forward pass
  idx_t* frontier_b = frontier + ncap;             // backward pass
  idx_t* cand = frontier_b + ncap;                 // pairs (a, b)
  const uint32_t ccap = ncap;
  // persistent
  unsigned char* pp = wsb + graph_region_a<idx_t>(ncap, hcap);
  idx_t* succ = reinterpret_cast<idx_t*>(pp);
  idx_t* pred = succ + (uint64_t)4 * ncap;
  pp += (((uint64_t)ncap * 8 * sizeof(idx_t)) + 15) & ~15ull;
  const uint32_t nbw = (ncap + 31) / 32;
  uint32_t* link = reinterpret_cast<uint32_t*>(pp);
  uint32_t* inq = link + nbw;
  uint32_t* inq_b = inq + nbw;
  uint32_t* removed = inq_b + nbw;
  const uint32_t n_removed_words = (uint32_t)((4 * (uint64_t)ncap + 2 + 31) / 32);
  uint32_t* scal = removed + n_removed_words;      // [0] candidate count, [1] pool flag
  // packed target, 8-byte aligned behind the scalars
  uint64_t* words = reinterpret_cast<uint64_t*>(
      wsb + (((uint64_t)(reinterpret_cast<unsigned char*>(scal + 8) - wsb) + 7) & ~7ull));

  const float INF = __int_as_float(0x7F800000);
  const float W_REF = 0.01f, W_ALT = 1.0f;
  auto weight = [&](uint32_t u, uint32_t v) -> float {
    if (u == src || v == snk) return W_REF;                       // cap edges
    return (u + 1 == v && v < n_ref) ? W_REF : W_ALT;             // reference edge i -> i+1
  };

  // ---- 1. prefix table: (k-1)-mer prefix -> node index per last base -----------------
  // EMPTY keys and NONE indices are all-ones bytes.  LDS tier: keys, node indices and succ / pred
  // are filled in one go and a thread remembers the table slot of each of its (at most MAXOWN) nodes,
  // so the index per (prefix, last base) is written at insertion and the duplicate check re-reads
  // it without probing again.  Global tier: only the key array first (the common bare-chain exit
  // needs no more), the rest in 1c.
  constexpr bool FUSED = !BIG;
  constexpr uint32_t MAXOWN = 8;               // nodes per thread: ncap < 2048 in the LDS tier
  const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  const uint64_t fill_bytes = graph_region_a<idx_t>(ncap, hcap) + ((((uint64_t)ncap * 8 * sizeof(idx_t)) + 15) & ~15ull);
  {
    uint4* q = reinterpret_cast<uint4*>(wsb);
    const uint64_t n16 = FUSED ? fill_bytes / 16 : (uint64_t)hcap / 2;       // hcap keys, 8 B each
    for (uint64_t x = tid; x < n16; x += NT) q[x] = ones;
  }
  {
    const uint64_t* srcw = a.packed + h_woff;
    for (uint32_t w = tid; w <= nwords; w += NT) words[w] = srcw[w];
  }
  // node j's k-mer: the target's own k-mers come from its packed words
  auto nk = [&](uint32_t j) -> uint64_t { return j < n_ref ? kmer_from_words(words, j, k) : nkx[j]; };
  for (uint32_t w = tid; w < nbw; w += NT) { link[w] = 0; inq[w] = 0; inq_b[w] = 0; }
  for (uint32_t w = tid; w < n_removed_words; w += NT) removed[w] = 0;
  if (tid < 8) scal[tid] = 0;
  __syncthreads();
  if (a.dbg == 9) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  uint32_t shared_prefix = 0;            // some other node has the same (k-1)-mer prefix
  uint32_t own_slot[MAXOWN];             // FUSED: 4 * slot + last base of this thread's nodes
  uint64_t own_x[MAXOWN];                // FUSED: their k-mers (2c' looks up the suffixes of the walk's nodes)
  if constexpr (FUSED) {
    if (m > MAXOWN * NT) {                 // cannot happen with the LDS tier's geometry
      if (tid == 0) { a.g_status[t] = T_NEEDS_BIG; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; hand_to_big(); }
      return;
    }
#pragma unroll
    for (uint32_t q = 0; q < MAXOWN; ++q) {
      const uint32_t j = tid + q * NT;
      own_slot[q] = 0;
      own_x[q] = 0;
      if (j < m) {
        bool wn;
        const uint64_t X = nk(j);
        own_x[q] = X;
        const int s = set_insert_lane(pkeys, hcap, X >> 2, &wn);
        if (s < 0 || !wn) shared_prefix = 1;
        own_slot[q] = 4u * (uint32_t)(s < 0 ? 0 : s) + (uint32_t)(X & 3);
        pidx[own_slot[q]] = (idx_t)j;      // a k-mer present twice: the later store wins
      }
    }
  } else {
    for (uint32_t j = tid; j < m; j += NT) {
      bool wn;
      const int s = set_insert_lane(pkeys, hcap, nk(j) >> 2, &wn);
      if (s < 0 || !wn) shared_prefix = 1;
    }
  }
  __syncthreads();

  // ---- 1b. the common case: nothing but the reference chain ---------------------------
  // m == n_ref, all prefixes distinct and the suffix of the last k-mer is not a prefix:
  // node j's only overlap edge goes to j+1 and nothing else exists (no duplicate k-mer
  // either).  Then both Dijkstra trees are the chain, every reference edge but the first
  // is stripped, and edges (source,0) and (0,1) both generate the one path 0..n_ref-1.
  {
    uint32_t not_pure = shared_prefix | (m != n_ref ? 1u : 0u) | ((a.dbg != 0 && !keep_pure) ? 1u : 0u);
    if (tid == 0 && !not_pure) {
      const uint64_t S = nk(m - 1) & a.pmask;
      uint32_t s = set_home(S, hcap);
      for (uint32_t step = 0; step < hcap; ++step) {
        const uint64_t kv = pkeys[s];
        if (kv == S) { not_pure = 1; break; }
        if (kv == EMPTY) break;
        if (++s == hcap) s = 0;
      }
    }
    if (!__syncthreads_or((int)not_pure)) {
      if (wave == 0) {
        uint32_t mincov = 0xFFFFFFFFu, maxcov = 0;
        for (uint32_t q = lane; q < n_ref; q += 64) {
          const uint32_t c = ncnt[q];
          mincov = c < mincov ? c : mincov;
          maxcov = c > maxcov ? c : maxcov;
        }
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = __shfl_xor(mincov, o); mincov = other < mincov ? other : mincov;
          const uint32_t om = __shfl_xor(maxcov, o); maxcov = om > maxcov ? om : maxcov;
        }
        if (lane == 0) {
          const unsigned long long pl = atomicAdd(&ctr[0], 1ull);
          const unsigned long long rl = atomicAdd(&ctr[1], 1ull);
          if (pl + 1 > pg_paths || rl + 1 > pg_runs) {
            atomicExch(ovf, 1ull);
            a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0;
          } else {
            const uint64_t pi = (uint64_t)pg * pg_paths + pl, ri = (uint64_t)pg * pg_runs + rl;
            a.r_start[ri] = 0; a.r_len[ri] = n_ref;
            a.p_target[pi] = t; a.p_runbase[pi] = ri; a.p_nruns[pi] = 1; a.p_len[pi] = n_ref; a.p_mincov[pi] = mincov;
            a.t_npaths[t] = 1; a.t_pathbase[t] = (uint32_t)pi; a.t_nruns[t] = 1;
            a.t_refmax[t] = maxcov;
          }
          a.t_eremoved[t] = n_ref >= 2 ? n_ref - 1 : 0;
          a.t_enonref[t] = 2;
          a.g_status[t] = T_OK;
        }
      }
      return;
    }
  }

  // ---- 1c. general case: node index per (prefix, last base); succ / pred start empty ----
  if constexpr (FUSED) {
    // a k-mer present twice (km/utils/common.py:55-59) shares one table entry: one of the
    // two nodes does not find its own index there
    uint32_t dup = 0;
#pragma unroll
    for (uint32_t q = 0; q < MAXOWN; ++q) {
      const uint32_t j = tid + q * NT;
      if (j < m && pidx[own_slot[q]] != (idx_t)j) dup = 1;
    }
    if (__syncthreads_or((int)dup)) {
      if (tid == 0) { a.status[t] = T_REPEAT; a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; }
      return;
    }
  } else {
    {
      uint4* q = reinterpret_cast<uint4*>(wsb);
      for (uint64_t x = (uint64_t)hcap / 2 + tid; x < fill_bytes / 16; x += NT) q[x] = ones;
    }
    __syncthreads();
    uint32_t dup = 0;
    for (uint32_t j = tid; j < m; j += NT) {
      const uint64_t X = nk(j);
      const uint64_t P = X >> 2;
      uint32_t s = set_home(P, hcap);
      for (uint32_t step = 0; step < hcap; ++step) {
        if (pkeys[s] == P) break;
        if (++s == hcap) s = 0;
      }
      pidx[4 * s + (uint32_t)(X & 3)] = (idx_t)j;
    }
    __syncthreads();
    // a k-mer present twice (km/utils/common.py:55-59) shares one table entry: one of the
    // two nodes does not find its own index there
    for (uint32_t j = tid; j < m; j += NT) {
      const uint64_t X = nk(j);
      const uint64_t P = X >> 2;
      uint32_t s = set_home(P, hcap);
      idx_t got = NONE;
      for (uint32_t step = 0; step < hcap; ++step) {
        const uint64_t kv = pkeys[s];
        if (kv == P) { got = pidx[4 * s + (uint32_t)(X & 3)]; break; }
        if (kv == EMPTY) break;
        if (++s == hcap) s = 0;
      }
      if (got != (idx_t)j) dup = 1;
    }
    if (__syncthreads_or((int)dup)) {
      if (tid == 0) { a.status[t] = T_REPEAT; a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; }
      return;
    }
  }
  if (a.dbg == 1) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // the two paths of "reference chain + one forward bubble a -> n_ref .. m-1 -> b" (2c, 2c')
  auto emit_bubble = [&](const uint32_t fa, const uint32_t fb) {
    // min coverage of the two paths: wave 0 the reference path, wave 1 the path through the bubble
    if (wave < 2) {
      uint32_t mc = 0xFFFFFFFFu;
      auto over = [&](uint32_t lo, uint32_t hi) {            // [lo, hi)
        for (uint32_t q = lo + lane; q < hi; q += 64) { const uint32_t c = ncnt[q]; mc = c < mc ? c : mc; }
      };
      if (wave == 0) over(0, n_ref);
      else { over(0, fa + 1); over(n_ref, m); over(fb, n_ref); }
      for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(mc, o); mc = other < mc ? other : mc; }
      if (lane == 0) scal[5 + wave] = mc;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned long long path_base = atomicAdd(&ctr[0], 2ull);
      unsigned long long run_base = atomicAdd(&ctr[1], 4ull);
      if (path_base + 2 > pg_paths || run_base + 4 > pg_runs) {
        atomicExch(ovf, 1ull);               // pools exhausted: host enlarges them and reruns the stage
        a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0;
      } else {
        path_base += (uint64_t)pg * pg_paths;
        run_base += (uint64_t)pg * pg_runs;
        a.r_start[run_base] = 0; a.r_len[run_base] = n_ref;
        a.p_target[path_base] = t; a.p_runbase[path_base] = run_base; a.p_nruns[path_base] = 1;
        a.p_len[path_base] = n_ref; a.p_mincov[path_base] = scal[5];
        a.r_start[run_base + 1] = 0; a.r_len[run_base + 1] = fa + 1;
        a.r_start[run_base + 2] = n_ref; a.r_len[run_base + 2] = m - n_ref;
        a.r_start[run_base + 3] = fb; a.r_len[run_base + 3] = n_ref - fb;
        a.p_target[path_base + 1] = t; a.p_runbase[path_base + 1] = run_base + 1; a.p_nruns[path_base + 1] = 3;
        a.p_len[path_base + 1] = (fa + 1) + (m - n_ref) + (n_ref - fb); a.p_mincov[path_base + 1] = scal[6];
        a.t_npaths[t] = 2; a.t_pathbase[t] = (uint32_t)path_base; a.t_nruns[t] = 4;
      }
      a.t_eremoved[t] = n_ref - 1;                 // the reference chain but its first edge
      a.t_enonref[t] = (m - n_ref) + 3;            // (source, 0), (0, 1), the bubble's m - n_ref + 1 edges
      a.g_status[t] = T_OK;
    }
  };

  // ---- 2c'. the same shape read off the prefix table, before the adjacency is built ------------
  // Edges are i -> j iff suffix(i) == prefix(j).  With (R1) the reference's (k-1)-mer prefixes
  // pairwise distinct and the last reference suffix no node's prefix, (R2) the first walk node
  // sharing its prefix with exactly one node, a reference node x in [1, n_ref-1] (both are children
  // of a = x - 1), (R3) every later walk node e having prefix(e) == suffix(e-1) and a prefix of its
  // own, and (R4) suffix(m-1) being the prefix of exactly one node, a reference node b > a — the
  // graph has the reference edges j -> j+1, the chain a -> n_ref -> .. -> m-1 -> b and nothing
  // else: any other edge would put a second node into a prefix slot that (R1)-(R4) say holds one.
  // All of it is in the table already: how many nodes share a node's prefix is the number of
  // occupied indices of its slot (kept in own_slot); (R3) and (R4) are one lookup per walk node of
  // its own suffix (its slot must hold exactly the next walk node, resp. b), (R1)'s rest one more.
  if constexpr (FUSED) {
    if (a.dbg == 0 && m > n_ref && n_ref >= 2) {
      uint32_t bad = 0;
#pragma unroll
      for (uint32_t q = 0; q < MAXOWN; ++q) {
        const uint32_t j = tid + q * NT;
        if (j >= m) continue;
        const uint32_t s4 = own_slot[q] & ~3u;
        uint32_t cnt = 0, other = NIL;
        for (uint32_t c = 0; c < 4; ++c) {
          const idx_t v = pidx[s4 + c];
          if (v == NONE) continue;
          ++cnt;
          if ((uint32_t)v != j) other = (uint32_t)v;
        }
        if (cnt > 2) bad = 1;
        else if (cnt == 2) {
          if (j == n_ref) { if (other >= 1 && other <= n_ref - 1) scal[3] = other - 1; else bad = 1; }
          else if (j < n_ref) { if (other != n_ref) bad = 1; }
          else bad = 1;                                           // a later walk node shares its prefix
        } else if (j == n_ref) bad = 1;                           // the first walk node hangs off nothing
        // the suffix of a walk node is the prefix of exactly the next one (R3), of exactly one
        // reference node for the last (R4); of nobody for the last reference node (R1)
        if (j >= n_ref - 1) {
          const uint64_t S = own_x[q] & a.pmask;
          uint32_t s = set_home(S, hcap), hit = NIL;
          for (uint32_t step = 0; step < hcap; ++step) {
            const uint64_t kv = pkeys[s];
            if (kv == S) { hit = s; break; }
            if (kv == EMPTY) break;
            if (++s == hcap) s = 0;
          }
          if (j == n_ref - 1) { if (hit != NIL) bad = 1; }
          else if (hit == NIL) bad = 1;
          else {
            uint32_t n_in = 0, only = NIL;
            for (uint32_t c = 0; c < 4; ++c) { const idx_t v = pidx[4 * hit + c]; if (v != NONE) { ++n_in; only = (uint32_t)v; } }
            if (n_in != 1) bad = 1;
            else if (j + 1 < m) { if (only != j + 1) bad = 1; }
            else if (only < n_ref) scal[4] = only;
            else bad = 1;
          }
        }
      }
      if (!__syncthreads_or((int)bad)) {
        const uint32_t fa = scal[3], fb = scal[4];
        if (fa < fb && fb <= n_ref - 1 && (uint64_t)(fb - fa) + 10 <= 100ull * (m - n_ref + 1)) {   // block-uniform
          emit_bubble(fa, fb);
          return;
        }
      }
    }
  }
  // ---- 2. adjacency: succ[4j+c] = node of kmer[j][1:]+c ; pred[4v+f] = j, f = first base of j
  for (uint32_t j = tid; j < m; j += NT) {
    const uint64_t X = nk(j);
    const uint64_t S = X & a.pmask;
    const uint32_t fb = (uint32_t)(X >> (2 * (k - 1))) & 3u;
    uint32_t s = set_home(S, hcap);
    for (uint32_t step = 0; step < hcap; ++step) {
      const uint64_t kv = pkeys[s];
      if (kv == S) {
        for (uint32_t c = 0; c < 4; ++c) {
          const idx_t v = pidx[4 * s + c];
          if (v != NONE && v != (idx_t)j) {                        // `if i != j`
            succ[4 * j + c] = v;
            pred[4 * (uint32_t)v + fb] = (idx_t)j;
          }
        }
        break;
      }
      if (kv == EMPTY) break;
      if (++s == hcap) s = 0;
    }
  }
  __syncthreads();
  if (a.dbg == 2) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // ---- 2b. link[j]: j -> j+1 is j's only out-edge and j+1's only in-edge ---------------
  for (uint32_t base = wave * 64; base < m; base += NT) {
    const uint32_t j = base + lane;
    bool lk = false;
    if (j + 1 < m && j != n_ref - 1) {                            // n_ref-1 also feeds the sink
      uint32_t outs = 0, ins = 0;
      bool to_next = false;
      for (uint32_t c = 0; c < 4; ++c) {
        const idx_t v = succ[4 * j + c];
        if (v != NONE) { ++outs; to_next |= (v == (idx_t)(j + 1)); }
        if (pred[4 * (j + 1) + c] != NONE) ++ins;
      }
      lk = (outs == 1) && to_next && (ins == 1);
    }
    const unsigned long long bm = __ballot(lk);
    if (lane == 0) {
      link[base >> 5] = (uint32_t)bm;
      if ((base >> 5) + 1 < nbw) link[(base >> 5) + 1] = (uint32_t)(bm >> 32);
    }
  }
  __syncthreads();          // the prefix table is dead from here on: region A is reused

  // ---- 2c. the one shape most variants have: the reference chain plus ONE forward bubble -------
  // Nodes n_ref..m-1 (the walk's own) form a single chain a -> n_ref -> ... -> m-1 -> b with
  // a < b on the reference, and there is no other irregular node.  Then (Graph.py:63-240 on this
  // graph; oracle/km_oracle.py: graph_paths): the reference route between a and b costs
  // (b - a) x 0.01 against (m - n_ref + 1) x 1.0 through the bubble, so — checked below with a
  // margin far above float32 rounding — both shortest-path trees keep the reference edges into b
  // and out of a; the chain walked from node 0 along after[] is the whole reference path, whose
  // edges are all stripped but (0, 1); every bubble edge then yields the path
  // 0..a, n_ref..m-1, b..n_ref-1 and the two surviving reference edges the reference path.
  // Two paths, written straight away; steps 3-6 are skipped.
  if constexpr (!BIG) {
    if (a.dbg == 0 && m > n_ref && n_ref >= 2 && wave == 0) {
      uint32_t z = 0;
      if (lane < nbw) {
        z = ~link[lane];
        const uint32_t lo = lane * 32;
        if (lo >= m) z = 0;
        else if (m - lo < 32) z &= (1u << (m - lo)) - 1u;
      }
      const unsigned long long words = __ballot(z != 0);
      uint32_t pos[4] = {0, 0, 0, 0}, np = 0;
      bool ok = __popcll(words) <= 4;
      for (unsigned long long r = words; ok && r; r &= r - 1) {
        const uint32_t w = (uint32_t)__ffsll((long long)r) - 1;
        for (uint32_t zz = lane_u32(z, w); zz; zz &= zz - 1) {
          if (np == 4) { ok = false; break; }
          pos[np++] = w * 32 + (uint32_t)__ffs((int)zz) - 1;
        }
      }
      // clear link bits exactly at a, b - 1 (the same for an insertion), n_ref - 1 and m - 1
      ok = ok && (np == 3 || np == 4) && pos[np - 1] == m - 1 && pos[np - 2] == n_ref - 1 && pos[np - 3] < n_ref - 1;
      uint32_t fa = 0, fb = 0;
      if (ok) {
        fa = pos[0];
        fb = pos[np - 3] + 1;
        auto degree = [&](const idx_t* adj, uint32_t u, uint32_t want0, uint32_t want1) -> bool {
          // the neighbours of u are exactly {want0, want1} (NIL = none)
          uint32_t cnt = 0, hit0 = 0, hit1 = 0;
          for (uint32_t c = 0; c < 4; ++c) {
            const idx_t v = adj[4 * u + c];
            if (v == NONE) continue;
            ++cnt;
            hit0 |= (uint32_t)v == want0;
            hit1 |= (uint32_t)v == want1;
          }
          const uint32_t wanted = (want0 != NIL) + (want1 != NIL && want1 != want0);
          return cnt == wanted && (want0 == NIL || hit0) && (want1 == NIL || hit1);
        };
        const uint32_t last = m - 1;
        ok = fb <= n_ref - 1 && fa < fb;
        ok = ok && degree(succ, fa, fa + 1, n_ref) && degree(pred, n_ref, fa, NIL);
        ok = ok && degree(succ, last, fb, NIL) && degree(pred, fb, fb - 1, last);
        if (fb - 1 != fa) ok = ok && degree(succ, fb - 1, fb, NIL) && degree(pred, fa + 1, fa, NIL);
        ok = ok && degree(pred, 0, NIL, NIL) && degree(succ, n_ref - 1, NIL, NIL);
        ok = ok && (uint64_t)(fb - fa) + 10 <= 100ull * (m - n_ref + 1);
      }
      if (lane == 0) { scal[2] = ok ? 1u : 0u; scal[3] = fa; scal[4] = fb; }
    }
    __syncthreads();
    if (scal[2]) {
      emit_bubble(scal[3], scal[4]);
      return;
    }
  }

  // first index e >= u with link[e] clear (the end of the chain through u)
  auto chain_end = [&](uint32_t u) -> uint32_t {
    uint32_t w = u >> 5;
    uint32_t z = ~link[w] & (0xFFFFFFFFu << (u & 31));
    while (z == 0 && w + 1 < nbw) { ++w; z = ~link[w]; }           // bit m-1 is always clear
    return z ? (w << 5) + (uint32_t)__ffs((int)z) - 1 : u;
  };
  // lowest s <= v with link[s..v-1] all set (the head of the chain through v)
  auto chain_head = [&](uint32_t v) -> uint32_t {
    uint32_t s = v;
    while (s > 0) {
      const uint32_t top = (s - 1) & 31, w = (s - 1) >> 5;
      const uint32_t z = ~link[w] & ((2u << top) - 1u);
      if (z) { s = (w << 5) + (31 - (uint32_t)__clz((int)z)) + 1; break; }
      s = w << 5;
    }
    return s;
  };

  for (uint32_t j = tid; j < n; j += NT) { dist_f[j] = INF; dist_b[j] = INF; }
  __syncthreads();

  if (a.dbg == 3) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // ---- 3. exact distances -------------------------------------------------------------
  // the two passes are independent: wave 0 runs the forward one, wave 1 the backward one
  if (wave < 2) {
  {
    const int dir = (int)wave;
    float* dist = dir ? dist_b : dist_f;
    const idx_t* adj = dir ? pred : succ;
    idx_t* fr = dir ? frontier_b : frontier;
    uint32_t* iq = dir ? inq_b : inq;
    uint32_t fcount = 0;
    uint32_t cur = dir ? n_ref - 1 : 0;
    float d = 0.0f + W_REF;                                         // cap edge from the root
    if (lane == 0) { dist[dir ? snk : src] = 0.0f; dist[cur] = d; }
    wsync();
    uint32_t guard = 0;
    while (guard++ <= n) {
      // (cur, d) is final: run along its chain in registers
      uint32_t end;
      if (dir == 0) {
        end = chain_end(cur);
        if (end > cur && end < n_ref && end < a.tref_len && d == a.tref[cur]) {
          // reference chain reached along the reference: its distances are the shared
          // partial sums (same float32 additions, done once on the host)
          for (uint32_t j = cur + 1 + lane; j <= end; j += 64) dist[j] = a.tref[j];
          d = a.tref[end];
        } else {
          uint32_t j = cur;
          while (j < end) {
            const uint32_t cnt = (end - j < 64u) ? end - j : 64u;
            float keep = 0.0f;
#pragma unroll 8
            for (uint32_t q = 0; q < cnt; ++q) {
              d = d + ((j + q + 1 < n_ref) ? W_REF : W_ALT);        // edge (j+q) -> (j+q+1)
              if (q == lane) keep = d;
            }
            if (lane < cnt) dist[j + 1 + lane] = keep;
            j += cnt;
          }
        }
      } else {
        end = chain_head(cur);
        if (end < cur && cur < n_ref && n_ref - 1 - end < a.tref_len && d == a.tref[n_ref - 1 - cur]) {
          for (uint32_t j = end + lane; j < cur; j += 64) dist[j] = a.tref[n_ref - 1 - j];
          d = a.tref[n_ref - 1 - end];
        } else {
          uint32_t j = cur;
          while (j > end) {
            const uint32_t cnt = (j - end < 64u) ? j - end : 64u;
            float keep = 0.0f;
#pragma unroll 8
            for (uint32_t q = 0; q < cnt; ++q) {
              d = d + ((j - q < n_ref) ? W_REF : W_ALT);            // edge (j-q-1) -> (j-q)
              if (q == lane) keep = d;
            }
            if (lane < cnt) dist[j - 1 - lane] = keep;
            j -= cnt;
          }
        }
      }
      const uint32_t u = end;
      // relax the edges leaving the chain: lanes 0..3 overlap edges, lane 4 the cap edge
      uint32_t v = 0;
      bool have = false;
      if (lane < 4 && u < m) {
        const idx_t x = adj[4 * u + lane];
        if (x != NONE) { v = x; have = true; }
      } else if (lane == 4) {
        if (dir == 0 && u == n_ref - 1) { v = snk; have = true; }
        if (dir == 1 && u == 0) { v = src; have = true; }
      }
      bool push = false;
      if (have) {
        const float nd = d + (dir ? weight(v, u) : weight(u, v));
        if (nd < dist[v]) {
          dist[v] = nd;
          // caps are never expanded; other nodes enter the frontier once
          if (v < m && !((iq[v >> 5] >> (v & 31)) & 1u)) push = true;
        }
      }
      const unsigned long long pm = __ballot(push);
      if (push) {
        fr[fcount + (uint32_t)__popcll(pm & ((1ull << lane) - 1))] = (idx_t)v;
        atomicOr(&iq[v >> 5], 1u << (v & 31));
      }
      fcount += (uint32_t)__popcll(pm);
      wsync();
      if (fcount == 0) break;
      // extract the frontier entry with the smallest distance
      uint32_t pos = 0;
      if (fcount > 1) {
        unsigned long long best = ~0ull;
        for (uint32_t base = 0; base < fcount; base += 64) {
          const uint32_t p = base + lane;
          unsigned long long key = ~0ull;
          if (p < fcount) key = ((unsigned long long)__float_as_uint(dist[fr[p]]) << 32) | p;
          for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other < key ? other : key;
          }
          best = key < best ? key : best;
        }
        pos = (uint32_t)(best & 0xFFFFFFFFu);
      }
      cur = fr[pos];
      d = dist[cur];
      wsync();
      if (lane == 0) fr[pos] = fr[fcount - 1];
      --fcount;
      wsync();
    }
  }
  }   // waves 0, 1
  __syncthreads();

  if (a.dbg == 4) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // ---- 4. predecessor arrays by the local rule ------------------------------------
  for (uint32_t j = tid; j < n; j += NT) {
    {
      idx_t best = NONE;
      float bv = INF, bd = INF;
      if (j >= 1 && j < m && ((link[(j - 1) >> 5] >> ((j - 1) & 31)) & 1u)) {
        if (dist_f[j] < INF) best = (idx_t)(j - 1);               // only in-edge
      } else if (j != src && dist_f[j] < INF) {
        auto consider = [&](uint32_t u) {
          const float du = dist_f[u];
          if (!(du < INF)) return;
          const float val = du + weight(u, j);
          if (best == NONE || val < bv || (val == bv && (du < bd || (du == bd && u < (uint32_t)best)))) {
            best = (idx_t)u; bv = val; bd = du;
          }
        };
        if (j == snk) consider(n_ref - 1);
        else {
          for (uint32_t c = 0; c < 4; ++c) { const idx_t u = pred[4 * j + c]; if (u != NONE) consider(u); }
          if (j == 0) consider(src);
        }
      }
      before[j] = best;
    }
    {
      idx_t best = NONE;
      float bv = INF, bd = INF;
      if (j + 1 < m && ((link[j >> 5] >> (j & 31)) & 1u)) {
        if (dist_b[j] < INF) best = (idx_t)(j + 1);               // only out-edge
      } else if (j != snk && dist_b[j] < INF) {
        auto consider = [&](uint32_t v) {
          const float dv = dist_b[v];
          if (!(dv < INF)) return;
          const float val = dv + weight(j, v);
          if (best == NONE || val < bv || (val == bv && (dv < bd || (dv == bd && v < (uint32_t)best)))) {
            best = (idx_t)v; bv = val; bd = dv;
          }
        };
        if (j == src) consider(0);
        else {
          for (uint32_t c = 0; c < 4; ++c) { const idx_t v = succ[4 * j + c]; if (v != NONE) consider(v); }
          if (j == n_ref - 1) consider(snk);
        }
      }
      after[j] = best;
    }
  }
  __syncthreads();

  // edge ids: 4u+c for overlap edges, 4m for source->0, 4m+1 for (n_ref-1)->sink
  auto edge_id = [&](uint32_t u, uint32_t v) -> uint32_t {
    if (u == src) return 4 * m;
    if (v == snk) return 4 * m + 1;
    for (uint32_t c = 0; c < 4; ++c) if (succ[4 * u + c] == (idx_t)v) return 4 * u + c;
    return NIL;
  };
  auto is_removed = [&](uint32_t e) -> bool { return (removed[e >> 5] >> (e & 31)) & 1u; };

  if (a.dbg == 5) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // ---- 5. strip reference edges (Graph.py:184-197) ---------------------------------
  // curs = nodes whose predecessor is the source; only node 0 has an edge from it.
  if (before[0] == (idx_t)src) {
    // common case: the sink-tree chain from node 0 is 0,1,...,n_ref-1,sink
    bool ok = true;
    for (uint32_t i = tid; i < n_ref; i += NT) {
      const uint32_t want = (i + 1 < n_ref) ? i + 1 : snk;
      if (after[i] != (idx_t)want) ok = false;
    }
    if (__syncthreads_and((int)ok)) {
      for (uint32_t i = 1 + tid; i < n_ref; i += NT) {       // first edge (0 -> 1) is kept
        const uint32_t e = edge_id(i, (i + 1 < n_ref) ? i + 1 : snk);
        if (e != NIL) atomicOr(&removed[e >> 5], 1u << (e & 31));
      }
    } else {
      if (tid == 0) {
        uint32_t cur = 0, last = NIL, hops = 0;
        while (after[cur] != NONE && hops++ <= n) {
          cur = after[cur];
          if (last != NIL && last != 0) {                      // `if last_cur and ...`
            const uint32_t e = edge_id(last, cur);
            if (e != NIL) removed[e >> 5] |= 1u << (e & 31);
          }
          last = cur;
        }
      }
    }
  }
  __syncthreads();

  if (a.dbg == 6) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // ---- 6. candidate edges and their unique representatives --------------------------
  const uint32_t n_edges = 4 * m + 2;
  uint32_t my_edges = 0, my_removed = 0;      // -v: edges of the graph / of them stripped (Graph.py:198, 231)
  for (uint32_t e = tid; e < n_edges; e += NT) {
    uint32_t ea, eb;
    bool exists;
    if (e == 4 * m) { ea = src; eb = 0; exists = true; }
    else if (e == 4 * m + 1) { ea = n_ref - 1; eb = snk; exists = true; }
    else { ea = e >> 2; const idx_t v = succ[e]; exists = (v != NONE); eb = v; }
    my_edges += exists ? 1u : 0u;
    my_removed += (exists && is_removed(e)) ? 1u : 0u;
    if (!exists || is_removed(e) || !(dist_f[ea] < INF) || !(dist_b[eb] < INF)) continue;
    // representative iff no candidate edge sits at an earlier generating position
    uint32_t x = ea, y = eb, hops = 0;
    bool keep = true;
    while (hops++ <= n) {
      if (after[x] != (idx_t)y) break;        // edge not on the sink tree: stop
      if (x == src) break;                    // position 0
      const uint32_t p = before[x];
      const uint32_t pe = edge_id(p, x);
      if (pe != NIL && !is_removed(pe)) { keep = false; break; }
      y = x; x = p;
    }
    if (keep) {
      const uint32_t slot = atomicAdd(&scal[0], 1u);
      if (slot < ccap) { cand[2 * slot] = (idx_t)ea; cand[2 * slot + 1] = (idx_t)eb; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { my_edges += __shfl_xor(my_edges, o); my_removed += __shfl_xor(my_removed, o); }
  if (lane == 0) { atomicAdd(&scal[2], my_edges); atomicAdd(&scal[7], my_removed); }
  __syncthreads();
  if (tid == 0) { a.t_eremoved[t] = scal[7]; a.t_enonref[t] = scal[2] - scal[7]; }
  const uint32_t n_cand = scal[0];
  if (n_cand > ccap) {
    if (tid == 0) { a.g_status[t] = (BIG && !a0.tids_n) ? T_INTERNAL : T_NEEDS_BIG; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; hand_to_big(); }
    return;
  }

  if (a.dbg == 7) { if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; } return; }
  // ---- 7. emit paths (caps stripped) as runs of consecutive node indices -------------
  // A path is walked chain by chain (wave-uniform): backwards from `a` along before[]
  // — inside a chain before[j] == j-1 — then forwards from `b` along after[].  Runs
  // that happen to be index-contiguous across a hop are merged.  `emit(start,len,idx)`
  // receives the merged runs with their position in path order.
  auto walk_path = [&](uint32_t ea, uint32_t eb, auto&& emit) -> uint32_t {
    // backward part: count merged runs first (a hop is contiguous iff before[head] == head-1)
    uint32_t nback = 0;
    if (ea != src) {
      uint32_t x = ea, hops = 0;
      nback = 1;
      while (hops++ <= n) {
        const uint32_t s = chain_head(x);
        const uint32_t p = before[s];
        if (p == src || p == (uint32_t)NONE) break;
        if (p + 1 != s) ++nback;
        x = p;
      }
    }
    const bool glue = (ea != src) && (eb != snk) && (ea + 1 == eb);   // junction a -> b contiguous
    // first forward merged run
    uint32_t f_lo = NIL, f_hi = NIL, fx = NIL;      // fx: first node after that run
    if (eb != snk) {
      f_lo = eb;
      uint32_t x = eb, hops = 0;
      while (hops++ <= n) {
        const uint32_t e = chain_end(x);
        f_hi = e;
        const uint32_t q = after[e];
        if (q == snk || q == (uint32_t)NONE) { fx = NIL; break; }
        if (q != e + 1) { fx = q; break; }
        x = q;
      }
    }
    // backward runs, discovered last-to-first
    if (ea != src) {
      uint32_t x = ea, hops = 0, idx = nback - 1;
      uint32_t hi = glue ? f_hi : ea;
      uint32_t lo = ea;
      while (hops++ <= n) {
        const uint32_t s = chain_head(x);
        lo = s;
        const uint32_t p = before[s];
        if (p == src || p == (uint32_t)NONE) break;
        if (p + 1 != s) { emit(lo, hi - lo + 1, idx); --idx; hi = p; }
        x = p;
      }
      emit(lo, hi - lo + 1, idx);
    }
    uint32_t total = nback;
    if (eb != snk) {
      if (!glue) { emit(f_lo, f_hi - f_lo + 1, total); ++total; }
      uint32_t x = fx, hops = 0;
      while (x != NIL && hops++ <= n) {
        const uint32_t lo = x;
        uint32_t hi = x, nxt = NIL, h2 = 0;
        while (h2++ <= n) {
          const uint32_t e = chain_end(hi);
          hi = e;
          const uint32_t q = after[e];
          if (q == snk || q == (uint32_t)NONE) { nxt = NIL; break; }
          if (q != e + 1) { nxt = q; break; }
          hi = q;
        }
        emit(lo, hi - lo + 1, total);
        ++total;
        x = nxt;
      }
    }
    return total;
  };

  // Emission is wave-uniform work: the candidates are dealt round-robin to the four waves.
  // pass 1: runs per path
  uint32_t* cnt_runs = reinterpret_cast<uint32_t*>(frontier);      // [n_cand], the frontiers are dead by now
  for (uint32_t p = wave; p < n_cand; p += NT / 64) {
    const uint32_t nr = walk_path(cand[2 * p], cand[2 * p + 1], [](uint32_t, uint32_t, uint32_t) {});
    if (lane == 0) cnt_runs[p] = nr;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
    for (uint32_t p = 0; p < n_cand; ++p) { const uint32_t c = cnt_runs[p]; cnt_runs[p] = total; total += c; }
    unsigned long long path_base = atomicAdd(&ctr[0], (unsigned long long)n_cand);
    unsigned long long run_base = atomicAdd(&ctr[1], (unsigned long long)total);
    uint32_t over = 0;
    if (path_base + n_cand > pg_paths || run_base + total > pg_runs) {
      atomicExch(ovf, 1ull);
      over = 1;
    }
    path_base += (uint64_t)pg * pg_paths;
    run_base += (uint64_t)pg * pg_runs;
    scal[2] = (uint32_t)path_base; scal[3] = (uint32_t)(path_base >> 32);
    scal[4] = (uint32_t)run_base; scal[5] = (uint32_t)(run_base >> 32);
    scal[6] = over; scal[7] = total;
  }
  __syncthreads();
  const uint64_t path_base = ((uint64_t)scal[3] << 32) | scal[2];
  const uint64_t run_base = ((uint64_t)scal[5] << 32) | scal[4];
  if (scal[6]) {      // pools exhausted: host enlarges them and reruns the stage
    if (tid == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; a.t_nruns[t] = 0; }
    return;
  }
  // pass 2: write runs, lengths and min coverage
  for (uint32_t p = wave; p < n_cand; p += NT / 64) {
    const uint64_t rcur = run_base + cnt_runs[p];
    uint32_t plen = 0, mincov = 0xFFFFFFFFu;
    const uint32_t nr = walk_path(cand[2 * p], cand[2 * p + 1],
                                  [&](uint32_t start, uint32_t len, uint32_t idx) {
      if (lane == 0) { a.r_start[rcur + idx] = start; a.r_len[rcur + idx] = len; }
      plen += len;
      for (uint32_t q = lane; q < len; q += 256) {      // four independent loads per lane
        uint32_t c0 = ncnt[start + q], c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu, c3 = 0xFFFFFFFFu;
        if (q + 64 < len) c1 = ncnt[start + q + 64];
        if (q + 128 < len) c2 = ncnt[start + q + 128];
        if (q + 192 < len) c3 = ncnt[start + q + 192];
        c0 = c0 < c1 ? c0 : c1; c2 = c2 < c3 ? c2 : c3; c0 = c0 < c2 ? c0 : c2;
        mincov = c0 < mincov ? c0 : mincov;
      }
    });
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t other = __shfl_xor(mincov, o);
      mincov = other < mincov ? other : mincov;
    }
    if (lane == 0) {
      const uint64_t pi = path_base + p;
      a.p_target[pi] = t;
      a.p_runbase[pi] = rcur;
      a.p_nruns[pi] = nr;
      a.p_len[pi] = plen;
      a.p_mincov[pi] = mincov;
    }
  }
  if (tid == 0) {
    a.g_status[t] = T_OK;
    a.t_npaths[t] = n_cand;
    a.t_pathbase[t] = (uint32_t)path_base;
    a.t_nruns[t] = scal[7];
  }
}

}  // namespace kmd
