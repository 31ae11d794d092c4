// graph_kernel.h — MutationFinder.graph_analysis + Graph for a batch of targets.
//
// Reference: km/utils/MutationFinder.py:496-572 (overlap edges weight 1, reference
// and cap edges weight 0.01), km/utils/Graph.py:63-119 (dense float32 Dijkstra,
// first-index argmin, strict '<' relaxation), :121-198 (two runs, strip reference
// edges), :200-240 (one source->sink path per remaining edge, unique set).
//
// One 64-lane workgroup per target; node numbering as written by the walk kernel,
// BigBang (source) = m, BigCrunch (sink) = m + 1 with m = n_nodes.
//
// How the dense O(n^2) algorithm is reproduced exactly on a sparse graph:
//  * dist[]  With strictly positive weights and monotone float32 addition the
//    distances Dijkstra ends with are the unique solution of
//    dist[j] = min_i fl(dist[i] + w_ij); any label-setting order yields them.  A
//    frontier Dijkstra over the <=4 successors (predecessors) per node computes
//    them with the same float32 additions (hop by hop along the path).
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
// to (start, len)) into pools reserved with one atomicAdd per target.
#pragma once
#include <type_traits>

#include "device_common.h"
#include "walk_kernel.h"

namespace kmd {

struct GraphArgs {
  int k;
  uint64_t kmask;
  const uint32_t* tids;      // BIG: targets to run; nullptr = blockIdx.x
  uint32_t n_targets;
  const uint64_t* node_kmer;
  const uint32_t* node_cnt;
  const uint64_t* node_base;
  const uint32_t* n_nodes;
  const uint32_t* n_ref;
  const uint32_t* status;
  // outputs
  uint32_t* g_status;        // per target: T_OK / T_NEEDS_BIG / T_INTERNAL
  uint32_t* t_npaths;        // per target
  uint32_t* t_pathbase;      // per target: first path record
  unsigned long long* counters;  // [0] paths used, [1] runs used, [2] overflow flag
  uint64_t path_pool, run_pool;  // capacities
  uint32_t* p_target;
  uint64_t* p_runbase;
  uint32_t* p_nruns;
  uint32_t* p_len;
  uint32_t* p_mincov;
  uint32_t* r_start;
  uint32_t* r_len;
  // geometry
  uint32_t ncap;   // max nodes incl. caps
  uint32_t hcap;   // hash slots, multiple of 64, >= 2 * ncap
  unsigned char* g_ws;
  uint64_t g_stride;
};

template <typename idx_t>
__host__ __device__ inline uint64_t graph_ws_bytes(uint32_t ncap, uint32_t hcap) {
  uint64_t b = 0;
  b += (uint64_t)hcap * 8;                 // keys
  b += (uint64_t)ncap * 4 * 3;             // dist_f, dist_b, cnt
  b += (uint64_t)((ncap + 31) / 32) * 4;   // inq bits
  b += (uint64_t)((4 * (uint64_t)ncap + 2 + 31) / 32) * 4;  // removed bits
  b += 16;                                 // scalars
  b += (uint64_t)hcap * sizeof(idx_t);     // hash -> node index
  b += (uint64_t)ncap * 4 * sizeof(idx_t) * 2;  // succ, pred
  b += (uint64_t)ncap * sizeof(idx_t) * 3;      // before, after, frontier
  b += (uint64_t)ncap * 2 * sizeof(idx_t);      // candidate edges (a, b)
  return (b + 15) & ~15ull;
}

__device__ inline int hash_find_lane(const uint64_t* keys, uint32_t cap, uint64_t key) {
  uint32_t s = set_home(key, cap);
  for (uint32_t step = 0; step < cap; ++step) {
    const uint64_t kv = keys[s];
    if (kv == key) return (int)s;
    if (kv == EMPTY) return -1;
    if (++s == cap) s = 0;
  }
  return -1;
}

template <bool BIG>
__global__ __launch_bounds__(64) void k_graph(GraphArgs a) {
  using idx_t = typename std::conditional<BIG, uint32_t, uint16_t>::type;
  constexpr idx_t NONE = (idx_t)~(idx_t)0;
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t t = a.tids ? a.tids[blockIdx.x] : blockIdx.x;

  if (a.status[t] != T_OK) {
    if (lane == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; }
    return;
  }
  const uint32_t m = a.n_nodes[t];
  const uint32_t n_ref = a.n_ref[t];
  const uint32_t n = m + 2, src = m, snk = m + 1;
  const uint32_t ncap = a.ncap, hcap = a.hcap;
  if (n > ncap || (uint64_t)2 * n > hcap || (!BIG && n >= 0xFFFFu)) {
    if (lane == 0) { a.g_status[t] = BIG ? T_INTERNAL : T_NEEDS_BIG; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; }
    return;
  }
  const uint64_t nb = a.node_base[t];
  const uint64_t* nk = a.node_kmer + nb;
  const int k = a.k;

  unsigned char* wsb;
  if constexpr (BIG) wsb = a.g_ws + (uint64_t)blockIdx.x * a.g_stride;
  else wsb = smem;
  uint64_t* keys = reinterpret_cast<uint64_t*>(wsb);
  float* dist_f = reinterpret_cast<float*>(keys + hcap);
  float* dist_b = dist_f + ncap;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(dist_b + ncap);
  uint32_t* inq = cnt + ncap;
  uint32_t* removed = inq + (ncap + 31) / 32;
  const uint32_t n_removed_words = (uint32_t)((4 * (uint64_t)ncap + 2 + 31) / 32);
  uint32_t* scal = removed + n_removed_words;      // [0] candidate count, [1] flag
  idx_t* hidx = reinterpret_cast<idx_t*>(scal + 4);
  idx_t* succ = hidx + hcap;
  idx_t* pred = succ + (uint64_t)4 * ncap;
  idx_t* before = pred + (uint64_t)4 * ncap;
  idx_t* after = before + ncap;
  idx_t* frontier = after + ncap;
  idx_t* cand = frontier + ncap;                   // pairs (a, b)
  const uint32_t ccap = ncap;

  const float INF = __int_as_float(0x7F800000);
  const float W_REF = 0.01f, W_ALT = 1.0f;
  auto weight = [&](uint32_t u, uint32_t v) -> float {
    if (u == src || v == snk) return W_REF;                       // cap edges
    return (u + 1 == v && v < n_ref) ? W_REF : W_ALT;             // reference edge i -> i+1
  };

  // ---- 1. node hash ---------------------------------------------------------------
  for (uint32_t s = lane; s < hcap; s += 64) keys[s] = EMPTY;
  for (uint32_t j = lane; j < n; j += 64) { dist_f[j] = INF; dist_b[j] = INF; }
  for (uint32_t w = lane; w < (ncap + 31) / 32; w += 64) inq[w] = 0;
  for (uint32_t w = lane; w < n_removed_words; w += 64) removed[w] = 0;
  if (lane < 4) scal[lane] = 0;
  __syncthreads();
  for (uint32_t j = lane; j < m; j += 64) {
    bool wn;
    const int s = set_insert_lane(keys, hcap, nk[j], &wn);
    if (s >= 0) hidx[s] = (idx_t)j;
    cnt[j] = a.node_cnt[nb + j];
  }
  __syncthreads();
  // ---- 2. (k-1)-overlap adjacency: succ[4j+c] = node of kmer[j][1:]+c, pred[4j+c] = c+kmer[j][:-1]
  for (uint32_t j = lane; j < m; j += 64) {
    const uint64_t X = nk[j];
    for (uint32_t c = 0; c < 4; ++c) {
      const uint64_t child = ((X << 2) | c) & a.kmask;
      int s = hash_find_lane(keys, hcap, child);
      idx_t v = (s >= 0) ? hidx[s] : NONE;
      if (v == (idx_t)j) v = NONE;                               // `if i != j`
      succ[4 * j + c] = v;
      const uint64_t par = (X >> 2) | ((uint64_t)c << (2 * (k - 1)));
      s = hash_find_lane(keys, hcap, par);
      idx_t u = (s >= 0) ? hidx[s] : NONE;
      if (u == (idx_t)j) u = NONE;
      pred[4 * j + c] = u;
    }
  }
  // the capping nodes have no overlap edges (their two cap edges are handled apart)
  if (lane < 8) { succ[4 * m + lane] = NONE; pred[4 * m + lane] = NONE; }
  __syncthreads();

  // ---- 3. exact distances: frontier Dijkstra, forward from source, backward from sink
  for (int dir = 0; dir < 2; ++dir) {
    float* dist = dir ? dist_b : dist_f;
    const idx_t* adj = dir ? pred : succ;
    uint32_t fcount = 0;
    for (uint32_t w = lane; w < (ncap + 31) / 32; w += 64) inq[w] = 0;
    if (lane == 0) {
      if (dir == 0) { dist[src] = 0.0f; dist[0] = 0.0f + W_REF; frontier[0] = (idx_t)0; }
      else { dist[snk] = 0.0f; dist[n_ref - 1] = 0.0f + W_REF; frontier[0] = (idx_t)(n_ref - 1); }
    }
    fcount = 1;
    __syncthreads();
    uint32_t guard = 0;
    while (fcount > 0 && guard++ <= n) {
      // select the frontier entry with the smallest distance
      uint32_t pos = 0;
      if (fcount > 1) {
        unsigned long long best = ~0ull;
        for (uint32_t base = 0; base < fcount; base += 64) {
          const uint32_t p = base + lane;
          unsigned long long key = ~0ull;
          if (p < fcount)
            key = ((unsigned long long)__float_as_uint(dist[frontier[p]]) << 32) | p;
          for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other < key ? other : key;
          }
          best = key < best ? key : best;
        }
        pos = (uint32_t)(best & 0xFFFFFFFFu);
      }
      const uint32_t u = frontier[pos];
      const float du = dist[u];
      __syncthreads();
      if (lane == 0) {
        frontier[pos] = frontier[fcount - 1];
      }
      --fcount;
      // relax: lanes 0..3 the overlap edges, lane 4 the cap edge
      uint32_t v = 0;
      bool have = false;
      if (lane < 4 && u < m) {              // caps are sinks of their own pass: never expanded
        const idx_t x = adj[4 * u + lane];
        if (x != NONE) { v = x; have = true; }
      } else if (lane == 4) {
        if (dir == 0 && u == n_ref - 1) { v = snk; have = true; }
        if (dir == 1 && u == 0) { v = src; have = true; }
      }
      bool push = false;
      if (have) {
        const float w = dir ? weight(v, u) : weight(u, v);
        const float nd = du + w;
        if (nd < dist[v]) {
          dist[v] = nd;
          const uint32_t bit = 1u << (v & 31);
          if (!(inq[v >> 5] & bit)) { push = true; }
        }
      }
      __syncthreads();
      const unsigned long long pm = __ballot(push);
      if (push) {
        const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1));
        frontier[fcount + rank] = (idx_t)v;
        atomicOr(&inq[v >> 5], 1u << (v & 31));
      }
      fcount += (uint32_t)__popcll(pm);
      __syncthreads();
    }
  }

  // ---- 4. predecessor arrays by the local rule ------------------------------------
  for (uint32_t j = lane; j < n; j += 64) {
    // before[j]: in-neighbour minimising (dist_f[u] + w(u,j), dist_f[u], u)
    {
      idx_t best = NONE;
      float bv = INF, bd = INF;
      if (j != src && dist_f[j] < INF) {
        auto consider = [&](uint32_t u) {
          const float d = dist_f[u];
          if (!(d < INF)) return;
          const float val = d + weight(u, j);
          if (best == NONE || val < bv || (val == bv && (d < bd || (d == bd && u < (uint32_t)best)))) {
            best = (idx_t)u; bv = val; bd = d;
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
      if (j != snk && dist_b[j] < INF) {
        auto consider = [&](uint32_t v) {
          const float d = dist_b[v];
          if (!(d < INF)) return;
          const float val = d + weight(j, v);
          if (best == NONE || val < bv || (val == bv && (d < bd || (d == bd && v < (uint32_t)best)))) {
            best = (idx_t)v; bv = val; bd = d;
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
    return 0xFFFFFFFFu;
  };
  auto is_removed = [&](uint32_t e) -> bool { return (removed[e >> 5] >> (e & 31)) & 1u; };

  // ---- 5. strip reference edges (Graph.py:184-197) ---------------------------------
  // curs = nodes whose predecessor is the source; only node 0 has an edge from it.
  if (before[0] == (idx_t)src) {
    // common case: the sink-tree chain from node 0 is 0,1,...,n_ref-1,sink
    bool ok = true;
    for (uint32_t i = lane; i < n_ref; i += 64) {
      const uint32_t want = (i + 1 < n_ref) ? i + 1 : snk;
      if (after[i] != (idx_t)want) ok = false;
    }
    if (__all((int)ok)) {
      for (uint32_t i = 1 + lane; i < n_ref; i += 64) {      // first edge (0 -> 1) is kept
        const uint32_t e = edge_id(i, (i + 1 < n_ref) ? i + 1 : snk);
        if (e != 0xFFFFFFFFu) atomicOr(&removed[e >> 5], 1u << (e & 31));
      }
    } else {
      if (lane == 0) {
        uint32_t cur = 0, last = 0xFFFFFFFFu, hops = 0;
        while (after[cur] != NONE && hops++ <= n) {
          cur = after[cur];
          if (last != 0xFFFFFFFFu && last != 0) {              // `if last_cur and ...`
            const uint32_t e = edge_id(last, cur);
            if (e != 0xFFFFFFFFu) removed[e >> 5] |= 1u << (e & 31);
          }
          last = cur;
        }
      }
    }
  }
  __syncthreads();

  // ---- 6. candidate edges and their unique representatives --------------------------
  const uint32_t n_edges = 4 * m + 2;
  bool overflow = false;
  for (uint32_t e0 = 0; e0 < n_edges; e0 += 64) {
    const uint32_t e = e0 + lane;
    bool keep = false;
    uint32_t ea = 0, eb = 0;
    if (e < n_edges) {
      bool exists;
      if (e == 4 * m) { ea = src; eb = 0; exists = true; }
      else if (e == 4 * m + 1) { ea = n_ref - 1; eb = snk; exists = true; }
      else { ea = e >> 2; const idx_t v = succ[e]; exists = (v != NONE); eb = v; }
      if (exists && !is_removed(e) && dist_f[ea] < INF && dist_b[eb] < INF) {
        // representative iff no candidate edge sits at an earlier generating position
        uint32_t x = ea, y = eb, hops = 0;
        keep = true;
        while (hops++ <= n) {
          if (after[x] != (idx_t)y) break;        // edge not on the sink tree: stop
          if (x == src) break;                    // position 0
          const uint32_t p = before[x];
          const uint32_t pe = edge_id(p, x);
          if (pe != 0xFFFFFFFFu && !is_removed(pe)) { keep = false; break; }
          y = x; x = p;
        }
      }
    }
    const unsigned long long km = __ballot(keep);
    const uint32_t basec = scal[0];
    if (keep) {
      const uint32_t slot = basec + (uint32_t)__popcll(km & ((1ull << lane) - 1));
      if (slot < ccap) { cand[2 * slot] = (idx_t)ea; cand[2 * slot + 1] = (idx_t)eb; }
      else overflow = true;
    }
    __syncthreads();
    if (lane == 0) scal[0] = basec + (uint32_t)__popcll(km);
    __syncthreads();
  }
  const uint32_t n_cand = scal[0];
  if (__any((int)overflow) || n_cand > ccap) {
    if (lane == 0) { a.g_status[t] = BIG ? T_INTERNAL : T_NEEDS_BIG; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; }
    return;
  }

  // ---- 7. emit paths (caps stripped) as runs of consecutive node indices -------------
  // pass 1: count runs / nodes / min coverage per path (one lane per path)
  uint32_t total_runs = 0;
  for (uint32_t p0 = 0; p0 < n_cand; p0 += 64) {
    const uint32_t p = p0 + lane;
    uint32_t nruns = 0;
    if (p < n_cand) {
      const uint32_t ea = cand[2 * p], eb = cand[2 * p + 1];
      uint32_t prev = 0xFFFFFFFFu, hops = 0;
      for (uint32_t x = ea; x != src && x != NONE && hops <= n; x = before[x], ++hops) {
        if (prev == 0xFFFFFFFFu || x + 1 != prev) ++nruns;      // walking backwards: x == prev-1 extends
        prev = x;
      }
      // junction a -> b continues the run iff b == a + 1
      prev = (ea == src) ? 0xFFFFFFFFu : ea;
      hops = 0;
      for (uint32_t x = eb; x != snk && x != NONE && hops <= n; x = after[x], ++hops) {
        if (prev == 0xFFFFFFFFu || prev + 1 != x) ++nruns;
        prev = x;
      }
    }
    for (int o = 32; o > 0; o >>= 1) nruns += __shfl_xor(nruns, o);
    total_runs += nruns;
  }
  unsigned long long run_base = 0, path_base = 0;
  if (lane == 0) {
    path_base = atomicAdd(&a.counters[0], (unsigned long long)n_cand);
    run_base = atomicAdd(&a.counters[1], (unsigned long long)total_runs);
    if (path_base + n_cand > a.path_pool || run_base + total_runs > a.run_pool) {
      atomicExch(&a.counters[2], 1ull);
      scal[1] = 1;
    }
  }
  path_base = __shfl(path_base, 0);
  run_base = __shfl(run_base, 0);
  __syncthreads();
  if (scal[1]) {        // pools exhausted: host enlarges them and reruns the stage
    if (lane == 0) { a.g_status[t] = T_OK; a.t_npaths[t] = 0; a.t_pathbase[t] = 0; }
    return;
  }
  // pass 2: write the records
  uint32_t run_cursor = 0;
  for (uint32_t p0 = 0; p0 < n_cand; p0 += 64) {
    const uint32_t p = p0 + lane;
    uint32_t nruns = 0, plen = 0, mincov = 0xFFFFFFFFu;
    uint32_t ea = 0, eb = 0;
    if (p < n_cand) {
      ea = cand[2 * p]; eb = cand[2 * p + 1];
      uint32_t prev = 0xFFFFFFFFu, hops = 0;
      for (uint32_t x = ea; x != src && x != NONE && hops <= n; x = before[x], ++hops) {
        if (prev == 0xFFFFFFFFu || x + 1 != prev) ++nruns;
        prev = x;
      }
      prev = (ea == src) ? 0xFFFFFFFFu : ea;
      hops = 0;
      for (uint32_t x = eb; x != snk && x != NONE && hops <= n; x = after[x], ++hops) {
        if (prev == 0xFFFFFFFFu || prev + 1 != x) ++nruns;
        prev = x;
      }
    }
    // exclusive prefix of nruns over lanes
    uint32_t incl = nruns;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o);
      if ((int)lane >= o) incl += up;
    }
    const uint32_t excl = incl - nruns;
    const uint32_t chunk_total = __shfl(incl, 63);
    if (p < n_cand) {
      const uint64_t rb = run_base + run_cursor + excl;
      // backward part: runs are discovered last-to-first; fill from its end
      uint32_t nback = 0;
      {
        uint32_t prev = 0xFFFFFFFFu, hops = 0;
        for (uint32_t x = ea; x != src && x != NONE && hops <= n; x = before[x], ++hops) {
          if (prev == 0xFFFFFFFFu || x + 1 != prev) ++nback;
          prev = x;
        }
      }
      {
        uint32_t prev = 0xFFFFFFFFu, hops = 0, ri = nback, rlen = 0;
        for (uint32_t x = ea; x != src && x != NONE && hops <= n; x = before[x], ++hops) {
          if (prev == 0xFFFFFFFFu || x + 1 != prev) {
            if (prev != 0xFFFFFFFFu) { a.r_start[rb + ri] = prev; a.r_len[rb + ri] = rlen; }
            --ri; rlen = 0;
          }
          ++rlen; ++plen;
          const uint32_t c = cnt[x];
          mincov = c < mincov ? c : mincov;
          prev = x;
        }
        if (prev != 0xFFFFFFFFu) { a.r_start[rb + ri] = prev; a.r_len[rb + ri] = rlen; }
      }
      {
        // forward part; the first node may extend the last backward run
        uint32_t prev = (ea == src) ? 0xFFFFFFFFu : ea;
        uint32_t ri = nback;          // index of the next new run
        uint32_t hops = 0;
        for (uint32_t x = eb; x != snk && x != NONE && hops <= n; x = after[x], ++hops) {
          if (prev == 0xFFFFFFFFu || prev + 1 != x) {
            a.r_start[rb + ri] = x; a.r_len[rb + ri] = 1; ++ri;
          } else {
            a.r_len[rb + ri - 1] += 1;
          }
          ++plen;
          const uint32_t c = cnt[x];
          mincov = c < mincov ? c : mincov;
          prev = x;
        }
      }
      const uint64_t pi = path_base + p;
      a.p_target[pi] = t;
      a.p_runbase[pi] = rb;
      a.p_nruns[pi] = nruns;
      a.p_len[pi] = plen;
      a.p_mincov[pi] = mincov;
    }
    run_cursor += chunk_total;
  }
  if (lane == 0) {
    a.g_status[t] = T_OK;
    a.t_npaths[t] = n_cand;
    a.t_pathbase[t] = (uint32_t)path_base;
  }
}

}  // namespace kmd
