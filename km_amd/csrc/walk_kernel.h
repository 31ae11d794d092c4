// walk_kernel.h — MutationFinder.__init__ / __extend for a batch of targets.
//
// Reference: km/utils/MutationFinder.py:100-124 (register the target's k-mers,
// extend from each) and :137-165 (recursive DFS with budgets), calling
// Jellyfish.get_child / query (km/utils/Jellyfish.py:47-72).
//
// Three kernels:
//
//  k_pack  (one wave per target)  ASCII -> 2-bit words, per-target initial status
//          (empty / bad base / node limit already hit by the target itself).
//
//  k_seed  (one THREAD per target k-mer, flat over the whole batch, no per-target
//          LDS state, full occupancy — the HBM-bound part and ~85 % of all logical
//          probes).  Thread (t, i) fetches ONE 16-byte sibling-bucket slot
//          (device_common.h) which yields the four child counts of
//          get_child(ref[i]) AND the count of ref[i+1], thresholds the children
//          and classifies the seed: "trivial" when it has no child or its only
//          child is the next target k-mer (then __extend registers nothing new),
//          else its bit is set in the target's flag bitmap and the target joins
//          the flagged list.  Trivial seeds cannot change node_data, so handling
//          them out of order is exact.
//
//  k_dfs   (one wave per FLAGGED target)  builds the target's node set in LDS
//          and walks the flagged seeds one after the other in target order with
//          the reference's exact DFS semantics (child order ACGT, len(stack) >
//          max_stack, breaks > max_break, node limit checked at every extension
//          call, registration of the whole stack on a rejoin or a loop).  All 64
//          lanes cooperate on each step: set membership is one LDS window read +
//          ballot, stack registration and un-marking are lane-strided.  Only
//          frames that still have untried children are kept as "branch frames"
//          (at most max_break of them), so unwinding jumps straight to the next
//          untried child.  BIG=true keeps the per-target state in a global
//          workspace for the few targets that outgrow the LDS budget.
//
// A k-mer occurring twice in a target (the reference's ValueError,
// km/utils/common.py:55-59) is detected where a hash of the target's k-mers is
// built anyway: in k_dfs for flagged targets and in the graph kernel for all.
//
// Node order written out (canonical order): target k-mers in target
// order, then registered k-mers in registration (stack) order.
#pragma once
#include "device_common.h"

namespace kmd {

constexpr uint32_t ST_NODE = 1, ST_ONSTACK = 2, ST_POPPED = 3;
// One 16-bit word per slot of k_dfs's node set: the state above in its two top bits and, for a node,
// its index (node i = i-th target k-mer, then registration order) — what the epilogue of k_dfs needs
// to read the graph's shape off the set.  (The large tier never looks at the index.)
__device__ inline uint16_t slot_meta(uint32_t state, uint32_t idx) { return (uint16_t)((state << 14) | (idx & 0x3FFFu)); }
__device__ inline uint32_t meta_state(uint16_t m) { return (uint32_t)m >> 14; }
__device__ inline uint32_t meta_index(uint16_t m) { return (uint32_t)m & 0x3FFFu; }
constexpr uint32_t T_OK = 0, T_NODE_LIMIT = 1, T_REPEAT = 2, T_EMPTY = 3, T_BAD_BASE = 4,
                   T_INTERNAL = 5, T_NEEDS_BIG = 100;
constexpr uint64_t DFS_STEP_LIMIT = 1ull << 32;
constexpr uint32_t SEED_BLOCK = 256;     // seeds per k_seed work item

constexpr uint32_t POOL_GROUPS = 64;       // path / run pools: bump-allocation counters, one 128-B line each
constexpr uint32_t POOL_CTR_STRIDE = 16;   // uint64 per group
constexpr uint32_t NOT_BARE = 0xFFFFFFFFu;

// What the epilogue of k_dfs writes (graph_kernel.h has the meaning of every array): kept in device
// memory and read once per wave, so that it costs k_dfs no registers while it walks.
struct EpiArgs {
  unsigned long long* counters;
  uint64_t path_pool, run_pool;
  uint32_t* p_target;
  uint64_t* p_runbase;
  uint32_t* p_nruns;
  uint32_t* p_len;
  uint32_t* p_mincov;
  uint32_t* r_start;
  uint32_t* r_len;
  uint32_t* g_status;
  uint32_t* t_npaths;
  uint32_t* t_pathbase;
  uint32_t* t_nruns;
  uint32_t* t_refmax;
  uint32_t* left;         // flagged targets the epilogue did not answer: k_graph's work list
  uint32_t* n_left;       // device counter
  uint32_t* t_eremoved;   // per target: reference edges the graph stage strips / edges it keeps — what the reference
  uint32_t* t_enonref;    // logs with -v (km/utils/Graph.py:198, 231), in our node order
};
constexpr uint32_t EPI_CHUNKS = 3;         // walk-discovered nodes the epilogue looks at: up to 192 (fast tier: 160)
constexpr uint32_t EPI_MAX_BUBBLES = 8;    // more than that: left to k_graph

struct __attribute__((aligned(16))) BranchFrame {
  uint4 c4;
  uint32_t depth, mask, brk, pad;
};

// Everything the three kernels share.
struct WalkArgs {
  TableView tab;
  const uint8_t* bases;       // ASCII input (k_pack only)
  const uint64_t* toff;       // base offsets (target lengths)
  uint64_t* packed;           // 2-bit packed targets
  const uint64_t* woff;       // word offsets into `packed`
  uint32_t n_targets;
  double ratio;
  int64_t n_cutoff;
  double nc;                       // (double)n_cutoff: the kernels compare against it (device_common.h: child_mask)
  uint64_t thr_below;         // sums < thr_below have the threshold thr_T (device_common.h: threshold_shortcut)
  uint32_t thr_T;
  uint32_t max_stack, max_break, max_node;
  // k_seed work items: one self-contained 128-byte record per SEED_BLOCK seeds, written
  // by k_pack: [0] target | first seed << 32, [1] n_ref | valid << 32, [2] node base,
  // [3] flag-word offset | packed-word offset << 32, [4..15] the 12 packed words covering the item's k-mers
  uint64_t* items;
  const uint32_t* item_off;   // first item of each target
  uint32_t n_items;
  // flagged seeds / targets
  uint32_t* flagbits;         // per target bitmap, words at fw_off[t]
  const uint64_t* fw_off;
  uint32_t* tflag;            // per target: already in the flagged list
  uint32_t* flagged;          // list of flagged targets (any order)
  uint32_t* n_flagged;        // device counter
  // what k_dfs needs to start on entry i of that list, in one 32-byte record written by k_seed
  // (instead of a chain of dependent header loads): {t, n_ref, node base lo / hi} {flag-word
  // offset, packed-word offset, 0, 0}
  uint4* flag_rec;
  uint32_t fast_extra;        // fast tier: a target's node storage holds n_ref + fast_extra nodes
  const EpiArgs* epi;         // fast tier: where the epilogue delivers paths (null: no epilogue, no list)
  // k_dfs target selection: list + count (device counter or host value)
  const uint32_t* list;
  const uint32_t* n_list_dev;
  uint32_t n_list_host;
  // per-target outputs
  uint64_t* node_kmer;
  uint32_t* node_cnt;
  uint64_t* node_base;             // where a target's nodes live now (k_pack resets it to node_base0; the large tier re-homes a target)
  uint32_t* node_cap;
  const uint64_t* node_base0;      // the fast-tier layout (never changes between two km_batch_set_targets)
  // large tier on the device: targets the fast kernels cannot hold are appended to big_walk (count in big_ctl[0]; k_graph's
  // to big_graph, big_ctl[1]) and finished by a second launch in the same stream, a slot of `big_entry` nodes each in
  // the region of the node pools that starts at big_region — no host round trip.  What does not fit there (more than
  // big_slots targets, a target above big_entry) keeps T_NEEDS_BIG and is finished by the host as before.
  uint32_t* big_ctl;
  uint32_t* big_walk;
  uint32_t big_slots;
  uint32_t big_entry;
  uint64_t big_region;
  uint32_t big_prep;               // k_dfs<BIG>: this launch works off big_walk and re-homes its targets itself
  // -v: where the walk met a k-mer that is on its stack and not yet a node ('Broke loop at kmer', km/utils/
  // MutationFinder.py:160-161): pairs {target, node index of that k-mer}, in walk order per target; loop_ctl[0] counts
  // them (the list holds loop_cap pairs, what does not fit is counted only).  t_eremoved / t_enonref: see EpiArgs.
  uint32_t* loop_list;
  uint32_t* loop_ctl;
  uint32_t loop_cap;
  uint32_t* t_eremoved;
  uint32_t* t_enonref;
  uint32_t* n_nodes;
  uint32_t* n_ref;
  uint32_t* status;
  unsigned long long* probes;      // per target, written by k_seed (one atomic per wave)
  unsigned long long* dfs_probes;  // per target, written by k_dfs
  unsigned long long* fetches;
  // k_dfs workspace geometry (LDS for the fast tier, per-block global for BIG)
  uint32_t hs_cap;      // node-set slots, multiple of 64 (walk-discovered k-mers, the stack, and the few target
                        // k-mers that lost their slot of the position table)
  uint32_t pcap;        // position-table slots (a power of two >= 64): the target's own k-mers by (k-1)-mer prefix
  uint32_t words_cap;   // packed-target words (even)
  uint32_t fcap;        // stack frames (even) >= max_stack + 2
  uint32_t bcap;        // branch frames >= max_break + 1
  uint32_t dbg;         // diagnostic ablation flags (KM_DEBUG_FLAGS): 1 = skip k_dfs work
  uint32_t spec;        // k_dfs: look a chain up along the target, one predicted step per lane (KM_SPECULATE=0: off; same results)
  unsigned char* g_ws;  // BIG only
  uint64_t g_stride;    // BIG only: bytes per block
  unsigned char* f_ws;  // fast tier: global scratch for the DFS stack frames (per block)
  uint64_t f_stride;
  unsigned long long* stamps;  // diagnostics (KM_SEED_STAMPS): 16 words per k_seed wave, else null
};

// Per-target k_dfs state.  BIG tier: everything in one global block.  Fast tier: node
// set, packed target and branch frames in LDS (walk_lds_bytes), stack frames in a
// global scratch (walk_frame_bytes) — frames are written on every push but read back
// only on a rejoin or an unwind.
__host__ __device__ inline uint64_t walk_frame_bytes(uint32_t fcap) {
  return (((uint64_t)fcap * 16) + 15) & ~15ull;       // k-mer 8 + count 4 + set slot 4
}
// pos_elt: bytes per entry of the position table (2 in the LDS tier, 4 in the large one)
__host__ __device__ inline uint64_t walk_lds_bytes(uint32_t hs_cap, uint32_t words_cap, uint32_t bcap, uint32_t pcap,
                                                   uint32_t pos_elt) {
  uint64_t b = 0;
  b += (uint64_t)hs_cap * 8;       // keys
  b += (uint64_t)words_cap * 8;    // packed target
  b += (uint64_t)bcap * 32;        // branch frames
  b += (uint64_t)hs_cap * 2;       // slot states + node indices (slot_meta)
  b += (uint64_t)pcap * pos_elt;   // position table
  return (b + 15) & ~15ull;
}
__host__ __device__ inline uint64_t walk_ws_bytes(uint32_t hs_cap, uint32_t words_cap,
                                                  uint32_t fcap, uint32_t bcap, uint32_t pcap) {
  return walk_lds_bytes(hs_cap, words_cap, bcap, pcap, 4) + walk_frame_bytes(fcap);
}

// Home slot in an LDS key set.  A cheap 32-bit mix (two multiplies) instead of a 64-bit
// finaliser: these sets are built and probed hundreds of times per target, and every
// instruction of k_dfs sits on one wave's critical path.
__device__ inline uint32_t set_home(uint64_t key, uint32_t cap) {
  uint32_t h = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x85EBCA6Bu);
  h *= 0x9E3779B1u;
  h ^= h >> 15;
  return __umulhi(h, cap);
}

// Per-lane insert (distinct lanes may insert concurrently).  Returns the slot;
// *was_new tells whether this call created the entry.
// SH = 2 (k_dfs's node set): the home slot follows from the k-mer's (k-1)-mer PREFIX, so that the k-mers
// sharing a prefix share one probe sequence — who shares a prefix is then read off one cluster.
template <int SH = 0>
__device__ inline int set_insert_lane(uint64_t* keys, uint32_t cap, uint64_t key, bool* was_new) {
  uint32_t s = set_home(key >> SH, cap);
  for (uint32_t step = 0; step < cap; ++step) {
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[s]),
                                       (unsigned long long)EMPTY, (unsigned long long)key);
    if (old == EMPTY) { *was_new = true; return (int)s; }
    if (old == key) { *was_new = false; return (int)s; }
    if (++s == cap) s = 0;
  }
  *was_new = false;
  return -1;
}

// Per-lane lookup (every lane its own key, no writers meanwhile): the slot of the key (*found)
// or the empty slot that ends its probe sequence; -1 if the set is full of other keys.
template <int SH = 0>
__device__ inline int set_lookup_lane(const uint64_t* keys, uint32_t cap, uint64_t key, bool* found) {
  uint32_t s = set_home(key >> SH, cap);
  for (uint32_t step = 0; step < cap; ++step) {
    const uint64_t kv = keys[s];
    if (kv == key) { *found = true; return (int)s; }
    if (kv == EMPTY) { *found = false; return (int)s; }
    if (++s == cap) s = 0;
  }
  *found = false;
  return -1;
}

// Wave-cooperative lookup (all 64 lanes call with the same key).  Returns the
// slot of the key (*found) or the empty slot where it would be inserted.
template <int SH = 0>
__device__ inline int set_find(const uint64_t* keys, uint32_t cap, uint64_t key, bool* found) {
  const uint32_t lane = (uint32_t)lane_id();
  uint32_t base = set_home(key >> SH, cap);
  for (uint32_t scanned = 0; scanned < cap + 64; scanned += 64) {
    uint32_t s = base + lane;
    if (s >= cap) s -= cap;
    const uint64_t kv = keys[s];
    const unsigned long long mm = __ballot(kv == key);
    const unsigned long long me = __ballot(kv == EMPTY);
    if (mm | me) {
      const int lm = mm ? (__ffsll((long long)mm) - 1) : 64;
      const int le = me ? (__ffsll((long long)me) - 1) : 64;
      uint32_t pos = base + (uint32_t)(lm < le ? lm : le);
      if (pos >= cap) pos -= cap;
      *found = lm < le;
      return (int)pos;
    }
    base += 64;
    if (base >= cap) base -= cap;
  }
  *found = false;
  return -1;
}

// ---------------------------------------------------------------------------- k_pack
// ASCII bases -> 2-bit words (32 bases per uint64_t, first base most significant),
// one extra zero word per target; per-target initial state.
constexpr uint32_t PACK_WAVES = 4;      // targets per k_pack block (one wave each): 4x fewer, fatter blocks
__global__ __launch_bounds__(64 * PACK_WAVES) void k_pack(WalkArgs a) {
  // a target's words are also kept in LDS (targets of up to PACK_LDS_WORDS * 32 bases): its item
  // records are then built from there instead of reading the words back from memory
  constexpr uint32_t PACK_LDS_WORDS = 192;
  __shared__ uint64_t lds_words[PACK_WAVES][PACK_LDS_WORDS];
  uint64_t* const mine = lds_words[threadIdx.x >> 6];
  const uint32_t t = blockIdx.x * PACK_WAVES + (threadIdx.x >> 6);
  const uint32_t lane = (uint32_t)lane_id();
  if (t >= a.n_targets) return;          // whole waves only; nothing below synchronises across waves
  // every per-target header word is requested here, together (three of them used to be loaded behind the packing
  // loop and behind its fences: four dependent round trips per wave instead of two)
  const uint64_t off = a.toff[t];
  const uint64_t L = a.toff[t + 1] - off;
  const uint64_t wo = a.woff[t];
  const uint64_t fwo = a.fw_off[t];
  const uint64_t nbase = a.node_base0[t];
  const uint64_t item0 = a.item_off[t];
  const uint32_t nwords = (uint32_t)((L + 31) >> 5);
  const uint32_t n_ref = (L >= (uint64_t)a.tab.k) ? (uint32_t)(L - a.tab.k + 1) : 0;
  uint32_t bad = 0;
  // 256 bases per round: a lane packs 4 consecutive bases into one byte, eight neighbouring
  // lanes are OR-ed into one 32-base word (bases past the end count as A = 0, which also
  // writes the extra zero word)
  for (uint32_t c = 0; c * 8 <= nwords; ++c) {
    const uint64_t p = (uint64_t)c * 256 + lane * 4;
    uint32_t byte = 0;
    uint32_t four = 0;                       // the lane's four characters, first one in the low byte
    if (p + 4 <= L) {
      __builtin_memcpy(&four, a.bases + off + p, 4);           // (one load; global memory takes it unaligned)
    } else {
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j)
        if (p + j < L) four |= (uint32_t)a.bases[off + p + j] << (8 * j);
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      uint32_t code = 0;
      if (p + j < L) {
        const uint32_t ch = (four >> (8 * j)) & 0xDFu;          // upper-case
        bad |= (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ? 1u : 0u;
        code = ((ch >> 1) ^ (ch >> 2)) & 3u;
      }
      byte = (byte << 2) | code;
    }
    uint64_t v = (uint64_t)byte << (56 - 8 * (lane & 7));
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    const uint32_t w = c * 8 + (lane >> 3);
    if ((lane & 7) == 0 && w <= nwords) {
      a.packed[wo + w] = v;
      if (w < PACK_LDS_WORDS) mine[w] = v;
    }
  }
  const bool in_lds = nwords < PACK_LDS_WORDS;
  for (uint32_t w = lane; w < (n_ref + 31) / 32; w += 64) a.flagbits[fwo + w] = 0;
  const int any_bad = __any((int)bad);
  uint32_t st = T_OK;
  if (n_ref == 0) st = T_EMPTY;
  else if (any_bad) st = T_BAD_BASE;
  else if (a.max_stack > 0 && n_ref > a.max_node) st = T_NODE_LIMIT;  // first __extend call exits
  // the packed words of this target (written by other lanes of this wave) are visible
  if (in_lds) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // one wave: LDS runs its operations in order
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  {
    const uint32_t n_it = (n_ref + SEED_BLOCK - 1) / SEED_BLOCK;
    uint64_t* rec0 = a.items + 16ull * item0;
    for (uint32_t x = lane; x < n_it * 16; x += 64) {
      const uint32_t q = x >> 4, f = x & 15;
      uint64_t v;
      if (f == 0) v = (uint64_t)t | ((uint64_t)(q * SEED_BLOCK) << 32);
      else if (f == 1) v = (uint64_t)n_ref | ((uint64_t)(st == T_OK ? 1u : 0u) << 32);
      else if (f == 2) v = nbase;
      else if (f == 3) v = fwo | (wo << 32);       // both below 2^32 (km_batch_create checks)
      else {
        const uint32_t w = q * (SEED_BLOCK / 32) + (f - 4);
        v = (w <= nwords) ? (in_lds ? mine[w] : a.packed[wo + w]) : 0ull;
      }
      rec0[x] = v;
    }
  }
  if (lane == 0) {
    a.status[t] = st;
    a.n_ref[t] = n_ref;
    a.n_nodes[t] = (st == T_OK || st == T_NODE_LIMIT) ? n_ref : 0;
    a.probes[t] = 0;
    a.dfs_probes[t] = 0;
    a.fetches[t] = 0;
    a.tflag[t] = 0;
    a.node_base[t] = nbase;                  // (a large-tier pass of the last run may have re-homed the target)
    a.node_cap[t] = n_ref + a.fast_extra;
    a.t_eremoved[t] = 0;
    a.t_enonref[t] = 0;
    if (t == 0) a.loop_ctl[0] = 0;
    if (t == 0 && a.big_ctl) { a.big_ctl[0] = 0; a.big_ctl[1] = 0; }
    if (t == 0) { a.n_flagged[0] = 0; a.n_flagged[1] = 0; a.n_flagged[2] = 0; }   // [1]: targets k_graph_pure hands to k_graph, [2]: those k_dfs's epilogue leaves to it
  }
}

// ---------------------------------------------------------------------------- k_seed
// Diagnostic time stamps of k_seed (only in the STAMPS instantiation): every stamp first
// drains the outstanding loads, so the difference of two stamps is the full latency of the
// stage between them.
#define KM_SEED_STAMP(n)                                                                  \
  do {                                                                                    \
    if constexpr (STAMPS) {                                                               \
      unsigned long long t_;                                                              \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" \
                   : "=s"(t_)::"memory");                                                 \
      ts[n] = t_;                                                                         \
    }                                                                                     \
  } while (0)

// One block per item record (256 consecutive seeds of one target), one lane per seed: its
// count, its get_child and the trivial / flagged decision.
//
// The kernel is bound by instruction issue, not by HBM (rocprof: the SIMDs issue ~75 % of
// the time, 1.7 M line reads per launch), so the work per seed is kept minimal:
//  * minimizers: every lane hashes ONE m-mer (the one starting at its base), the 256 + w keys
//    go to LDS once, and a lane's minimizer is the minimum of the w keys after its own
//    (one LDS round trip: w/2 ds_read2 + w/2 v_min3);
//  * the lookup is key -> directory word -> the aligned home pair, three dependent loads with
//    no loop (device_common.h: the build guarantees the pair).
// FETCHES: count the table slots read per target (km_batch_sizes_t.table_fetches; KM_RUN_COUNT_FETCHES) — two
// ballots and one more atomic per wave, 2 % of the pipelined step, so only on request
template <bool STAMPS, int K, bool FETCHES>
__global__ __launch_bounds__(SEED_BLOCK) void k_seed(WalkArgs a) {
  static_assert(SEED_BLOCK == 256, "the item record holds 12 words = 256 seeds + k - 1 + 1 bases");
  constexpr uint32_t RAW = SEED_BLOCK + 32;                // selection keys of the item's m-mers
  __shared__ uint32_t raw[RAW];
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long real0 = 0;
  uint32_t probes_first = 0;                               // diagnostics: slots read by this lane's lookup
  if constexpr (STAMPS) real0 = __builtin_amdgcn_s_memrealtime();
  KM_SEED_STAMP(0);
  const TableView tab = specialized_view<K>(a.tab);
  const int k = tab.k;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t W = (uint32_t)tab.w;
  const uint64_t* rec = a.items + 16ull * blockIdx.x;
  KM_SEED_STAMP(1);
  // 64 bits of the target starting at item-relative base q
  auto bits_at = [&](uint32_t q) -> uint64_t {
    const uint32_t w = q >> 5, sh = (q & 31) * 2;
    const uint64_t hi = rec[4 + w], lo = rec[5 + w];
    return (hi << sh) | ((lo >> 1) >> (63 - sh));
  };
  const uint64_t x = bits_at(tid);
  KM_SEED_STAMP(2);

  // ---- minimizers.  raw[q] = selection key of the m-mer starting at base q; the (k-1)-mer
  // starting at base q has the windows q .. q+w-1.  This lane needs X[1:] (base tid + 1).
  raw[tid] = mmer_scan_key(tab, x, tid);
  if (tid < W) raw[SEED_BLOCK + tid] = mmer_scan_key(tab, bits_at(SEED_BLOCK + tid), SEED_BLOCK + tid);
  // the record header is only needed from here on: its load overlapped the bases above
  const uint64_t h0 = rec[0], h1 = rec[1];
  const uint64_t nb = rec[2], fwo = rec[3] & 0xFFFFFFFFull;
  const uint32_t wo32 = (uint32_t)(rec[3] >> 32);
  if ((uint32_t)(h1 >> 32) == 0u) return;                  // target not walkable (status != OK); block-uniform
  const uint32_t t = (uint32_t)h0;
  const uint32_t n_ref = (uint32_t)h1;
  const uint32_t i = (uint32_t)(h0 >> 32) + tid;           // the first seed of an item is a multiple of 256
  __syncthreads();
  uint32_t m_rest;                                         // min over bases tid+1 .. tid+w-1
  uint32_t m_last;                                         // key of base tid+w
  if (W == 16) {
    const uint32_t* r = raw + tid + 1;
    const uint32_t m0 = min(min(r[0], r[1]), r[2]), m1 = min(min(r[3], r[4]), r[5]);
    const uint32_t m2 = min(min(r[6], r[7]), r[8]), m3 = min(min(r[9], r[10]), r[11]);
    const uint32_t m4 = min(min(r[12], r[13]), r[14]);
    m_rest = min(min(min(m0, m1), m2), min(m3, m4));
    m_last = r[15];
  } else {
    m_rest = ~0u;
    for (uint32_t j = 1; j < W; ++j) m_rest = min(m_rest, raw[tid + j]);
    m_last = raw[tid + W];
  }
  const uint32_t u_child = (min(m_rest, m_last) & SEL_POS) - (tid + 1);
  KM_SEED_STAMP(3);

  // ---- key, directory word, home pair.  Thread 0 of a target's first item also needs
  // query(ref[0]) (its (k-1)-mer prefix starts at base 0): one more independent chain.
  const uint64_t X = x >> (64 - 2 * k);
  const Key g = key_from_window(tab, X & tab.pmask, u_child);
  const DirPair dw = *reinterpret_cast<const DirPair*>(tab.dir + g.bucket);
  const bool head = i == 0;
  Key gq = g;
  DirPair dq = dw;
  if (head) {
    const uint32_t u_first = min(raw[0], m_rest) & SEL_POS;
    gq = key_from_window(tab, X >> 2, u_first);
    dq = *reinterpret_cast<const DirPair*>(tab.dir + gq.bucket);
  }
  KM_SEED_STAMP(4);
  const uint32_t S = (uint32_t)bucket_slots(dw.lo, dw.hi);
  const Slot* base = tab.slots + 2ull * dw.lo;
  const uint32_t pos = (uint32_t)home_slot(tab, g, S);     // 0 for an empty bucket
  // loaded unconditionally (an empty bucket reads some valid slots and ignores them)
  const Slot* b0 = (S ? base : tab.slots) + pos;
  const uint4 first = *reinterpret_cast<const uint4*>(b0);
  const uint4 second = *reinterpret_cast<const uint4*>(b0 + 1);
  uint32_t Sq = 0, posq = 0;
  const Slot* baseq = tab.slots;
  uint4 firstq = first, secondq = second;
  if (head) {
    Sq = (uint32_t)bucket_slots(dq.lo, dq.hi);
    baseq = tab.slots + 2ull * dq.lo;
    posq = (uint32_t)home_slot(tab, gq, Sq);
    const Slot* bq = (Sq ? baseq : tab.slots) + posq;
    firstq = *reinterpret_cast<const uint4*>(bq);
    secondq = *reinterpret_cast<const uint4*>(bq + 1);
  }
  KM_SEED_STAMP(5);

  uint32_t fetch_l = 0;
  bool valid = false, triv = false, triv_child = false;
  if (i < n_ref) {
    valid = true;
    uint4 c4 = make_uint4(0, 0, 0, 0);
    if (S) c4 = bucket_resolve2(tab, g, base, S, pos, first, second, &fetch_l);
    c4 = finish_children(tab, X, g.flip, c4);
    KM_SEED_STAMP(6);
    if constexpr (STAMPS) probes_first = fetch_l;
    // (the k-mer itself is not stored: every consumer reads the target's own k-mers from its
    // packed words; node_kmer[] holds walk-discovered nodes only)
    uint32_t nextb = 4;
    if (i + 1 < n_ref) {
      // last base of ref[i+1] = base k of this lane's 32-base window (k <= 31), else reloaded
      if (k < 32) {
        nextb = (uint32_t)(x >> (62 - 2 * k)) & 3u;
      } else {
        const uint32_t p = tid + (uint32_t)k;
        nextb = (uint32_t)(rec[4 + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u;
      }
      a.node_cnt[nb + i + 1] = pick4(c4, nextb);
    }
    if (head) {                                            // node_data[ref[0]] = jf.query(ref[0])
      uint4 cq = make_uint4(0, 0, 0, 0);
      if (Sq) cq = bucket_resolve2(tab, gq, baseq, Sq, posq, firstq, secondq, &fetch_l);
      const uint32_t sb = (uint32_t)(X & 3);
      uint32_t v = pick4(cq, gq.flip ? 3 - sb : sb);
      if (v == COUNT_ESCAPE) v = overflow_count(tab, X);
      a.node_cnt[nb] = v;
    }
    if (a.max_stack > 0) {
      const uint32_t mask = child_mask(c4, a.ratio, a.nc);
      triv = (mask == 0) || (nextb < 4 && mask == (1u << nextb));
      triv_child = triv && mask != 0;
      if (!triv) {
        atomicOr(&a.flagbits[fwo + (i >> 5)], 1u << (i & 31));
        if (atomicExch(&a.tflag[t], 1u) == 0u) {
          const uint32_t at = atomicAdd(a.n_flagged, 1u);
          a.flagged[at] = t;
          a.flag_rec[2 * at] = make_uint4(t, n_ref, (uint32_t)nb, (uint32_t)(nb >> 32));
          a.flag_rec[2 * at + 1] = make_uint4((uint32_t)fwo, wo32, 0u, 0u);
        }
      }
    }
  }
  // one atomic per wave for the per-target counters.  Logical probes per seed:
  // node_data[s] = jf.query(s) (1); for a trivial seed also get_child (4) and the
  // re-query of [seed] when it has a child (1).  Slots read: 1 or 2 per lookup, more only in
  // a table whose build gave up the pair bound.
  const unsigned long long probes_w = (unsigned long long)__popcll(__ballot(valid)) +
                                      4ull * __popcll(__ballot(triv)) + __popcll(__ballot(triv_child));
  unsigned long long fetch_w = 0;
  if constexpr (FETCHES || STAMPS) {
    fetch_w = (unsigned long long)__popcll(__ballot(fetch_l >= 1)) + __popcll(__ballot(fetch_l >= 2));
    if (__any(fetch_l > 2)) {
      uint32_t extra = fetch_l > 2 ? fetch_l - 2 : 0;
      for (int o = 32; o > 0; o >>= 1) extra += __shfl_xor(extra, o);
      fetch_w += extra;
    }
  }
  if (lane == 0 && probes_w) {
    atomicAdd(&a.probes[t], probes_w);
    if constexpr (FETCHES || STAMPS) atomicAdd(&a.fetches[t], fetch_w);
  }
  if constexpr (STAMPS) {
    KM_SEED_STAMP(7);
    const unsigned long long real1 = __builtin_amdgcn_s_memrealtime();
    uint32_t pmax = probes_first, psum = probes_first;
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
      const uint32_t v = __shfl_xor(pmax, o2);
      pmax = v > pmax ? v : pmax;
      psum += __shfl_xor(psum, o2);
    }
    if (lane == 0) {
      unsigned long long* o = a.stamps + 16ull * (blockIdx.x * (SEED_BLOCK / 64) + (tid >> 6));
      for (int q = 0; q < 8; ++q) o[q] = ts[q];
      o[8] = real0; o[9] = real1;
      o[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: wave, SIMD, CU, SH, SE
      o[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
      o[12] = pmax; o[13] = psum;
    }
  }
}

// ---------------------------------------------------------------------------- k_dfs
// Diagnostics: -DKM_DFS_COUNTERS (tools/dfs_lifetimes.py with DFS_COUNTERS=1) counts and times, per wave, what the
// walk did (the KM_DC / KM_DT macros below; nothing in the product build).  KM_SEED_STAMPS in the environment records
// when a wave started and ended its phases (s_memrealtime, 100 MHz) in the normal build.
// (the register allocator is told how many waves a SIMD is to hold: 512 / 4 = 128 vector registers)
#ifndef KM_DFS_WAVES_PER_EU
#define KM_DFS_WAVES_PER_EU 4
#endif
template <bool BIG, int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KM_DFS_WAVES_PER_EU))) void k_dfs(WalkArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t lane = (uint32_t)lane_id();
  // diagnostics (KM_SEED_STAMPS in the environment): when this wave started and ended, 100 MHz clock
  unsigned long long life0 = 0, life1 = 0, life2 = 0, lifeB = 0, lifeC = 0;
  if (a.stamps) life0 = __builtin_amdgcn_s_memrealtime();
#ifdef KM_DFS_COUNTERS
  uint32_t dc_nonres = 0, dt_nonres = 0, dc_bigS = 0;
  uint32_t dt_post = 0, dt_pre = 0, dt_chain = 0, dt_tail = 0, dt_pop = 0, dt_seed0 = 0;
  uint32_t dc_slow = 0, dc_spec = 0, dc_rec = 0, dc_bload = 0, dc_gen = 0, dc_runs = 0, dc_full = 0, dc_v0 = 0, dc_steps = 0, dc_noalign = 0;
  uint32_t dt_spec = 0, dt_book = 0, dt_gen = 0, dt_bload = 0, dt_rejoin = 0, dt_unwind = 0, dt_align = 0, dt_t0 = 0;   // 10 ns units
#define KM_DC(x) (++(x))
#define KM_DCN(x, v) ((x) += (v))
#define KM_DT0() (dt_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime())
#define KM_DT(x) do { const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memrealtime(); (x) += n_ - dt_t0; dt_t0 = n_; } while (0)
#else
#define KM_DC(x) do {} while (0)
#define KM_DCN(x, v) do {} while (0)
#define KM_DT0() do {} while (0)
#define KM_DT(x) do {} while (0)
#endif
  const TableView tab = specialized_view<K>(a.tab);
  const int k = tab.k;
  // the target: fast tier — entry blockIdx.x of k_seed's list of flagged targets, with what the
  // walk needs of it in one record (requested together with the length of the list); large tier —
  // a list of target ids from the host
  uint32_t t, n_ref, node_cap;
  uint64_t nb, wo, fwo;
  if constexpr (BIG) {
    const uint32_t n_list = a.n_list_dev ? *a.n_list_dev : a.n_list_host;
    if (blockIdx.x >= n_list) return;
    if (a.big_prep && blockIdx.x >= a.big_slots) return;
    t = a.list[blockIdx.x];
    if (a.status[t] != T_OK && a.status[t] != T_NEEDS_BIG) return;
    n_ref = a.n_ref[t];
    nb = a.node_base[t];
    node_cap = a.node_cap[t];
    // (whatever loop breaks the fast tier's abandoned attempt at this target logged are void: a marker says so)
    if (lane == 0 && a.status[t] == T_NEEDS_BIG) {
      const uint32_t at = atomicAdd(&a.loop_ctl[0], 1u);
      if (at < a.loop_cap) { a.loop_list[2 * at] = t; a.loop_list[2 * at + 1] = 0xFFFFFFFFu; }
    }
    if (a.big_prep) {
      // the device's own large tier: slot blockIdx.x of the region; the seed kernel's counts move there
      const uint64_t need = (uint64_t)max(n_ref, a.max_node + a.max_stack) + 1;
      if (need > (uint64_t)a.big_entry || a.status[t] != T_NEEDS_BIG) return;          // (left to the host)
      const uint64_t nb_new = a.big_region + (uint64_t)blockIdx.x * a.big_entry;
      for (uint32_t j = lane; j < n_ref; j += 64) a.node_cnt[nb_new + j] = a.node_cnt[nb + j];
      if (lane == 0) { a.node_base[t] = nb_new; a.node_cap[t] = a.big_entry; }
      nb = nb_new;
      node_cap = a.big_entry;
      __syncthreads();
    }
    wo = a.woff[t];
    fwo = a.fw_off[t];
  } else {
    const uint4 r0 = a.flag_rec[2 * blockIdx.x], r1 = a.flag_rec[2 * blockIdx.x + 1];
    const uint32_t n_list = *a.n_list_dev;
    // The grid follows the number of flagged targets the batch's last delivery reported (kmgpu.hip: launch_dfs), not
    // the batch size.  Should more be flagged than there are blocks, block 0 hands the rest to the large tier, as
    // any target is that this tier cannot hold (the next run's grid is larger).
    if (blockIdx.x == 0 && n_list > gridDim.x) {
      for (uint32_t e = gridDim.x + lane; e < n_list; e += 64) {
        const uint32_t te = a.flag_rec[2 * e].x;
        a.status[te] = T_NEEDS_BIG;
        if (a.epi != nullptr) { a.epi->t_refmax[te] = NOT_BARE; a.epi->left[atomicAdd(a.epi->n_left, 1u)] = te; }
        if (a.big_ctl) { const uint32_t at = atomicAdd(&a.big_ctl[0], 1u); if (at < a.big_slots) a.big_walk[at] = te; }
      }
    }
    if (blockIdx.x >= n_list) return;
    t = r0.x; n_ref = r0.y; nb = ((uint64_t)r0.w << 32) | r0.z;
    fwo = r1.x; wo = r1.y;
    node_cap = n_ref + a.fast_extra;
  }

  unsigned char* wsb;
  unsigned char* fwb;
  if constexpr (BIG) {
    wsb = a.g_ws + (uint64_t)blockIdx.x * a.g_stride;
    fwb = wsb + walk_lds_bytes(a.hs_cap, a.words_cap, a.bcap, a.pcap, 4);
  } else {
    wsb = smem;
    fwb = a.f_ws + (uint64_t)blockIdx.x * a.f_stride;
  }
  const uint32_t cap = a.hs_cap;
  uint64_t* keys = reinterpret_cast<uint64_t*>(wsb);
  uint64_t* words = keys + cap;
  BranchFrame* bf = reinterpret_cast<BranchFrame*>(words + a.words_cap);
  uint16_t* state = reinterpret_cast<uint16_t*>(bf + a.bcap);
  // The target's own k-mers are NOT entered into the node set: a table of POSITIONS hashed by (k-1)-mer prefix,
  // written with plain stores (whoever stores last owns a slot), answers "is y the target's k-mer i" by reading the
  // slot and comparing y with the packed target at that position.  The k-mers that lost their slot (a fifth of
  // them at load 1/2) go into the node set as before.  Set-up was the compare-and-swap of all of a target's k-mers —
  // bound by the LDS atomic rate of a CU whose waves build their sets at the same time — and the 64-bit keys of
  // those k-mers were three quarters of a wave's LDS.
  using pos_t = typename std::conditional<BIG, uint32_t, uint16_t>::type;
  constexpr uint32_t POS_NONE = (uint32_t)(pos_t)~(pos_t)0, NO_NODE = 0xFFFFFFFFu;
  pos_t* pos = reinterpret_cast<pos_t*>(state + cap);
  const uint32_t pcap = a.pcap;
  // its hash looks at the low 32 bits of the prefix only (the last 16 bases: one multiply, and the set-up reads them
  // straight off the packed target with 32-bit operations); pcap is a power of two
  const uint32_t pshift = 32u - (uint32_t)__ffs((int)pcap) + 1u;
  auto pos_home = [&](uint32_t p32) -> uint32_t { return (p32 * 0x9E3779B1u) >> pshift; };
  uint64_t* fk = reinterpret_cast<uint64_t*>(fwb);
  uint32_t* fc = reinterpret_cast<uint32_t*>(fk + a.fcap);
  uint32_t* fs = fc + a.fcap;
  // Single-wave workgroup.  step_sync: orders LDS traffic between lanes (the node set,
  // states, branch frames) — LDS executes a wave's operations in order, so only the
  // compiler must be held back.  mem_sync: additionally makes lane 0's frame stores
  // (global memory) visible before other lanes read them.
  auto step_sync = [&]() {
    if constexpr (BIG) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto mem_sync = [&]() { __syncthreads(); };

  // a flagged target is walkable: n_ref >= 1 and L = n_ref + k - 1 bases
  const uint32_t nwords = (uint32_t)(((uint64_t)n_ref + (uint32_t)k - 1 + 31) >> 5);
  const uint32_t nflag = (n_ref + 31) >> 5;
  const uint32_t* flag = a.flagbits + fwo;
  if (nwords + 1 > a.words_cap || n_ref > node_cap || (uint64_t)n_ref * 4 > (uint64_t)pcap * 3 || n_ref >= POS_NONE) {
    if (lane == 0) {
      a.status[t] = BIG ? T_INTERNAL : T_NEEDS_BIG;
      if constexpr (!BIG) {
        // (with the epilogue on, a flagged target's t_refmax is this kernel's to write — k_graph_pure leaves it
        // alone — and k_graph finds the target through the list of what the epilogue left)
        if (a.epi != nullptr) { a.epi->t_refmax[t] = NOT_BARE; a.epi->left[atomicAdd(a.epi->n_left, 1u)] = t; }
        if (a.big_ctl) { const uint32_t at = atomicAdd(&a.big_ctl[0], 1u); if (at < a.big_slots) a.big_walk[at] = t; }
      }
    }
    return;
  }
  if (a.dbg & 1u) return;

  // ---- per-target state: packed target + node set of the target's k-mers ------------
  // the flag words are requested now (lane w: word w of each chunk of 64) and looked at after the set is built
  uint32_t flag0 = 0;
  if (lane < nflag) flag0 = flag[lane];
  for (uint32_t w = lane; w <= nwords; w += 64) words[w] = a.packed[wo + w];
  __syncthreads();                                           // the packed words
  if (a.stamps) lifeB = __builtin_amdgcn_s_memrealtime();
  if constexpr (BIG) {
    for (uint32_t s = lane; s < cap; s += 64) { keys[s] = EMPTY; state[s] = 0; }
    for (uint32_t s = lane; s < pcap; s += 64) pos[s] = (pos_t)POS_NONE;
  } else {
    // LDS: 16 bytes per store (cap and pcap are multiples of 64: whole uint4s in all three arrays)
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    uint4* kq = reinterpret_cast<uint4*>(keys);
    for (uint32_t s = lane; s < cap / 2; s += 64) kq[s] = ones;
    uint4* sq = reinterpret_cast<uint4*>(state);
    for (uint32_t s = lane; s < cap / 8; s += 64) sq[s] = make_uint4(0u, 0u, 0u, 0u);
    uint4* pq = reinterpret_cast<uint4*>(pos);
    for (uint32_t s = lane; s < pcap / 8; s += 64) pq[s] = ones;
  }
  __syncthreads();
  auto kmer_at = [&](uint32_t i) -> uint64_t {
    const uint32_t w = i >> 5, sh = (i & 31) * 2;
    const uint64_t hi = words[w], lo = words[w + 1];
    const uint64_t x = sh ? ((hi << sh) | (lo >> (64 - sh))) : hi;
    return x >> (64 - 2 * k);
  };
  // low 32 bits of the (k-1)-mer prefix of the target's k-mer i: bases i + k - 17 .. i + k - 2, read as 32-bit pieces
  // of the packed words (piece j = bases 16 j .. 16 j + 15 is element j ^ 1 of the little-endian 64-bit words)
  auto prefix32_at = [&](uint32_t i) -> uint32_t {
    if (k >= 17) {
      const uint32_t b0 = i + (uint32_t)k - 17u, j = b0 >> 4, sh = (b0 & 15u) * 2u;
      const uint32_t* w32 = reinterpret_cast<const uint32_t*>(words);
      const uint32_t hi = w32[j ^ 1u], lo = w32[(j + 1u) ^ 1u];
      return sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;
    }
    return (uint32_t)(kmer_at(i) >> 2);
  };
  // the index of the target k-mer y, if y is one that owns its slot of the position table (the others are in the node set)
  // (branch-free: an empty slot compares against the target's first k-mer and is discarded — callers that look several
  // k-mers up side by side get their LDS reads in flight together)
  auto ref_index = [&](uint64_t y) -> uint32_t {
    const uint32_t p_ = (uint32_t)pos[pos_home((uint32_t)(y >> 2))];
    const bool some = p_ != POS_NONE;
    return (kmer_at(some ? p_ : 0u) == y && some) ? p_ : NO_NODE;
  };
  // y is probably a node (a target k-mer, or a k-mer sitting in its home slot of the node set): a hint
  auto node_hint = [&](uint64_t y) -> bool { return ref_index(y) != NO_NODE || keys[set_home(y >> 2, cap)] == y; };
  // ---- for the speculation of the chain runs (below): where do x's last SPEC_ELL bases occur in the target?  Lane l
  // keeps the SPEC_ELL-mers starting at positions l, l + 64, ... (16 bits each, two per register: the first
  // 64 * 2 * SPEC_PK positions of the target; a longer target is simply not searched beyond them)
  constexpr uint32_t SPEC_ELL = 8, SPEC_PK = 4;
  const uint32_t Lb = n_ref + (uint32_t)k - 1;               // bases of the target
  auto win64 = [&](uint32_t p) -> uint64_t {                 // 32 bases from target position p (zeros past the end)
    const uint32_t w = p >> 5, sh = (p & 31) * 2;
    const uint64_t hi = words[w], lo = words[w + 1];
    return sh ? ((hi << sh) | (lo >> (64 - sh))) : hi;
  };
  const bool spec_ell_ok = a.spec != 0 && (uint32_t)k > 2 * SPEC_ELL && Lb > SPEC_ELL;
  uint32_t ell_pk[SPEC_PK];
#pragma unroll
  for (uint32_t q = 0; q < SPEC_PK; ++q) ell_pk[q] = 0;
  if (spec_ell_ok) {
#pragma unroll
    for (uint32_t q = 0; q < 2 * SPEC_PK; ++q) {
      const uint32_t p = lane + 64u * q;
      uint32_t v = 0;
      if (p + SPEC_ELL < Lb) v = (uint32_t)(win64(p) >> (64 - 2 * SPEC_ELL));
      ell_pk[q >> 1] |= (q & 1u) ? (v << 16) : v;
    }
  }
  if (a.stamps) lifeC = __builtin_amdgcn_s_memrealtime();
  uint32_t dup = 0;
  // (R1) of the epilogue below — no two of the target's k-mers share their (k-1)-mer prefix, and the
  // last one's suffix is nobody's prefix — is read off this very build: the set hashes a k-mer by its
  // prefix, so k-mers sharing one follow the same probe sequence and the later one meets the earlier.
  uint32_t shared = 0;
  // Pass 1: every target k-mer stores its index at the slot of its prefix (plain stores; the last one wins).
  // Pass 2: whoever does not read its own index back has lost the slot — to a k-mer with another prefix, or to
  // one sharing its prefix (then the graph is not the plain chain: `shared`) or to its own twin (`dup`) — and is
  // entered into the node set with compare-and-swap, as every target k-mer used to be.
  uint32_t n_losers = 0;                                     // wave-uniform
#pragma unroll 2
  for (uint32_t i = lane; i < n_ref; i += 64) pos[pos_home(prefix32_at(i))] = (pos_t)i;
  __syncthreads();
  for (uint32_t base = 0; base < n_ref; base += 64u * 32u) {       // (wave-uniform; one trip for targets of up to 2 048 k-mers)
    // who lost its slot: up to 32 k-mers per lane, their reads of the table in flight together
    uint32_t lost_bits = 0;
#pragma unroll 8
    for (uint32_t q = 0; q < 32; ++q) {
      const uint32_t i = base + 64u * q + lane;
      if (base + 64u * q >= n_ref) break;                             // wave-uniform
      if (i < n_ref && (uint32_t)pos[pos_home(prefix32_at(i))] != i) lost_bits |= 1u << q;
    }
    // the losers, one after the other per lane (1.5 on average): compared with the owner of their slot, then entered
    for (uint32_t bits = lost_bits; __any((int)(bits != 0)); bits &= bits - 1) {
      const bool mine = bits != 0;
      n_losers += (uint32_t)__popcll(__ballot(mine));
      if (mine) {
        const uint32_t i = base + 64u * ((uint32_t)__ffs((int)bits) - 1) + lane;
        const uint64_t kk = kmer_at(i);
        const uint64_t xw = kmer_at((uint32_t)pos[pos_home((uint32_t)(kk >> 2))]);   // (the slot was written: by this k-mer, if by no other)
        if (xw == kk) dup = 1;
        else if ((xw >> 2) == (kk >> 2)) shared = 1;
        uint32_t sl = set_home(kk >> 2, cap);
        for (uint32_t step = 0; step < cap && !dup; ++step) {
          const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[sl]), (unsigned long long)EMPTY,
                                                   (unsigned long long)kk);
          if (old == EMPTY) { state[sl] = slot_meta(ST_NODE, i); break; }
          if (old == kk) { dup = 1; break; }
          if ((old >> 2) == (kk >> 2)) shared = 1;
          if (++sl == cap) sl = 0;
        }
      }
    }
  }
  __syncthreads();
  bool ref_pure = false;
  if constexpr (!BIG) {
    if (a.epi != nullptr && n_ref >= 2) {
      if (lane == 0) {                                       // whoever has the last suffix as its prefix owns its slot of the
        const uint64_t S = kmer_at(n_ref - 1) & tab.pmask;   // position table or sits in its cluster of the node set
        const uint32_t w = (uint32_t)pos[pos_home((uint32_t)S)];
        if (w != POS_NONE && (kmer_at(w) >> 2) == S) shared = 1;
        uint32_t sl = set_home(S, cap);
        for (uint32_t step = 0; step < cap; ++step) {
          const uint64_t kv = keys[sl];
          if (kv == EMPTY) break;
          if ((kv >> 2) == S) { shared = 1; break; }
          if (++sl == cap) sl = 0;
        }
      }
      ref_pure = !__any((int)shared);
    }
  }

  uint32_t st = T_OK;
  uint32_t n_nodes = n_ref;
  DirCache dcache = {~0u, 0u, 0u};
  PendingLookup pend;
  pend.valid = false;
  // the bucket the chain runs hold in the lanes outlives a run: the next one mostly starts where
  // this one ended (its rejoin hints may then be stale — they are hints)
  BucketLanes bl;
  bl.valid = false; bl.resident = false; bl.bucket = 0; bl.S = 0;
#pragma unroll
  for (uint32_t i = 0; i < BUCKET_LANES_SETS; ++i) { bl.tag[i] = EMPTY; bl.info[i] = 0; }
  const ChildRule rule = {a.ratio, a.nc, a.thr_below, a.thr_T};
  uint64_t probes_u = 0;     // wave-uniform
  uint32_t fetch_u = 0;
  if (__any((int)dup)) st = T_REPEAT;

  if (a.stamps) life1 = __builtin_amdgcn_s_memrealtime();
  // ---- exact DFS from every flagged seed, in target order ----------------------------
  if (st == T_OK) {
    uint32_t set_count = n_losers;
    const uint32_t set_limit = (uint32_t)(((uint64_t)cap * 3) >> 2);
    // (a target whose k-mers crowd the position table — few distinct prefixes — leaves no room for the walk here)
    if (set_count + 8 > set_limit) st = BIG ? T_INTERNAL : T_NEEDS_BIG;
    uint64_t steps = 0;
    for (uint32_t w = 0; w < nflag && st == T_OK; ++w) {
      // (word w of the flag bitmap: lane w & 63 of its chunk of 64 words holds it)
      if (w >= 64 && (w & 63) == 0) flag0 = (w + lane < nflag) ? flag[w + lane] : 0u;
      uint32_t bits = lane_u32(flag0, w & 63);
      while (bits && st == T_OK) {
        const uint32_t b = (uint32_t)__ffs((int)bits) - 1;
        bits &= bits - 1;
        const uint32_t i = w * 32 + b;
        KM_DT0();
        uint64_t cur = kmer_at(i);
        // (the seed is a node for good: its frame's set slot is never looked at)
        if (lane == 0) { fk[0] = cur; fc[0] = 0; fs[0] = 0u; }
        uint32_t depth = 1, reg = 1, bsp = 0, parent_brk = 0, mask = 0, brk = 0;
        uint4 c4 = make_uint4(0, 0, 0, 0);
        bool need_expand = true;
        step_sync();
        KM_DT(dt_seed0);
        while (true) {
          if (++steps > DFS_STEP_LIMIT) { st = T_INTERNAL; break; }
          if (need_expand) {
            need_expand = false;
            KM_DT0();
            if (n_nodes > a.max_node) { st = T_NODE_LIMIT; break; }
            if (!(pend.valid && pend.X == cur)) children_issue_wave(tab, cur, &dcache, &pend);
            c4 = children_finish_wave(tab, pend, &fetch_u);
            pend.valid = false;
            probes_u += 4;
            KM_DC(dc_gen);
            KM_DT(dt_gen);
            mask = child_mask(c4, a.ratio, a.nc);
            brk = parent_brk;
            if (__popc(mask) > 1) {
              ++brk;
              if (brk > a.max_break) mask = 0;
            }
            // the walk most likely continues with the first kept child: request its lookup
            // now, its latency overlaps the bookkeeping of this step
            // (unless that child is a k-mer of the target — for a seed it mostly is, its successor on the target: a
            // rejoin, never expanded; the chain that follows is looked up by the run's own means)
            const uint64_t first_child = ((cur << 2) | ((uint32_t)__ffs((int)mask) - 1)) & tab.kmask;
            if (mask && depth + 1 <= a.max_stack &&
                !__builtin_amdgcn_readfirstlane((int)(ref_index(first_child) != NO_NODE))) {
              children_issue_wave(tab, first_child, &dcache, &pend);
            }
            KM_DT(dt_post);
          }
          // ---- chain run.  cur has exactly one child left to take.  As long as that keeps
          // being so, only the lookups depend on each other; the node-set probe, the insertion
          // and the stack frame of every child on the chain do not, so the chain is walked with
          // the lookups alone (child j of the run recorded in lane j) and booked afterwards by
          // all lanes at once.  The run assumes every child to be new; the booking finds the
          // first one that is not (a rejoin or a loop), keeps what precedes it and hands that
          // child to the general step below, which treats it exactly as before.  What was
          // looked up past it is dropped: lookups have no side effects.
          if (mask != 0 && (mask & (mask - 1)) == 0 && n_nodes <= a.max_node) {
            KM_DT0();
            int64_t room64 = 64;
            room64 = min(room64, (int64_t)a.max_stack - (int64_t)depth);
            room64 = min(room64, (int64_t)a.fcap - (int64_t)depth);
            room64 = min(room64, (int64_t)set_limit - (int64_t)set_count);
            const uint32_t room = room64 > 0 ? (uint32_t)room64 : 0u;
            uint32_t n = 0;
            uint64_t rkey = 0;                             // lane j: child j of the run
            uint32_t rcnt = 0;
            uint64_t x = cur;                              // top of the (virtual) stack
            uint32_t c = (uint32_t)__ffs((int)mask) - 1;   // its one child left, and that child's count
            uint32_t cnt = pick4(c4, c);
            bool expanded = false;                         // (c4, mask) = a finished expansion of x
            // State of a step: x has exactly the child c (count cnt) left to take; child = x[1:] + c;
            // T = tag of the group of child[1:] (what the expansion of child looks up); hint: child
            // sits in its home slot of the node set, i.e. is most likely a rejoin — stop there and
            // let the general step decide (only a hint: misses are caught by the booking below).
            // From the second step on all of it comes precomputed out of the slot that named the
            // child (device_common.h: BucketLanes); only the first step works it out here.
            uint64_t child = ((x << 2) | c) & tab.kmask;
            auto tag_of_suffix = [&](uint64_t ch) -> uint64_t {      // tag of the group of ch[1:]: what the expansion of ch looks up
              const uint64_t P = ch & tab.pmask;
              uint32_t flip_;
              return group_tag(tab, P, revcomp(P, k - 1), &flip_);
            };
            auto slow_state = [&](uint64_t ch, uint64_t* T_out, bool* hint_out) {
              *T_out = tag_of_suffix(ch);
              *hint_out = __builtin_amdgcn_readfirstlane((int)node_hint(ch)) != 0;
            };
            uint64_t T;
            bool hint;
            slow_state(child, &T, &hint);
            KM_DT(dt_pre);
            // the rest of a step — the tag is not among the lanes, or its slot does not name a single
            // child — kept out of the loop's hot path; true: the run goes on, false: (c4, mask) hold the
            // expansion of x and the run ends
            auto step_slow = [&](bool was_hit) -> bool {
              bool have_c4 = false;                        // c4 = the expansion of x (set by the speculation below)
              KM_DC(dc_slow);
              KM_DT0();
              if (!was_hit && a.spec) {
                // ---- speculation along the target.  Behind a variant the walk runs back onto the target: the
                // k-mers of the chain are then x shifted by the target's own bases, known in advance — only
                // whether the table agrees is not.  So the steps are looked up TOGETHER, lane j taking
                // y_j = x + the next j bases of the target (one full lookup per lane, as k_seed does: key,
                // directory word, home pair), and verified in order afterwards: step j stands iff get_child of
                // y_j keeps exactly the predicted base (km/utils/Jellyfish.py:55-72).  What stands is recorded
                // as children of the run, exactly as if it had been walked one lookup after the other; the
                // first y_j that does not follow the prediction is a real k-mer of the chain with its real
                // expansion, and the run goes on from there.  Lookups have no side effects: a wrong alignment
                // costs their latency and nothing else.  Where the target's bases line up with x:
                //  (a) a substitution does not shift the walk against the target: x, s steps behind the seed i,
                //      then ends at target position i + k - 1 + s;
                //  (b) else the last SPEC_ELL bases of x occur in the target (first occurrence).
                constexpr uint32_t SPEC_NONE = 0xFFFFFFFFu;
                uint32_t e = SPEC_NONE;                                  // target position of the base predicted next
                // the k-mer at stack position s is the seed (target k-mer i) shifted by s bases: x, at position
                // depth - 1 + n, ends at target position i + k - 1 + that if the walk has not shifted against the
                // target — at the first k-mer off the target that is the hypothesis, later on x's last bases say
                // (only near the seed: further on, x's last SPEC_ELL bases are looked for anywhere in the target, (b))
                const uint32_t sx = depth - 1 + n;
                if (sx <= 4 || !spec_ell_ok) {
                  const uint32_t e_sub = i + (uint32_t)k + sx;
                  if (e_sub < Lb && e_sub >= (uint32_t)k) {
                    const uint64_t d = x ^ kmer_at(e_sub - (uint32_t)k);
                    const uint32_t agree = d ? ((uint32_t)__ffsll((long long)d) - 1) >> 1 : (uint32_t)k;
                    // (the seed's other child — the k-mer at stack position 1 — ends on the substituted base: the
                    // bases behind it agree, that one does not)
                    if ((sx <= 4 && agree == sx - 1) || agree >= SPEC_ELL) e = e_sub;
                  }
                }
                if (e == SPEC_NONE && spec_ell_ok) {
                  const uint32_t sfx = (uint32_t)x & ((1u << (2 * SPEC_ELL)) - 1);
                  uint32_t hits = 0;                                     // bit q: the SPEC_ELL-mer at lane + 64 q is x's suffix
#pragma unroll
                  for (uint32_t q = 0; q < 2 * SPEC_PK; ++q) {
                    const uint32_t v = (q & 1u) ? (ell_pk[q >> 1] >> 16) : (ell_pk[q >> 1] & 0xFFFFu);
                    if (v == sfx && lane + 64u * q + SPEC_ELL < Lb) hits |= 1u << q;
                  }
                  if (__any((int)(hits != 0))) {
                    for (uint32_t q = 0; q < 2 * SPEC_PK; ++q) {         // first occurrence in target order
                      const unsigned long long mm = __ballot((hits >> q) & 1u);
                      if (mm) { e = 64u * q + (uint32_t)__ffsll((long long)mm) - 1 + SPEC_ELL; break; }
                    }
                  }
                }
                if (e == SPEC_NONE) KM_DC(dc_noalign);
                KM_DT(dt_align);
                if (e != SPEC_NONE) {
                  // how many of x's last bases are the target's bases before e: the chain is back on the target (a
                  // rejoin, surely a node) after k - lam more steps
                  uint32_t lam;
                  if (e >= (uint32_t)k) {
                    const uint64_t d = x ^ kmer_at(e - (uint32_t)k);
                    lam = d ? ((uint32_t)__ffsll((long long)d) - 1) >> 1 : (uint32_t)k;
                  } else {
                    const uint64_t d = (x ^ (win64(0) >> (64 - 2 * e))) & ((1ull << (2 * e)) - 1);   // (1 <= e < k <= 32)
                    lam = d ? ((uint32_t)__ffsll((long long)d) - 1) >> 1 : e;
                  }
                  if (lam >= (uint32_t)k) {
                    // x itself is a k-mer of the target, i.e. a node: the booking ends the run before it
                    c4 = make_uint4(0, 0, 0, 0);
                    mask = 0;
                    expanded = true;
                    return false;
                  }
                  // lanes 0 .. J-1 look up y_0 .. y_{J-1}; y_J is the target's k-mer at e + J - k when J = k - lam
                  const uint32_t J_full = (uint32_t)k - lam;
                  uint32_t J = min(J_full, 31u);
                  J = min(J, Lb - e);
                  J = min(J, room - n + 1);                              // (n <= room here)
                  if (J >= 3) {
                    const uint64_t w64 = win64(e);
                    uint64_t yj = x;
                    if (lane >= 1 && lane < J) yj = ((x << (2 * lane)) | (w64 >> (64 - 2 * lane))) & tab.kmask;
                    const uint32_t pb = lane < J ? (uint32_t)(w64 >> (62 - 2 * lane)) & 3u : 4u;   // (e + lane < Lb)
                    uint32_t fl = 0;
                    const uint4 cj = forward_children(tab, yj, &fl);
                    const uint32_t mj = child_mask(cj, a.ratio, a.nc);
                    const bool single = mj != 0 && (mj & (mj - 1)) == 0;
                    const uint32_t cbj = single ? (uint32_t)__ffs((int)mj) - 1 : 0u;
                    const uint32_t cntj = pick4(cj, cbj);
                    const unsigned long long badm = __ballot(!(single && cbj == pb));   // (pb = 4 from lane J on)
                    const uint32_t V = (uint32_t)__ffsll((long long)badm) - 1;    // steps that stand (<= J)
                    const uint32_t R = V == J ? J - 1 : V;               // children recorded: y_1 .. y_R
                    {
                      const uint64_t ys = (uint64_t)__shfl((unsigned long long)yj, (int)((lane - n + 1) & 63u));
                      const uint32_t cs = (uint32_t)__shfl((int)cntj, (int)((lane - n) & 63u));
                      if (lane >= n && lane < n + R) { rkey = ys; rcnt = cs; }
                    }
                    if (lane >= J) fl = 0;
                    for (int o = 32; o > 0; o >>= 1) fl += __shfl_xor(fl, o);
                    fetch_u += fl;
                    n += R;
                    KM_DC(dc_spec); KM_DCN(dc_rec, R);
                    KM_DT(dt_spec);
                    if (V == 0) KM_DC(dc_v0);
                    if (V == J && J == J_full) KM_DC(dc_full);
                    if (V == J) {
                      // every step stands: y_{J-1} is the top of the run and has exactly the child y_J left —
                      // a k-mer of the target (a node: the run ends before it) unless the lanes ran out first
                      x = lane_u64(yj, J - 1);
                      c = lane_u32(cbj, J - 1);
                      cnt = lane_u32(cntj, J - 1);
                      child = ((x << 2) | c) & tab.kmask;
                      if (J == J_full) { hint = true; T = EMPTY; }
                      else slow_state(child, &T, &hint);
                      return true;
                    }
                    x = lane_u64(yj, V);
                    c4 = make_uint4(lane_u32(cj.x, V), lane_u32(cj.y, V), lane_u32(cj.z, V), lane_u32(cj.w, V));
                    have_c4 = true;
                  }
                }
              }
              if (have_c4) {
                // (c4 is x's expansion)
              } else if (!was_hit) {
                // not among the lanes: the group lives in another bucket (the minimizer changed), in a
                // bucket too large for the lanes, or does not exist.  Its key is worked out once: the
                // directory word of its bucket (kept in dcache) and, already on its way, its home pair.
                PendingLookup p;
                children_issue_wave(tab, x, &dcache, &p);
                SlotHit h2;
                h2.hit = false; h2.info = 0;
                if (!bl.valid || p.g.bucket != bl.bucket) {
                  bucket_load_wave(tab, rule, p.g.bucket, dcache.lo, dcache.hi, node_hint, &bl, &fetch_u);
                  KM_DC(dc_bload);
                  KM_DT(dt_bload);
                  if (bl.resident) h2 = bucket_find_wave(bl, T);
                }
                if (h2.info & SLOT_SINGLE) {
                  c = h2.info & 3u;
                  cnt = h2.info >> 16;
                  hint = (h2.info & SLOT_HINT) != 0;
                  child = ((x << 2) | c) & tab.kmask;
                  T = tag_of_suffix(child);
                  return true;
                }
                if (!h2.hit) {
                  c4 = children_finish_wave(tab, p, &fetch_u);   // (all zero if the group does not exist)
                  KM_DC(dc_nonres); KM_DT(dt_nonres);
#ifdef KM_DFS_COUNTERS
                  if (bl.S > dc_bigS) dc_bigS = bl.S;
#endif
                } else {
                  c4 = forward_children_wave(tab, x, &dcache, &fetch_u);
                }
              } else {
                // none, several, or an escaped count: the run ends here (or nearly); the counts
                // themselves are not kept in the lanes
                c4 = forward_children_wave(tab, x, &dcache, &fetch_u);
              }
              KM_DT0();
              const uint32_t xm = child_mask(c4, a.ratio, a.nc);
              if (xm != 0 && (xm & (xm - 1)) == 0) {
                c = (uint32_t)__ffs((int)xm) - 1;
                cnt = pick4(c4, c);
                child = ((x << 2) | c) & tab.kmask;
                slow_state(child, &T, &hint);
                KM_DT(dt_tail);
                return true;
              }
              mask = xm;
              expanded = true;
              return false;
            };
            // Two loops: the inner one only takes steps the lanes answer — it never writes the bucket
            // the lanes hold, so those registers are loop-invariant there (with the bucket change inside
            // the same loop the compiler copied all of them on every step); the outer one runs once per
            // bucket change or other slow step.
            for (;;) {
              bool was_hit = false, stop = false;
              KM_DT0();
              for (;;) {
                if (n >= room || hint) { stop = true; break; }
                if (lane == n) { rkey = child; rcnt = cnt; }
                ++n;
                x = child;
                // ---- the expansion of x: its group among the slots the lanes hold
                SlotHit h;
                h.hit = false; h.info = 0;
                if (bl.resident) h = bucket_find_wave(bl, T);   // (resident implies valid)
                if (!(h.info & SLOT_SINGLE)) { was_hit = h.hit; break; }   // (info is 0 on a miss)
                c = h.info & 3u;
                cnt = h.info >> 16;
                hint = (h.info & SLOT_HINT) != 0;
                child = ((x << 2) | c) & tab.kmask;
                T = tag_of_suffix(child);
              }
              KM_DT(dt_chain);
              if (stop) break;
              if (!step_slow(was_hit)) break;
            }
            if (!expanded) {
              // x still has its one child c to take (no room left, or the hint): all the general
              // step needs of the expansion is that child's count
              c4 = make_uint4(cnt, cnt, cnt, cnt);
              mask = 1u << c;
            }
            const uint64_t x_end = x;
            if (n) {
              steps += n;
              KM_DC(dc_runs); KM_DCN(dc_steps, n);
              KM_DT0();
              // ---- booking: first child of the run that is already a node or on the stack
              const bool act = lane < n;
              bool found = false;
              int slot = -1;
              uint32_t st8 = 0;
              if (act) {
                if (ref_index(rkey) != NO_NODE) {            // a k-mer of the target: a node
                  found = true; st8 = ST_NODE; slot = 0;
                } else {
                  slot = set_lookup_lane<2>(keys, cap, rkey, &found);
                  if (found) st8 = meta_state(state[slot]);
                }
              }
              bool stop = act && (slot < 0 || (found && (st8 == ST_NODE || st8 == ST_ONSTACK)));
              // ... or repeats an earlier child of the run (a loop).  Every child of a run is a
              // function of the one before it (its only kept child), so a run that repeats a
              // k-mer is periodic from there on and its LAST k-mer occurs earlier as well: one
              // comparison tells whether the pairwise search is needed at all.
              if (__any((int)(lane + 1 < n && rkey == lane_u64(rkey, n - 1)))) {
                for (uint32_t i = 0; i + 1 < n; ++i)
                  stop = stop || (act && lane > i && rkey == lane_u64(rkey, i));
              }
              const unsigned long long stops = __ballot(stop);
              const uint32_t f = stops ? (uint32_t)__ffsll((long long)stops) - 1 : n;
              bool fresh = false;
              if (lane < f) {
                if (!found) {
                  slot = set_insert_lane<2>(keys, cap, rkey, &fresh);
                }
                if (slot >= 0) {
                  state[slot] = slot_meta(ST_ONSTACK, 0);
                  fk[depth + lane] = rkey; fc[depth + lane] = rcnt; fs[depth + lane] = (uint32_t)slot;
                }
              }
              if (__any((int)(lane < f && slot < 0))) { st = BIG ? T_INTERNAL : T_NEEDS_BIG; break; }
              set_count += (uint32_t)__popcll(__ballot(fresh));
              probes_u += 4ull * f;
              parent_brk = brk;
              if (f == n) {
                // the whole run stands: x is the top of the stack, (c4, mask) its expansion
                depth += n;
                cur = x_end;
                if (__popc(mask) > 1) {
                  ++brk;
                  if (brk > a.max_break) mask = 0;
                }
                if (mask && depth + 1 <= a.max_stack && !(hint && !expanded)) {   // the general step continues: request its lookup (not for a child that is about to rejoin)
                  const uint64_t next = ((cur << 2) | ((uint32_t)__ffs((int)mask) - 1)) & tab.kmask;
                  if (!(pend.valid && pend.X == next)) children_issue_wave(tab, next, &dcache, &pend);
                }
              } else {
                // child f is not new: its parent (child f - 1, or cur) is the top of the stack and
                // has exactly this child left
                const uint64_t kf = lane_u64(rkey, f);
                const uint32_t cf = lane_u32(rcnt, f);
                depth += f;
                if (f) cur = lane_u64(rkey, f - 1);
                c4 = make_uint4(cf, cf, cf, cf);
                mask = 1u << (uint32_t)(kf & 3);
                pend.valid = false;
              }
              step_sync();
              KM_DT(dt_book);
            }
          }
          if (mask == 0) {
            if (bsp == 0) break;                         // DFS from this seed is done
            KM_DT0();
            --bsp;
            mem_sync();
            const BranchFrame f = bf[bsp];
            for (uint32_t j = f.depth + lane; j < depth; j += 64) {
              const uint32_t s = fs[j];
              if (meta_state(state[s]) == ST_ONSTACK) state[s] = slot_meta(ST_POPPED, 0);
            }
            depth = f.depth;
            if (reg > depth) reg = depth;
            cur = fk[depth - 1];
            c4 = f.c4; mask = f.mask; brk = f.brk;
            step_sync();
            KM_DT(dt_unwind);
            continue;
          }
          KM_DT0();
          const uint32_t c = (uint32_t)__ffs((int)mask) - 1;
          mask &= mask - 1;
          const uint64_t child = ((cur << 2) | c) & tab.kmask;
          const uint32_t ccnt = pick4(c4, c);
          bool found;
          int slot = 0;
          uint32_t stt;
          if (__builtin_amdgcn_readfirstlane((int)(ref_index(child) != NO_NODE))) {   // a k-mer of the target: a node
            found = true; stt = ST_NODE;
          } else {
            slot = set_find<2>(keys, cap, child, &found);
            if (slot < 0) { st = BIG ? T_INTERNAL : T_NEEDS_BIG; break; }
            stt = found ? meta_state(state[slot]) : 0u;
          }
          if (found && (stt == ST_NODE || stt == ST_ONSTACK)) {
            // rejoin (or loop): for p in stack: node_data[p] = jf.query(p)
            probes_u += depth;
            if (stt == ST_ONSTACK) {
              // a loop: the child is on the stack and not yet a node — the reference logs it (-v).  It becomes a
              // node right below: its index follows from its stack position (the frame whose set slot this is)
              uint32_t jc = 0xFFFFFFFFu;
              mem_sync();
              for (uint32_t j0 = reg; j0 < depth && jc == 0xFFFFFFFFu; j0 += 64) {       // wave-uniform
                const uint32_t j = j0 + lane;
                const unsigned long long hit_ = __ballot(j < depth && fs[j] == (uint32_t)slot);
                if (hit_) jc = j0 + (uint32_t)__ffsll((long long)hit_) - 1;
              }
              if (lane == 0 && jc != 0xFFFFFFFFu) {
                const uint32_t at = atomicAdd(&a.loop_ctl[0], 1u);
                if (at < a.loop_cap) { a.loop_list[2 * at] = t; a.loop_list[2 * at + 1] = n_nodes + (jc - reg); }
              }
            }
            if (reg < depth) {
              const uint32_t add = depth - reg;
              if (n_nodes + add > node_cap) { st = BIG ? T_INTERNAL : T_NEEDS_BIG; break; }
              mem_sync();
              for (uint32_t j = reg + lane; j < depth; j += 64) {
                const uint64_t kx = fk[j];
                a.node_kmer[nb + n_nodes + (j - reg)] = kx;
                a.node_cnt[nb + n_nodes + (j - reg)] = fc[j];
                state[fs[j]] = slot_meta(ST_NODE, n_nodes + (j - reg));
              }
              n_nodes += add;
              reg = depth;
              step_sync();
            }
            KM_DT(dt_rejoin);
          } else if (depth + 1 <= a.max_stack) {
            // __extend(stack + [child], breaks)
            if (!found && set_count + 1 > set_limit) {
              // drop the POPPED tombstones: rebuild the set from nodes + live stack
              __syncthreads();
              for (uint32_t s = lane; s < cap; s += 64) { keys[s] = EMPTY; state[s] = 0; }
              __syncthreads();
              bool wn;
              for (uint32_t j = lane; j < n_nodes; j += 64) {
                const uint64_t kk = (j < n_ref) ? kmer_at(j) : a.node_kmer[nb + j];
                if (j < n_ref && (uint32_t)pos[pos_home((uint32_t)(kk >> 2))] == j) continue;   // it owns its slot of the position table
                const int s2 = set_insert_lane<2>(keys, cap, kk, &wn);
                if (s2 >= 0) state[s2] = slot_meta(ST_NODE, j);
              }
              __syncthreads();
              for (uint32_t j = 1 + lane; j < depth; j += 64) {       // (frame 0 is the seed: a node)
                const int s2 = set_insert_lane<2>(keys, cap, fk[j], &wn);
                if (s2 >= 0) {
                  if (wn) state[s2] = slot_meta(ST_ONSTACK, 0);
                  fs[j] = (uint32_t)s2;
                }
              }
              __syncthreads();
              set_count = n_losers + (n_nodes - n_ref) + (depth - reg);
              if (set_count + 1 > set_limit) { st = BIG ? T_INTERNAL : T_NEEDS_BIG; break; }
              slot = set_find<2>(keys, cap, child, &found);
              if (slot < 0 || found) { st = T_INTERNAL; break; }
            }
            if (depth >= a.fcap) { st = BIG ? T_INTERNAL : T_NEEDS_BIG; break; }   // fast tier: bounded frames
            if (mask != 0) {
              if (bsp >= a.bcap) { st = BIG ? T_INTERNAL : T_NEEDS_BIG; break; }
              if (lane == 0) {
                BranchFrame f;
                f.c4 = c4; f.depth = depth; f.mask = mask; f.brk = brk; f.pad = 0;
                bf[bsp] = f;
              }
              ++bsp;
            }
            if (!found) ++set_count;
            if (lane == 0) {
              if (!found) keys[slot] = child;
              state[slot] = slot_meta(ST_ONSTACK, 0);
              fk[depth] = child; fc[depth] = ccnt; fs[depth] = (uint32_t)slot;
            }
            parent_brk = brk;
            ++depth;
            cur = child;
            need_expand = true;
            step_sync();
            KM_DT(dt_rejoin);
          }
          // else: the child's __extend returns at once (len(stack) > max_stack)
        }
        if (st != T_OK) break;
        KM_DT0();
        mem_sync();
        for (uint32_t j = 1 + lane; j < depth; j += 64) {
          const uint32_t s = fs[j];
          if (meta_state(state[s]) == ST_ONSTACK) state[s] = slot_meta(ST_POPPED, 0);
        }
        step_sync();
        KM_DT(dt_pop);
        // the next seed's __extend call (there is one unless this was the last
        // target k-mer) checks the node limit first
        if (n_nodes > a.max_node && i + 1 < n_ref) st = T_NODE_LIMIT;
      }
    }
  }

  // ---- epilogue: the graph of nearly every flagged target, read off the node set --------------
  // MutationFinder.graph_analysis + Graph (km/utils/MutationFinder.py:496-572, km/utils/Graph.py:63-240)
  // on "reference chain + forward bubbles" is a closed form (graph_kernel.h 2c has the argument for one
  // bubble; tests/test_bubble_theory.py checks the conditions below against the oracle for one and for
  // several): edges are i -> j iff suffix(i) == prefix(j), so with
  //  (R1) no two reference nodes sharing their (k-1)-mer prefix, nothing behind the last reference suffix,
  //  (R2) a walk node sharing its prefix with at most one other node, and that one a reference node x >= 1
  //       (both are children of a = x - 1): the walk node heads a bubble hanging off a,
  //  (R3) the suffix of every walk node e being the prefix of exactly one node: e + 1, not itself a head,
  //  (R4) or a reference node b, which ends the bubble (the next walk node, if any, heads the next): b > a
  //       with (b - a) x 0.01 well below the bubble's cost, or b <= a (a tandem duplication: the
  //       bubble leads back; both shortest-path trees stay on the reference chain then as well),
  // the paths are the reference path and, per bubble, 0..a, the bubble, b..n_ref-1 (and, for a bubble whose last
  // node points at its own head beside b = a + 1 — a loop — the same with the bubble twice).  The node set of the
  // walk answers "which node is this k-mer" (state NODE + index per slot), and "who has prefix P" is
  // the four k-mers P+A, P+C, P+G, P+T: no prefix table, no adjacency, no Dijkstra.  Everything else
  // (loops, dead ends, nested bubbles, a bubble cheaper than the reference route, more than
  // EPI_MAX_BUBBLES of them) goes to k_graph through the `left` list, untouched.
  if (a.stamps) life2 = __builtin_amdgcn_s_memrealtime();
  bool answered = false;
  if constexpr (!BIG) {
    const uint32_t m = n_nodes, n_walk = n_nodes - n_ref;
    if (a.epi != nullptr && ref_pure && st == T_OK && n_ref >= 2 && n_walk <= 64u * EPI_CHUNKS && m < 0x3FFFu) {
      constexpr uint32_t NONE = 0xFFFFFFFFu;
      constexpr uint32_t REF_REGS = 8;                     // reference counts a lane keeps (targets of up to 512 k-mers)
      // the nodes whose k-mer has the (k-1)-mer prefix P, the node `self` aside: how many, and one of
      // them.  They all sit in the probe sequence that starts at P's home slot.
      auto nodes_with_prefix = [&](uint64_t P, uint32_t self, uint32_t* one, uint32_t* first = nullptr) -> uint32_t {
        uint32_t cnt = 0, sl = set_home(P, cap);
        {                                                    // the target k-mer that owns P's slot of the position table
          const uint32_t w = (uint32_t)pos[pos_home((uint32_t)P)];
          if (w != POS_NONE && w != self && (kmer_at(w) >> 2) == P) {
            if (first) *first = w;
            cnt = 1;
            *one = w;
          }
        }
        for (uint32_t step = 0; step < cap; ++step) {
          const uint64_t kv = keys[sl];
          if (kv == EMPTY) break;
          if ((kv >> 2) == P) {
            const uint16_t mv = state[sl];
            if (meta_state(mv) == ST_NODE && meta_index(mv) != self) {
              if (cnt == 0 && first) *first = meta_index(mv);
              ++cnt;
              *one = meta_index(mv);                       // (the last one found)
            }
          }
          if (++sl == cap) sl = 0;
        }
        return cnt;
      };
      const uint32_t* ncnt = a.node_cnt + nb;
      // ---- everything that comes from memory is requested first: where to write, the k-mers and
      // counts of the walk's nodes (lane l of chunk q owns node n_ref + 64 q + l), the reference's counts
      const EpiArgs ea = *a.epi;
      __syncthreads();                                     // node_kmer / node_cnt were stored by this wave
      uint64_t w_x[EPI_CHUNKS];
      uint32_t w_c[EPI_CHUNKS], rc[REF_REGS];
#pragma unroll
      for (uint32_t q = 0; q < EPI_CHUNKS; ++q) {
        const uint32_t e = n_ref + 64u * q + lane;
        w_x[q] = 0; w_c[q] = 0xFFFFFFFFu;
        if (e < m) { w_x[q] = a.node_kmer[nb + e]; w_c[q] = ncnt[e]; }
      }
#pragma unroll
      for (uint32_t q = 0; q < REF_REGS; ++q) {
        const uint32_t j = lane + 64u * q;
        rc[q] = 0xFFFFFFFFu;
        if (j < n_ref) rc[q] = ncnt[j];
      }
      uint32_t bad = 0;
      const uint64_t last_suffix = kmer_at(n_ref - 1) & tab.pmask;    // (R1): no walk node behind it either
      uint32_t w_a[EPI_CHUNKS], w_nx[EPI_CHUNKS], w_lp[EPI_CHUNKS];
      unsigned long long Hm[EPI_CHUNKS], Em[EPI_CHUNKS];
      // pass 1 — who shares a walk node's prefix: at most one node, a reference node x >= 1 (R2)
#pragma unroll
      for (uint32_t q = 0; q < EPI_CHUNKS; ++q) {
        const uint32_t e = n_ref + 64u * q + lane;
        bool head = false;
        w_a[q] = 0; w_nx[q] = NONE; w_lp[q] = NONE;
        if (64u * q < n_walk) {                            // wave-uniform
          if (e < m) {
            const uint64_t X = w_x[q];
            if ((X >> 2) == last_suffix) bad = 1;
            uint32_t other = NONE;
            const uint32_t sh = nodes_with_prefix(X >> 2, e, &other);
            if (sh > 1) bad = 1;
            if (sh == 1) {
              if (other >= n_ref || other < 1) bad = 1;    // two walk nodes share a prefix / off the source
              head = true;
              w_a[q] = other - 1;
            }
          }
        }
        Hm[q] = __ballot(head);
      }
      // pass 2 — who is behind a walk node's suffix (R3, R4).  Node e + 1 is, when its prefix is that
      // suffix; the others behind it would share e + 1's prefix, i.e. make it a head (pass 1).  So a
      // node followed by a non-head e + 1 with the right prefix is settled without a scan; the rest
      // (the last node of every bubble) scan the cluster of their suffix: exactly one node, a
      // reference node b.
#pragma unroll
      for (uint32_t q = 0; q < EPI_CHUNKS; ++q) {
        const uint32_t e = n_ref + 64u * q + lane;
        bool is_end = false;
        if (64u * q < n_walk) {                            // wave-uniform
          const uint64_t X = w_x[q];
          // the next node's k-mer: lane + 1, or lane 0 of the next chunk
          uint64_t Xn = (uint64_t)__shfl_down((unsigned long long)X, 1);
          if (q + 1 < EPI_CHUNKS) { const uint64_t first_next = lane_u64(w_x[q + 1], 0); if (lane == 63) Xn = first_next; }
          const unsigned long long Hn = (Hm[q] >> 1) | (q + 1 < EPI_CHUNKS ? (Hm[q + 1] << 63) : 0ull);
          const bool next_head = (Hn >> lane) & 1ull;
          const uint64_t S = X & tab.pmask;
          const bool chained = e + 1 < m && (Xn >> 2) == S && !next_head;
          if (e < m && !chained) {
            uint32_t nx = NONE, n1 = NONE;
            const uint32_t sc = nodes_with_prefix(S, NONE, &nx, &n1);
            if (sc == 2) {
              // two nodes behind the end of a bubble: a reference node b and a walk node — the bubble's OWN head,
              // if this is a loop (a tandem duplication a little shorter than k: the head shares its prefix with
              // b = a + 1, so whatever points at b points at the head as well; checked below, where the head is
              // known).  One more path then: through the bubble twice (tests/test_bubble_theory.py).
              const uint32_t lo = n1 < nx ? n1 : nx, hi = n1 < nx ? nx : n1;
              if (lo < n_ref && hi >= n_ref) { nx = lo; w_lp[q] = hi; }
              else bad = 1;
            } else if (sc != 1 || nx >= n_ref) {
              bad = 1;                                     // (a walk node behind it that is not e + 1, or e + 1 a head as well)
            }
            is_end = true;
            w_nx[q] = nx;
          }
          if (e < m && chained) w_nx[q] = e + 1;
        }
        Em[q] = __ballot(is_end);
      }
      // a head follows every end and nothing else; the first walk node is a head, the last an end
      uint32_t n_bub = 0;
#pragma unroll
      for (uint32_t q = 0; q < EPI_CHUNKS; ++q) {
        const uint32_t e = n_ref + 64u * q + lane;
        if (e < m) {
          const unsigned long long Hn = (Hm[q] >> 1) | (q + 1 < EPI_CHUNKS ? (Hm[q + 1] << 63) : 0ull);
          const bool next_head = (Hn >> lane) & 1ull;
          const bool is_end = (Em[q] >> lane) & 1ull;
          if (e + 1 < m ? (next_head != is_end) : !is_end) bad = 1;
        }
        n_bub += (uint32_t)__popcll(Em[q]);
      }
      if (n_walk && !(Hm[0] & 1ull)) bad = 1;
      // per bubble (at its end node): a from its head, b, and — for a forward bubble — the margin that
      // keeps both shortest-path trees on the reference edges
      uint32_t my_a = 0, my_s = 0, my_e = 0, my_b = 0;     // lane r: bubble r (in node order)
      bool my_loop = false;                                // ... and whether it leads back to its own head as well
      if (!__any((int)bad) && n_bub <= EPI_MAX_BUBBLES) {
        uint32_t rank_base = 0;
#pragma unroll
        for (uint32_t q = 0; q < EPI_CHUNKS; ++q) {
          if (Em[q] == 0ull) continue;                     // wave-uniform
          // head position at or before this lane's node, over the chunks up to q
          uint32_t hpos = NONE;
#pragma unroll
          for (uint32_t q2 = 0; q2 <= q; ++q2) {
            const unsigned long long hm = q2 == q ? (Hm[q2] & (~0ull >> (63 - lane))) : Hm[q2];
            if (hm) hpos = 64u * q2 + 63u - (uint32_t)__clzll((long long)hm);
          }
          // a of that head: held by the lane that owns it
          uint32_t a_here = 0;
#pragma unroll
          for (uint32_t q2 = 0; q2 <= q; ++q2) {
            const uint32_t v = (uint32_t)__shfl((int)w_a[q2], (int)(hpos & 63u));
            if (hpos != NONE && (hpos >> 6) == q2) a_here = v;
          }
          const bool is_end = (Em[q] >> lane) & 1ull;
          const uint32_t e = n_ref + 64u * q + lane;
          if (is_end) {
            const uint32_t s_node = n_ref + hpos, b_node = w_nx[q];
            if (hpos == NONE || b_node > n_ref - 1 ||
                (a_here < b_node && (uint64_t)(b_node - a_here) + 10 > 100ull * (e - s_node + 2))) bad = 1;
            if (w_lp[q] != NONE && (w_lp[q] != s_node || b_node != a_here + 1)) bad = 1;   // not this bubble's own loop
          }
          // hand bubble r to lane r
          for (unsigned long long em = Em[q]; em; em &= em - 1) {
            const uint32_t l = (uint32_t)__ffsll((long long)em) - 1;
            const uint32_t r = rank_base + (uint32_t)__popcll(Em[q] & ((1ull << l) - 1ull));
            const uint32_t va = lane_u32(a_here, l), vs = lane_u32(hpos, l), vb = lane_u32(w_nx[q], l);
            const uint32_t vl = lane_u32(w_lp[q], l);
            if (lane == r) { my_a = va; my_s = n_ref + vs; my_e = n_ref + 64u * q + l; my_b = vb; my_loop = vl != NONE; }
          }
          rank_base += (uint32_t)__popcll(Em[q]);
        }
      }
      if (!__any((int)bad) && n_bub <= EPI_MAX_BUBBLES) {
        // ---- emission (graph_kernel.h: emit_bubble, for every bubble).  The pool space is asked for
        // now; the answer is needed only after the coverages are known
        const uint32_t pg = t % POOL_GROUPS;
        const uint64_t pg_paths = ea.path_pool / POOL_GROUPS, pg_runs = ea.run_pool / POOL_GROUPS;
        // (a looped bubble has two paths: once and twice through it — three and four runs)
        const unsigned long long loops = __ballot(lane < n_bub && my_loop);
        const uint32_t n_loop = (uint32_t)__popcll(loops);
        const unsigned long long want_paths = 1ull + n_bub + n_loop, want_runs = 1ull + 3ull * n_bub + 4ull * n_loop;
        unsigned long long pb = 0, rb = 0;
        if (lane == 0) {
          unsigned long long* ctr = ea.counters + (uint64_t)pg * POOL_CTR_STRIDE;
          pb = atomicAdd(&ctr[0], want_paths);
          rb = atomicAdd(&ctr[1], want_runs);
        }
        uint32_t ref_min = 0xFFFFFFFFu, ref_max = 0;
#pragma unroll
        for (uint32_t q = 0; q < REF_REGS; ++q) {
          const uint32_t j = lane + 64u * q;
          if (j < n_ref) { ref_min = rc[q] < ref_min ? rc[q] : ref_min; ref_max = rc[q] > ref_max ? rc[q] : ref_max; }
        }
        for (uint32_t j = lane + 64u * REF_REGS; j < n_ref; j += 64) {       // longer targets
          const uint32_t cv = ncnt[j];
          ref_min = cv < ref_min ? cv : ref_min;
          ref_max = cv > ref_max ? cv : ref_max;
        }
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t x0 = __shfl_xor(ref_min, o); ref_min = x0 < ref_min ? x0 : ref_min;
          const uint32_t x1 = __shfl_xor(ref_max, o); ref_max = x1 > ref_max ? x1 : ref_max;
        }
        uint32_t my_mc = 0xFFFFFFFFu;
        for (uint32_t r = 0; r < n_bub; ++r) {             // wave-uniform
          const uint32_t ba = lane_u32(my_a, r), bs = lane_u32(my_s, r), be = lane_u32(my_e, r), bb = lane_u32(my_b, r);
          uint32_t mc = 0xFFFFFFFFu;                       // min over 0..a, s..e, b..n_ref-1
#pragma unroll
          for (uint32_t q = 0; q < REF_REGS; ++q) {
            const uint32_t j = lane + 64u * q;
            if (j < n_ref && (j <= ba || j >= bb)) mc = rc[q] < mc ? rc[q] : mc;
          }
          for (uint32_t j = lane + 64u * REF_REGS; j < n_ref; j += 64)
            if (j <= ba || j >= bb) { const uint32_t cv = ncnt[j]; mc = cv < mc ? cv : mc; }
#pragma unroll
          for (uint32_t q = 0; q < EPI_CHUNKS; ++q) {
            const uint32_t e = n_ref + 64u * q + lane;
            if (e >= bs && e <= be) mc = w_c[q] < mc ? w_c[q] : mc;
          }
          for (int o = 32; o > 0; o >>= 1) { const uint32_t x0 = __shfl_xor(mc, o); mc = x0 < mc ? x0 : mc; }
          if (lane == r) my_mc = mc;
        }
        pb = lane_u64(pb, 0);
        rb = lane_u64(rb, 0);
        if (pb + want_paths > pg_paths || rb + want_runs > pg_runs) {
          // pools exhausted: the host enlarges them and reruns the graph stage over every flagged target
          if (lane == 0) {
            atomicExch(ea.counters + (uint64_t)POOL_GROUPS * POOL_CTR_STRIDE, 1ull);
            ea.t_npaths[t] = 0; ea.t_pathbase[t] = 0; ea.t_nruns[t] = 0; ea.t_refmax[t] = NOT_BARE; ea.g_status[t] = T_OK;
          }
        } else {
          pb += (uint64_t)pg * pg_paths;
          rb += (uint64_t)pg * pg_runs;
          if (lane == 0) {
            ea.r_start[rb] = 0; ea.r_len[rb] = n_ref;
            ea.p_target[pb] = t; ea.p_runbase[pb] = rb; ea.p_nruns[pb] = 1; ea.p_len[pb] = n_ref; ea.p_mincov[pb] = ref_min;
            ea.t_npaths[t] = (uint32_t)want_paths; ea.t_pathbase[t] = (uint32_t)pb; ea.t_nruns[t] = (uint32_t)want_runs;
            ea.t_refmax[t] = n_bub ? NOT_BARE : ref_max;
            ea.g_status[t] = T_OK;
            // (the sink tree from node 0 is the reference chain: every one of its edges but the first is stripped;
            // what stays are the two of the caps' side — (source, 0), (0, 1) — and the bubbles' own)
            ea.t_eremoved[t] = n_ref - 1;
            ea.t_enonref[t] = 2u + n_walk + n_bub + n_loop;
          }
          if (lane < n_bub) {
            const uint64_t r0 = rb + 1 + 3ull * lane, p0 = pb + 1 + lane;
            ea.r_start[r0] = 0; ea.r_len[r0] = my_a + 1;
            ea.r_start[r0 + 1] = my_s; ea.r_len[r0 + 1] = my_e - my_s + 1;
            ea.r_start[r0 + 2] = my_b; ea.r_len[r0 + 2] = n_ref - my_b;
            ea.p_target[p0] = t; ea.p_runbase[p0] = r0; ea.p_nruns[p0] = 3;
            ea.p_len[p0] = (my_a + 1) + (my_e - my_s + 1) + (n_ref - my_b); ea.p_mincov[p0] = my_mc;
            if (my_loop) {                                 // 0..a, the bubble twice, b..n_ref-1: the same nodes, the same coverage
              const uint32_t lr = (uint32_t)__popcll(loops & ((1ull << lane) - 1ull));
              const uint64_t r1 = rb + 1 + 3ull * n_bub + 4ull * lr, p1 = pb + 1 + n_bub + lr;
              ea.r_start[r1] = 0; ea.r_len[r1] = my_a + 1;
              ea.r_start[r1 + 1] = my_s; ea.r_len[r1 + 1] = my_e - my_s + 1;
              ea.r_start[r1 + 2] = my_s; ea.r_len[r1 + 2] = my_e - my_s + 1;
              ea.r_start[r1 + 3] = my_b; ea.r_len[r1 + 3] = n_ref - my_b;
              ea.p_target[p1] = t; ea.p_runbase[p1] = r1; ea.p_nruns[p1] = 4;
              ea.p_len[p1] = (my_a + 1) + 2 * (my_e - my_s + 1) + (n_ref - my_b); ea.p_mincov[p1] = my_mc;
            }
          }
        }
        answered = true;
      }
    }
    if (a.epi != nullptr && !answered && lane == 0) {
      a.epi->t_refmax[t] = NOT_BARE;
      a.epi->left[atomicAdd(a.epi->n_left, 1u)] = t;
    }
  }

  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + 32ull * blockIdx.x;
    o[0] = life0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = t; o[3] = n_nodes - n_ref; o[4] = life1; o[5] = life2; o[6] = 0; o[7] = lifeB; o[8] = lifeC;
#ifdef KM_DFS_COUNTERS
    o[19] = dt_spec; o[20] = dt_book; o[21] = dt_gen; o[22] = dt_bload; o[23] = dt_rejoin; o[24] = dt_unwind; o[25] = dt_align;
    o[26] = dc_nonres; o[27] = dt_nonres; o[28] = dc_bigS;
    o[29] = ((unsigned long long)dt_post << 32) | dt_pre; o[30] = ((unsigned long long)dt_chain << 32) | dt_tail;
    o[6] = ((unsigned long long)dt_pop << 32) | dt_seed0;
    o[9] = dc_slow; o[10] = dc_spec; o[11] = dc_rec; o[12] = dc_bload; o[13] = dc_gen; o[14] = dc_runs; o[15] = dc_full; o[16] = dc_v0; o[17] = dc_steps; o[18] = dc_noalign;
#endif
    o[31] = 0x6c6966655f646673ull;
  }
  if (lane == 0) {
    a.status[t] = st;
    if constexpr (!BIG) {
      if (st == T_NEEDS_BIG && a.big_ctl) { const uint32_t at = atomicAdd(&a.big_ctl[0], 1u); if (at < a.big_slots) a.big_walk[at] = t; }
    }
    if (st != T_NEEDS_BIG) {     // a large-tier rerun restarts from the seed kernel's counters
      a.n_nodes[t] = n_nodes;
      a.dfs_probes[t] = probes_u;
      a.fetches[t] += fetch_u;
    }
  }
}

}  // namespace kmd
