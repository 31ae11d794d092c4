/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * Plain-C restatement of the reference's find_mutation hot path, fast enough to check
 * every target of a full-size batch.  Same canonical order as oracle/km_oracle.py (target
 * k-mers registered and extended in target order, alternative paths sorted by index
 * sequence); it is pinned against that module and, through it, against the golden vectors
 * of the unmodified reference (tests/test_oracle_c.py).
 *
 *   ko_query / children        <- km/utils/Jellyfish.py:47-53, 55-72
 *   walk (register + extend)   <- km/utils/MutationFinder.py:100-124, 137-165
 *   dense float32 Dijkstra     <- km/utils/Graph.py:63-119 (O(n^2), first-index argmin, strict <)
 *   strip reference edges      <- km/utils/Graph.py:184-197 (incl. the `if last_cur` quirk)
 *   unique shortest paths      <- km/utils/Graph.py:200-240
 *   min coverage               <- km/utils/MutationFinder.py:490-494, 639
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build: make -C oracle   ->  oracle/_build/libkmoracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int k, canonical;
  uint64_t cap, mask;   /* open addressing, cap = power of two */
  uint64_t* keys;
  uint32_t* vals;       /* 0 = empty */
  uint64_t probes;      /* logical probes: one per ko_query */
} ko_db;

static uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static uint64_t revcomp(uint64_t x, int k) {
  uint64_t r = 0;
  for (int i = 0; i < k; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
  return r;
}

ko_db* ko_open(const uint64_t* keys, const uint32_t* counts, uint64_t n, int k, int canonical) {
  ko_db* d = (ko_db*)calloc(1, sizeof *d);
  d->k = k; d->canonical = canonical;
  d->cap = 16; while (d->cap < 2 * n + 2) d->cap <<= 1;
  d->mask = d->cap - 1;
  d->keys = (uint64_t*)calloc(d->cap, 8);
  d->vals = (uint32_t*)calloc(d->cap, 4);
  for (uint64_t i = 0; i < n; ++i) {
    if (!counts[i]) continue;
    uint64_t s = mix(keys[i]) & d->mask;
    while (d->vals[s] && d->keys[s] != keys[i]) s = (s + 1) & d->mask;
    d->keys[s] = keys[i]; d->vals[s] = counts[i];          /* last record wins, like a dict */
  }
  return d;
}
void ko_close(ko_db* d) { if (d) { free(d->keys); free(d->vals); free(d); } }

uint32_t ko_query(ko_db* d, uint64_t kmer) {
  d->probes++;
  if (d->canonical) { uint64_t r = revcomp(kmer, d->k); if (r < kmer) kmer = r; }
  uint64_t s = mix(kmer) & d->mask;
  while (d->vals[s]) { if (d->keys[s] == kmer) return d->vals[s]; s = (s + 1) & d->mask; }
  return 0;
}

/* children kept by get_child, ACGT order: bit c of the result; counts in cnt[4] */
static int children(ko_db* d, uint64_t kmer, double ratio, int64_t n_cutoff, uint32_t cnt[4]) {
  const uint64_t kmask = d->k >= 32 ? ~0ULL : ((1ULL << (2 * d->k)) - 1);
  uint64_t sum = 0;
  for (int c = 0; c < 4; ++c) { cnt[c] = ko_query(d, ((kmer << 2) | (uint64_t)c) & kmask); sum += cnt[c]; }
  const double t = (double)sum * ratio, nc = (double)n_cutoff;
  const double thr = nc > t ? nc : t;                      /* Python max(t, nc) */
  int m = 0;
  for (int c = 0; c < 4; ++c) if ((double)cnt[c] >= thr) m |= 1 << c;
  return m;
}

/* ---- per-target node dictionary (insertion ordered) ---- */
typedef struct {
  uint64_t* kmer; uint32_t* count; uint32_t n, cap_nodes;
  uint64_t hcap, hmask; uint32_t* hidx;                    /* hash -> node index + 1 */
} nodes_t;

static void nodes_init(nodes_t* nd, uint32_t cap_nodes) {
  nd->cap_nodes = cap_nodes; nd->n = 0;
  nd->kmer = (uint64_t*)malloc(8ull * cap_nodes); nd->count = (uint32_t*)malloc(4ull * cap_nodes);
  nd->hcap = 16; while (nd->hcap < 2ull * cap_nodes + 2) nd->hcap <<= 1;
  nd->hmask = nd->hcap - 1;
  nd->hidx = (uint32_t*)calloc(nd->hcap, 4);
}
static void nodes_free(nodes_t* nd) { free(nd->kmer); free(nd->count); free(nd->hidx); }
static int nodes_find(const nodes_t* nd, uint64_t kmer) {
  uint64_t s = mix(kmer) & nd->hmask;
  while (nd->hidx[s]) { if (nd->kmer[nd->hidx[s] - 1] == kmer) return (int)nd->hidx[s] - 1; s = (s + 1) & nd->hmask; }
  return -1;
}
static void nodes_set(nodes_t* nd, uint64_t kmer, uint32_t count) {
  int i = nodes_find(nd, kmer);
  if (i >= 0) { nd->count[i] = count; return; }
  nd->kmer[nd->n] = kmer; nd->count[nd->n] = count;
  uint64_t s = mix(kmer) & nd->hmask;
  while (nd->hidx[s]) s = (s + 1) & nd->hmask;
  nd->hidx[s] = ++nd->n;
}

typedef struct {
  ko_db* db; nodes_t* nd; double ratio; int64_t n_cutoff;
  uint32_t max_stack, max_break, max_node; uint64_t* stack; int limit_hit;
} walk_t;

static void extend(walk_t* w, uint32_t depth, uint32_t breaks) {
  if (w->limit_hit) return;
  if (depth > w->max_stack) return;
  if (w->nd->n > w->max_node) { w->limit_hit = 1; return; }
  uint32_t cnt[4];
  const int m = children(w->db, w->stack[depth - 1], w->ratio, w->n_cutoff, cnt);
  if (__builtin_popcount((unsigned)m) > 1) { if (++breaks > w->max_break) return; }
  const uint64_t kmask = w->db->k >= 32 ? ~0ULL : ((1ULL << (2 * w->db->k)) - 1);
  for (int c = 0; c < 4 && !w->limit_hit; ++c) {
    if (!((m >> c) & 1)) continue;
    const uint64_t child = ((w->stack[depth - 1] << 2) | (uint64_t)c) & kmask;
    int known = nodes_find(w->nd, child) >= 0;
    for (uint32_t j = 0; !known && j < depth; ++j) known = (w->stack[j] == child);
    if (known) {
      for (uint32_t j = 0; j < depth; ++j) nodes_set(w->nd, w->stack[j], ko_query(w->db, w->stack[j]));
    } else {
      w->stack[depth] = child;
      extend(w, depth + 1, breaks);
    }
  }
}

/* status codes shared with the product: 0 ok, 1 node limit, 2 repeated k-mer, 3 too short */
typedef struct {
  uint32_t status, n_ref, n_nodes, n_paths;
  uint64_t probes;
  uint64_t* node_kmer; uint32_t* node_count;               /* [n_nodes]  (malloc'd) */
  uint32_t* path_off;  uint32_t* path_nodes; uint32_t* path_min_cov;   /* CSR over paths */
} ko_result;

void ko_free_result(ko_result* r) {
  free(r->node_kmer); free(r->node_count); free(r->path_off); free(r->path_nodes); free(r->path_min_cov);
  memset(r, 0, sizeof *r);
}

static void dijkstra_prev(const float* w, uint32_t n, uint32_t start, int transpose, int32_t* prev) {
  float* dist = (float*)malloc(4ull * n);
  uint8_t* todo = (uint8_t*)malloc(n);
  for (uint32_t i = 0; i < n; ++i) { prev[i] = -1; dist[i] = INFINITY; todo[i] = 1; }
  dist[start] = 0.0f;
  for (uint32_t it = 0; it < n; ++it) {
    uint32_t best = n; float bd = 0;
    for (uint32_t i = 0; i < n; ++i) if (todo[i] && (best == n || dist[i] < bd)) { best = i; bd = dist[i]; }
    const uint32_t i = best;
    for (uint32_t j = 0; j < n; ++j) {
      const float wij = transpose ? w[(uint64_t)j * n + i] : w[(uint64_t)i * n + j];
      const float ndist = wij + dist[i];
      if (ndist < dist[j]) { dist[j] = ndist; prev[j] = (int32_t)i; }
    }
    todo[i] = 0;
  }
  free(dist); free(todo);
}

typedef struct { uint32_t* v; uint32_t len; } path_t;
static int path_cmp(const void* a, const void* b) {
  const path_t* p = (const path_t*)a; const path_t* q = (const path_t*)b;
  const uint32_t m = p->len < q->len ? p->len : q->len;
  for (uint32_t i = 0; i < m; ++i) if (p->v[i] != q->v[i]) return p->v[i] < q->v[i] ? -1 : 1;
  return p->len < q->len ? -1 : (p->len > q->len ? 1 : 0);
}

static void graph_paths(const nodes_t* nd, uint32_t n_ref, int k, ko_result* r) {
  const uint32_t m = nd->n, n = m + 2, src = m, snk = m + 1;
  const uint64_t pmask = (1ULL << (2 * (k - 1))) - 1;
  float* w = (float*)malloc(4ull * n * n);
  uint8_t* edge = (uint8_t*)calloc((uint64_t)n * n, 1);
  for (uint64_t x = 0; x < (uint64_t)n * n; ++x) w[x] = INFINITY;
  /* (k-1)-overlap edges, weight 1 */
  for (uint32_t i = 0; i < m; ++i)
    for (uint32_t j = 0; j < m; ++j)
      if (i != j && (nd->kmer[i] & pmask) == (nd->kmer[j] >> 2)) { w[(uint64_t)i * n + j] = 1.0f; edge[(uint64_t)i * n + j] = 1; }
  for (uint32_t i = 0; i + 1 < n_ref; ++i) { w[(uint64_t)i * n + i + 1] = 0.01f; edge[(uint64_t)i * n + i + 1] = 1; }
  w[(uint64_t)src * n + 0] = 0.01f; edge[(uint64_t)src * n + 0] = 1;
  w[(uint64_t)(n_ref - 1) * n + snk] = 0.01f; edge[(uint64_t)(n_ref - 1) * n + snk] = 1;
  int32_t* before = (int32_t*)malloc(4ull * n); int32_t* after = (int32_t*)malloc(4ull * n);
  dijkstra_prev(w, n, src, 0, before);
  dijkstra_prev(w, n, snk, 1, after);
  /* strip reference edges */
  for (uint32_t c0 = 0; c0 < n; ++c0) {
    if (before[c0] != (int32_t)src) continue;
    int32_t cur = (int32_t)c0, last = -1;
    while (after[cur] != -1) {
      cur = after[cur];
      if (last > 0 && edge[(uint64_t)last * n + cur]) edge[(uint64_t)last * n + cur] = 0;   /* `if last_cur` skips None and 0 */
      last = cur;
    }
  }
  /* one path per remaining edge, unique */
  uint32_t np = 0, cap = 16;
  path_t* paths = (path_t*)malloc(cap * sizeof *paths);
  uint32_t* tmp = (uint32_t*)malloc(8ull * n + 16);
  for (uint32_t a = 0; a < n; ++a) for (uint32_t b = 0; b < n; ++b) {
    if (!edge[(uint64_t)a * n + b]) continue;
    uint32_t la = 0; int32_t x = (int32_t)a;
    while (1) { tmp[la++] = (uint32_t)x; if (before[x] == -1) break; x = before[x]; }
    if (tmp[la - 1] != src) continue;
    for (uint32_t i = 0; i < la / 2; ++i) { uint32_t t = tmp[i]; tmp[i] = tmp[la - 1 - i]; tmp[la - 1 - i] = t; }
    uint32_t len = la; x = (int32_t)b;
    while (1) { tmp[len++] = (uint32_t)x; if (after[x] == -1) break; x = after[x]; }
    if (tmp[len - 1] != snk) continue;
    path_t p; p.len = len - 2; p.v = (uint32_t*)malloc(4ull * (p.len ? p.len : 1));
    memcpy(p.v, tmp + 1, 4ull * p.len);                    /* caps stripped */
    int dup = 0;
    for (uint32_t q = 0; q < np && !dup; ++q) dup = (path_cmp(&paths[q], &p) == 0);
    if (dup) { free(p.v); continue; }
    if (np == cap) { cap *= 2; paths = (path_t*)realloc(paths, cap * sizeof *paths); }
    paths[np++] = p;
  }
  qsort(paths, np, sizeof *paths, path_cmp);
  r->n_paths = np;
  r->path_off = (uint32_t*)malloc(4ull * (np + 1));
  uint64_t tot = 0; for (uint32_t q = 0; q < np; ++q) tot += paths[q].len;
  r->path_nodes = (uint32_t*)malloc(4ull * (tot ? tot : 1));
  r->path_min_cov = (uint32_t*)malloc(4ull * (np ? np : 1));
  uint32_t off = 0;
  for (uint32_t q = 0; q < np; ++q) {
    r->path_off[q] = off;
    uint32_t mc = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < paths[q].len; ++i) { r->path_nodes[off + i] = paths[q].v[i]; if (nd->count[paths[q].v[i]] < mc) mc = nd->count[paths[q].v[i]]; }
    r->path_min_cov[q] = mc; off += paths[q].len; free(paths[q].v);
  }
  r->path_off[np] = off;
  free(paths); free(tmp); free(before); free(after); free(w); free(edge);
}

/* seq: base codes 0..3, length L.  stages: 1 = walk only, 3 = walk + graph. */
int ko_analyse(ko_db* d, const uint8_t* seq, uint32_t L, double ratio, int64_t n_cutoff,
               uint32_t max_stack, uint32_t max_break, uint32_t max_node, int stages, ko_result* r) {
  memset(r, 0, sizeof *r);
  const int k = d->k;
  if (L < (uint32_t)k) { r->status = 3; return 0; }
  const uint32_t n_ref = L - k + 1;
  r->n_ref = n_ref;
  const uint64_t kmask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t* ref = (uint64_t*)malloc(8ull * n_ref);
  uint64_t x = 0;
  for (uint32_t i = 0; i < L; ++i) { x = ((x << 2) | seq[i]) & kmask; if (i + 1 >= (uint32_t)k) ref[i + 1 - k] = x; }
  nodes_t nd;
  nodes_init(&nd, (n_ref > max_node ? n_ref : max_node) + max_stack + 8);
  const uint64_t p0 = d->probes;
  for (uint32_t i = 0; i < n_ref; ++i) {
    if (nodes_find(&nd, ref[i]) >= 0) { r->status = 2; free(ref); nodes_free(&nd); return 0; }   /* get_ref_kmer raises first */
    nodes_set(&nd, ref[i], 0);
  }
  for (uint32_t i = 0; i < n_ref; ++i) nd.count[i] = ko_query(d, ref[i]);
  walk_t w; w.db = d; w.nd = &nd; w.ratio = ratio; w.n_cutoff = n_cutoff;
  w.max_stack = max_stack; w.max_break = max_break; w.max_node = max_node; w.limit_hit = 0;
  w.stack = (uint64_t*)malloc(8ull * (max_stack + 2));
  for (uint32_t i = 0; i < n_ref && !w.limit_hit; ++i) { w.stack[0] = ref[i]; extend(&w, 1, 0); }
  free(w.stack); free(ref);
  r->probes = d->probes - p0;
  r->n_nodes = nd.n;
  r->node_kmer = (uint64_t*)malloc(8ull * nd.n); r->node_count = (uint32_t*)malloc(4ull * nd.n);
  memcpy(r->node_kmer, nd.kmer, 8ull * nd.n); memcpy(r->node_count, nd.count, 4ull * nd.n);
  if (w.limit_hit) { r->status = 1; nodes_free(&nd); return 0; }
  if (stages & 2) graph_paths(&nd, n_ref, k, r);
  nodes_free(&nd);
  return 0;
}
