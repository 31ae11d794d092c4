// report.cpp — host side of SURVEY.md §8f-1: variant naming, path quantification, cluster
// quantification and the TSV rows of `km find_mutation`, for ALL targets of a fetched batch in
// one call.  Plain C++ (no device code): the GPU leaves ~0.3 ms of work per 10 000 targets, the
// Python restatement of this part (km_amd/report.py, kept as the readable specification and
// pinned to the reference's golden TSVs) ~0.5 ms per variant target.
//
// Mirrors, function by function:
//   split_paths / name_variant  <- km/utils/MutationFinder.py:190-373, 405-488
//   fit_paths                   <- km/utils/PathQuant.py:93-154 (least squares, then the
//                                  projected gradient refinement, float64 throughout)
//   cluster_groups              <- km/utils/MutationFinder.py:651-723
//   target rows + their order   <- km/utils/MutationFinder.py:575-648, 726-833,
//                                  km/utils/PathQuant.py:37-49
// The least-squares start is the minimum-norm solution with numpy's rcond=None cut-off: from the eigenvectors
// of the small matrix A^T A while the paths are well separated, through a one-sided Jacobi SVD of A itself for a
// near-singular fit; it agrees with LAPACK's gelsd to ~1e-13 relative, far inside the printed %.3f / %.1f —
// except on rounding ties and near-singular fits, which are flagged.
// Paths arrive as runs of consecutive nodes and are worked on as such wherever the reference's arithmetic allows
// it (which path is the reference, where a variant leaves and rejoins it, the patterns of a fit, the spelling);
// the node vectors remain for the slices of a cluster.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kmgpu.h"

namespace {

// set while formatting a target whose printed values depend on the last bits of the solver
thread_local bool g_tie = false;
thread_local long g_fit_iters = 0;

typedef int32_t Node;                // a node index of one target (targets are far below 2^31 nodes)
typedef std::vector<Node> Path;

struct Target {
  const char* name;
  const char* seq;      // as given (not NUL terminated)
  size_t seq_len;
  int k;
  int64_t n_ref;
  const uint64_t* kmers;    // all nodes, or NULL: then node i < n_ref is the k-mer at base i of seq
  const uint64_t* extra;    // ... and node n_ref + j is extra[j]
  const uint32_t* counts;   // NULL for a bare-reference target delivered lean: only ref_max is known
  uint32_t ref_max;         // max count over the target's own k-mers (bare-reference targets)
  int64_t n_nodes;
  const Path* paths;        // n_paths of them, in the worker's scratch
  size_t n_paths;
  const uint32_t* min_cov;
  const uint8_t* is_ref;    // per path: it is 0, 1, .. n_ref-1 (read off the path's runs when it was laid out)
  bool plain;               // seq is upper-case ACGT throughout: its k-mers spell as they stand
};

struct Split { int64_t start, end_ref, end_var, end_ovl; };

const char LAST[4] = {'A', 'C', 'G', 'T'};

inline uint64_t base_code(char c) { return ((uint64_t)(c >> 1) ^ (uint64_t)(c >> 2)) & 3u; }   // either case
inline uint64_t kmer_of(const Target& t, int64_t node) {
  if (t.kmers) return t.kmers[node];
  if (node >= t.n_ref) return t.extra[node - t.n_ref];
  uint64_t x = 0;
  for (int j = 0; j < t.k; ++j) x = (x << 2) | base_code(t.seq[node + j]);
  return x;
}
inline char tail_of(const Target& t, int64_t node) {
  if (t.kmers) return LAST[t.kmers[node] & 3];
  if (node >= t.n_ref) return LAST[t.extra[node - t.n_ref] & 3];
  return LAST[base_code(t.seq[node + t.k - 1])];
}

// Python index semantics of numpy fancy indexing on a 1-D array of length n
inline bool wrap_index(int64_t i, int64_t n, int64_t* out) {
  if (i < 0) i += n;
  if (i < 0 || i >= n) return false;
  *out = i;
  return true;
}

// km_amd/report.py: split_paths.  Returns false where Python raises IndexError.
// (`alt_runs`: the stretches of consecutive nodes of `alt` as (first, last + 1) pairs, usable when `ref` is the
// path 0, 1, .. nr-1 — then where the two part and where they meet again is read off the stretches, not the nodes.)
bool split_paths(const Path& ref, const Path& alt, int k, Split* s, const std::vector<int64_t>* alt_runs = nullptr) {
  const int64_t nr = (int64_t)ref.size(), na = (int64_t)alt.size();
  const int64_t m = std::min(nr, na);
  int64_t start = m;
  if (alt_runs) {
    // alt[i] == i holds for a whole stretch or for none of it
    int64_t at = 0;
    for (size_t q = 0; q + 1 < alt_runs->size() && at < m; q += 2) {
      const int64_t a = (*alt_runs)[q], b = (*alt_runs)[q + 1];
      if (b <= a) continue;
      if (a != at) { start = at; break; }
      at += b - a;
    }
  } else {
    for (int64_t i = 0; i < m; ++i)
      if (ref[i] != alt[i]) { start = i; break; }
  }
  const int64_t room = m - (start + k) + 1;
  int64_t same = 0;
  if (room > 0) {
    same = room;
    if (alt_runs) {
      // alt[j] == j + (nr - na), counted from the end: again a whole stretch or none of it
      int64_t end = na, got = 0;
      for (size_t q = alt_runs->size(); q >= 2 && got < room; q -= 2) {
        const int64_t a = (*alt_runs)[q - 2], b = (*alt_runs)[q - 1];
        if (b <= a) continue;
        if (b - end != nr - na) break;
        got += b - a;
        end -= b - a;
      }
      same = std::min(got, room);
    } else {
      for (int64_t i = 0; i < room; ++i)
        if (ref[nr - 1 - i] != alt[na - 1 - i]) { same = i; break; }
    }
  }
  const int64_t end_ref = nr - same, end_var = na - same;
  const int64_t room2 = end_ref - start;
  int64_t more = 0;
  if (room2 > 0) {
    const int64_t reach = std::min(room2, end_var + na);
    int64_t first = -1;
    for (int64_t st = 0; st < reach; ++st) {
      int64_t ri, ai;
      if (!wrap_index(end_ref - 1 - st, nr, &ri) || !wrap_index(end_var - 1 - st, na, &ai)) return false;
      if (ref[ri] != alt[ai]) { first = st; break; }
    }
    if (first >= 0) more = first;
    else if (reach < room2) return false;          // "list index out of range"
    else more = room2;
  }
  s->start = start;
  s->end_ref = end_ref;
  s->end_var = end_var;
  s->end_ovl = end_ref - more;
  return true;
}

// Python slice a[lo:hi] with non-negative bounds
inline void slice(const Path& a, int64_t lo, int64_t hi, Path* out) {
  const int64_t n = (int64_t)a.size();
  if (lo < 0) lo = std::max<int64_t>(0, lo + n);
  if (hi < 0) hi = std::max<int64_t>(0, hi + n);
  lo = std::min(lo, n);
  hi = std::min(hi, n);
  out->clear();
  if (hi > lo) out->assign(a.begin() + lo, a.begin() + hi);
}

// Thin SVD of the n x m matrix held column by column in u (m small) by one-sided Jacobi
// rotations (Hestenes): on return the columns of u are sigma_j * u_j, v holds the right
// singular vectors in its columns and sig the singular values.  Works on A itself — forming
// A^T A would square the condition number and make an exactly rank-deficient problem (a tandem
// duplication path next to its double: sigma_3 ~ 1e-15) look full rank.
void jacobi_svd(std::vector<std::vector<double>>& u, int m, std::vector<double>& v, std::vector<double>& sig) {
  const size_t n = m ? u[0].size() : 0;
  v.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) v[(size_t)i * m + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < m; ++p)
      for (int q = p + 1; q < m; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        double* const cp = u[(size_t)p].data();
        double* const cq = u[(size_t)q].data();
        for (size_t i = 0; i < n; ++i) {
          alpha += cp[i] * cp[i];
          beta += cq[i] * cq[i];
          gamma += cp[i] * cq[i];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
        // one of the two is a null column already (norm below 1e-14 of the other's: under the cut-off on the
        // singular values, so it is dropped below whatever it points at): rotating it against the other only
        // stirs rounding noise, sweep after sweep, and an exactly rank-deficient fit never came to rest
        if (std::min(alpha, beta) <= 1e-28 * std::max(alpha, beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (size_t i = 0; i < n; ++i) {
          const double up = cp[i], uq = cq[i];
          cp[i] = c * up - sn * uq;
          cq[i] = sn * up + c * uq;
        }
        for (int r = 0; r < m; ++r) {
          const double vp = v[(size_t)r * m + p], vq = v[(size_t)r * m + q];
          v[(size_t)r * m + p] = c * vp - sn * vq;
          v[(size_t)r * m + q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  sig.resize((size_t)m);
  for (int j = 0; j < m; ++j) {
    double s2 = 0;
    for (size_t i = 0; i < n; ++i) s2 += u[(size_t)j][i] * u[(size_t)j][i];
    sig[(size_t)j] = std::sqrt(s2);
  }
}

// Per-thread working storage: every container below keeps its capacity from target to target, so a worker
// allocates while it warms up and then no more (a 10 000-target batch used to make ~400 000 small allocations,
// most of them freed by another thread than the one that made them).
struct RowRec {
  size_t off, len;            // the row's text in the worker's output buffer (no trailing newline in len)
  std::string name;           // "Type\tvariant"
  std::string mincov;         // as printed
  std::string note;           // "vs_ref" / "cluster N n=M"
};

struct Group { int64_t lo, hi; std::vector<int> members; };

struct Scratch {
  std::vector<Path> paths;
  std::vector<std::vector<int64_t>> path_runs;   // per path: its stretches of consecutive nodes as (first, last + 1) pairs
  std::vector<int64_t> ref_run, cref_run;        // the same for `ref` and `cref` (one stretch each)
  std::vector<const std::vector<int64_t>*> set_runs;   // for the paths of `set`: their stretches when known, else NULL
  std::vector<uint32_t> min_cov;
  std::vector<uint8_t> is_ref;         // per path of the target
  std::vector<uint32_t> counts32;      // a target's counts when the view carries them as 16-bit values
  const uint32_t* cnt = nullptr;       // the target's counts, and whether one of them is 2^24 or more (then a count
  bool cnt_wide = false;               // enters a fit as its nearest float32, PathQuant.py:99)
  Path ref, cref;
  struct Event { int64_t pos; int32_t c, d; };
  std::vector<Event> events;
  std::vector<int64_t> seg;            // the stretches of nodes that lie on at least one path of the fit (lo, hi pairs)
  std::vector<int32_t> cmap;
  std::vector<int32_t> row;
  std::vector<Path> clipped;
  std::vector<const Path*> set;
  std::vector<int32_t> pat;
  std::vector<double> pat_sum, pat_n, g, atb, ev, coef, rvaf, grad, resid;
  std::vector<RowRec> rows;
  size_t n_rows = 0;
  std::vector<size_t> order;
  std::string tmp, gone, neu, name, cref_seq;
  std::vector<int> todo;
  std::vector<Group> groups;
  size_t n_groups = 0;
  std::vector<Split> diffs;
};

// km_amd/report.py: fit_paths over the node counts as float32 values (w.cnt: summed stretch by stretch) followed
// by -1, -1 (the two capping nodes, on no path).  Leaves coef and rvaf in the scratch (rvaf == coef when every
// coefficient is 0).
//
// contrib[i][c] = occurrences of node i on path c.  The rows of that matrix take few distinct values (a node
// lies on the reference only, on a variant path only, on both, ...): everything works on those PATTERNS —
// pattern q with its row, the number of nodes N_q that have it and the sum S_q of their counts — instead of on
// the ~500 nodes, and the patterns are read off the RUNS of the paths (a path is a few stretches of consecutive
// node indices): between two neighbouring run ends the row is constant.  Patterns are numbered in the order of
// their first node, as a scan over the nodes would find them; the all-zero row (nodes on none of the paths,
// the two capping nodes) adds exact zeros to every sum below and is left out.  S_q comes from running sums of
// the counts — integers, so exact in any order.  Returns the number of patterns (w.pat, w.pat_sum, w.pat_n).
size_t fit_patterns(Scratch& w, const std::vector<const Path*>& paths) {
  const size_t M = paths.size();
  std::vector<Scratch::Event>& ev = w.events;
  ev.clear();
  const bool have_runs = w.set_runs.size() == M;
  for (size_t c = 0; c < M; ++c) {
    if (have_runs && w.set_runs[c]) {                   // the stretches are known: no scan of the path
      const std::vector<int64_t>& rr = *w.set_runs[c];
      for (size_t i = 0; i + 1 < rr.size(); i += 2) {
        if (rr[i + 1] <= rr[i]) continue;
        ev.push_back(Scratch::Event{rr[i], (int32_t)c, 1});
        ev.push_back(Scratch::Event{rr[i + 1], (int32_t)c, -1});
      }
      continue;
    }
    const Path& p = *paths[c];
    const size_t n = p.size();
    const Node* pd = p.data();
    size_t i = 0;
    while (i < n) {
      size_t j = i + 1;
      while (j < n && pd[j] == pd[j - 1] + 1) ++j;
      ev.push_back(Scratch::Event{pd[i], (int32_t)c, 1});
      ev.push_back(Scratch::Event{pd[j - 1] + 1, (int32_t)c, -1});
      i = j;
    }
  }
  const size_t n_ev = ev.size();
  Scratch::Event* e = ev.data();
  for (size_t i = 1; i < n_ev; ++i) {                   // a handful of events: insertion sort by position (the order
    const Scratch::Event x = e[i];                      // of equal positions does not matter: they are applied together)
    size_t j = i;
    for (; j > 0 && e[j - 1].pos > x.pos; --j) e[j] = e[j - 1];
    e[j] = x;
  }
  // at most one pattern per gap between events
  w.pat.resize((n_ev + 1) * M);
  w.pat_sum.resize(n_ev + 1);
  w.pat_n.resize(n_ev + 1);
  w.row.assign(M, 0);
  int32_t* pat = w.pat.data();
  int32_t* row = w.row.data();
  double *pat_sum = w.pat_sum.data(), *pat_n = w.pat_n.data();
  // the sum of the float32 values of the counts of the nodes lo .. hi-1: integers, exact in any order
  const uint32_t* cnt = w.cnt;
  const bool wide = w.cnt_wide;
  auto stretch_sum = [&](int64_t lo, int64_t hi) -> uint64_t {
    uint64_t acc = 0;
    if (!wide) {
      for (int64_t i = lo; i < hi; ++i) acc += cnt[i];
    } else {
      for (int64_t i = lo; i < hi; ++i) { const uint32_t c = cnt[i]; acc += c < (1u << 24) ? (uint64_t)c : (uint64_t)(float)c; }
    }
    return acc;
  };
  size_t n_pat = 0;
  int64_t active = 0, prev = 0;
  size_t last = 0;                                      // neighbours mostly share their pattern
  w.seg.clear();
  for (size_t x = 0; x < n_ev;) {
    const int64_t pos = e[x].pos;
    if (active && pos > prev) {                         // nodes prev .. pos-1 have the row `row`
      w.seg.push_back(prev); w.seg.push_back(pos);
      auto same = [&](size_t q) {
        for (size_t c = 0; c < M; ++c) if (pat[q * M + c] != row[c]) return false;
        return true;
      };
      size_t q = last;
      if (!(q < n_pat && same(q)))
        for (q = 0; q < n_pat; ++q) if (same(q)) break;
      if (q == n_pat) {
        for (size_t c = 0; c < M; ++c) pat[q * M + c] = row[c];
        pat_sum[q] = 0.0; pat_n[q] = 0.0;
        ++n_pat;
      }
      pat_sum[q] += (double)stretch_sum(prev, pos);
      pat_n[q] += (double)(pos - prev);
      last = q;
    }
    for (; x < n_ev && e[x].pos == pos; ++x) { row[(size_t)e[x].c] += e[x].d; active += e[x].d; }
    prev = pos;
  }
  return n_pat;
}

// The numbers of the fit from the patterns.  FM = the number of paths when it is 2, 3 or 4 (the loops over paths
// then unroll over arrays on the stack; the operations and their order are those of the general form, FM = 0,
// so the results are the same bit for bit).
template <int FM>
void fit_solve(Scratch& w, const std::vector<const Path*>& paths, const size_t n_pat, int64_t n_total,
               const uint32_t* raw_counts) {
  const size_t M = FM ? (size_t)FM : paths.size();
  const int m = (int)M;
  constexpr size_t CAP = FM ? (size_t)FM : 1;
  double g_s[CAP * CAP], atb_s[CAP], ev_s[CAP * CAP], coef_s[CAP], grad_s[CAP];
  double *g = g_s, *atb = atb_s, *ev = ev_s, *coef = coef_s, *grad = grad_s;
  if (!FM) {
    w.g.resize(M * M); w.atb.resize(M); w.ev.resize(M * M); w.coef.resize(M); w.grad.resize(M);
    g = w.g.data(); atb = w.atb.data(); ev = w.ev.data(); coef = w.coef.data(); grad = w.grad.data();
  }
  const int32_t* pat = w.pat.data();
  const double *pat_sum = w.pat_sum.data(), *pat_n = w.pat_n.data();
  // Minimum-norm least squares with numpy's rcond=None cut-off (eps * max(n, m) * sigma_max on the singular
  // values).  A^T A = sum_q N_q pat_q pat_q^T and A^T b = sum_q S_q pat_q are sums of small integers times
  // counts, the singular values of A the square roots of the eigenvalues of the m x m matrix A^T A, its
  // eigenvectors the right singular vectors: coef = sum_j V_j (V_j . A^T b) / lambda_j over the kept j.
  // Squaring the condition number is harmless while the paths are well separated; a fit with a small or
  // vanishing singular value (near-identical or identical paths: a rank-deficient cluster) goes through
  // the one-sided Jacobi SVD of A itself, as every fit did before.
  for (size_t i = 0; i < M; ++i) coef[i] = 0.0;
  bool solved = false;
  {
    for (size_t i = 0; i < M * M; ++i) g[i] = 0.0;
    for (size_t i = 0; i < M; ++i) atb[i] = 0.0;
    for (size_t q = 0; q < n_pat; ++q)
      for (size_t r = 0; r < M; ++r) {
        atb[r] += (double)pat[q * M + r] * pat_sum[q];
        for (size_t c = 0; c < M; ++c) g[r * M + c] += pat_n[q] * (double)pat[q * M + r] * (double)pat[q * M + c];
      }
    // cyclic Jacobi on the symmetric g: g -> diag(lambda), ev columns = eigenvectors
    for (size_t i = 0; i < M * M; ++i) ev[i] = 0.0;
    for (size_t i = 0; i < M; ++i) ev[i * M + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
      double off = 0.0, diag = 0.0;
      for (size_t p2 = 0; p2 < M; ++p2) { diag += g[p2 * M + p2] * g[p2 * M + p2]; for (size_t q2 = p2 + 1; q2 < M; ++q2) off += g[p2 * M + q2] * g[p2 * M + q2]; }
      if (off <= 1e-30 * diag) break;
      for (size_t p2 = 0; p2 < M; ++p2)
        for (size_t q2 = p2 + 1; q2 < M; ++q2) {
          const double apq = g[p2 * M + q2];
          if (apq == 0.0) continue;
          const double theta = (g[q2 * M + q2] - g[p2 * M + p2]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
          const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
          for (size_t r = 0; r < M; ++r) {                // columns p2, q2
            const double gp = g[r * M + p2], gq = g[r * M + q2];
            g[r * M + p2] = c * gp - sn * gq;
            g[r * M + q2] = sn * gp + c * gq;
          }
          for (size_t r = 0; r < M; ++r) {                // rows p2, q2
            const double gp = g[p2 * M + r], gq = g[q2 * M + r];
            g[p2 * M + r] = c * gp - sn * gq;
            g[q2 * M + r] = sn * gp + c * gq;
          }
          for (size_t r = 0; r < M; ++r) {
            const double vp = ev[r * M + p2], vq = ev[r * M + q2];
            ev[r * M + p2] = c * vp - sn * vq;
            ev[r * M + q2] = sn * vp + c * vq;
          }
        }
    }
    double lmax = 0.0, lmin = std::numeric_limits<double>::infinity();
    for (size_t j = 0; j < M; ++j) { lmax = std::max(lmax, g[j * M + j]); lmin = std::min(lmin, g[j * M + j]); }
    if (lmax > 0 && lmin > 1e-9 * lmax) {                 // sigma_min > 3e-5 sigma_max: every singular value is kept
      for (size_t j = 0; j < M; ++j) {
        double proj = 0.0;
        for (size_t r = 0; r < M; ++r) proj += ev[r * M + j] * atb[r];
        proj /= g[j * M + j];
        for (size_t r = 0; r < M; ++r) coef[r] += ev[r * M + j] * proj;
      }
      solved = true;
    }
  }
  if (!solved) {
    // only over the nodes that lie on a path, in their order: the rows of every other node (the two capping nodes
    // among them) are zero and stay zero under the rotations — exact zeros in every sum below
    w.cmap.resize((size_t)n_total);
    size_t n_on = 0;
    for (size_t sgi = 0; sgi + 1 < w.seg.size(); sgi += 2)
      for (int64_t i = w.seg[sgi]; i < w.seg[sgi + 1]; ++i) w.cmap[(size_t)i] = (int32_t)n_on++;
    std::vector<std::vector<double>> u(M, std::vector<double>(n_on, 0.0));
    for (size_t c = 0; c < M; ++c)
      for (int64_t node : *paths[c]) u[c][(size_t)w.cmap[(size_t)node]] += 1.0;
    std::vector<float> counts(n_on);
    for (size_t sgi = 0; sgi + 1 < w.seg.size(); sgi += 2)
      for (int64_t i = w.seg[sgi]; i < w.seg[sgi + 1]; ++i) counts[(size_t)w.cmap[(size_t)i]] = (float)raw_counts[i];
    std::vector<double> v, sig;
    jacobi_svd(u, m, v, sig);
    double smax = 0.0;
    for (double sj : sig) smax = std::max(smax, sj);
    const double cutoff = std::numeric_limits<double>::epsilon() * (double)std::max<int64_t>(n_total, m) * smax;
    for (size_t j = 0; j < M; ++j) {
      const double sj = sig[j];
      // a singular value near the cut-off, or a kept one that small, leaves the answer to the
      // last bits of the solver: let the caller recompute this target with numpy
      if (smax > 0 && sj > 1e-14 * smax && sj < 1e-6 * smax) g_tie = true;
      if (!(sj > cutoff)) continue;
      double proj = 0.0;                              // (sigma_j u_j) . b / sigma_j^2
      for (size_t i = 0; i < n_on; ++i) proj += u[j][i] * (double)counts[i];
      proj /= sj * sj;
      for (size_t r = 0; r < M; ++r) coef[r] += v[r * M + j] * proj;
    }
  }
  for (size_t c = 0; c < M; ++c) if (coef[c] < 0) coef[c] = 0;
  // Projected gradient refinement (km/utils/PathQuant.py:111-142): grad_c = 2/n * sum_i (count_i - est_i) * contrib_ic
  // with est_i = sum_c contrib_ic * coef_c, step 0.1, until max |grad| <= 0.01, evaluated per pattern:
  // grad_c = 2/n * sum_q pat_qc * (S_q - N_q * est_q).  The same numbers in exact arithmetic; in floating point
  // the sums are grouped differently, which matters as little as the difference between two BLAS builds does
  // to the reference itself (printed values near a rounding tie are flagged and recomputed with numpy: err 100).
  for (size_t c = 0; c < M; ++c) grad[c] = 0.0;
  w.resid.resize(n_pat);
  double* resid = w.resid.data();
  double step = std::numeric_limits<double>::infinity();
  long iters = 0;
  while (step > 0.01) {
    for (size_t q = 0; q < n_pat; ++q) {
      double e = 0.0;
      for (size_t c = 0; c < M; ++c) e += (double)pat[q * M + c] * coef[c];
      resid[q] = pat_sum[q] - pat_n[q] * e;          // sum over the pattern's rows of (count - est)
    }
    for (size_t c = 0; c < M; ++c) {
      double sres = 0.0;
      for (size_t q = 0; q < n_pat; ++q) sres += 2.0 * resid[q] * (double)pat[q * M + c];
      grad[c] = sres / (double)n_total;
    }
    for (size_t c = 0; c < M; ++c) coef[c] += 0.1 * grad[c];
    for (size_t c = 0; c < M; ++c) if (coef[c] < 0) { grad[c] = 0; coef[c] = 0; }
    step = 0.0;
    bool nan = false;
    for (size_t c = 0; c < M; ++c) {
      if (grad[c] != grad[c]) nan = true;
      step = std::max(step, std::fabs(grad[c]));
    }
    ++iters;
    if (nan) break;                                 // np.max of a NaN is NaN; NaN > 0.01 is False
  }
  g_fit_iters += iters;
  double mx = -std::numeric_limits<double>::infinity(), sum = 0.0;
  for (size_t c = 0; c < M; ++c) { mx = std::max(mx, coef[c]); sum += coef[c]; }
  if (FM) w.coef.assign(coef, coef + M);
  w.rvaf.resize(M);
  if (mx == 0) w.rvaf = w.coef;
  else
    for (size_t i = 0; i < M; ++i) w.rvaf[i] = w.coef[i] / sum;
}

void fit_paths(Scratch& w, const std::vector<const Path*>& paths, int64_t n_total, const uint32_t* raw_counts) {
  const size_t n_pat = fit_patterns(w, paths);
  switch (paths.size()) {
    case 2: fit_solve<2>(w, paths, n_pat, n_total, raw_counts); break;
    case 3: fit_solve<3>(w, paths, n_pat, n_total, raw_counts); break;
    case 4: fit_solve<4>(w, paths, n_pat, n_total, raw_counts); break;
    default: fit_solve<0>(w, paths, n_pat, n_total, raw_counts); break;
  }
}

// Exact least-squares answers are ratios with small denominators (means of integer counts), so
// a printed value can sit exactly on a rounding tie (x.x5 for %.1f): which way it falls is
// then decided by the last-bit rounding errors of the solver.  Such rows are flagged and the
// caller recomputes that target with numpy, whose LAPACK path is the reference's.
inline void note_tie(double v, double scale) {
  if (!(v == v) || std::fabs(v) > 1e15) return;
  const double f = std::fabs(v) * scale;
  if (std::fabs(f - std::floor(f) - 0.5) < 1e-6) g_tie = true;
}

// "%.<decimals>f" % v for decimals 1 or 3, digit for digit what printf / Python print: the double is m * 2^e
// exactly, so round-half-even of m * 10^decimals * 2^e is integer arithmetic (printf is correctly rounded on the
// exact value as well).  Returns the number of characters, 0 when the value is out of the range handled here.
inline int fixed_digits(double v, int decimals, char* buf) {
  uint64_t bits;
  memcpy(&bits, &v, sizeof bits);
  const bool neg = (bits >> 63) != 0;
  const int ex = (int)((bits >> 52) & 0x7FF);
  uint64_t mant = bits & ((1ull << 52) - 1);
  if (ex == 0x7FF || ex >= 1023 + 50) return 0;          // nan, inf, >= 2^50: left to snprintf
  int e = ex - 1075;
  if (ex == 0) e = -1074; else mant |= 1ull << 52;
  const uint64_t scale = decimals == 3 ? 1000 : 10;
  const uint64_t P = mant * scale;                        // < 2^63
  uint64_t Q;
  if (e >= 0) {
    Q = P << e;
  } else {
    const int sh = -e;
    if (sh >= 64) {
      Q = 0;                                              // P < 2^63 <= half an ulp of the last printed digit
    } else {
      Q = P >> sh;
      const uint64_t rem = P & ((1ull << sh) - 1), half = 1ull << (sh - 1);
      if (rem > half || (rem == half && (Q & 1))) ++Q;
    }
  }
  char tmp[32];
  int n = 0;
  uint64_t frac = Q % scale, whole = Q / scale;
  for (int i = 0; i < decimals; ++i) { tmp[n++] = (char)('0' + frac % 10); frac /= 10; }
  tmp[n++] = '.';
  do { tmp[n++] = (char)('0' + whole % 10); whole /= 10; } while (whole);
  if (neg) tmp[n++] = '-';
  for (int i = 0; i < n; ++i) buf[i] = tmp[n - 1 - i];
  return n;
}

inline void put_float(std::string& out, const char* spec, double v) {
  if (v != v) { out += "nan"; return; }
  const int decimals = spec[2] == '3' ? 3 : 1;
  note_tie(v, decimals == 3 ? 1000.0 : 10.0);
  char buf[64];
  int n = fixed_digits(v, decimals, buf);
  if (n == 0) n = snprintf(buf, sizeof buf, spec, v);
  out.append(buf, (size_t)n);
}

inline void put_int(std::string& out, long long v) {
  char buf[24];
  char* e = buf + sizeof buf;
  char* p = e;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) *--p = '-';
  out.append(p, (size_t)(e - p));
}

// the sequence a path spells, appended to `out`
// (`runs`: the path's stretches of consecutive nodes as (first, last + 1) pairs, when the caller has them)
void put_spell(std::string& out, const Target& t, const Path& p, bool whole_first,
               const std::vector<int64_t>* runs = nullptr) {
  if (p.empty()) return;
  const size_t at = out.size();
  const size_t head = whole_first ? (size_t)t.k : 1;
  out.resize(at + head + p.size() - 1);
  char* o = &out[at];
  if (whole_first) {
    if (t.plain && !t.kmers && p[0] < t.n_ref) {
      memcpy(o, t.seq + p[0], (size_t)t.k);
    } else {
      uint64_t kmer = kmer_of(t, p[0]);
      for (int i = t.k - 1; i >= 0; --i) { o[i] = LAST[kmer & 3]; kmer >>= 2; }
    }
  } else {
    o[0] = tail_of(t, p[0]);
  }
  o += head;
  const size_t n = p.size();
  if (t.kmers) {
    for (size_t i = 1; i < n; ++i) *o++ = LAST[t.kmers[p[i]] & 3];
    return;
  }
  auto own = [&](int64_t lo, int64_t hi) {                   // the last bases of the target's own k-mers lo .. hi-1
    const char* src = t.seq + lo + t.k - 1;
    const size_t len = (size_t)(hi - lo);
    if (t.plain) memcpy(o, src, len);
    else
      for (size_t q = 0; q < len; ++q) o[q] = LAST[base_code(src[q])];
    o += len;
  };
  if (runs) {
    bool head_done = false;                                 // p[0] is spelled already
    for (size_t q = 0; q + 1 < runs->size(); q += 2) {
      int64_t lo = (*runs)[q];
      const int64_t hi = (*runs)[q + 1];
      if (hi <= lo) continue;
      if (!head_done) { ++lo; head_done = true; }
      const int64_t mid = std::min(hi, std::max(lo, t.n_ref));
      if (lo < mid) own(lo, mid);
      for (int64_t node = mid; node < hi; ++node) *o++ = LAST[t.extra[node - t.n_ref] & 3];
    }
    return;
  }
  for (size_t i = 1; i < n;) {
    const int64_t node = p[i];
    if (node >= t.n_ref) { *o++ = LAST[t.extra[node - t.n_ref] & 3]; ++i; continue; }
    size_t j = i + 1;                                       // a stretch of the target's own consecutive k-mers
    while (j < n && p[j] == p[j - 1] + 1 && p[j] < t.n_ref) ++j;
    own(node, node + (int64_t)(j - i));
    i = j;
  }
}

// 0 ok, 1 IndexError, 2 "mutation identification could be incorrect", 3 assertion
// (`known`: split_paths(ref, alt) when the caller has it already)
int name_variant(Scratch& w, const Target& t, const Path& ref, const Path& alt, int64_t offset, std::string* out,
                 const Split* known = nullptr) {
  Split sp;
  if (known) sp = *known;
  else if (!split_paths(ref, alt, t.k, &sp)) return 1;
  // ref[start:end_ref] and alt[start:end_var] as Python slices them
  auto clamp = [](int64_t n, int64_t lo, int64_t hi, int64_t* a, int64_t* b) {
    if (lo < 0) lo = std::max<int64_t>(0, lo + n);
    if (hi < 0) hi = std::max<int64_t>(0, hi + n);
    lo = std::min(lo, n);
    hi = std::min(hi, n);
    *a = lo;
    *b = std::max(lo, hi);
  };
  int64_t r0, r1, a0, a1;
  clamp((int64_t)ref.size(), sp.start, sp.end_ref, &r0, &r1);
  clamp((int64_t)alt.size(), sp.start, sp.end_var, &a0, &a1);
  if ((int64_t)ref.size() - (r1 - r0) + (a1 - a0) != (int64_t)alt.size()) return 2;
  std::string &gone = w.gone, &neu = w.neu;
  gone.resize((size_t)(r1 - r0));
  for (int64_t i = r0; i < r1; ++i) gone[(size_t)(i - r0)] = tail_of(t, ref[(size_t)i]);
  neu.resize((size_t)(a1 - a0));
  for (int64_t i = a0; i < a1; ++i) neu[(size_t)(i - a0)] = tail_of(t, alt[(size_t)i]);
  if (!gone.empty()) {
    if (gone == neu) return 3;
    // while gone[-(cut+1):] == neu[-(cut+1):]: cut += 1 — with gone != neu that is the length of the common suffix
    // (once cut + 1 exceeds the shorter string the two slices differ in length)
    size_t cut = 0;
    const size_t lim = std::min(gone.size(), neu.size());
    while (cut < lim && gone[gone.size() - 1 - cut] == neu[neu.size() - 1 - cut]) ++cut;
    if (cut) {
      gone.resize(cut >= gone.size() ? 0 : gone.size() - cut);
      neu.resize(cut >= neu.size() ? 0 : neu.size() - cut);
    }
  }
  const char* kind;
  if (sp.end_ref == sp.end_var) kind = sp.start == sp.end_ref ? "Reference" : "Substitution";
  else if (sp.start == sp.end_ovl) kind = "ITD";
  else if (sp.end_ref < sp.end_var) kind = gone.empty() ? "Insertion" : "Indel";
  else kind = neu.empty() ? "Deletion" : "Indel";
  if (!strcmp(kind, "Reference")) { *out = "Reference\t"; return 0; }
  for (char& c : gone) c = (char)tolower((unsigned char)c);
  out->clear();
  *out += kind; *out += '\t';
  put_int(*out, (long long)(sp.start + t.k + offset)); *out += ':';
  *out += gone; *out += '/'; *out += neu; *out += ':';
  put_int(*out, (long long)(sp.end_ref + 1 + offset));
  return 0;
}

// One TSV row (no newline) appended to `out`; `seq` is either given or spelled from `seq_path`.
void put_row(std::string& out, const char* db, const char* query, const std::string& name, double rvaf, double expr,
             long long min_cov, long long off, const Target& t, const Path* seq_path, const char* seq, size_t seq_len,
             double ref_expr, const char* ref_seq, size_t ref_len, const char* note,
             const std::vector<int64_t>* seq_runs = nullptr) {
  out += db; out += '\t'; out += query; out += '\t'; out += name; out += '\t';
  put_float(out, "%.3f", rvaf); out += '\t'; put_float(out, "%.1f", expr); out += '\t';
  put_int(out, min_cov); out += '\t'; put_int(out, off); out += '\t';
  if (seq_path) put_spell(out, t, *seq_path, true, seq_runs);
  else out.append(seq, seq_len);
  out += '\t'; put_float(out, "%.1f", ref_expr); out += '\t'; out.append(ref_seq, ref_len); out += '\t'; out += note;
}

// ---- the sort key of km_amd/report.py: row_key ------------------------------------------
// natural(text) = re.split("([0-9]+)", text) with the digit runs converted to int and the rest lower-cased;
// two such lists compare element by element (text, int, text, ... on both sides), the shorter one first when
// one is a prefix of the other.  Done here on the two strings directly.
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

int cmp_nat(const char* a, size_t na, const char* b, size_t nb) {
  size_t i = 0, j = 0;
  while (true) {
    size_t ie = i, je = j;
    while (ie < na && !is_digit(a[ie])) ++ie;
    while (je < nb && !is_digit(b[je])) ++je;
    const size_t la = ie - i, lb = je - j, n = std::min(la, lb);
    for (size_t q = 0; q < n; ++q) {
      const unsigned char ca = (unsigned char)tolower((unsigned char)a[i + q]), cb = (unsigned char)tolower((unsigned char)b[j + q]);
      if (ca != cb) return ca < cb ? -1 : 1;
    }
    if (la != lb) return la < lb ? -1 : 1;
    i = ie; j = je;
    const bool ea = i >= na, eb = j >= nb;
    if (ea || eb) return ea && eb ? 0 : (ea ? -1 : 1);      // the list that ends here is the shorter one
    while (ie < na && is_digit(a[ie])) ++ie;
    while (je < nb && is_digit(b[je])) ++je;
    while (i < ie - 1 && a[i] == '0') ++i;                  // integers: compare the values
    while (j < je - 1 && b[j] == '0') ++j;
    if (ie - i != je - j) return ie - i < je - j ? -1 : 1;
    const int c = memcmp(a + i, b + j, ie - i);
    if (c) return c < 0 ? -1 : 1;
    i = ie; j = je;                                         // a text element follows on both sides (maybe "")
  }
}

// key = [natural(w) for w in info.split(" ")] + [natural(query), natural(variant), natural(type), natural(min_cov)],
// the first component descending.  All rows of a target share the query.
bool row_less(const RowRec& a, const RowRec& b) {
  struct View { const char* p; size_t n; };
  auto comps = [](const RowRec& r, View* v) -> int {
    int n = 0;
    size_t p = 0;
    while (true) {
      const size_t q = r.note.find(' ', p);
      if (q == std::string::npos) { if (n < 12) v[n++] = View{r.note.data() + p, r.note.size() - p}; break; }
      if (n < 12) v[n++] = View{r.note.data() + p, q - p};
      p = q + 1;
    }
    const size_t tab = r.name.find('\t');
    v[n++] = View{"", 0};                                   // the query: equal on both sides
    v[n++] = View{r.name.data() + tab + 1, r.name.size() - tab - 1};
    v[n++] = View{r.name.data(), tab};
    v[n++] = View{r.mincov.data(), r.mincov.size()};
    return n;
  };
  View va[16], vb[16];
  const int na = comps(a, va), nb = comps(b, vb);
  const int n = std::min(na, nb);
  for (int i = 0; i < n; ++i) {
    int c = cmp_nat(va[i].p, va[i].n, vb[i].p, vb[i].n);
    if (i == 0) c = -c;                           // the first component sorts descending
    if (c) return c < 0;
  }
  return na < nb;
}

// km_amd/report.py: cluster_groups
// (the groups land in w.groups[0 .. w.n_groups): the vectors keep their capacity from target to target)
void cluster_groups(const std::vector<Split>& diffs, Scratch& w) {
  std::vector<int>& todo = w.todo;
  todo.resize(diffs.size());
  for (size_t i = 0; i < diffs.size(); ++i) todo[i] = (int)i;
  w.n_groups = 0;
  auto first_overlap = [&](int64_t lo, int64_t hi) -> int {
    for (int v : todo) {
      const int64_t s = diffs[(size_t)v].start, e = diffs[(size_t)v].end_ref;
      if (e < lo || s > hi) continue;
      if (lo == hi && hi == s && s == e) continue;          // terminal ITD, ignored in cluster mode
      if (hi == e && (lo == hi || s == e)) continue;        // quasi-terminal ITD
      return v;
    }
    return -1;
  };
  while (!todo.empty()) {
    const int seed = todo.front();
    todo.erase(todo.begin());
    if (w.n_groups == w.groups.size()) w.groups.emplace_back();
    Group& g = w.groups[w.n_groups++];
    g.lo = diffs[(size_t)seed].start;
    g.hi = diffs[(size_t)seed].end_ref;
    g.members.clear();
    g.members.push_back(seed);
    int v = first_overlap(g.lo, g.hi);
    while (v != -1) {
      todo.erase(std::find(todo.begin(), todo.end(), v));
      g.members.push_back(v);
      g.lo = std::min(g.lo, diffs[(size_t)v].start);
      g.hi = std::max(g.hi, diffs[(size_t)v].end_ref);
      v = first_overlap(g.lo, g.hi);
    }
  }
}

inline RowRec& new_row(Scratch& w, const std::string& out) {
  if (w.n_rows == w.rows.size()) w.rows.emplace_back();
  RowRec& r = w.rows[w.n_rows++];
  r.off = out.size();
  return r;
}

// km_amd/report.py: target_rows.  Appends the target's rows (each newline-terminated, in the reference's order)
// to `out`; 0 ok, else the error code of name_variant / split_paths (then `out` is left as it was).
int target_rows(Scratch& w, const Target& t, const char* db, std::string& out) {
  const int k = t.k;
  const int64_t n_ref = t.n_ref, n_total = t.n_nodes + 2;
  const size_t out0 = out.size();
  const double nan = std::numeric_limits<double>::quiet_NaN();
  const size_t ref_len = std::min<size_t>(t.seq_len, (size_t)(n_ref + k - 1));
  static const std::string REFERENCE = "Reference\t";
  if (!t.counts) {
    // lean delivery of a bare-reference target: its single row needs the path's min coverage and
    // whether every count is 0 (then PathQuant's rVAF aliases coef and both print nan)
    const double expr0 = t.ref_max == 0 ? nan : -1.0;
    if (t.n_paths != 1 || !t.is_ref[0]) return 3;
    put_row(out, db, t.name, REFERENCE, nan, expr0, t.min_cov[0], 0, t, nullptr, t.seq, ref_len, expr0, t.seq, ref_len, "vs_ref");
    out.push_back('\n');
    return 0;
  }
  uint32_t ref_max = 0;
  {
    // the reference fits float32 counts (PathQuant.py:99): the fits sum exactly those values — the counts
    // themselves below 2^24 (every count of a real database), the nearest float32 above
    const uint32_t* cnt = t.counts;
    uint32_t all_max = 0;
    for (int64_t i = 0; i < n_ref; ++i) ref_max = std::max(ref_max, cnt[i]);
    for (int64_t i = n_ref; i < t.n_nodes; ++i) all_max = std::max(all_max, cnt[i]);
    all_max = std::max(all_max, ref_max);
    w.cnt = cnt;
    w.cnt_wide = all_max >= (1u << 24);
  }
  Path& ref = w.ref;                                     // 0 .. n_ref-1: what is there from the last target stays
  {
    const size_t have = ref.size();
    ref.resize((size_t)n_ref);
    for (size_t i = have; i < (size_t)n_ref; ++i) ref[i] = (Node)i;
    w.ref_run.clear();
    if (n_ref > 0) { w.ref_run.push_back(0); w.ref_run.push_back(n_ref); }
  }
  const double ref_expr = ref_max == 0 ? nan : -1.0;
  if (t.n_paths == 1 && t.is_ref[0]) {
    put_row(out, db, t.name, REFERENCE, nan, ref_expr, t.min_cov[0], 0, t, nullptr, t.seq, ref_len, ref_expr, t.seq, ref_len, "vs_ref");
    out.push_back('\n');
    return 0;
  }
  w.n_rows = 0;
  auto fail = [&](int rc) { out.resize(out0); return rc; };
  auto close_row = [&](RowRec& r, const std::string& name, long long mc, const char* note) {
    r.len = out.size() - r.off;
    out.push_back('\n');                                  // every row is terminated: blocks concatenate into the TSV
    r.name = name;
    r.mincov.clear(); put_int(r.mincov, mc);
    r.note = note;
  };
  std::string& name = w.name;
  w.diffs.resize(t.n_paths);
  for (size_t pi = 0; pi < t.n_paths; ++pi) {
    const Path& p = t.paths[pi];
    if (t.is_ref[pi]) {
      RowRec& r = new_row(w, out);
      put_row(out, db, t.name, REFERENCE, nan, ref_expr, t.min_cov[pi], 0, t, nullptr, t.seq, ref_len, ref_expr, t.seq, ref_len, "vs_ref");
      close_row(r, REFERENCE, t.min_cov[pi], "vs_ref");
      continue;
    }
    w.set.clear(); w.set.push_back(&p); w.set.push_back(&ref);
    w.set_runs.clear(); w.set_runs.push_back(&w.path_runs[pi]); w.set_runs.push_back(&w.ref_run);
    fit_paths(w, w.set, n_total, t.counts);
    if (!split_paths(ref, p, k, &w.diffs[pi], &w.path_runs[pi])) return fail(1);   // (kept for the clusters below)
    const int rc = name_variant(w, t, ref, p, 0, &name, &w.diffs[pi]);
    if (rc) return fail(rc);
    RowRec& r = new_row(w, out);
    put_row(out, db, t.name, name, w.rvaf[0], w.coef[0], t.min_cov[pi], 0, t, &p, nullptr, 0, w.coef[1], t.seq, ref_len, "vs_ref",
            &w.path_runs[pi]);
    close_row(r, name, t.min_cov[pi], "vs_ref");
  }
  if (t.n_paths) {
    std::vector<Split>& diffs = w.diffs;
    for (size_t pi = 0; pi < t.n_paths; ++pi)              // the reference against itself: nothing differs
      if (t.is_ref[pi]) diffs[pi] = Split{n_ref, n_ref, n_ref, n_ref};
    cluster_groups(diffs, w);
    int num = 0;
    for (size_t gi = 0; gi < w.n_groups; ++gi) {
      const Group& g = w.groups[gi];
      if (g.members.size() == 1 && t.is_ref[(size_t)g.members[0]]) continue;
      ++num;
      int64_t size = 0;
      for (int v : g.members)
        size = std::max<int64_t>(size, std::llabs(diffs[(size_t)v].end_var - diffs[(size_t)v].end_ref + 1));
      const int64_t off = std::max<int64_t>(0, g.lo - size);
      Path& cref = w.cref;
      slice(ref, off, g.hi, &cref);
      if (w.clipped.size() < g.members.size()) w.clipped.resize(g.members.size());
      const size_t n_clip = g.members.size();
      for (size_t q = 0; q < n_clip; ++q) {
        const int v = g.members[q];
        // (a path that IS the reference was not laid out: `ref` holds the same nodes)
        slice(t.is_ref[(size_t)v] ? ref : t.paths[(size_t)v], off, diffs[(size_t)v].end_var + g.hi - diffs[(size_t)v].end_ref, &w.clipped[q]);
      }
      w.set.clear();
      w.set.push_back(&cref);
      for (size_t q = 0; q < n_clip; ++q) w.set.push_back(&w.clipped[q]);
      w.cref_run.clear();                                  // a slice of 0 .. n_ref-1
      if (!cref.empty()) { w.cref_run.push_back(cref.front()); w.cref_run.push_back(cref.back() + 1); }
      w.set_runs.assign(w.set.size(), nullptr);
      w.set_runs[0] = &w.cref_run;
      fit_paths(w, w.set, n_total, t.counts);
      std::string& cref_seq = w.cref_seq;                  // the cluster's reference sequence
      cref_seq.clear();
      put_spell(cref_seq, t, cref, true);
      char note[64];
      snprintf(note, sizeof note, "cluster %d n=%d", num, (int)n_clip);
      for (size_t q = 0; q < n_clip; ++q) {
        const Path& p = w.clipped[q];
        if (p.empty()) return fail(4);                     // min() of an empty sequence
        uint32_t mc = 0xFFFFFFFFu;
        for (int64_t node : p) mc = std::min(mc, t.counts[node]);
        const int rc = name_variant(w, t, cref, p, off, &name);
        if (rc) return fail(rc);
        RowRec& r = new_row(w, out);
        put_row(out, db, t.name, name, w.rvaf[q + 1], w.coef[q + 1], mc, off, t, &p, nullptr, 0, w.coef[0],
                cref_seq.data(), cref_seq.size(), note);
        close_row(r, name, mc, note);
      }
    }
  }
  // the reference's row order; rows were appended in the order they were made
  const size_t nr = w.n_rows;
  w.order.resize(nr);
  for (size_t i = 0; i < nr; ++i) w.order[i] = i;
  if (nr > 1) std::stable_sort(w.order.begin(), w.order.end(), [&](size_t a, size_t b) { return row_less(w.rows[a], w.rows[b]); });
  bool in_place = true;
  for (size_t i = 0; i < nr; ++i) in_place = in_place && w.order[i] == i;
  if (in_place) return 0;                                 // they were made in that order
  w.tmp.clear();
  for (size_t i = 0; i < nr; ++i) {
    const RowRec& r = w.rows[w.order[i]];
    w.tmp.append(out, r.off, r.len + 1);
  }
  out.resize(out0);
  out += w.tmp;
  return 0;
}

// A freed text buffer is kept for the next call (two at most): a 10 000-target batch prints ~15 MB, and a fresh
// allocation of that size is a fresh mapping whose every page faults on first touch, on every call.
struct TextCache {
  std::mutex mu;
  char* buf[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  ~TextCache() { free(buf[0]); free(buf[1]); }
};
TextCache g_text_cache;
constexpr size_t TEXT_HEADER = 16;       // the capacity sits in front of the text handed out

char* text_alloc(size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_text_cache.mu);
    for (int i = 0; i < 2; ++i)
      if (g_text_cache.buf[i] && g_text_cache.cap[i] >= bytes) {
        char* p = g_text_cache.buf[i];
        g_text_cache.buf[i] = nullptr;
        return p + TEXT_HEADER;
      }
  }
  const size_t cap = bytes + bytes / 8 + 4096;
  char* p = (char*)malloc(cap + TEXT_HEADER);
  if (!p) return nullptr;
  memcpy(p, &cap, sizeof cap);
  return p + TEXT_HEADER;
}

void text_free(char* text) {
  if (!text) return;
  char* p = text - TEXT_HEADER;
  size_t cap;
  memcpy(&cap, p, sizeof cap);
  {
    std::lock_guard<std::mutex> lk(g_text_cache.mu);
    for (int i = 0; i < 2; ++i)
      if (!g_text_cache.buf[i]) { g_text_cache.buf[i] = p; g_text_cache.cap[i] = cap; return; }
    // both places taken: keep the larger buffers
    const int small = g_text_cache.cap[0] <= g_text_cache.cap[1] ? 0 : 1;
    if (g_text_cache.cap[small] < cap) { std::swap(g_text_cache.buf[small], p); g_text_cache.cap[small] = cap; }
  }
  free(p);
}

// ---- the team of worker threads, kept between calls.  A consumer reports batch after batch (tools/kmclient.cpp: one
// call per 10 000 targets every 2 ms): starting sixteen threads and faulting in sixteen fresh megabyte buffers per call
// was a fifth of a call.  One team per process, taken by one call at a time (a second concurrent call starts
// threads of its own, as every call used to); the workers' row buffers keep their capacity.
struct Team {
  std::mutex in_use;                         // held by the call that runs on the team
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> threads;          // workers 1 .. (the caller is worker 0)
  std::vector<std::string> buf;              // per worker, reused
  std::function<void(unsigned)> job;
  unsigned n_active = 0, pending = 0;
  unsigned long long generation = 0;
  bool quit = false;
  // A new worker starts on the CPU of the thread that made it, and a thread that only ever runs for a fraction of
  // a millisecond at a time is not worth moving to the scheduler's periodic balancing: measured, a whole team sat
  // on ONE CPU for hundreds of calls (every call then took the single-thread time), in about half of all process
  // starts.  So each worker moves itself once, when it starts, to its own CPU of those the process may use
  // (counted on from the CPU of its maker), and then takes the full mask back: a first placement, not a pin.
  // KM_REPORT_SPREAD=0 leaves it to the scheduler.
  static void spread(unsigned me, int maker_cpu) {
    int stride = 1;                                     // KM_REPORT_SPREAD=s: every s-th allowed CPU (0: leave it to the scheduler)
    if (const char* e = getenv("KM_REPORT_SPREAD")) { stride = atoi(e); if (stride <= 0) return; }
    cpu_set_t all;
    CPU_ZERO(&all);
    if (sched_getaffinity(0, sizeof all, &all) != 0) return;
    int allowed[CPU_SETSIZE], n = 0, at = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &all)) { if (c == maker_cpu) at = n; allowed[n++] = c; }
    if (n < 2) return;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(allowed[(int)(((long)at + (long)me * stride) % n)], &one);
    if (sched_setaffinity(0, sizeof one, &one) == 0) sched_setaffinity(0, sizeof all, &all);
  }
  void loop(unsigned me, int maker_cpu) {
    spread(me, maker_cpu);
    unsigned long long seen = 0;
    for (;;) {
      std::function<void(unsigned)> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return quit || (generation != seen && me < n_active); });
        if (quit) return;
        seen = generation;
        f = job;
      }
      f(me);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  // run job(0 .. n - 1), job(0) on the calling thread
  void run(unsigned n, const std::function<void(unsigned)>& f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      while (threads.size() + 1 < n) { const unsigned me = (unsigned)threads.size() + 1; const int cpu = sched_getcpu(); threads.emplace_back([this, me, cpu] { loop(me, cpu); }); }
      job = f;
      n_active = n;
      pending = n - 1;
      ++generation;
    }
    cv_go.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
    n_active = 0;
  }
  ~Team() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_go.notify_all();
    for (std::thread& th : threads) th.join();
  }
};
Team& team() {
  static Team* t = new Team;                 // (never destroyed: its threads must not outlive a static's destructor at exit)
  return *t;
}

}  // namespace

extern "C" int km_report_rows(const km_report_in_t* in, char** text_out, uint64_t** row_off_out,
                              int32_t** err_out) {
  if (!in || !text_out || !row_off_out || !err_out || !in->res) return KM_E_ARG;
  const km_batch_out_t& r = *in->res;
  // counts: 32-bit, or 16-bit + the list of the exact counts >= 65535 (KM_DELIVER_COUNT16; the list's length is
  // in the sizes, so those are required then)
  const bool c16 = !r.node_count && r.node_count16;
  const uint32_t n_esc = c16 && in->sizes ? in->sizes->n_count_escapes : 0;
  if (c16 && (!in->sizes || (n_esc && (!r.count_esc_node || !r.count_esc_value)))) return KM_E_ARG;
  if (!r.status || !r.n_ref || !r.node_off || (!r.node_kmer && (!r.extra_off || !r.extra_kmer)) ||
      (!r.node_count && !r.node_count16) || !r.path_off || !r.run_off ||
      !r.run_start || !r.run_len || !r.path_min_cov || (in->n_targets && (!in->bases || !in->base_off || !in->names)))
    return KM_E_ARG;
  if (in->k < 2 || in->k > 32) return KM_E_K;
  const uint32_t n = in->n_targets;
  // The view's offset arrays against the lengths of what they index, when the caller gave them.
  // (Round 2 lost a test process to a heap overwrite in fit_paths: a work-in-progress delivery handed
  // over a target with variant paths but no counts, and contrib[path node] was written past a
  // two-element vector.  The per-target checks below make such a view an error code.)
  if (in->sizes) {
    const km_batch_sizes_t& z = *in->sizes;
    if (z.n_targets != n) return KM_E_ARG;
    if (n) {
      if (r.node_off[0] != 0 || r.node_off[n] != z.n_nodes || r.path_off[0] != 0 || r.path_off[n] != z.n_paths) return KM_E_ARG;
      if (r.extra_off && (r.extra_off[0] != 0 || r.extra_off[n] != z.n_extra)) return KM_E_ARG;
      if (r.run_off[0] != 0 || r.run_off[z.n_paths] != z.n_runs) return KM_E_ARG;
    }
    for (uint32_t t = 0; t < n; ++t) {
      if (r.node_off[t + 1] < r.node_off[t] || r.path_off[t + 1] < r.path_off[t]) return KM_E_ARG;
      if (r.extra_off && r.extra_off[t + 1] < r.extra_off[t]) return KM_E_ARG;
    }
    for (uint32_t p = 0; p < z.n_paths; ++p)
      if (r.run_off[p + 1] < r.run_off[p]) return KM_E_ARG;
    for (uint32_t q = 0; q < n_esc; ++q)
      if (r.count_esc_node[q] >= z.n_nodes || (q && r.count_esc_node[q] <= r.count_esc_node[q - 1])) return KM_E_ARG;
  }
  uint64_t* row_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
  int32_t* err = (int32_t*)malloc(sizeof(int32_t) * std::max<size_t>(1, n));
  if (!row_off || !err) { free(row_off); free(err); return KM_E_NOMEM; }
  try {
    // Targets are independent.  A team of host threads (KM_REPORT_THREADS, default min(cores, 16)) pulls CHUNKS of
    // consecutive targets off a shared counter; a worker appends the rows of its chunks to ONE buffer of its own
    // and notes, per target, how many bytes it wrote (row_off[t + 1] for now).  When every chunk is done the
    // first worker turns the lengths into offsets and takes the text buffer, then the same team copies the
    // chunks into place.  One team, no per-target strings.
    const bool trace = getenv("KM_TRACE_HOST") != nullptr;
    constexpr uint32_t CHUNK = 32;
    const uint32_t n_chunks = (n + CHUNK - 1) / CHUNK;
    struct ChunkRec { uint32_t worker; size_t at; };
    struct WorkerTrace { double first_us = 0, last_us = 0; uint32_t chunks = 0; int cpu = -1; char pad[40]; };
    std::vector<WorkerTrace> wtrace;                      // KM_TRACE_HOST: when each worker started and ended, on which CPU
    std::vector<ChunkRec> chunk(n_chunks);
    unsigned n_thr = std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    if (const char* e = getenv("KM_REPORT_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) n_thr = (unsigned)v; }
    n_thr = std::min<unsigned>(n_thr, std::max<uint32_t>(1, n / 64));     // small batches: no threads
    // the team's buffers when this call gets the team, buffers of its own otherwise
    std::unique_lock<std::mutex> team_lock(team().in_use, std::try_to_lock);
    const bool on_team = team_lock.owns_lock() && n_thr > 1;
    std::vector<std::string> own_buf;
    if (on_team) { if (team().buf.size() < n_thr) team().buf.resize(n_thr); }
    else own_buf.resize(n_thr);
    std::vector<std::string>& wbuf = on_team ? team().buf : own_buf;
    if (trace) wtrace.resize(n_thr);
    std::atomic<uint32_t> next(0), next_copy(0), arrived(0);
    std::atomic<bool> failed(false);
    std::atomic<int> phase(0);            // 1: offsets and the text buffer are ready, -1: allocation failed
    char* text = nullptr;
    const char* db = in->db_name ? in->db_name : "";
    const auto tr0 = std::chrono::steady_clock::now();
    std::chrono::steady_clock::time_point tr1 = tr0;
    auto work = [&](unsigned me) {
      try {
        Scratch w;
        std::string& out = wbuf[me];
        out.clear();
        out.reserve((size_t)n * 1600 / n_thr + 65536);
        if (trace) { wtrace[me].first_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); wtrace[me].cpu = sched_getcpu(); }
        for (uint32_t c = next.fetch_add(1); c < n_chunks; c = next.fetch_add(1)) {
          if (trace) ++wtrace[me].chunks;
          chunk[c] = ChunkRec{me, out.size()};
          for (uint32_t ti = c * CHUNK; ti < std::min<uint32_t>(n, (c + 1) * CHUNK); ++ti) {
            err[ti] = 0;
            const size_t before = out.size();
            auto done = [&]() { row_off[ti + 1] = out.size() - before; };
            row_off[ti + 1] = 0;
            if (r.status[ti] != KM_T_OK) continue;
            Target t;
            t.plain = false;
            t.name = in->names[ti] ? in->names[ti] : "";
            t.seq = (const char*)in->bases + in->base_off[ti];
            t.seq_len = (size_t)(in->base_off[ti + 1] - in->base_off[ti]);
            t.k = in->k;
            t.n_ref = r.n_ref[ti];
            t.kmers = r.node_kmer ? r.node_kmer + r.node_off[ti] : nullptr;
            t.extra = r.node_kmer ? nullptr : r.extra_kmer + r.extra_off[ti];
            t.counts = r.node_count ? r.node_count + r.node_off[ti] : nullptr;
            // ---- this target's slice of the view must hang together before any of it is used
            if (r.node_off[ti + 1] < r.node_off[ti] || r.path_off[ti + 1] < r.path_off[ti] ||
                in->base_off[ti + 1] < in->base_off[ti]) { err[ti] = 5; continue; }
            t.n_nodes = (int64_t)(r.node_off[ti + 1] - r.node_off[ti]);
            t.ref_max = 0;
            const bool lean = t.n_nodes == 0 && t.n_ref > 0;
            if (lean) {
              if (!r.ref_max_cov || r.ref_max_cov[ti] == 0xFFFFFFFFu) { err[ti] = 5; continue; }   // counts missing
              t.counts = nullptr;                      // bare-reference target, delivered lean
              t.ref_max = r.ref_max_cov[ti];
              t.n_nodes = t.n_ref;
            }
            if (t.n_nodes < t.n_ref || (int64_t)t.seq_len < t.n_ref + t.k - 1) { err[ti] = 5; continue; }
            if (c16 && !lean) {
              // 16-bit counts: this target's as 32-bit values, the exact ones patched in from the escape list
              const uint64_t n0 = r.node_off[ti], n1 = r.node_off[ti + 1];
              w.counts32.resize((size_t)t.n_nodes);
              for (int64_t i = 0; i < t.n_nodes; ++i) w.counts32[(size_t)i] = r.node_count16[n0 + (uint64_t)i];
              if (n_esc) {
                const uint64_t* e0 = r.count_esc_node;
                for (const uint64_t* e = std::lower_bound(e0, e0 + n_esc, n0); e < e0 + n_esc && *e < n1; ++e)
                  w.counts32[(size_t)(*e - n0)] = r.count_esc_value[e - e0];
              }
              t.counts = w.counts32.data();
            }
            if (!r.node_kmer) {
              if (r.extra_off[ti + 1] < r.extra_off[ti] ||
                  (int64_t)(r.extra_off[ti + 1] - r.extra_off[ti]) != t.n_nodes - t.n_ref) { err[ti] = 5; continue; }
            }
            const uint32_t p0 = r.path_off[ti], p1 = r.path_off[ti + 1];
            if (lean && p1 == p0 + 1 && r.run_off[p0 + 1] == r.run_off[p0] + 1 && r.run_start[r.run_off[p0]] == 0 &&
                (int64_t)r.run_len[r.run_off[p0]] == t.n_ref) {
              // the common case by far — a bare-reference target delivered lean, its one path the single run
              // 0 .. n_ref-1: its one row needs neither the path's nodes nor any count
              static const std::string REFERENCE = "Reference\t";
              const double nan0 = std::numeric_limits<double>::quiet_NaN();
              const double expr0 = t.ref_max == 0 ? nan0 : -1.0;
              const size_t len0 = (size_t)(t.n_ref + t.k - 1);
              put_row(out, db, t.name, REFERENCE, nan0, expr0, r.path_min_cov[p0], 0, t, nullptr, t.seq, len0, expr0,
                      t.seq, len0, "vs_ref");
              out.push_back('\n');
              done();
              continue;
            }
            const size_t n_paths = p1 - p0;
            if (w.paths.size() < n_paths) w.paths.resize(n_paths);
            t.paths = w.paths.data();
            t.n_paths = n_paths;
            w.min_cov.assign(r.path_min_cov + p0, r.path_min_cov + p1);
            t.min_cov = w.min_cov.data();
            w.is_ref.resize(n_paths);
            t.is_ref = w.is_ref.data();
            if (w.path_runs.size() < n_paths) w.path_runs.resize(n_paths);
            bool consistent = true;
            for (uint32_t p = p0; p < p1 && consistent; ++p) {
              Path& path = w.paths[p - p0];
              std::vector<int64_t>& runs = w.path_runs[p - p0];
              runs.clear();
              if (r.run_off[p + 1] < r.run_off[p]) { consistent = false; break; }
              size_t total = 0;
              for (uint64_t q = r.run_off[p]; q < r.run_off[p + 1]; ++q) {
                // every node of a path is one of this target's nodes (fit_paths indexes by it)
                if ((int64_t)r.run_start[q] + (int64_t)r.run_len[q] > t.n_nodes) { consistent = false; break; }
                total += r.run_len[q];
              }
              if (!consistent) break;
              bool chain = true;                           // every run starts where the one before it ended, from 0
              size_t at = 0;
              for (uint64_t q = r.run_off[p]; q < r.run_off[p + 1]; ++q) {
                chain = chain && (size_t)r.run_start[q] == at;
                const int64_t first = (int64_t)r.run_start[q];
                runs.push_back(first); runs.push_back(first + r.run_len[q]);
                at += r.run_len[q];
              }
              const bool is_ref = chain && (int64_t)total == t.n_ref;                  // the path 0, 1, .. n_ref-1
              w.is_ref[p - p0] = is_ref;
              if (is_ref) { path.clear(); continue; }      // never read: target_rows has 0 .. n_ref-1 of its own
              path.resize(total);                          // (what the last target left in it is overwritten, not cleared)
              at = 0;
              for (uint64_t q = r.run_off[p]; q < r.run_off[p + 1]; ++q) {
                Node* dst = path.data() + at;
                const int64_t first = (int64_t)r.run_start[q];
                const uint32_t len = r.run_len[q];
                for (uint32_t j = 0; j < len; ++j) dst[j] = (Node)(first + j);
                at += len;
              }
            }
            if (!consistent) { err[ti] = 5; continue; }
            if (lean && !(n_paths == 1 && w.is_ref[0])) { err[ti] = 5; continue; }
            {
              // (the scan the per-base spelling would otherwise repeat for every row of the target)
              const size_t len = (size_t)(t.n_ref + t.k - 1);
              const uint8_t* sq = (const uint8_t*)t.seq;
              uint8_t other = 0;
              for (size_t i = 0; i < len; ++i) {
                const uint8_t c = sq[i];
                other |= (uint8_t)((c != 'A') & (c != 'C') & (c != 'G') & (c != 'T'));
              }
              t.plain = other == 0;
            }
            g_tie = false;
            g_fit_iters = 0;
            const auto tt0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            err[ti] = target_rows(w, t, db, out);
            if (trace) {
              const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count();
              if (ms > 2.0) fprintf(stderr, "[km host] report: target %u took %.1f ms (%zu paths, %lld nodes, err %d; %ld gradient iterations)\n", ti, ms, n_paths, (long long)t.n_nodes, err[ti], g_fit_iters);
            }
            if (err[ti]) continue;
            if (g_tie) err[ti] = 100;                    // rows are still delivered
            done();
          }
        }
      } catch (...) {
        failed = true;
      }
      if (trace) wtrace[me].last_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count();
      // ---- every chunk is written: the last worker to arrive lays out the text
      if (arrived.fetch_add(1) + 1 == n_thr) {
        tr1 = std::chrono::steady_clock::now();
        size_t total = 0;
        row_off[0] = 0;
        for (uint32_t ti = 0; ti < n; ++ti) { const size_t len = row_off[ti + 1]; row_off[ti + 1] = (total += len); }
        text = failed ? nullptr : text_alloc(total + 1);
        if (text) text[total] = 0;
        phase.store(text ? 1 : -1, std::memory_order_release);
      } else {
        int spins = 0;
        while (phase.load(std::memory_order_acquire) == 0)
          if (++spins > 64) std::this_thread::yield();
      }
      if (phase.load(std::memory_order_acquire) != 1) return;
      for (uint32_t c = next_copy.fetch_add(1); c < n_chunks; c = next_copy.fetch_add(1)) {
        const uint32_t t0 = c * CHUNK, t1 = std::min<uint32_t>(n, (c + 1) * CHUNK);
        const size_t bytes = (size_t)(row_off[t1] - row_off[t0]);
        if (bytes) memcpy(text + row_off[t0], wbuf[chunk[c].worker].data() + chunk[c].at, bytes);
      }
    };
    if (n_thr <= 1) {
      work(0);
    } else if (on_team) {
      team().run(n_thr, work);
    } else {
      std::vector<std::thread> pool;
      for (unsigned q = 1; q < n_thr; ++q) pool.emplace_back(work, q);
      work(0);
      for (std::thread& th : pool) th.join();
    }
    if (failed || !text) { free(row_off); free(err); text_free(text); return KM_E_NOMEM; }
    if (trace)
      fprintf(stderr, "[km host] report: %u targets, %u threads, rows %.1f ms, text assembly %.1f ms, %llu bytes\n", n, n_thr,
              std::chrono::duration<double, std::milli>(tr1 - tr0).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr1).count(),
              (unsigned long long)row_off[n]);
    if (trace) {
      fprintf(stderr, "[km host] report workers (cpu: first chunk at us .. last done at us, chunks):");
      for (unsigned q = 0; q < n_thr; ++q) fprintf(stderr, " %d:%.0f..%.0f,%u", wtrace[q].cpu, wtrace[q].first_us, wtrace[q].last_us, wtrace[q].chunks);
      fprintf(stderr, "\n");
    }
    *text_out = text;
    *row_off_out = row_off;
    *err_out = err;
    return KM_OK;
  } catch (...) {
    free(row_off);
    free(err);
    return KM_E_NOMEM;
  }
}

extern "C" void km_report_free(char* text, uint64_t* row_off, int32_t* err) {
  text_free(text);
  free(row_off);
  free(err);
}
