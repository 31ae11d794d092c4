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
// The least-squares start is the minimum-norm solution through a one-sided Jacobi SVD with
// numpy's rcond=None cut-off; it agrees with LAPACK's gelsd to ~1e-13 relative, far inside the
// printed %.3f / %.1f — except on rounding ties and near-singular fits, which are flagged.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kmgpu.h"

namespace {

// set while formatting a target whose printed values depend on the last bits of the solver
thread_local bool g_tie = false;
thread_local double g_fit_ms = 0.0;
thread_local long g_fit_iters = 0;

typedef std::vector<int64_t> Path;

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
  std::vector<Path> paths;
  std::vector<uint32_t> min_cov;
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
bool split_paths(const Path& ref, const Path& alt, int k, Split* s) {
  const int64_t nr = (int64_t)ref.size(), na = (int64_t)alt.size();
  const int64_t m = std::min(nr, na);
  int64_t start = m;
  for (int64_t i = 0; i < m; ++i)
    if (ref[i] != alt[i]) { start = i; break; }
  const int64_t room = m - (start + k) + 1;
  int64_t same = 0;
  if (room > 0) {
    same = room;
    for (int64_t i = 0; i < room; ++i)
      if (ref[nr - 1 - i] != alt[na - 1 - i]) { same = i; break; }
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

std::string unpack(uint64_t kmer, int k) {
  std::string s((size_t)k, 'A');
  for (int i = k - 1; i >= 0; --i) { s[(size_t)i] = LAST[kmer & 3]; kmer >>= 2; }
  return s;
}

std::string spell(const Target& t, const Path& p, bool whole_first) {
  if (p.empty()) return std::string();
  std::string s = whole_first ? unpack(kmer_of(t, p[0]), t.k) : std::string(1, tail_of(t, p[0]));
  s.reserve(s.size() + p.size());
  for (size_t i = 1; i < p.size(); ++i) s.push_back(tail_of(t, p[i]));
  return s;
}

inline std::string suffix(const std::string& s, size_t n) { return n >= s.size() ? s : s.substr(s.size() - n); }

// 0 ok, 1 IndexError, 2 "mutation identification could be incorrect", 3 assertion
int name_variant(const Target& t, const Path& ref, const Path& alt, int64_t offset, std::string* out) {
  Split sp;
  if (!split_paths(ref, alt, t.k, &sp)) return 1;
  Path only_ref, only_var;
  slice(ref, sp.start, sp.end_ref, &only_ref);
  slice(alt, sp.start, sp.end_var, &only_var);
  if ((int64_t)ref.size() - (int64_t)only_ref.size() + (int64_t)only_var.size() != (int64_t)alt.size()) return 2;
  std::string gone = spell(t, only_ref, false), neu = spell(t, only_var, false);
  if (!gone.empty()) {
    if (gone == neu) return 3;
    size_t cut = 0;
    while (suffix(gone, cut + 1) == suffix(neu, cut + 1)) ++cut;
    if (cut) {
      gone = cut >= gone.size() ? std::string() : gone.substr(0, gone.size() - cut);
      neu = cut >= neu.size() ? std::string() : neu.substr(0, neu.size() - cut);
    }
  }
  const char* kind;
  if (sp.end_ref == sp.end_var) kind = sp.start == sp.end_ref ? "Reference" : "Substitution";
  else if (sp.start == sp.end_ovl) kind = "ITD";
  else if (sp.end_ref < sp.end_var) kind = gone.empty() ? "Insertion" : "Indel";
  else kind = neu.empty() ? "Deletion" : "Indel";
  if (!strcmp(kind, "Reference")) { *out = "Reference\t"; return 0; }
  for (char& c : gone) c = (char)tolower((unsigned char)c);
  char buf[64];
  snprintf(buf, sizeof buf, "%lld:", (long long)(sp.start + t.k + offset));
  *out = std::string(kind) + "\t" + buf + gone + "/" + neu;
  snprintf(buf, sizeof buf, ":%lld", (long long)(sp.end_ref + 1 + offset));
  *out += buf;
  return 0;
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
        for (size_t i = 0; i < n; ++i) {
          alpha += u[(size_t)p][i] * u[(size_t)p][i];
          beta += u[(size_t)q][i] * u[(size_t)q][i];
          gamma += u[(size_t)p][i] * u[(size_t)q][i];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (size_t i = 0; i < n; ++i) {
          const double up = u[(size_t)p][i], uq = u[(size_t)q][i];
          u[(size_t)p][i] = c * up - sn * uq;
          u[(size_t)q][i] = sn * up + c * uq;
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

// km_amd/report.py: fit_paths.  counts: float32 values of the node counts followed by -1, -1
// (the two capping nodes).  Returns coef and rvaf (rvaf == coef when every coefficient is 0).
void fit_paths(const std::vector<const Path*>& paths, const std::vector<float>& counts, int64_t n_total,
               std::vector<double>* coef_out, std::vector<double>* rvaf_out) {
  const int m = (int)paths.size();
  const auto fit_t0 = std::chrono::steady_clock::now();
  // contrib[i][c] = occurrences of node i on path c, one row per node
  std::vector<int32_t> contrib((size_t)n_total * (size_t)m, 0);
  for (int c = 0; c < m; ++c)
    for (int64_t node : *paths[(size_t)c]) contrib[(size_t)node * (size_t)m + (size_t)c] += 1;
  // The rows of contrib take few distinct values (a node lies on the reference only, on a variant path
  // only, on both, ...): everything below works on those PATTERNS — pattern q with its row, the number of
  // nodes N_q that have it and the sum S_q of their counts — instead of on the ~500 nodes.
  std::vector<std::vector<int32_t>> pat;
  std::vector<double> pat_sum, pat_n;
  {
    std::vector<int32_t> row((size_t)m);
    size_t last = 0;                                    // neighbours mostly share their pattern
    for (int64_t i = 0; i < n_total; ++i) {
      for (int c = 0; c < m; ++c) row[(size_t)c] = contrib[(size_t)i * (size_t)m + (size_t)c];
      size_t q = last < pat.size() && pat[last] == row ? last : 0;
      if (!(q < pat.size() && pat[q] == row))
        for (q = 0; q < pat.size(); ++q) if (pat[q] == row) break;
      if (q == pat.size()) { pat.push_back(row); pat_sum.push_back(0.0); pat_n.push_back(0.0); }
      pat_sum[q] += (double)counts[(size_t)i];
      pat_n[q] += 1.0;
      last = q;
    }
  }
  // Minimum-norm least squares with numpy's rcond=None cut-off (eps * max(n, m) * sigma_max on the singular
  // values).  A^T A = sum_q N_q pat_q pat_q^T and A^T b = sum_q S_q pat_q are sums of small integers times
  // counts, the singular values of A the square roots of the eigenvalues of the m x m matrix A^T A, its
  // eigenvectors the right singular vectors: coef = sum_j V_j (V_j . A^T b) / lambda_j over the kept j.
  // Squaring the condition number is harmless while the paths are well separated; a fit with a small or
  // vanishing singular value (near-identical or identical paths: a rank-deficient cluster) goes through
  // the one-sided Jacobi SVD of A itself, as every fit did before.
  std::vector<double> coef((size_t)m, 0.0);
  bool solved = false;
  {
    std::vector<std::vector<double>> g((size_t)m, std::vector<double>((size_t)m, 0.0));
    std::vector<double> atb((size_t)m, 0.0);
    for (size_t q = 0; q < pat.size(); ++q)
      for (int r = 0; r < m; ++r) {
        atb[(size_t)r] += (double)pat[q][(size_t)r] * pat_sum[q];
        for (int c = 0; c < m; ++c) g[(size_t)r][(size_t)c] += pat_n[q] * (double)pat[q][(size_t)r] * (double)pat[q][(size_t)c];
      }
    // cyclic Jacobi on the symmetric g: g -> diag(lambda), ev columns = eigenvectors
    std::vector<double> ev((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) ev[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
      double off = 0.0, diag = 0.0;
      for (int p2 = 0; p2 < m; ++p2) { diag += g[(size_t)p2][(size_t)p2] * g[(size_t)p2][(size_t)p2]; for (int q2 = p2 + 1; q2 < m; ++q2) off += g[(size_t)p2][(size_t)q2] * g[(size_t)p2][(size_t)q2]; }
      if (off <= 1e-30 * diag) break;
      for (int p2 = 0; p2 < m; ++p2)
        for (int q2 = p2 + 1; q2 < m; ++q2) {
          const double apq = g[(size_t)p2][(size_t)q2];
          if (apq == 0.0) continue;
          const double theta = (g[(size_t)q2][(size_t)q2] - g[(size_t)p2][(size_t)p2]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
          const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
          for (int r = 0; r < m; ++r) {                  // columns p2, q2
            const double gp = g[(size_t)r][(size_t)p2], gq = g[(size_t)r][(size_t)q2];
            g[(size_t)r][(size_t)p2] = c * gp - sn * gq;
            g[(size_t)r][(size_t)q2] = sn * gp + c * gq;
          }
          for (int r = 0; r < m; ++r) {                  // rows p2, q2
            const double gp = g[(size_t)p2][(size_t)r], gq = g[(size_t)q2][(size_t)r];
            g[(size_t)p2][(size_t)r] = c * gp - sn * gq;
            g[(size_t)q2][(size_t)r] = sn * gp + c * gq;
          }
          for (int r = 0; r < m; ++r) {
            const double vp = ev[(size_t)r * m + p2], vq = ev[(size_t)r * m + q2];
            ev[(size_t)r * m + p2] = c * vp - sn * vq;
            ev[(size_t)r * m + q2] = sn * vp + c * vq;
          }
        }
    }
    double lmax = 0.0, lmin = std::numeric_limits<double>::infinity();
    for (int j = 0; j < m; ++j) { lmax = std::max(lmax, g[(size_t)j][(size_t)j]); lmin = std::min(lmin, g[(size_t)j][(size_t)j]); }
    if (lmax > 0 && lmin > 1e-9 * lmax) {                 // sigma_min > 3e-5 sigma_max: every singular value is kept
      for (int j = 0; j < m; ++j) {
        double proj = 0.0;
        for (int r = 0; r < m; ++r) proj += ev[(size_t)r * m + j] * atb[(size_t)r];
        proj /= g[(size_t)j][(size_t)j];
        for (int r = 0; r < m; ++r) coef[(size_t)r] += ev[(size_t)r * m + j] * proj;
      }
      solved = true;
    }
  }
  if (!solved) {
    std::vector<std::vector<double>> u((size_t)m, std::vector<double>((size_t)n_total));
    for (int c = 0; c < m; ++c)
      for (int64_t i = 0; i < n_total; ++i) u[(size_t)c][(size_t)i] = (double)contrib[(size_t)i * (size_t)m + (size_t)c];
    std::vector<double> v, sig;
    jacobi_svd(u, m, v, sig);
    double smax = 0.0;
    for (double sj : sig) smax = std::max(smax, sj);
    const double cutoff = std::numeric_limits<double>::epsilon() * (double)std::max<int64_t>(n_total, m) * smax;
    for (int j = 0; j < m; ++j) {
      const double sj = sig[(size_t)j];
      // a singular value near the cut-off, or a kept one that small, leaves the answer to the
      // last bits of the solver: let the caller recompute this target with numpy
      if (smax > 0 && sj > 1e-14 * smax && sj < 1e-6 * smax) g_tie = true;
      if (!(sj > cutoff)) continue;
      double proj = 0.0;                              // (sigma_j u_j) . b / sigma_j^2
      for (int64_t i = 0; i < n_total; ++i) proj += u[(size_t)j][(size_t)i] * (double)counts[(size_t)i];
      proj /= sj * sj;
      for (int r = 0; r < m; ++r) coef[(size_t)r] += v[(size_t)r * m + j] * proj;
    }
  }
  for (double& c : coef) if (c < 0) c = 0;
  // Projected gradient refinement (km/utils/PathQuant.py:111-142): grad_c = 2/n * sum_i (count_i - est_i) * contrib_ic
  // with est_i = sum_c contrib_ic * coef_c, step 0.1, until max |grad| <= 0.01, evaluated per pattern:
  // grad_c = 2/n * sum_q pat_qc * (S_q - N_q * est_q).  The same numbers in exact arithmetic; in floating point
  // the sums are grouped differently, which matters as little as the difference between two BLAS builds does
  // to the reference itself (printed values near a rounding tie are flagged and recomputed with numpy: err 100).
  const size_t n_pat = pat.size();
  std::vector<double> grad((size_t)m), resid(n_pat);
  double step = std::numeric_limits<double>::infinity();
  while (step > 0.01) {
    for (size_t q = 0; q < n_pat; ++q) {
      double e = 0.0;
      for (int c = 0; c < m; ++c) e += (double)pat[q][(size_t)c] * coef[(size_t)c];
      resid[q] = pat_sum[q] - pat_n[q] * e;          // sum over the pattern's rows of (count - est)
    }
    for (int c = 0; c < m; ++c) {
      double sres = 0.0;
      for (size_t q = 0; q < n_pat; ++q) sres += 2.0 * resid[q] * (double)pat[q][(size_t)c];
      grad[(size_t)c] = sres / (double)n_total;
    }
    for (int c = 0; c < m; ++c) coef[(size_t)c] += 0.1 * grad[(size_t)c];
    for (int c = 0; c < m; ++c) if (coef[(size_t)c] < 0) { grad[(size_t)c] = 0; coef[(size_t)c] = 0; }
    step = 0.0;
    bool nan = false;
    for (int c = 0; c < m; ++c) {
      if (grad[(size_t)c] != grad[(size_t)c]) nan = true;
      step = std::max(step, std::fabs(grad[(size_t)c]));
    }
    ++g_fit_iters;
    if (nan) break;                                 // np.max of a NaN is NaN; NaN > 0.01 is False
  }
  g_fit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - fit_t0).count();
  *coef_out = coef;
  double mx = -std::numeric_limits<double>::infinity(), sum = 0.0;
  for (double c : coef) { mx = std::max(mx, c); sum += c; }
  if (mx == 0) *rvaf_out = coef;
  else {
    rvaf_out->resize(coef.size());
    for (size_t i = 0; i < coef.size(); ++i) (*rvaf_out)[i] = coef[i] / sum;
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

std::string fmt_float(const char* spec, double v) {
  if (v != v) return "nan";
  note_tie(v, spec[2] == '3' ? 1000.0 : 10.0);
  char buf[64];
  snprintf(buf, sizeof buf, spec, v);
  return buf;
}

std::string format_row(const char* db, const char* query, const std::string& name, double rvaf, double expr,
                       long long min_cov, long long off, const std::string& seq, double ref_expr,
                       const std::string& ref_seq, const std::string& note) {
  std::string r;
  r.reserve(seq.size() + ref_seq.size() + 128);
  r += db; r += '\t'; r += query; r += '\t'; r += name; r += '\t';
  r += fmt_float("%.3f", rvaf); r += '\t'; r += fmt_float("%.1f", expr); r += '\t';
  r += std::to_string(min_cov); r += '\t'; r += std::to_string(off); r += '\t';
  r += seq; r += '\t'; r += fmt_float("%.1f", ref_expr); r += '\t'; r += ref_seq; r += '\t'; r += note;
  return r;
}

// ---- the sort key of km_amd/report.py: row_key ------------------------------------------
struct NatTok { bool is_int; std::string s; unsigned long long v; };
typedef std::vector<NatTok> Nat;

Nat natural(const std::string& text) {           // re.split("([0-9]+)", text) with ints converted
  Nat out;
  size_t i = 0;
  while (true) {
    size_t j = i;
    while (j < text.size() && !(text[j] >= '0' && text[j] <= '9')) ++j;
    NatTok t{false, text.substr(i, j - i), 0};
    for (char& c : t.s) c = (char)tolower((unsigned char)c);
    out.push_back(t);
    if (j >= text.size()) break;
    size_t e = j;
    while (e < text.size() && text[e] >= '0' && text[e] <= '9') ++e;
    NatTok d{true, std::string(), strtoull(text.substr(j, e - j).c_str(), nullptr, 10)};
    out.push_back(d);
    i = e;
    if (i >= text.size()) { out.push_back(NatTok{false, std::string(), 0}); break; }
  }
  return out;
}

int cmp_nat(const Nat& a, const Nat& b) {
  const size_t n = std::min(a.size(), b.size());
  for (size_t i = 0; i < n; ++i) {
    if (a[i].is_int && b[i].is_int) {
      if (a[i].v != b[i].v) return a[i].v < b[i].v ? -1 : 1;
    } else {
      const int c = a[i].s.compare(b[i].s);
      if (c) return c < 0 ? -1 : 1;
    }
  }
  return a.size() == b.size() ? 0 : (a.size() < b.size() ? -1 : 1);
}

struct RowKey { std::vector<Nat> comps; };

RowKey row_key(const std::string& row) {
  std::vector<std::string> f;
  size_t i = 0;
  while (true) {
    size_t j = row.find('\t', i);
    if (j == std::string::npos) { f.push_back(row.substr(i)); break; }
    f.push_back(row.substr(i, j - i));
    i = j + 1;
  }
  RowKey k;
  const std::string& info = f[11];
  size_t p = 0;
  while (true) {
    size_t q = info.find(' ', p);
    if (q == std::string::npos) { k.comps.push_back(natural(info.substr(p))); break; }
    k.comps.push_back(natural(info.substr(p, q - p)));
    p = q + 1;
  }
  k.comps.push_back(natural(f[1]));
  k.comps.push_back(natural(f[3]));
  k.comps.push_back(natural(f[2]));
  k.comps.push_back(natural(f[6]));
  return k;
}

bool key_less(const RowKey& a, const RowKey& b) {
  const size_t n = std::min(a.comps.size(), b.comps.size());
  for (size_t i = 0; i < n; ++i) {
    int c = cmp_nat(a.comps[i], b.comps[i]);
    if (i == 0) c = -c;                           // the first component sorts descending
    if (c) return c < 0;
  }
  return a.comps.size() < b.comps.size();
}

bool is_reference(const Path& p, int64_t n_ref) {
  if ((int64_t)p.size() != n_ref) return false;
  for (int64_t i = 0; i < n_ref; ++i) if (p[(size_t)i] != i) return false;
  return true;
}

// km_amd/report.py: cluster_groups
struct Group { int64_t lo, hi; std::vector<int> members; };
void cluster_groups(const std::vector<Split>& diffs, std::vector<Group>* out) {
  std::vector<int> todo(diffs.size());
  for (size_t i = 0; i < diffs.size(); ++i) todo[i] = (int)i;
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
    Group g{diffs[(size_t)seed].start, diffs[(size_t)seed].end_ref, {seed}};
    int v = first_overlap(g.lo, g.hi);
    while (v != -1) {
      todo.erase(std::find(todo.begin(), todo.end(), v));
      g.members.push_back(v);
      g.lo = std::min(g.lo, diffs[(size_t)v].start);
      g.hi = std::max(g.hi, diffs[(size_t)v].end_ref);
      v = first_overlap(g.lo, g.hi);
    }
    out->push_back(g);
  }
}

// km_amd/report.py: target_rows.  0 ok, else the error code of name_variant / split_paths.
int target_rows(const Target& t, const char* db, std::vector<std::string>* rows_out) {
  const int k = t.k;
  const int64_t n_ref = t.n_ref, n_total = t.n_nodes + 2;
  if (!t.counts) {
    // lean delivery of a bare-reference target: its single row needs the path's min coverage and
    // whether every count is 0 (then PathQuant's rVAF aliases coef and both print nan)
    const double nan0 = std::numeric_limits<double>::quiet_NaN();
    const double expr0 = t.ref_max == 0 ? nan0 : -1.0;
    const std::string ref_seq0(t.seq, std::min<size_t>(t.seq_len, (size_t)(n_ref + k - 1)));
    rows_out->clear();
    if (t.paths.size() != 1 || !is_reference(t.paths[0], n_ref)) return 3;
    rows_out->push_back(format_row(db, t.name, "Reference\t", nan0, expr0, t.min_cov[0], 0, ref_seq0, expr0,
                                   ref_seq0, "vs_ref"));
    return 0;
  }
  std::vector<float> counts((size_t)n_total);
  for (int64_t i = 0; i < t.n_nodes; ++i) counts[(size_t)i] = (float)t.counts[i];
  counts[(size_t)n_total - 2] = counts[(size_t)n_total - 1] = -1.0f;
  Path ref((size_t)n_ref);
  for (int64_t i = 0; i < n_ref; ++i) ref[(size_t)i] = i;
  const std::string ref_seq(t.seq, std::min<size_t>(t.seq_len, (size_t)(n_ref + k - 1)));
  uint32_t ref_max = 0;
  for (int64_t i = 0; i < n_ref; ++i) ref_max = std::max(ref_max, t.counts[i]);
  const double nan = std::numeric_limits<double>::quiet_NaN();
  const double ref_expr = ref_max == 0 ? nan : -1.0;
  std::vector<std::string>& rows = *rows_out;
  rows.clear();
  if (t.paths.size() == 1 && is_reference(t.paths[0], n_ref)) {
    rows.push_back(format_row(db, t.name, "Reference\t", nan, ref_expr, t.min_cov[0], 0, ref_seq, ref_expr,
                              ref_seq, "vs_ref"));
    return 0;
  }
  std::vector<double> coef, rvaf;
  for (size_t pi = 0; pi < t.paths.size(); ++pi) {
    const Path& p = t.paths[pi];
    if (is_reference(p, n_ref)) {
      rows.push_back(format_row(db, t.name, "Reference\t", nan, ref_expr, t.min_cov[pi], 0, ref_seq, ref_expr,
                                ref_seq, "vs_ref"));
      continue;
    }
    fit_paths({&p, &ref}, counts, n_total, &coef, &rvaf);
    std::string name;
    int rc = name_variant(t, ref, p, 0, &name);
    if (rc) return rc;
    rows.push_back(format_row(db, t.name, name, rvaf[0], coef[0], t.min_cov[pi], 0, spell(t, p, true), coef[1],
                              ref_seq, "vs_ref"));
  }
  if (!t.paths.empty()) {
    std::vector<Split> diffs(t.paths.size());
    for (size_t pi = 0; pi < t.paths.size(); ++pi)
      if (!split_paths(ref, t.paths[pi], k, &diffs[pi])) return 1;
    std::vector<Group> groups;
    cluster_groups(diffs, &groups);
    int num = 0;
    for (const Group& g : groups) {
      if (g.members.size() == 1 && is_reference(t.paths[(size_t)g.members[0]], n_ref)) continue;
      ++num;
      int64_t size = 0;
      for (int v : g.members)
        size = std::max<int64_t>(size, std::llabs(diffs[(size_t)v].end_var - diffs[(size_t)v].end_ref + 1));
      const int64_t off = std::max<int64_t>(0, g.lo - size);
      Path cref;
      slice(ref, off, g.hi, &cref);
      std::vector<Path> clipped(g.members.size());
      for (size_t q = 0; q < g.members.size(); ++q) {
        const int v = g.members[q];
        slice(t.paths[(size_t)v], off, diffs[(size_t)v].end_var + g.hi - diffs[(size_t)v].end_ref, &clipped[q]);
      }
      std::vector<const Path*> set;
      set.push_back(&cref);
      for (const Path& c : clipped) set.push_back(&c);
      fit_paths(set, counts, n_total, &coef, &rvaf);
      const std::string cref_seq = spell(t, cref, true);
      char note[64];
      snprintf(note, sizeof note, "cluster %d n=%d", num, (int)clipped.size());
      for (size_t q = 0; q < clipped.size(); ++q) {
        const Path& p = clipped[q];
        if (p.empty()) return 4;                         // min() of an empty sequence
        uint32_t mc = 0xFFFFFFFFu;
        for (int64_t node : p) mc = std::min(mc, t.counts[node]);
        std::string name;
        int rc = name_variant(t, cref, p, off, &name);
        if (rc) return rc;
        rows.push_back(format_row(db, t.name, name, rvaf[q + 1], coef[q + 1], mc, off, spell(t, p, true),
                                  coef[0], cref_seq, note));
      }
    }
  }
  std::vector<RowKey> keys(rows.size());
  std::vector<size_t> order(rows.size());
  for (size_t i = 0; i < rows.size(); ++i) { keys[i] = row_key(rows[i]); order[i] = i; }
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key_less(keys[a], keys[b]); });
  std::vector<std::string> sorted(rows.size());
  for (size_t i = 0; i < rows.size(); ++i) sorted[i].swap(rows[order[i]]);
  rows.swap(sorted);
  return 0;
}

}  // namespace

extern "C" int km_report_rows(const km_report_in_t* in, char** text_out, uint64_t** row_off_out,
                              int32_t** err_out) {
  if (!in || !text_out || !row_off_out || !err_out || !in->res) return KM_E_ARG;
  const km_batch_out_t& r = *in->res;
  if (!r.status || !r.n_ref || !r.node_off || (!r.node_kmer && (!r.extra_off || !r.extra_kmer)) ||
      !r.node_count || !r.path_off || !r.run_off ||
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
  }
  uint64_t* row_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
  int32_t* err = (int32_t*)malloc(sizeof(int32_t) * std::max<size_t>(1, n));
  if (!row_off || !err) { free(row_off); free(err); return KM_E_NOMEM; }
  try {
    // targets are independent: a few host threads (KM_REPORT_THREADS, default min(cores, 16))
    // pull them off a shared counter, each block of rows is kept per target and concatenated
    const bool trace = getenv("KM_TRACE_HOST") != nullptr;
    std::vector<std::string> block(n);
    std::atomic<uint32_t> next(0);
    std::atomic<bool> failed(false);
    auto work = [&]() {
      std::vector<std::string> rows;
      try {
        for (uint32_t ti = next.fetch_add(1); ti < n; ti = next.fetch_add(1)) {
          err[ti] = 0;
          if (r.status[ti] != KM_T_OK) continue;
          Target t;
          t.name = in->names[ti] ? in->names[ti] : "";
          t.seq = (const char*)in->bases + in->base_off[ti];
          t.seq_len = (size_t)(in->base_off[ti + 1] - in->base_off[ti]);
          t.k = in->k;
          t.n_ref = r.n_ref[ti];
          t.kmers = r.node_kmer ? r.node_kmer + r.node_off[ti] : nullptr;
          t.extra = r.node_kmer ? nullptr : r.extra_kmer + r.extra_off[ti];
          t.counts = r.node_count + r.node_off[ti];
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
          if (!r.node_kmer) {
            if (r.extra_off[ti + 1] < r.extra_off[ti] ||
                (int64_t)(r.extra_off[ti + 1] - r.extra_off[ti]) != t.n_nodes - t.n_ref) { err[ti] = 5; continue; }
          }
          const uint32_t p0 = r.path_off[ti], p1 = r.path_off[ti + 1];
          if (lean && p1 == p0 + 1 && r.run_off[p0 + 1] == r.run_off[p0] + 1 && r.run_start[r.run_off[p0]] == 0 &&
              (int64_t)r.run_len[r.run_off[p0]] == t.n_ref) {
            // the common case by far — a bare-reference target delivered lean, its one path the single run
            // 0 .. n_ref-1: its one row needs neither the path's nodes nor any count
            const double nan0 = std::numeric_limits<double>::quiet_NaN();
            const double expr0 = t.ref_max == 0 ? nan0 : -1.0;
            const std::string ref_seq0(t.seq, (size_t)(t.n_ref + t.k - 1));
            block[ti] = format_row(in->db_name ? in->db_name : "", t.name, "Reference\t", nan0, expr0, r.path_min_cov[p0], 0,
                                   ref_seq0, expr0, ref_seq0, "vs_ref");
            block[ti].push_back('\n');
            continue;
          }
          t.paths.resize(p1 - p0);
          t.min_cov.assign(r.path_min_cov + p0, r.path_min_cov + p1);
          bool consistent = true;
          for (uint32_t p = p0; p < p1 && consistent; ++p) {
            Path& path = t.paths[p - p0];
            path.clear();
            if (r.run_off[p + 1] < r.run_off[p]) { consistent = false; break; }
            for (uint64_t q = r.run_off[p]; q < r.run_off[p + 1]; ++q) {
              // every node of a path is one of this target's nodes (fit_paths indexes by it)
              if ((int64_t)r.run_start[q] + (int64_t)r.run_len[q] > t.n_nodes) { consistent = false; break; }
              for (uint32_t j = 0; j < r.run_len[q]; ++j) path.push_back((int64_t)r.run_start[q] + j);
            }
          }
          if (!consistent) { err[ti] = 5; continue; }
          if (lean && !(t.paths.size() == 1 && is_reference(t.paths[0], t.n_ref))) { err[ti] = 5; continue; }
          g_tie = false;
          g_fit_ms = 0.0; g_fit_iters = 0;
          const auto tt0 = std::chrono::steady_clock::now();
          err[ti] = target_rows(t, in->db_name ? in->db_name : "", &rows);
          if (trace) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count();
            if (ms > 2.0) fprintf(stderr, "[km host] report: target %u took %.1f ms (%zu paths, %lld nodes, err %d; gradient loops %.1f ms, %ld iterations)\n", ti, ms, t.paths.size(), (long long)t.n_nodes, err[ti], g_fit_ms, g_fit_iters);
          }
          if (err[ti]) continue;
          if (g_tie) err[ti] = 100;                    // rows are still delivered
          std::string& out = block[ti];
          for (size_t i = 0; i < rows.size(); ++i) {
            out += rows[i];
            out.push_back('\n');                     // every row is terminated: blocks concatenate into the TSV
          }
        }
      } catch (...) {
        failed = true;
      }
    };
    const auto tr0 = std::chrono::steady_clock::now();
    unsigned n_thr = std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    if (const char* e = getenv("KM_REPORT_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) n_thr = (unsigned)v; }
    n_thr = std::min<unsigned>(n_thr, std::max<uint32_t>(1, n / 64));     // small batches: no threads
    if (n_thr <= 1) {
      work();
    } else {
      std::vector<std::thread> pool;
      for (unsigned q = 0; q < n_thr; ++q) pool.emplace_back(work);
      for (std::thread& th : pool) th.join();
    }
    const auto tr1 = std::chrono::steady_clock::now();
    if (failed) { free(row_off); free(err); return KM_E_NOMEM; }
    // one buffer: offsets first, then the blocks are copied into place by the same number of threads
    // (a 10 000-target batch prints ~15 MB; concatenating through a std::string and copying that again
    // cost more than producing the rows)
    size_t total = 0;
    for (uint32_t ti = 0; ti < n; ++ti) { row_off[ti] = total; total += block[ti].size(); }
    row_off[n] = total;
    char* buf = (char*)malloc(total + 1);
    if (!buf) { free(row_off); free(err); return KM_E_NOMEM; }
    std::atomic<uint32_t> next_copy(0);
    auto copy_work = [&]() {
      for (uint32_t lo = next_copy.fetch_add(256); lo < n; lo = next_copy.fetch_add(256))
        for (uint32_t ti = lo; ti < std::min<uint32_t>(n, lo + 256); ++ti)
          if (!block[ti].empty()) memcpy(buf + row_off[ti], block[ti].data(), block[ti].size());
    };
    if (n_thr <= 1) {
      copy_work();
    } else {
      std::vector<std::thread> pool;
      for (unsigned q = 0; q < n_thr; ++q) pool.emplace_back(copy_work);
      for (std::thread& th : pool) th.join();
    }
    buf[total] = 0;
    if (trace)
      fprintf(stderr, "[km host] report: %u targets, %u threads, rows %.1f ms, text assembly %.1f ms, %zu bytes\n", n, n_thr,
              std::chrono::duration<double, std::milli>(tr1 - tr0).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr1).count(), total);
    *text_out = buf;
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
  free(text);
  free(row_off);
  free(err);
}
