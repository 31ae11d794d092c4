// CPU-only driver for km_report_rows (csrc/report.cpp), built with -fsanitize=address,undefined by
// tests/test_report_native.py: reads a result view dumped by the test (plain binary, see load()),
// runs the full view, its lean form, and a series of deliberately inconsistent views.  Every
// inconsistent view must come back as an error code (KM_E_ARG for the call, or err 5 for the
// target), never as an out-of-bounds access — the sanitizers turn those into a non-zero exit.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kmgpu.h"

struct View {
  uint32_t n = 0, k = 0;
  std::vector<uint8_t> bases;
  std::vector<uint64_t> base_off, node_off, extra_off, run_off, extra_kmer;
  std::vector<uint32_t> status, n_ref, node_count, path_off, run_start, run_len, path_min_cov, ref_max;
  std::vector<std::string> names;
};

template <typename T>
static bool rd(FILE* f, std::vector<T>& v) {
  uint64_t n;
  if (fread(&n, 8, 1, f) != 1) return false;
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

static bool load(const char* path, View& v) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  uint32_t hdr[2];
  bool ok = fread(hdr, 4, 2, f) == 2;
  v.n = hdr[0]; v.k = hdr[1];
  ok = ok && rd(f, v.bases) && rd(f, v.base_off) && rd(f, v.status) && rd(f, v.n_ref) && rd(f, v.node_off) &&
       rd(f, v.node_count) && rd(f, v.extra_off) && rd(f, v.extra_kmer) && rd(f, v.path_off) && rd(f, v.run_off) &&
       rd(f, v.run_start) && rd(f, v.run_len) && rd(f, v.path_min_cov) && rd(f, v.ref_max);
  fclose(f);
  for (uint32_t t = 0; t < v.n; ++t) v.names.push_back("t" + std::to_string(t));
  return ok;
}

// exact-size heap copies: an access one element past any array is a sanitizer report
template <typename T>
static T* dup(const std::vector<T>& v) {
  T* p = (T*)malloc(sizeof(T) * (v.size() ? v.size() : 1));
  if (v.size()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

static uint64_t g_text_hash = 0;          // FNV-1a of the text of the last served call
// as16: 0 = 32-bit counts; 1 = 16-bit counts + escape list; 2 = ... with the list out of order; 3 = ... with an
// entry beyond the nodes
static int run(const View& v, bool with_sizes, int* n_err5, int* n_rows, const char* label, int as16 = 0) {
  km_batch_out_t out;
  memset(&out, 0, sizeof out);
  std::vector<uint16_t> c16;
  std::vector<uint64_t> esc_n;
  std::vector<uint32_t> esc_v;
  if (as16) {
    for (size_t i = 0; i < v.node_count.size(); ++i) {
      const uint32_t c = v.node_count[i];
      c16.push_back((uint16_t)(c >= 0xFFFFu ? 0xFFFFu : c));
      if (c >= 0xFFFFu) { esc_n.push_back(i); esc_v.push_back(c); }
    }
    if (as16 == 2 && esc_n.size() >= 2) std::swap(esc_n[0], esc_n[1]);
    if (as16 == 3 && !esc_n.empty()) esc_n.back() = v.node_count.size() + 7;
  }
  out.status = dup(v.status); out.n_ref = dup(v.n_ref); out.node_off = dup(v.node_off);
  out.node_count = dup(v.node_count); out.extra_off = dup(v.extra_off); out.extra_kmer = dup(v.extra_kmer);
  out.path_off = dup(v.path_off); out.run_off = dup(v.run_off); out.run_start = dup(v.run_start);
  out.run_len = dup(v.run_len); out.path_min_cov = dup(v.path_min_cov); out.ref_max_cov = dup(v.ref_max);
  std::vector<const char*> names;
  for (const std::string& s : v.names) names.push_back(s.c_str());
  uint8_t* bases = dup(v.bases);
  uint64_t* base_off = dup(v.base_off);
  km_batch_sizes_t sz;
  memset(&sz, 0, sizeof sz);
  sz.n_targets = v.n; sz.n_nodes = v.node_count.size(); sz.n_paths = (uint32_t)v.path_min_cov.size();
  sz.n_runs = v.run_start.size(); sz.n_extra = v.extra_kmer.size();
  uint16_t* d16 = nullptr; uint64_t* den = nullptr; uint32_t* dev = nullptr;
  if (as16) {
    free(out.node_count);
    out.node_count = nullptr;
    d16 = dup(c16); den = dup(esc_n); dev = dup(esc_v);
    out.node_count16 = d16; out.count_esc_node = den; out.count_esc_value = dev;
    sz.n_count_escapes = (uint32_t)esc_n.size();
  }
  km_report_in_t in;
  memset(&in, 0, sizeof in);
  in.n_targets = v.n; in.bases = bases; in.base_off = base_off; in.names = names.data(); in.db_name = "view.jf";
  in.k = (int32_t)v.k; in.res = &out; in.sizes = with_sizes ? &sz : nullptr;
  char* text = nullptr;
  uint64_t* row_off = nullptr;
  int32_t* err = nullptr;
  const int rc = km_report_rows(&in, &text, &row_off, &err);
  *n_err5 = 0; *n_rows = 0;
  if (rc == KM_OK) {
    for (uint32_t t = 0; t < v.n; ++t) *n_err5 += err[t] == 5;
    g_text_hash = 1469598103934665603ull;
    for (uint64_t i = 0; i < row_off[v.n]; ++i) { *n_rows += text[i] == '\n'; g_text_hash = (g_text_hash ^ (unsigned char)text[i]) * 1099511628211ull; }
    km_report_free(text, row_off, err);
  }
  printf("%-44s sizes=%d rc=%d err5=%d rows=%d\n", label, (int)with_sizes, rc, *n_err5, *n_rows);
  free(out.status); free(out.n_ref); free(out.node_off); free(out.node_count); free(out.extra_off); free(out.extra_kmer);
  free(out.path_off); free(out.run_off); free(out.run_start); free(out.run_len); free(out.path_min_cov); free(out.ref_max_cov);
  free(bases); free(base_off); free(d16); free(den); free(dev);
  return rc;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  View full;
  if (!load(argv[1], full)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  int e5, rows, fails = 0;
  auto expect = [&](bool cond, const char* what) { if (!cond) { printf("  UNEXPECTED: %s\n", what); ++fails; } };
  // 1. the view as delivered, with and without the lengths
  int rc = run(full, true, &e5, &rows, "full view");
  const int full_rows = rows;
  expect(rc == KM_OK && e5 == 0 && rows > 0, "full view must be served");
  rc = run(full, false, &e5, &rows, "full view");
  expect(rc == KM_OK && e5 == 0 && rows == full_rows, "the same without lengths");
  // 2. lean: bare-reference targets lose their counts (node_off collapses), as the delivery kernels do it
  View lean = full;
  {
    std::vector<uint32_t> cnt;
    std::vector<uint64_t> noff(1, 0);
    for (uint32_t t = 0; t < full.n; ++t) {
      const bool bare = full.ref_max[t] != 0xFFFFFFFFu;
      if (!bare) cnt.insert(cnt.end(), full.node_count.begin() + full.node_off[t], full.node_count.begin() + full.node_off[t + 1]);
      noff.push_back(cnt.size());
    }
    lean.node_count = cnt; lean.node_off = noff;
  }
  rc = run(lean, true, &e5, &rows, "lean view");
  expect(rc == KM_OK && e5 == 0 && rows == full_rows, "lean view must give the same rows");
  // 3. what round 2's work-in-progress delivery handed over: a target with variant paths, counts missing
  uint32_t tv = full.n;
  for (uint32_t t = 0; t < full.n; ++t) if (full.path_off[t + 1] - full.path_off[t] > 1) { tv = t; break; }
  if (tv < full.n) {
    View bad = full;
    const uint64_t a = bad.node_off[tv], b = bad.node_off[tv + 1];
    bad.node_count.erase(bad.node_count.begin() + a, bad.node_count.begin() + b);
    for (uint32_t t = tv + 1; t <= bad.n; ++t) bad.node_off[t] -= b - a;
    bad.ref_max[tv] = 7;                                   // (a stale value, as it was)
    rc = run(bad, true, &e5, &rows, "variant target without counts");
    expect(rc == KM_OK && e5 == 1, "must be err 5 for that target only");
    bad.ref_max[tv] = 0xFFFFFFFFu;
    rc = run(bad, false, &e5, &rows, "variant target without counts, NOT_BARE");
    expect(rc == KM_OK && e5 == 1, "must be err 5 for that target only");
    // 4. a path node beyond the target's nodes
    View bad2 = full;
    const uint64_t q = bad2.run_off[bad2.path_off[tv] + 1];
    bad2.run_start[q] += 100000;
    rc = run(bad2, true, &e5, &rows, "path node beyond the target's nodes");
    expect(rc == KM_OK && e5 == 1, "must be err 5 for that target only");
    bad2 = full;
    bad2.run_len[q] = 0x7FFFFFFFu;
    rc = run(bad2, false, &e5, &rows, "run length beyond the target's nodes");
    expect(rc == KM_OK && e5 == 1, "must be err 5 for that target only");
    // 5. extra_off disagreeing with node_off / n_ref
    View bad3 = full;
    bad3.n_ref[tv] += 3;
    rc = run(bad3, true, &e5, &rows, "n_ref vs extra_off mismatch");
    expect(rc == KM_OK && e5 >= 1, "must be err 5");
  }
  // 6. offsets against the lengths (caught for the whole call when the lengths are given)
  {
    View bad = full;
    bad.node_off[full.n] += 5;
    rc = run(bad, true, &e5, &rows, "node_off[n] beyond node_count");
    expect(rc == KM_E_ARG, "KM_E_ARG");
    bad = full;
    if (full.n >= 2) { bad.path_off[1] = bad.path_off[full.n] + 9; }
    rc = run(bad, true, &e5, &rows, "path_off not monotone");
    expect(rc == KM_E_ARG, "KM_E_ARG");
    bad = full;
    bad.run_off[bad.run_off.size() - 1] += 4;
    rc = run(bad, true, &e5, &rows, "run_off beyond run_start");
    expect(rc == KM_E_ARG, "KM_E_ARG");
    bad = full;
    bad.extra_off[full.n] += 2;
    rc = run(bad, true, &e5, &rows, "extra_off beyond extra_kmer");
    expect(rc == KM_E_ARG, "KM_E_ARG");
  }
  // 7. 16-bit counts + escape list (KM_DELIVER_COUNT16): the same text as the 32-bit form — as delivered (no count
  // reaches 65535) and with every count times 400 (most do); a list out of order or pointing beyond the nodes is
  // an argument error
  {
    rc = run(full, true, &e5, &rows, "full view");
    const uint64_t h32 = g_text_hash;
    rc = run(full, true, &e5, &rows, "full view, 16-bit counts", 1);
    expect(rc == KM_OK && e5 == 0 && rows == full_rows && g_text_hash == h32, "16-bit counts must give the same text");
    rc = run(lean, true, &e5, &rows, "lean view, 16-bit counts", 1);
    expect(rc == KM_OK && e5 == 0 && rows == full_rows && g_text_hash == h32, "lean 16-bit counts must give the same text");
    View big = lean;
    size_t over = 0;
    for (uint32_t& c : big.node_count) { c *= 400u; over += c >= 0xFFFFu; }
    rc = run(big, true, &e5, &rows, "counts x 400");
    const uint64_t hbig = g_text_hash;
    expect(rc == KM_OK && over > 2, "the scaled view must be served and hold escapes");
    rc = run(big, true, &e5, &rows, "counts x 400, 16-bit + escape list", 1);
    expect(rc == KM_OK && g_text_hash == hbig, "escaped counts must give the same text");
    rc = run(big, true, &e5, &rows, "escape list out of order", 2);
    expect(rc == KM_E_ARG, "KM_E_ARG");
    rc = run(big, true, &e5, &rows, "escape beyond the nodes", 3);
    expect(rc == KM_E_ARG, "KM_E_ARG");
    rc = run(full, false, &e5, &rows, "16-bit counts without the lengths", 1);
    expect(rc == KM_E_ARG, "KM_E_ARG (the list's length travels in the sizes)");
  }
  printf("%s\n", fails ? "VIEWS FAILED" : "VIEWS OK");
  return fails ? 1 : 0;
}
