// CPU-only timing driver for km_report_rows: loads a view written by make_bench_view.py, runs it R times
// with T threads (KM_REPORT_THREADS) and prints ms per call and a hash of the text.
//   g++ -O2 -std=c++17 -pthread -o /tmp/report_time tests/host/report_time.cpp km_amd/csrc/report.cpp
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kmgpu.h"

template <typename T>
static bool rd(FILE* f, std::vector<T>& v) {
  uint64_t n;
  if (fread(&n, 8, 1, f) != 1) return false;
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const bool lean = argc > 3 && atoi(argv[3]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint32_t hdr[2];
  std::vector<uint8_t> bases;
  std::vector<uint64_t> base_off, node_off, extra_off, run_off, extra_kmer;
  std::vector<uint32_t> status, n_ref, node_count, path_off, run_start, run_len, path_min_cov, ref_max;
  bool ok = fread(hdr, 4, 2, f) == 2;
  ok = ok && rd(f, bases) && rd(f, base_off) && rd(f, status) && rd(f, n_ref) && rd(f, node_off) && rd(f, node_count) &&
       rd(f, extra_off) && rd(f, extra_kmer) && rd(f, path_off) && rd(f, run_off) && rd(f, run_start) && rd(f, run_len) &&
       rd(f, path_min_cov) && rd(f, ref_max);
  fclose(f);
  if (!ok) return 2;
  const uint32_t n = hdr[0];
  if (extra_kmer.empty()) extra_kmer.reserve(1);           // data() must not be NULL
  if (node_count.empty()) node_count.reserve(1);
  if (lean) {                      // bare-reference targets lose their counts, as the lean delivery does
    std::vector<uint64_t> noff(n + 1, 0);
    std::vector<uint32_t> cnt;
    for (uint32_t t = 0; t < n; ++t) {
      if (ref_max[t] == 0xFFFFFFFFu) cnt.insert(cnt.end(), node_count.begin() + node_off[t], node_count.begin() + node_off[t + 1]);
      noff[t + 1] = cnt.size();
    }
    node_off = noff; node_count = cnt;
  }
  std::vector<std::string> names_s;
  std::vector<const char*> names;
  for (uint32_t t = 0; t < n; ++t) names_s.push_back("target_" + std::to_string(t));
  for (auto& s : names_s) names.push_back(s.c_str());
  km_batch_out_t out;
  memset(&out, 0, sizeof out);
  out.status = status.data(); out.n_ref = n_ref.data(); out.node_off = node_off.data(); out.node_count = node_count.data();
  out.extra_off = extra_off.data(); out.extra_kmer = extra_kmer.data(); out.path_off = path_off.data();
  out.run_off = run_off.data(); out.run_start = run_start.data(); out.run_len = run_len.data();
  out.path_min_cov = path_min_cov.data(); out.ref_max_cov = ref_max.data();
  km_report_in_t in;
  memset(&in, 0, sizeof in);
  in.n_targets = n; in.bases = bases.data(); in.base_off = base_off.data(); in.names = names.data();
  in.db_name = "synthetic.jf"; in.k = (int32_t)hdr[1]; in.res = &out;
  double best = 1e30, sum = 0;
  uint64_t h = 0, bytes = 0;
  for (int r = 0; r < reps; ++r) {
    char* text = nullptr; uint64_t* row_off = nullptr; int32_t* err = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = km_report_rows(&in, &text, &row_off, &err);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != KM_OK) { fprintf(stderr, "rc %d\n", rc); return 1; }
    best = ms < best ? ms : best; sum += ms;
    if (r == 0) {
      bytes = row_off[n];
      h = 1469598103934665603ull;
      for (uint64_t i = 0; i < bytes; ++i) { h ^= (unsigned char)text[i]; h *= 1099511628211ull; }
      int nerr = 0;
      for (uint32_t t = 0; t < n; ++t) nerr += err[t] != 0;
      printf("targets %u, bytes %llu, flagged %d, fnv %016llx\n", n, (unsigned long long)bytes, nerr, (unsigned long long)h);
    }
    km_report_free(text, row_off, err);
  }
  printf("ms per call: best %.3f mean %.3f  (%.2f us per target, best)\n", best, sum / reps, best * 1000.0 / n);
  return 0;
}
