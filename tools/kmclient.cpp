// kmclient — a consumer of libkmgpu.so's C-ABI with no interpreter in the process (include/kmgpu.h only).
//
//   kmclient pump <dir> <steps> <warmup> <inflight> [repeats]
//       the pipelined step bench.py times as `value`: <inflight> batches on their own streams, each with its
//       own 10 000-target set resident in HBM, km_batch_pump over them with lean delivery.  The program to put
//       behind `rocprofv3 --kernel-trace --memory-copy-trace --` (tools/collect_evidence.sh).
//   kmclient e2e <dir> <steps> <warmup> <inflight> [repeats]
//       end to end: every step hands FRESH host strings to km_batch_set_targets (H2D), runs walk + path search +
//       lean delivery, and km_report_rows turns the delivered view into TSV text on a second thread while the
//       next batches run (km/tools/find_mutation.py:47-58 over successive batches of a catalog).  Prints the
//       rate and an FNV-1a hash of all the text, so that a caller can compare it with another path's.
//
// <dir> holds keys.npy (uint64), counts.npy (uint32), targets.npy (uint8 codes 0..3, shape [n, L]) as
// bench.py --dump-case writes them.  One JSON line on stdout.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/kmgpu.h"

#define CHECK(call)                                                                     \
  do {                                                                                  \
    const int rc_ = (call);                                                             \
    if (rc_ != KM_OK) {                                                                 \
      fprintf(stderr, "%s failed: %s (%s)\n", #call, km_strerror(rc_), km_last_error()); \
      exit(3);                                                                          \
    }                                                                                   \
  } while (0)

struct Npy {
  std::vector<unsigned char> data;
  std::vector<uint64_t> shape;
  size_t elem = 0;
};

static Npy load_npy(const std::string& path) {
  Npy a;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  unsigned char hdr[10];
  if (fread(hdr, 1, 10, f) != 10 || memcmp(hdr, "\x93NUMPY", 6) != 0) { fprintf(stderr, "%s: not a .npy file\n", path.c_str()); exit(2); }
  size_t hlen = hdr[8] | (hdr[9] << 8);
  if (hdr[6] >= 2) {                                  // version 2: 4-byte header length
    unsigned char more[2];
    if (fread(more, 1, 2, f) != 2) exit(2);
    hlen |= ((size_t)more[0] << 16) | ((size_t)more[1] << 24);
  }
  std::string h(hlen, ' ');
  if (fread(&h[0], 1, hlen, f) != hlen) exit(2);
  const size_t d = h.find("'descr':");
  const size_t q0 = h.find('\'', d + 8), q1 = h.find('\'', q0 + 1);
  const std::string descr = h.substr(q0 + 1, q1 - q0 - 1);
  a.elem = (size_t)atoi(descr.c_str() + 2);
  if (descr[0] != '<' && descr[0] != '|') { fprintf(stderr, "%s: big-endian data\n", path.c_str()); exit(2); }
  if (h.find("'fortran_order': False") == std::string::npos) { fprintf(stderr, "%s: fortran order\n", path.c_str()); exit(2); }
  const size_t s0 = h.find('(', h.find("'shape':")), s1 = h.find(')', s0);
  uint64_t total = 1;
  for (size_t p = s0 + 1; p < s1;) {
    while (p < s1 && (h[p] == ' ' || h[p] == ',')) ++p;
    if (p >= s1) break;
    const uint64_t v = strtoull(h.c_str() + p, nullptr, 10);
    a.shape.push_back(v);
    total *= v;
    while (p < s1 && h[p] != ',') ++p;
  }
  a.data.resize(total * a.elem);
  if (total && fread(a.data.data(), a.elem, total, f) != total) { fprintf(stderr, "%s: short read\n", path.c_str()); exit(2); }
  fclose(f);
  return a;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: kmclient pump|e2e <dir> <steps> <warmup> <inflight> [repeats]\n");
    return 2;
  }
  const std::string mode = argv[1], dir = argv[2];
  const int steps = atoi(argv[3]), warmup = atoi(argv[4]), n_fl = atoi(argv[5]);
  const int repeats = argc > 6 ? atoi(argv[6]) : 3;
  if (steps < 1 || n_fl < 1 || n_fl > 16) return 2;
  const int K = 31;
  Npy keys = load_npy(dir + "/keys.npy"), counts = load_npy(dir + "/counts.npy"), tg = load_npy(dir + "/targets.npy");
  if (keys.elem != 8 || counts.elem != 4 || tg.elem != 1 || tg.shape.size() != 2) { fprintf(stderr, "unexpected array types\n"); return 2; }
  const uint64_t n_keys = keys.shape[0], n_all = tg.shape[0], L = tg.shape[1];
  const uint32_t T = (uint32_t)(n_all / (uint64_t)n_fl);
  if (T == 0) return 2;
  // ASCII targets, as a caller holds them
  std::vector<uint8_t> ascii(tg.data.size());
  for (size_t i = 0; i < ascii.size(); ++i) ascii[i] = (uint8_t)"ACGT"[tg.data[i] & 3];
  std::vector<uint64_t> off(T + 1);
  for (uint32_t t = 0; t <= T; ++t) off[t] = (uint64_t)t * L;
  std::vector<std::string> name_store(T);
  std::vector<const char*> names(T);
  for (uint32_t t = 0; t < T; ++t) { name_store[t] = "syn_t" + std::to_string(t); names[t] = name_store[t].c_str(); }

  const double t_up = now_s();
  kmjf_t* db = nullptr;
  CHECK(kmjf_from_records((const uint64_t*)keys.data.data(), (const uint32_t*)counts.data.data(), n_keys, K, 1, &db));
  CHECK(kmjf_upload(db, 0));
  const double upload_s = now_s() - t_up;
  keys.data.clear(); keys.data.shrink_to_fit();
  counts.data.clear(); counts.data.shrink_to_fit();
  km_params_t prm;
  memset(&prm, 0, sizeof prm);
  prm.ratio = 0.05; prm.count = 5; prm.max_stack = 500; prm.max_break = 10; prm.max_node = 10000;
  std::vector<km_batch_t*> bs((size_t)n_fl, nullptr);
  std::vector<void*> streams((size_t)n_fl, nullptr);
  for (int q = 0; q < n_fl; ++q) {
    CHECK(km_batch_create(db, &prm, T, (uint64_t)T * L, &bs[(size_t)q]));
    CHECK(km_stream_create(0, &streams[(size_t)q]));
    CHECK(km_batch_set_targets(bs[(size_t)q], ascii.data() + (uint64_t)q * T * L, off.data(), T));
  }
  const int flags = KM_STAGE_WALK | KM_STAGE_GRAPH | KM_RUN_DELIVER | KM_DELIVER_LEAN |
                    (getenv("KMCLIENT_COUNT32") ? 0 : KM_DELIVER_COUNT16);

  if (mode == "pump") {
    CHECK(km_batch_pump(bs.data(), streams.data(), n_fl, warmup > n_fl ? warmup : n_fl, flags));
    std::vector<double> ms;
    for (int r = 0; r < repeats; ++r) {
      const double t0 = now_s();
      CHECK(km_batch_pump(bs.data(), streams.data(), n_fl, steps, flags));
      ms.push_back((now_s() - t0) / steps * 1e3);
    }
    double best = ms[0], worst = ms[0];
    for (double v : ms) { best = v < best ? v : best; worst = v > worst ? v : worst; }
    km_batch_sizes_t sz;
    CHECK(km_batch_result(bs[0], nullptr, &sz));
    printf("{\"mode\": \"pump\", \"targets_per_step\": %u, \"batches_in_flight\": %d, \"steps\": %d, \"repeats\": %d, "
           "\"ms_per_step_min\": %.6f, \"ms_per_step_max\": %.6f, \"targets_per_s_best\": %.1f, \"logical_probes_per_step\": %llu, "
           "\"table_upload_s\": %.3f}\n", T, n_fl, steps, repeats, best, worst, T / (best * 1e-3),
           (unsigned long long)sz.logical_probes, upload_s);
  } else {
    // ---- streaming: main thread = set_targets (H2D of fresh strings) + run; reporter thread = wait for the
    // delivery of each batch in order, km_report_rows on its view, text appended to one buffer
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> state((size_t)n_fl, 0);             // 0 free, 1 running (GPU), reported -> 0
    std::vector<char*> texts;
    std::vector<size_t> text_len;
    uint64_t rows_total = 0, flagged_err = 0, bytes_total = 0, bytes_seen = 0;
    double t_slot = 0, t_set = 0, t_run = 0, t_result = 0, t_report = 0;   // where the two threads spend a step
    long n_acc = 0;
    std::atomic<long> next_report(0);
    long total_steps = 0;
    bool done = false;
    auto reporter = [&](long first) {
      for (long i = first;; ++i) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return i < total_steps || done; });
          if (i >= total_steps && done) return;
        }
        const size_t q = (size_t)(i % n_fl);
        km_batch_out_t view;
        km_batch_sizes_t sz;
        const double r0 = now_s();
        CHECK(km_batch_result(bs[q], &view, &sz));
        const double r1 = now_s();
        km_report_in_t in;
        memset(&in, 0, sizeof in);
        in.n_targets = T; in.bases = ascii.data() + (uint64_t)(i % n_fl) * T * L; in.base_off = off.data();
        in.names = names.data(); in.db_name = "synthetic.jf"; in.k = K; in.res = &view; in.sizes = &sz;
        char* txt = nullptr;
        uint64_t* row_off = nullptr;
        int32_t* err = nullptr;
        CHECK(km_report_rows(&in, &txt, &row_off, &err));
        t_result += r1 - r0; t_report += now_s() - r1;
        for (uint32_t t = 0; t < T; ++t) flagged_err += err[t] != 0;
        // a consumer would write() the text out and release it; here the text of every distinct target set (the
        // first n_fl steps of a repeat) is kept to be hashed outside the timed region, the rest goes straight back
        bytes_seen += row_off[T];
        if (i - first < n_fl) {
          texts.push_back(txt);
          text_len.push_back((size_t)row_off[T]);
          txt = nullptr;
        }
        km_report_free(txt, row_off, err);
        {
          std::lock_guard<std::mutex> lk(mu);
          state[q] = 0;
          next_report = i + 1;
        }
        cv.notify_all();
      }
    };
    auto run_steps = [&](int n) {
      for (char* p : texts) km_report_free(p, nullptr, nullptr);
      texts.clear();
      text_len.clear();
      std::thread rep(reporter, total_steps);
      for (int i = 0; i < n; ++i) {
        const size_t q = (size_t)((total_steps) % n_fl);
        const double m0 = now_s();
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return state[q] == 0; });
        }
        const double m1 = now_s();
        // fresh host strings every step: the set this slot would have in a catalog run
        CHECK(km_batch_set_targets(bs[q], ascii.data() + (uint64_t)q * T * L, off.data(), T));
        const double m2 = now_s();
        CHECK(km_batch_run(bs[q], flags, streams[q]));
        t_slot += m1 - m0; t_set += m2 - m1; t_run += now_s() - m2; ++n_acc;
        {
          std::lock_guard<std::mutex> lk(mu);
          state[q] = 1;
          ++total_steps;
        }
        cv.notify_all();
      }
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return next_report.load() == total_steps; });
        done = true;
      }
      cv.notify_all();
      rep.join();
      done = false;
    };
    run_steps(warmup > n_fl ? warmup : n_fl);
    std::vector<double> ms;
    uint64_t fnv = 0;
    for (int r = 0; r < repeats; ++r) {
      t_slot = t_set = t_run = t_result = t_report = 0; n_acc = 0;
      const double t0 = now_s();
      run_steps(steps);
      ms.push_back((now_s() - t0) / steps * 1e3);
      rows_total = bytes_total = 0;
      fnv = 1469598103934665603ull;
      if (r + 1 == repeats)
        for (size_t b = 0; b < texts.size(); ++b) {             // one pass over every distinct target set
          bytes_total += text_len[b];
          for (size_t i = 0; i < text_len[b]; ++i) { const unsigned char ch = (unsigned char)texts[b][i]; fnv = (fnv ^ ch) * 1099511628211ull; rows_total += ch == '\n'; }
        }
    }
    double best = ms[0], worst = ms[0];
    for (double v : ms) { best = v < best ? v : best; worst = v > worst ? v : worst; }
    printf("{\"mode\": \"e2e\", \"targets_per_step\": %u, \"batches_in_flight\": %d, \"steps\": %d, \"repeats\": %d, "
           "\"ms_per_step_min\": %.6f, \"ms_per_step_max\": %.6f, \"targets_per_s_best\": %.1f, \"hashed_steps\": %zu, \"tsv_rows\": %llu, "
           "\"tsv_bytes\": %llu, \"tsv_fnv1a\": \"%016llx\", \"tsv_bytes_all_steps\": %llu, \"targets_with_report_flags\": %llu, \"table_upload_s\": %.3f, "
           "\"main_thread_ms_per_step\": {\"wait_for_free_batch\": %.3f, \"km_batch_set_targets\": %.3f, \"km_batch_run\": %.3f}, "
           "\"report_thread_ms_per_step\": {\"km_batch_result_wait\": %.3f, \"km_report_rows\": %.3f}, "
           "\"includes\": \"km_batch_set_targets of host strings (H2D) every step, walk + path search, lean delivery (D2H), "
           "km_report_rows on a second thread, text in one buffer\"}\n",
           T, n_fl, steps, repeats, best, worst, T / (best * 1e-3), texts.size(), (unsigned long long)rows_total,
           (unsigned long long)bytes_total, (unsigned long long)fnv, (unsigned long long)bytes_seen, (unsigned long long)flagged_err, upload_s,
           t_slot / n_acc * 1e3, t_set / n_acc * 1e3, t_run / n_acc * 1e3, t_result / n_acc * 1e3, t_report / n_acc * 1e3);
  }
  for (int q = 0; q < n_fl; ++q) {
    CHECK(km_batch_destroy(bs[(size_t)q]));
    CHECK(km_stream_destroy(streams[(size_t)q]));
  }
  CHECK(kmjf_close(db));
  return 0;
}
