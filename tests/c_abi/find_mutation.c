/* A plain-C client of libkmgpu.so: the C-level sequence of INTEGRATION.md §2 for one target
 * file against one .jf database, printing the TSV rows.  Built and run by
 * tests/test_gpu_parity.py::test_plain_c_client (gcc, no Python, no torch in the process). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmgpu.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != KM_OK) {                                                                  \
      fprintf(stderr, "%s failed: %s (%s)\n", #call, km_strerror(rc_), km_last_error()); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

/* concatenated sequence lines of a FASTA file, upper-cased */
static char* read_fasta(const char* path, size_t* len) {
  FILE* f = fopen(path, "r");
  if (!f) return NULL;
  size_t cap = 1 << 16, n = 0;
  char* s = (char*)malloc(cap);
  char line[4096];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '>') continue;
    for (char* p = line; *p && *p != '\n' && *p != '\r'; ++p) {
      if (n + 1 >= cap) s = (char*)realloc(s, cap *= 2);
      s[n++] = (*p >= 'a' && *p <= 'z') ? (char)(*p - 32) : *p;
    }
  }
  fclose(f);
  *len = n;
  return s;
}

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s <target.fa> <db.jf> <query-name>\n", argv[0]); return 2; }
  size_t len = 0;
  char* seq = read_fasta(argv[1], &len);
  if (!seq) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

  kmjf_t* db = NULL;
  CHECK(kmjf_load(argv[2], 0, &db));
  kmjf_info_t info;
  CHECK(kmjf_info(db, &info));

  km_params_t par = {0.05, 5, 500, 10, 10000, 0};            /* -p -c -s -b -n defaults */
  km_batch_t* b = NULL;
  CHECK(km_batch_create(db, &par, 1, len + 64, &b));
  uint64_t off[2] = {0, (uint64_t)len};
  CHECK(km_batch_set_targets(b, (const uint8_t*)seq, off, 1));
  /* kernels + device-side compaction + one asynchronous D2H into the batch's pinned buffer;
   * `out` then points into that buffer (nothing is copied or reorganised on the host) */
  CHECK(km_batch_run(b, KM_STAGE_WALK | KM_STAGE_GRAPH | KM_RUN_DELIVER, NULL));
  km_batch_sizes_t sz;
  km_batch_out_t out;
  CHECK(km_batch_result(b, &out, &sz));
  if (out.status[0] != KM_T_OK) { fprintf(stderr, "target status %u\n", out.status[0]); return 3; }

  const char* names[1] = {argv[3]};
  km_report_in_t in;
  memset(&in, 0, sizeof in);
  in.n_targets = 1;
  in.bases = (const uint8_t*)seq;
  in.base_off = off;
  in.names = names;
  in.db_name = argv[2];
  in.k = info.k;
  in.res = &out;
  char* text = NULL;
  uint64_t* row_off = NULL;
  int32_t* err = NULL;
  CHECK(km_report_rows(&in, &text, &row_off, &err));
  fwrite(text, 1, (size_t)row_off[1], stdout);
  fprintf(stderr, "nodes %llu paths %u logical probes %llu report flag %d\n", (unsigned long long)sz.n_nodes,
          sz.n_paths, (unsigned long long)sz.logical_probes, err[0]);
  km_report_free(text, row_off, err);
  CHECK(km_batch_destroy(b));
  CHECK(kmjf_close(db));
  free(seq);
  return 0;
}
