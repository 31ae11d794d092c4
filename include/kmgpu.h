/* kmgpu.h — C-ABI of libkmgpu.so: the MI355X (gfx950) implementation of km's
 * `find_mutation` hot path.  Plain pointers and sizes only; every function
 * returns an int status (KM_OK == 0) and never throws across the boundary.
 *
 * The reference (iric-soft/km, pure Python) has no FFI of its own: its seam is
 * the duck-type of km/utils/Jellyfish.py plus the SWIG surface of the
 * third-party Jellyfish binding beneath it.  Each entry point below names the
 * reference interface it replaces (file:line into the reference tree).  The
 * ctypes binding a km maintainer would add is shown in INTEGRATION.md and
 * shipped as km_amd/lib.py.
 *
 * Threading: a handle is used from one host thread at a time.  Device work is
 * issued on the caller's HIP stream where a `stream` argument exists
 * (a hipStream_t passed as void*; NULL = the default stream).
 *
 * k-mer encoding (same as Jellyfish keys): A=0 C=1 G=2 T=3, two bits per base,
 * first base in the most significant used bits of a uint64_t (k <= 32).
 */
#ifndef KMGPU_H
#define KMGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
#define KM_OK          0
#define KM_E_IO        1   /* cannot open / read the file                          */
#define KM_E_FORMAT    2   /* not a Jellyfish `binary/sorted` file, bad header     */
#define KM_E_K         3   /* k > 32 (key does not fit a uint64_t) or k < 2        */
#define KM_E_ARG       4   /* NULL / out-of-range argument                         */
#define KM_E_HIP       5   /* a HIP runtime call failed (see km_last_error)        */
#define KM_E_NOMEM     6   /* host or device allocation failed                     */
#define KM_E_STATE     7   /* call order violated (e.g. table not uploaded)        */
#define KM_E_CAPACITY  8   /* a caller-provided output buffer is too small         */

/* ---- per-target status written by the walk ------------------------------ */
#define KM_T_OK          0
#define KM_T_NODE_LIMIT  1  /* len(node_data) > max_node at an extension call:
                               km/utils/MutationFinder.py:143-148 (host turns it
                               back into the same sys.exit message)              */
#define KM_T_REPEAT_KMER 2  /* a k-mer occurs twice in the target: ValueError of
                               km/utils/common.py:55-59                          */
#define KM_T_EMPTY       3  /* target shorter than k: `assert len(ref_mer)`,
                               km/utils/Sequence.py:46                           */
#define KM_T_BAD_BASE    4  /* a character outside ACGTacgt                      */
#define KM_T_INTERNAL    5  /* workspace exhausted even in the large tier        */

typedef struct kmjf kmjf_t;         /* one k-mer count database (host records + device table) */
typedef struct km_batch km_batch_t; /* device workspace for a batch of targets                */

typedef struct {
  int32_t  k;            /* k-mer length (key_len / 2)                        */
  int32_t  canonical;    /* header["canonical"]  (km/utils/Jellyfish.py:45)   */
  uint64_t n_records;    /* records held (count > 0)                          */
  uint64_t n_slots;      /* device table capacity in 16-byte slots (0 = not uploaded) */
  uint64_t n_groups;     /* occupied slots                                    */
  uint64_t table_bytes;  /* n_slots * 16 + bucket directory + side table of counts >= 65535 */
  int32_t  device;       /* HIP device ordinal of the table, -1 if none       */
  int32_t  max_probe;    /* slots a lookup may read: 2 (the home pair) unless a minimizer
                            bucket was too heavy to keep that bound; like n_slots a function of
                            the record SET (any order of the records gives the same table
                            geometry: the build's settle pass)    */
} kmjf_info_t;

/* Walk parameters = the CLI flags of km/argparser/find_mutation.py:5-39 as they
 * reach Jellyfish(cutoff=ratio, n_cutoff=count) and
 * MutationFinder(refpath, jf, steps, branchs, nodes)
 * (km/tools/find_mutation.py:29,49-51). */
typedef struct {
  double   ratio;      /* -p/--ratio   : child kept iff count >= max(sum*ratio, count) */
  int64_t  count;      /* -c/--count                                                   */
  uint32_t max_stack;  /* -s/--steps                                                   */
  uint32_t max_break;  /* -b/--branchs                                                 */
  uint32_t max_node;   /* -n/--nodes                                                   */
  uint32_t reserved;
} km_params_t;

typedef struct {
  uint32_t n_targets;
  uint32_t n_paths;        /* over all targets                                 */
  uint64_t n_nodes;        /* over all targets, caps excluded                  */
  uint64_t n_runs;         /* path run-length records over all paths           */
  uint64_t logical_probes; /* reference-semantics Jellyfish.query calls        */
  uint64_t table_fetches;  /* 16-byte table slots actually read by the walk (KM_RUN_COUNT_FETCHES; else 0) */
  uint32_t n_big_tier;     /* targets that needed the large-workspace pass     */
  uint32_t n_flagged;      /* targets with at least one non-trivial seed       */
  uint64_t seed_probes;    /* logical probes answered by the k_seed kernel     */
  uint64_t n_extra;        /* walk-discovered nodes over all targets (n_nodes minus the targets' own k-mers) */
  uint32_t n_count_escapes;/* entries of count_esc_node / count_esc_value (KM_DELIVER_COUNT16), else 0            */
  uint32_t reserved;
} km_batch_sizes_t;

/* Host-side result arrays.  km_batch_fetch() fills caller-allocated arrays (numpy; any
 * pointer may be NULL to skip that output; sizes come from km_batch_sizes());
 * km_batch_result() instead points them into the batch's pinned delivery buffer (no copy;
 * node_kmer is NULL there: a target's own k-mers are not shipped back to the caller who
 * supplied them, node i < n_ref is the k-mer at base i of the target, and the
 * walk-discovered nodes n_ref.. are in extra_kmer).
 *
 * Node order per target (canonical order): the target's own
 * k-mers in target order (node i == k-mer at position i), then walk-discovered
 * k-mers in registration order.  The two capping nodes of
 * km/utils/MutationFinder.py:97-98,122-123 are implicit: BigBang == n_nodes,
 * BigCrunch == n_nodes + 1.
 *
 * A path (km/utils/Graph.py:220-240, caps stripped as in
 * km/utils/MutationFinder.py:562) is a list of node indices, delivered
 * run-length encoded: consecutive indices (i, i+1, ...) collapse into one
 * (start, length) run.  Paths of one target are sorted by their index sequence. */
typedef struct {
  uint32_t* status;        /* [n_targets]   KM_T_*                                      */
  uint32_t* aux;           /* [n_targets]   REPEAT_KMER: unused (host re-derives pos)   */
  uint32_t* n_ref;         /* [n_targets]   k-mers in the target                        */
  uint64_t* probes;        /* [n_targets]   logical probes                              */
  uint64_t* node_off;      /* [n_targets+1] CSR offsets into node_kmer/node_count       */
  uint64_t* node_kmer;     /* [n_nodes]     packed k-mers                               */
  uint32_t* node_count;    /* [n_nodes]     counts (Jellyfish.query)                    */
  uint32_t* path_off;      /* [n_targets+1] CSR offsets into the per-path arrays        */
  uint64_t* run_off;       /* [n_paths+1]   CSR offsets into run_start/run_len          */
  uint32_t* run_start;     /* [n_runs]                                                  */
  uint32_t* run_len;       /* [n_runs]                                                  */
  uint32_t* path_len;      /* [n_paths]     nodes on the path                           */
  uint32_t* path_min_cov;  /* [n_paths]     min count along the path
                                            (km/utils/MutationFinder.py:639,802)        */
  uint64_t* extra_off;     /* [n_targets+1] CSR offsets into extra_kmer
                                            (extra_off[t+1]-extra_off[t] == nodes of t - n_ref[t]) */
  uint64_t* extra_kmer;    /* [n_extra]     packed k-mers of the walk-discovered nodes     */
  uint32_t* ref_max_cov;   /* [n_targets]   bare-reference targets (the only path is the target's own
                                            k-mer chain, no walk-discovered node): max count over
                                            the target's k-mers; 0xFFFFFFFF for every other target */
  /* KM_DELIVER_COUNT16 (km_batch_result only): node_count is NULL and the counts arrive as 16-bit values,
   * same indexing (node_off); a count >= 65535 reads 0xFFFF there and its exact value is in the escape
   * list, sorted by node index (global index into node_count16).  Half the bytes of a delivery are counts. */
  uint16_t* node_count16;    /* [n_nodes]                                                 */
  uint64_t* count_esc_node;  /* [n_count_escapes] ascending                               */
  uint32_t* count_esc_value; /* [n_count_escapes]                                         */
} km_batch_out_t;

/* ---- database: replaces Jellyfish.__init__ (km/utils/Jellyfish.py:23-45) and
 *      the binding's QueryMerFile / MerDNA.k() (km/utils/Jellyfish.py:24-25) --- */
int kmjf_open(const char* path, kmjf_t** out);
/* Build a database from in-memory records (synthetic workloads, tests, the
 * receiving side of a broadcast).  keys/counts are copied. */
int kmjf_from_records(const uint64_t* keys, const uint32_t* counts, uint64_t n,
                      int k, int canonical, kmjf_t** out);
/* An empty handle whose records live only on a device (multi-GPU receive side). */
int kmjf_create(int k, int canonical, kmjf_t** out);
int kmjf_close(kmjf_t* h);
int kmjf_info(const kmjf_t* h, kmjf_info_t* info);
/* Borrow the host record arrays (valid until kmjf_close). */
int kmjf_records(const kmjf_t* h, const uint64_t** keys, const uint32_t** counts, uint64_t* n);

/* Open + upload in one go, without a host copy of the records: the record area of the
 * memory-mapped file is copied to HBM as it is, unpacked and inserted there (the path for
 * sample-after-sample runs, example/run_leucegene.sh:29-35).  kmjf_records() then reports none. */
int kmjf_load(const char* path, int device, kmjf_t** out);

/* Build the HBM-resident table on `device` from the host records. */
int kmjf_upload(kmjf_t* h, int device);
/* Build the table from record arrays that already sit in device memory
 * (e.g. received through an RCCL broadcast).  The arrays are only read. */
int kmjf_upload_from_device(kmjf_t* h, int device, const uint64_t* d_keys,
                            const uint32_t* d_counts, uint64_t n, void* stream);

/* One database on several GPUs of this process (BASELINE config 4: targets sharded, table replicated; the
 * read-only handle every target of km/tools/find_mutation.py:29,47-58 shares).  `h` holds host records
 * (kmjf_open / kmjf_from_records).  They are uploaded to devices[0], cross the links ONCE as one RCCL
 * broadcast of the 12-byte records (not of the 8.8x larger table), and every device builds its own table:
 * replicas[0] = h (now uploaded on devices[0]), replicas[1..n-1] = new handles for devices[1..n-1]
 * (kmjf_close each).  n == 1 is kmjf_upload.  RCCL is loaded when first needed (librccl.so.1), the library
 * does not link against it; KM_E_HIP if it is missing or a collective fails, KM_E_ARG for n < 1, a null
 * argument or a device named twice.  One process per GPU (torch.distributed, MPI): broadcast the records
 * with the launcher's own collective and call kmjf_upload_from_device (km_amd/dist.py does). */
int kmjf_broadcast(kmjf_t* h, const int* devices, int n, kmjf_t** replicas);

/* ---- lookups: replace Jellyfish.query (km/utils/Jellyfish.py:47-53; also the
 *      loop of common.get_cov, km/utils/common.py:73-92) and
 *      Jellyfish.get_child (km/utils/Jellyfish.py:55-72) ---------------------- */
/* Host arrays in / out. */
int kmjf_query_batch(kmjf_t* h, const uint64_t* kmers, uint64_t n, uint32_t* counts);
/* mask bit c (A=0..T=3) set <=> child `kmer[1:]+c` (forward) or `c+kmer[:-1]`
 * (forward == 0) is kept; counts4[4*i+c] = its count. */
int kmjf_children_batch(kmjf_t* h, const uint64_t* kmers, uint64_t n, double ratio,
                        int64_t n_cutoff, int forward, uint8_t* mask, uint32_t* counts4);
/* Same, device arrays, asynchronous on `stream`. */
int kmjf_query_batch_dev(kmjf_t* h, const uint64_t* d_kmers, uint64_t n, uint32_t* d_counts,
                         void* stream);
int kmjf_children_batch_dev(kmjf_t* h, const uint64_t* d_kmers, uint64_t n, double ratio,
                            int64_t n_cutoff, int forward, uint8_t* d_mask,
                            uint32_t* d_counts4, void* stream);

/* ---- batched walk + path search: replaces, for many targets at once, the loop
 *      body of km/tools/find_mutation.py:47-53:
 *        MutationFinder.__init__ / __extend (km/utils/MutationFinder.py:87-165)
 *        MutationFinder.graph_analysis     (km/utils/MutationFinder.py:496-572)
 *        Graph.init_paths / all_shortest   (km/utils/Graph.py:63-240)
 *        get_counts + min                  (km/utils/MutationFinder.py:490-494,639) */
int km_batch_create(kmjf_t* h, const km_params_t* params, uint32_t max_targets,
                    uint64_t max_total_bases, km_batch_t** out);
int km_batch_destroy(km_batch_t* b);
/* Targets as concatenated ASCII bases (ACGT, either case); offsets[n+1]. Copies H2D. */
int km_batch_set_targets(km_batch_t* b, const uint8_t* bases, const uint64_t* offsets,
                         uint32_t n_targets);
/* Same with device-resident arrays (copied device-to-device, async on stream). */
int km_batch_set_targets_dev(km_batch_t* b, const uint8_t* d_bases, const uint64_t* offsets_host,
                             uint32_t n_targets, void* stream);
#define KM_STAGE_WALK  1
#define KM_STAGE_GRAPH 2
/* OR into `stages`: capture the step into a hipGraph on first use and replay it with one
 * launch afterwards (until the targets change).  Per-kernel HIP-event timings are not
 * available for replayed steps. */
#define KM_RUN_HIPGRAPH 4
/* OR into `stages`: also enqueue result delivery behind the kernels, on the same stream — the
 * results are compacted on the device into their final layout (CSR, paths sorted) and cross
 * PCIe with one asynchronous copy into the batch's pinned buffer; km_batch_result() then only
 * waits for that copy.  Without this flag km_batch_result() / km_batch_fetch() deliver on
 * demand. */
#define KM_RUN_DELIVER 8
/* With KM_RUN_DELIVER: lean delivery.  A bare-reference target (ref_max_cov[t] != 0xFFFFFFFF;
 * typically 70 % of a batch) prints one `Reference` row whose only data are path_min_cov and
 * whether every count is 0 (km/utils/MutationFinder.py:575-648, km/utils/PathQuant.py:144-154):
 * its node_count rows are omitted (node_off[t+1] == node_off[t]) and do not cross PCIe.  Every
 * other target is delivered in full.  km_report_rows accepts both forms; km_batch_fetch always
 * returns every node (re-delivering if the last delivery was lean). */
#define KM_DELIVER_LEAN 16
/* With KM_RUN_DELIVER: node counts cross PCIe as 16-bit values + a short list of the exact counts >= 65535
 * (km_batch_out_t.node_count16 / count_esc_*; at most 2048 of those per delivery, else the library quietly
 * delivers the 32-bit form).  km_report_rows reads either form; km_batch_fetch always fills 32-bit counts. */
#define KM_DELIVER_COUNT16 128
/* Count the 16-byte table slots the walk reads (km_batch_sizes_t.table_fetches; 0 without this flag): a
 * diagnostic — two ballots and one more atomic per wave of k_seed, 2 % of a pipelined step. */
#define KM_RUN_COUNT_FETCHES 256
/* Record the HIP events km_batch_timings reads (seven event records per run; off by default). */
#define KM_RUN_TIMED 32
/* With KM_RUN_TIMED: record the STAGE boundaries only (walk start / end, graph end, delivery) — the walk stage as it
 * runs in production, without the two event records between its three kernels; km_batch_timings [3], [4], [5] are 0. */
#define KM_RUN_TIMED_STAGES 512
/* Kept for callers of round 2: a batch's kernels now ALWAYS run in `stream`, in order (the pass over the
 * unflagged targets, k_graph_pure, used to run beside k_dfs on a side stream unless this flag was given;
 * it follows k_dfs, inside the graph stage of km_batch_timings).  The flag changes nothing. */
#define KM_RUN_SERIAL 64
/* Launch the kernels asynchronously on `stream` (no host synchronisation unless
 * a target overflows the fast tier, in which case the large-tier pass needs one). */
int km_batch_run(km_batch_t* b, int stages, void* stream);
int km_batch_sync(km_batch_t* b);
/* Sizes of the arrays km_batch_fetch fills (a full delivery; km_batch_result reports the sizes
 * of the delivery it returns, which may be lean).  NOTE: after a LEAN km_batch_result both
 * km_batch_sizes and km_batch_fetch deliver again, in full, into the same pinned buffer (which may
 * be reallocated): views handed out by that km_batch_result are invalid afterwards — copy what is
 * still needed first, or ask for the full form only. */
int km_batch_sizes(km_batch_t* b, km_batch_sizes_t* sizes);
int km_batch_fetch(km_batch_t* b, const km_batch_out_t* out);
/* Zero-copy variant: waits for the delivery of the last run (finishing, if some target needed
 * it, the large-workspace tier first) and points `view` into the batch's pinned host buffer;
 * the arrays stay valid until the next km_batch_run / km_batch_set_targets on this batch.
 * Either output may be NULL.  This is what `km find_mutation` needs per target
 * (km/tools/find_mutation.py:49-58) and what km_report_rows consumes. */
int km_batch_result(km_batch_t* b, km_batch_out_t* view, km_batch_sizes_t* sizes);
/* `steps` runs over `n` batches in flight, round robin (batch i % n on streams[i % n]): before a
 * batch is run again its previous delivery is awaited, at the end every batch's.  The loop of a
 * pipelined consumer (km/tools/find_mutation.py:47-58 over successive batches) without an
 * interpreter between the launches. */
int km_batch_pump(km_batch_t* const* batches, void* const* streams, int n, int steps, int stages);
/* What the reference logs with -v from inside the walk and the graph, for the last run of `b` (any output may be
 * NULL): removed_ref_edges[n_targets] / nonref_edges[n_targets] — "Removed %d ref edges." (km/utils/Graph.py:198)
 * and "%d edges in non-ref edge set." (km/utils/Graph.py:231), in this library's node order (the reference's own
 * numbers move by one with its hash seed: its `if last_cur` skips whichever node happens to have index 0);
 * *n_loop_breaks — how often the walk met a k-mer on its stack that was not yet a node ("Broke loop at kmer",
 * km/utils/MutationFinder.py:160-161); loop_pairs[2 * min(*n_loop_breaks, 4096, loop_cap)] — {target, node index
 * of that k-mer} in walk order per target.  Waits for the run (and finishes what it left to the host). */
int km_batch_graph_log(km_batch_t* b, uint32_t* removed_ref_edges, uint32_t* nonref_edges,
                       uint32_t* n_loop_breaks, uint32_t* loop_pairs, uint32_t loop_cap);
/* Durations (ms) of the last run (it must have carried KM_RUN_TIMED; zeros otherwise) measured
 * with HIP events on the launch stream:
 * [0] walk stage (k_pack + k_seed + k_dfs), [1] graph stage, [2] walk + graph,
 * [3] k_seed, [4] k_pack, [5] k_dfs, [6] the delivery kernels (k_out_scan + k_out_pack),
 * [7] the device-to-host copy ([6], [7]: 0 unless the run carried KM_RUN_DELIVER). */
int km_batch_timings(km_batch_t* b, float* ms8);

/* ---- host reporting: replaces, for all targets of a fetched batch at once, the per-target
 *      tail of km/tools/find_mutation.py:53-58 — MutationFinder.graph_analysis' naming and
 *      quantification (km/utils/MutationFinder.py:190-373, 405-488, 575-833), PathQuant
 *      (km/utils/PathQuant.py:37-49, 93-154) and the row order.  Pure host code. */
typedef struct {
  uint32_t n_targets;
  const uint8_t* bases;          /* target sequences as given (ASCII), concatenated           */
  const uint64_t* base_off;      /* [n_targets+1]                                             */
  const char* const* names;      /* [n_targets] NUL-terminated query names                    */
  const char* db_name;           /* the Database column                                       */
  int32_t k;
  int32_t reserved;
  const km_batch_out_t* res;     /* arrays of km_batch_fetch or km_batch_result: everything but
                                    aux / probes / path_len; node_kmer may be NULL when
                                    extra_off / extra_kmer are given                          */
  const km_batch_sizes_t* sizes; /* the lengths of those arrays (km_batch_sizes / km_batch_result), or NULL.
                                    With them every offset array is checked against its array's length
                                    before anything is read (KM_E_ARG on a mismatch); without them the
                                    offsets are taken at their word.  Either way every node index of
                                    a path is checked against its target's node count: a target
                                    whose view is inconsistent gets err 5 and no rows, it is never
                                    read or written out of bounds.                                */
} km_report_in_t;
/* text: the TSV rows of every KM_T_OK target, each row terminated by '\n' (the text as a whole is
 * what `km find_mutation` prints between its header and its trailer);
 * row_off[t] .. row_off[t+1] is the block of target t (empty for other statuses);
 * err[t] != 0 where the reference would have raised while naming a variant
 * (1 IndexError, 2 "mutation identification could be incorrect", 3 AssertionError,
 * 4 ValueError, 5 = the view of this target is inconsistent — offsets not monotone, a path node
 * beyond the target's nodes, counts missing for a target that has variant paths; no rows then), or
 * 100 = rows delivered, but a printed rVAF / expression sits
 * within 1e-6 of a %.3f / %.1f rounding tie, where the last bits of the least-squares solver
 * decide the digit: a caller that needs the reference's exact text recomputes that target with
 * numpy (km_amd.lib.report_rows does).  Release the three arrays with km_report_free.
 * Threads: a team of KM_REPORT_THREADS workers (default min(cores, 16)) kept between calls; each worker places
 * itself on its own CPU once, when it is started, and keeps the process's affinity mask (KM_REPORT_SPREAD=0:
 * placement is left to the scheduler). */
int km_report_rows(const km_report_in_t* in, char** text, uint64_t** row_off, int32_t** err);
void km_report_free(char* text, uint64_t* row_off, int32_t* err);

/* Diagnostics: with KM_SEED_STAMPS set in the environment k_seed records, per wave, eight
 * s_memtime stamps (start, header, bases, minimizer scan, directory words, slots, resolved,
 * end), two s_memrealtime stamps (start, end) and the HW_ID placement, 16 words per wave.
 * dst == NULL only reports the size.  Stamped runs are slower; never use them for timing. */
int km_batch_debug_stamps(km_batch_t* b, uint64_t* dst, uint64_t cap_words, uint64_t* n_words);
/* Diagnostics: device counters of the last run — out4[0] flagged targets (k_seed), [1] unflagged targets
 * the pure-chain pass handed to k_graph, [2] flagged targets the epilogue of k_dfs left to k_graph, [3] 0. */
int km_batch_debug_counts(km_batch_t* b, uint32_t* out4);

/* ---- measurement helpers (bench.py at N = 1 holds no device buffers of its own) ------------- */
int km_device_sync(int device);                                    /* hipDeviceSynchronize on `device`          */
/* device-to-device copy of `bytes` bytes, `reps` times: read + write GB/s (the box's large-copy
 * rate beside the 8 TB/s spec) */
int km_device_copy_GBs(int device, uint64_t bytes, int reps, double* gbs);
/* the two probe kernels alone (rows A2 / A3: Jellyfish.query, get_child — km/utils/Jellyfish.py:47-72)
 * over `n` host k-mers: average launch time over `reps` launches each, and how many have count 0 */
int km_probe_bench(kmjf_t* h, const uint64_t* kmers, uint64_t n, int reps, double ratio, int64_t n_cutoff,
                   double* query_ms, double* children_ms, uint64_t* n_zero);

/* ---- misc ---------------------------------------------------------------- */
const char* km_strerror(int code);
const char* km_last_error(void);   /* thread-local detail of the last failure */
int km_device_count(int* n);
/* A non-blocking HIP stream on `device` for callers that do not bring their own. */
int km_stream_create(int device, void** stream);
int km_stream_destroy(void* stream);
const char* km_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KMGPU_H */
