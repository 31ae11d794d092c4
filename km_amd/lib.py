"""ctypes binding of libkmgpu.so (include/kmgpu.h).

This is the whole Python<->native boundary: plain pointers and sizes, numpy
arrays own every in/out buffer.  There is NO CPU fallback — if the HIP library
is missing or cannot be loaded, :func:`load` raises.
"""

import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KM_LIBRARY: another build of the same library (diagnostics builds of tools/); it must exist
LIB_PATH = os.environ.get("KM_LIBRARY") or os.path.join(_HERE, "libkmgpu.so")

KM_OK = 0
KM_STAGE_WALK, KM_STAGE_GRAPH, KM_RUN_HIPGRAPH, KM_RUN_DELIVER, KM_DELIVER_LEAN, KM_RUN_TIMED = 1, 2, 4, 8, 16, 32
KM_DELIVER_COUNT16 = 128        # node counts cross PCIe as 16-bit values + the list of the exact counts >= 65535
KM_RUN_COUNT_FETCHES = 256      # count the table slots read (sizes.table_fetches; a diagnostic, 2 % of a step)
KM_RUN_SERIAL = 64
KM_RUN_TIMED_STAGES = 512       # with KM_RUN_TIMED: stage boundaries only (no event records between the walk stage's kernels)
T_OK, T_NODE_LIMIT, T_REPEAT_KMER, T_EMPTY, T_BAD_BASE, T_INTERNAL = range(6)

# every symbol include/kmgpu.h declares (tests check the library exports them all)
SYMBOLS = [
    "kmjf_open", "kmjf_load", "kmjf_from_records", "kmjf_create", "kmjf_close", "kmjf_info", "kmjf_records",
    "kmjf_upload", "kmjf_upload_from_device", "kmjf_broadcast", "kmjf_query_batch", "kmjf_children_batch",
    "kmjf_query_batch_dev", "kmjf_children_batch_dev", "km_batch_create", "km_batch_destroy",
    "km_batch_set_targets", "km_batch_set_targets_dev", "km_batch_run", "km_batch_sync",
    "km_batch_sizes", "km_batch_fetch", "km_batch_result", "km_batch_timings", "km_batch_graph_log", "km_batch_pump", "km_batch_debug_stamps", "km_batch_debug_counts",
    "km_device_sync", "km_device_copy_GBs", "km_probe_bench",
    "km_report_rows", "km_report_free", "km_strerror", "km_last_error",
    "km_device_count", "km_stream_create", "km_stream_destroy", "km_version",
]


class KmError(RuntimeError):
    def __init__(self, code, detail):
        self.code = code
        super().__init__("libkmgpu: %s" % detail)


class JfInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("canonical", C.c_int32), ("n_records", C.c_uint64),
                ("n_slots", C.c_uint64), ("n_groups", C.c_uint64), ("table_bytes", C.c_uint64),
                ("device", C.c_int32), ("max_probe", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("ratio", C.c_double), ("count", C.c_int64), ("max_stack", C.c_uint32),
                ("max_break", C.c_uint32), ("max_node", C.c_uint32), ("reserved", C.c_uint32)]


class BatchSizes(C.Structure):
    _fields_ = [("n_targets", C.c_uint32), ("n_paths", C.c_uint32), ("n_nodes", C.c_uint64),
                ("n_runs", C.c_uint64), ("logical_probes", C.c_uint64),
                ("table_fetches", C.c_uint64), ("n_big_tier", C.c_uint32), ("n_flagged", C.c_uint32),
                ("seed_probes", C.c_uint64), ("n_extra", C.c_uint64),
                ("n_count_escapes", C.c_uint32), ("reserved", C.c_uint32)]


_P16 = C.POINTER(C.c_uint16)
_P32 = C.POINTER(C.c_uint32)
_P64 = C.POINTER(C.c_uint64)


class BatchOut(C.Structure):
    _fields_ = [("status", _P32), ("aux", _P32), ("n_ref", _P32), ("probes", _P64),
                ("node_off", _P64), ("node_kmer", _P64), ("node_count", _P32),
                ("path_off", _P32), ("run_off", _P64), ("run_start", _P32), ("run_len", _P32),
                ("path_len", _P32), ("path_min_cov", _P32), ("extra_off", _P64), ("extra_kmer", _P64),
                ("ref_max_cov", _P32),
                ("node_count16", _P16), ("count_esc_node", _P64), ("count_esc_value", _P32)]


class ReportIn(C.Structure):
    """km_report_in_t (include/kmgpu.h)."""
    _fields_ = [("n_targets", C.c_uint32), ("bases", C.c_void_p), ("base_off", C.c_void_p),
                ("names", C.POINTER(C.c_char_p)), ("db_name", C.c_char_p), ("k", C.c_int32),
                ("reserved", C.c_int32), ("res", C.POINTER(BatchOut)), ("sizes", C.POINTER(BatchSizes))]


_lib = None

# Live native handles, closed at interpreter exit while the HIP runtime is still up: batches first
# (they reference their database), then databases.  Without this a handle that is still alive when
# Python tears its modules down is destroyed by __del__ in arbitrary order, possibly after the HIP
# runtime has unloaded its state (round 2: a SIGSEGV inside __cxa_finalize under rocprofv3).
_LIVE_BATCHES = weakref.WeakSet()
_LIVE_DATABASES = weakref.WeakSet()


def close_all_handles():
    for b in list(_LIVE_BATCHES):
        try:
            b.close()
        except Exception:
            pass
    for d in list(_LIVE_DATABASES):
        try:
            d.close()
        except Exception:
            pass


atexit.register(close_all_handles)


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same
    SONAME as /opt/rocm's): whichever is loaded first serves both libkmgpu.so and torch, and torch
    finds no GPU when it is the system one.  So when torch is installed but not imported yet (a later
    `import torch` — km_amd.dist, bench.py — must keep working) ITS runtime is loaded first.
    KM_HIP_RUNTIME=system keeps the ROCm installation's (a process that will never import torch: a
    single-GPU `km find_mutation`, a C consumer); KM_HIP_RUNTIME=torch insists on torch's and fails
    loudly without it.  Returns a short description of what was bound (KM_VERBOSE_LOAD=1 prints it)."""
    import importlib.util
    want = os.environ.get("KM_HIP_RUNTIME", "")
    if "torch" in sys.modules:
        return "torch already imported: its HIP runtime serves libkmgpu.so"
    if want == "system":
        return "system HIP runtime (KM_HIP_RUNTIME=system)"
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        if want == "torch":
            raise ImportError("km_amd: KM_HIP_RUNTIME=torch but torch is not installed")
        return "system HIP runtime (no torch installed)"
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        if want == "torch":
            raise ImportError("km_amd: KM_HIP_RUNTIME=torch but %s does not exist" % cand)
        return "system HIP runtime (torch ships no libamdhip64.so)"
    try:
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        if want == "torch":
            raise ImportError("km_amd: cannot load %s: %s" % (cand, exc))
        sys.stderr.write("km_amd: could not preload torch's HIP runtime (%s): %s — libkmgpu.so binds the system one; "
                         "a later `import torch` may then find no GPU (KM_HIP_RUNTIME=system silences this)\n" % (cand, exc))
        return "system HIP runtime (preload of torch's failed)"
    return "torch's bundled HIP runtime %s (KM_HIP_RUNTIME=system to use the ROCm installation's)" % cand


HIP_RUNTIME_BOUND = None


def load():
    """Load libkmgpu.so (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("km_amd: %s is missing — run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (there is no CPU fallback)" % LIB_PATH)
    global HIP_RUNTIME_BOUND
    HIP_RUNTIME_BOUND = _share_hip_runtime_with_torch()
    if os.environ.get("KM_VERBOSE_LOAD"):
        sys.stderr.write("km_amd: %s; library %s\n" % (HIP_RUNTIME_BOUND, LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    vp, cp = C.c_void_p, C.c_char_p
    u32, u64, i32, i64, dbl = C.c_uint32, C.c_uint64, C.c_int, C.c_int64, C.c_double
    sig = {
        "kmjf_open": [cp, C.POINTER(vp)],
        "kmjf_load": [cp, i32, C.POINTER(vp)],
        "kmjf_from_records": [vp, vp, u64, i32, i32, C.POINTER(vp)],
        "kmjf_create": [i32, i32, C.POINTER(vp)],
        "kmjf_close": [vp],
        "kmjf_info": [vp, C.POINTER(JfInfo)],
        "kmjf_records": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)],
        "kmjf_upload": [vp, i32],
        "kmjf_upload_from_device": [vp, i32, vp, vp, u64, vp],
        "kmjf_broadcast": [vp, C.POINTER(i32), i32, C.POINTER(vp)],
        "kmjf_query_batch": [vp, vp, u64, vp],
        "kmjf_children_batch": [vp, vp, u64, dbl, i64, i32, vp, vp],
        "kmjf_query_batch_dev": [vp, vp, u64, vp, vp],
        "kmjf_children_batch_dev": [vp, vp, u64, dbl, i64, i32, vp, vp, vp],
        "km_batch_create": [vp, C.POINTER(Params), u32, u64, C.POINTER(vp)],
        "km_batch_destroy": [vp],
        "km_batch_set_targets": [vp, vp, vp, u32],
        "km_batch_set_targets_dev": [vp, vp, vp, u32, vp],
        "km_batch_run": [vp, i32, vp],
        "km_batch_sync": [vp],
        "km_batch_sizes": [vp, C.POINTER(BatchSizes)],
        "km_batch_fetch": [vp, C.POINTER(BatchOut)],
        "km_batch_result": [vp, C.POINTER(BatchOut), C.POINTER(BatchSizes)],
        "km_batch_timings": [vp, C.POINTER(C.c_float)],
        "km_batch_graph_log": [vp, vp, vp, C.POINTER(u32), vp, u32],
        "km_batch_pump": [C.POINTER(vp), C.POINTER(vp), i32, i32, i32],
        "km_batch_debug_stamps": [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)],
        "km_batch_debug_counts": [vp, C.POINTER(C.c_uint32)],
        "km_report_rows": [C.POINTER(ReportIn), C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint64)),
                           C.POINTER(C.POINTER(C.c_int32))],
        "km_device_count": [C.POINTER(i32)],
        "km_device_sync": [i32],
        "km_device_copy_GBs": [i32, u64, i32, C.POINTER(dbl)],
        "km_probe_bench": [vp, vp, u64, i32, dbl, i64, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(u64)],
        "km_stream_create": [i32, C.POINTER(vp)],
        "km_stream_destroy": [vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    for name in ("km_strerror", "km_last_error", "km_version"):
        getattr(lib, name).restype = cp
    lib.km_strerror.argtypes = [i32]
    lib.km_report_free.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.km_report_free.restype = None
    _lib = lib
    return lib


def pump(batches, streams, steps, stages):
    """km_batch_pump: `steps` runs over the batches in flight, round robin, deliveries awaited."""
    lib = load()
    n = len(batches)
    bs = (C.c_void_p * n)(*[b._b for b in batches])
    sts = (C.c_void_p * n)(*[C.c_void_p(s or 0) for s in streams])
    check(lib.km_batch_pump(bs, sts, n, int(steps), int(stages)))


def device_sync(device=0):
    check(load().km_device_sync(int(device)))


def device_copy_GBs(device=0, nbytes=1 << 30, reps=10):
    out = C.c_double()
    check(load().km_device_copy_GBs(int(device), int(nbytes), int(reps), C.byref(out)))
    return float(out.value)


def probe_bench(db, kmers, reps=10, ratio=0.05, n_cutoff=5):
    """(query_ms, children_ms, n_zero): k_query / k_children alone over `kmers` (numpy uint64)."""
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
    q, c, z = C.c_double(), C.c_double(), C.c_uint64()
    check(load().km_probe_bench(db._h, ptr(kmers), kmers.size, int(reps), float(ratio), int(n_cutoff),
                                C.byref(q), C.byref(c), C.byref(z)))
    return float(q.value), float(c.value), int(z.value)


def stream_create(device=0):
    """A non-blocking HIP stream handle (int) from the library."""
    st = C.c_void_p()
    check(load().km_stream_create(int(device), C.byref(st)))
    return st.value


def stream_destroy(stream):
    check(load().km_stream_destroy(C.c_void_p(stream)))


def check(rc):
    if rc != KM_OK:
        lib = load()
        detail = lib.km_last_error().decode() or lib.km_strerror(rc).decode()
        raise KmError(rc, detail)


def ptr(arr):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)


def typed_ptr(arr, ctype):
    return None if arr is None else arr.ctypes.data_as(C.POINTER(ctype))


class Database:
    """One k-mer count database: host records + HBM-resident table.

    Replaces the ``QueryMerFile`` handle of km/utils/Jellyfish.py:24."""

    def __init__(self, handle):
        self._h = handle
        self._lib = load()
        _LIVE_DATABASES.add(self)

    @classmethod
    def open(cls, path):
        lib = load()
        h = C.c_void_p()
        check(lib.kmjf_open(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path, device=0):
        """Open + upload without a host copy of the records (direct file -> HBM ingestion)."""
        lib = load()
        h = C.c_void_p()
        check(lib.kmjf_load(os.fsencode(path), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_records(cls, keys, counts, k, canonical=True):
        lib = load()
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        assert keys.shape == counts.shape
        h = C.c_void_p()
        check(lib.kmjf_from_records(ptr(keys), ptr(counts), keys.size, int(k), int(bool(canonical)),
                                    C.byref(h)))
        return cls(h)

    @classmethod
    def empty(cls, k, canonical=True):
        lib = load()
        h = C.c_void_p()
        check(lib.kmjf_create(int(k), int(bool(canonical)), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h is not None:
            self._lib.kmjf_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self):
        inf = JfInfo()
        check(self._lib.kmjf_info(self._h, C.byref(inf)))
        return inf

    def records(self):
        """(keys, counts) copies of the host records."""
        kp, cp_, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        check(self._lib.kmjf_records(self._h, C.byref(kp), C.byref(cp_), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
        keys = np.ctypeslib.as_array(C.cast(kp, _P64), shape=(n.value,)).copy()
        counts = np.ctypeslib.as_array(C.cast(cp_, _P32), shape=(n.value,)).copy()
        return keys, counts

    def upload(self, device=0):
        check(self._lib.kmjf_upload(self._h, int(device)))
        return self

    def broadcast(self, devices):
        """One process, several GPUs (kmjf_broadcast): the host records go to devices[0], cross the links once as
        one RCCL broadcast and every device builds its own table.  Returns one Database per device; the first is
        this one."""
        devices = [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)
        out = (C.c_void_p * max(1, len(devices)))()
        check(self._lib.kmjf_broadcast(self._h, arr, len(devices), out))
        return [self] + [Database(C.c_void_p(out[i])) for i in range(1, len(devices))]

    def upload_from_device(self, device, d_keys_ptr, d_counts_ptr, n, stream=None):
        check(self._lib.kmjf_upload_from_device(self._h, int(device), C.c_void_p(d_keys_ptr),
                                                C.c_void_p(d_counts_ptr), int(n),
                                                C.c_void_p(stream or 0)))
        return self

    def query_dev(self, d_kmers_ptr, n, d_counts_ptr, stream=None):
        """Device arrays in / out, asynchronous on `stream`."""
        check(self._lib.kmjf_query_batch_dev(self._h, C.c_void_p(d_kmers_ptr), int(n),
                                             C.c_void_p(d_counts_ptr), C.c_void_p(stream or 0)))

    def children_dev(self, d_kmers_ptr, n, ratio, n_cutoff, d_mask_ptr, d_counts4_ptr, forward=True,
                     stream=None):
        check(self._lib.kmjf_children_batch_dev(self._h, C.c_void_p(d_kmers_ptr), int(n), float(ratio),
                                                int(n_cutoff), int(bool(forward)),
                                                C.c_void_p(d_mask_ptr), C.c_void_p(d_counts4_ptr),
                                                C.c_void_p(stream or 0)))

    def query(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(kmers.size, dtype=np.uint32)
        check(self._lib.kmjf_query_batch(self._h, ptr(kmers), kmers.size, ptr(out)))
        return out

    def children(self, kmers, ratio, n_cutoff, forward=True):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        mask = np.zeros(kmers.size, dtype=np.uint8)
        counts4 = np.zeros((kmers.size, 4), dtype=np.uint32)
        check(self._lib.kmjf_children_batch(self._h, ptr(kmers), kmers.size, float(ratio),
                                            int(n_cutoff), int(bool(forward)), ptr(mask),
                                            ptr(counts4)))
        return mask, counts4


class Batch:
    """Device workspace that walks + path-searches many targets per launch."""

    def __init__(self, db, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                 max_targets=1024, max_total_bases=1 << 20):
        self._lib = load()
        self.db = db
        self.params = Params(float(ratio), int(count), int(max_stack), int(max_break),
                             int(max_node), 0)
        self._b = C.c_void_p()
        check(self._lib.km_batch_create(db._h, C.byref(self.params), int(max_targets),
                                        int(max_total_bases), C.byref(self._b)))
        self.n_targets = 0
        _LIVE_BATCHES.add(self)

    def close(self):
        if self._b is not None and self._b.value:
            self._lib.km_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_targets(self, seqs):
        """seqs: list of str/bytes (ACGT)."""
        blob, offs = pack_sequences(seqs)
        self.set_targets_packed(blob, offs)

    def set_targets_packed(self, blob, offsets):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        check(self._lib.km_batch_set_targets(self._b, ptr(blob), ptr(offsets), offsets.size - 1))
        self.n_targets = offsets.size - 1

    def set_targets_dev(self, d_bases_ptr, offsets, stream=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        check(self._lib.km_batch_set_targets_dev(self._b, C.c_void_p(d_bases_ptr), ptr(offsets),
                                                 offsets.size - 1, C.c_void_p(stream or 0)))
        self.n_targets = offsets.size - 1

    def run(self, stages=KM_STAGE_WALK | KM_STAGE_GRAPH, stream=None):
        check(self._lib.km_batch_run(self._b, int(stages), C.c_void_p(stream or 0)))

    def sync(self):
        check(self._lib.km_batch_sync(self._b))

    def timings(self):
        ms = (C.c_float * 8)()
        check(self._lib.km_batch_timings(self._b, ms))
        return tuple(float(x) for x in ms)

    def graph_log(self, n_targets):
        """What the reference logs with -v from inside the walk and the graph (km_batch_graph_log): per target the
        reference edges stripped and the edges kept, and per target the node indices at which the walk broke a
        loop, in walk order.  -> (removed uint32[n], nonref uint32[n], {target: [node index, ...]}, n_loop_breaks)."""
        n = int(n_targets)
        removed = np.zeros(max(1, n), dtype=np.uint32)
        nonref = np.zeros(max(1, n), dtype=np.uint32)
        cap = 4096
        pairs = np.zeros(2 * cap, dtype=np.uint32)
        n_loops = C.c_uint32(0)
        check(self._lib.km_batch_graph_log(self._b, ptr(removed), ptr(nonref), C.byref(n_loops), ptr(pairs), cap))
        loops = {}
        for t, node in pairs[:2 * min(cap, n_loops.value)].reshape(-1, 2).tolist():
            if node == 0xFFFFFFFF:               # the large tier starts this target's walk over
                loops.pop(t, None)
            else:
                loops.setdefault(t, []).append(node)
        n_real = sum(len(v) for v in loops.values())
        return removed[:n], nonref[:n], loops, n_real

    def debug_counts(self):
        """(flagged targets, unflagged ones handed to k_graph, flagged ones the epilogue of k_dfs left to k_graph)."""
        out = (C.c_uint32 * 4)()
        check(self._lib.km_batch_debug_counts(self._b, out))
        return int(out[0]), int(out[1]), int(out[2])

    def debug_stamps(self):
        """k_seed time stamps (KM_SEED_STAMPS diagnostics): uint64 array [n_waves, 16]."""
        n = C.c_uint64(0)
        check(self._lib.km_batch_debug_stamps(self._b, None, 0, C.byref(n)))
        out = np.zeros(int(n.value), dtype=np.uint64)
        if n.value:
            check(self._lib.km_batch_debug_stamps(self._b, out.ctypes.data_as(C.c_void_p), n, C.byref(n)))
        return out.reshape(-1, 16)

    def sizes(self):
        """Sizes of a FULL delivery.  After a lean result() this (like fetch()) delivers again into the
        same pinned buffer: arrays returned by that result() must not be used afterwards."""
        s = BatchSizes()
        check(self._lib.km_batch_sizes(self._b, C.byref(s)))
        return s

    def wait_result(self):
        """Wait for the delivery of the last run (km_batch_result without building views);
        returns the batch sizes."""
        s = BatchSizes()
        check(self._lib.km_batch_result(self._b, None, C.byref(s)))
        return s

    def result(self):
        """Results of the last run as numpy views INTO the batch's pinned delivery buffer (no
        copy; valid until the next run / set_targets on this batch).  `node_kmer` is absent:
        node i < n_ref of a target is the k-mer at base i of the target, the walk-discovered
        nodes are in `extra_kmer` (CSR `extra_off`)."""
        out, s = BatchOut(), BatchSizes()
        check(self._lib.km_batch_result(self._b, C.byref(out), C.byref(s)))
        n = s.n_targets

        def view(p, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(p, shape=(int(count),))

        extra = {}
        if not out.node_count and out.node_count16:
            # KM_DELIVER_COUNT16: the 16-bit form as delivered, and the counts as every consumer here reads them
            c16 = view(out.node_count16, s.n_nodes, np.uint16)
            esc_n = view(out.count_esc_node, s.n_count_escapes, np.uint64)
            esc_v = view(out.count_esc_value, s.n_count_escapes, np.uint32)
            counts = c16.astype(np.uint32)
            if s.n_count_escapes:
                counts[esc_n.astype(np.int64)] = esc_v
            extra = {"node_count16": c16, "count_esc_node": esc_n, "count_esc_value": esc_v, "node_count": counts}
        res = {
            "status": view(out.status, n, np.uint32), "n_ref": view(out.n_ref, n, np.uint32),
            "probes": view(out.probes, n, np.uint64), "node_off": view(out.node_off, n + 1, np.uint64),
            "extra_off": view(out.extra_off, n + 1, np.uint64),
            "ref_max_cov": view(out.ref_max_cov, n, np.uint32),
            "node_count": extra["node_count"] if extra else view(out.node_count, s.n_nodes, np.uint32),
            "extra_kmer": view(out.extra_kmer, s.n_extra, np.uint64),
            "path_off": view(out.path_off, n + 1, np.uint32),
            "run_off": view(out.run_off, s.n_paths + 1, np.uint64),
            "run_start": view(out.run_start, s.n_runs, np.uint32),
            "run_len": view(out.run_len, s.n_runs, np.uint32),
            "path_len": view(out.path_len, s.n_paths, np.uint32),
            "path_min_cov": view(out.path_min_cov, s.n_paths, np.uint32),
            "logical_probes": int(s.logical_probes), "table_fetches": int(s.table_fetches),
            "n_big_tier": int(s.n_big_tier), "n_flagged": int(s.n_flagged),
        }
        res.update(extra)
        return res

    def fetch(self, nodes=True, paths=True):
        """Copy results to host numpy arrays (dict), node k-mers rebuilt in full."""
        s = self.sizes()
        n = s.n_targets
        r = {
            "status": np.zeros(n, np.uint32), "n_ref": np.zeros(n, np.uint32),
            "probes": np.zeros(n, np.uint64), "node_off": np.zeros(n + 1, np.uint64),
            "logical_probes": int(s.logical_probes), "table_fetches": int(s.table_fetches),
            "n_big_tier": int(s.n_big_tier),
        }
        out = BatchOut()
        out.status = typed_ptr(r["status"], C.c_uint32)
        out.n_ref = typed_ptr(r["n_ref"], C.c_uint32)
        out.probes = typed_ptr(r["probes"], C.c_uint64)
        out.node_off = typed_ptr(r["node_off"], C.c_uint64)
        if nodes:
            r["node_kmer"] = np.zeros(max(1, s.n_nodes), np.uint64)
            r["node_count"] = np.zeros(max(1, s.n_nodes), np.uint32)
            out.node_kmer = typed_ptr(r["node_kmer"], C.c_uint64)
            out.node_count = typed_ptr(r["node_count"], C.c_uint32)
        if paths:
            r["path_off"] = np.zeros(n + 1, np.uint32)
            r["run_off"] = np.zeros(s.n_paths + 1, np.uint64)
            r["run_start"] = np.zeros(max(1, s.n_runs), np.uint32)
            r["run_len"] = np.zeros(max(1, s.n_runs), np.uint32)
            r["path_len"] = np.zeros(max(1, s.n_paths), np.uint32)
            r["path_min_cov"] = np.zeros(max(1, s.n_paths), np.uint32)
            out.path_off = typed_ptr(r["path_off"], C.c_uint32)
            out.run_off = typed_ptr(r["run_off"], C.c_uint64)
            out.run_start = typed_ptr(r["run_start"], C.c_uint32)
            out.run_len = typed_ptr(r["run_len"], C.c_uint32)
            out.path_len = typed_ptr(r["path_len"], C.c_uint32)
            out.path_min_cov = typed_ptr(r["path_min_cov"], C.c_uint32)
        check(self._lib.km_batch_fetch(self._b, C.byref(out)))
        if nodes:
            r["node_kmer"] = r["node_kmer"][: s.n_nodes]
            r["node_count"] = r["node_count"][: s.n_nodes]
        if paths:
            r["run_start"] = r["run_start"][: s.n_runs]
            r["run_len"] = r["run_len"][: s.n_runs]
            r["path_len"] = r["path_len"][: s.n_paths]
            r["path_min_cov"] = r["path_min_cov"][: s.n_paths]
        return r


def expand_path(res, p):
    """Node indices of path `p` (global path number) from the run-length records."""
    a, b = int(res["run_off"][p]), int(res["run_off"][p + 1])
    if b == a:
        return np.zeros(0, np.int64)
    return np.concatenate([np.arange(s, s + l, dtype=np.int64)
                           for s, l in zip(res["run_start"][a:b].tolist(),
                                           res["run_len"][a:b].tolist())])


REPORT_ERRORS = {1: IndexError("list index out of range"),
                 2: Exception("mutation identification could be incorrect"),
                 3: AssertionError(), 4: ValueError("min() arg is an empty sequence"),
                 5: RuntimeError("km_report_rows: the result view of this target is inconsistent")}


def _python_rows(res, t, name, seq, k, db_name):
    """Rows of target t through the Python restatement (km_amd/report.py)."""
    from . import report
    a, e = int(res["node_off"][t]), int(res["node_off"][t + 1])
    p0, p1 = int(res["path_off"][t]), int(res["path_off"][t + 1])
    paths = [expand_path(res, p) for p in range(p0, p1)]
    seq = seq if isinstance(seq, str) else bytes(seq).decode("ascii")
    if "node_kmer" in res:
        kmers = np.asarray(res["node_kmer"][a:e])
    else:                                              # delivery view: own k-mers from the sequence
        from . import kmer as km
        x0, x1 = int(res["extra_off"][t]), int(res["extra_off"][t + 1])
        kmers = np.concatenate([km.sliding_kmers(km.encode(seq), k)[:int(res["n_ref"][t])],
                                np.asarray(res["extra_kmer"][x0:x1])])
    tr = report.TargetResult(name, seq, k, int(res["n_ref"][t]), kmers,
                             np.asarray(res["node_count"][a:e]), paths,
                             np.asarray(res["path_min_cov"][p0:p1]).tolist())
    return report.target_rows(tr, db_name)


def pack_sequences(seqs):
    """(blob, offsets): the sequences as one uint8 array + uint64 offsets[n+1]."""
    enc = [s.encode("ascii") if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    np.cumsum([len(e) for e in enc], out=offs[1:])
    return np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8), offs


def report_rows(res, names, seqs, k, db_name, packed=None):
    """TSV rows of every target of a fetched batch (dict from Batch.fetch) through the C++
    reporting path: returns a list with, per target, a list of row strings (empty unless the
    target's status is KM_T_OK) or the exception the reference would have raised.
    `packed` = pack_sequences(seqs) when the caller already has it."""
    text, row_off, special = report_text(res, names, seqs, k, db_name, packed)
    result = []
    for t in range(len(names)):
        if t in special:
            result.append(special[t])
        else:
            a, e = int(row_off[t]), int(row_off[t + 1])
            result.append(text[a:e].splitlines() if e > a else [])
    return result


def report_text(res, names, seqs, k, db_name, packed=None):
    """The same as one string: (text, row_off, special).  text[row_off[t]:row_off[t+1]] is the block
    of target t, every row terminated by a newline; `special` maps the few targets that are NOT
    served by the text to their row list (numpy recomputation on a rounding tie) or to the exception
    the reference would have raised.  When `special` is empty, `text` is the whole TSV body."""
    lib = load()
    n = len(names)
    blob, offs = packed if packed is not None else pack_sequences(seqs)
    out = BatchOut()
    keep = []
    for field, ctype in (("status", C.c_uint32), ("n_ref", C.c_uint32), ("probes", C.c_uint64),
                         ("node_off", C.c_uint64), ("node_kmer", C.c_uint64), ("node_count", C.c_uint32),
                         ("path_off", C.c_uint32), ("run_off", C.c_uint64), ("run_start", C.c_uint32),
                         ("run_len", C.c_uint32), ("path_len", C.c_uint32), ("path_min_cov", C.c_uint32),
                         ("extra_off", C.c_uint64), ("extra_kmer", C.c_uint64), ("ref_max_cov", C.c_uint32)):
        if field not in res:
            continue                                   # a delivery view has no node_kmer, a fetch no extra_*
        if field == "node_count" and "node_count16" in res:
            continue                                   # the 16-bit form goes to the library as it came
        arr = np.ascontiguousarray(res[field], dtype=np.dtype(ctype))
        if arr.size == 0:
            arr = np.zeros(1, dtype=arr.dtype)
        keep.append(arr)
        setattr(out, field, typed_ptr(arr, ctype))
    n_esc = 0
    if "node_count16" in res:
        n_esc = int(np.asarray(res["count_esc_node"]).size)
        for field, ctype in (("node_count16", C.c_uint16), ("count_esc_node", C.c_uint64), ("count_esc_value", C.c_uint32)):
            arr = np.ascontiguousarray(res[field], dtype=np.dtype(ctype))
            if arr.size == 0:
                arr = np.zeros(1, dtype=arr.dtype)
            keep.append(arr)
            setattr(out, field, typed_ptr(arr, ctype))
    inp = ReportIn()
    inp.n_targets = n
    inp.bases = blob.ctypes.data
    inp.base_off = offs.ctypes.data
    name_arr = (C.c_char_p * max(1, n))(*[nm.encode() for nm in names])
    inp.names = name_arr
    inp.db_name = str(db_name).encode()
    inp.k = int(k)
    inp.res = C.pointer(out)
    # the lengths of the arrays, so that the library checks every offset against them first
    sz = BatchSizes()
    sz.n_targets = n
    sz.n_nodes = int(np.asarray(res["node_count"]).size)
    sz.n_paths = int(np.asarray(res["path_min_cov"]).size)
    sz.n_runs = int(np.asarray(res["run_start"]).size)
    sz.n_extra = int(np.asarray(res["extra_kmer"]).size) if "extra_kmer" in res else 0
    sz.n_count_escapes = n_esc
    inp.sizes = C.pointer(sz)
    text = C.c_void_p()
    row_off = C.POINTER(C.c_uint64)()
    err = C.POINTER(C.c_int32)()
    check(lib.km_report_rows(C.byref(inp), C.byref(text), C.byref(row_off), C.byref(err)))
    try:
        total = int(row_off[n])
        blob_out = C.string_at(text, total).decode("ascii")
        offs_out = np.ctypeslib.as_array(row_off, shape=(n + 1,)).copy()
        errs = np.ctypeslib.as_array(err, shape=(max(1, n),))[:n]
        special = {}
        for t in np.nonzero(errs)[0].tolist():
            if errs[t] == 100:
                # a printed value sits on a %.1f / %.3f rounding tie: the digit depends on the
                # last bits of the least-squares solver — recompute with numpy, like the reference
                special[t] = _python_rows(res, t, names[t], seqs[t], k, db_name)
            else:
                special[t] = REPORT_ERRORS.get(int(errs[t]), RuntimeError("report error %d" % errs[t]))
    finally:
        lib.km_report_free(text, row_off, err)
    return blob_out, offs_out, special
