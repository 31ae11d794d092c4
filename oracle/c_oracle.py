"""TEST INFRASTRUCTURE — ctypes wrapper of the plain-C oracle (oracle/km_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libkmoracle.so")


class Result(C.Structure):
    _fields_ = [("status", C.c_uint32), ("n_ref", C.c_uint32), ("n_nodes", C.c_uint32),
                ("n_paths", C.c_uint32), ("probes", C.c_uint64),
                ("node_kmer", C.POINTER(C.c_uint64)), ("node_count", C.POINTER(C.c_uint32)),
                ("path_off", C.POINTER(C.c_uint32)), ("path_nodes", C.POINTER(C.c_uint32)),
                ("path_min_cov", C.POINTER(C.c_uint32))]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", HERE])
        lib = C.CDLL(LIB)
        lib.ko_open.restype = C.c_void_p
        lib.ko_open.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        lib.ko_close.argtypes = [C.c_void_p]
        lib.ko_query.restype = C.c_uint32
        lib.ko_query.argtypes = [C.c_void_p, C.c_uint64]
        lib.ko_analyse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_int64,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Result)]
        lib.ko_free_result.argtypes = [C.POINTER(Result)]
        _lib = lib
    return _lib


class COracle:
    def __init__(self, keys, counts, k, canonical=True):
        self.lib = load()
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        self.k = k
        self.h = self.lib.ko_open(keys.ctypes.data, counts.ctypes.data, keys.size, k, int(canonical))

    def __del__(self):
        try:
            self.lib.ko_close(self.h)
        except Exception:
            pass

    def analyse(self, codes, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                graph=True):
        """codes: uint8 base codes of one target.  Returns a dict like
        km_oracle.analyse_target (k-mers packed, paths as lists)."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        r = Result()
        self.lib.ko_analyse(self.h, codes.ctypes.data, codes.size, float(ratio), int(count),
                            int(max_stack), int(max_break), int(max_node), 3 if graph else 1,
                            C.byref(r))
        out = {"status": int(r.status), "n_ref": int(r.n_ref), "probes": int(r.probes)}
        n = int(r.n_nodes)
        out["kmers"] = np.ctypeslib.as_array(r.node_kmer, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        out["counts"] = np.ctypeslib.as_array(r.node_count, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        paths, mc = [], []
        if r.n_paths:
            off = np.ctypeslib.as_array(r.path_off, shape=(r.n_paths + 1,))
            nodes = np.ctypeslib.as_array(r.path_nodes, shape=(int(off[-1]),))
            mc = np.ctypeslib.as_array(r.path_min_cov, shape=(r.n_paths,)).tolist()
            paths = [nodes[off[i]:off[i + 1]].tolist() for i in range(r.n_paths)]
        out["paths"], out["min_cov"] = paths, mc
        self.lib.ko_free_result(C.byref(r))
        return out
