"""ctypes binding of the C ABI in include/poasta_amd.h (libpoasta_amd.so, built in-tree by hipcc).

There is no fallback: if the shared library is missing or cannot be loaded the import of any compute
entry point raises, and on a box without a HIP device every compute call returns POA_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POA_LIB_PATH") or os.path.join(_HERE, "libpoasta_amd.so")  # override: A/B runs of two builds
CSRC = os.path.join(_HERE, "csrc")

POA_OK = 0
ERRORS = {-1: "POA_ERR_INVALID_ARG", -2: "POA_ERR_NOT_A_DAG", -3: "POA_ERR_NO_DEVICE", -4: "POA_ERR_HIP",
          -5: "POA_ERR_CAPACITY", -6: "POA_ERR_OUT_OF_MEMORY", -7: "POA_ERR_UNSUPPORTED"}
POA_NONE = 0xFFFFFFFF
FLAG_AMBIGUOUS, FLAG_START_QUIRK, FLAG_REF_PANIC, FLAG_SHORT_QUERY, FLAG_TRUNCATED, FLAG_EMPTY_GRAPH = 1, 2, 4, 8, 16, 32

# every symbol include/poasta_amd.h declares (checked by tests/test_abi.py)
EXPORTS = ["poa_version", "poa_last_error", "poa_device_count", "poa_graph_create", "poa_graph_destroy",
           "poa_graph_rows", "poa_graph_node_rows", "poa_graph_update", "poa_align_batch", "poa_align_batch_ex", "poa_align_batch_2piece", "poa_align_batch_2piece_ex", "poa_planes_2piece", "poa_release_cache", "poa_batch_create", "poa_batch_run",
           "poa_batch_run_ex",
           "poa_batch_fetch", "poa_batch_stats", "poa_batch_device_results", "poa_batch_fetch_search_counters", "poa_batch_last_layout", "poa_batch_fetch_planes", "poa_batch_destroy"]


class PoaCosts2(C.Structure):
    _fields_ = [("mismatch", C.c_uint8), ("gap_open1", C.c_uint8), ("gap_extend1", C.c_uint8), ("gap_open2", C.c_uint8),
                ("gap_extend2", C.c_uint8), ("wide_planes", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class PoaCosts(C.Structure):
    _fields_ = [("mismatch", C.c_uint8), ("gap_open", C.c_uint8), ("gap_extend", C.c_uint8), ("reserved", C.c_uint8)]


class PoaBound(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("value", C.c_uint32)]


class PoaConfig(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("heuristic", C.c_uint32), ("pruning", C.c_uint32), ("queue_entries_per_cell", C.c_float),
                ("flags", C.c_uint32), ("span", C.c_uint32), ("qry_free_begin", PoaBound), ("qry_free_end", PoaBound),
                ("graph_free_begin", PoaBound), ("graph_free_end", PoaBound), ("tune", C.c_uint32 * 32)]


# poa_config_t.tune (include/poasta_amd.h: POA_TUNE_*): the library reads no environment variable; A/B scripts and the variant
# sweep set POA_<NAME> and this binding copies them into the config of every call (tune_from_env).
TUNE_KEYS = ("PLANES", "COMPACT", "PACKED", "RELATIVE", "PX", "MF", "MW", "PXMW", "FWD_QUADS", "FUSE_TB", "TB_GROUP", "TB_DEPTH",
             "EXACT_IMPL", "EXACT_LANES", "EXACT_LDS", "WS_LANES", "WS_GROUP", "WS_WAVES", "WS_RING_GLOBAL", "WS_STATIC",
             "WS_CHUNK_CAP", "WS_PROF", "PS_LANES", "PS_LEAN", "TIMING", "WS_ADAPT", "WS_REC")
EXACT_IMPLS = {"lane": 1, "wave": 2, "flat": 3}


def tune_from_env(cfg=None, **overrides):
    """Fill cfg.tune from POA_<NAME> environment variables and keyword overrides (exact_impl="flat", ws_lanes=1, ...).
    Returns cfg (a dense-mode PoaConfig if none was given), or None if nothing is set and no cfg was given."""
    vals = {}
    for i, k in enumerate(TUNE_KEYS):
        v = overrides.get(k.lower(), os.environ.get("POA_" + k))
        if v is None or v == "":
            continue
        vals[i] = EXACT_IMPLS[v] if (k == "EXACT_IMPL" and v in EXACT_IMPLS) else int(v)
    if cfg is None:
        if not vals:
            return None
        cfg = PoaConfig()
    for i, v in vals.items():
        cfg.tune[i] = v + 1
    return cfg


BOUND_UNBOUNDED, BOUND_INCLUDED, BOUND_EXCLUDED = 0, 1, 2
SPAN_GLOBAL, SPAN_ENDS_FREE = 0, 1


MODE_DENSE, MODE_EXACT, MODE_HYBRID = 0, 1, 2
HEURISTIC_DIJKSTRA, HEURISTIC_MINGAP = 0, 1
FLAG_EXACT_OVERFLOW = 0x40
CFG_FULL_PLANES = 1


class PoaStats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("bases", C.c_uint64), ("plane_bytes", C.c_uint64), ("n_queries", C.c_uint32),
                ("n_chunks", C.c_uint32), ("n_forward_launches", C.c_uint32), ("n_flagged", C.c_uint32),
                ("ms_forward", C.c_float), ("ms_traceback", C.c_float), ("ms_h2d", C.c_float), ("ms_d2h", C.c_float),
                ("ms_exact", C.c_float), ("n_exact", C.c_uint32), ("ms_total", C.c_float), ("n_runs", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ }


class PoaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "POA_ERR"), code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP engine for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "poasta_amd.h")]
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("poasta_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.poa_version.restype = C.c_char_p
    L.poa_last_error.restype = C.c_char_p
    L.poa_device_count.restype = C.c_int
    L.poa_graph_create.restype = C.c_int
    L.poa_graph_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.poa_graph_destroy.argtypes = [vp]
    L.poa_graph_destroy.restype = None
    L.poa_graph_rows.restype = C.c_uint32
    L.poa_graph_rows.argtypes = [vp]
    L.poa_graph_node_rows.restype = C.c_int
    L.poa_graph_node_rows.argtypes = [vp, vp]
    L.poa_align_batch.restype = C.c_int
    L.poa_align_batch.argtypes = [vp, C.POINTER(PoaCosts), C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, vp,
                                  C.POINTER(PoaStats), C.c_int]
    L.poa_align_batch_ex.restype = C.c_int
    L.poa_align_batch_ex.argtypes = [vp, C.POINTER(PoaCosts), C.POINTER(PoaConfig), C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, vp,
                                     C.POINTER(PoaStats), C.c_int]
    L.poa_align_batch_2piece.restype = C.c_int
    L.poa_align_batch_2piece.argtypes = [vp, C.POINTER(PoaCosts2), C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, vp,
                                         C.POINTER(PoaStats), C.c_int]
    L.poa_align_batch_2piece_ex.restype = C.c_int
    L.poa_align_batch_2piece_ex.argtypes = [vp, C.POINTER(PoaCosts2), C.POINTER(PoaConfig), C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, vp,
                                            C.POINTER(PoaStats), vp, C.c_int]
    L.poa_planes_2piece.restype = C.c_int
    L.poa_planes_2piece.argtypes = [vp, C.POINTER(PoaCosts2), vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_int]
    L.poa_batch_run_ex.restype = C.c_int
    L.poa_batch_run_ex.argtypes = [vp, C.POINTER(PoaCosts), C.POINTER(PoaConfig), vp]
    L.poa_batch_create.restype = C.c_int
    L.poa_batch_create.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.poa_batch_run.restype = C.c_int
    L.poa_batch_run.argtypes = [vp, C.POINTER(PoaCosts), vp]
    L.poa_batch_fetch.restype = C.c_int
    L.poa_batch_fetch.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, C.POINTER(PoaStats)]
    L.poa_batch_stats.restype = C.c_int
    L.poa_batch_stats.argtypes = [vp, C.POINTER(PoaStats)]
    L.poa_batch_fetch_search_counters.restype = C.c_int
    L.poa_batch_fetch_search_counters.argtypes = [vp, vp]
    L.poa_batch_last_layout.restype = C.c_int
    L.poa_batch_last_layout.argtypes = [vp, vp]
    L.poa_batch_device_results.restype = C.c_int
    L.poa_batch_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.poa_batch_fetch_planes.restype = C.c_int
    L.poa_batch_fetch_planes.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.poa_batch_destroy.argtypes = [vp]
    L.poa_batch_destroy.restype = None
    _lib = L
    return L


def check(rc):
    if rc != POA_OK:
        raise PoaError(rc, lib().poa_last_error().decode(errors="replace"))
