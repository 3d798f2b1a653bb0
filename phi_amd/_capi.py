"""ctypes binding of include/phi_amd.h (libphi_amd.so).

The library is the product: if it is missing or cannot be loaded this module raises, it never
falls back to a CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHI_AMD_LIB") or os.path.join(_HERE, "libphi_amd.so")   # (PHI_AMD_LIB: an experimental build)

PHI_OK = 0
PHI_ERR_INVALID, PHI_ERR_NOMEM, PHI_ERR_DEVICE, PHI_ERR_STATE = -1, -2, -3, -4
PHI_ERR_UNSUPPORTED, PHI_ERR_WALK, PHI_ERR_OVERFLOW = -5, -6, -7
PHI_FLAG_QCLP, PHI_FLAG_MIXED = 1, 2
PHI_COMM_ID_BYTES = 128

# every symbol include/phi_amd.h declares
SYMBOLS = [
    "phi_strerror", "phi_last_error", "phi_ctx_create", "phi_ctx_destroy", "phi_set_stream", "phi_set_params",
    "phi_set_graph", "phi_add_reads", "phi_add_reads_device", "phi_reset_reads", "phi_reads_stats", "phi_hits_buffer",
    "phi_spectrum_export", "phi_spectrum_import", "phi_spectrum_set_size", "phi_solve", "phi_path_sequence",
    "phi_sketch", "phi_walk_minimizers", "phi_walk_sharing", "phi_kept_anchors", "phi_prof_enable", "phi_prof_read",
    "phi_host_register", "phi_host_unregister", "phi_set_solve_budget", "phi_device_synchronize", "phi_walk_text_upload", "phi_walk_text_resolve", "phi_walk_entries",
    "phi_index_stats", "phi_solve_stats", "phi_comm_unique_id", "phi_comm_init", "phi_comm_info", "phi_comm_allreduce_hits", "phi_comm_exchange", "phi_comm_destroy",
    "phi_reads_text_begin", "phi_add_reads_text", "phi_reads_text_end", "phi_reads_text_detach_carry", "phi_reads_text_last_batch",
    "phi_text_park_create", "phi_text_park_pin", "phi_text_park_add", "phi_text_park_add_async", "phi_text_park_wait", "phi_text_park_bytes", "phi_text_park_fetch", "phi_text_park_release", "phi_text_park_destroy",
    "phi_add_reads_text_parked",
    "phi_peers_create", "phi_peers_join", "phi_peers_allreduce_hits", "phi_peers_exchange", "phi_peers_destroy",
    "phi_ipc_unique_id", "phi_ipc_init", "phi_ipc_info", "phi_ipc_allreduce_hits", "phi_ipc_flush", "phi_ipc_exchange", "phi_ipc_check", "phi_ipc_destroy",
]


class PhiIndexInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_entries", "walk_bases", "n_classes", "class_bases", "n_class_records",
                                         "n_walk_minimizers", "n_distinct_minimizers")] + [("sketch_gpu_ms", C.c_double)]


class PhiSolveInfo(C.Structure):
    _fields_ = [("n_dp_anchors", C.c_int64), ("n_events", C.c_int64), ("n_steps", C.c_int32), ("n_blocks", C.c_int32),
                ("dp_mode", C.c_int32), ("max_classes", C.c_int32), ("mean_classes", C.c_double)]


class PhiResult(C.Structure):
    _fields_ = [
        ("objective", C.c_int64), ("upper_bound", C.c_int64), ("optimal", C.c_int32), ("n_dp_runs", C.c_int32),
        ("n_covered", C.c_int64),
        ("n_path", C.c_int64), ("path_vtx", C.POINTER(C.c_int32)), ("path_hap", C.POINTER(C.c_int32)),
        ("recombination_count", C.c_int32), ("n_switches", C.c_int32), ("hap_len", C.c_int64),
        ("n_walks", C.c_int32), ("n_minimizers", C.POINTER(C.c_int64)), ("n_anchors", C.POINTER(C.c_int64)),
        ("spectrum_size", C.c_int64), ("filtered", C.c_int64), ("retained", C.c_int64), ("n_in_model", C.c_int64),
    ]


_lib = None


def load():
    """Load libphi_amd.so and declare the prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m phi_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.phi_strerror.restype = C.c_char_p
    L.phi_strerror.argtypes = [C.c_int]
    L.phi_last_error.restype = C.c_char_p
    L.phi_last_error.argtypes = [vp]
    L.phi_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.phi_ctx_destroy.restype = None
    L.phi_ctx_destroy.argtypes = [vp]
    L.phi_set_stream.argtypes = [vp, vp]
    L.phi_set_params.argtypes = [vp, i32, i32, C.c_float, i32, C.c_uint32]
    L.phi_set_graph.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp]
    L.phi_add_reads.argtypes = [vp, vp, vp, i64]
    L.phi_add_reads_device.argtypes = [vp, vp, vp, i64, i64]
    L.phi_reset_reads.argtypes = [vp]
    L.phi_reads_text_begin.argtypes = [vp, i64]
    L.phi_add_reads_text.argtypes = [vp, vp, i64, C.POINTER(i32)]
    L.phi_reads_text_end.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.phi_reads_text_detach_carry.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.phi_reads_text_last_batch.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.phi_text_park_create.argtypes = [i32, C.POINTER(vp)]
    L.phi_text_park_pin.argtypes = [vp, vp, C.c_size_t]
    L.phi_text_park_add.argtypes = [vp, vp, i64, C.POINTER(i32)]
    L.phi_text_park_add_async.argtypes = [vp, vp, i64, C.POINTER(i32)]
    L.phi_text_park_wait.argtypes = [vp, i32]
    L.phi_text_park_bytes.argtypes = [vp, i32]
    L.phi_text_park_bytes.restype = i64
    L.phi_text_park_fetch.argtypes = [vp, i32, vp, i64]
    L.phi_text_park_release.argtypes = [vp, i32]
    L.phi_text_park_destroy.argtypes = [vp]
    L.phi_text_park_destroy.restype = None
    L.phi_add_reads_text_parked.argtypes = [vp, vp, i32, C.POINTER(i32)]
    L.phi_reads_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.phi_hits_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.phi_spectrum_export.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.phi_spectrum_import.argtypes = [vp, vp, i64]
    L.phi_spectrum_set_size.argtypes = [vp, i64]
    L.phi_solve.argtypes = [vp, C.POINTER(PhiResult)]
    L.phi_set_solve_budget.argtypes = [vp, i64]
    L.phi_index_stats.argtypes = [vp, C.POINTER(PhiIndexInfo)]
    L.phi_solve_stats.argtypes = [vp, C.POINTER(PhiSolveInfo)]
    L.phi_comm_unique_id.argtypes = [vp, C.c_size_t]
    L.phi_comm_init.argtypes = [vp, vp, i32, i32]
    L.phi_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.phi_comm_allreduce_hits.argtypes = [vp]
    L.phi_comm_exchange.argtypes = [vp]
    L.phi_comm_destroy.argtypes = [vp]
    L.phi_peers_create.argtypes = [i32, C.POINTER(vp)]
    L.phi_peers_join.argtypes = [vp, vp, i32]
    L.phi_peers_allreduce_hits.argtypes = [vp]
    L.phi_peers_exchange.argtypes = [vp]
    L.phi_peers_destroy.argtypes = [vp]
    L.phi_ipc_unique_id.argtypes = [vp, C.c_size_t]
    L.phi_ipc_init.argtypes = [vp, C.c_char_p, i32, i32]
    L.phi_ipc_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.phi_ipc_allreduce_hits.argtypes = [vp]
    L.phi_ipc_exchange.argtypes = [vp]
    L.phi_ipc_flush.argtypes = [vp]
    L.phi_ipc_check.argtypes = [vp]
    L.phi_ipc_destroy.argtypes = [vp]
    L.phi_path_sequence.argtypes = [vp, vp, i64]
    L.phi_sketch.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, i64, C.POINTER(i64)]
    L.phi_walk_minimizers.argtypes = [vp, i32, vp, vp, i64, C.POINTER(i64)]
    L.phi_walk_sharing.argtypes = [vp, vp, i32, C.POINTER(i64)]
    L.phi_kept_anchors.argtypes = [vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.phi_prof_enable.argtypes = [vp, C.c_int]
    L.phi_host_register.argtypes = [vp, vp, C.c_size_t]
    L.phi_host_unregister.argtypes = [vp, vp]
    L.phi_prof_read.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i64)]
    L.phi_device_synchronize.argtypes = [vp]
    L.phi_walk_text_upload.argtypes = [vp, vp, i32]
    L.phi_walk_text_resolve.argtypes = [vp, C.c_char_p, i32, vp, i64, i32, vp, C.POINTER(C.c_uint32)]
    L.phi_walk_entries.argtypes = [vp, vp, i64, C.POINTER(i64)]
    for name in SYMBOLS:
        f = getattr(L, name)          # AttributeError here = the library does not export the ABI
        if f.restype is C.c_int and name not in ("phi_strerror", "phi_last_error", "phi_ctx_destroy"):
            f.restype = C.c_int
    _lib = L
    return L
