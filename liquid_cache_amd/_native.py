"""ctypes binding of libliquid_cache_amd.so (the C ABI in include/liquid_cache_amd.h).

The product path has NO fallback: if the in-tree HIP library is missing this module raises at import time of the
symbols, and if no HIP device is present every compute call returns LC_ERR_DEVICE (surfaced as LiquidCacheError).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LC_LIB_PATH: A/B runs of two in-tree builds of the same library (kernel tuning); the default is the in-tree build
LIB_PATH = os.environ.get("LC_LIB_PATH") or os.path.join(_HERE, "libliquid_cache_amd.so")

LC_OK, LC_NOT_STAGED, LC_UNSUPPORTED, LC_NEEDS_BACKING = 0, 1, 2, 3
LC_ERR_INVALID, LC_ERR_CORRUPT, LC_ERR_DEVICE, LC_ERR_OOM, LC_ERR_NO_SYMTAB = -1, -2, -3, -4, -5
OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_LIKE, OP_NOT_LIKE = range(8)
LIT_I64, LIT_U64, LIT_F32, LIT_F64, LIT_BYTES, LIT_I128, LIT_BOOL = range(7)
HINT_NONE, HINT_SUBSTRING_SEARCH, HINT_PREDICATE_COLUMN = 0, 1, 2
OPT_SIGNATURE_INDEX, OPT_ROW_LISTS, OPT_HOST_BUILT_INDEX, OPT_LIKE_PIPELINE_MIN_ENTRIES, OPT_LIKE_PATH, OPT_LIKE_MANY_HINT = 1, 2, 3, 4, 5, 6
OPT_LIKE_INDEX_BUDGET_BYTES, OPT_LIKE_INDEX_CACHE = 7, 8
OPT_LIKE_INDEX_ASYNC, OPT_SCAN_CACHE, OPT_COMM_SHARED_MEMORY = 9, 10, 11
HITS_COUNTERS_ZEROED, GATHER_SLOTTED, GATHER_SLOT_BYTES = 1, 2, 128  # flags of the hit-list calls
HITS_PARTITIONED, HITS_PARTITIONS, HITS_COUNTER_STRIDE = 4, 16, 16  # the partitioned list form (round 6)


class LiquidCacheError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"liquid_cache_amd status {status}: {message}")
        self.status = status


class FilterStep(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_terms", C.c_uint32), ("scans", C.POINTER(C.c_void_p)), ("preds", C.c_void_p)]


class Predicate(C.Structure):
    _fields_ = [("op", C.c_int32), ("lit_tag", C.c_int32), ("lit", C.c_void_p), ("lit_len", C.c_uint64)]


class DeviceInfo(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("compute_units", C.c_int32), ("hbm_total_bytes", C.c_uint64),
                ("hbm_staged_bytes", C.c_uint64), ("staged_entries", C.c_uint64), ("name", C.c_char * 64),
                ("gcn_arch", C.c_char * 32)]


class EntryInfo(C.Structure):
    _fields_ = [("logical_type", C.c_int32), ("physical_type", C.c_int32), ("len", C.c_uint32),
                ("nullable", C.c_int32), ("all_null", C.c_int32), ("bit_width", C.c_int32), ("dict_len", C.c_uint32),
                ("has_fingerprints", C.c_int32), ("device_bytes", C.c_uint64), ("algorithmic_pred_bytes", C.c_uint64),
                ("squeezed_date_field", C.c_int32), ("clamped_from_bit_width", C.c_int32),
                ("quantized_from_bit_width", C.c_int32), ("reserved0", C.c_int32), ("quantized_bucket_width", C.c_uint64)]


class ScanInfo(C.Structure):
    _fields_ = [("entries", C.c_uint64), ("rows", C.c_uint64), ("mask_words", C.c_uint64), ("entry_bytes", C.c_uint64),
                ("index_bytes", C.c_uint64), ("unigram_index_bytes", C.c_uint64), ("ctx_index_bytes", C.c_uint64),
                ("ctx_slab_bytes", C.c_uint64), ("index_build_ms", C.c_double), ("like_plans", C.c_uint32), ("is_byte_view", C.c_int32),
                ("max_bit_width", C.c_int32), ("index_build_pending", C.c_int32), ("last_like_kernel", C.c_int32), ("reserved", C.c_int32)]


LIKE_KERNEL_NAMES = {0: "none", 1: "k_str_pred", 2: "k_like_lean", 3: "k_like_flat", 4: "k_like_scanall", 5: "k_like_scanall<unigram>"}


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p),
                        ("flags", C.c_int64), ("n_children", C.c_int64), ("children", C.c_void_p),
                        ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]
ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64),
                       ("n_buffers", C.c_int64), ("n_children", C.c_int64), ("buffers", C.c_void_p),
                       ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p),
                       ("private_data", C.c_void_p)]

# every symbol include/liquid_cache_amd.h declares (tests check that the built library exports all of them)
EXPORTED_SYMBOLS = [
    "lc_ctx_create", "lc_ctx_destroy", "lc_ctx_set_option", "lc_entry_index_to_bytes", "lc_stage_indexed", "lc_scan_explain", "lc_scan_group_partials", "lc_insert_arrow_batch", "lc_insert_arrow_batch_device", "lc_comm_unique_id", "lc_comm_init",
    "lc_comm_destroy", "lc_comm_rank", "lc_comm_world", "lc_comm_allreduce_count", "lc_comm_allgather_mask", "lc_last_error", "lc_device_info_get", "lc_version", "lc_symtab_set",
    "lc_stage", "lc_evict", "lc_entry_info_get", "lc_transcode_arrow", "lc_insert_arrow", "lc_free", "lc_symtab_get",
    "lc_eval_predicate", "lc_eval_predicate_batch", "lc_get_with_selection", "lc_get_date_part_with_selection", "lc_scan_date_part", "lc_scan_gather_bytes_plan", "lc_scan_gather_bytes", "lc_scan_gather_bytes_async", "lc_mask_and_then", "lc_scan_create",
    "lc_scan_destroy", "lc_scan_mask_words", "lc_scan_rows", "lc_scan_entries", "lc_scan_algorithmic_bytes",
    "lc_scan_traffic_model", "lc_scan_eval_and", "lc_scan_eval_count", "lc_scan_eval_or", "lc_eval_predicate_or", "lc_insert_arrow_device", "lc_entry_to_liquid_bytes", "lc_squeeze_date", "lc_squeeze_clamp", "lc_squeeze_quantize", "lc_scan_aggregate", "lc_scan_sum_product", "lc_scan_eval_filter",
    "lc_scan_segment_offsets", "lc_scan_eval", "lc_scan_gather_fixed", "lc_device_alloc", "lc_device_free",
    "lc_device_memset", "lc_device_to_host", "lc_host_to_device", "lc_stream_synchronize",
    "lc_stream_create", "lc_stream_destroy",
    "lc_scan_eval_hits", "lc_scan_mask_to_hits", "lc_scan_gather_fixed_hits", "lc_scan_gather_bytes_hits", "lc_scan_info_get",
    "lc_scan_filter_hits", "lc_scan_index_wait", "lc_scan_eval_count_groups", "lc_eval_predicate_row_groups", "lc_hits_compact",
]
# include/liquid_cache_amd_bench.h: bench / test aids, built into their own library (never part of the product .so)
BENCH_SYMBOLS = ["lc_synth_url_batch", "lc_synth_int64_batch", "lc_synth_phrase_batch", "lc_synth_title_batch",
                 "lc_calibrate_read", "lc_probe_stream_read", "lc_debug_row_lists", "lc_bench_eval_timed", "lc_bench_gather_bytes_hits_timed",
                 "lc_bench_rowgroup_run", "lc_bench_entry_calls", "lc_bench_rowgroup_many"]


class RowGroupStats(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("first_pass_s", C.c_double), ("total_s", C.c_double), ("call_us_mean", C.c_double),
                ("calls", C.c_uint64), ("hits", C.c_uint64), ("units", C.c_uint64), ("passes", C.c_uint32),
                ("threads", C.c_uint32)]

_lib = None
_bench = None
BENCH_LIB_PATH = os.path.join(_HERE, "libliquid_cache_amd_bench.so")


def load_bench():
    """The bench / test aid library (synthetic ClickBench-shaped column generators, PMC calibration kernels)."""
    global _bench
    if _bench is not None:
        return _bench
    load()  # the aid library links the product library
    if not os.path.exists(BENCH_LIB_PATH):
        raise ImportError(f"{BENCH_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    B = C.CDLL(BENCH_LIB_PATH)
    vp, u64, i32, sz = C.c_void_p, C.c_uint64, C.c_int32, C.c_size_t
    for name in ("lc_synth_url_batch", "lc_synth_title_batch", "lc_synth_phrase_batch"):
        getattr(B, name).restype = sz
        getattr(B, name).argtypes = [u64, u64, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, sz]
    B.lc_synth_int64_batch.restype = None
    B.lc_synth_int64_batch.argtypes = [u64, u64, C.c_uint32, i32, C.c_int64, vp]
    B.lc_calibrate_read.restype = i32
    B.lc_calibrate_read.argtypes = [vp, u64, i32, i32]
    B.lc_probe_stream_read.restype = i32
    B.lc_probe_stream_read.argtypes = [vp, u64, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    B.lc_debug_row_lists.restype = sz
    B.lc_debug_row_lists.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, sz]
    B.lc_bench_gather_bytes_hits_timed.restype = i32
    B.lc_bench_gather_bytes_hits_timed.argtypes = [vp, vp, vp, vp, u64, vp, vp, u64, vp, C.c_uint32, vp, i32, C.POINTER(C.c_float)]
    B.lc_bench_eval_timed.restype = i32
    B.lc_bench_eval_timed.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, u64, C.POINTER(C.c_float)]
    B.lc_bench_rowgroup_run.restype = i32
    B.lc_bench_rowgroup_run.argtypes = [vp, u64, C.POINTER(u64), C.POINTER(u64), vp, i32, i32, i32, i32, C.POINTER(RowGroupStats)]
    B.lc_bench_entry_calls.restype = i32
    B.lc_bench_entry_calls.argtypes = [vp, u64, C.POINTER(u64), vp, i32, i32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(u64)]
    _bench = B
    return B


def load():
    """Load the in-tree shared library; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). liquid_cache_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, i32, sz = C.c_void_p, C.c_uint64, C.c_int32, C.c_size_t
    P = C.POINTER
    L.lc_version.restype = C.c_char_p
    L.lc_last_error.restype = C.c_char_p; L.lc_last_error.argtypes = [vp]
    L.lc_ctx_create.restype = i32; L.lc_ctx_create.argtypes = [P(C.c_int32), i32, u64, P(vp)]
    L.lc_ctx_destroy.restype = None; L.lc_ctx_destroy.argtypes = [vp]
    L.lc_ctx_set_option.restype = i32; L.lc_ctx_set_option.argtypes = [vp, i32, C.c_int64]
    L.lc_device_info_get.restype = i32; L.lc_device_info_get.argtypes = [vp, P(DeviceInfo)]
    L.lc_symtab_set.restype = i32; L.lc_symtab_set.argtypes = [vp, u64, vp, sz]
    L.lc_symtab_get.restype = i32; L.lc_symtab_get.argtypes = [vp, u64, P(vp), P(sz)]
    L.lc_stage.restype = i32; L.lc_stage.argtypes = [vp, u64, P(u64), P(vp), P(sz), P(u64)]
    L.lc_stage_indexed.restype = i32; L.lc_stage_indexed.argtypes = [vp, u64, P(u64), P(vp), P(sz), P(u64), P(vp), P(sz)]
    L.lc_entry_index_to_bytes.restype = i32; L.lc_entry_index_to_bytes.argtypes = [vp, u64, P(vp), P(sz)]
    L.lc_evict.restype = i32; L.lc_evict.argtypes = [vp, u64, P(u64)]
    L.lc_entry_info_get.restype = i32; L.lc_entry_info_get.argtypes = [vp, u64, P(EntryInfo)]
    L.lc_transcode_arrow.restype = i32; L.lc_transcode_arrow.argtypes = [vp, vp, vp, i32, u64, P(vp), P(sz)]
    L.lc_insert_arrow.restype = i32; L.lc_insert_arrow.argtypes = [vp, u64, vp, vp, i32, u64]
    L.lc_insert_arrow_batch.restype = i32; L.lc_insert_arrow_batch.argtypes = [vp, u64, P(u64), P(vp), P(vp), P(i32), P(u64)]
    L.lc_insert_arrow_device.restype = i32; L.lc_insert_arrow_device.argtypes = [vp, u64, P(u64), P(vp), P(vp)]
    L.lc_insert_arrow_batch_device.restype = i32; L.lc_insert_arrow_batch_device.argtypes = [vp, u64, P(u64), P(vp), P(vp), P(i32), P(u64)]
    L.lc_entry_to_liquid_bytes.restype = i32; L.lc_entry_to_liquid_bytes.argtypes = [vp, u64, P(vp), P(sz)]
    L.lc_squeeze_clamp.restype = i32; L.lc_squeeze_clamp.argtypes = [vp, u64, P(u64), P(u64)]
    L.lc_scan_eval_filter.restype = i32; L.lc_scan_eval_filter.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, P(vp), vp]
    L.lc_scan_sum_product.restype = i32; L.lc_scan_sum_product.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lc_scan_aggregate.restype = i32; L.lc_scan_aggregate.argtypes = [vp, vp, vp, vp, vp]
    L.lc_squeeze_quantize.restype = i32; L.lc_squeeze_quantize.argtypes = [vp, u64, P(u64), P(u64)]
    L.lc_squeeze_date.restype = i32; L.lc_squeeze_date.argtypes = [vp, u64, P(u64), i32]
    L.lc_free.restype = None; L.lc_free.argtypes = [vp]
    L.lc_eval_predicate.restype = i32
    L.lc_eval_predicate.argtypes = [vp, u64, P(Predicate), vp, vp, vp, P(C.c_uint32), P(C.c_int32)]
    L.lc_eval_predicate_batch.restype = i32
    L.lc_eval_predicate_batch.argtypes = [vp, u64, P(u64), P(Predicate), P(vp), P(vp), P(vp), P(C.c_uint32),
                                          P(C.c_int32), P(C.c_int32)]
    L.lc_get_with_selection.restype = i32; L.lc_get_with_selection.argtypes = [vp, u64, vp, vp, vp]
    L.lc_get_date_part_with_selection.restype = i32
    L.lc_get_date_part_with_selection.argtypes = [vp, u64, vp, i32, vp, vp]
    L.lc_scan_date_part.restype = i32; L.lc_scan_date_part.argtypes = [vp, vp, vp, u64, i32, vp]
    L.lc_scan_gather_bytes_plan.restype = i32
    L.lc_scan_gather_bytes_plan.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, P(u64), P(u64), vp]
    L.lc_scan_gather_bytes.restype = i32; L.lc_scan_gather_bytes.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    L.lc_scan_gather_bytes_async.restype = i32
    L.lc_scan_gather_bytes_async.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, vp, u64, vp]
    L.lc_mask_and_then.restype = i32; L.lc_mask_and_then.argtypes = [vp, vp, u64, vp, u64, vp]
    L.lc_scan_create.restype = i32; L.lc_scan_create.argtypes = [vp, u64, P(u64), P(vp)]
    L.lc_scan_destroy.restype = None; L.lc_scan_destroy.argtypes = [vp]
    for name in ("lc_scan_mask_words", "lc_scan_rows", "lc_scan_entries"):
        getattr(L, name).restype = u64; getattr(L, name).argtypes = [vp]
    L.lc_scan_algorithmic_bytes.restype = u64; L.lc_scan_algorithmic_bytes.argtypes = [vp, P(Predicate), i32]
    L.lc_scan_traffic_model.restype = i32; L.lc_scan_traffic_model.argtypes = [vp, P(Predicate), i32, P(u64), P(u64)]
    L.lc_comm_unique_id.restype = i32; L.lc_comm_unique_id.argtypes = [vp, vp]
    L.lc_comm_init.restype = i32; L.lc_comm_init.argtypes = [vp, i32, i32, vp, P(vp)]
    L.lc_comm_destroy.restype = None; L.lc_comm_destroy.argtypes = [vp]
    L.lc_comm_rank.restype = i32; L.lc_comm_rank.argtypes = [vp]
    L.lc_comm_world.restype = i32; L.lc_comm_world.argtypes = [vp]
    L.lc_comm_allreduce_count.restype = i32; L.lc_comm_allreduce_count.argtypes = [vp, vp, vp]
    L.lc_comm_allgather_mask.restype = i32; L.lc_comm_allgather_mask.argtypes = [vp, vp, u64, vp, P(u64), vp]
    L.lc_scan_group_partials.restype = i32; L.lc_scan_group_partials.argtypes = [vp, vp, vp, i32, vp, vp, u64, vp, vp]
    L.lc_scan_explain.restype = i32; L.lc_scan_explain.argtypes = [vp, P(Predicate), C.c_char_p, sz]
    L.lc_scan_segment_offsets.restype = P(u64); L.lc_scan_segment_offsets.argtypes = [vp]
    L.lc_scan_eval.restype = i32; L.lc_scan_eval.argtypes = [vp, vp, P(Predicate), vp, vp, vp, vp]
    L.lc_scan_eval_and.restype = i32; L.lc_scan_eval_and.argtypes = [vp, vp, P(Predicate), C.c_uint32, vp, vp, vp, vp]
    L.lc_scan_eval_or.restype = i32; L.lc_scan_eval_or.argtypes = [vp, C.c_uint32, P(vp), P(Predicate), vp, vp, vp, vp, vp]
    L.lc_eval_predicate_or.restype = i32
    L.lc_eval_predicate_or.argtypes = [vp, C.c_uint32, P(u64), P(Predicate), vp, vp, vp, P(C.c_uint32), P(C.c_int32)]
    L.lc_scan_eval_count.restype = i32
    L.lc_scan_eval_count.argtypes = [vp, vp, P(Predicate), C.c_uint32, vp, vp, vp, vp, vp]
    L.lc_scan_gather_fixed.restype = i32; L.lc_scan_gather_fixed.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    L.lc_scan_info_get.restype = i32; L.lc_scan_info_get.argtypes = [vp, P(ScanInfo)]
    L.lc_scan_index_wait.restype = i32; L.lc_scan_index_wait.argtypes = [vp]
    L.lc_scan_eval_count_groups.restype = i32
    L.lc_scan_eval_count_groups.argtypes = [vp, vp, P(Predicate), C.c_uint32, vp, C.c_uint32, P(C.c_uint32), vp, vp, vp, vp, vp]
    L.lc_eval_predicate_row_groups.restype = i32
    L.lc_eval_predicate_row_groups.argtypes = [vp, u64, P(u64), C.c_uint32, P(C.c_uint32), P(Predicate), C.c_uint32, P(u64), P(u64), u64,
                                               P(u64)]
    L.lc_scan_eval_hits.restype = i32
    L.lc_scan_eval_hits.argtypes = [vp, vp, P(Predicate), C.c_uint32, vp, vp, u64, vp, vp, vp, vp, C.c_uint32, vp]
    L.lc_scan_filter_hits.restype = i32
    L.lc_scan_filter_hits.argtypes = [vp, vp, P(Predicate), vp, vp, u64, vp, u64, vp, C.c_uint32, vp]
    L.lc_scan_mask_to_hits.restype = i32; L.lc_scan_mask_to_hits.argtypes = [vp, vp, vp, vp, u64, vp, vp, C.c_uint32, vp]
    L.lc_scan_gather_fixed_hits.restype = i32; L.lc_scan_gather_fixed_hits.argtypes = [vp, vp, vp, vp, u64, vp, vp, C.c_uint32, vp]
    L.lc_hits_compact.restype = i32; L.lc_hits_compact.argtypes = [vp, vp, vp, u64, vp, u64, vp, vp]
    L.lc_scan_gather_bytes_hits.restype = i32
    L.lc_scan_gather_bytes_hits.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, u64, vp, C.c_uint32, vp]
    L.lc_device_alloc.restype = i32; L.lc_device_alloc.argtypes = [vp, u64, P(vp)]
    L.lc_device_free.restype = i32; L.lc_device_free.argtypes = [vp, vp]
    L.lc_device_memset.restype = i32; L.lc_device_memset.argtypes = [vp, vp, i32, u64, vp]
    L.lc_device_to_host.restype = i32; L.lc_device_to_host.argtypes = [vp, vp, vp, u64, vp]
    L.lc_host_to_device.restype = i32; L.lc_host_to_device.argtypes = [vp, vp, vp, u64, vp]
    L.lc_stream_synchronize.restype = i32; L.lc_stream_synchronize.argtypes = [vp, vp]
    L.lc_stream_create.restype = i32; L.lc_stream_create.argtypes = [vp, P(vp)]
    L.lc_stream_destroy.restype = i32; L.lc_stream_destroy.argtypes = [vp, vp]
    _lib = L
    return L


def check(status: int, ctx=None):
    if status < 0 or status in (LC_UNSUPPORTED, LC_NEEDS_BACKING):
        msg = load().lc_last_error(ctx).decode(errors="replace")
        raise LiquidCacheError(status, msg)
    return status
