"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/lo_common.h.  Importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from the product package (liquid_cache_amd/).

Every entry point works on the reference's own serialized LiquidArray bytes ("Liquid IPC",
/root/reference/src/core/src/liquid_array/ipc.rs:158-236) plus Arrow-convention buffers
(LSB-first bitmaps, i32 offsets) held in numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# operators (lo_common.h enum lo_op)
EQ, NE, LT, LE, GT, GE, LIKE, NOT_LIKE = range(8)
OP_NAMES = {"eq": EQ, "ne": NE, "lt": LT, "le": LE, "gt": GT, "ge": GE, "like": LIKE, "not_like": NOT_LIKE,
            "==": EQ, "!=": NE, "<": LT, "<=": LE, ">": GT, ">=": GE}
# literal tags
LIT_I64, LIT_U64, LIT_F32, LIT_F64, LIT_BYTES, LIT_I128, LIT_BOOL = range(7)
# physical ids (ipc.rs:28-45)
PHYS = {"int8": 0, "int16": 1, "int32": 2, "int64": 3, "uint8": 4, "uint16": 5, "uint32": 6, "uint64": 7,
        "float32": 8, "float64": 9, "date32": 10, "date64": 11, "timestamp[s]": 12, "timestamp[ms]": 13,
        "timestamp[us]": 14, "timestamp[ns]": 15}
PHYS_NP = {0: np.int8, 1: np.int16, 2: np.int32, 3: np.int64, 4: np.uint8, 5: np.uint16, 6: np.uint32,
           7: np.uint64, 8: np.float32, 9: np.float64, 10: np.int32, 11: np.int64, 12: np.int64, 13: np.int64,
           14: np.int64, 15: np.int64}
LOGICAL_INTEGER, LOGICAL_FLOAT, LOGICAL_BYTE_VIEW, LOGICAL_DECIMAL = 1, 2, 4, 6
# ArrowByteType (byte_view_array/mod.rs:113-122)
BT_UTF8, BT_UTF8VIEW, BT_DICT16_BINARY, BT_DICT16_UTF8, BT_BINARY, BT_BINARYVIEW = range(6)
DATE_YEAR, DATE_MONTH, DATE_DAY, DATE_DOW = range(4)


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc if missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "liboracle.so"], check=True)
    return _LIB_PATH


class SymTab(C.Structure):
    _fields_ = [("n", C.c_int32), ("len", C.c_uint8 * 256), ("sym", C.c_uint64 * 256)]


class ArrayInfo(C.Structure):
    _fields_ = [("logical", C.c_int32), ("phys", C.c_int32), ("len", C.c_uint32), ("nullable", C.c_int32),
                ("all_null", C.c_int32), ("bit_width", C.c_int32), ("value_width", C.c_int32),
                ("lane_bits", C.c_int32), ("reference", C.c_uint64), ("bitpacked_off", C.c_uint64),
                ("decimal_is256", C.c_int32), ("decimal_precision", C.c_int32), ("decimal_scale", C.c_int32),
                ("alp_e", C.c_int32), ("alp_f", C.c_int32), ("patch_len", C.c_uint64),
                ("patch_indices_off", C.c_uint64), ("patch_values_off", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, sz, i32, i64, u8p = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_void_p
        L.lo_fl_index.restype = sz; L.lo_fl_index.argtypes = [i32, sz, sz]
        L.lo_fl_pack.argtypes = [i32, i32, vp, vp]
        L.lo_fl_unpack.argtypes = [i32, i32, vp, vp]
        L.lo_bitpack_size.restype = sz; L.lo_bitpack_size.argtypes = [i32, i32, sz]
        L.lo_bitpack.restype = sz; L.lo_bitpack.argtypes = [i32, i32, vp, sz, vp]
        L.lo_bitunpack.argtypes = [i32, i32, vp, sz, vp]
        L.lo_get_bit_width.restype = i32; L.lo_get_bit_width.argtypes = [C.c_uint64]
        L.lo_array_info_get.restype = i32; L.lo_array_info_get.argtypes = [vp, sz, C.POINTER(ArrayInfo)]
        L.lo_ipc_header_read.restype = i32; L.lo_ipc_header_read.argtypes = [vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for name in ("lo_prim_encode_bound", "lo_float_encode_bound"):
            getattr(L, name).restype = sz; getattr(L, name).argtypes = [i32, sz]
        L.lo_decimal_encode_bound.restype = sz; L.lo_decimal_encode_bound.argtypes = [sz]
        L.lo_prim_encode.restype = i64; L.lo_prim_encode.argtypes = [i32, vp, vp, sz, vp, sz]
        L.lo_float_encode.restype = i64; L.lo_float_encode.argtypes = [i32, vp, vp, sz, vp, sz]
        L.lo_decimal_encode.restype = i64; L.lo_decimal_encode.argtypes = [i32, i32, i32, vp, vp, sz, vp, sz]
        L.lo_fixed_to_arrow.restype = i32; L.lo_fixed_to_arrow.argtypes = [vp, sz, vp, vp]
        L.lo_fixed_filter.restype = i64; L.lo_fixed_filter.argtypes = [vp, sz, vp, vp, vp, C.POINTER(C.c_int)]
        L.lo_fixed_eval_predicate.restype = i64
        L.lo_fixed_eval_predicate.argtypes = [vp, sz, i32, i32, vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.lo_date_component.restype = C.c_int32; L.lo_date_component.argtypes = [i32, C.c_int32]
        L.lo_ymd_to_epoch_days.restype = C.c_int32; L.lo_ymd_to_epoch_days.argtypes = [C.c_int32, C.c_uint32, C.c_uint32]
        L.lo_timestamp_to_days.restype = C.c_int32; L.lo_timestamp_to_days.argtypes = [C.c_int64, i32]
        L.lo_date_lossy_days.restype = C.c_int32; L.lo_date_lossy_days.argtypes = [i32, C.c_int32]
        L.lo_and_then.restype = sz; L.lo_and_then.argtypes = [vp, sz, vp, sz, vp]
        L.lo_prep_null_mask.argtypes = [vp, vp, sz, vp]
        L.lo_like_match.restype = i32; L.lo_like_match.argtypes = [vp, sz, vp, sz]
        L.lo_symtab_load.restype = i32; L.lo_symtab_load.argtypes = [vp, sz, C.POINTER(SymTab)]
        L.lo_symtab_save.restype = sz; L.lo_symtab_save.argtypes = [C.POINTER(SymTab), vp]
        L.lo_fsst_train.argtypes = [vp, vp, sz, C.POINTER(SymTab)]
        L.lo_fsst_compress.restype = sz; L.lo_fsst_compress.argtypes = [C.POINTER(SymTab), vp, sz, vp]
        L.lo_fsst_decompress.restype = sz; L.lo_fsst_decompress.argtypes = [C.POINTER(SymTab), vp, sz, vp, sz]
        L.lo_bv_encode_bound.restype = sz; L.lo_bv_encode_bound.argtypes = [sz, sz]
        L.lo_bv_encode.restype = i64; L.lo_bv_encode.argtypes = [i32, vp, vp, vp, sz, C.POINTER(SymTab), i32, vp, sz]
        L.lo_bv_encode_dict.restype = i64
        L.lo_bv_encode_dict.argtypes = [i32, vp, vp, sz, vp, vp, sz, C.POINTER(SymTab), i32, vp, sz]
        L.lo_bv_eval_predicate.restype = i64
        L.lo_bv_eval_predicate.argtypes = [vp, sz, C.POINTER(SymTab), i32, i32, vp, sz, vp, vp, vp, C.POINTER(C.c_int)]
        L.lo_bv_dict_results.restype = i32
        L.lo_bv_dict_results.argtypes = [vp, sz, C.POINTER(SymTab), i32, vp, sz, vp, C.POINTER(C.c_uint32)]
        L.lo_bv_filter_to_arrow.restype = i64
        L.lo_bv_filter_to_arrow.argtypes = [vp, sz, C.POINTER(SymTab), vp, vp, vp, sz, vp, C.POINTER(sz), C.POINTER(C.c_int)]
        L.lo_fingerprint.restype = C.c_uint32; L.lo_fingerprint.argtypes = [vp, sz]
        _lib = L
    return _lib


# ---------------------------------------------------------------- helpers
def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b.view(np.uint8).reshape(-1))
    return np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(0, np.uint8)


def pack_bits(bools) -> np.ndarray:
    """bool array -> LSB-first Arrow bitmap (uint8)."""
    b = np.asarray(bools, dtype=bool)
    return np.packbits(b, bitorder="little") if b.size else np.zeros(0, np.uint8)


def unpack_bits(bitmap: np.ndarray, n: int) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=bool)
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def _check(rc, what):
    if rc < 0:
        raise RuntimeError(f"oracle {what} failed: {rc}")
    return rc


@dataclass
class BoolResult:
    """A BooleanArray: `values` and `validity` as bool arrays of length popcount(selection)."""
    values: np.ndarray
    validity: Optional[np.ndarray]

    def filter_mask(self) -> np.ndarray:
        """prep_null_mask_filter: nulls -> False."""
        return self.values if self.validity is None else (self.values & self.validity)


# ---------------------------------------------------------------- fastlanes
def fl_index(tbits: int, row: int, lane: int) -> int:
    return lib().lo_fl_index(tbits, row, lane)


def bitpack(values: np.ndarray, W: int) -> np.ndarray:
    v = np.ascontiguousarray(values)
    tb = v.dtype.itemsize * 8
    out = np.zeros(lib().lo_bitpack_size(tb, W, v.size), np.uint8)
    lib().lo_bitpack(tb, W, _ptr(v), v.size, _ptr(out))
    return out


def bitunpack(packed: np.ndarray, W: int, n: int, dtype) -> np.ndarray:
    out = np.zeros(n, dtype=dtype)
    lib().lo_bitunpack(out.dtype.itemsize * 8, W, _ptr(np.ascontiguousarray(packed)), n, _ptr(out))
    return out


def get_bit_width(v: int) -> int:
    return lib().lo_get_bit_width(int(v))


# ---------------------------------------------------------------- encode
def encode_primitive(phys: int, values: np.ndarray, validity: Optional[np.ndarray] = None) -> bytes:
    """validity: bool array or None."""
    v = np.ascontiguousarray(values, dtype=PHYS_NP[phys])
    vb = pack_bits(validity) if validity is not None else None
    if vb is not None and vb.size == 0:
        vb = np.zeros(1, np.uint8)
    if phys in (8, 9):
        cap = lib().lo_float_encode_bound(phys, v.size)
        out = np.zeros(cap, np.uint8)
        n = _check(lib().lo_float_encode(phys, _ptr(v), _ptr(vb), v.size, _ptr(out), cap), "float_encode")
    else:
        cap = lib().lo_prim_encode_bound(phys, v.size)
        out = np.zeros(cap, np.uint8)
        n = _check(lib().lo_prim_encode(phys, _ptr(v), _ptr(vb), v.size, _ptr(out), cap), "prim_encode")
    return out[:n].tobytes()


def _i128_array(ints) -> np.ndarray:
    out = np.zeros((len(ints), 2), dtype=np.uint64)
    for i, x in enumerate(ints):
        x = int(x) & ((1 << 128) - 1)
        out[i, 0] = x & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = x >> 64
    return out


def encode_decimal(unscaled, validity=None, precision=15, scale=2, is256=False) -> bytes:
    v = _i128_array([0 if x is None else x for x in unscaled])
    if validity is None and any(x is None for x in unscaled):
        validity = [x is not None for x in unscaled]
    vb = pack_bits(validity) if validity is not None else None
    if vb is not None and vb.size == 0:
        vb = np.zeros(1, np.uint8)
    n_rows = len(unscaled)
    cap = lib().lo_decimal_encode_bound(n_rows)
    out = np.zeros(cap, np.uint8)
    n = _check(lib().lo_decimal_encode(int(is256), precision, scale, _ptr(v), _ptr(vb), n_rows, _ptr(out), cap),
               "decimal_encode")
    return out[:n].tobytes()


def strings_to_arrow(strings) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """list of bytes/str/None -> (offsets i32[n+1], data u8, validity bool or None)."""
    offs = np.zeros(len(strings) + 1, np.int32)
    chunks = []
    valid = np.ones(len(strings), bool)
    total = 0
    for i, s in enumerate(strings):
        if s is None:
            valid[i] = False
        else:
            b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
            chunks.append(b)
            total += len(b)
        offs[i + 1] = total
    data = np.frombuffer(b"".join(chunks), np.uint8).copy() if total else np.zeros(0, np.uint8)
    return offs, data, (None if valid.all() else valid)


def fsst_train(offsets: np.ndarray, data: np.ndarray) -> SymTab:
    st = SymTab()
    d = data if data.size else np.zeros(1, np.uint8)
    lib().lo_fsst_train(_ptr(d), _ptr(np.ascontiguousarray(offsets, np.int32)), len(offsets) - 1, C.byref(st))
    return st


def symtab_bytes(st: SymTab) -> bytes:
    out = np.zeros(1 + 9 * 256, np.uint8)
    n = lib().lo_symtab_save(C.byref(st), _ptr(out))
    return out[:n].tobytes()


def symtab_load(b: bytes) -> SymTab:
    st = SymTab()
    a = _u8(b)
    _check(lib().lo_symtab_load(_ptr(a), a.size, C.byref(st)), "symtab_load")
    return st


def fsst_compress(st: SymTab, s: bytes) -> bytes:
    a = _u8(s) if len(s) else np.zeros(1, np.uint8)
    out = np.zeros(2 * len(s) + 8, np.uint8)
    n = lib().lo_fsst_compress(C.byref(st), _ptr(a), len(s), _ptr(out))
    return out[:n].tobytes()


def fsst_decompress(st: SymTab, c: bytes) -> bytes:
    a = _u8(c) if len(c) else np.zeros(1, np.uint8)
    out = np.zeros(8 * len(c) + 8, np.uint8)
    n = lib().lo_fsst_decompress(C.byref(st), _ptr(a), len(c), _ptr(out), out.size)
    return out[:n].tobytes()


def encode_byte_view(strings, st: Optional[SymTab] = None, fingerprints: bool = False,
                     arrow_type: int = BT_UTF8) -> Tuple[bytes, SymTab]:
    """train_from_arrow / from_string_array: returns (liquid bytes, symbol table)."""
    offs, data, valid = strings_to_arrow(strings)
    if st is None:
        # train on non-null values (conversions.rs:212-227)
        nn = [s for s in strings if s is not None]
        o2, d2, _ = strings_to_arrow(nn)
        st = fsst_train(o2, d2)
    vb = pack_bits(valid) if valid is not None else None
    cap = lib().lo_bv_encode_bound(len(strings), data.size)
    out = np.zeros(cap, np.uint8)
    d = data if data.size else np.zeros(1, np.uint8)
    n = _check(lib().lo_bv_encode(arrow_type, _ptr(offs), _ptr(d), _ptr(vb), len(strings), C.byref(st),
                                  int(fingerprints), _ptr(out), cap), "bv_encode")
    return out[:n].tobytes(), st


def encode_byte_view_dict(keys, key_validity, dict_values, st: Optional[SymTab] = None, fingerprints=False,
                          arrow_type: int = BT_DICT16_UTF8) -> Tuple[bytes, SymTab]:
    """from_unique_dict_array: keys (u16, garbage allowed under nulls) + unique dictionary values."""
    doffs, ddata, _ = strings_to_arrow(dict_values)
    if st is None:
        st = fsst_train(doffs, ddata)
    k = np.ascontiguousarray(keys, np.uint16)
    vb = pack_bits(key_validity) if key_validity is not None else None
    cap = lib().lo_bv_encode_bound(max(len(k), len(dict_values)), ddata.size)
    out = np.zeros(cap, np.uint8)
    d = ddata if ddata.size else np.zeros(1, np.uint8)
    n = _check(lib().lo_bv_encode_dict(arrow_type, _ptr(k), _ptr(vb), len(k), _ptr(doffs), _ptr(d),
                                       len(dict_values), C.byref(st), int(fingerprints), _ptr(out), cap),
               "bv_encode_dict")
    return out[:n].tobytes(), st


# ---------------------------------------------------------------- inspect / decode
def array_info(liquid: bytes) -> ArrayInfo:
    a = _u8(liquid)
    info = ArrayInfo()
    _check(lib().lo_array_info_get(_ptr(a), a.size, C.byref(info)), "array_info")
    return info


def logical_type(liquid: bytes) -> int:
    a = _u8(liquid)
    lg, ph = C.c_int(), C.c_int()
    _check(lib().lo_ipc_header_read(_ptr(a), a.size, C.byref(lg), C.byref(ph)), "ipc_header")
    return lg.value


def _fixed_dtype(info: ArrayInfo):
    if info.logical == LOGICAL_DECIMAL:
        return None
    return PHYS_NP[info.phys]


def _i128_to_ints(raw: np.ndarray):
    r = raw.reshape(-1, 2)
    out = []
    for lo_, hi in r:
        x = (int(hi) << 64) | int(lo_)
        if x >= 1 << 127:
            x -= 1 << 128
        out.append(x)
    return out


def to_arrow_fixed(liquid: bytes):
    """Returns (values ndarray | list[int] for decimals, validity bool ndarray | None)."""
    a = _u8(liquid)
    info = array_info(liquid)
    n = info.len
    raw = np.zeros(n * info.value_width + 16, np.uint8)
    vb = np.zeros((n + 7) // 8 + 1, np.uint8)
    _check(lib().lo_fixed_to_arrow(_ptr(a), a.size, _ptr(raw), _ptr(vb)), "to_arrow")
    valid = unpack_bits(vb, n) if info.nullable else None
    dt = _fixed_dtype(info)
    if dt is None:
        return _i128_to_ints(raw[: n * 16].view(np.uint64)), valid
    return raw[: n * info.value_width].view(dt).copy(), valid


def filter_fixed(liquid: bytes, selection: np.ndarray):
    a = _u8(liquid)
    info = array_info(liquid)
    n = info.len
    sel = pack_bits(selection)
    if sel.size == 0:
        sel = np.zeros(1, np.uint8)
    raw = np.zeros(n * info.value_width + 16, np.uint8)
    vb = np.zeros((n + 7) // 8 + 1, np.uint8)
    nl = C.c_int()
    k = _check(lib().lo_fixed_filter(_ptr(a), a.size, _ptr(sel), _ptr(raw), _ptr(vb), C.byref(nl)), "filter")
    valid = unpack_bits(vb, k) if nl.value else None
    dt = _fixed_dtype(info)
    if dt is None:
        return _i128_to_ints(raw[: k * 16].view(np.uint64)), valid
    return raw[: k * info.value_width].view(dt).copy(), valid


def _literal(info_logical: int, phys: int, literal):
    """-> (tag, buffer ndarray)."""
    if isinstance(literal, (bytes, bytearray, str)):
        b = literal.encode() if isinstance(literal, str) else bytes(literal)
        return LIT_BYTES, (np.frombuffer(b, np.uint8).copy() if b else np.zeros(1, np.uint8)), len(b)
    if isinstance(literal, bool):
        return LIT_BOOL, np.array([1 if literal else 0], np.uint8), 1
    if info_logical == LOGICAL_DECIMAL:
        return LIT_I128, _i128_array([literal]).reshape(-1), 16
    if info_logical == LOGICAL_FLOAT:
        if phys == 8:
            return LIT_F32, np.array([literal], np.float32), 4
        return LIT_F64, np.array([literal], np.float64), 8
    if phys in (4, 5, 6, 7):
        return LIT_U64, np.array([int(literal)], np.uint64), 8
    return LIT_I64, np.array([int(literal)], np.int64), 8


def eval_predicate(liquid: bytes, op: int, literal, selection: Optional[np.ndarray] = None,
                   symtab: Optional[SymTab] = None) -> BoolResult:
    """cache.eval_predicate(id, expr).with_selection(sel): BooleanArray of length popcount(sel)."""
    a = _u8(liquid)
    lg = logical_type(liquid)
    sel = None
    if selection is not None:
        sel = pack_bits(selection)
        if sel.size == 0:
            sel = np.zeros(1, np.uint8)
    nl = C.c_int()
    if lg == LOGICAL_BYTE_VIEW:
        tag, buf, ln = _literal(lg, 0, literal)
        # length from the keys section
        n = byte_view_len(liquid)
        ov = np.zeros((n + 7) // 8 + 1, np.uint8)
        ovalid = np.zeros((n + 7) // 8 + 1, np.uint8)
        st = symtab if symtab is not None else SymTab()
        k = _check(lib().lo_bv_eval_predicate(_ptr(a), a.size, C.byref(st), op, tag, _ptr(buf), ln, _ptr(sel),
                                              _ptr(ov), _ptr(ovalid), C.byref(nl)), "bv_eval_predicate")
    else:
        info = array_info(liquid)
        n = info.len
        tag, buf, ln = _literal(lg, info.phys, literal)
        ov = np.zeros((n + 7) // 8 + 1, np.uint8)
        ovalid = np.zeros((n + 7) // 8 + 1, np.uint8)
        k = _check(lib().lo_fixed_eval_predicate(_ptr(a), a.size, op, tag, _ptr(buf), _ptr(sel), _ptr(ov),
                                                 _ptr(ovalid), C.byref(nl)), "eval_predicate")
    return BoolResult(unpack_bits(ov, k), unpack_bits(ovalid, k) if nl.value else None)


def bench_eval_batches(blobs, symtabs, op: int, literal, threads: int = 1) -> int:
    """Native loop over many batches (lo_bench.c): rows whose predicate is true.  `symtabs[i]` is the SymTab of byte-view
    batch i or None for fixed-width batches.  Used by bench.py's cpu_baseline leg only."""
    n = len(blobs)
    arrs = [_u8(b) for b in blobs]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * n)(*[a.size for a in arrs])
    is_str = symtabs is not None and any(s is not None for s in symtabs)
    st_ptrs = (C.c_void_p * n)(*[C.addressof(s) if s is not None else None for s in symtabs]) if is_str else None
    lg = logical_type(blobs[0])
    info = None if lg == LOGICAL_BYTE_VIEW else array_info(blobs[0])
    tag, buf, ln = _literal(lg, 0 if info is None else info.phys, literal)
    fn = lib().lo_bench_eval_batches
    fn.restype = C.c_int64
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    r = fn(n, ptrs, lens, st_ptrs, op, tag, _ptr(buf), ln, threads)
    if r < 0:
        raise RuntimeError("lo_bench_eval_batches failed")
    return int(r)


def bench_eval_batches_masks(blobs, symtabs, op: int, literal, seg_offsets: np.ndarray, threads: int = 1):
    """bench_eval_batches that also returns the hit mask in scan layout (u64 words, batch i at seg_offsets[i]) and the
    per-batch hit counts: (total, mask words, counts).  Used by bench.py's checker leg and the full-size tests."""
    n = len(blobs)
    arrs = [_u8(b) for b in blobs]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * n)(*[a.size for a in arrs])
    is_str = symtabs is not None and any(s is not None for s in symtabs)
    st_ptrs = (C.c_void_p * n)(*[C.addressof(s) if s is not None else None for s in symtabs]) if is_str else None
    lg = logical_type(blobs[0])
    info = None if lg == LOGICAL_BYTE_VIEW else array_info(blobs[0])
    tag, buf, ln = _literal(lg, 0 if info is None else info.phys, literal)
    so = np.ascontiguousarray(seg_offsets, np.uint64)
    assert so.size == n + 1
    mask = np.zeros(max(int(so[-1]), 1), np.uint64)
    counts = np.zeros(max(n, 1), np.uint32)
    fn = lib().lo_bench_eval_batches_masks
    fn.restype = C.c_int64
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p]
    r = fn(n, ptrs, lens, st_ptrs, op, tag, _ptr(buf), ln, threads, so.ctypes.data, mask.ctypes.data, counts.ctypes.data)
    if r < 0:
        raise RuntimeError("lo_bench_eval_batches_masks failed")
    return int(r), mask[: int(so[-1])], counts[:n]


def byte_view_len(liquid: bytes) -> int:
    a = _u8(liquid)
    # header(16) + view header(20) -> fsst at 40; keys section follows (serialization.rs:223-252)
    fsst_size = int(np.frombuffer(a[28:32].tobytes(), np.uint32)[0])
    cur = (40 + fsst_size + 7) & ~7
    return int(np.frombuffer(a[cur:cur + 4].tobytes(), np.uint32)[0])


def dict_results(liquid: bytes, symtab: SymTab, op: int, needle: bytes):
    """Per-dictionary-entry results of compare_with on the unfiltered array -> (bool[D], n_candidates)."""
    a = _u8(liquid)
    nd = _u8(needle) if len(needle) else np.zeros(1, np.uint8)
    out = np.zeros(70000, np.uint8)
    c = C.c_uint32()
    _check(lib().lo_bv_dict_results(_ptr(a), a.size, C.byref(symtab), op, _ptr(nd), len(needle), _ptr(out),
                                    C.byref(c)), "dict_results")
    return out, c.value


def filter_byte_view(liquid: bytes, symtab: SymTab, selection: Optional[np.ndarray] = None):
    """get().with_selection(): -> list of bytes | None."""
    a = _u8(liquid)
    n = byte_view_len(liquid)
    sel = None
    if selection is not None:
        sel = pack_bits(selection)
        if sel.size == 0:
            sel = np.zeros(1, np.uint8)
    dl = C.c_size_t()
    nl = C.c_int()
    k = _check(lib().lo_bv_filter_to_arrow(_ptr(a), a.size, C.byref(symtab), _ptr(sel), None, None, 0, None,
                                           C.byref(dl), C.byref(nl)), "bv_filter size")
    offs = np.zeros(k + 1, np.int32)
    data = np.zeros(dl.value + 8, np.uint8)
    vb = np.zeros((n + 7) // 8 + 1, np.uint8)
    _check(lib().lo_bv_filter_to_arrow(_ptr(a), a.size, C.byref(symtab), _ptr(sel), _ptr(offs), _ptr(data),
                                       data.size, _ptr(vb), C.byref(dl), C.byref(nl)), "bv_filter")
    valid = unpack_bits(vb, k) if nl.value else np.ones(k, bool)
    raw = data.tobytes()
    return [raw[offs[i]:offs[i + 1]] if valid[i] else None for i in range(k)]


# ---------------------------------------------------------------- masks / dates
def and_then(left: np.ndarray, right: np.ndarray) -> np.ndarray:
    """boolean_buffer_and_then on bool arrays."""
    lb, rb = pack_bits(left), pack_bits(right)
    if rb.size == 0:
        rb = np.zeros(1, np.uint8)
    out = np.zeros(max(lb.size, 1), np.uint8)
    n = lib().lo_and_then(_ptr(lb if lb.size else np.zeros(1, np.uint8)), len(left), _ptr(rb), len(right), _ptr(out))
    return unpack_bits(out, n)


def date_component(field: int, days: int) -> int:
    return lib().lo_date_component(field, int(days))


def date_lossy_days(field: int, component: int) -> int:
    return lib().lo_date_lossy_days(field, int(component))


def timestamp_to_days(value: int, unit: int) -> int:
    return lib().lo_timestamp_to_days(int(value), unit)


def like_match(s: bytes, pattern: bytes) -> bool:
    a = _u8(s) if len(s) else np.zeros(1, np.uint8)
    p = _u8(pattern) if len(pattern) else np.zeros(1, np.uint8)
    return bool(lib().lo_like_match(_ptr(a), len(s), _ptr(p), len(pattern)))


def fingerprint(s: bytes) -> int:
    a = _u8(s) if len(s) else np.zeros(1, np.uint8)
    return lib().lo_fingerprint(_ptr(a), len(s))


# ------------------------------------------------------------------------------------------------------------------
# Quantize squeeze (test infrastructure like everything in this directory)
#   squeeze:   LiquidPrimitiveArray::squeeze, IntegerSqueezePolicy::Quantize    primitive_array.rs:455-498
#   predicate: LiquidPrimitiveQuantizedArray::try_eval_predicate_inner          hybrid_primitive_array.rs:487-665
#   selection: try_eval_predicate filters first, then evaluates                 hybrid_primitive_array.rs:700-760
# ------------------------------------------------------------------------------------------------------------------
class NeedsBacking(Exception):
    """SqueezeResult::Err(NeedsBacking): the squeezed data cannot decide, the caller reads the full bytes."""


def quantize_squeeze(values, validity, signed: bool):
    """(buckets, reference, bucket_width, new_bit_width) of the Quantize policy, or None when the array is not
    squeezable (original width < 8, primitive_array.rs:600-611).  `values`: python ints / numpy; `validity`: bools."""
    vals = [int(v) for v in values]
    valid = [True] * len(vals) if validity is None else [bool(b) for b in validity]
    present = [v for v, ok in zip(vals, valid) if ok]
    if not present:
        return None
    reference = min(present)
    offsets = [(v - reference) if ok else 0 for v, ok in zip(vals, valid)]
    orig_bw = get_bit_width(max(offsets))
    if orig_bw < 8:
        return None
    new_bw = max(orig_bw // 2, 1)
    count = 1 << new_bw
    range_size = min(max(offsets) + 1, (1 << 64) - 1)
    width = max(-(-range_size // count), 1)
    buckets = [min(o // width, count - 1) for o in offsets]
    return buckets, reference, width, new_bw


def quantized_eval(buckets, validity, reference: int, width: int, op: int, k: int, selection=None,
                   decimal: bool = False) -> BoolResult:
    """Result over the selected rows, or raises NeedsBacking on the first valid selected row in k's bucket that the
    bucket does not decide.  decimal=True: LiquidDecimalQuantizedArray (decimal_array.rs:416-512), the same rule except
    that its `less_side` also lists Eq (:462-465) — rows of lower buckets answer true to `= k`."""
    n = len(buckets)
    valid = [True] * n if validity is None else [bool(b) for b in validity]
    rows = [i for i in range(n) if selection is None or selection[i]]
    below_const = {EQ: False, NE: True, LT: False, LE: False, GT: True, GE: True}[op]
    out = []
    if k < reference:
        out = [valid[i] and below_const for i in rows]
    else:
        rel = k - reference
        q, r = divmod(rel, width)
        less_side = {EQ: decimal, NE: True, LT: True, LE: True, GT: False, GE: False}[op]
        greater_side = {EQ: False, NE: True, LT: False, LE: False, GT: True, GE: True}[op]
        if op == LT:
            equal = False if r == 0 else None
        elif op == LE:
            equal = True if r + 1 == width else None
        elif op == GT:
            equal = False if r + 1 == width else None
        elif op == GE:
            equal = True if r == 0 else None
        else:
            equal = None
        for i in rows:
            if not valid[i]:
                out.append(False)
                continue
            b = buckets[i]
            if b < q:
                out.append(less_side)
            elif b > q:
                out.append(greater_side)
            elif equal is None:
                raise NeedsBacking()
            else:
                out.append(equal)
    has_nulls = validity is not None
    return BoolResult(np.array(out, dtype=bool), np.array([valid[i] for i in rows], dtype=bool) if has_nulls else None)


# ------------------------------------------------------------------------------------------------------------------
# Float Quantize squeeze (test infrastructure)
#   squeeze:   LiquidFloatArray::squeeze, FloatSqueezePolicy::Quantize            float_array.rs:338-395
#   predicate: LiquidFloatQuantizedArray::try_eval_predicate_inner                float_array.rs:825-953
#   selection: try_eval_predicate filters the buckets first (filter_inner :786-792) but keeps the UNFILTERED patch indices
#              (new_from_filtered :772-784), so with patches a selection has no well-defined result: not restated.
# Restated as written, including that the bucket bounds are computed as (bucket << shift) + reference (:919-922) although
# the buckets were cut at multiples of 2^shift of the ABSOLUTE encoded value (:363-368) — the bounds sit
# reference mod 2^shift too high.
# ------------------------------------------------------------------------------------------------------------------
_F10_32 = [np.float32(x) for x in (1.0, 10.0, 100.0, 1000.0, 10000.0, 100000.0, 1000000.0, 10000000.0, 100000000.0,
                                   1000000000.0, 10000000000.0)]
_IF10_32 = [np.float32(x) for x in (1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001,
                                    0.000000001, 0.0000000001)]


def _wrap(v: int, bits: int) -> int:
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def _alp_decode(i: int, e: int, f: int, bits: int):
    """LiquidFloatType::decode_single (float_array.rs:109-125): (i as F) * F10[f] * IF10[e], no fused multiply-add."""
    if bits == 32:
        return np.float32(np.float32(np.float32(i) * _F10_32[f]) * _IF10_32[e])
    return np.float64(np.float64(i) * _F10_64[f]) * _IF10_64[e]


_F10_64 = [np.float64(float("1e%d" % k)) for k in range(24)]
_IF10_64 = [np.float64(float("1e-%d" % k)) for k in range(24)]


def float_parts(liquid: bytes):
    """(info, packed-domain offsets as python ints, validity bools | None, patch indices, patch values) of a LiquidFloatArray."""
    info = array_info(liquid)
    assert info.logical == LOGICAL_FLOAT
    a = _u8(liquid)
    n = info.len
    sec = int(info.bitpacked_off)
    has_nulls = int(a[sec + 5])
    nulls_len = int(np.frombuffer(a[sec + 6: sec + 10].tobytes(), np.uint32)[0])
    values_len = int(np.frombuffer(a[sec + 10: sec + 14].tobytes(), np.uint32)[0])
    voff = sec + ((16 + nulls_len + 7) // 8) * 8
    udt = np.uint32 if info.phys == 8 else np.uint64
    if info.all_null or info.bit_width == 0:
        offs = [0] * n
    else:
        offs = [int(x) for x in bitunpack(a[voff: voff + values_len], info.bit_width, n, udt)]
    valid = unpack_bits(a[sec + 16: sec + 16 + nulls_len], n).tolist() if has_nulls else None
    pl = int(info.patch_len)
    pidx = np.frombuffer(a[int(info.patch_indices_off): int(info.patch_indices_off) + 8 * pl].tobytes(), np.uint64).tolist()
    fdt = np.float32 if info.phys == 8 else np.float64
    w = 4 if info.phys == 8 else 8
    pval = np.frombuffer(a[int(info.patch_values_off): int(info.patch_values_off) + w * pl].tobytes(), fdt).copy()
    return info, offs, valid, pidx, pval


def float_quantize_squeeze(liquid: bytes):
    """dict(buckets, shift, new_bw, reference, e, f, bits, validity, patch_idx, patch_val, overflow) or None when the
    array is not squeezable (no bit width or < 8 bits, float_array.rs:343-346)."""
    info, offs, valid, pidx, pval = float_parts(liquid)
    if info.all_null or info.bit_width < 8:
        return None
    bits = 32 if info.phys == 8 else 64
    ref = _wrap(int(info.reference), bits)
    new_bw = info.bit_width // 2
    shift = info.bit_width - new_bw
    qmin = ref >> shift
    buckets = [(_wrap(_wrap(ref + o, bits) >> shift, bits) - qmin) & ((1 << bits) - 1) for o in offs]
    return dict(buckets=buckets, shift=shift, new_bw=new_bw, reference=ref, e=info.alp_e, f=info.alp_f, bits=bits,
                validity=valid, patch_idx=[int(i) for i in pidx], patch_val=pval,
                overflow=any(b >= (1 << new_bw) for b in buckets))


def float_quantized_eval(q, op: int, k, selection=None) -> BoolResult:
    """try_eval_predicate (filter_inner, then try_eval_predicate_inner): BoolResult over the selected rows, or raises
    NeedsBacking when a valid, unpatched, selected row's bucket bounds do not decide the comparison.  A selection over an
    array WITH patches is not restated (see the header of this section)."""
    if selection is not None:
        if q["patch_idx"]:
            raise ValueError("selection over a float-quantized array with patches: undefined in the reference")
        keep = [i for i, s_ in enumerate(selection) if s_]
        q = dict(q, buckets=[q["buckets"][i] for i in keep],
                 validity=None if q["validity"] is None else [q["validity"][i] for i in keep])
    bits = q["bits"]
    ft = np.float32 if bits == 32 else np.float64
    k = ft(k)
    n = len(q["buckets"])
    valid = [True] * n if q["validity"] is None else q["validity"]
    patched = set(q["patch_idx"])
    out = [False] * n

    def decide(lo, hi):
        if op == EQ:
            return False if (k < lo or k > hi) else None
        if op == NE:
            return True if (k < lo or k > hi) else None
        if op == LT:
            return False if k <= lo else (True if hi < k else None)
        if op == LE:
            return False if k < lo else (True if hi <= k else None)
        if op == GT:
            return True if k < lo else (False if hi <= k else None)
        return True if k <= lo else (False if hi < k else None)

    with np.errstate(all="ignore"):
        for i, b in enumerate(q["buckets"]):
            if not valid[i] or i in patched:
                continue
            lo_i = _wrap((b << q["shift"]) + q["reference"], bits)
            hi_i = _wrap(((b + 1) << q["shift"]) + q["reference"], bits)
            d = decide(_alp_decode(lo_i, q["e"], q["f"], bits), _alp_decode(hi_i, q["e"], q["f"], bits))
            if d is None:
                raise NeedsBacking()
            out[i] = bool(d)
        for i, pv in zip(q["patch_idx"], q["patch_val"]):
            pv = ft(pv)
            out[i] = bool({EQ: pv == k, NE: pv != k, LT: pv < k, LE: pv <= k, GT: pv > k, GE: pv >= k}[op])
    vals = np.array(out, bool)
    return BoolResult(vals, None if q["validity"] is None else np.array(valid, bool))
