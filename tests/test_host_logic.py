"""CPU tests of the host side: C-ABI surface, Arrow->Liquid transcoder, LiquidExpr validation, sharding.

No compute call is made without a GPU: the host-only context must FAIL LOUDLY for staging / evaluation.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import liquid_cache_amd as lc
from liquid_cache_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    return set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", text)) - {"lc_status"}


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_library_exports_every_declared_symbol(product_lib):
    declared = _declared("liquid_cache_amd.h")
    assert declared == set(N.EXPORTED_SYMBOLS)
    exported = _exported(N.LIB_PATH)
    missing = declared - exported
    assert not missing, f"symbols declared in include/liquid_cache_amd.h but not exported: {sorted(missing)}"
    for name in declared:
        assert hasattr(product_lib, name)
    assert b"gfx950" in product_lib.lc_version()


def test_bench_aids_live_in_their_own_library(product_lib):
    """Data generators and profiling aids (include/liquid_cache_amd_bench.h) are NOT part of the product library."""
    declared = _declared("liquid_cache_amd_bench.h")
    assert declared == set(N.BENCH_SYMBOLS)
    assert not (declared & _exported(N.LIB_PATH)), "bench aids leaked into the product library"
    assert declared <= _exported(N.BENCH_LIB_PATH)
    B = N.load_bench()
    for name in declared:
        assert hasattr(B, name)
    # and nothing in the product library's exports is undeclared
    assert _exported(N.LIB_PATH) <= _declared("liquid_cache_amd.h"), sorted(_exported(N.LIB_PATH) - _declared("liquid_cache_amd.h"))


def test_library_embeds_gfx950_code_object():
    blob = open(N.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_fixed_pred" in blob and b"k_str_pred" in blob


def test_host_only_context_fails_loudly_for_compute(product_lib):
    cache = lc.LiquidCacheBuilder.new().with_host_only().build()
    liquid = cache.transcode(pa.array([1, 2, 3], type=pa.int64()))
    assert liquid is not None
    with pytest.raises(lc.LiquidCacheError) as e:
        cache.stage([1], [liquid])
    assert e.value.status == N.LC_ERR_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(lc.LiquidCacheError):
        cache.scan([1])
    with pytest.raises(lc.LiquidCacheError):
        lc.boolean_buffer_and_then(cache, [True, False, True], [True, False])
    # nothing can be staged, so lookups answer "not cached" (None) exactly like the reference — never a CPU result
    expr = lc.LiquidExpr.try_new(">", 1, pa.int64())
    assert cache.eval_predicate(1, expr).read() is None and cache.get(1).read() is None
    cache.close()


def test_no_gpu_means_no_default_context(product_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lc.LiquidCacheError) as e:
        lc.LiquidCacheBuilder.new().build()
    assert e.value.status == N.LC_ERR_DEVICE


@pytest.fixture()
def host_cache(product_lib):
    c = lc.LiquidCacheBuilder.new().with_host_only().build()
    yield c
    c.close()


INTS = [("int8", pa.int8()), ("int16", pa.int16()), ("int32", pa.int32()), ("int64", pa.int64()),
        ("uint8", pa.uint8()), ("uint16", pa.uint16()), ("uint32", pa.uint32()), ("uint64", pa.uint64())]


@pytest.mark.parametrize("name,dtype", INTS)
def test_transcoder_integers_byte_identical_to_oracle(host_cache, oracle, name, dtype):
    lo = oracle
    rng = np.random.default_rng(len(name))
    np_dtype = lo.PHYS_NP[lo.PHYS[name]]
    info = np.iinfo(np_dtype)
    for n in (0, 1, 100, 1024, 8192, 3000):
        span = int(rng.integers(1, 2 ** min(40, np.dtype(np_dtype).itemsize * 8 - 1)))
        base = int(rng.integers(int(info.min) // 2, int(info.max) // 2 - span))
        vals = (rng.integers(0, span, size=n) + base).astype(np_dtype)
        for nullable in (False, True):
            valid = (rng.random(n) < 0.7) if nullable else None
            arr = pa.array(vals, type=dtype, mask=None if valid is None else ~valid)
            got = host_cache.transcode(arr)
            zeroed = np.where(valid, vals, 0).astype(np_dtype) if valid is not None else vals
            want = lo.encode_primitive(lo.PHYS[name], zeroed, valid)
            if valid is None or n == 0:
                assert got == want
            dec, dvalid = lo.to_arrow_fixed(got)
            if valid is None:
                assert dec.tolist() == vals.tolist()
            elif dvalid is None:
                assert valid.all() and dec.tolist() == vals.tolist()
            else:
                assert dvalid.tolist() == valid.tolist() and dec[valid].tolist() == vals[valid].tolist()


def test_transcoder_sliced_arrays_and_temporal_types(host_cache, oracle):
    lo = oracle
    base = pa.array([None, 5, 6, None, 8, 9, 10], type=pa.int32())
    sl = base.slice(1, 5)  # non-zero offset, validity bit offset 1 (bit_pack_array.rs:215-223)
    dec, valid = lo.to_arrow_fixed(host_cache.transcode(sl))
    assert valid.tolist() == [True, True, False, True, True] and dec[valid].tolist() == [5, 6, 8, 9]
    for t, name in ((pa.date32(), "date32"), (pa.date64(), "date64"), (pa.timestamp("s"), "timestamp[s]"),
                    (pa.timestamp("ms"), "timestamp[ms]"), (pa.timestamp("us"), "timestamp[us]"),
                    (pa.timestamp("ns"), "timestamp[ns]")):
        arr = pa.array([1, 86400000, None, -5], type=pa.int64() if t != pa.date32() else pa.int32()).cast(t)
        b = host_cache.transcode(arr)
        assert lo.array_info(b).phys == lo.PHYS[name]
        dec, valid = lo.to_arrow_fixed(b)
        assert dec[valid].tolist() == [1, 86400000, -5]
    # timezone-aware timestamps and booleans stay Arrow (transcode.rs:104-107, :150-154)
    assert host_cache.transcode(pa.array([1, 2], type=pa.timestamp("us", tz="UTC"))) is None
    assert host_cache.transcode(pa.array([True, False])) is None


def test_transcoder_floats_decimals(host_cache, oracle):
    lo = oracle
    import decimal
    rng = np.random.default_rng(1)
    for dt, pat in ((np.float32, pa.float32()), (np.float64, pa.float64())):
        vals = rng.normal(size=5000).astype(dt).round(3)
        vals[:3] = [np.nan, np.inf, -np.inf]
        dec, _ = lo.to_arrow_fixed(host_cache.transcode(pa.array(vals, type=pat)))
        np.testing.assert_array_equal(dec, vals)
    d = [decimal.Decimal("0.05"), None, decimal.Decimal("123456.78"), decimal.Decimal("0.00")]
    b = host_cache.transcode(pa.array(d, type=pa.decimal128(15, 2)))
    info = lo.array_info(b)
    assert (info.logical, info.decimal_precision, info.decimal_scale) == (lo.LOGICAL_DECIMAL, 15, 2)
    dec, valid = lo.to_arrow_fixed(b)
    assert [x for x, v in zip(dec, valid) if v] == [5, 12345678, 0]
    # negative decimals do not fit u64: the reference uses LiquidFixedLenByteArray for those (out of scope here)
    assert host_cache.transcode(pa.array([decimal.Decimal("-1.00")], type=pa.decimal128(15, 2))) is None


def test_transcoder_strings_roundtrip_and_symbol_table_per_path(host_cache, oracle):
    lo = oracle
    urls = pq.read_table(os.path.join(GOLD, "nano_hits_cols.parquet"), columns=["URL"])["URL"].combine_chunks()
    a, b = urls.slice(0, 8192), urls.slice(8192, 8192)
    la = host_cache.transcode(a, lc.CacheExpression.SUBSTRING_SEARCH, path_id=77)
    st = lo.symtab_load(host_cache.symbol_table(77))
    lb = host_cache.transcode(b, lc.CacheExpression.SUBSTRING_SEARCH, path_id=77)  # reuses the trained table
    assert host_cache.symbol_table(77) == lo.symtab_bytes(st)
    assert lo.filter_byte_view(la, st) == [u.encode() for u in a.to_pylist()]
    assert lo.filter_byte_view(lb, st) == [u.encode() for u in b.to_pylist()]
    assert len(la) < 0.3 * sum(len(u) for u in a.to_pylist())  # dictionary + FSST actually compress
    r = lo.eval_predicate(la, lo.LIKE, b"%tours%", symtab=st)
    assert int(r.values.sum()) == sum("tours" in u for u in a.to_pylist())
    # no hint -> no fingerprints section
    plain = host_cache.transcode(a, None, path_id=77)
    assert int(np.frombuffer(plain[32:36], np.uint32)[0]) == 0 and int(np.frombuffer(la[32:36], np.uint32)[0]) > 0
    for t in (pa.binary(), pa.string_view(), pa.binary_view()):
        vals = ["a", None, "bc", "a", ""]
        arr = pa.array([None if v is None else (v.encode() if "binary" in str(t) else v) for v in vals], type=t)
        lq = host_cache.transcode(arr, None, path_id=5)
        got = lo.filter_byte_view(lq, lo.symtab_load(host_cache.symbol_table(5)))
        assert got == [None if v is None else v.encode() for v in vals]
    dict_arr = pa.DictionaryArray.from_arrays(pa.array([0, 1, None, 1, 0], type=pa.uint16()), pa.array(["x", "yy"]))
    lq = host_cache.transcode(dict_arr, None, path_id=6)
    assert lo.filter_byte_view(lq, lo.symtab_load(host_cache.symbol_table(6))) == [b"x", b"yy", None, b"yy", b"x"]


def test_liquid_expr_validation_matrix():
    """Mirrors LiquidExpr::try_new / supports_expr (liquid_expr.rs:65-148) and its tests (:210-258)."""
    E, H = lc.LiquidExpr.try_new, lc.CacheExpression.SUBSTRING_SEARCH
    assert E("=", "x", pa.string()) is not None                      # validates_byte_comparison_with_literal
    assert E("like", "%abc%", pa.string()) is None                   # rejects_byte_like_without_substring_hint
    assert E("like", "%abc%", pa.string(), H) is not None            # accepts_byte_like_with_substring_hint
    assert E(">", 42, pa.int32()) is not None                        # validates_numeric_comparison
    assert E("like", "%a%", pa.int32(), H) is None
    assert E(">", "x", pa.int64()) is None
    assert E("=", 3, pa.string()) is None
    assert E(None, True, pa.string()) is not None and E(None, True, pa.int32()) is None
    for t in (pa.int8(), pa.uint64(), pa.float32(), pa.float64(), pa.date32(), pa.date64(), pa.decimal128(15, 2),
              pa.timestamp("ms")):
        for op in ("=", "!=", "<", "<=", ">", ">="):
            assert E(op, 1, t) is not None, (t, op)
    assert E(">", 1, pa.timestamp("ms", tz="UTC")) is None
    assert E("=", b"x", pa.dictionary(pa.uint16(), pa.string())) is not None
    assert E("=", "x", pa.string_view()) is not None and E("=", b"x", pa.binary_view()) is not None
    assert E("ilike", "%a%", pa.string(), H) is None


def test_entry_id_packing():
    e = lc.ParquetArrayID.new(3, 7, 13, 42)
    assert int(e) == (3 << 48) | (7 << 32) | (13 << 16) | 42       # id.rs:15-21
    assert lc.ParquetArrayID.column_access_path(e) == int(e) >> 16
    with pytest.raises(ValueError):
        lc.ParquetArrayID.new(1 << 16, 0, 0, 0)


def test_row_range_sharding_keeps_columns_together():
    from liquid_cache_amd import sharding as sh
    ids = [lc.ParquetArrayID.new(f, rg, col, b) for f in range(2) for rg in range(3) for b in range(5) for col in (1, 13)]
    for world in (1, 2, 3, 8):
        shards = sh.assign_row_ranges(ids, world)
        assert sorted(sum(shards, [])) == sorted(int(e) for e in ids)
        owner = {}
        for r, s in enumerate(shards):
            for e in s:
                assert owner.setdefault(sh.row_range_key(e), r) == r  # all columns of a row range on one rank
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 2 * 2  # balanced within one row range
        # contiguous runs in (file, row group, batch) order
        last = None
        for s in shards:
            for e in s:
                k = sh.row_range_key(e)
                assert last is None or k >= last
                last = k


def test_expression_hint_mirror():
    """CacheExpression::extract_date32 / Date32Field mirror (cache/expressions.rs:82-84, squeezed_date32_array.rs)."""
    import liquid_cache_amd as lc
    h = lc.CacheExpression.extract_date32(lc.Date32Field.MONTH)
    assert isinstance(h, lc.ExtractDate32) and h.field == 1
    assert lc.CacheExpression.extract_date32("year").field == 0
    assert lc.CacheExpression.extract_date32("DayOfWeek").field == 3


def test_bench_defaults_name_the_metric_workload():
    """bench.py with no flags = BASELINE.json's metric configuration; the full-size GPU tests stage with these args."""
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.workload, a.rows, a.batch_size, a.needle) == (1, "url_like", 99_997_497, 8192, "google")
    for name in ("uniques", "row_group_batches", "needle_ppm", "no_fingerprints", "int_bits", "seed", "steps", "warmup"):
        assert hasattr(a, name)


def test_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: both headers must compile as C99 (and as C++) without torch / HIP types."""
    import shutil
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "liquid_cache_amd.h"\n#include "liquid_cache_amd_bench.h"\n'
                   'int main(void) { return (int)sizeof(lc_predicate) * 0; }\n')
    inc = os.path.join(ROOT, "include")
    if shutil.which("gcc"):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                       check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)], check=True)


def test_bench_cpu_baseline_counts_are_the_substring_truth():
    """bench.py's checker leg: the C oracle's COUNT(*) over regenerated + transcoded batches of the bench column — for the
    headline needle and for the extra parity needles — equals a plain substring test on the raw strings.  Batches of row
    group 29 (symbol table trained on its first batch, as in the bench): the table on which `mail` exposed the GPU walker's
    speculative matches; the oracle must not share such a defect with the kernel it checks."""
    import bench
    args = bench.parse_args([])
    first = 29 * args.row_group_batches
    cache = lc.LiquidCacheBuilder.new().with_host_only().build()
    L = N.load()
    bs = args.batch_size
    needles = ("google",) + bench.EXTRA_PARITY_NEEDLES
    got = {n: 0 for n in needles}
    want = {n: 0 for n in needles}
    from oracle import liquid_oracle as lo
    offs = np.zeros(bs + 1, np.int32)
    data = np.zeros(bs * 512, np.uint8)
    for b in (first, first + 3, first + 12):
        n = N.load_bench().lc_synth_url_batch(args.seed, b, bs, args.uniques, args.needle_ppm, offs.ctypes.data, data.ctypes.data, data.size)
        raw = data[:n].tobytes()
        strs = [raw[offs[i]: offs[i + 1]] for i in range(bs)]
        arr = pa.StringArray.from_buffers(bs, pa.py_buffer(offs.copy()), pa.py_buffer(data[:n].copy()))
        eid = lc.ParquetArrayID.new(0, b // args.row_group_batches, 13, b % args.row_group_batches)
        path = lc.ParquetArrayID.column_access_path(eid)
        blob = cache.transcode(arr, lc.CacheExpression.SUBSTRING_SEARCH, path)
        st = lo.symtab_load(cache.symbol_table(path))
        for nd in needles:
            got[nd] += int(lo.bench_eval_batches([blob], [st], lo.LIKE, ("%" + nd + "%").encode(), 1))
            want[nd] += sum(nd.encode() in s for s in strs)
    cache.close()
    assert got == want and want["mail"] > 0 and want["ru/"] > want["mail"]


def test_inverted_row_lists_layout(product_lib):
    """The row lists lc_stage attaches to byte-view entries: offsets[d + 1] is a running sum, the rows of key k
    are exactly the VALID rows whose key is k (any order within a key), rows under nulls and keys >= d are not listed."""
    rng = np.random.default_rng(12)
    for n, d, p_null in ((8192, 2200, 0.05), (8192, 1, 0.0), (77, 500, 0.5), (1, 1, 0.0), (8192, 8192, 0.0),
                         (16384, 3000, 0.1), (65535, 9000, 0.0), (65535, 1, 0.0)):  # batch sizes up to 65,535 rows
        keys = rng.integers(0, d, size=n).astype(np.uint16)
        valid = rng.random(n) >= p_null
        keys[~valid] = rng.integers(0, 65536, size=int((~valid).sum())).astype(np.uint16)   # garbage under nulls
        bitmap = np.packbits(valid, bitorder="little")
        out = np.zeros(d + 1 + n + 32, np.uint16)
        got = N.load_bench().lc_debug_row_lists(keys.ctypes.data, bitmap.ctypes.data if p_null else None, n, d,
                                            out.ctypes.data, out.size)
        assert got == out.size
        off, rows = out[: d + 1].astype(np.int64), out[d + 1: d + 1 + n]
        assert off[0] == 0 and (np.diff(off) >= 0).all() and off[d] == int(valid.sum())
        for k in set(rng.integers(0, d, size=min(d, 200)).tolist()) | {0, d - 1}:
            want = np.nonzero(valid & (keys == k))[0]
            assert sorted(rows[off[k]: off[k + 1]].tolist()) == want.tolist(), (n, d, k)
    # out of contract: more rows than an entry with lists may have, no dictionary, a short buffer
    big = np.zeros(65536, np.uint16)
    out = np.zeros(70000, np.uint16)
    assert N.load_bench().lc_debug_row_lists(big.ctypes.data, None, 65536, 10, out.ctypes.data, out.size) == 0  # u16 offsets
    assert N.load_bench().lc_debug_row_lists(big.ctypes.data, None, 100, 0, out.ctypes.data, out.size) == 0
    assert N.load_bench().lc_debug_row_lists(big.ctypes.data, None, 100, 10, out.ctypes.data, 50) == 0


def test_entry_map_against_unordered_map(tmp_path):
    """csrc/lc_internal.hpp's EntryMap (open addressing + prefetching find_many behind lc_ctx::entries) == std::unordered_map
    under 400,000 random emplace / erase / find / find_many operations over ParquetArrayID-shaped keys."""
    import subprocess
    exe = str(tmp_path / "entry_map_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "entry_map_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "entry map ok" in r.stdout, r.stdout + r.stderr
