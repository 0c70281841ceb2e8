"""CPU model of k_str_pred's lane-parallel LIKE walk (words of 8 compressed bytes, every word first walked from state 0 /
"next byte is a code", start states then corrected to a fixpoint) on batches of the bench's synthetic URL column.

Old rule: a match found by ANY of the walks counts.  New rule: only the state at the fixpoint counts.  With the old rule the
model reproduces the over-count the GPU showed for `%mail%` (bench.py --needle mail, per-batch counts dumped with
LC_DUMP_COUNTS) exactly; with the new rule it equals the ground truth.  No GPU needed: python scripts/micro/emulate_like_walk.py"""
import os, sys, ctypes as C, numpy as np, pyarrow as pa
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import liquid_cache_amd as lc
from liquid_cache_amd import _native as N
from oracle import liquid_oracle as lo
L = N.load()
cache = lc.LiquidCacheBuilder.new().with_host_only().build()
bs=8192; seed=42; uniques=2200; ppm=159
needle=sys.argv[1].encode() if len(sys.argv)>1 else b"mail"; m=len(needle)
class BV(C.Structure):
    _fields_=[("arrow_type",C.c_int32),("n",C.c_uint32),("d",C.c_uint32),("nullable",C.c_int32),("all_null",C.c_int32),
              ("keys",C.POINTER(C.c_uint16)),("key_validity",C.POINTER(C.c_uint8)),("fsst",C.POINTER(C.c_uint8)),("fsst_len",C.c_uint32),
              ("uncompressed_bytes",C.c_uint64),("offsets",C.POINTER(C.c_uint32)),("prefix_keys",C.POINTER(C.c_uint8)),
              ("shared_prefix",C.POINTER(C.c_uint8)),("shared_prefix_len",C.c_uint32),("fingerprints",C.POINTER(C.c_uint8)),
              ("compact_offsets_size",C.c_uint32),("offset_bytes",C.c_int32)]
ol=lo.lib()
ol.lo_bv_parse.argtypes=[C.c_void_p,C.c_size_t,C.POINTER(BV)]; ol.lo_bv_parse.restype=C.c_int
def kmp_delta(nd):
    m=len(nd); fail=[0]*(m+2); k=0
    for i in range(1,m):
        while k>0 and nd[i]!=nd[k]: k=fail[k]
        if nd[i]==nd[k]: k+=1
        fail[i+1]=k
    delta=[[0]*256 for _ in range(m+1)]
    for s in range(m+1):
        for b in range(256):
            if s==m: v=m
            elif nd[s]==b: v=s+1
            elif s==0: v=0
            else: v=delta[fail[s]][b]
            delta[s][b]=v
    return delta
delta=kmp_delta(needle)
def bigram_bit(a,b): return ((((a<<8)|b)*40503)>>7)&127 & 0x7f
def bb(a,b): return (((((a<<8)|b)*40503)&0xFFFFFFFF)>>7)&127
nbits=set(bb(needle[i],needle[i+1]) for i in range(m-1))
def run(b, gpu_extra):
    offs = np.zeros(bs + 1, np.int32); data = np.zeros(bs * 512, np.uint8)
    n = L.lc_synth_url_batch(seed, b, bs, uniques, ppm, offs.ctypes.data, data.ctypes.data, data.size)
    arr = pa.StringArray.from_buffers(bs, pa.py_buffer(offs[: bs + 1].copy()), pa.py_buffer(data[:n].copy()))
    eid = lc.ParquetArrayID.new(0, b // 54, 13, b % 54)
    path = lc.ParquetArrayID.column_access_path(eid)
    blob = cache.transcode(arr, lc.CacheExpression.SUBSTRING_SEARCH, path)
    st = lo.symtab_load(cache.symbol_table(path))
    a=np.frombuffer(blob,np.uint8).copy()
    bv=BV(); rc=ol.lo_bv_parse(a.ctypes.data,a.size,C.byref(bv)); assert rc==0,rc
    d=bv.d; keys=np.ctypeslib.as_array(bv.keys,(bv.n,)).copy(); off=np.ctypeslib.as_array(bv.offsets,(d+1,)).copy()
    fsst=bytes(np.ctypeslib.as_array(bv.fsst,(bv.fsst_len,)))
    sym=[int(st.sym[c]).to_bytes(8,'little')[:st.len[c]] for c in range(256)]
    # fold: code transitions
    def code_next(s,c):
        cur=s
        for by in sym[c]: cur=delta[cur][by]
        return cur
    rows_per=np.bincount(keys,minlength=d)
    over=0; over_fix=0; fp_vals=[]
    for v in range(d):
        cb=fsst[off[v]:off[v+1]]
        # decode + truth
        dec=bytearray(); i=0
        while i<len(cb):
            c=cb[i]
            if c==255: dec.append(cb[i+1]); i+=2
            else: dec+=sym[c]; i+=1
        truth = needle in dec
        sig=set(bb(dec[i],dec[i+1]) for i in range(len(dec)-1))
        if not nbits<=sig: 
            assert not truth
            continue
        # states: (s, lit) lit=1 means next byte is literal
        def walk(state, word):
            s,lit=state
            for c in word:
                if s==m and not lit:  # absorbing incl role? image: row m: entry[255] -> m (stays), codes -> m
                    s=m; continue
                if lit: s=delta[s][c]; lit=0
                elif c==255: lit=1 if s!=m else 0
                else: s=code_next(s,c)
            return (s,lit)
        words=[cb[i:i+8] for i in range(0,max(len(cb),1),8)] or [b'']
        nwd=len(words)
        s_in=[(0,0)]*nwd; e=[walk((0,0),w) for w in words]
        hit=[x==(m,0) for x in e]
        while True:
            prev=[(0,0)]+e[:-1]
            prev=[(0,0) if (i==0 or p==(m,0)) else p for i,p in enumerate(prev)]
            ch=[prev[i]!=s_in[i] for i in range(nwd)]
            if not any(ch): break
            for i in range(nwd):
                if ch[i]:
                    s_in[i]=prev[i]; e[i]=walk(s_in[i],words[i]); hit[i]=hit[i] or e[i]==(m,0)
        gpu=any(hit); fix=any(x==(m,0) for x in e)
        if gpu and not truth: over+=rows_per[v]; fp_vals.append((v,bytes(dec)[:80]))
        if fix!=truth: over_fix+=1
    print('batch',b,'emulated overcount rows',over,'observed',gpu_extra,'fixed-mismatch values',over_fix, fp_vals[:2])
# rows the GPU counted beyond the ground truth (round 2, before the fix); batch 1566 trains the row group's symbol table
OBSERVED = {1566: 6, 1567: 5, 1569: 18, 1578: 132}
if len(sys.argv) > 2:   # broad check of the new rule: python emulate_like_walk.py <needle> <first row group> <row groups>
    for rg in range(int(sys.argv[2]), int(sys.argv[2]) + int(sys.argv[3])):
        run(rg * 54, -1)
        run(rg * 54 + 1 + rg % 53, -1)
else:
    for b, extra in OBSERVED.items():
        run(b, extra)
