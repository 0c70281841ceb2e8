"""Per-kernel resource usage (scratch bytes per lane, VGPRs, LDS) read from the gfx950 code objects embedded in a built
library's .hip_fatbin section — what the compiler decided, without recompiling.
usage: python scripts/kernel_resources.py [path/to/lib.so] [--min-scratch N]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path):
    """The amdgcn ELF images of every offload bundle in the library (one bundle per translation unit)."""
    out = subprocess.run([READELF, "-S", "-W", so_path], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.hip_fatbin\s+\S+\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", out)
    if not m:
        return []
    off, size = int(m.group(1), 16), int(m.group(2), 16)
    with open(so_path, "rb") as f:
        f.seek(off)
        data = f.read(size)
    images = []
    pos = data.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and s:
                images.append((triple, data[pos + o:pos + o + s]))
        pos = data.find(MAGIC, pos + 1)
    return images


def kernels(so_path):
    """{kernel symbol: {"scratch": bytes per lane, "vgprs": n, "sgprs": n, "lds": bytes per workgroup}}"""
    import yaml
    res = {}
    for _, image in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(image)
            tmp.flush()
            notes = subprocess.run([READELF, "--notes", tmp.name], capture_output=True, text=True, check=True).stdout
        lines = notes.splitlines()
        try:
            first = next(i for i, l in enumerate(lines) if l.strip() == "---") + 1
            last = next(i for i in range(first, len(lines)) if lines[i].strip() == "...")
        except StopIteration:
            continue
        meta = yaml.safe_load("\n".join(lines[first:last])) or {}
        for k in meta.get("amdhsa.kernels", []):
            res[k[".name"]] = {"scratch": int(k.get(".private_segment_fixed_size", 0)), "vgprs": int(k.get(".vgpr_count", 0)),
                               "sgprs": int(k.get(".sgpr_count", 0)), "lds": int(k.get(".group_segment_fixed_size", 0)),
                               "dynamic_stack": bool(k.get(".uses_dynamic_stack", False))}
    return res


def main():
    import argparse
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap = argparse.ArgumentParser()
    ap.add_argument("library", nargs="?", default=os.path.join(here, "liquid_cache_amd", "libliquid_cache_amd.so"))
    ap.add_argument("--min-scratch", type=int, default=0)
    a = ap.parse_args()
    ks = kernels(a.library)
    for name in sorted(ks, key=lambda n: -ks[n]["scratch"]):
        k = ks[name]
        if k["scratch"] >= a.min_scratch:
            print("%6d B scratch  %3d VGPRs  %6d B LDS  %s" % (k["scratch"], k["vgprs"], k["lds"], name[:150]))


if __name__ == "__main__":
    main()
