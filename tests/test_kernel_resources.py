"""What the compiler decided for the hot kernels, read from the built library's gfx950 code objects (no GPU needed).

Round 5 lost 1.4x on every narrow-integer scan to ONE extra struct field: RegEntryArgs grew past the 16 dwords the calling
convention passes in registers, every call site of the per-width functions got a by-value stack copy, and the kernels went from
8-24 bytes of scratch per lane to 2.3-4.6 KB — the instruction stream of the passes unchanged, the waves in flight capped by the
scratch reservation.  No parity test can see that; this one does."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_resources as KR  # noqa: E402

LIB = os.path.join(ROOT, "liquid_cache_amd", "libliquid_cache_amd.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(KR.READELF) and os.path.exists(LIB)),
                                reason="needs the built library and llvm-readelf")

# kernels on the paths bench.py times per scan: scratch beyond a few spilled dwords means a register array or a by-value
# argument went to memory
HOT = ("k_fixed_pred_reg", "k_fixed_chain", "k_fixed_pred", "k_like_flat", "k_like_scanall", "k_str_pred", "k_mask_to_hits",
       "k_str_gather_hits", "k_fixed_gather_hits", "k_sum_product", "k_date_lossy")


@pytest.fixture(scope="module")
def kernels():
    ks = KR.kernels(LIB)
    assert len(ks) > 100, "the library's code objects were not found (%d kernels)" % len(ks)
    return ks


def test_hot_kernels_keep_their_state_in_registers(kernels):
    seen = set()
    for name, k in kernels.items():
        for h in HOT:
            if h in name:
                seen.add(h)
                assert k["scratch"] <= 64, "%s: %d bytes of scratch per lane" % (name, k["scratch"])
                assert not k["dynamic_stack"], name
    assert {"k_fixed_pred_reg", "k_fixed_chain", "k_like_flat", "k_str_pred"} <= seen


def test_no_kernel_reserves_kilobytes_of_scratch(kernels):
    worst = max(kernels.items(), key=lambda kv: kv[1]["scratch"])
    assert worst[1]["scratch"] <= 256, "%s: %d bytes of scratch per lane" % (worst[0], worst[1]["scratch"])


def test_register_resident_kernels_use_no_lds(kernels):
    # (the ballot-through-LDS forms are A/B builds; the shipped kernels park ballots in lanes)
    for name, k in kernels.items():
        if "k_fixed_pred_reg" in name:
            assert k["lds"] == 0, "%s: %d bytes of LDS" % (name, k["lds"])


def test_index_builder_keeps_two_workgroups_per_cu(kernels):
    """k_flat_build runs 16-wave workgroups, two per CU: that needs 8 wave slots per SIMD, i.e. at most 64 VGPRs AND at most 80
    SGPRs — 81 are allocated as 96 (+16 for the trap handler) and leave a SIMD 7 slots.  Round 6 lost a third of the builder's
    speed to six SGPRs (profiles/r6/ab_flat_build.txt); nothing but the SQ counters' wave-cycles showed it."""
    seen = 0
    for name, k in kernels.items():
        if "k_flat_build" in name:
            seen += 1
            assert k["vgprs"] <= 64 and k["sgprs"] <= 80, "%s: %d VGPRs, %d SGPRs" % (name, k["vgprs"], k["sgprs"])
            assert k["scratch"] == 0, name
    assert seen >= 2
