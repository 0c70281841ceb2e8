"""The drop-in boundary from a plain C program: tests/c_abi/readme_vector.c is compiled with gcc against
include/liquid_cache_amd.h, linked with the in-tree libliquid_cache_amd.so and (on a GPU box) run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "readme_vector.c")


def _build(tmp_path, product_lib):
    exe = str(tmp_path / "readme_vector")
    libdir = os.path.join(ROOT, "liquid_cache_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                    "-L", libdir, "-l:libliquid_cache_amd.so", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_c_program_compiles_and_links(tmp_path, product_lib):
    assert os.path.exists(_build(tmp_path, product_lib))


@pytest.mark.gpu
def test_c_program_runs_the_readme_vector(tmp_path, product_lib):
    exe = _build(tmp_path, product_lib)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c abi ok" in r.stdout


def test_device_fsst_encoder_model_equals_host_encoder(tmp_path):
    """The encoder k_bv_build runs per dictionary value (lc_fsst_device.hpp: table layout, matcher, compression loop — the
    same source the kernel compiles) executed on the CPU: its code stream equals FsstEncoder::compress for 19,200 values
    over 12 trained tables (URLs, all byte values, four-letter alphabets, escape-heavy text; own and foreign data)."""
    csrc = os.path.join(ROOT, "liquid_cache_amd", "csrc")
    exe = str(tmp_path / "fsst_device_encoder_model")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_abi", "fsst_device_encoder_model.cpp"),
                    os.path.join(csrc, "lc_fsst.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "device encoder model ok" in r.stdout


def test_concurrent_callers_program_compiles_and_links(tmp_path, product_lib):
    exe = str(tmp_path / "concurrent_callers")
    libdir = os.path.join(ROOT, "liquid_cache_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "concurrent_callers.c"), "-o", exe,
                    "-L", libdir, "-l:libliquid_cache_amd.so", "-Wl,-rpath," + libdir], check=True)
    assert os.path.exists(exe)


def test_rowgroup_reader_program_compiles_and_links(tmp_path, product_lib):
    exe = str(tmp_path / "rowgroup_reader")
    libdir = os.path.join(ROOT, "liquid_cache_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "rowgroup_reader.c"), "-o", exe,
                    "-L", libdir, "-l:libliquid_cache_amd.so", "-Wl,-rpath," + libdir], check=True)
    assert os.path.exists(exe)
