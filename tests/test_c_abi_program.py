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
