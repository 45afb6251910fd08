"""The C-ABI is usable from plain C: tests/c_client/battle_client.c includes include/magent_runtime_api.h, links the
library and plays an episode with nothing but pointers and ints."""
import os
import subprocess

import pytest

import helpers as H

SRC = os.path.join(H.ROOT, "tests", "c_client", "battle_client.c")
INC = os.path.join(H.ROOT, "include")


def _build(tmp_path, lib, name):
    exe = str(tmp_path / name)
    libdir, libfile = os.path.dirname(lib), os.path.basename(lib)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", INC, SRC, "-o", exe, "-L", libdir, "-l:" + libfile,
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_is_valid_c():
    subprocess.check_call(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-x", "c",
                           os.path.join(INC, "magent_runtime_api.h")])


def test_c_client_links_against_the_product_library(tmp_path):
    """link only (no GPU here): every symbol the client uses resolves in libmagent.so"""
    if not os.path.exists(H.HIP_LIB):
        pytest.skip("libmagent.so not built")
    _build(tmp_path, H.HIP_LIB, "client_hip")


def test_c_client_runs_on_the_oracle(tmp_path):
    exe = _build(tmp_path, H.ensure_oracle(), "client_oracle")
    out = subprocess.run([exe, "30", "120", "5"], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == 5 and out[0].startswith("step 0 done 0 num 120 120 checksum ")


@pytest.mark.gpu
def test_c_client_hip_matches_oracle(tmp_path):
    want = subprocess.run([_build(tmp_path, H.ensure_oracle(), "client_oracle"), "60", "900", "12"],
                          capture_output=True, text=True, check=True).stdout
    got = subprocess.run([_build(tmp_path, H.HIP_LIB, "client_hip"), "60", "900", "12"],
                         capture_output=True, text=True, check=True).stdout
    assert got == want and got.count("checksum") == 12
