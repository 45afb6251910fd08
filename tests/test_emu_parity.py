"""The engine's HIP kernels, run lane by lane on the CPU (tests/hipemu), against the oracle.

The build container has no GPU, so the `-m gpu` parity suite cannot run here.  tests/hipemu compiles the SAME sources
(magent_amd/csrc/*.hip, unchanged) as plain C++ against a stand-in <hip/hip_runtime.h>: workgroups run one after another,
their threads as fibers that meet at __syncthreads() / wave ballots.  This checks the kernels' logic -- indexing, the
attack / move fixed points, atomics protocols, barrier placement -- before a GPU minute is spent; it is test
infrastructure like oracle/ (only tests/ and tools/fuzz_parity.py load it), it is not a CPU path of the product, and it
does not replace the GPU suite, which runs the hipcc build on the MI355X.

HIPEMU_SCRAMBLE=<seed> runs lanes and workgroups in a pseudo-random order: results must not depend on it.
"""
import os
import subprocess
import sys

import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# a slice that touches every phase family: dense attack chains, striped moves, multi-cell bodies, rules, goals, turn_mode, food
SLICE = ["battle_small_dense", "battle_brawl", "battle_walls", "gather", "forest", "tri_rect", "pursuit_dense", "bodies", "quad"]


@pytest.fixture(scope="module")
def emu():
    return H.ensure_emu()


@pytest.mark.parametrize("name", SLICE)
def test_emulated_kernels_match_oracle(emu, name):
    sc = H.scenarios()[name]
    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, emu), name + " (hipemu)")


def test_emulated_kernels_do_not_depend_on_lane_order(emu):
    """the same under two scrambled lane / workgroup orders (a subprocess each: the order is fixed when the library starts)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "for n in ('battle_brawl', 'tri_rect', 'bodies'):\n"
            "    sc = H.scenarios()[n]\n"
            "    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, H.ensure_emu()), n + ' (hipemu, scrambled)')\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    for seed in ("1", "7"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_SCRAMBLE=seed, OMP_NUM_THREADS="1"),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]
