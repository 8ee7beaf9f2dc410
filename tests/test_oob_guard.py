"""``-m gpu``: two instruments against out-of-bounds accesses of the stage entries (tests/oob_runner.py).

1. Every pointer argument inside NaN-filled margins: a value read from outside an argument that REACHES a result turns it into a NaN, a
   write outside an argument changes a margin.  torch's caching allocator would otherwise serve such accesses silently from a neighbouring
   tensor.  This instrument does NOT see a read whose value nobody uses — the out-of-bounds LDS-DMA read car_linear16.hip had until commit
   0d74f26 passes it (profiles/round5_oob_guard.md, profiles/round6_oob_guard.md).
2. For that class: the library built with -DCAR_BOUNDS (tools/build_bounds.py), in which the LDS-DMA / buffer-load / row-load helpers
   compare their source range with the extent their launcher passed and TRAP outside it.  The second test below re-introduces the
   pre-0d74f26 bug into today's car_linear16.hip and checks that this build dies on it (and that the product's kernels do not)."""
import os
import subprocess
import sys

import pytest

import oob_runner as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", [n for fam in O.FAMILIES.values() for n in fam])
def test_entries_stay_inside_their_buffers(name):
    O.CASES[name](True)


def _family(fam, **env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oob_runner.py"), fam], capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, **env), cwd=ROOT)


def test_bounds_build_traps_on_the_reverted_linear16_and_not_on_the_product():
    """-DCAR_BOUNDS: the x3 and fused families run to the end on today's kernels; with commit 0d74f26 reverted in car_linear16.hip the x3
    family dies (a trap) in a narrow column group's case, which the NaN-margin harness lets pass."""
    built = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_bounds.py"), "--reintroduce-0d74f26"], capture_output=True, text=True,
                           timeout=900, cwd=ROOT)
    assert built.returncode == 0, built.stdout + built.stderr
    dev = os.path.join(ROOT, "tools", "_dev")
    for fam in ("x3", "fused", "exchange"):
        ok = _family(fam, CAR_OOB_FULL_LIB=os.path.join(dev, "libcar_bounds.so"))
        assert ok.returncode == 0 and f"DONE {fam}: 0 failed" in ok.stdout, ok.stdout[-1500:] + ok.stderr[-1500:]
    blind = _family("x3", CAR_OOB_LIB=os.path.join(dev, "liboldlin16.so"))
    assert blind.returncode == 0 and "DONE x3: 0 failed" in blind.stdout, "the NaN-margin harness was not expected to see the discarded read"
    caught = _family("x3", CAR_OOB_LIB=os.path.join(dev, "liboldlin16_bounds.so"))
    assert caught.returncode != 0 and "DONE x3" not in caught.stdout, caught.stdout[-1500:]
    assert "OK x3_nt18" in caught.stdout and "RUN x3_nt4" in caught.stdout and "OK x3_nt4" not in caught.stdout, caught.stdout[-1500:]
