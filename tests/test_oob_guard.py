"""``-m gpu``: every pointer argument of the stage entries flush against unmapped device memory (tests/oob_runner.py,
tests/host/guard_alloc.cpp).  torch's caching allocator hides reads past a buffer — the out-of-bounds LDS-DMA read that
car_linear16.hip had until commit 0d74f26 was found by reading the code, no test could see it; here such an access is a GPU page
fault that kills the family's subprocess (tools/oob_selfcheck.sh shows that the old kernel does die here).  One process per family of
cases: a fault takes the whole HIP context with it; both placements (buffer end / buffer start at the edge of mapped memory) run in it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNNER = os.path.join(ROOT, "tests", "oob_runner.py")


def run_family(name, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, RUNNER, name], capture_output=True, text=True, timeout=900, env=e)


@pytest.mark.parametrize("family", ["x3", "linear", "gather", "fused", "tail"])
def test_entries_stay_inside_their_buffers(family):
    r = run_family(family)
    if r.returncode == 77:
        pytest.skip(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "HIP virtual-memory API unavailable")
    assert r.returncode == 0 and f"DONE {family}" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
