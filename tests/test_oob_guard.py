"""``-m gpu``: every pointer argument of the stage entries inside NaN-filled margins (tests/oob_runner.py).  torch's caching allocator
hides reads past a buffer — the out-of-bounds LDS-DMA read that car_linear16.hip had until commit 0d74f26 was found by reading the
code, no test could see it: here a value read from outside an argument that reaches a result turns it into a NaN, and a write outside
an argument changes a margin (tools/oob_selfcheck.sh shows that the old kernel fails here)."""
import pytest

import oob_runner as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for fam in O.FAMILIES.values() for n in fam])
def test_entries_stay_inside_their_buffers(name):
    O.CASES[name](True)
