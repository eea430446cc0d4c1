"""The selection network of the 5 x 5 median kernel (imageprocessing_amd/csrc/median_nets.inc) selects the median of
every input: by the 0-1 principle it is enough that it does so for every input of zeros and ones, and those are all
tried here (2^25 inputs, 64 per machine word)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _networks():
    text = open(os.path.join(ROOT, "imageprocessing_amd", "csrc", "median_nets.inc")).read()
    out = {}
    for name in ("MED25",):
        net = re.search(r"#define %s_NET \{(.*)\}" % name, text).group(1)
        pairs = [(int(a), int(b)) for a, b in re.findall(r"\{(\d+), (\d+)\}", net)]
        assert len(pairs) == int(re.search(r"#define %s_LEN (\d+)" % name, text).group(1))
        out[name] = (pairs, int(re.search(r"#define %s_OUT (\d+)" % name, text).group(1)))
    return out


@pytest.mark.parametrize("name,n", [("MED25", 25)])
def test_network_selects_the_median_of_every_zero_one_input(name, n):
    pairs, out = _networks()[name]
    total = 1 << n
    idx = np.arange(total, dtype=np.uint64)
    # bit k of word j of v[i] = value of input i in test vector 64 j + k
    v = [np.packbits(((idx >> np.uint64(i)) & np.uint64(1)).astype(np.uint8), bitorder="little").view(np.uint64) if total >= 64
         else None for i in range(n)]
    ones = np.zeros(total, np.uint8)
    for i in range(n):
        ones += ((idx >> np.uint64(i)) & np.uint64(1)).astype(np.uint8)
    expect = np.packbits((ones > n // 2).astype(np.uint8), bitorder="little").view(np.uint64)
    for a, b in pairs:
        lo, hi = v[a] & v[b], v[a] | v[b]
        v[a], v[b] = lo, hi
    assert np.array_equal(v[out], expect)
