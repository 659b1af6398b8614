"""The top tree handed to shq_tree_build_domain as geometry against the reference's own key arithmetic.

oracle/_ref/libpeano_ref.so is libgadget/utils/peano.cpp compiled where it lies (`make -C oracle ref`).  It must reproduce the golden
keys of the reference's tests/test_peano.cpp (tests/golden/peano_keys.json); with it, a top tree cut in Peano-Hilbert key space the way
domain.cpp does (TopNodes with StartKey / Shift / Daughter / Leaf) is turned into the daughter-per-octant table by the loop of
INTEGRATION.md (force_create_node_for_topnode's `sub`, forcetree.cpp:881), and for random positions the top leaf found by geometric
descent through that table (what the device build does) must be domain_get_topleaf(PEANO(pos)) (domain.h:68-76)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import common as cm
from shenqi_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpeano_ref.so")
BITS = 21


@pytest.fixture(scope="module")
def peano():
    if not os.path.exists(LIB):
        if os.path.isdir("/root/reference/libgadget"):
            pytest.fail("oracle/_ref/libpeano_ref.so missing although /root/reference is present: run `make -C oracle ref`")
        pytest.skip("no reference and no prebuilt peano library")
    lib = C.CDLL(LIB)
    lib.ref_peano_hilbert_key.restype = C.c_uint64
    lib.ref_peano_hilbert_key.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.ref_PEANO.restype = C.c_uint64
    lib.ref_PEANO.argtypes = [C.c_void_p, C.c_double]
    return lib


def test_reference_key_reproduces_its_golden_values(peano):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "peano_keys.json")))["keys"]
    for i, g in enumerate(gold):
        p = np.array([i % 4, (i // 4) % 4, (i // 16) % 4], dtype=np.float64)
        assert peano.ref_PEANO(p.ctypes.data, 4.0) == g


def key_space_topnodes(rng, maxdepth=3, psplit=0.7):
    """TopNodes the way domain.cpp refines them: node = key range [StartKey, StartKey + 2^Shift), daughters = its eight consecutive
    sub-ranges (Daughter + k covers StartKey + k 2^(Shift-3)), leaves numbered in key order"""
    nodes = [dict(StartKey=0, Shift=3 * BITS, Daughter=-1, Leaf=-1, depth=0)]
    queue = [0]
    while queue:
        t = queue.pop(0)
        nd = nodes[t]
        if nd["depth"] == 0 or (nd["depth"] < maxdepth and rng.random() < psplit):
            nd["Daughter"] = len(nodes)
            for k in range(8):
                nodes.append(dict(StartKey=nd["StartKey"] + (k << (nd["Shift"] - 3)), Shift=nd["Shift"] - 3, Daughter=-1, Leaf=-1, depth=nd["depth"] + 1))
            queue.extend(range(nd["Daughter"], nd["Daughter"] + 8))
    leaves = sorted((nd["StartKey"], t) for t, nd in enumerate(nodes) if nd["Daughter"] < 0)
    for l, (_, t) in enumerate(leaves):
        nodes[t]["Leaf"] = l
    return nodes


def geo_from_topnodes(nodes, peano):
    """the fill loop of INTEGRATION.md (force_create_node_for_topnode, forcetree.cpp:868-930)"""
    geo = np.zeros(len(nodes), dtype=capi.TOPNODE_GEO_DTYPE)
    geo["daughter"] = -1
    geo["leaf"] = -1

    def fill(topnode, bits, x, y, z):
        d = nodes[topnode]["Daughter"]
        geo["leaf"][topnode] = nodes[topnode]["Leaf"] if d < 0 else -1
        if d < 0:
            return
        for i in range(2):
            for j in range(2):
                for k in range(2):
                    count = i + 2 * j + 4 * k
                    sub = 7 & peano.ref_peano_hilbert_key((x << 1) + i, (y << 1) + j, (z << 1) + k, bits)
                    geo["daughter"][topnode][count] = d + sub
                    fill(d + sub, bits + 1, 2 * x + i, 2 * y + j, 2 * z + k)

    fill(0, 1, 0, 0, 0)
    return geo


def domain_get_topleaf(key, nodes):
    no = 0
    while nodes[no]["Daughter"] >= 0:
        no = nodes[no]["Daughter"] + ((key - nodes[no]["StartKey"]) >> (nodes[no]["Shift"] - 3))
    return nodes[no]["Leaf"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_geometric_descent_finds_the_reference_topleaf(peano, seed):
    rng = np.random.default_rng(seed)
    nodes = key_space_topnodes(rng)
    geo = geo_from_topnodes(nodes, peano)
    assert sorted(int(x) for x in geo["leaf"] if x >= 0) == list(range(sum(1 for nd in nodes if nd["Daughter"] < 0)))
    box = cm.BOX
    pos = rng.random((4000, 3)) * box
    got = cm.topleaf_of(pos, geo, box)
    want = np.array([domain_get_topleaf(peano.ref_PEANO(np.ascontiguousarray(p).ctypes.data, box), nodes) for p in pos])
    assert np.array_equal(got, want)
    assert len(np.unique(want)) > 8
