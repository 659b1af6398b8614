"""Comparison of a downloaded device tree (NODE array, pre-order from `firstnode`) with the tree of oracle/toptree_build.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import toptree_build as tb  # noqa: E402


def geo_list(geo):
    return [([int(x) for x in geo["daughter"][t]], int(geo["leaf"][t])) for t in range(len(geo))]


def compare(nodes, firstnode, lastnode_dev, t, lastnode_orc, moments=True):
    """every node the threaded walk of the oracle tree reaches against the downloaded node of the same pre-order number:
    geometry, type, flags, particle lists, links (through the numbering), and — moments=True — mass / cofm / hmax to the bit"""
    number = {}
    order = []
    for no, nd in tb.preorder(t):
        number[no] = firstnode + len(order)
        order.append((no, nd))
    assert len(order) == len(nodes), (len(order), len(nodes))
    for k, (no, nd) in enumerate(order):
        g = nodes[k]
        assert g["len"] == nd.len and tuple(g["center"]) == tuple(nd.center), k
        ctype = (int(g["flags"]) >> 3) & 3
        assert ctype == nd.ChildType, (k, ctype, nd.ChildType)
        assert (int(g["flags"]) >> 1) & 1 == nd.TopLevel and int(g["flags"]) & 1 == nd.InternalTopLevel, k
        assert int(g["sibling"]) == (number[nd.sibling] if nd.sibling >= 0 else -1), k
        assert int(g["father"]) == (number[nd.father] if nd.father >= 0 else -1), k
        if nd.ChildType == tb.PARTICLE:
            assert int(g["noccupied"]) == nd.nocc
            assert list(g["suns"][:nd.nocc]) == nd.suns[:nd.nocc], k
        elif nd.ChildType == tb.PSEUDO:
            assert int(g["suns"][0]) - lastnode_dev == nd.suns[0] - lastnode_orc, k
        else:
            kids = [s for s in nd.suns if s >= 0]
            assert [int(x) for x in g["suns"][:len(kids)]] == [number[s] for s in kids], k
            assert all(int(x) == -1 for x in g["suns"][len(kids):]), k
        if moments:
            assert g["mass"] == nd.mass and tuple(g["cofm"]) == tuple(nd.cofm) and g["hmax"] == nd.hmax, (k, nd.ChildType, g["mass"], nd.mass, g["cofm"], nd.cofm)
    return number
