"""The FOF restatement (oracle/fof.py) against the reference's own fixtures: both particle set-ups of
libgadget/tests/test_fof.cpp and every BOOST_TEST on their catalogues (one task)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fof as ofof  # noqa: E402
import fof_fixtures as fx  # noqa: E402


def _run(f):
    n = len(f["pos"])
    return ofof.fof(f["pos"], f["vel"], f["mass"], f["types"], f["ids"], np.zeros(n, dtype=bool), np.zeros(n), f["box"], f["linkl"], f["minlength"])


def test_fof_line_fixture():
    f = fx.line(numpart=128 * 128)        # the reference runs 512^2; the set-up scales (spacing / linking length = const * N^(-2/3))
    minid, groups, grnr = _run(f)
    fx.check_line(groups, grnr, f)
    assert (minid == 1).all()


def test_fof_halos_fixture():
    f = fx.halos()
    minid, groups, grnr = _run(f)
    fx.check_halos(groups, grnr, f)


def test_secondary_attaches_to_nearest_primary_and_gives_up_beyond_4_linking_lengths():
    rng = np.random.default_rng(1)
    box, linkl = 100.0, 1.0
    # one clump of DM, gas at increasing distances from it
    dm = 50 + 0.3 * rng.normal(size=(30, 3))
    gas = np.array([[50.2, 50, 50], [53.0, 50, 50], [57.5, 50, 50], [80.0, 50, 50]])
    pos = np.concatenate([dm, gas])
    types = np.array([1] * 30 + [0] * 4, dtype=np.uint8)
    ids = np.arange(100, 134, dtype=np.uint64)
    n = len(pos)
    minid, groups, grnr = ofof.fof(pos, np.zeros((n, 3)), np.ones(n), types, ids, np.zeros(n, dtype=bool), np.zeros(n), box, linkl, 5)
    assert (minid[:30] == 100).all()
    # search radii 0.4, 0.8, 1.6, 3.2, 6.4 (the first one >= 4 linking lengths is still searched): 0.2 and 3.0 and 6.x attach, 30 does not
    assert minid[30] == 100 and minid[31] == 100
    assert minid[32] == (100 if np.sqrt(((pos[:30] - pos[32]) ** 2).sum(1)).min() <= 6.4 else ids[32])
    assert minid[33] == ids[33]
    assert groups[0]["LenType"][0] == int((minid[30:] == 100).sum()) and groups[0]["Length"] == 30 + groups[0]["LenType"][0]
