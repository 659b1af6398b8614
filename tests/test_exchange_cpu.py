"""The restatement of the particle exchange (oracle/exchange.py) against the four cases of the reference's
libgadget/tests/test_exchange.cpp, with 1, 2, 3 and 5 tasks emulated in one process."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import exchange as oex  # noqa: E402
import exchange_fixtures as fx  # noqa: E402

CASES = {"all": [8] * 6, "zero_slots": [8, 0, 8, 0, 8, 0]}


def run(ntask, ntype, layout, garbage=False):
    tasks = [list(fx.setup_task(r, ntask, ntype)) for r in range(ntask)]
    tot = ntask * sum(ntype)
    if garbage:
        for T in tasks:                      # slots_mark_garbage(0, ...), test_exchange.cpp:136
            T[0]["Flags"][0] |= 1
            T[2][0]["ReverseLink"][T[0]["PI"][0]] = len(T[0]) + 100
        tot -= ntask
    objs = [oex.Task(T[0], T[1], T[2], T[3]) for T in tasks]
    targets = [layout(T[0], T[1], ntask) for T in tasks]
    oex.domain_exchange(objs, targets)
    return [(o.parts, o.numpart, o.slots, o.slot_size) for o in objs], tot


@pytest.mark.parametrize("ntask", [1, 2, 3, 5])
@pytest.mark.parametrize("case", ["all", "zero_slots"])
def test_exchange(ntask, case):
    tasks, tot = run(ntask, CASES[case], fx.layout_id_mod)
    fx.check_after(tasks, ntask, tot)


@pytest.mark.parametrize("ntask", [2, 4])
def test_exchange_with_garbage(ntask):
    tasks, tot = run(ntask, CASES["all"], fx.layout_id_mod, garbage=True)
    fx.check_after(tasks, ntask, tot)


@pytest.mark.parametrize("ntask", [2, 4])
def test_exchange_uneven(ntask):
    tasks, tot = run(ntask, CASES["all"], fx.layout_uneven)
    fx.check_after(tasks, ntask, tot, uneven=True)
    assert tasks[0][3][0] == fx.NUMPART1 * ntask       # the gas slots of task 0 grew to hold everybody's gas (test_exchange.cpp:179)
