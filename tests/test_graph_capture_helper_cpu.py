"""graph_capture (the package's wrapper of torch.cuda.graph): garbage is collected before the capture opens and Python's cyclic
collector is off until it has closed — a library object finalised inside a capture frees device memory, which the capture's global
mode forbids (the process aborts: seen once in a GPU suite, profiles/r05/gpu_suite_mid.log).  Checked here with a stand-in for
torch.cuda.graph; the real captures are tests/test_gpu_graph_capture.py's."""
import gc

import b3w_testlib as T


def test_collector_is_off_inside_and_back_on_after(monkeypatch):
    import torch
    m = T.pkg()
    seen = []

    class Standin:
        def __init__(self, graph, stream=None, **kw):
            seen.append(("init", graph, stream, kw))

        def __enter__(self):
            seen.append(("enter", gc.isenabled()))
            return "inner"

        def __exit__(self, *exc):
            seen.append(("exit", gc.isenabled(), exc[0]))
            return False

    monkeypatch.setattr(torch.cuda, "graph", Standin)

    class Cycle:
        def __init__(self):
            self.me = self
    dropped = []
    import weakref
    c = Cycle()
    w = weakref.ref(c, lambda _: dropped.append(1))
    del c                                                      # cyclic garbage from "an earlier test"
    assert gc.isenabled()
    with m.graph_capture("g", stream="s", pool=7) as inner:
        assert inner == "inner" and not gc.isenabled() and dropped == [1] and w() is None      # collected BEFORE the capture opened
    assert gc.isenabled()
    assert seen == [("init", "g", "s", {"pool": 7}), ("enter", False), ("exit", False, None)]
    try:
        with m.graph_capture("g"):
            raise ValueError("inside")
    except ValueError:
        pass
    assert gc.isenabled() and seen[-1][:2] == ("exit", False) and seen[-1][2] is ValueError
    gc.disable()                                               # a caller who had it off keeps it off
    try:
        with m.graph_capture("g"):
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
