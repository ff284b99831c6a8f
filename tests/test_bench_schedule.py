"""bench.py's pass scheduler (run_passes) and step accounting, without a GPU: every pass of the sequence is started exactly
once and finished in order, never more than max_in_flight are enqueued, and a context is never re-used while its previous
pass is still in flight; K steps are dealt into balanced batches (plan_batches)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


class FakeSlot:
    def __init__(self, name):
        self.name = name


def simulate(sequence, max_in_flight, monkeypatch):
    log, pending = [], []

    def start(g, prompt, n_greedy):
        assert g not in pending, "context re-used while its pass is in flight"
        pending.append(g)
        assert len(pending) <= max_in_flight
        log.append(("start", g.name))

    def finish(g):
        assert pending and pending[0] is g, "results must come back in order"
        pending.pop(0)
        log.append(("finish", g.name))
        return g.name

    monkeypatch.setattr(bench, "clip_start", start)
    monkeypatch.setattr(bench, "clip_finish", finish)
    last = bench.run_passes(sequence, [1, 2, 3], 51, max_in_flight)
    assert not pending
    return log, last


def test_run_passes_orders_and_bounds(monkeypatch):
    slots = [FakeSlot("a"), FakeSlot("b"), FakeSlot("c")]
    rem = FakeSlot("rem")
    for n_full in (0, 1, 2, 3, 7):
        for with_rem in (False, True):
            seq = [slots[i % 3] for i in range(n_full)] + ([rem] if with_rem else [])
            if not seq:
                continue
            for inflight in (1, 2, 3):
                log, last = simulate(seq, inflight, monkeypatch)
                assert [n for k, n in log if k == "start"] == [s.name for s in seq]
                assert [n for k, n in log if k == "finish"] == [s.name for s in seq]
                assert last == seq[-1].name
                if inflight == 1:
                    assert log == [x for s in seq for x in (("start", s.name), ("finish", s.name))]


def test_same_slot_back_to_back_waits(monkeypatch):
    s = FakeSlot("only")
    log, _ = simulate([s, s, s], 3, monkeypatch)
    assert log == [("start", "only"), ("finish", "only")] * 3


def test_step_accounting():
    for steps in range(1, 30):
        for c in (1, 2, 3, 4):
            n_full, rem = divmod(steps, c)
            assert n_full * c + rem == steps and 0 <= rem < c


def test_plan_batches_is_balanced_and_complete():
    """Every clip pass lands in exactly one batch, no batch exceeds C clips, sizes differ by at most one, and the number of
    batches is a multiple of the contexts in flight whenever there are enough passes (no context idles through the tail):
    the driver's 20 passes become 10 + 10, not 16 + 4 (measured: 7391 vs 7172 audio-s/s, profiles/r03_ab_variants.txt)."""
    assert bench.plan_batches(20, 16, 2) == [10, 10]
    assert bench.plan_batches(32, 16, 2) == [16, 16]
    assert bench.plan_batches(64, 16, 2) == [16, 16, 16, 16]
    assert bench.plan_batches(1, 16, 2) == [1]
    assert bench.plan_batches(0, 16, 2) == []
    for steps in range(1, 80):
        for C in (1, 4, 16, 18):
            for inflight in (1, 2, 3):
                sizes = bench.plan_batches(steps, C, inflight)
                assert sum(sizes) == steps and all(0 < n <= C for n in sizes)
                assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
                if steps >= inflight:
                    assert len(sizes) % inflight == 0 or len(sizes) == steps      # (one clip per batch: nothing left to round up with)
                assert len(sizes) <= -(-steps // C) + inflight - 1
