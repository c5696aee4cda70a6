"""A model of the direct transport's flag protocol (csrc/comm_direct.hip) under arbitrary interleavings — no GPU needed.

Round 4 found two races on hardware (eight processes time-slicing one GPU): a workgroup overwrote box space that workgroups OTHER
than its partner were still reading, because the per-workgroup flags prove nothing about the neighbours' ranges once two calls
differ in length (the ranges of workgroup b are cut from the call's length). The kernels now wait, at the start of every call of the
all-reduce family, until every workgroup of every peer has published the previous call, and a send waits for the credit of ALL of
the receiver's workgroups. This test restates both protocols as little state machines — one per (rank, workgroup), every remote
store / flag store / poll a separate step — runs them under a seeded random scheduler with call lengths that change from call to
call, and checks the property the hardware test can only sample: a reader never sees a unit written by a call other than its own.
The FIRST version of the protocol (per-partner flags only) is modelled too and must be caught, so the checker is known to bite."""
import random

import pytest

G = 4  # workgroups per launch (32 on the device; the argument does not depend on the number)


def elem_range(length, b):  # csrc/comm_direct.hip: elem_range<char> (16-byte units)
    units = (length + 15) // 16
    e0 = min(units * b // G * 16, length)
    e1 = min(units * (b + 1) // G * 16, length)
    return e0, e1


class World:
    def __init__(self, n):
        self.n = n
        self.flag = [[[0] * G for _ in range(n)] for _ in range(n)]   # flag[owner][src][wg]  (arrival words in owner's ctrl)
        self.ack = [[[0] * G for _ in range(n)] for _ in range(n)]    # ack[owner = sender][receiver][wg]
        self.inbox = [[[{} for _ in range(n)] for _ in range(2)] for _ in range(n)]  # inbox[owner][parity][src][unit] = call id
        self.pbox = [[{} for _ in range(n)] for _ in range(n)]        # pbox[owner][src][unit] = message id
        self.violations = []


def reduce_family(w, r, b, lengths, wait_all):
    """Workgroup b of rank r: calls 1 .. len(lengths); call s pushes range b of a `lengths[s-1]`-byte slice into every peer's
    inbox[s & 1], publishes, waits for its partners, reads its own range of every peer's contribution."""
    n = w.n
    for s, length in enumerate(lengths, start=1):
        par = s & 1
        if wait_all:  # wait_all_workgroups(flagS, s - 1): every workgroup of every peer has left call s - 2
            for p in range(n):
                if p != r:
                    for wb in range(G):
                        while w.flag[r][p][wb] < s - 1:
                            yield
        e0, e1 = elem_range(length, b)
        for u in range(e0, e1, 16):  # phase 1: one remote store per unit and peer (each a scheduling point)
            for p in range(n):
                if p != r:
                    w.inbox[p][par][r][u] = s
                    yield
        for p in range(n):  # publish (after release_stores + barrier: all of the workgroup's stores are performed)
            if p != r:
                w.flag[p][r][b] = s
                yield
        for p in range(n):  # wait for the partner workgroups
            if p != r:
                while w.flag[r][p][b] < s:
                    yield
        for u in range(e0, e1, 16):  # phase 2: read my range of every peer's slice
            for p in range(n):
                if p != r:
                    got = w.inbox[r][par][p].get(u)
                    if got != s:
                        w.violations.append(("reduce", r, b, s, p, u, got))
                    yield


def sender(w, r, b, peer, lengths, credit_all):
    for s, length in enumerate(lengths, start=1):
        if credit_all:  # wait_all_workgroups(ackP, s - 1, peer)
            for wb in range(G):
                while w.ack[r][peer][wb] < s - 1:
                    yield
        else:
            while w.ack[r][peer][b] < s - 1:
                yield
        e0, e1 = elem_range(length, b)
        for u in range(e0, e1, 16):
            w.pbox[peer][r][u] = s
            yield
        w.flag[peer][r][b] = s
        yield


def receiver(w, r, b, peer, lengths):
    for s, length in enumerate(lengths, start=1):
        while w.flag[r][peer][b] < s:
            yield
        e0, e1 = elem_range(length, b)
        for u in range(e0, e1, 16):
            got = w.pbox[r][peer].get(u)
            if got != s:
                w.violations.append(("recv", r, b, s, u, got))
            yield
        w.ack[peer][r][b] = s  # credit into the SENDER's ctrl
        yield


def run(procs, rng, limit=400_000):
    live = list(procs)
    steps = 0
    while live:
        g = rng.choice(live) if rng.random() < 0.7 else live[0]  # (biased: lets one workgroup lag far behind)
        try:
            next(g)
        except StopIteration:
            live.remove(g)
        steps += 1
        assert steps < limit, "the model dead-locked (or live-locked)"


def lengths_for(rng, calls):
    return [rng.choice([4000, 4004, 4016, 8000, 1 << 12, 3 << 11, 100]) for _ in range(calls)]


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_family_never_reads_another_calls_data(world):
    for seed in range(60):
        rng = random.Random(seed)
        lengths = lengths_for(rng, 6)
        w = World(world)
        run([reduce_family(w, r, b, lengths, True) for r in range(world) for b in range(G)], rng)
        assert not w.violations, (seed, lengths, w.violations[:3])


def test_the_first_version_of_the_reduce_protocol_is_caught():
    """Per-partner flags only (round 4's first version): some interleaving lets a fast workgroup overwrite a range a slow neighbour
    of the peer still reads — the checker must find it, or it proves nothing about the fixed protocol."""
    caught = 0
    for seed in range(200):
        rng = random.Random(seed)
        w = World(2)
        run([reduce_family(w, r, b, lengths_for(rng, 6), False) for r in range(2) for b in range(G)], rng)
        caught += bool(w.violations)
    assert caught > 0


def test_send_recv_credits_cover_every_workgroup():
    for seed in range(100):
        rng = random.Random(seed)
        lengths = lengths_for(rng, 6)
        w = World(2)
        procs = [sender(w, 0, b, 1, lengths, True) for b in range(G)] + [receiver(w, 1, b, 0, lengths) for b in range(G)]
        run(procs, rng)
        assert not w.violations, (seed, lengths, w.violations[:3])


def test_per_partner_credits_are_caught():
    """4 000 -> 4 004 bytes is 250 -> 251 sixteen-byte units: the very pair of lengths of the hardware failure."""
    caught = 0
    for seed in range(300):
        rng = random.Random(seed)
        lengths = [4000, 4004, 4008, 4012, 8000, 100]
        w = World(2)
        procs = [sender(w, 0, b, 1, lengths, False) for b in range(G)] + [receiver(w, 1, b, 0, lengths) for b in range(G)]
        run(procs, rng)
        caught += bool(w.violations)
    assert caught > 0
