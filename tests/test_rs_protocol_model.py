"""CPU: a randomized-schedule MODEL of the exchange protocol of the row-split persistent loop (csrc/dsd_loop_rs.hpp: data is the flag, three
slots per ring, reset one item ahead and drained before the next publish).  The GPU tests (tests/test_gpu_rs.py) show that the kernel
computes the right numbers on the schedules the hardware happened to produce; this model explores schedules it may not have produced.

Model.  Workgroups are sequential agents (tile, g) running the kernel's program for a few evaluations x layers over the rings x (l >= 1),
x0 (layer-0 input), gate, skip sum, head tile, next-x tile - the same owners, readers, item numbering and reset points as the kernel.  Memory
is weakly ordered the way write-through stores are: a store sits in its workgroup's pending list and lands at a random later scheduler
step, in any order relative to the other pending stores (same address: in program order); `drain` lands all of a workgroup's stores
(s_waitcnt vmcnt(0)).  A slice has two words that arrive independently.  A gather polls until no word of the slices it needs holds the
SENTINEL and then REQUIRES every word to be exactly (ring, item, owner): a stale word (an old item taken for the new one) or an early
read fails the run.  The scheduler picks a random runnable agent and lands random pending stores; agents may be arbitrarily slow.

Checked: the protocol as implemented passes thousands of random schedules for several geometries; each of four mutations of it - two slots
instead of three, the reset issued without the drain before the publish, the x reset issued before the item was staged, the gate reset issued
before the gather - is CAUGHT (a stale or early read, or a reader stuck on a word that was cleared under it)."""
import random

import pytest

SENT = None


class Mem:
    def __init__(self, rng):
        self.cells = {}
        self.pending = {}           # agent -> list of [key, value, lands_at]
        self.rng = rng
        self.now = 0

    def store(self, agent, key, value):
        # most stores land within a few scheduler steps, a few take very long (a write-through store has no latency bound but the drain)
        r = self.rng.random()
        delay = self.rng.randint(1, 6) if r < 0.9 else self.rng.randint(6, 200) if r < 0.98 else self.rng.randint(200, 4_000)
        self.pending.setdefault(agent, []).append([key, value, self.now + delay])

    def land_some(self):
        self.now += 1
        for agent, lst in self.pending.items():
            i = 0
            while i < len(lst):
                key, value, at = lst[i]
                first_for_key = all(lst[j][0] != key for j in range(i))     # same address: program order
                if first_for_key and at <= self.now:
                    self.cells[key] = value
                    lst.pop(i)
                else:
                    i += 1

    def drain(self, agent):
        for key, value, _ in self.pending.pop(agent, []):
            self.cells[key] = value

    def load(self, key):
        return self.cells.get(key, SENT)


class Stale(Exception):
    pass


def agent_program(mem, me, tile, g, ntiles, G, L, n_evals, mut):
    """Generator: yields whenever the agent could be pre-empted (after every memory operation / poll)."""
    slots = 2 if mut == 'two_slots' else 3
    res_owner = g < max(1, G // 2)                  # G = 16: half of the workgroups finish residual rows, the others skip rows
    skip_owner = (not res_owner) or G <= 2
    head_owner = g < max(1, G // 2)                 # rows of head a / c
    out_owner = g == 0                              # rows of head b
    nbrs = [t for t in (tile - 1, tile + 1) if 0 <= t < ntiles]

    def put(ring, item, t, owner, value):
        for w in range(2):
            mem.store(me, (ring, item % slots, t, owner, w), value)

    def gather(ring, item, sources):
        """sources: list of (tile, owner).  Poll until nothing is the sentinel, then every word must be the expected item."""
        while True:
            vals = [(s, mem.load((ring, item % slots, s[0], s[1], w))) for s in sources for w in range(2)]
            yield
            if all(v is not SENT for _, v in vals):
                for s, v in vals:
                    if v != (ring, item, s[1]):
                        raise Stale(f'{me} read {v} for {(ring, item, s)}')
                return

    x_owners = lambda t: [(t, o) for o in range(G) if o < max(1, G // 2)]
    all_of = lambda t: [(t, o) for o in range(G)]
    for e in range(n_evals):
        for l in range(L):
            last = l == L - 1
            # ---- conv phase: stage x_l (own tile + both neighbours)
            ring, item = ('x0', e) if l == 0 else ('x', e * (L - 1) + l - 1)
            owners = (lambda t: [(t, o) for o in range(G) if o < max(1, G // 2)]) if True else None
            srcs = owners(tile) + [s for t in nbrs for s in owners(t)]
            if mut == 'x_reset_early' and l >= 1 and res_owner:
                put('x', item + 2, tile, g, SENT)
                yield
            yield from gather(ring, item, srcs)
            if l >= 1 and res_owner and mut != 'x_reset_early':
                put('x', item + 2, tile, g, SENT)                       # obligation (a): everybody is done with item - 1
                yield
            if l == 1:
                if head_owner:
                    put('x0', e + 2, tile, g, SENT); put('h', e + 2, tile, g, SENT)
                if skip_owner:
                    put('s', e + 2, tile, g, SENT)
                if out_owner:
                    put('p', e + 2, tile, g, SENT)
                yield
            # ---- gate: drain, publish, gather, reset
            if mut != 'no_drain':
                mem.drain(me)
            put('g', e * L + l, tile, g, ('g', e * L + l, g))
            yield
            if mut == 'gate_reset_early':
                put('g', e * L + l + 2, tile, g, SENT)
                yield
            yield from gather('g', e * L + l, all_of(tile))
            if mut != 'gate_reset_early':
                put('g', e * L + l + 2, tile, g, SENT)
                yield
            # ---- out phase: drain, publish x'
            if mut != 'no_drain':
                mem.drain(me)
            if res_owner and not last:
                put('x', e * (L - 1) + l, tile, g, ('x', e * (L - 1) + l, g))
                yield
        # ---- head: skip sum -> a -> b -> c (next evaluation's x0)
        if skip_owner:
            mem.drain(me)
            put('s', e, tile, g, ('s', e, g))
            yield
        if head_owner:
            yield from gather('s', e, [(tile, o) for o in range(G) if (not o < max(1, G // 2)) or G <= 2])
            mem.drain(me)
            put('h', e, tile, g, ('h', e, g))
            yield
        if out_owner:
            yield from gather('h', e, x_owners(tile))
            mem.drain(me)
            if e + 1 < n_evals:
                put('p', e, tile, g, ('p', e, g))
            yield
        if e + 1 < n_evals and head_owner:
            yield from gather('p', e, [(tile, 0)])
            mem.drain(me)
            put('x0', e + 1, tile, g, ('x0', e + 1, g))
            yield


def run(seed, ntiles, G, L, n_evals, mut=None, max_steps=2_000_000):
    rng = random.Random(seed)
    mem = Mem(rng)
    agents = {}
    for t in range(ntiles):
        for g in range(G):
            me = (t, g)
            agents[me] = agent_program(mem, me, t, g, ntiles, G, L, n_evals, mut)
            if g < max(1, G // 2):                  # evaluation 0's input projection comes from global memory: published up front
                for w in range(2):
                    mem.cells[('x0', 0, t, g, w)] = ('x0', 0, g)
    # a few agents are persistently slow (uneven load)
    slow = {a for a in agents if rng.random() < 0.3}
    steps = 0
    while agents:
        steps += 1
        if steps > max_steps:
            return 'stuck'
        mem.land_some()
        a = rng.choice(list(agents))
        if a in slow and rng.random() < 0.8:
            continue
        try:
            next(agents[a])
        except StopIteration:
            mem.drain(a)
            del agents[a]
        except Stale as ex:
            return f'stale: {ex}'
    return 'ok'


@pytest.mark.parametrize('ntiles,G,L,n_evals', [(1, 2, 3, 3), (2, 2, 3, 4), (3, 4, 2, 4), (2, 4, 4, 3), (3, 2, 5, 3)])
def test_the_protocol_survives_random_schedules(ntiles, G, L, n_evals):
    for seed in range(80):
        assert run(seed, ntiles, G, L, n_evals) == 'ok', (seed, ntiles, G, L, n_evals)


@pytest.mark.parametrize('mut', ['two_slots', 'no_drain', 'x_reset_early', 'gate_reset_early'])
def test_every_mutation_of_the_protocol_is_caught(mut):
    """The model has teeth: each way of weakening the protocol produces a stale / early read or a stuck reader on some schedule."""
    for seed in range(400):
        if run(seed, 3, 4, 4, 4, mut=mut, max_steps=150_000) != 'ok':
            return
    raise AssertionError(f'mutation {mut} survived 400 random schedules: the model would not catch it')
