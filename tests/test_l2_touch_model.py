"""CPU: the turn-taking of the L2 touch (csrc/dsd_loop_split.hpp, L2Touch::next / SplitPipeR::touch / SplitPipeF::touch) restated in Python.
Wave q of an XCD's nwx waves (q = 4 * workgroup-in-XCD + wave) keeps r = (q - per * g) mod nwx as a running counter, g the chunk `ahead`
steps in front of the one being multiplied: at every step it reads t = r, fetches 64 lines at byte offset (g mod gtot) * chunk_bytes +
t * 8192 if t < per, and advances r by -per (one wrap).  Checked for both streams (planes: 48 KiB chunks, per = 6; pair format: 32 KiB,
per = 4) and several launch sizes (32, 19, 10, 2 workgroups per XCD): over two whole evaluations of L layers every line of every chunk is
fetched by EXACTLY one wave per evaluation, `ahead` chunks before it is multiplied, inside the buffer - across the layer boundaries and the
wrap from the last layer to layer 0 of the next evaluation - and the work is spread evenly."""
import pytest


def touches(q, nwx, L, per, chunk_bytes, ahead, n_evals=2):
    """[(evaluation, step, chunk fetched, byte offset)] of wave q; a step = a chunk being multiplied, in stream order."""
    gtot = 64 * L
    r = (q - per * ahead) % nwx
    out = []
    for e in range(n_evals):
        for l in range(L):
            for gbase, n in ((64 * l, 48), (64 * l + 48, 16)):       # pipe1: the conv's 48 chunks, pipe2: the out-projection's 16
                for kc in range(n):
                    t = r
                    r -= per
                    if r < 0:
                        r += nwx
                    if t < per:
                        g = gbase + kc + ahead
                        if g >= gtot:
                            g -= gtot
                        out.append((e, gbase + kc, g, g * chunk_bytes + t * 8192))
    return out


@pytest.mark.parametrize('per,chunk_bytes', [(6, 4 * 12288), (4, 4 * 8192)], ids=['planes', 'pair'])
@pytest.mark.parametrize('ahead', [4, 8, 16])
@pytest.mark.parametrize('L,nwx', [(20, 128), (20, 76), (20, 40), (2, 8), (3, 128)])
def test_every_line_is_fetched_once_per_evaluation_and_ahead_of_its_use(per, chunk_bytes, ahead, L, nwx):
    assert chunk_bytes == per * 8192 and per <= nwx
    gtot = 64 * L
    seen = {}
    counts = []
    for q in range(nwx):
        mine = touches(q, nwx, L, per, chunk_bytes, ahead)
        counts.append(len(mine))
        for e, step, g, off in mine:
            assert 0 <= off and off + 64 * 128 <= gtot * chunk_bytes     # inside the buffer: 64 lanes x 128-byte lines
            assert (step + ahead) % gtot == g                            # `ahead` chunks in front of the chunk being multiplied
            e_of_chunk = e + (1 if step + ahead >= gtot else 0)          # the chunk belongs to the evaluation it is multiplied in
            for lane in range(64):
                key = (e_of_chunk, off // 128 + lane)
                assert key not in seen, (q, key, seen.get(key))
                seen[key] = q
    lines = gtot * chunk_bytes // 128
    assert sum(1 for k in seen if k[0] == 1) == lines                    # the second evaluation: every line of every chunk, fetched during
    assert max(counts) - min(counts) <= 1                                # ... the first one's tail or itself; the work is spread evenly
