"""CPU: the turn-taking of the L2 touch (csrc/dsd_loop_split.hpp, L2Touch / SplitPipeR::touch / SplitPipeF::touch) restated in Python.
In the kernel wave q of an XCD (q = 4 * workgroup-in-XCD + wave, 0 .. 127) executes, at chunk kc of the pipe whose chunk 0 has global index
gbase, one 64-line fetch iff t = (q - per * g) & 127 < per, g = gbase + kc + ahead, at byte offset (g mod gtot) * chunk_bytes + t * 8192.
Checked here for both streams (planes: 48 KiB chunks, per = 6; pair format: 32 KiB, per = 4): over one whole evaluation of L layers every
line of every chunk is fetched by EXACTLY one wave, `ahead` chunks before it is multiplied, inside the buffer - including across the layer
boundaries and the wrap from the last layer to layer 0 of the next evaluation."""
import pytest


def touches(q, L, per, chunk_bytes, ahead):
    """(step at which it is issued, byte offset) for every fetch wave q issues during one evaluation; a step = a chunk being multiplied."""
    gtot = 64 * L
    out = []
    for l in range(L):
        for gbase, n in ((64 * l, 48), (64 * l + 48, 16)):           # pipe1: the conv's 48 chunks, pipe2: the out-projection's 16
            for kc in range(n):
                g = gbase + kc + ahead
                t = (q - per * g) & 127
                if t < per:
                    if g >= gtot:
                        g -= gtot
                    out.append((gbase + kc, g, g * chunk_bytes + t * 8192))
    return out


@pytest.mark.parametrize('per,chunk_bytes', [(6, 4 * 12288), (4, 4 * 8192)], ids=['planes', 'pair'])
@pytest.mark.parametrize('ahead', [4, 8, 16])
@pytest.mark.parametrize('L', [20, 2])
def test_every_line_is_fetched_once_and_ahead_of_its_use(per, chunk_bytes, ahead, L):
    assert chunk_bytes == per * 8192 and (per * 64 * L) % 128 == 0      # the wrap must not move the turn
    gtot = 64 * L
    seen = {}
    for q in range(128):
        for step, g, off in touches(q, L, per, chunk_bytes, ahead):
            assert 0 <= off and off + 64 * 128 <= gtot * chunk_bytes     # inside the buffer: 64 lanes x 128-byte lines
            assert (step + ahead) % gtot == g                            # `ahead` chunks in front of the chunk being multiplied
            for lane in range(64):
                line = off // 128 + lane
                assert line not in seen, (q, g, line, seen.get(line))
                seen[line] = q
    assert len(seen) == gtot * chunk_bytes // 128                        # every line of every chunk of every layer
    per_wave = [len(touches(q, L, per, chunk_bytes, ahead)) for q in range(128)]
    assert max(per_wave) - min(per_wave) <= 1                            # the work is spread evenly: per * gtot / 128 fetches per wave
