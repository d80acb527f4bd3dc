"""CPU: the Philox4x32-10 restatement (oracle/philox_oracle.py) against the known-answer vectors published with the generator
(Random123 kat_vectors, philox4x32 10 rounds), and the shape of the derived N(0,1) stream."""
import numpy as np

from oracle import philox_oracle as P


def _kat(ctr, key):
    out = P.philox4x32_10([np.array([c], dtype=np.uint32) for c in ctr], key)
    return [int(v[0]) for v in out]


def test_philox_known_answers():
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_normal_stream_statistics_and_determinism():
    z = P.philox_normal(1234, 7, 200000)
    assert np.array_equal(z, P.philox_normal(1234, 7, 200000))
    assert not np.array_equal(z[:1000], P.philox_normal(1234, 8, 1000))
    assert not np.array_equal(z[:1000], P.philox_normal(1235, 7, 1000))
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 3).mean())) < 0.03 and abs(float((z ** 4).mean()) - 3.0) < 0.06
    assert np.isfinite(z).all() and float(np.abs(z).max()) < 6.0
