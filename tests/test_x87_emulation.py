"""The double-arithmetic emulation of the reference's x87 extended-precision expressions
(path_tracer_amd/csrc/pt_x87.hpp) against real long double arithmetic (oracle por_x87)."""
import numpy as np
import pytest

import path_tracer_amd as pt
from _oracle import _p


def both(port, kind, a, b=None, dev=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros_like(a)
    ref = np.zeros_like(a)
    assert pt.lib().pt_x87_eval(kind, dev, _p(a), _p(b), a.size, _p(out)) == 0
    port.lib.por_x87(kind, _p(a), _p(b), a.size, _p(ref))
    same = (out.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(out) & np.isnan(ref))
    return same, out, ref


def cases(n, rng):
    yield 0, np.exp(rng.uniform(-8, 8, n)), None
    yield 0, np.array([1.0, 2.0, 0.5, 3.0, 7.0, 1.0000000000000002, 0.9999999999999999, 1e-30, 1e30]), None
    yield 1, rng.integers(0, 2**31, n).astype(np.float64), None
    yield 1, np.array([0, 1, 2, 3, 2**31 - 1, 2**31 - 2, 2**30, 2**18 - 1, 2**18, 2**18 + 1], dtype=np.float64), None
    yield 2, np.exp(rng.uniform(-30, 8, n)), None
    yield 3, np.exp(rng.uniform(-30, 8, n)), None
    for eta in (1 / 1.5, 1.0, 0.9):
        yield 4, np.full(n, eta), rng.uniform(0, 1, n)
    yield 4, rng.uniform(0.3, 1.0, n), rng.uniform(0, 1, n) ** 4
    edge = np.concatenate([np.zeros(4), np.ones(4), np.full(4, 1e-9), np.full(4, 1e-5), rng.uniform(0, 1e-3, 500)])
    yield 4, np.full(edge.size, 1 / 1.5), edge
    yield 4, np.full(edge.size, 1.0), edge


def test_host_emulation_is_exact(port):
    rng = np.random.default_rng(1)
    for kind, a, b in cases(400_000, rng):
        same, out, ref = both(port, kind, a, b)
        assert same.all(), (kind, a[~same][:3], out[~same][:3], ref[~same][:3])


def test_2u_minus_1_needs_no_emulation(port):
    # exhaustive over every draw below 2^19: plain double u*2-1 equals the x87 result
    k = np.arange(0, 2**19, dtype=np.float64)
    ref = np.zeros_like(k)
    port.lib.por_x87(1, _p(k), _p(k), k.size, _p(ref))
    plain = (k.astype(np.int64) / 2147483647.0) * 2.0 - 1.0
    assert np.array_equal(plain.view(np.uint64), ref.view(np.uint64))


def test_double_rounding_cases_are_exercised(port):
    # the plain double expressions do differ from the reference's: the emulation is not vacuous
    rng = np.random.default_rng(2)
    a = np.exp(rng.uniform(-8, 8, 400_000))
    ref = np.zeros_like(a)
    port.lib.por_x87(0, _p(a), _p(a), a.size, _p(ref))
    assert 20 < ((1.0 / a) != ref).sum() < 400
    port.lib.por_x87(2, _p(a), _p(a), a.size, _p(ref))
    assert ((a * 1.0000001) != ref).mean() > 0.2


@pytest.mark.gpu
def test_device_emulation_is_exact(port):
    rng = np.random.default_rng(3)
    for kind, a, b in cases(1_000_000, rng):
        same, out, ref = both(port, kind, a, b, dev=1)
        assert same.all(), (kind, a[~same][:3], out[~same][:3], ref[~same][:3])
