"""
CPU tests of DeviceCov's aliasing rules (ipp-rl_amd/_device_array.py) on a stand-in engine: a write must never land in
the cached host copy while the device copy stays the live one (np.asarray(state)[...] = 0 followed by a map
operation used the stale device matrix).
"""
import numpy as np
import pytest

from ipp_rl_amd._device_array import DeviceCov, SlotStore
from ipp_rl_amd import _runtime


class _Tensor:
    def __init__(self, a): self.a = a
    def detach(self): return self
    def numel(self): return self.a.size
    def double(self): return _Tensor(self.a.astype(np.float64))
    def cpu(self): return self
    def numpy(self): return self.a


class FakeEngine:
    """Dense slots on the host: what read_cov / write_cov / read_diag of IPPEngine do, without a GPU."""

    def __init__(self, n, slots):
        self.slots = {s: np.zeros((n, n), dtype=np.float32) for s in range(slots)}
        self.writes = 0

    def read_cov(self, slot): return _Tensor(self.slots[slot].copy())
    def read_diag(self, slot): return _Tensor(np.diag(self.slots[slot]).copy())
    def write_cov(self, slot, host):
        self.slots[slot] = np.asarray(host, dtype=np.float32).copy()
        self.writes += 1


def make(n=6, slots=4):
    eng = FakeEngine(n, slots + 2)
    store = SlotStore(eng, 2, slots)
    rs = np.random.RandomState(0)
    a = rs.uniform(size=(n, n))
    p = DeviceCov(store, n, host=(a + a.T).copy())
    slot = p.device_slot(store)  # uploaded: the device copy is live, the host copy a cache
    assert slot is not None and eng.writes == 1
    return eng, store, p


def device_matrix(eng, store, p):
    """What the next map operation would read."""
    return eng.slots[p.device_slot(store)].astype(np.float64)


def test_views_that_escape_an_attached_state_are_read_only():
    eng, store, p = make()
    for view in (np.asarray(p), p[0], p[1:3], p.T, p.ravel(), next(iter(p))):
        assert isinstance(view, np.ndarray) and not view.flags.writeable
        with pytest.raises(ValueError):
            view[...] = 0.0
    assert np.asarray(p, dtype=np.float32).flags.writeable  # conversions and copies are the caller's own
    assert p.copy().flags.writeable and np.array(p).flags.writeable
    assert isinstance(p[0, 0], float) or np.isscalar(p[0, 0])


@pytest.mark.parametrize("write", ["setitem", "mask_rows", "ufunc_out", "fill_diagonal", "iadd", "copyto"])
def test_every_write_path_reaches_the_next_map_operation(write):
    eng, store, p = make()
    want = np.asarray(p).copy()
    if write == "setitem":
        p[0, 3] = 7.0; want[0, 3] = 7.0
    elif write == "mask_rows":  # planning/common/features.py:98-99
        msk = np.array([True, False, True, True, False, True])
        p[~msk, :] = 0; p[:, ~msk] = 0
        want[~msk, :] = 0; want[:, ~msk] = 0
    elif write == "ufunc_out":
        np.multiply(p, 2.0, out=p); want *= 2.0
    elif write == "fill_diagonal":
        np.fill_diagonal(p, 1.5); np.fill_diagonal(want, 1.5)
    elif write == "iadd":
        p += 1.0; want += 1.0
    elif write == "copyto":
        np.copyto(p, np.ones((6, 6))); want[...] = 1.0
    assert np.allclose(np.asarray(p), want)
    assert np.allclose(device_matrix(eng, store, p), want, atol=1e-6)  # re-uploaded: the device copy is not stale
    assert np.allclose(np.diag(p), np.diag(want), atol=1e-6) and abs(np.trace(p) - np.trace(want)) < 1e-5


def test_detached_state_hands_out_its_matrix_like_an_ndarray():
    eng, store, p = make()
    p[0, 0] = 3.0  # detaches
    a = np.asarray(p)
    assert a.flags.writeable
    a[1, 1] = 9.0  # plain ndarray aliasing: the host matrix IS the state now
    assert np.isclose(device_matrix(eng, store, p)[1, 1], 9.0)


def test_compat_slot_budget_scales_with_the_map():
    assert _runtime.state_slots_for(2500) == _runtime.STATE_SLOTS
    assert 2 <= _runtime.state_slots_for(200 * 200) <= 2 + _runtime.STATE_BYTES // (4 * 40000 * 40000)
    assert _runtime.state_slots_for(10 ** 6) == 2
