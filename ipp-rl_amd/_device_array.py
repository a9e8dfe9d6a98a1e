"""
DeviceCov: the N x N covariance the drop-in classes hand to the reference's planners, resident on the GPU.

The reference's callers treat a map state as a dense float64 ndarray: they pass it back into
simulate_prediction_step / Mapping.update_grid_map (planning/mcts_zero/mcts.py:239, planning/mcts_mission.py:228-246),
read np.diag(state) / np.trace(state) (planning/common/rewards.py:11,23-30), keep many of them alive as tree-node states
(mcts.py:16-21) and only rarely look at the whole matrix (hash(str(state)), network input planes).  Shipping 25 MB down
and up again around every call made the drop-in path no faster than NumPy (29 ms per chained predict step at 50x50).

A DeviceCov is what those calls now return: an array-LIKE object (it implements __array__, __array_function__, indexing
and arithmetic) that owns one dense state slot of the compat engine.  Passed back into a map operation it is used in
place on the device (the step writes a NEW slot, like the reference allocates a new P'); np.diag / np.trace / .diagonal()
read the engine's cached diagonal (10 KB); anything else -- str(), slicing, arithmetic, pickling, deepcopy -- materialises
the float64 matrix on the host once and caches it.  Writing through it (state[~mask, :] = 0, features.py:98-99; ufuncs with
out=state; np.fill_diagonal / np.copyto / np.put / np.place / np.putmask on it) lands in the host copy and detaches the
device slot, so the next map operation uploads the modified matrix: aliasing semantics of the reference are kept.  While a
device slot is attached, every array or view that ESCAPES (np.asarray(state), state[i], state.T, state.ravel(), iteration)
is READ-ONLY: a write through it would change the cached host copy only and the next map operation would silently use the
stale device copy -- it raises "assignment destination is read-only" instead; once detached, the host matrix is the
state and is handed out writable like any ndarray.  Slots are recycled when their DeviceCov is garbage-collected; when all are in use the least
recently used state is moved to the host.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Optional

import numpy as np


_MUTATING = (np.fill_diagonal, np.copyto, np.put, np.place, np.putmask)  # array functions that write into their first argument


class SlotStore:
    """Dense state slots [first, first + count) of one compat engine, handed out to DeviceCov objects."""

    def __init__(self, engine, first: int, count: int):
        self.engine, self.first, self.count = engine, first, count
        self.free = list(range(first + count - 1, first - 1, -1))
        self.live = OrderedDict()  # slot -> weakref(DeviceCov), least recently used first
        self.uploads = self.downloads = self.evictions = 0

    def touch(self, slot: int):
        if slot in self.live:
            self.live.move_to_end(slot)

    def acquire(self, owner: "DeviceCov") -> int:
        if not self.free:
            self._evict()
        slot = self.free.pop()
        self.live[slot] = weakref.ref(owner)
        return slot

    def release(self, slot: int):
        if self.live.pop(slot, None) is not None:
            self.free.append(slot)

    def _evict(self):
        for slot, ref in list(self.live.items()):  # least recently used first
            owner = ref()
            if owner is None:
                self.release(slot)
                return
            if owner._pinned:
                continue
            owner._to_host()
            self.evictions += 1
            return
        raise RuntimeError(f"all {self.count} device state slots are pinned by running map operations")


def _release_slot(store_ref, slot):
    store = store_ref()
    if store is not None:
        store.release(slot)


class DeviceCov:
    """Array-like N x N float64 covariance living in a device slot (see the module docstring)."""

    __array_priority__ = 1000.0

    def __init__(self, store: SlotStore, n: int, slot: Optional[int] = None, host: Optional[np.ndarray] = None):
        self._store, self._n = store, int(n)
        self._slot, self._host, self._pinned = None, host, False
        self._finalizer = None
        if slot is not None:
            self._attach(slot)

    # ---------------------------------------------------------------- slot management
    @classmethod
    def new_on_device(cls, store: SlotStore, n: int) -> "DeviceCov":
        """An empty state whose slot a map operation is about to write."""
        obj = cls(store, n)
        obj._attach(store.acquire(obj))
        return obj

    def _attach(self, slot: int):
        self._slot = slot
        self._store.live[slot] = weakref.ref(self)
        self._finalizer = weakref.finalize(self, _release_slot, weakref.ref(self._store), slot)

    def _detach(self):
        if self._slot is not None:
            if self._finalizer is not None:
                self._finalizer.detach()
            self._store.release(self._slot)
            self._slot = None

    def _to_host(self) -> np.ndarray:
        """Materialise (once) and give the slot back."""
        host = self._materialise()
        self._detach()
        return host

    def _materialise(self) -> np.ndarray:
        if self._host is None:
            from ._runtime import to_host64

            self._host = to_host64(self._store.engine.read_cov(self._slot))
            self._store.downloads += 1
        return self._host

    def _view(self) -> np.ndarray:
        """The matrix as callers may see it: the host copy itself once the device slot is gone, a read-only view of the cache
        while the device copy is the live one."""
        host = self._materialise()
        if self._slot is None:
            return host
        ro = host.view()
        ro.flags.writeable = False
        return ro

    def _writable(self) -> np.ndarray:
        """The host matrix as the one live copy (for in-place writes): the device slot is dropped."""
        host = self._materialise()
        self._detach()
        return host

    def device_slot(self, store: SlotStore) -> Optional[int]:
        """Slot holding this state in `store`'s engine, uploading the host copy if it was evicted or modified; None when
        the object belongs to another engine (the caller then treats it as a plain array)."""
        if store is not self._store:
            return None
        if self._slot is None:
            slot = self._store.acquire(self)
            self._attach(slot)
            self._store.engine.write_cov(slot, self._host)
            self._store.uploads += 1
        self._store.touch(self._slot)
        return self._slot

    # ---------------------------------------------------------------- ndarray surface
    shape = property(lambda self: (self._n, self._n))
    ndim = 2
    dtype = np.dtype(np.float64)
    size = property(lambda self: self._n * self._n)
    T = property(lambda self: self._view().T)

    def __len__(self):
        return self._n

    def __array__(self, dtype=None, copy=None):
        a = self._view()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def diagonal(self) -> np.ndarray:
        if self._host is not None or self._slot is None:
            return np.einsum("ii->i", self._materialise()).copy()
        self._store.touch(self._slot)
        return self._store.engine.read_diag(self._slot).detach().cpu().numpy().astype(np.float64)

    def __array_function__(self, func, types, args, kwargs):
        # the two things planners ask of a state all the time: served from the engine's cached diagonal
        if func is np.diag and len(args) == 1 and not kwargs and args[0] is self:
            return self.diagonal()
        if func is np.trace and len(args) == 1 and not kwargs and args[0] is self:
            return float(self.diagonal().sum())
        if func in _MUTATING and args and isinstance(args[0], DeviceCov):  # in-place writes into the state: host copy, slot dropped
            args = (args[0]._writable(),) + tuple(args[1:])
        args = tuple(np.asarray(a) if isinstance(a, DeviceCov) else a for a in args)
        kwargs = {k: ((v._writable() if k == "out" else np.asarray(v)) if isinstance(v, DeviceCov) else v) for k, v in kwargs.items()}
        return func(*args, **kwargs)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(np.asarray(a) if isinstance(a, DeviceCov) else a for a in inputs)
        if "out" in kwargs:  # (written in place: the host copy becomes the state)
            kwargs["out"] = tuple(o._writable() if isinstance(o, DeviceCov) else o for o in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __getitem__(self, idx):
        return self._view()[idx]

    def __setitem__(self, idx, value):
        # callers that write into a state (features.py:98-99) change the host copy; the device copy is dropped so that
        # the next map operation sees the modification
        self._writable()[idx] = value

    def __iter__(self):
        return iter(self._view())

    def copy(self, order="C"):
        return self._materialise().copy(order=order)

    def astype(self, dtype, **kw):
        return self._materialise().astype(dtype, **kw)

    def flatten(self, order="C"):
        return self._materialise().flatten(order=order)

    def ravel(self, order="C"):
        return self._view().ravel(order=order)

    def __repr__(self):
        return repr(self._materialise())

    def __str__(self):
        return str(self._materialise())

    def __copy__(self):
        return self._materialise().copy()

    def __deepcopy__(self, memo):
        return self._materialise().copy()

    def __reduce__(self):  # pickled as the plain float64 matrix (workers upload it into their own engine)
        return (np.array, (self._materialise(),))

    def _binary(name):  # noqa: N805
        def op(self, other):
            return getattr(self._materialise(), name)(np.asarray(other) if isinstance(other, DeviceCov) else other)

        op.__name__ = name
        return op

    for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__matmul__",
               "__rmatmul__", "__pow__", "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__"):
        locals()[_n] = _binary(_n)
    del _n, _binary
    __hash__ = None  # like ndarray

    def _inplace(name):  # noqa: N805
        def op(self, other):
            getattr(self._writable(), name)(np.asarray(other) if isinstance(other, DeviceCov) else other)
            return self

        op.__name__ = name
        return op

    for _n in ("__iadd__", "__isub__", "__imul__", "__itruediv__"):
        locals()[_n] = _inplace(_n)
    del _n, _inplace

    def __neg__(self):
        return -self._materialise()

    def __abs__(self):
        return abs(self._materialise())

    def min(self, *a, **k):
        return self._materialise().min(*a, **k)

    def max(self, *a, **k):
        return self._materialise().max(*a, **k)

    def sum(self, *a, **k):
        return self._materialise().sum(*a, **k)
