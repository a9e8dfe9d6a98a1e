"""
Replay engine for tests/golden/call_trace.npz (recorded by tests/golden/gen_golden.py::gen_call_trace): the ordered attribute
reads / calls the reference's planners make on the class surface, re-issued on any set of stand-in objects -- this repo's drop-in
classes on the GPU (tests/test_hip_call_trace.py) or oracle-backed stand-ins on the CPU (tests/test_call_trace_oracle.py) -- with
every result compared.  An argument that was the RESULT of an earlier event is the replaying side's OWN result of that event.
"""
import json

import numpy as np

TOL = 1e-5


def _digest(a):
    """tests/golden/gen_golden.py::_Recorder.digest"""
    a = np.asarray(a, dtype=np.float64)
    flat = a.ravel()
    idx = np.random.RandomState(a.size).randint(0, a.size, 64)
    parts = [flat[idx], a.reshape(a.shape[0], -1).sum(axis=1)]
    if a.ndim == 2 and a.shape[0] == a.shape[1]:
        parts.append(np.diag(a))
    return np.concatenate(parts)


class _Replay:
    def __init__(self, g, prefix):
        self.g, self.p = g, prefix
        self.events = json.loads(bytes(g[prefix + "_trace"]).decode())
        self.ours = {}      # array number -> OUR object for it (result of the event that produced it)
        self.checked = 0
        self.worst = 0.0

    def stored(self, i):
        key = f"{self.p}_a{i:04d}"
        return self.g[key] if key in self.g.files else None

    def arg(self, d):
        if "a" in d:
            i = d["a"]
            if i in self.ours:
                return self.ours[i]
            a = self.stored(i)
            assert a is not None, f"argument array {i} is neither an earlier result nor stored"
            return np.array(a)
        if "t" in d:
            return [self.arg(x) for x in d["t"]]
        if "d" in d:
            return {k: self.arg(x) for k, x in d["d"].items()}
        return d.get("v")

    def check(self, d, got, where):
        if "a" in d:
            i = d["a"]
            self.ours.setdefault(i, got)
            ref = self.stored(i)
            arr = np.asarray(got, dtype=np.float64)
            if ref is not None:
                assert arr.shape == ref.shape, (where, arr.shape, ref.shape)
                err = float(np.max(np.abs(arr - ref))) if ref.size else 0.0
                assert err < TOL, (where, i, err)
            else:
                dg, shape = self.g[f"{self.p}_d{i:04d}"], tuple(self.g[f"{self.p}_s{i:04d}"])
                assert arr.shape == shape, (where, arr.shape, shape)
                mine = _digest(arr)
                err = float(np.max(np.abs(mine[:64] - dg[:64])))                     # 64 entries
                err = max(err, float(np.max(np.abs(mine[64 + shape[0]:] - dg[64 + shape[0]:]))) if len(dg) > 64 + shape[0] else 0.0)  # diagonal
                assert err < TOL, (where, i, err)
                rows = float(np.max(np.abs(mine[64:64 + shape[0]] - dg[64:64 + shape[0]])))
                assert rows < TOL * max(1, arr.shape[-1]) ** 0.5 * 4, (where, i, rows)  # row sums of n entries
            self.worst = max(self.worst, err)
            self.checked += 1
        elif "t" in d:
            assert len(got) == len(d["t"]), where
            for x, y in zip(d["t"], got):
                self.check(x, y, where)
        elif "d" in d:
            for k, x in d["d"].items():
                self.check(x, got[k], where)
        elif "v" in d:
            v = d["v"]
            if isinstance(v, float):
                assert abs(float(got) - v) < TOL * max(1.0, abs(v)), (where, got, v)
                self.checked += 1
            elif v is None:
                assert got is None, where
            else:
                assert got == v, (where, got, v)

    def rng(self, ev):
        i = ev["rng"]
        keys, pos, gauss = self.g[f"{self.p}_a{i:04d}"], self.g[f"{self.p}_a{i + 1:04d}"], self.g[f"{self.p}_a{i + 2:04d}"]
        np.random.set_state(("MT19937", keys.astype(np.uint32), int(pos[0]), int(pos[1]), float(gauss[0])))



def run_trace(g, prefix, objs, fns, simulate):
    """objs: {"mapping": .., "mapping.grid_map": .., "mapping.sensor": .., "mapping.sensor.sensor_simulation": ..}; fns: the
    planning.common helpers by name; simulate(state, prev, action, uav, info) -> (reward, next_state).  Returns (replay, event counts)."""
    rp = _Replay(g, prefix)
    kinds = {}
    for n, ev in enumerate(rp.events):
        where = (n, ev["k"], ev["o"], ev["n"])
        kinds[where[1:]] = kinds.get(where[1:], 0) + 1
        if ev["k"] == "get":
            rp.check(ev["r"], getattr(objs[ev["o"]], ev["n"]), where)
        elif ev["k"] == "set":
            setattr(objs[ev["o"]], ev["n"], rp.arg(ev["args"][0]))
        elif ev["o"] == "fn" and ev["n"] == "simulate_prediction_step":
            state, prev, action, uav, info = [rp.arg(a) for a in ev["args"]]
            rp.check(ev["r"], simulate(state, np.asarray(prev), np.asarray(action), uav, info), where)
        elif ev["o"] == "fn":
            args = [rp.arg(a) for a in ev.get("args", [])]
            kw = {k: rp.arg(v) for k, v in ev.get("kw", {}).items()}
            rp.check(ev["r"], fns[ev["n"]](*args, **kw), where)
        else:
            if "rng" in ev:
                rp.rng(ev)  # the planner drew from NumPy's stream in between: the observation noise starts where it stood
            args = [rp.arg(a) for a in ev.get("args", [])]
            kw = {k: rp.arg(v) for k, v in ev.get("kw", {}).items()}
            rp.check(ev["r"], getattr(objs[ev["o"]], ev["n"])(*args, **kw), where)
    return rp, kinds
