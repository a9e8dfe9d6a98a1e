"""
Host-side bookkeeping for ipp_tree_step: node ids, parents and paths of a forest of search trees whose states
live on the device as path-local factor columns (csrc/k_tree.h).  The counterpart of the reference's per-node
`state` matrices (planning/mcts_zero/mcts.py:16-21, planning/mcts_mission.py:25-31): a planner keeps its own
statistics (visit counts, values, priors) per node id and asks this pool for predict / expand steps in batches.
"""
from typing import List, Optional, Sequence

import numpy as np

from ..engine import IPPEngine


class TreeNodePool:
    def __init__(self, engine: IPPEngine, node_capacity: int):
        self.engine = engine
        self.capacity = int(node_capacity)
        self.depth_cap = engine.TREE_DEPTH
        self.parent = np.full(self.capacity, -1, dtype=np.int32)   # parent node, -1 = child of the root env
        self.root = np.full(self.capacity, -1, dtype=np.int32)     # env slot of the tree's root
        self.depth = np.zeros(self.capacity, dtype=np.int32)       # 1 = child of the root
        self.action = np.zeros((self.capacity, 3))                 # waypoint that led to the node
        self._next = 0

    def clear(self) -> None:
        """Forget every node (device storage is overwritten by later expansions)."""
        self._next = 0

    def __len__(self) -> int:
        return self._next

    def allocate(self, n: int = 1):
        """Reserve n node ids for a caller that issues ipp_tree_step itself (planning/mcts_zero/mcts.py)."""
        if self._next + n > self.capacity:
            raise RuntimeError(f"node pool exhausted ({self.capacity} nodes)")
        first = self._next
        self._next += n
        return first if n == 1 else np.arange(first, first + n, dtype=np.int32)

    def path(self, node: Optional[int]) -> List[int]:
        """Node ids from the root's child down to `node` (empty for the root itself), -1 padded to the engine's depth."""
        ids: List[int] = []
        while node is not None and node >= 0:
            ids.append(int(node))
            node = int(self.parent[node])
        ids.reverse()
        return ids + [-1] * (self.depth_cap - len(ids))

    def _gather(self, roots: Sequence[int], parents: Sequence[Optional[int]]):
        paths = np.array([self.path(p) for p in parents], dtype=np.int32).reshape(len(roots), self.depth_cap)
        return np.asarray(roots, dtype=np.int32), paths

    def predict(self, roots, parents, actions, prev_actions, **kw):
        """Reward of actions[i] from the state of node parents[i] (None / -1: the root env roots[i]); nothing recorded."""
        r, paths = self._gather(roots, parents)
        return self.engine.tree_step(r, paths, actions, prev_actions, new_ids=None, **kw)

    def expand(self, roots, parents, actions, prev_actions, **kw):
        """Like predict, and every step becomes a new child node; returns (reward, status, new node ids)."""
        n = len(roots)
        for p in parents:
            d = 0 if p is None or p < 0 else int(self.depth[p])
            if d + 1 > self.depth_cap:
                raise ValueError(f"a path holds at most {self.depth_cap} nodes")
        if self._next + n > self.capacity:
            raise RuntimeError(f"node pool exhausted ({self.capacity} nodes)")
        new = np.arange(self._next, self._next + n, dtype=np.int32)
        r, paths = self._gather(roots, parents)
        reward, status = self.engine.tree_step(r, paths, actions, prev_actions, new_ids=new, **kw)
        acts = np.asarray(actions, dtype=np.float64).reshape(n, 3) if not hasattr(actions, "cpu") else actions.cpu().numpy()
        for k in range(n):
            p = parents[k]
            self.parent[new[k]] = -1 if p is None else int(p)
            self.root[new[k]] = int(r[k])
            self.depth[new[k]] = 1 if p is None or p < 0 else int(self.depth[p]) + 1
            self.action[new[k]] = acts[k]
        self._next += n
        return reward, status, new
