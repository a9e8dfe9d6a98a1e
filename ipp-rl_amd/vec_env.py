"""
VecIPPEnv -- batched driver over the HIP step engine: B independent IPP environments in opaque env
slots on one GPU.

One env step = the reference's self-play step sequence fused (planning/mcts_zero/episode_generators.py:
137-146): simulate_prediction_step (reward) + sensor.take_measurement + mapping.update_grid_map.
Episodes are ``episode_steps`` long (max_episode_steps, config/example.yaml:64); an episode reset draws a
new Gaussian-random-field ground truth and (optionally) shuffled prior hyper-parameters
(shuffle_prior_cov, mapping/mappings.py:238-240) and returns the UAV to Mission.init_action
(planning/missions.py:69).

Randomness: parity runs pass NumPy legacy-stream normals in (``white_noise`` / ``meas_noise`` arguments);
throughput runs let the engine's Philox generator fill them on the device (stated in reports).
Device noise is keyed on the GLOBAL env id (ipp_fill_normal_rows: Philox counter = global env id x row length +
element, subsequence = stream kind + the env's episode index / the step index), so an env's ground truths and
measurement noise do not depend on how the batch is sharded over GPUs (tests/test_hip_sharding.py).
"""
from __future__ import annotations

from typing import Optional, Sequence

import os

import numpy as np

from . import _ffi
from .engine import EngineConfig, IPPEngine

INIT_ACTION = (2.0, 2.0, 14.0)  # planning/missions.py:69


def shard_range(total: int, rank: int, world: int):
    """Contiguous env-id range of `rank`: [rank*total/world, (rank+1)*total/world)."""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def cell_centre_actions(cfg: EngineConfig, step: int, env_lo: int, env_hi: int, total_envs: int,
                        altitudes: Sequence[float]) -> np.ndarray:
    """
    Synthetic workload of SURVEY 8(d): uniform over the N * len(altitudes) cell-centre actions, drawn from
    RandomState(10_000 + step) for ALL `total_envs` envs and sliced, so a shard sees the same actions
    whatever the GPU count.
    """
    rs = np.random.RandomState(10_000 + step)
    col = rs.randint(0, cfg.x_dim, total_envs)
    row = rs.randint(0, cfg.y_dim, total_envs)
    lev = rs.randint(0, len(altitudes), total_envs)
    alts = np.asarray(altitudes, dtype=np.float64)
    acts = np.stack([cfg.resolution * col + 0.5 * cfg.resolution, cfg.resolution * row + 0.5 * cfg.resolution,
                     alts[lev]], axis=1)
    return acts[env_lo:env_hi]


class VecIPPEnv:
    NOISE_RING = 16

    def __init__(self, cfg: EngineConfig, num_envs: int, state: str = "factor", episode_steps: int = 40,
                 device: str = "cuda:0", seed: int = 1234, env_id_offset: int = 0, shuffle_prior_cov: bool = False,
                 rank_cap: Optional[int] = None, stagger: bool = False, tile_threads: int = 0,
                 adaptive: bool = True, use_flight_time: bool = True, window_rows: int = 0, fused_reset: bool = True,
                 parts: int = 1, arena=None):
        import torch

        self.torch = torch
        self.cfg = cfg
        self.num_envs = int(num_envs)
        self.episode_steps = int(episode_steps)
        self.seed = int(seed)
        self.env_id_offset = int(env_id_offset)
        self.shuffle_prior_cov = shuffle_prior_cov
        self.adaptive, self.use_flight_time = adaptive, use_flight_time
        rank_cap = int(rank_cap) if rank_cap else 9 * self.episode_steps
        self.engine = IPPEngine(cfg, capacity=self.num_envs, state=state, rank_cap=rank_cap, device=device,
                                tile_threads=tile_threads, window_rows=window_rows, fixed_prior=not shuffle_prior_cov,
                                arena=arena)
        dev = self.engine.device
        self.device = dev
        B = self.num_envs
        self.prev = torch.tensor(INIT_ACTION, dtype=torch.float64, device=dev).repeat(B, 1)
        self.init_prev = self.prev.clone()
        # stagger: env e starts its first episode at phase e % T so that every batched step sees the
        # stationary mix of ranks 0 .. (T-1)*m (throughput runs); otherwise all envs run in lock step
        self.phase = (torch.arange(B, device=dev) + self.env_id_offset) % self.episode_steps if stagger else \
            torch.zeros(B, dtype=torch.int64, device=dev)
        self.t = 0
        self.episode = np.zeros(B, dtype=np.int64)  # completed-episode counters (host side, never read by kernels)
        # dispatch orders of the step launches, one per schedule phase (staggered runs): env e has done (t + phase_e) % T steps
        # of its episode when step t starts; descending = longest items first (engine.set_item_order)
        self._orders = None
        if stagger and os.environ.get("IPP_ITEM_ORDER", "1") != "0":
            ph = (np.arange(B, dtype=np.int64) + self.env_id_offset) % self.episode_steps
            self._orders = [torch.as_tensor(np.argsort(-((t + ph) % self.episode_steps), kind="stable").astype(np.int32), device=dev)
                            for t in range(self.episode_steps)]
        # parts > 1: a FIXED partition of the envs into `parts` groups (every group holds every episode phase), each stepped by
        # its own launch on its own stream (step_async / wait): only an env's OWN step t + 1 has to follow its step t, so the
        # next launch of one group fills the slots that the slowest items of the other group still leave empty
        # (IPPEngine.step_parts).  Their dispatch orders list group 0's items (heaviest first), then group 1's, ...
        self.parts = 1
        self._part_begin = None
        self._part_streams = None
        if parts > 1 and stagger and self._orders is not None and self.engine.info.fused_step == 1 and B >= 2 * parts:
            self.parts = int(parts)
            grp = ((np.arange(B, dtype=np.int64) + self.env_id_offset) // self.episode_steps) % self.parts
            ph = (np.arange(B, dtype=np.int64) + self.env_id_offset) % self.episode_steps
            self._orders_parts = []
            for t in range(self.episode_steps):
                w = (t + ph) % self.episode_steps
                key = grp * (2 * self.episode_steps) + (self.episode_steps - 1 - w)  # group, then descending weight, then env id
                self._orders_parts.append(torch.as_tensor(np.argsort(key, kind="stable").astype(np.int32), device=dev))
            self._part_begin = [0] + [int(x) for x in np.cumsum(np.bincount(grp, minlength=self.parts))]
            self._part_envs = [torch.as_tensor(np.nonzero(grp == g)[0].astype(np.int64), device=dev) for g in range(self.parts)]
            self._part_streams, self._side_pick = self._pick_streams(dev, self.parts)
            self._part_done = [torch.cuda.Event() for _ in range(self.parts)]
            self._ev_inputs = torch.cuda.Event()
            self._async_pending = False
            self._main_dirty = True  # the caller's stream holds work on the env slots that the part streams have not waited for
        self._reset_ids_by_phase = None
        if stagger:
            ph = self.phase.cpu().numpy()
            self._reset_ids_host = [np.nonzero(ph == p)[0].astype(np.int32) for p in range(self.episode_steps)]
            self._reset_ids_by_phase = [torch.as_tensor(i, device=dev) for i in self._reset_ids_host]
        # fused_reset: scheduled resets folded into the step launch (ipp_step_autoreset): per phase, the index of every
        # env's new ground truth among the staged fields, -1 for the envs that carry on (built on first use).  On by default
        # since (a) the waves of a resetting item wait for their own stores instead of an agent-scope release fence (which
        # wrote the L2 back: +13 us per step kernel) and (b) the staging meets the main stream once per block of steps, not
        # once per step: 4096 envs of 50x50: 23.1-23.9 M env-steps/s against 22.6 M with the separate reset launch.
        # Shuffled priors can ride along (ipp_set_reset_prior: the per-episode (sigma^2, l) of every staged ground truth).
        # Only where the engine runs the fused step kernel: on the split path of large batches (k_prepare + k_gain_factor in
        # chunks) the resets are a launch of their own either way and the folded form is slower (32768 envs: 24.9 vs 28.7 M).
        # (with shuffled priors the folded form is available -- fused_reset="always" -- but not the default: measured equal
        # to the separate launch, 19.4-19.5 vs 19.3-20.3 M env-steps/s at window 12; a partitioned env takes it: step_async needs the
        # resets inside the step launches)
        self._fused_reset = bool(fused_reset and stagger and state == "factor" and
                                 (not shuffle_prior_cov or fused_reset == "always" or self.parts > 1) and
                                 self.engine.info.fused_step == 1 and self.engine.info.window_rows > 0 and
                                 4 * B * self.episode_steps <= (64 << 20))
        self._reset_src_by_phase = {}
        self._prior_ring = {}
        self._white = torch.empty((B, cfg.n_cells), dtype=torch.float32, device=dev)
        # staggered runs prepare the next resets' ground truths on a side stream while the step kernels run
        # (the GRF convolution is fp64-compute-bound, the step is HBM-bound: they overlap on the chip); the field
        # for the resets after step t+1 is started at the beginning of step t, so it has two steps to finish
        self._side = None
        if stagger:
            self._side = self._side_pick if self._part_streams is not None else self._pick_streams(dev, 1)[1]
        self._grf_rows = None if os.environ.get("IPP_GRF_ROWS", "1") != "0" else False  # (False: the engine has no in-generator noise / A/B)
        self._gt_flip_ok = os.environ.get("IPP_GT_FLIP", "1") != "0"  # (A/B: 0 = staged buffers + copies at the resets)
        self.alt_blocks = 0
        # Staging in BLOCKS of K steps: the fields of all resets of block b + 1 are generated while block b runs, into one of
        # two buffer sets, and the streams meet ONCE per block (main waits for the block's `ready` event before its first step
        # and records `free` behind its last; the side stream waits for `free` before it refills the set).  With an event
        # wait and an event record per step the stream protocol cost 12-20 us of every step (4096 envs of 50x50: 0.186 ms per
        # step against 0.171 without events, profiles/r02_experiments.txt).  2 K <= episode_steps: within the staging
        # horizon an env resets at most once, so its episode counter read at staging time names the right ground truth.
        if stagger:
            n_max = max(int(i.numel()) for i in self._reset_ids_by_phase)
            # (cap of the two staged sets: 2 GiB -- with 256 MiB configs[2], 2048 resets of 40 KB per step, staged block by block of ONE step
            # and paid the streams' meeting every step: 0.4695 against 0.4436 ms per step with blocks of eight, profiles/r06_experiments.txt 14)
            K = max(1, min(8, self.episode_steps // 2, (2 << 30) // max(1, n_max * cfg.n_cells * 8)))
            if os.environ.get("IPP_BLK_K"):  # (A/B: steps per staging block)
                K = max(1, min(int(os.environ["IPP_BLK_K"]), self.episode_steps // 2))
            self._blk_K = K
            # (the K fields of a set are rows [j n_max, (j + 1) n_max) of ONE buffer: a block can be generated by one launch.  The buffers are
            # allocated on FIRST USE: where the fields go straight into the envs' alternate planes -- every BASELINE config -- nothing ever reads
            # them, and at configs[2] they are 2 x 1.3 GB + as much white-noise scratch that would go back to the driver when the env closes,
            # cutting up the device memory the next large arena's 1-GiB chunks come from)
            self._staged_sets = None
            self._staged = None
            self._blk_nmax = n_max
            self._blk_ids = {}  # first step of a block mod the episode length -> device int32 [K n_max] row ids (-1: padding)
            self._staged_white = None
            self._blk_ready = [torch.cuda.Event() for _ in range(2)]
            # (`free` of a buffer set: one event per stream that steps the batch -- the caller's, or one per part)
            self._blk_free = [[torch.cuda.Event() for _ in range(self.parts)] for _ in range(2)]
            for evs in self._blk_free:
                for ev in evs:
                    ev.record(torch.cuda.current_stream(dev))
            self._blk_tag = [-1, -1]   # block index staged in each buffer set
            self._blk_prog = [(-1, 0), (-1, 0)]  # (block, fields staged so far) of a block that is being staged field by field
            self._blk_alt = [False, False]       # the set's block went straight into the envs' ALTERNATE ground-truth planes (resets flip)
            self._blk_waited = -1      # block whose `ready` event the main stream has waited for
        # measurement noise for NOISE_RING steps per generator launch
        self._noise_ring = torch.empty((self.NOISE_RING, B, self.engine.meas_cap), dtype=torch.float32, device=dev)
        self._noise_pos = 0
        self._noise_fills = 0
        if self.parts > 1:  # two rings: the next one is filled on the caller's stream while the part streams read this one
            self._noise_rings = [self._noise_ring, torch.empty_like(self._noise_ring)]
            self._ring_ready = [torch.cuda.Event() for _ in range(2)]
            self._ring_free = [[torch.cuda.Event() for _ in range(self.parts)] for _ in range(2)]
            self._ring_filled = [-1, -1]  # fill index held by each ring
            self._ring_cur = 0
        self.reward = torch.empty(B, dtype=torch.float32, device=dev)
        self.status = torch.empty(B, dtype=torch.int32, device=dev)
        self._flags = (4 if adaptive else 0) | (8 if use_flight_time else 0)

    # ------------------------------------------------------------------ resets
    def _prior_scale(self, ids_host: np.ndarray, episode_index):
        """(sigma^2, l) ~ U(0.8, 1.2) x nominal per env and episode (mapping/mappings.py:238-240), from a counter hash
        of (seed, GLOBAL env id, episode index, which draw): vectorised, no generator state, the same for an env
        whatever the sharding.  Returns float64 [n, 2] (episode_index: scalar or [n])."""
        if not self.shuffle_prior_cov:
            return None
        gid = np.asarray(ids_host, dtype=np.uint64) + np.uint64(self.env_id_offset)
        epi = np.broadcast_to(np.asarray(episode_index, dtype=np.uint64), gid.shape)
        out = np.empty((len(gid), 2))
        with np.errstate(over="ignore"):
            for d, nominal in enumerate((self.cfg.signal_variance, self.cfg.length_scale)):
                x = (gid * np.uint64(0x9E3779B97F4A7C15) + epi * np.uint64(0xD1B54A32D192ED03) +
                     np.uint64((self.seed * 2 + d) & (2 ** 64 - 1)) * np.uint64(0x8CB92BA72F3D8DD7))
                x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)  # splitmix64 finaliser
                x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
                x ^= x >> np.uint64(31)
                u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)
                out[:, d] = nominal * (0.8 + 0.4 * u)
        return out

    PRIOR_RING = 16

    def _prior_scale_scheduled(self, phase: int):
        """Device [n, 2] prior scales for the next episode of the envs of a scheduled reset group; drawn PRIOR_RING
        episodes at a time so that the host work and the upload are off the per-step path."""
        ids = self._reset_ids_host[phase]
        epis = self.episode[ids] if len(ids) else np.zeros(0, dtype=np.int64)
        if len(ids) and not np.all(epis == epis[0]):
            # a hand-made reset(env_ids=subset) moved some counters of this group: every env draws for ITS OWN episode index
            # (the prior then matches the ground truth's episode and does not depend on which env is first in the shard)
            return self.torch.as_tensor(self._prior_scale(ids, epis), dtype=self.torch.float64, device=self.device)
        epi = int(epis[0]) if len(ids) else 0
        hit = self._prior_ring.get(phase)
        if hit is None or not (hit[0] <= epi < hit[0] + self.PRIOR_RING):
            block = np.stack([self._prior_scale(ids, epi + k) for k in range(self.PRIOR_RING)])
            hit = (epi, self.torch.as_tensor(block, dtype=self.torch.float64, device=self.device))
            self._prior_ring[phase] = hit
        return hit[1][epi - hit[0]]

    def reset(self, env_ids=None, white_noise=None, gt=None, prior_scale=None, _phase=None):
        """Reset the given slots (all when None).  white_noise / gt: [n, H, W] NumPy or tensor (parity)."""
        torch = self.torch
        if self.parts > 1:
            self.wait()
            self._main_dirty = True
        if _phase is not None:  # scheduled reset of step(): index tensors prepared at construction
            ids = self._reset_ids_by_phase[_phase]
        elif env_ids is None:
            ids = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
        else:
            ids = torch.as_tensor(env_ids, dtype=torch.int32, device=self.device)
        n = int(ids.numel())
        if n == 0:
            return
        if gt is None and white_noise is None:
            white_noise = self._white_for(ids, self._ids_host(ids, env_ids, _phase), self._white[:n])
        if prior_scale is None and self.shuffle_prior_cov:
            if _phase is not None:
                prior_scale = self._prior_scale_scheduled(_phase)
            else:
                ih = self._ids_host(ids, env_ids, _phase)
                prior_scale = self._prior_scale(ih, self.episode[np.asarray(ih, dtype=np.int64)])
        # the reset kernel also returns the UAVs to Mission.init_action (ipp_reset_episode)
        self.engine.reset(env_ids=ids, prior_scale=prior_scale, gt=gt, white_noise=white_noise, prev=self.prev,
                          init_action=INIT_ACTION)
        if _phase is not None:
            self.episode[self._reset_ids_host[_phase]] += 1
        else:
            self._invalidate_staging()
            if env_ids is None:
                self.episode += 1
            elif not torch.is_tensor(env_ids):
                self.episode[np.asarray(env_ids, dtype=np.int64)] += 1
            else:
                self.episode[ids.cpu().numpy()] += 1

    def _reset_src(self, p: int):
        src = self._reset_src_by_phase.get(p)
        if src is None:
            host = np.full(self.num_envs, -1, dtype=np.int32)
            host[self._reset_ids_host[p]] = np.arange(len(self._reset_ids_host[p]), dtype=np.int32)
            src = self._reset_src_by_phase[p] = self.torch.as_tensor(host, device=self.device)
        return src

    def _phase_ending_at(self, t: int) -> int:
        """Phase whose envs finish their episode with step index t: env e has done (t + 1 + phase_e) steps."""
        return (self.episode_steps - ((t + 1) % self.episode_steps)) % self.episode_steps

    def _staged_buffers(self):
        """The two staged buffer sets (K fields of n_max rows each) and their white-noise scratch, allocated on first use."""
        if self._staged_sets is None:
            torch, K, nm, N = self.torch, self._blk_K, self._blk_nmax, self.cfg.n_cells
            self._staged_sets = [torch.empty((K * nm, N), dtype=torch.float32, device=self.device) for _ in range(2)]
            self._staged = [self._staged_sets[q // K][(q % K) * nm:(q % K + 1) * nm] for q in range(2 * K)]
        return self._staged_sets

    def _staged_field(self, q: int):
        self._staged_buffers()
        return self._staged[q]

    def _staged_white_field(self, q: int):
        if self._staged_white is None:
            self._staged_white = [self.torch.empty((self._blk_nmax, self.cfg.n_cells), dtype=self.torch.float32, device=self.device)
                                  for _ in range(2 * self._blk_K)]
        return self._staged_white[q]

    def _stage_block(self, b: int, upto: Optional[int] = None):
        """Start, on the side stream, the ground truths of the resets of the steps [b K, b K + upto) into buffer set b % 2 (upto = K: the
        whole block, then its `ready` event).  Called with growing `upto` it stages a block field by field: the prefetch of the NEXT
        block is spread over the steps of the running one, one field behind every step's launches -- staged at once (eight generator
        calls, 160-190 us of host time) in front of a step's launches it was a bubble of that length whenever the host had no lead
        over the device, i.e. at the start of every timed region (tools/host_step_times.py)."""
        torch = self.torch
        K, set_ = self._blk_K, b % 2
        upto = K if upto is None else min(int(upto), K)
        j0 = self._blk_prog[set_][1] if self._blk_prog[set_][0] == b else 0
        if j0 == 0:
            for ev in self._blk_free[set_]:
                self._side.wait_event(ev)
        if j0 == 0:
            self._blk_alt[set_] = False
        if j0 == 0 and self._grf_rows is not False and self._stage_block_once(b):
            upto = K  # (the whole block went out as ONE generator launch)
            j0 = K
        self._blk_prog[set_] = (b, max(j0, upto))
        with torch.cuda.stream(self._side):
            for j in range(j0, upto):
                p = self._phase_ending_at(b * K + j)
                n = int(self._reset_ids_by_phase[p].numel())
                if n == 0:
                    continue
                buf = set_ * K + j
                # the envs of a scheduled reset share their episode index: the generator draws the white noise itself where it can
                # (50x50 / 100x100: no [n, N] noise array written and read back), else fill + generate
                epi = self.episode[self._reset_ids_host[p]]
                if self._grf_rows is not False and len(epi) and np.all(epi == epi[0]):
                    self._grf_rows = self.engine.generate_grf_rows(n, self.seed, self.GT_STREAM + int(epi[0]), self._staged_field(buf)[:n],
                                                                   row_ids=self._reset_ids_by_phase[p], row_offset=self.env_id_offset,
                                                                   stream=self._side)
                    if self._grf_rows:
                        continue
                white = self._white_for(self._reset_ids_by_phase[p], self._reset_ids_host[p], self._staged_white_field(buf)[:n])
                self.engine.generate_grf(white, out=self._staged_field(buf)[:n], stream=self._side)
            if upto == K:
                self._blk_ready[set_].record(self._side)
        if upto == K:
            self._blk_tag[set_] = b

    def _stage_block_once(self, b: int) -> bool:
        """All ground truths of block b by ONE launch of the generator that draws its own white noise (50x50 / 100x100 grids): field
        (j, i) = row j n_max + i of the set's buffer, row id = the env, subsequence = GT_STREAM + the episode index of the row's PHASE (per-group
        offsets passed by value: the phases of a block may be in different episodes; no upload in front of the launch, which on a
        stream that waits for the part streams would stall the host), padding rows skipped.  Eight launches of ~102 fields cost
        eight host calls (160-190 us) and 8 x 11 us of the side stream; one launch of 819 fields 0.020 ms.  The same fields bit for
        bit (one Philox definition, keyed on the global env id).  False: this grid has no such generator."""
        torch = self.torch
        K, set_, nm = self._blk_K, b % 2, self._blk_nmax
        if K * nm > int(self.engine.max_batch):
            return False
        key = (b * K) % self.episode_steps  # (the phases of a block's steps depend on its first step modulo the episode length)
        ids = self._blk_ids.get(key)
        phases = [self._phase_ending_at(b * K + j) for j in range(K)]
        if ids is None:
            host = np.full(K * nm, -1, dtype=np.int32)
            for j, p in enumerate(phases):
                h = self._reset_ids_host[p]
                host[j * nm:j * nm + len(h)] = h
            ids = self._blk_ids[key] = torch.as_tensor(host, device=self.device)
        epi = []
        for p in phases:  # (the envs of a scheduled reset share their episode index; a hand-made reset of a subset breaks that: per-phase launches then)
            e_p = self.episode[self._reset_ids_host[p]]
            if len(e_p) and not np.all(e_p == e_p[0]):
                return False
            epi.append(int(e_p[0]) if len(e_p) else 0)
        # where the resets are folded into the step launches the fields go straight into the envs' alternate planes and a reset is a
        # flip (no copy of H W floats in and out per reset); else into the set's buffer, which the separate reset launch copies
        # (2 K > episode_steps: a block is staged when its first step arrives, behind the `free` event of block b - 2 only -- while
        # block b - 1, whose resets flip the very planes this launch would write, may still be running: staged buffers + copies then)
        alt = bool(self._fused_reset) and self._gt_flip_ok and 2 * K <= self.episode_steps
        ok = self.engine.generate_grf_rows(K * nm, self.seed, self.GT_STREAM, None if alt else self._staged_buffers()[set_], row_ids=ids,
                                           row_offset=self.env_id_offset, stream=self._side, group_rows=nm, group_subsequence=epi)
        if not ok:
            self._grf_rows = False
            return False
        self._grf_rows = True
        self._blk_alt[set_] = alt
        self.alt_blocks += int(alt)  # (blocks of ground truths generated straight into the alternate planes: tests, reports)
        return True

    def _prefetch_next_block(self, b: int, j: int):
        """Behind the launches of step j of block b: field j of block b + 1 (only when an env cannot reset twice inside the two blocks,
        2 K <= episode_steps: with shorter episodes block b + 1 would be staged before block b's resets have moved the episode
        counters and would repeat its ground truths; it is then staged whole when its first step arrives)."""
        if self._blk_tag[(b + 1) % 2] != b + 1 and 2 * self._blk_K <= self.episode_steps:
            if os.environ.get("IPP_STAGE_SPREAD", "1") == "0":  # A/B: the whole block at once, behind the first step's launches
                self._stage_block(b + 1)
            else:
                self._stage_block(b + 1, upto=j + 1)

    def _invalidate_staging(self):
        """A reset outside the schedule moved episode counters: staged fields may name the wrong episodes."""
        if self._side is None:
            return
        main = self.torch.cuda.current_stream(self.device)
        if self.parts > 1:
            self.wait()  # (the part streams' reads of the sets are in front of this point of the caller's stream)
        for set_ in range(2):
            self._blk_tag[set_] = -1
            self._blk_prog[set_] = (-1, 0)
            self._blk_alt[set_] = False
            for ev in self._blk_free[set_]:
                ev.record(main)  # (everything that read the set is in front of this point of the stream)
        self._blk_waited = -1

    GT_STREAM, NOISE_STREAM = 1 << 40, 2 << 40  # subsequence = stream kind + episode index / step index

    def _ids_host(self, ids, env_ids, phase):
        if phase is not None:
            return self._reset_ids_host[phase]
        if env_ids is None:
            return np.arange(self.num_envs, dtype=np.int32)
        return ids.cpu().numpy() if self.torch.is_tensor(env_ids) else np.asarray(env_ids, dtype=np.int32).ravel()

    def _white_for(self, ids_dev, ids_host, out):
        """Ground-truth white noise of the NEXT episode of the given envs: row = global env id, subsequence = that env's
        episode index (the envs of one scheduled reset share it: one launch; a mixed set takes one launch per index)."""
        epi = self.episode[np.asarray(ids_host, dtype=np.int64)]
        uniq = np.unique(epi)
        if len(uniq) == 1:
            self.engine.normal_rows(out, self.cfg.n_cells, self.seed, self.GT_STREAM + int(uniq[0]), row_ids=ids_dev,
                                    row_offset=self.env_id_offset)
            return out
        for e in uniq:  # (hand-made reset sets only; each launch fills the rows of one episode index)
            sel = np.nonzero(epi == e)[0]
            tmp = self.torch.empty((len(sel), self.cfg.n_cells), dtype=self.torch.float32, device=self.device)
            self.engine.normal_rows(tmp, self.cfg.n_cells, self.seed, self.GT_STREAM + int(e),
                                    row_ids=np.asarray(ids_host)[sel].astype(np.int32), row_offset=self.env_id_offset)
            out[self.torch.as_tensor(sel, device=self.device)] = tmp
        return out

    # ------------------------------------------------------------------ stepping
    def step(self, actions, meas_noise=None, env_ids=None, auto_reset: bool = True, after_step_hook=None):
        """
        actions: [B, 3] float64 (NumPy or device tensor).  Returns (reward, status) device tensors.
        With stagger=True and auto_reset, the envs whose episode ends after this step are reset on a fixed
        schedule known to the host (no device->host sync in the loop).
        """
        torch = self.torch
        if self.parts > 1:
            if env_ids is None and auto_reset and meas_noise is None and after_step_hook is None and \
                    (self._fused_reset or self._reset_ids_by_phase is None):
                # the synchronous call on a partitioned batch: the parts' launches run side by side, the caller's stream
                # continues behind both (same results, one join per step)
                self.step_async(actions)
                self.wait()
                return self.reward, self.status
            self.wait()  # any other call form runs on the caller's stream, behind everything the part streams hold
            self._main_dirty = True
        a = self.engine._dev(actions, torch.float64).reshape(-1, 3)
        main = torch.cuda.current_stream(self.device)
        scheduled = None
        blk = None
        prefetch = None
        if auto_reset and self._reset_ids_by_phase is not None:
            K = self._blk_K
            b, j = divmod(self.t, K)
            set_ = b % 2
            if self._blk_tag[set_] != b:  # first step (or the schedule was disturbed): stage this block now
                self._stage_block(b)
                self._blk_waited = -1
            if self._blk_waited != b:
                main.wait_event(self._blk_ready[set_])
                self._blk_waited = b
            p = self._phase_ending_at(self.t)
            n = int(self._reset_ids_by_phase[p].numel())
            scheduled = (p, set_ * K + j, n) if n > 0 else None
            blk = (set_, j == K - 1)
            prefetch = (b, j)  # (the next block's fields are generated while this block runs: behind this step's launches)
        elif self._reset_ids_by_phase is not None:
            # a step outside the reset schedule (auto_reset=False): its staged fields stay unused, but when it is the LAST step of
            # its block the block's `free` event is still recorded below -- else the side stream would refill this buffer set
            # behind an older event while resets of the main stream may still be reading it
            b_, j_ = divmod(self.t, self._blk_K)
            blk = (b_ % 2, j_ == self._blk_K - 1) if self._blk_tag[b_ % 2] == b_ else None
        if meas_noise is None and self.parts > 1:
            nz = self._noise_plane_parts(main, None)
        elif meas_noise is None:
            if self._noise_pos == 0:
                # plane p of the ring = step (fills * NOISE_RING + p); row = global env id
                self.engine.normal_rows(self._noise_ring, self.engine.meas_cap, self.seed, self.NOISE_STREAM +
                                        self._noise_fills * self.NOISE_RING, row_offset=self.env_id_offset)
                self._noise_fills += 1
            nz = self._noise_ring[self._noise_pos]
            self._noise_pos = (self._noise_pos + 1) % self.NOISE_RING
        else:
            nz = meas_noise
        # full-batch steps let the kernel store the new previous waypoint (IPP_UPDATE_PREV): no copy launch; with staged
        # ground truths the scheduled resets ride in the same launch as well (ipp_step_autoreset)
        fused = None
        if self._fused_reset and env_ids is None and scheduled is not None:
            p, k, n = scheduled
            fused = dict(reset_src=self._reset_src(p), reset_gt=None if self._blk_alt[k // self._blk_K] else self._staged_field(k)[:n], init_action=INIT_ACTION)
            if self.shuffle_prior_cov:  # priors of the episodes this launch starts, row i for staged field i
                self.engine.set_reset_prior(self._prior_scale_scheduled(p))
        if self._orders is not None and env_ids is None:
            # heaviest items first: an env's stored columns grow with the steps since its reset
            self.engine.set_item_order(self._orders[self.t % self.episode_steps])
        self.engine.step(a, self.prev, env_ids=env_ids, meas_noise=nz, adaptive=self.adaptive,
                         use_flight_time=self.use_flight_time, reward_out=self.reward, status_out=self.status,
                         update_prev=env_ids is None, **(fused or {}))
        if meas_noise is None and self.parts > 1:
            self._noise_done_parts(None)
        if fused is not None:
            self.episode[self._reset_ids_host[scheduled[0]]] += 1
            scheduled = None
        if env_ids is not None:
            self.prev[torch.as_tensor(env_ids, device=self.device).long()] = a
        self.t += 1
        if after_step_hook is not None:
            after_step_hook()
        if scheduled is not None:
            p, k, n = scheduled
            # (a block that went into the alternate planes has no buffer to copy from: the separate reset draws the same field again)
            self.reset(gt=None if self._blk_alt[k // self._blk_K] else self._staged_field(k)[:n], _phase=p)
        if prefetch is not None:
            self._prefetch_next_block(*prefetch)
        if blk is not None and blk[1]:
            for ev in self._blk_free[blk[0]]:
                ev.record(main)  # last step of the block: its buffer set may be refilled
        return self.reward, self.status

    # ------------------------------------------------------------------ partitioned batch: one launch and one stream per part
    def _pick_streams(self, dev, n_parts: int):
        """The streams of the part launches (n_parts > 1) and of the ground-truth staging, one hardware queue each.
        The runtime binds a stream to one of a few hardware queues (four by default) at first use; two streams on ONE queue
        take turns, whatever the events say.  Measured at 4096 envs of 50x50 with two groups (profiles/r04_experiments.txt
        item 12): staging stream and both part streams on three queues 59.5 us per step (68.8 M env-steps/s), staging stream
        on a part stream's queue 70-75 us (every GRF launch holds that group's next step back), both part streams on one
        queue 106-108 us -- slower than the single launch (88-91 us).  Which streams share a queue is not visible through the
        API, so it is measured: ipp_probe_stream_pair runs two chains of dependent 30-us launches, twice as long on a shared
        queue.  New streams are classed against one representative per queue found so far (a handful of probes, ~0.5 ms
        each) until there are enough queues; the part streams avoid the caller's queue when they can.
        IPP_PARTS_PROBE=0: streams as they come (A/B).  Returns (part streams | None, staging stream)."""
        torch = self.torch
        want_parts = n_parts if n_parts > 1 else 0
        self.queue_report = None
        if os.environ.get("IPP_PARTS_PROBE", "1") == "0" or not hasattr(self.engine._lib, "ipp_probe_stream_pair"):
            self._queues = None
            return ([torch.cuda.Stream(device=dev) for _ in range(want_parts)] or None), torch.cuda.Stream(device=dev)
        launches = 12
        main = torch.cuda.current_stream(dev)
        if not want_parts:
            # one launch per step on the caller's stream: the staging stream only has to stay off the caller's hardware queue -- one
            # probe per candidate instead of the classification below
            self._queues = None
            thr = max(launches * 0.030 * 1.5, 0.75 * self.engine.probe_stream_pair(main, main, launches))
            st = None
            for _ in range(4):
                st = torch.cuda.Stream(device=dev)
                if self.engine.probe_stream_pair(st, main, launches) <= thr:
                    break
            return None, st
        # "shared" = the two chains ran one after the other: calibrated on ONE stream paired with itself (2 x 12 launches in a row), so
        # that other work on the device while the env is built stretches both sides of the comparison (nominal: 0.72 ms -> 0.54 ms)
        shared = max(launches * 0.030 * 1.5, 0.75 * min(self.engine.probe_stream_pair(main, main, launches) for _ in range(2)))

        def same_queue(a, b):
            """Two chains of dependent launches on a and b: one after the other (shared queue) or side by side?  A result within
            15 % of the threshold is measured again (other work on the device stretches a probe)."""
            for _ in range(3):
                t = self.engine.probe_stream_pair(a, b, launches)
                if abs(t - shared) > 0.15 * shared:
                    break
            return t > shared

        reps, members = [main], [[]]  # queue 0 = the caller's
        pool = []
        while len(pool) < 16 and len(reps) - 1 < want_parts + 1:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                torch.zeros(16, device=dev).add_(1)  # first use binds the queue
            pool.append(st)
            for q, r in enumerate(reps):
                if same_queue(st, r):
                    members[q].append(st)
                    break
            else:
                reps.append(st)
                members.append([st])
        others = [m[0] for m in members[1:]]  # one stream per queue that is not the caller's
        picks = others[:want_parts]
        free = others[want_parts:]
        while len(picks) < want_parts:  # fewer queues than groups: share the caller's, then anything
            picks.append(members[0][0] if members[0] and members[0][0] not in picks else torch.cuda.Stream(device=dev))
        side = free[0] if free else (members[0][0] if members[0] and members[0][0] not in picks else
                                     (others[-1] if others and not want_parts else torch.cuda.Stream(device=dev)))
        # parts_distinct: every group got a hardware queue of its own (if not -- fewer queues than groups, GPU_MAX_HW_QUEUES -- the
        # partitioned schedule is slower than one launch per step: callers that only want throughput should step synchronously)
        self._queues = {"n_queues": len(reps), "pool": pool, "members": members, "probes_ms_shared": shared,
                        "parts_distinct": len(others) >= want_parts}
        queue_of = {id(st): q for q, ms in enumerate(members) for st in ms}
        # plain numbers for reports (bench.py puts them on its line): where the staging stream ended up decides whether a
        # ground-truth launch holds back a group's next step (43-47 M env-steps/s instead of 55-56 M at configs[1])
        self.queue_report = {"n_queues": len(reps), "parts_distinct": bool(len(others) >= want_parts),
                             "staging_shares_a_part_queue": bool(queue_of.get(id(side), -1) in {queue_of.get(id(p), -2) for p in picks}),
                             "staging_on_callers_queue": bool(queue_of.get(id(side), -1) == 0),
                             "probe_ms_shared": float(shared), "streams_probed": len(pool)}
        return (picks or None), side

    @property
    def part_queues_distinct(self) -> bool:
        """False when two part streams had to share a hardware queue (their launches then take turns: step_async still gives
        the same results, but slower than step())."""
        return self._queues is None or bool(self._queues.get("parts_distinct", True))

    def part_stream(self, p: int):
        """The stream that steps part p (run that part's policy on it and no event is needed around step_async)."""
        return self._part_streams[p]

    def part_envs(self, p: int):
        """Env indices of part p (device int64 tensor; fixed for the lifetime of the env)."""
        return self._part_envs[p]

    def step_async(self, actions, inputs_ready: bool = False):
        """
        One env step of the whole batch as `parts` launches, part p on part_stream(p); returns at once, nothing joins the
        streams: part p's results (reward / status entries of part_envs(p), its env slots) are complete behind
        part_stream(p) -- wait(p) makes the caller's stream wait for them.  Same results as step(), bit for bit.
        actions: [B, 3] float64 DEVICE tensor.  inputs_ready=False: the part streams first wait for the caller's stream at
        this point (actions written on it are seen); True: the caller vouches that actions[part_envs(p)] are complete for
        part_stream(p) (written on that stream, or synchronised earlier): no event at all.
        Lifetime: the engine keeps the tensors of the last calls alive until the part streams are past the launches that read
        them (IPPEngine.step_parts), so a fresh `actions` tensor per call may be dropped right after the call.
        """
        torch = self.torch
        if self.parts <= 1:
            raise RuntimeError("step_async needs VecIPPEnv(parts > 1, stagger=True) on a fused-step engine")
        if not (self._fused_reset or self._reset_ids_by_phase is None):
            raise RuntimeError("step_async needs the scheduled resets inside the step launch (fused_reset)")
        a = actions if (torch.is_tensor(actions) and actions.dtype == torch.float64 and actions.is_cuda and actions.is_contiguous()) \
            else self.engine._dev(actions, torch.float64).reshape(-1, 3).contiguous()
        streams = self._part_streams
        prefetch = None
        if not inputs_ready or self._main_dirty:
            self._main_dirty = False
            self._ev_inputs.record(torch.cuda.current_stream(self.device))
            for st in streams:
                st.wait_event(self._ev_inputs)
        scheduled = None
        blk = None
        if self._reset_ids_by_phase is not None:
            K = self._blk_K
            b, j = divmod(self.t, K)
            set_ = b % 2
            if self._blk_tag[set_] != b:
                self._stage_block(b)
                self._blk_waited = -1
            if self._blk_waited != b:
                for st in streams:
                    st.wait_event(self._blk_ready[set_])
                self._blk_waited = b
            p = self._phase_ending_at(self.t)
            n = int(self._reset_ids_by_phase[p].numel())
            scheduled = (p, set_ * K + j, n) if n > 0 else None
            blk = (set_, j == K - 1)
            prefetch = (b, j)
        # (the caller's stream is looked up only when a noise ring is refilled: torch.cuda.current_stream costs ~5 us)
        nz = self._noise_plane_parts(None if self._noise_pos else torch.cuda.current_stream(self.device), streams)
        fused = {}
        if scheduled is not None:
            p, k, n = scheduled
            fused = dict(reset_src=self._reset_src(p), reset_gt=None if self._blk_alt[k // self._blk_K] else self._staged_field(k)[:n], init_action=INIT_ACTION)
            if self.shuffle_prior_cov:
                self.engine.set_reset_prior(self._prior_scale_scheduled(p))
        self.engine.set_item_order(self._orders_parts[self.t % self.episode_steps])
        self.engine.step_parts(a, self.prev, nz, self._flags | _ffi.IPP_UPDATE_PREV, self.reward, self.status, self._part_begin, streams, **fused)
        if scheduled is not None:
            self.episode[self._reset_ids_host[scheduled[0]]] += 1
        self._noise_done_parts(streams)
        self.t += 1
        if blk is not None and blk[1]:
            for q, st in enumerate(streams):
                self._blk_free[blk[0]][q].record(st)
        self._async_pending = True
        if prefetch is not None:
            self._prefetch_next_block(*prefetch)

    def _noise_plane_parts(self, main, streams):
        """Measurement noise of this step on a partitioned batch: plane `pos` of ring (fill index % 2); the other ring is
        refilled on the caller's stream one ring ahead.  streams: the part streams that will read it (None: the caller's).
        Same planes as the single ring of parts == 1 (plane p of fill f = step f * NOISE_RING + p)."""
        pos = self._noise_pos
        if pos == 0:
            f = self._ring_cur = self._noise_fills
            self._noise_fills += 1
            r = f % 2
            for want, ring in ((f, r), (f + 1, 1 - r)):
                if self._ring_filled[ring] != want:
                    for ev in self._ring_free[ring]:  # (recorded behind the last step that read this ring)
                        main.wait_event(ev)
                    self.engine.normal_rows(self._noise_rings[ring], self.engine.meas_cap, self.seed,
                                            self.NOISE_STREAM + want * self.NOISE_RING, row_offset=self.env_id_offset)
                    self._ring_ready[ring].record(main)
                    self._ring_filled[ring] = want
            if streams is not None:
                for st in streams:
                    st.wait_event(self._ring_ready[r])
        return self._noise_rings[self._ring_cur % 2][pos]

    def _noise_done_parts(self, streams):
        self._noise_pos = (self._noise_pos + 1) % self.NOISE_RING
        if self._noise_pos == 0:
            main = self.torch.cuda.current_stream(self.device)
            for q in range(self.parts):
                self._ring_free[self._ring_cur % 2][q].record(streams[q] if streams is not None else main)

    def wait(self, part: Optional[int] = None):
        """The caller's (current) stream waits for everything issued on part_stream(part) (all parts when None)."""
        if self.parts <= 1 or not self._async_pending:
            return
        main = self.torch.cuda.current_stream(self.device)
        for q in (range(self.parts) if part is None else [part]):
            self._part_done[q].record(self._part_streams[q])
            main.wait_event(self._part_done[q])
        if part is None:
            self._async_pending = False

    def close(self):
        """Joins the part streams and the staging stream, then releases the engine (and with it the arena): nothing of this env may
        still be running on a stream the allocator does not know about when the memory goes back."""
        if getattr(self, "engine", None) is None:
            return
        try:
            self.wait()
            for st in (self._part_streams or []):
                st.synchronize()
            if self._side is not None:
                self._side.synchronize()
        except Exception:
            pass
        self.engine.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ views
    def mean(self, env):
        self.wait()
        return self.engine.read_mean(env)

    def diag(self, env):
        self.wait()
        return self.engine.read_diag(env)

    def ground_truth(self, env):
        self.wait()
        return self.engine.read_gt(env)

    def covariance(self, env):
        self.wait()
        return self.engine.read_cov(env)

    def metrics(self, env_ids=None):
        self.wait()
        return self.engine.metrics(env_ids=env_ids)
