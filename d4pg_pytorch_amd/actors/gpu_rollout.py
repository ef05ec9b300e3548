"""Explorers resident on the learner's GPU: policy, exploration noise, environment, n-step assembly and
replay insertion without a host hop.

The reference's explorers are CPU processes that push one pickled transition per environment step through a
pipe (``models/agent.py:51-154``).  For environments that exist as device kernels (``envs/device_pendulum.py``)
the whole loop runs on the GPU instead: a *segment* of ``T`` steps of ``E`` environments is

    T x [ actor_infer32 (online actor; OU noise and the clip in its last epilogue) -> environment step ]
    -> n-step targets of the [T, E] segment (``csrc/nstep.hip``)
    -> replay rows (``rollout_rows_kernel``)

i.e. TWO launches per environment step: the Philox position of the noise is the environment's step counter, which the
environment's kernel advances itself (``envs/device_pendulum.py::StepCounter``).  ``fuse_noise=False`` keeps noise
and counter in launches of their own (four per step), for the test that both orders of work produce the same rows.

captured once as a HIP graph and replayed; the rows then enter the HBM replay with one ``replay_insert``
launch on the learner's stream.  The learner's update occupies a fraction of the 256 CUs, so the segment
graph runs beside it on its own stream.

n-step windows do not cross a segment boundary: the last ``n - 1`` steps of a segment bootstrap after fewer
steps (their own ``gamma^k``), exactly like the steps before an episode end (``models/agent.py:109-123``).
"""
from __future__ import annotations

import gc
from typing import Dict, Optional, Tuple

import torch

from ..ops import hip as H
from .gpu_actor import PackedActor32


class DeviceRollout:
    def __init__(self, cfg: Dict, env, replay, device="cuda", segment: int = 32, seed: int = 0,
                 explore: bool = True, use_graph: bool = True, fuse_noise: bool = True):
        self.cfg, self.env, self.replay = cfg, env, replay
        self.device = torch.device(device)
        self.E, self.T = env.E, int(segment)
        self.S, self.A = int(cfg["state_dim"]), int(cfg["action_dim"])
        assert env.state_dim == self.S and env.action_dim == self.A
        if not PackedActor32.supported(self.S, self.A, int(cfg["dense_size"])):
            raise ValueError("the device rollout needs a policy the single-launch inference kernel supports (<= 64 actions)")
        self.actor = PackedActor32(self.S, self.A, int(cfg["dense_size"]), self.device, rows=self.E)
        self.explore = bool(explore)
        self.n, self.gamma = int(cfg["n_step_returns"]), float(cfg["discount_rate"])
        if self.explore and float(cfg["ou_min_sigma"]) != float(cfg["ou_max_sigma"]):
            # the host agents anneal sigma over ou_decay_period steps of an episode (utils/utils.py:32 of the
            # reference); the noise kernel here takes one sigma, so a schedule would silently train differently
            raise ValueError("device explorers draw OU noise with a constant sigma: set ou_min_sigma == ou_max_sigma "
                             "(the reference's own setting) or use host explorers")
        if int(segment) < 4 * self.n:
            import warnings
            warnings.warn(f"gpu_rollout_segment={segment} with n_step_returns={self.n}: the last n - 1 steps of every "
                          f"segment bootstrap early, i.e. {100.0 * (self.n - 1) / int(segment):.0f} % of the rows "
                          f"(use a segment of at least {4 * self.n} steps to keep that below 25 %)", stacklevel=2)
        self.action_scale = float(cfg["action_high"]) if cfg.get("scale_action") else 1.0
        T, E, S = self.T, self.E, self.S
        self.lda = 4 * ((self.A + 3) // 4)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)   # noqa: E731
        self.obs = z(T + 1, E, S)              # obs[t]: what the policy saw at step t; obs[T] seeds the next segment
        self.next_obs = z(T, E, S)
        self.act_raw = z(E, self.lda)
        self.act = z(T, E, self.lda)
        self.reward, self.done, self.trunc = z(T, E), z(T, E), z(T, E)
        self.R, self.gk, self.d = z(T, E), z(T, E), z(T, E)
        self.nxt = z(T, E, dt=torch.int32)
        self.rows = z(T * E, replay.row_ld)
        self.ou_x = z(E, self.lda)
        self.ou_counter = z(1, dt=torch.int64)
        env.noise = self.ou_x              # a new episode starts from zero noise (utils/utils.py:21-22 reset())
        self.fuse_noise = bool(fuse_noise) and hasattr(getattr(env, "counter", None), "paired")
        if self.fuse_noise:
            env.counter.paired = True      # the environment's kernel advances the counter: no launch for it
        #: kernel nodes of the captured segment graph: T x (policy, [noise,] environment [, counter]) + n-step targets
        #: + rows (the first observations move in a copy node)
        self.graph_kernels = 0
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.stream = torch.cuda.Stream(device=self.device)
        self.done_event = torch.cuda.Event()
        self.use_graph = bool(use_graph)
        self._graph = None
        self.env_steps = 0
        self._in_flight = self._started = False
        with torch.cuda.stream(self.stream):
            self.env.reset(self.obs[0])
        self.stream.synchronize()

    # ------------------------------------------------------------------ weights
    def refresh_weights(self, flat_ptr: int, actor_base: int = 0) -> None:
        """Repack the policy from flat fp32 parameters on this device.  Ordered after the caller's stream,
        ahead of the next segment."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            self.actor.pack_from(flat_ptr, actor_base)

    # ------------------------------------------------------------------ segment
    def _ou(self, t: int) -> None:
        cfg = self.cfg
        a = H.OuArgs()
        a.x, a.action, a.out = self.ou_x.data_ptr(), self.act_raw.data_ptr(), self.act[t].data_ptr()
        # the Philox position of the noise is the environment's own step counter where it has one (advanced by the
        # environment's kernel once per step; different stream constants and seeds keep the two draws apart)
        shared = getattr(self.env, "counter", None)
        a.sigma, a.seed = None, self.seed
        a.counter = shared.ptr if shared is not None else self.ou_counter.data_ptr()
        a.n, a.A = self.E * self.lda, self.lda
        a.theta, a.mu = float(cfg["ou_theta"]), float(cfg["ou_mu"])
        a.sigma_const = float(cfg["ou_max_sigma"]) if self.explore else 0.0
        a.low, a.high = float(cfg["action_low"]), float(cfg["action_high"])
        H.launch("d4pg_ou_noise_shared_counter" if shared is not None else "d4pg_ou_noise", a)

    def _infer_with_noise(self, t: int) -> None:
        cfg = self.cfg
        a = self.actor.infer_args(self.obs[t], self.act[t], self.E, self.action_scale)
        a.ou.x, a.ou.counter, a.ou.seed = self.ou_x.data_ptr(), self.env.counter.ptr, self.seed
        a.ou.theta, a.ou.mu = float(cfg["ou_theta"]), float(cfg["ou_mu"])
        a.ou.sigma = float(cfg["ou_max_sigma"]) if self.explore else 0.0
        a.ou.low, a.ou.high = float(cfg["action_low"]), float(cfg["action_high"])
        self.actor.infer(a, self.E)

    def _enqueue_segment(self) -> None:
        T, E = self.T, self.E
        self.obs[0].copy_(self.obs[T])
        if self.fuse_noise:
            self.env.counter.rewind()          # nothing in steady state: every segment ends on the slot it started from
        for t in range(T):
            if self.fuse_noise:
                self._infer_with_noise(t)
            else:
                self.actor.infer(self.actor.infer_args(self.obs[t], self.act_raw, E, self.action_scale), E)
                self._ou(t)
            self.env.step(self.act[t], self.obs[t + 1], self.next_obs[t], self.reward[t], self.done[t], self.trunc[t])
        if self.fuse_noise:
            self.env.counter.rewind()          # one tiny launch per SEGMENT if T is odd
        na = H.NstepArgs()
        na.reward, na.done, na.trunc = self.reward.data_ptr(), self.done.data_ptr(), self.trunc.data_ptr()
        na.R, na.gk, na.nxt, na.d = self.R.data_ptr(), self.gk.data_ptr(), self.nxt.data_ptr(), self.d.data_ptr()
        na.T, na.E, na.n, na.gamma = T, E, self.n, self.gamma
        H.launch("d4pg_nstep", na)
        ra = H.RolloutRowsArgs()
        ra.obs, ra.act, ra.next_obs = self.obs.data_ptr(), self.act.data_ptr(), self.next_obs.data_ptr()
        ra.R, ra.gk, ra.nxt, ra.d, ra.rows = (self.R.data_ptr(), self.gk.data_ptr(), self.nxt.data_ptr(),
                                              self.d.data_ptr(), self.rows.data_ptr())
        ra.T, ra.E, ra.S, ra.A, ra.ld_act, ra.row_ld = T, E, self.S, self.A, self.lda, self.replay.row_ld
        H.launch("d4pg_rollout_rows", ra)

    def _capture(self) -> None:
        # the first segment starts from the reset observations, which collect() parks in obs[T]
        with torch.cuda.stream(self.stream):
            self._enqueue_segment()                 # warm-up (module load); its rows are real and are inserted
        self.stream.synchronize()
        gc.collect()
        was = gc.isenabled()
        gc.disable()                                # a collected tensor would be freed during capture
        try:
            g = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                self._enqueue_segment()
            self.graph_kernels = H.graph_kernel_nodes(g)
            g.instantiate()
            self._graph = g
        finally:
            if was:
                gc.enable()

    def launch(self, flat_ptr: Optional[int] = None, actor_base: int = 0) -> None:
        """Enqueue one segment on the rollout's stream, ordered behind what the caller's stream holds now (the
        insert of the previous segment's rows, a consistent set of weights).  With ``flat_ptr`` the policy is
        first repacked from those flat fp32 parameters (the learner's own buffer or an IPC mapping)."""
        assert not self._in_flight, "drain() the previous segment first"
        main = torch.cuda.current_stream(self.device)
        if not self._started:
            self.obs[self.T].copy_(self.obs[0])     # a segment starts from obs[T] of the one before
            self._started = True
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)
        if flat_ptr is not None:
            with torch.cuda.stream(self.stream):
                self.actor.pack_from(flat_ptr, actor_base)
        if self.use_graph and self._graph is None:
            self._capture()                        # makes the first segment itself
        elif self.use_graph:
            with torch.cuda.stream(self.stream):
                self._graph.replay()
        else:
            with torch.cuda.stream(self.stream):
                self._enqueue_segment()
        self.done_event.record(self.stream)
        self._in_flight = True

    def drain(self, insert: bool = True) -> int:
        """The caller's stream waits (on the device) for the segment in flight and inserts its rows into the
        replay.  Returns the environment steps they hold."""
        if not self._in_flight:
            return 0
        torch.cuda.current_stream(self.device).wait_event(self.done_event)
        if insert:
            self.replay.insert(self.rows)
        self._in_flight = False
        self.env_steps += self.T * self.E
        return self.T * self.E

    def collect(self, insert: bool = True) -> int:
        """One segment start to finish: ``T * E`` environment steps and as many replay rows."""
        self.launch()
        return self.drain(insert)

    def collect_overlapped(self, flat_ptr: Optional[int] = None, actor_base: int = 0) -> int:
        """Insert the rows of the segment launched by the previous call and launch the next one, which then
        runs on its own stream beside whatever the caller enqueues next (the learner's updates leave most of
        the 256 CUs idle).  Returns the environment steps inserted by this call."""
        n = self.drain()
        self.launch(flat_ptr, actor_base)
        return n

    @property
    def launches_per_step(self) -> float:
        """Kernel launches per environment step of a replayed segment, the per-segment launches (n-step targets, rows,
        a counter rewind after an odd number of steps) spread over its steps."""
        return self.graph_kernels / self.T

    def episode_stats(self) -> Tuple[Optional[float], int]:
        return self.env.episode_stats()
