"""Actor process: environment loop, exploration, n-step assembly, checkpoints.

Roles are the reference's (``models/agent.py:12-184``):

* ``exploration`` agents act with the online policy plus OU noise and are the
  only producers of replay data;
* the ``exploitation`` agent (agent 0) acts greedily with the TARGET policy,
  logs the evaluation reward and is the only process that writes actor
  checkpoints (periodically, or when the episode reward beats the best by
  ``save_reward_threshold``).

Differences by design: an agent steps ``envs_per_agent`` environments at once
with batched policy inference; transitions leave through a lossless
shared-memory ring in the replay row format; weights arrive through the
seq-locked slab (every agent sees every published version); the stop flag ends
the loop (the reference runs every agent to 1M env steps regardless,
``models/agent.py:60``); stored ``next_state`` is normalised like ``state``.
"""
from __future__ import annotations

import os
import time
from typing import Dict, Optional

import numpy as np

from ..envs import create_env_wrapper
from ..parallel.ipc import ControlBlock, ShmRing, SlabReader, WeightSlab
from ..utils.checkpoint import actor_checkpoint_path, save_actor
from ..utils.logger import Logger
from .native import CpuActor, NStepAssembler, OUNoise


def make_policy(config: Dict, device: str):
    """CPU agents use the native C++ forward; GPU agents the HIP inference path."""
    scale = float(config["action_high"]) if config.get("scale_action") else 1.0
    if str(device).startswith("cuda"):
        from .gpu_actor import GpuActor
        return GpuActor(config["state_dim"], config["action_dim"], config["dense_size"],
                        max_batch=max(1, int(config.get("envs_per_agent", 1))), device=device, action_scale=scale)
    return CpuActor(config["state_dim"], config["action_dim"], config["dense_size"], action_scale=scale)


class Agent:
    def __init__(self, config: Dict, policy, ctrl: ControlBlock, n_agent: int = 0,
                 agent_type: str = "exploration", log_dir: str = "", ring: Optional[ShmRing] = None,
                 slab: Optional[WeightSlab] = None, seed: Optional[int] = None, hb_slot: Optional[int] = None):
        self.config = config
        self.n_agent = n_agent
        self.agent_type = agent_type
        self.max_steps = int(config["max_ep_length"])
        self.num_episode_save = int(config["num_episode_save"])
        self.ctrl = ctrl
        self.ring = ring
        self.local_episode = 0
        self.log_dir = log_dir
        self.E = int(config.get("envs_per_agent", 1))
        seed = int(config["random_seed"]) + 1000 * (n_agent + 1) if seed is None else seed
        self.env_wrapper = create_env_wrapper(config, self.E, seed=seed)
        self.ou_noise = OUNoise(self.E, config["action_dim"], config["action_low"], config["action_high"],
                                theta=config["ou_theta"], mu=config["ou_mu"], max_sigma=config["ou_max_sigma"],
                                min_sigma=config["ou_min_sigma"], decay_period=config["ou_decay_period"],
                                seed=seed + 17)
        self.nstep = NStepAssembler(self.E, config["n_step_returns"], config["state_dim"], config["action_dim"],
                                    config["discount_rate"])
        self.actor = policy
        self.reader = SlabReader(slab) if slab is not None else None
        # slab layout: [online actor | target actor]; the exploiter follows the target
        self._half = 1 if agent_type == "exploitation" else 0
        self.logger = Logger(os.path.join(log_dir, f"agent-{n_agent}"), config.get("log_backend", "tb")) if log_dir else None
        self.best_reward = -float("inf")
        self.episode_rewards = []
        self.env_steps = 0
        self.weight_versions_seen = 0
        self.hb_slot = n_agent + 1 if hb_slot is None else int(hb_slot)
        self._gpu_version = 0
        self.shared_weights = False         # True once the policy follows the learner's device buffer (HIP IPC)

    # ------------------------------------------------------------------ weights
    def _try_attach_shared(self) -> None:
        """GPU agent on the learner's machine: map the learner's weight buffer (``parallel/hip_ipc.py``), local
        memory on the learner's own GPU, peer memory over xGMI on another one."""
        self._attach_tried = True
        if not (hasattr(self.actor, "attach_shared") and self.config.get("share_weights_ipc", True)):
            return
        try:
            from ..parallel.hip_ipc import SharedDeviceBuffer, read_handle
            got = read_handle(self.ctrl.name)
            if got is None:
                self._attach_tried = False           # the learner has not exported yet: look again next time
                return
            handle, numel, dev_index = got
            mine = self.actor.device.index or 0
            if mine != dev_index:
                # An agent on ANOTHER GPU of the machine maps the same buffer as peer memory: its pack kernel then
                # reads the learner's weights over xGMI, still without a copy through the host.  Only where the
                # runtime reports peer access between the two devices; the host slab otherwise.
                import torch
                if not torch.cuda.can_device_access_peer(mine, dev_index):
                    return
            shared = SharedDeviceBuffer.open(handle, numel, self.actor.device)
            if mine != dev_index:
                shared.tensor[:1].clone()            # a first read through the mapping: fails here, not in the rollout
            self.actor.attach_shared(shared, self._half)
            self.shared_weights = True
            if self.logger is not None:
                self.logger.scalar_summary("agent/shared_device_weights", 1.0, 0)
        except Exception as exc:      # pragma: no cover - depends on the host's IPC support
            print(f"[agent {self.n_agent}] device weight sharing unavailable ({exc}); using the host slab", flush=True)

    def update_actor_learner(self) -> bool:
        """Pull the newest published weights (``models/agent.py:37-49``)."""
        if not getattr(self, "_attach_tried", False):
            self._try_attach_shared()
        if self.shared_weights:
            v = self.ctrl.load(ControlBlock.GPU_WEIGHT_VERSION)
            if v == self._gpu_version:
                return False
            for _ in range(3):
                self.actor.refresh_shared(v)      # (synchronises: the pack has read the slot when it returns)
                # The slot of publication v is rewritten by publication v + 2, whose copies are enqueued before the
                # version counter moves: the pack was overtaken iff that publication had been ISSUED by now.  Two
                # more attempts at most: past that a policy torn between consecutive versions is accepted, as the
                # reference's exploiter does on every step (unsynchronised shared memory).
                if self.ctrl.load(ControlBlock.GPU_WEIGHT_ISSUED) < v + 2 or self.ctrl.stop:
                    break
                v = max(v, self.ctrl.load(ControlBlock.GPU_WEIGHT_VERSION))
            self._gpu_version = v
            self.weight_versions_seen += 1
            return True
        if self.reader is None:
            return False
        if not self.reader.poll():
            return False
        n = self.reader.buf.size // 2
        self.actor.load_flat(self.reader.buf[self._half * n:(self._half + 1) * n])
        self.weight_versions_seen += 1
        return True

    # ---------------------------------------------------------------------- run
    def run(self) -> None:
        cfg, E = self.config, self.E
        env, ctrl = self.env_wrapper, self.ctrl
        explore = self.agent_type == "exploration"
        max_env_steps = int(cfg["max_env_steps_per_agent"])
        push_timeout = 10_000 if cfg.get("lossless_replay_queue", True) else 0
        self.update_actor_learner()
        obs = env.reset()
        self.ou_noise.reset()
        ep_reward = np.zeros(E, dtype=np.float64)
        ep_steps = np.zeros(E, dtype=np.int64)
        ep_start = np.full(E, time.time())
        print_mark = 0
        t_start = time.time()
        while not ctrl.stop and self.env_steps < max_env_steps:
            action = np.asarray(self.actor.act(obs), dtype=np.float32).reshape(E, -1)
            if explore:
                action = self.ou_noise.get_action(action, ep_steps)
            next_obs, reward, done = env.step(action)
            self.env_steps += E
            ep_reward += reward
            ep_steps += 1
            trunc = (ep_steps >= self.max_steps) & ~done
            rows = self.nstep.push(env.normalise_state(obs), action, env.normalise_reward(reward),
                                   env.normalise_state(next_obs), done, trunc)
            if explore and self.ring is not None and rows.shape[0]:
                pushed = self.ring.push(rows, timeout_ms=push_timeout)
                if pushed < rows.shape[0] and ctrl.stop:
                    break
            ctrl.add(ControlBlock.ENV_STEPS, E)
            obs = next_obs
            finished = np.nonzero(done | trunc)[0]
            for e in finished:
                self._end_episode(int(e), float(ep_reward[e]), time.time() - ep_start[e])
                obs[e] = env.reset_env(int(e))
                self.ou_noise.reset(int(e))
                ep_reward[e], ep_steps[e], ep_start[e] = 0.0, 0, time.time()
            if self.env_steps // 5000 >= print_mark and self.episode_rewards:
                print_mark = self.env_steps // 5000 + 1
                print(f"[agent {self.n_agent}] reward {self.episode_rewards[-1]:.2f} env_steps {self.env_steps} "
                      f"time {int(time.time() - t_start)}s", flush=True)
            ctrl.heartbeat(self.hb_slot, int(time.time() * 1000))
        if self.ring is not None:
            self.ring.mark_closed()
        if self.logger is not None:
            self.logger.close()
        print(f"Agent {self.n_agent} done.", flush=True)

    def _end_episode(self, e: int, episode_reward: float, seconds: float) -> None:
        cfg = self.config
        self.local_episode += 1
        self.ctrl.add(ControlBlock.GLOBAL_EPISODE, 1)
        step = self.ctrl.update_step
        if self.logger is not None:
            self.logger.scalar_summary("agent/reward", episode_reward, step)
            self.logger.scalar_summary("agent/episode_timing", seconds, step)
        self.episode_rewards.append(episode_reward)
        if self.local_episode % 100 == 0:
            print(f"Agent: {self.n_agent}  episode {self.local_episode}", flush=True)
        outperformed = episode_reward - self.best_reward > cfg["save_reward_threshold"]
        time_to_save = self.local_episode % self.num_episode_save == 0
        if self.n_agent == 0 and (time_to_save or outperformed):
            if episode_reward > self.best_reward:
                self.best_reward = episode_reward
                print("best_reward", self.best_reward, flush=True)
            self.save(f"local_episode_{self.local_episode}_reward_{self.best_reward:4f}")
        # explorers refresh every update_agent_ep episodes; the exploiter tracks the target continuously
        if self.agent_type != "exploration" or self.local_episode % int(cfg["update_agent_ep"]) == 0:
            if not self.ctrl.stop:
                self.update_actor_learner()

    # --------------------------------------------------------------------- save
    def save(self, checkpoint_name: str) -> str:
        path = os.path.join(self.log_dir, f"agent_{self.n_agent}", checkpoint_name + ".pt")
        meta = {"env": self.config["env"], "model": self.config["model"], "local_episode": self.local_episode,
                "best_reward": float(self.best_reward), "update_step": int(self.ctrl.update_step),
                "state_dim": self.config["state_dim"], "action_dim": self.config["action_dim"],
                "dense_size": self.config["dense_size"], "policy": "target_actor"}
        return save_actor(path, self.actor.state_dict(), meta)

    def save_replay_gif(self, output_path: str, env_index: int = 0) -> str:
        """Roll the greedy policy once and write an animated GIF
        (``models/agent.py:163-184`` + ``utils/utils.py:37-52``)."""
        from ..utils.misc import frames_to_gif
        env = create_env_wrapper(self.config, 1, seed=0)
        state = env.reset()
        frames = []
        for _ in range(self.max_steps):
            action = np.asarray(self.actor.act(state), dtype=np.float32).reshape(1, -1)
            state, _, done = env.step(action)
            frames.append(env.render(0))
            if done[0]:
                break
        return frames_to_gif(frames, output_path)
