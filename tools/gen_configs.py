#!/usr/bin/env python
"""Writes ``configs/*.yml``: one file per (environment, algorithm) pair the
reference ships (10 environments x {d4pg, d3pg, ddpg}, same file names, same
flat keys and values, so either code base can read them) plus ``configs/mi355x/``
with the MI355X-specific stress configurations.  The tables below are the only
source; edit them and re-run instead of editing the generated files."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "configs")

# file stem, env id, state_dim, action_dim, (v_min, v_max) for d4pg
ENVS = [
    ("pendulum", "Pendulum-v0", 3, 1, (-1000.0, 0.0)),
    ("lunar_lander_continuous", "LunarLanderContinuous-v2", 8, 2, (-200.0, 200.0)),
    ("bipedal", "BipedalWalker-v2", 24, 4, (-100.0, 300.0)),
    ("inverted_pendulum", "InvertedPendulum-v2", 4, 1, (0.0, 1000.0)),
    ("inverted_double_pendulum", "InvertedDoublePendulum-v2", 11, 1, (0.0, 10000.0)),
    ("reacher", "Reacher-v2", 11, 2, (-12.0, -3.0)),
    ("hopper", "Hopper-v2", 11, 3, (0.0, 3500.0)),
    ("walker2d", "Walker2d-v2", 17, 6, (0.0, 5000.0)),
    ("half_cheetah", "HalfCheetah-v2", 17, 6, (0.0, 10000.0)),
    ("ant", "Ant-v2", 111, 8, (0.0, 4000.0)),
]
ACTION_RANGE = {("pendulum", "d4pg"): 2, ("pendulum", "d3pg"): 2, ("pendulum", "ddpg"): 2,
                ("ant", "d3pg"): 2, ("ant", "ddpg"): 2}
# n_step_returns / max_ep_length of the scalar-critic variants
N_STEP_1 = {"pendulum", "lunar_lander_continuous", "half_cheetah", "ant"}
EP_400 = {"pendulum", "lunar_lander_continuous", "ant"}

TEMPLATE = """# {title}
# Flat key/value file; every key is optional (defaults in d4pg_pytorch_amd/config.py).

# --- environment ---
env: {env}
state_dim: {S}
action_dim: {A}
action_low: {lo}
action_high: {hi}
num_agents: {agents}          # agent 0 exploits (target policy), the others explore
random_seed: 2019

# --- training ---
model: {model}
batch_size: 256
num_steps_train: 100_000      # learner updates
max_ep_length: {ep}           # steps before an episode is truncated
replay_mem_size: 1000000      # replay capacity (ring, oldest rows overwritten)
{prio}discount_rate: 0.99
n_step_returns: {n}
update_agent_ep: 1            # explorers refresh their weights every k episodes
replay_queue_size: 64
batch_queue_size: 64
{per}num_episode_save: 100
device: cuda                  # learner (and exploiter) device
agent_device: cpu             # explorer device
{sbod}save_reward_threshold: 1      # save when the evaluation reward beats the best by more than this

# --- networks ---
critic_learning_rate: 0.0005
actor_learning_rate: 0.0005
dense_size: 400               # width of both hidden layers
final_layer_init: 0.003
{dist}tau: 0.001                    # Polyak coefficient of the target networks

# --- misc ---
results_path: results
"""


def render(stem, env, S, A, vr, model):
    rng = ACTION_RANGE.get((stem, model), 1)
    d4 = model == "d4pg"
    n = 5 if d4 else (1 if stem in N_STEP_1 or (model == "ddpg" and stem == "bipedal") else 5)
    ep = 1000 if d4 else (400 if stem in EP_400 else 1000)
    agents = 2 if model == "ddpg" else 4
    # Two deliberate corrections of the reference's files: its walker2d_d4pg.yml has num_agents: 1
    # (no explorer -> no data) and its hopper_d4pg.yml has state_dim: 1 (the observation has 11
    # entries).  The loader also repairs both when it reads the reference's own files.
    S_out = S
    dist = ("num_atoms: 51                 # atoms of the categorical value distribution\n"
            f"v_min: {vr[0]}\nv_max: {vr[1]}\n") if d4 else ""
    return TEMPLATE.format(title=f"{env} / {model.upper()}", env=env, S=S_out, A=A, lo=-rng, hi=rng, agents=agents,
                           model=model, ep=ep, n=n, per="replay_memory_prioritized: 0\n" if d4 else "",
                           sbod="save_buffer_on_disk: 0\n" if d4 else "", dist=dist,
                           prio=("priority_alpha: 0.6           # 0 = uniform, 1 = fully proportional\n"
                                 "priority_beta_start: 0.4      # importance exponent, annealed linearly ...\n"
                                 "priority_beta_end: 1.0        # ... to this value over num_steps_train\n") if d4 else "")


MI355X = {
    "lunar_lander_d4pg_1gpu.yml": """# LunarLanderContinuous-shape D4PG, one MI355X learner, HBM replay, HIP-graph update.
env: LunarLanderContinuous-v2
model: d4pg
num_agents: 4
batch_size: 256
n_step_returns: 5
v_min: -200.0
v_max: 200.0
device: cuda
replay_device: hbm
compute_dtype: fp32       # reference precision: fused row-local pipeline on exact f32 MFMA (the default); 'bf16' = the same
                          # pipeline with bf16 MFMA operands, 1.5x (batch 256) to 3x (batch 4096) faster, opt-in
use_hip_graph: true
""",
    "bipedal_d4pg_per_8explorers.yml": """# BipedalWalker-shape D4PG with prioritised replay: the sum-tree lives in HBM, 8 CPU explorers feed it.
env: BipedalWalker-v2
model: d4pg
num_agents: 9
replay_memory_prioritized: 1
priority_alpha: 0.6
priority_beta_start: 0.4
priority_beta_end: 1.0
v_min: -100.0
v_max: 300.0
device: cuda
replay_device: hbm
compute_dtype: fp32       # reference precision: fused row-local pipeline on exact f32 MFMA (the default); 'bf16' = the same
                          # pipeline with bf16 MFMA operands, 1.5x (batch 256) to 3x (batch 4096) faster, opt-in
""",
    "walker2d_d4pg_dp8.yml": """# Walker2d D4PG, 8 data-parallel learners (one per MI355X) over RCCL, replicated update on the all-gathered batch.
env: Walker2d-v2
model: d4pg
num_agents: 3
num_learners: 8
batch_size: 256          # per learner rank
v_min: 0.0               # the range discounted returns take on the native task (gamma 0.99)
v_max: 400.0
device: cuda
replay_device: hbm
compute_dtype: fp32       # reference precision: fused row-local pipeline on exact f32 MFMA (the default); 'bf16' = the same
                          # pipeline with bf16 MFMA operands, 1.5x (batch 256) to 3x (batch 4096) faster, opt-in
""",
    "half_cheetah_d4pg_dp8.yml": """# HalfCheetah D4PG, 8 data-parallel learners.
env: HalfCheetah-v2
model: d4pg
num_agents: 3
num_learners: 8
batch_size: 256
n_step_returns: 5
v_min: -50.0             # the range discounted returns take on the native task (gamma 0.99)
v_max: 600.0
device: cuda
replay_device: hbm
compute_dtype: fp32       # reference precision: fused row-local pipeline on exact f32 MFMA (the default); 'bf16' = the same
                          # pipeline with bf16 MFMA operands, 1.5x (batch 256) to 3x (batch 4096) faster, opt-in
""",
    "ant_stress_d4pg.yml": """# Synthetic Ant-shape stress: obs 111, act 8, batch 4096, 10M-transition HBM replay per rank.
env: Ant-v2
synthetic_env: true
model: d4pg
num_agents: 3
num_learners: 8
batch_size: 4096
replay_mem_size: 10000000
replay_memory_prioritized: 1
v_min: 0.0
v_max: 4000.0
device: cuda
replay_device: hbm
compute_dtype: fp32       # the reference's precision (models/d4pg/d4pg.py:73-77); ant_stress_d4pg_bf16.yml is the bf16-operand sibling
envs_per_agent: 64
""",
    "ant_stress_d4pg_bf16.yml": """# bf16-operand sibling of ant_stress_d4pg.yml.  Synthetic Ant-shape stress: obs 111, act 8, batch 4096, 10M-transition HBM replay per rank.
env: Ant-v2
synthetic_env: true
model: d4pg
num_agents: 3
num_learners: 8
batch_size: 4096
replay_mem_size: 10000000
replay_memory_prioritized: 1
v_min: 0.0
v_max: 4000.0
device: cuda
replay_device: hbm
compute_dtype: bf16       # throughput stress: bf16 MFMA operands (fp32 masters / Adam / loss), lower precision than the reference
envs_per_agent: 64
""",
    "ant_stress_d3pg.yml": """# Same stress shape with the scalar (D3PG) critic path.
env: Ant-v2
synthetic_env: true
model: d3pg
num_agents: 3
num_learners: 8
batch_size: 4096
replay_mem_size: 10000000
n_step_returns: 1
device: cuda
replay_device: hbm
compute_dtype: fp32       # the reference's precision; set bf16 for the bf16-operand pipeline (lower precision than the reference)
envs_per_agent: 64
""",
    "pendulum_d4pg_device_explorers.yml": """# Pendulum D4PG with the explorers ON the learner's GPU: policy, OU noise, dynamics, n-step rows and replay insert
# are kernels (no CPU explorer process, no host hop); the GPU exploiter follows the learner's weights through HIP IPC.
env: Pendulum-v0
model: d4pg
num_agents: 1             # the exploiter only
action_low: -2
action_high: 2
v_min: -20.0              # support in the learner's units (reward / 100, gamma 0.99)
v_max: 0.0
device: cuda
replay_device: hbm
gpu_rollout_envs: 64      # environments stepped on the GPU
gpu_rollout_segment: 32   # steps per captured rollout segment
gpu_rollout_interval: 32  # learner updates between two segments (64 env steps per update)
num_steps_train: 50000
""",
    "walker2d_d4pg_device_explorers.yml": """# Walker2d D4PG with the explorers ON the learner's GPU (csrc/planar_env.hip: the articulated-body simulator as a
# kernel, one thread per environment); the GPU exploiter plays the host simulator (envs/planar.py) with the
# learner's target policy, read through HIP IPC.  (Hopper, HalfCheetah, BipedalWalker, Ant: the files next to this one.)
env: Walker2d-v2
model: d4pg
num_agents: 1             # the exploiter only
v_min: 0.0                # the range discounted returns take (gamma 0.99, reward per step up to about 4)
v_max: 400.0
device: cuda
replay_device: hbm
gpu_rollout_envs: 256     # environments stepped on the GPU
gpu_rollout_segment: 32   # steps per captured rollout segment
gpu_rollout_interval: 32  # learner updates between two segments (256 env steps per update)
num_steps_train: 200000
""",
    "walker2d_d4pg_rollout_workers_8gpu.yml": """# Walker2d D4PG on one 8-GPU node: the learner and its HBM replay on GPU 0, a rollout worker PROCESS on each of GPUs 1 - 7
# (actors/rollout_worker.py).  A worker steps its environments as kernels on its own GPU, packs the online actor
# straight out of the learner's HIP-IPC weight slots (peer memory over xGMI) and writes every segment's rows into a
# staging ring in the learner's HBM (parallel/row_mailbox.py); the learner's GPU runs updates only.  The GPU exploiter
# plays the host simulator with the target policy.  On fewer GPUs: --set rollout_devices=[1,2,3]; [0] runs the worker
# beside the learner on one GPU.
env: Walker2d-v2
model: d4pg
num_agents: 1             # the exploiter only
v_min: 0.0
v_max: 400.0
device: cuda
replay_device: hbm
rollout_devices: [1, 2, 3, 4, 5, 6, 7]
gpu_rollout_envs: 1024    # environments per worker
gpu_rollout_segment: 32   # steps per captured rollout segment (two launches per step)
updates_per_env_step: 0.002        # the learner never waits for data at 7 x 1024 environments; this only bounds it from above
max_restarts_per_process: 3
max_restarts_per_hour: 20
num_steps_train: 200000
""",
    "bipedal_d4pg_device_explorers.yml": """# BipedalWalker-v2 D4PG with the explorers ON the learner's GPU (the simulator as a kernel, one thread per environment); the
# GPU exploiter plays the host simulator with the learner's target policy, read through HIP IPC.
env: BipedalWalker-v2
model: d4pg
state_dim: 24
action_dim: 4
num_agents: 1             # the exploiter only
max_ep_length: 1600
v_min: -100.0             # the range discounted returns take (gamma 0.99)
v_max: 300.0
device: cuda
replay_device: hbm
gpu_rollout_envs: 256     # environments stepped on the GPU
gpu_rollout_segment: 32   # steps per captured rollout segment
gpu_rollout_interval: 32  # learner updates between two segments (256 env steps per update)
num_steps_train: 200000
""",
    "half_cheetah_d4pg_device_explorers.yml": """# HalfCheetah-v2 D4PG with the explorers ON the learner's GPU (the simulator as a kernel, one thread per environment); the
# GPU exploiter plays the host simulator with the learner's target policy, read through HIP IPC.
env: HalfCheetah-v2
model: d4pg
state_dim: 17
action_dim: 6
num_agents: 1             # the exploiter only
max_ep_length: 1000
v_min: -50.0             # the range discounted returns take (gamma 0.99)
v_max: 600.0
device: cuda
replay_device: hbm
gpu_rollout_envs: 256     # environments stepped on the GPU
gpu_rollout_segment: 32   # steps per captured rollout segment
gpu_rollout_interval: 32  # learner updates between two segments (256 env steps per update)
num_steps_train: 200000
""",
    "hopper_d4pg_device_explorers.yml": """# Hopper-v2 D4PG with the explorers ON the learner's GPU (the simulator as a kernel, one thread per environment); the
# GPU exploiter plays the host simulator with the learner's target policy, read through HIP IPC.
env: Hopper-v2
model: d4pg
state_dim: 11
action_dim: 3
num_agents: 1             # the exploiter only
max_ep_length: 1000
v_min: 0.0             # the range discounted returns take (gamma 0.99)
v_max: 400.0
device: cuda
replay_device: hbm
gpu_rollout_envs: 256     # environments stepped on the GPU
gpu_rollout_segment: 32   # steps per captured rollout segment
gpu_rollout_interval: 32  # learner updates between two segments (256 env steps per update)
num_steps_train: 200000
""",
    "ant_d4pg_device_explorers.yml": """# Ant-v2 D4PG with the explorers ON the learner's GPU (the simulator as a kernel, one thread per environment); the
# GPU exploiter plays the host simulator with the learner's target policy, read through HIP IPC.
env: Ant-v2
model: d4pg
state_dim: 111
action_dim: 8
num_agents: 1             # the exploiter only
max_ep_length: 1000
v_min: 0.0             # the range discounted returns take (gamma 0.99)
v_max: 300.0
device: cuda
replay_device: hbm
gpu_rollout_envs: 256     # environments stepped on the GPU
gpu_rollout_segment: 32   # steps per captured rollout segment
gpu_rollout_interval: 32  # learner updates between two segments (256 env steps per update)
num_steps_train: 200000
""",
    "pendulum_d4pg_cpu.yml": """# Pendulum D4PG entirely on the CPU: learner + sampler + agent processes (plumbing, no GPU).
env: Pendulum-v0
model: d4pg
num_agents: 4
action_low: -2
action_high: 2
v_min: -1000.0
v_max: 0.0
device: cpu
agent_device: cpu
replay_device: host
""",
}


def main():
    os.makedirs(OUT, exist_ok=True)
    for stem, env, S, A, vr in ENVS:
        for model in ("d4pg", "d3pg", "ddpg"):
            with open(os.path.join(OUT, f"{stem}_{model}.yml"), "w") as fh:
                fh.write(render(stem, env, S, A, vr, model))
    os.makedirs(os.path.join(OUT, "mi355x"), exist_ok=True)
    for name, text in MI355X.items():
        with open(os.path.join(OUT, "mi355x", name), "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main()
