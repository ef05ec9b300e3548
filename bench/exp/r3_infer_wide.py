"""Device rollout steps/s with the policy on the 4x4x1 kernel (16-row tiles from 2048 states) against the 16x16x4 kernel."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.models.core import build_core
from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
from d4pg_pytorch_amd.actors.gpu_rollout import DeviceRollout
from d4pg_pytorch_amd.envs.device_envs import make_device_env

for env in ("LunarLanderContinuous-v2", "Walker2d-v2", "Pendulum-v0"):
    for E in (2048, 4096, 8192):
        for wide_from in (1 << 30, 1):
            os.environ["D4PG_INFER_WIDE_FROM"] = str(wide_from)
            cfg = default_config(env=env, model="d4pg", device="cuda", env_backend="native", gpu_rollout_envs=E,
                                 replay_mem_size=1_000_000, v_min=-100.0, v_max=100.0)
            core = build_core(cfg, "cuda")
            rp = HbmReplay(cfg, "cuda", core.dev_state)
            ro = DeviceRollout(cfg, make_device_env(cfg, E, "cuda", 11), rp, "cuda", segment=32, seed=12)
            ro.refresh_weights(core.online.flat.data_ptr(), core.layout.actor.base)
            for _ in range(3):
                ro.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = sum(ro.collect() for _ in range(20))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"env": env, "envs": E, "policy_kernel": "16x16x4" if ro.actor.wide else "4x4x1",
                              "env_steps_per_sec": round(n / dt), "us_per_step_of_all_envs": round(dt / (20 * 32) * 1e6, 2)}), flush=True)
            del ro, rp, core
            torch.cuda.empty_cache()
