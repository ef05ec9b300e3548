"""Time of one device step of 4096 environments of each planar task for the library named by D4PG_HIP_LIB."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from d4pg_pytorch_amd.envs import planar
from d4pg_pytorch_amd.envs.device_envs import DevicePlanar
E = 4096
for cls in (planar.HopperVec, planar.Walker2dVec, planar.HalfCheetahVec, planar.BipedalWalkerVec):
    dev = DevicePlanar(cls, E, "cuda", seed=1)
    S, A = dev.state_dim, dev.action_dim
    obs, nobs = torch.zeros(E, S, device="cuda"), torch.zeros(E, S, device="cuda")
    rew, done, trunc = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    act = (torch.rand(E, A, device="cuda") * 2 - 1)
    dev.reset(obs)
    for _ in range(20):
        dev.step(act, obs, nobs, rew, done, trunc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        dev.step(act, obs, nobs, rew, done, trunc)
    torch.cuda.synchronize()
    print(cls.__name__, round((time.perf_counter() - t0) / 300 * 1e6, 1), "us per step (kernel + counter launch, no graph); finite:", bool(torch.isfinite(dev.state).all()))
