"""One step of the device ant (the library named by D4PG_HIP_LIB) against the C++ host build, and the time of a step
of 4096 environments: the check used on builds of csrc/ant_env.hip with another loop order / optimisation level."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch
from d4pg_pytorch_amd.envs.device_envs import DeviceAnt
from d4pg_pytorch_amd.envs.native_planar import NativeAntVec
E = 4096
dev = DeviceAnt(E, "cuda", seed=5, time_limit=1000)
obs = torch.zeros(E, 111, device="cuda"); dev.reset(obs); torch.cuda.synchronize()
host = NativeAntVec(E, seed=0, time_limit=1000); host.reset()
nobs, rew, done, trunc = torch.zeros(E, 111, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
g = torch.Generator().manual_seed(0)
worst = 0.0
for t in range(30):
    st = dev.state.cpu().double().numpy()
    s = host.sim; s.pos[:], s.quat[:], s.qj[:], s.u[:] = st[:, 0:3], st[:, 3:7], st[:, 7:15], st[:, 15:29]
    act = (torch.rand(E, 8, generator=g) * 2 - 1).cuda()
    dev.step(act, obs, nobs, rew, done, trunc); torch.cuda.synchronize()
    o_h, r_h, d_h = host._step(act.cpu().numpy())
    d = np.abs(nobs.cpu().numpy()[:, :27] - o_h[:, :27])
    close = (d <= 5e-3 + 5e-3 * np.abs(o_h[:, :27])).all(1)
    worst = max(worst, 1.0 - close.mean())
print("worst fraction of environments off by more than rounding in one step:", round(worst, 4))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    dev.step(act, obs, nobs, rew, done, trunc)
torch.cuda.synchronize()
print("us per step of 4096 ants (kernel + counter launch, no graph):", round((time.perf_counter() - t0) / 200 * 1e6, 1))
