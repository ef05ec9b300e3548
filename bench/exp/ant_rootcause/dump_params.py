"""Writes AntParams (csrc/ant_sim.h) of the Python ant model and a batch of start states for the stand-alone drivers
of this directory: params.bin (the struct's bytes), states.bin ([E][32] fp32), actions.bin ([T][E][8] fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from d4pg_pytorch_amd.envs.native_planar import NativeAntVec, ant_params

out = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__))
E, T = 256, 60
env = NativeAntVec(E, seed=3, time_limit=1000)
env.reset()
open(os.path.join(out, "params.bin"), "wb").write(bytes(ant_params(env)))
s = env.sim
st = np.zeros((E, 32), np.float32)
st[:, 0:3], st[:, 3:7], st[:, 7:15], st[:, 15:29] = s.pos, s.quat, s.qj, s.u
st.tofile(os.path.join(out, "states.bin"))
np.random.default_rng(0).uniform(-1, 1, (T, E, 8)).astype(np.float32).tofile(os.path.join(out, "actions.bin"))
# states with the legs on the ground (contacts active), 30 steps into an episode
rng = np.random.default_rng(1)
for _ in range(30):
    env.step(rng.uniform(-1, 1, (E, 8)).astype(np.float32))
s = env.sim
st[:, 0:3], st[:, 3:7], st[:, 7:15], st[:, 15:29] = s.pos, s.quat, s.qj, s.u
st.tofile(os.path.join(out, "states_contact.bin"))
print("wrote", out, E, T)
