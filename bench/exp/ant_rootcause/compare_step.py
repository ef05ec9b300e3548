"""Holds each device build's state after one step (ant_step_bisect.hip) against the host build's."""
import sys, numpy as np
ref = np.fromfile(sys.argv[1], np.float32).reshape(-1, 30)
names = ["pos"] * 3 + ["quat"] * 4 + ["qj"] * 8 + ["v"] * 3 + ["w"] * 3 + ["qd"] * 8 + ["reward"]
for path in sys.argv[2:]:
    x = np.fromfile(path, np.float32).reshape(-1, 30)
    err = np.abs(x - ref) / (5e-3 + 5e-3 * np.abs(ref))
    bad = (err > 1).any(1) | (~np.isfinite(x)).any(1)
    worst = {}
    for i, n in enumerate(names):
        worst[n] = max(worst.get(n, 0.0), float(np.nanmax(np.abs(x[:, i] - ref[:, i]))))
    print(f"{path}: environments off by more than rounding {bad.sum()} / {len(x)}; non-finite {(~np.isfinite(x)).any(1).sum()}; "
          f"max abs difference by group {({k: round(v, 5) for k, v in worst.items()})}")
