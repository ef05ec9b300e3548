"""Holds every device build's (M, Q, u') of bench/exp/ant_rootcause/ant_bisect.hip against the host build's."""
import sys, numpy as np
ref = np.fromfile(sys.argv[1], np.float32).reshape(-1, 133)
names = [f"M[{a}][{b}]" for a in range(14) for b in range(a + 1)] + [f"Q[{a}]" for a in range(14)] + [f"ud[{a}]" for a in range(14)]
for path in sys.argv[2:]:
    x = np.fromfile(path, np.float32).reshape(-1, 133)
    err = np.abs(x - ref) / (1e-3 + np.abs(ref))
    bad_env = (err[:, :119] > 1e-3).any(1)
    bad_ud = (~np.isfinite(x[:, 119:])).any(1) | (err[:, 119:] > 5e-2).any(1)
    print(f"{path}: envs with a wrong M/Q entry {bad_env.sum()} / {len(x)}, with wrong or non-finite u' {bad_ud.sum()}; "
          f"max rel err M/Q {np.nanmax(err[:, :119]):.3g}, u' {np.nanmax(err[:, 119:]):.3g}")
    if bad_env.any():
        e = int(np.argmax(bad_env))
        cols = np.nonzero(err[e, :119] > 1e-3)[0]
        print("  first bad env", e, "entries:", [(names[c], float(x[e, c]), float(ref[e, c])) for c in cols[:12]])
        worst = np.argsort(-(err[:, :119] > 1e-3).sum(0))[:12]
        print("  entries most often wrong:", [(names[c], int((err[:, c] > 1e-3).sum())) for c in worst])
