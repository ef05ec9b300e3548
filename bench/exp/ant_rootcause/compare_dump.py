"""First sub-step's system of the looped -O3 build against the looped -O1 build (ant_step_bisect.hip -DDUMP_SYSTEM):
which entries of M / Q / u' differ, and in how many environments."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], np.float32).reshape(-1, 163)[:, 30:]
b = np.fromfile(sys.argv[2], np.float32).reshape(-1, 163)[:, 30:]
names = [f"M[{r}][{c}]" for r in range(14) for c in range(r + 1)] + [f"Q[{r}]" for r in range(14)] + [f"ud[{r}]" for r in range(14)]
err = np.abs(a - b) / (1e-3 + np.abs(a))
bad = (err > 1e-2) | ~np.isfinite(b)
print("environments whose first sub-step differs:", bad.any(1).sum(), "of", len(a), "; non-finite entries in -O3:", (~np.isfinite(b)).sum())
cnt = bad.sum(0)
order = np.argsort(-cnt)
print("entries wrong most often:", [(names[i], int(cnt[i])) for i in order[:30] if cnt[i]])
e = int(np.argmax(bad.any(1)))
print("environment", e, [(names[i], float(a[e, i]), float(b[e, i])) for i in np.nonzero(bad[e])[0][:20]])
