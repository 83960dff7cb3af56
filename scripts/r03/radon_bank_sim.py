"""CPU model of the LDS bank conflicts of the tiled Radon forward kernel's tap reads (ds_read_b128: 16-lane groups, 16 bank
quads of 16 B): cycles per read instruction for a lane -> ray map and a window pitch, averaged over angles and steps.
Usage: python scripts/r03/radon_bank_sim.py"""
import numpy as np

G, A = 724, 720
th = np.deg2rad(np.arange(A) * 180.0 / A)
rng = np.random.default_rng(0)


def cycles(cols, rows, pitch):
    """cols, rows: [S, 16] integer cell of each lane of a group at S sample steps -> mean service cycles of the group"""
    addr = rows * pitch + cols
    bank = addr & 15
    out = np.zeros(cols.shape[0])
    for b in range(16):
        m = bank == b
        # distinct addresses in bank b per step
        a = np.where(m, addr, -10**9)
        a = np.sort(a, axis=1)
        distinct = (np.diff(a, axis=1) != 0).sum(1) + 1 - (~m).any(1)      # minus the sentinel bucket when present
        distinct = np.where(m.any(1), distinct, 0)
        out = np.maximum(out, distinct)
    return out.mean()


def angle_cost(c, s, perm, pitch, nsteps=64):
    """one wave (64 rays from a random block, `perm[lane]` = ray within the block), nsteps consecutive steps"""
    j0 = rng.integers(0, G - 64)
    i0 = rng.integers(0, G - nsteps)
    j = j0 + perm[None, :]                       # [1, 64]
    i = (i0 + np.arange(nsteps))[:, None]        # [S, 1]
    xj = -1 + 2 * j / (G - 1)
    xi = -1 + 2 * i / (G - 1)
    ix = ((c * xj + s * xi) + 1) * 0.5 * (G - 1)
    iy = ((-s * xj + c * xi) + 1) * 0.5 * (G - 1)
    if abs(s) > abs(c):
        ix, iy = iy, ix                          # SWAP family: the transposed image
    col, row = np.floor(ix).astype(np.int64) + 4096, np.floor(iy).astype(np.int64) + 4096
    tot = 0.0
    for g in range(4):
        tot += cycles(col[:, 16 * g:16 * g + 16], row[:, 16 * g:16 * g + 16], pitch)
    return tot


def evaluate(name, perm_of_angle, pitch):
    cs = [angle_cost(np.cos(t), np.sin(t), perm_of_angle(k), pitch) for k, t in enumerate(th[::6])]
    print(f"{name:40s} pitch {pitch:4d}: {np.mean(cs):.2f} cycles / read (4 = conflict-free), worst angle {np.max(cs):.2f}")
    return np.array(cs)


ident = np.arange(64)
base = evaluate("adjacent rays (current)", lambda k: ident, 128)
for p in (129, 132, 136, 120, 127):
    evaluate("adjacent rays", lambda k: ident, p)
best = np.full(len(th[::6]), 1e9)
pick = np.zeros(len(th[::6]), dtype=int)
for m in range(1, 64, 2):
    perm = (m * ident) % 64
    cs = evaluate(f"ray = {m} * lane mod 64", lambda k: perm, 128) if m in (1, 3, 5, 7, 9, 11, 13, 15, 17, 21, 23, 27, 29, 31) else None
    if cs is not None:
        pick = np.where(cs < best, m, pick)
        best = np.minimum(best, cs)
print("best multiplier per angle:", best.mean(), "histogram of picks:", np.unique(pick, return_counts=True))


def angle_cost_skew(c, s, beta, nsteps=64, trials=3):
    """window rows shifted by round(beta * row) columns (per-chunk skew), pitch = 0 mod 16"""
    tot = 0.0
    for _ in range(trials):
        j0 = rng.integers(0, G - 64); i0 = rng.integers(0, G - nsteps)
        j = j0 + ident[None, :]; i = (i0 + np.arange(nsteps))[:, None]
        xj = -1 + 2 * j / (G - 1); xi = -1 + 2 * i / (G - 1)
        ix = ((c * xj + s * xi) + 1) * 0.5 * (G - 1); iy = ((-s * xj + c * xi) + 1) * 0.5 * (G - 1)
        if abs(s) > abs(c):
            ix, iy = iy, ix
        col, row = np.floor(ix).astype(np.int64) + 4096, np.floor(iy).astype(np.int64) + 4096
        col = col + np.round(beta * (row & 15)).astype(np.int64)     # rows of a 16-row band
        for g in range(4):
            tot += cycles(col[:, 16 * g:16 * g + 16], row[:, 16 * g:16 * g + 16], 128)
    return tot / trials


print("\nper-angle best row skew (shift = round(beta * row)):")
betas = np.linspace(-1.5, 1.5, 25)
res = []
for k, t in enumerate(th[::12]):
    c, s = np.cos(t), np.sin(t)
    costs = [angle_cost_skew(c, s, b) for b in betas]
    kb = int(np.argmin(costs))
    res.append((np.rad2deg(t), betas[kb], costs[kb], costs[12]))
res = np.array(res)
print("mean best", res[:, 2].mean(), "mean beta=0", res[:, 3].mean())
for r in res[::5]:
    print(f"theta {r[0]:6.1f}  best beta {r[1]:+.3f}  cycles {r[2]:.2f}  (beta 0: {r[3]:.2f})")
