"""CPU: the order-free form of the reference's nearest-hit update against the sequential loop it replaces.

Triangle::Intersect (Serial/geometry.h:162-175) does, for every passing triangle of a voxel in ascending index,
    if (t_ < global_t) { global_t = t_; global_triangle_near = this; }          (t_ double, global_t float)
so the outcome depends on the test order through the rounding of global_t. The HIP path collects a voxel's passing
triangles as (f, j0, L) in any order (VoxelBest in rt_kernels.hip) and applies that summary to the ray's state. This
file restates both in numpy scalars and compares them on candidate lists built to hit every tie: distances that round
to the same float from below and from above, exact floats, duplicates, +-inf, NaN, and incoming states equal to f."""
import numpy as np

F = np.float32
INF = F(np.inf)


def sequential(t_in, id_in, cands):
    """The reference's loop: cands = [(index, t double)] in ascending index."""
    t, i = F(t_in), id_in
    for idx, tc in cands:
        if np.float64(tc) < np.float64(t):
            t, i = F(tc), idx
    return t, i


def summary(cands):
    """VoxelBest accumulated in the given (arbitrary) order."""
    f, j0, L = INF, 0xffffffff, -1
    for idx, tc in cands:
        with np.errstate(over="ignore"):
            fj = F(tc)
        below = idx if np.float64(tc) < np.float64(fj) else -1
        if fj < f:
            f, j0, L = fj, idx, below
        elif fj == f:
            j0, L = min(j0, idx), max(L, below)
    return f, j0, L


def apply(t_in, id_in, s):
    f, j0, L = s
    t, i = F(t_in), id_in
    if f < t:
        return f, (L if L >= 0 else j0)
    if f == t and L >= 0:
        return t, L
    return t, i


def _neighbours(x, rng):
    """Doubles that round to the float x from below / above / exactly, and ones that just miss it."""
    x = F(x)
    up, dn = np.nextafter(x, INF), np.nextafter(x, -INF)
    hi_half = (np.float64(up) - np.float64(x)) / 2
    lo_half = (np.float64(x) - np.float64(dn)) / 2
    k = rng.integers(0, 7)
    if k == 0:
        return np.float64(x)
    if k == 1:
        return np.float64(x) - lo_half * rng.uniform(0.01, 0.99)      # rounds up to x, below it as a double
    if k == 2:
        return np.float64(x) + hi_half * rng.uniform(0.01, 0.99)      # rounds down to x, above it as a double
    if k == 3:
        return np.float64(x) - lo_half * rng.uniform(1.01, 1.9)       # rounds to the float below
    if k == 4:
        return np.float64(x) + hi_half * rng.uniform(1.01, 1.9)       # rounds to the float above
    if k == 5:
        return np.nextafter(np.float64(x), -np.inf)
    return np.nextafter(np.float64(x), np.inf)


def test_order_free_rule_equals_the_sequential_update():
    rng = np.random.default_rng(5)
    n_cases = 0
    for trial in range(30000):
        base = F(rng.choice([1.0, 0.1, 3.0000002, 1e-3, 117.25, -2.5, 16777216.0]) * rng.choice([1.0, 1.0, 1.0, 1.0000001]))
        n = int(rng.integers(0, 9))
        ids = sorted(rng.choice(1000, size=n, replace=False).tolist())
        cands = []
        for idx in ids:
            r = rng.random()
            if r < 0.80:
                tc = _neighbours(base, rng)
            elif r < 0.90:
                tc = np.float64(base) * rng.uniform(0.5, 2.0)
            elif r < 0.94:
                tc = np.float64(np.inf) * rng.choice([-1.0, 1.0])
            elif r < 0.97:
                tc = np.float64(np.nan)
            else:
                tc = np.float64(rng.choice([1e300, -1e300, 3.5e38, 0.0, -0.0]))      # overflow to +-inf as floats, zeros
            cands.append((idx, tc))
        r = rng.random()
        if r < 0.3:
            t_in, id_in = INF, -1
        elif r < 0.8:
            t_in, id_in = F(_neighbours(base, rng)), 5000
        else:
            t_in, id_in = F(np.float64(base) * rng.uniform(0.5, 2.0)), 5000
        with np.errstate(over="ignore"):
            want = sequential(t_in, id_in, cands)
            for _ in range(3):
                perm = [cands[k] for k in rng.permutation(len(cands))]
                got = apply(t_in, id_in, summary(perm))
                assert (got[0] == want[0] or (np.isnan(got[0]) and np.isnan(want[0]))) and got[1] == want[1], (t_in, id_in, cands, want, got)
        n_cases += 1
    assert n_cases == 30000


def test_summaries_of_parts_combine():
    """Several lanes may each hold part of a voxel's records: merging their summaries (smaller f wins; on equal f the
    smaller j0 and the larger L) is the summary of the whole."""
    rng = np.random.default_rng(6)

    def merge(a, b):
        if a[0] < b[0]:
            return a
        if b[0] < a[0]:
            return b
        return a[0], min(a[1], b[1]), max(a[2], b[2])

    for trial in range(5000):
        base = F(rng.choice([1.0, 0.3, 9.75]))
        ids = sorted(rng.choice(200, size=int(rng.integers(0, 10)), replace=False).tolist())
        cands = [(i, _neighbours(base, rng)) for i in ids]
        cut = int(rng.integers(0, len(cands) + 1))
        whole = summary(cands)
        parts = merge(summary(cands[:cut]), summary(cands[cut:]))
        assert whole == parts or (whole[0] == parts[0] == INF)
