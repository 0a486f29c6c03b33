"""Seeded synthetic event streams shaped like the reference's datasets (no dataset blobs
are needed or read).  Statistics measured on ICEWS18 valid+test: 23 033 entities, 256 relations,
240 timestamps of unit 24, ~1 550 facts per timestamp (sigma ~300), Zipf-like entity popularity
(log-log slope ~ -0.9), relation slope ~ -2.2, ~11 % of a timestamp's facts repeated from the
previous one."""
import numpy as np

SHAPES = {
    # name: (num_ent, num_rels, num_t, per_t, sigma_t, time_unit, ent_slope, rel_slope, repeat)
    'ICEWS18': (23033, 256, 240, 1550, 300, 24, 0.9, 2.2, 0.11),
    'ICEWS14': (12498, 260, 365, 900, 200, 1, 0.9, 2.2, 0.11),
    'WIKI': (12554, 24, 211, 6200, 600, 1, 0.7, 1.5, 0.85),
    'YAGO': (10623, 10, 178, 908, 150, 1, 0.7, 1.2, 0.91),
    'GDELT': (7691, 240, 2750, 730, 120, 15, 0.9, 2.0, 0.20),
}


def zipf_probs(n, slope):
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** slope
    return p / p.sum()


def make_stream(shape='ICEWS18', seed=999, num_t=None):
    """Returns (quads[int64 n,4] sorted by time, num_ent, num_rels, time_unit)."""
    num_ent, num_rels, T, per_t, sigma, unit, es, rs, rep = SHAPES[shape]
    if num_t is not None:
        T = num_t
    rng = np.random.RandomState(seed)
    ent_perm = rng.permutation(num_ent)            # popularity rank -> entity id
    rel_perm = rng.permutation(num_rels)
    pe, pr = zipf_probs(num_ent, es), zipf_probs(num_rels, rs)
    out, prev = [], None
    for k in range(T):
        m = max(8, int(rng.normal(per_t, sigma)))
        n_rep = int(m * rep) if prev is not None else 0
        n_new = m - n_rep
        s = ent_perm[rng.choice(num_ent, n_new, p=pe)]
        o = ent_perm[rng.choice(num_ent, n_new, p=pe)]
        r = rel_perm[rng.choice(num_rels, n_new, p=pr)]
        q = np.stack((s, r, o), axis=1)
        if n_rep:
            q = np.concatenate((q, prev[rng.choice(len(prev), n_rep)]))
        q = q[rng.permutation(len(q))]
        prev = q
        out.append(np.concatenate((q, np.full((len(q), 1), k * unit)), axis=1))
    return np.concatenate(out).astype(np.int64), num_ent, num_rels, unit
