"""TEST INFRASTRUCTURE ONLY: helpers shared by tools/make_golden.py and tests/ -- tiny synthetic
event streams and (un)flattening of the reference's nested history lists into npz-storable arrays.
"""
import numpy as np


def tiny_stream(seed, num_ent, num_rels, num_t, per_t, time_unit=1, t0=0):
    """A small synthetic quadruple stream [(s, r, o, t)], sorted by t, Zipf-ish entity popularity,
    with repeated (s, o) pairs and multi-edges so that induced subgraphs have non-trivial degree."""
    rng = np.random.RandomState(seed)
    pop = 1.0 / np.arange(1, num_ent + 1) ** 0.8
    pop /= pop.sum()
    quads = []
    for k in range(num_t):
        m = max(2, int(per_t + rng.randint(-per_t // 4, per_t // 4 + 1)))
        s = rng.choice(num_ent, size=m, p=pop)
        o = rng.choice(num_ent, size=m, p=pop)
        r = rng.randint(0, num_rels, size=m)
        t = np.full(m, t0 + k * time_unit)
        q = np.stack((s, r, o, t), axis=1)
        if k > 0 and len(quads[-1]) > 3:                     # some facts repeat from the previous step
            rep = quads[-1][rng.choice(len(quads[-1]), size=max(1, m // 8))].copy()
            rep[:, 3] = t0 + k * time_unit
            q = np.concatenate((q, rep))
        quads.append(q)
    return np.concatenate(quads).astype(np.int64)


def flatten_histories(hist, hist_t):
    """nested lists -> (seq_ptr[M+1], step_t[S], nbr_ptr[S+1], nbr[K,2])."""
    seq_ptr, step_t, nbr_ptr, nbr = [0], [], [0], []
    for h, ht in zip(hist, hist_t):
        for a, t in zip(h, ht):
            a = np.asarray(a, dtype=np.int64).reshape(-1, 2)
            nbr.append(a)
            nbr_ptr.append(nbr_ptr[-1] + len(a))
            step_t.append(int(t))
        seq_ptr.append(len(step_t))
    nbr = np.concatenate(nbr) if nbr else np.zeros((0, 2), np.int64)
    return (np.asarray(seq_ptr, np.int64), np.asarray(step_t, np.int64),
            np.asarray(nbr_ptr, np.int64), nbr.astype(np.int64))


def unflatten_histories(seq_ptr, step_t, nbr_ptr, nbr):
    hist, hist_t = [], []
    for i in range(len(seq_ptr) - 1):
        h, ht = [], []
        for k in range(seq_ptr[i], seq_ptr[i + 1]):
            h.append(np.asarray(nbr[nbr_ptr[k]:nbr_ptr[k + 1]], dtype=np.int64))
            ht.append(int(step_t[k]))
        hist.append(h)
        hist_t.append(ht)
    return hist, hist_t


def histories_equal(a, b):
    (ha, ta), (hb, tb) = a, b
    if len(ha) != len(hb):
        return False
    for x, y, u, v in zip(ha, hb, ta, tb):
        if [int(q) for q in u] != [int(q) for q in v] or len(x) != len(y):
            return False
        for p, q in zip(x, y):
            if not np.array_equal(np.asarray(p), np.asarray(q)):
                return False
    return True


# tiny datasets used by the golden fixtures (tools/make_golden.py) and the tests
DATASETS = {
    'tiny': dict(seed=11, num_ent=40, num_rels=5, num_t=20, per_t=14, time_unit=1),
    'small': dict(seed=23, num_ent=150, num_rels=12, num_t=30, per_t=50, time_unit=24),
}


def make_params(seed, shapes, scale=None):
    """Deterministic (numpy RandomState => stable across torch versions) parameter values.
    shapes: dict name -> shape.  scale=None: xavier-like bound per tensor, else uniform(-scale, scale)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(int(x) for x in shapes[name])
        if scale is not None:
            a = scale
        elif len(shp) >= 2:
            a = np.sqrt(2.0) * np.sqrt(6.0 / (shp[0] + shp[1]))
        else:
            a = 0.1
        out[name] = rng.uniform(-a, a, size=shp).astype(np.float32)
    return out


def split_dataset(name):
    """The train/valid/test split of a tiny dataset (70/15/15 % of the timestamps)."""
    cfg = DATASETS[name]
    q = tiny_stream(cfg['seed'], cfg['num_ent'], cfg['num_rels'], cfg['num_t'], cfg['per_t'],
                    time_unit=cfg['time_unit'])
    times = np.unique(q[:, 3])
    n_tr = int(len(times) * 0.7)
    n_va = int(len(times) * 0.85)
    tr = q[q[:, 3] < times[n_tr]]
    va = q[(q[:, 3] >= times[n_tr]) & (q[:, 3] < times[n_va])]
    te = q[q[:, 3] >= times[n_va]]
    return cfg, tr, va, te


BIG = 20000
NSAMP = 4096


def sample_idx(n):
    """Positions at which big gradient tensors are sampled in the fixtures."""
    return np.random.RandomState(n % 100003).randint(0, n, size=NSAMP)


def check_packed(npz, key, arr, rtol, atol):
    """Compare `arr` with a fixture entry written by tools/make_golden.py:pack_tensor.
    Returns (ok, max_abs_err, detail)."""
    arr = np.asarray(arr)
    if key in npz:
        ref = npz[key]
        err = float(np.max(np.abs(arr - ref))) if ref.size else 0.0
        return bool(np.allclose(arr, ref, rtol=rtol, atol=atol)), err, 'full'
    ref_s = npz[key + '__samp']
    got_s = arr.reshape(-1)[sample_idx(arr.size)]
    err = float(np.max(np.abs(got_s - ref_s)))
    nrm = float(np.linalg.norm(arr.astype(np.float64)))
    ok = np.allclose(got_s, ref_s, rtol=rtol, atol=atol) and \
        abs(nrm - float(npz[key + '__norm'])) <= rtol * 10 * float(npz[key + '__norm']) + atol
    return bool(ok), err, 'sampled'
