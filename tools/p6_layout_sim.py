#!/usr/bin/env python
"""CPU model of the planes GEMM's data movement (csrc/gemm_p6.h): the T16 tiled plane format in HBM -> the 1 KB LDS-DMA
pieces -> the half-stage image in LDS -> the MFMA fragments read with ds_read_b128 (K-contiguous role) or
ds_read_b64_tr_b16 (K-strided role).  Checks (a) that every lane of every fragment receives the matrix element the
32x32x16 MFMA layout expects, for both roles, and (b) LDS bank conflicts of the fragment reads against the lane groups of
MI355X_MICROARCH.md's LDS table.  Pure numpy; also run by tests/test_planes_layout_cpu.py."""
import numpy as np


def t16_off(row, col, tc_count):
    """element offset of (row, col) inside one T16 plane with tc_count tiles per tile row"""
    i, j, tc = row & 15, col & 15, col >> 4
    ip = i ^ ((tc & 1) << 2)
    hh = (j >> 3) ^ ((i >> 3) & 1)
    return ((row >> 4) * tc_count + tc) * 256 + ip * 16 + hh * 8 + (j & 7)


def to_t16(mat):
    """[R, C] (multiples of 16) -> flat T16 plane"""
    r, c = mat.shape
    out = np.zeros(r * c, dtype=mat.dtype)
    rows, cols = np.meshgrid(np.arange(r), np.arange(c), indexing='ij')
    out[t16_off(rows, cols, c // 16)] = mat
    return out


def dma_image(plane, tc_count, tr, r0, kb, n_tiles):
    """The LDS image (in ELEMENTS: n_tiles * 256) of one operand's half-stage: r0 = first row (K-contiguous role) / first
    column (K-strided role) of the tile, kb = the 16-wide k block.  Piece p, lane l -> 16 bytes (8 elements) at image offset
    p * 512 + l * 8 from tile 2 p + (l >> 5), element offset (l & 31) * 8 inside it."""
    img = np.zeros(n_tiles * 256, dtype=plane.dtype)
    for p in range(n_tiles // 2):
        for l in range(64):
            ti = 2 * p + (l >> 5)
            if not tr:
                src = ((r0 // 16 + ti) * tc_count + kb) * 256 + (l & 31) * 8
            else:
                src = (kb * tc_count + r0 // 16 + ti) * 256 + (l & 31) * 8
            img[p * 512 + l * 8:p * 512 + l * 8 + 8] = plane[src:src + 8]
    return img


def frag_addr_contig(lane, w0, t, par):
    """BYTE address (inside the operand's image) of lane's ds_read_b128 for MFMA row block t of a wave whose rows start at
    w0; par = parity of the global k block."""
    r = w0 + 32 * t + (lane & 31)
    return (r >> 4) * 512 + (((r & 15) ^ (4 * par)) * 32) + ((((lane >> 5) ^ ((r >> 3) & 1))) * 16)


def frag_addr_tr(lane, w0, t, u):
    """BYTE address of lane's u-th ds_read_b64_tr_b16 (u = 0, 1) for MFMA column block t of a wave whose columns start at w0"""
    sl = lane & 15
    c = w0 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3)
    kk = 8 * (lane >> 5) + 4 * u + (sl >> 2)
    return (c >> 4) * 512 + ((kk ^ (4 * ((c >> 4) & 1))) * 32) + ((((c >> 3) & 1) ^ ((kk >> 3) & 1)) * 16) + ((c >> 2) & 1) * 8


def read_fragment_contig(img, w0, t, par):
    """-> [64 lanes, 8] elements: lane l must hold row w0 + 32 t + (l & 31), k = 8 (l >> 5) .. + 7"""
    out = np.zeros((64, 8), dtype=img.dtype)
    for l in range(64):
        a = frag_addr_contig(l, w0, t, par) // 2
        out[l] = img[a:a + 8]
    return out


def read_fragment_tr(img, w0, t):
    """ds_read_b64_tr_b16 x 2: in a 16-lane group, lane i receives element i & 3 of the 8-byte chunks addressed by lanes
    (i >> 2) + 4 j, j = 0..3 (measured in round 2, tools/probes/tr_probe.hip)."""
    out = np.zeros((64, 8), dtype=img.dtype)
    for u in range(2):
        chunks = np.zeros((64, 4), dtype=img.dtype)
        for l in range(64):
            a = frag_addr_tr(l, w0, t, u) // 2
            chunks[l] = img[a:a + 4]
        for l in range(64):
            g0, i = l & ~15, l & 15
            for j in range(4):
                out[l, 4 * u + j] = chunks[g0 + (i >> 2) + 4 * j][i & 3]
    return out


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in g] for g in B128_GROUPS]


def conflicts_b128(addrs):
    """extra LDS cycles of one ds_read_b128 wave-instruction: per lane group, (max lanes on one bank) - 1 summed"""
    extra = 0
    for g in B128_GROUPS:
        banks = {}
        for l in g:
            for d in range(4):
                banks.setdefault(((addrs[l] // 4) + d) % 64, set()).add(addrs[l] + 4 * d)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def conflicts_tr64(addrs):
    extra = 0
    for g in (range(0, 32), range(32, 64)):
        banks = {}
        for l in g:
            for d in range(2):
                banks.setdefault(((addrs[l] // 4) + d) % 64, set()).add(addrs[l] + 4 * d)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def check(verbose=False):
    rng = np.random.RandomState(0)
    R, C = 512, 96                    # stored matrix (multiples of 16)
    mat = rng.randint(1, 30000, (R, C)).astype(np.int32)
    plane = to_t16(mat)
    assert len(np.unique(plane)) == len(np.unique(mat))
    tcn = C // 16
    # K-contiguous role: rows = the tile's M index (256 rows from r0), k = columns
    for r0 in (0, 256):
        for kb in range(C // 16):
            img = dma_image(plane, tcn, False, r0, kb, 16)
            for wm in range(4):
                for t in range(2):
                    f = read_fragment_contig(img, wm * 64, t, kb & 1)
                    for l in range(64):
                        row = r0 + wm * 64 + 32 * t + (l & 31)
                        k = kb * 16 + 8 * (l >> 5)
                        assert np.array_equal(f[l], mat[row, k:k + 8]), ('contig', r0, kb, wm, t, l)
    # K-strided role: k = rows (16 per half-stage), the tile's M index = columns; use the transposed problem on `mat2`
    R2, C2 = 64, 512
    mat2 = rng.randint(1, 30000, (R2, C2)).astype(np.int32)
    plane2 = to_t16(mat2)
    tcn2 = C2 // 16
    for c0 in (0, 256):
        for kb in range(R2 // 16):
            img = dma_image(plane2, tcn2, True, c0, kb, 16)
            for wm in range(4):
                for t in range(2):
                    f = read_fragment_tr(img, wm * 64, t)
                    for l in range(64):
                        col = c0 + wm * 64 + 32 * t + (l & 31)
                        k = kb * 16 + 8 * (l >> 5)
                        assert np.array_equal(f[l], mat2[k:k + 8, col]), ('strided', c0, kb, wm, t, l)
    # bank conflicts of the fragment reads
    worst_c = max(conflicts_b128([frag_addr_contig(l, w0, t, par) for l in range(64)])
                  for w0 in (0, 64, 128, 192) for t in range(2) for par in range(2))
    worst_t = max(conflicts_tr64([frag_addr_tr(l, w0, t, u) for l in range(64)])
                  for w0 in (0, 64, 128, 192) for t in range(2) for u in range(2))
    if verbose:
        print('fragments correct in both roles; extra LDS cycles per read: b128 %d, tr64 %d' % (worst_c, worst_t))
    return worst_c, worst_t


if __name__ == '__main__':
    check(verbose=True)
