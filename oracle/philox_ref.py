"""TEST INFRASTRUCTURE (checker only; never imported by the product path).

numpy restatement of the counter-based normal draws of csrc/sampler_kernels.hip (`philox_normal4` and the element
numbering of jodo_sampler_step_rng / jodo_dpm_update_rng): Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw:
"Parallel Random Numbers: As Easy as 1, 2, 3", SC'11; Random123 `philox4x32_R(10, ...)`) followed by Box-Muller on
24-bit uniforms.  The reference draws its noise with torch.randn (models/utils.py:67-99); there is no reference
counterpart of the generator itself, so it is pinned by the Random123 known-answer vectors
(tests/test_host_logic.py::test_philox_known_answers) and the kernel is pinned against this file.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
RNG_POS, RNG_FEAT, RNG_EDGE = 0, 1, 2


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def normal4(seed, draw, stream, elem):
    """Four N(0,1) float32 draws per element: elem int array [...] -> float32 [..., 4] (Box-Muller pairs (0,1), (2,3))."""
    elem = np.asarray(elem, dtype=np.uint64)
    ctr = np.stack([elem & MASK, elem >> np.uint64(32), np.full(elem.shape, draw, np.uint64), np.full(elem.shape, stream, np.uint64)],
                   axis=-1).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    u = philox4x32_10(ctr, key)
    out = np.empty(elem.shape + (4,), np.float32)
    for p in range(2):
        u1 = ((u[..., 2 * p] >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
        u2 = (u[..., 2 * p + 1] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        r = np.sqrt(np.float32(-2.0) * np.log(u1.astype(np.float64))).astype(np.float32)
        ang = np.float64(2.0) * u2.astype(np.float64) * np.pi
        out[..., 2 * p] = r * np.cos(ang).astype(np.float32)
        out[..., 2 * p + 1] = r * np.sin(ang).astype(np.float32)
    return out


def node_noise(seed, draw, n_nodes, N, nd):
    """The node noise jodo_sampler_step_rng applies: [B, N, 3 + nd], positions masked and centre-of-mass free
    (models/utils.py:38-45, 67-90 applied to the in-kernel draws), features masked."""
    B = len(n_nodes)
    atom = np.arange(B * N, dtype=np.uint64).reshape(B, N)
    pos = normal4(seed, draw, RNG_POS, atom)[..., :3]
    k = np.arange(nd)
    feat4 = normal4(seed, draw, RNG_FEAT, atom[..., None] * np.uint64(64) + (k // 4).astype(np.uint64))      # [B,N,nd,4]
    feat = np.take_along_axis(feat4, (k % 4)[None, None, :, None], axis=-1)[..., 0]
    mask = (np.arange(N)[None, :] < np.asarray(n_nodes)[:, None]).astype(np.float32)[..., None]
    pos = pos * mask
    pos = pos - pos.sum(1, keepdims=True) / np.asarray(n_nodes, np.float32)[:, None, None] * mask
    return np.concatenate([pos, feat * mask], axis=-1).astype(np.float32)


def edge_noise(seed, draw, n_nodes, N, ch):
    """The edge noise of jodo_sampler_step_rng: [B, N, N, ch], symmetric, zero diagonal, masked (models/utils.py:93-99):
    entry (a, c) takes channel f of the draw of cell (b, max(a,c), min(a,c))."""
    B = len(n_nodes)
    a = np.arange(N)
    lo, hi = np.maximum(a[:, None], a[None, :]), np.minimum(a[:, None], a[None, :])
    cell = (np.arange(B, dtype=np.uint64)[:, None, None] * np.uint64(N) + lo.astype(np.uint64)[None]) * np.uint64(N) + hi.astype(np.uint64)[None]
    z = normal4(seed, draw, RNG_EDGE, cell)[..., :ch]
    m = (a[None, :] < np.asarray(n_nodes)[:, None])
    em = (m[:, :, None] & m[:, None, :] & (a[:, None] != a[None, :])[None]).astype(np.float32)
    return (z * em[..., None]).astype(np.float32)
