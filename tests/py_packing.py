"""Weight packing for the MI355X kernels: PyTorch [out, in] matrices -> MFMA A-operand streams.

Every dense projection on the hot path is executed on `v_mfma_f32_32x32x2_f32` in the *transposed*
orientation  D[out_feature, item] = sum_k W[out_feature, k] * X[k, item]  with

  * the 32 "items" (edges or nodes) of a wave in lanes (item j = lane & 31),
  * the weight matrix as the A operand (lane l supplies W[row l & 31][k-slot l >> 5]),
  * the activations as the B operand, ONE register per k-step: at step R lanes 0-31 supply
    input slot (R, half 0) and lanes 32-63 slot (R, half 1) of their item.

The accumulator of an output block of 32 features then holds, in lane (j, half h), register s, the
output row  i = (s & 3) + 8 * (s >> 2) + 4 * h  of that block — and can be fed straight back as the
B operand of the next projection (register s of half h = k-slot (s, h)), so chains of projections
never leave the register file.  All that is needed is that the weights are laid out for it; that
is what this file does, once, at model-load time.

Slot conventions ("natural-half" layout): a feature vector of F = 32*nb features lives in
registers R = 0 .. F/2-1 of both halves with
        feature(R, h) = (R // 16) * 32 + h * 16 + (R % 16)
so each half-lane loads/stores runs of 16 consecutive floats and memory stays in natural order.
Other maps (concatenated inputs, the 14x18 attention-head arrangement) are expressed as explicit
`in_map` / `out_map` index arrays.

Packed layout of one projection: float32 [n_out_blocks][ksteps/4][64 lanes][4]  (one 16-byte load
per lane feeds four consecutive MFMAs).
"""
import numpy as np

LANES = 64


def natural_in_map(n_features):
    """[ksteps, 2] -> input column; n_features must be a multiple of 32."""
    assert n_features % 32 == 0
    R = np.arange(n_features // 2)
    m = np.empty((n_features // 2, 2), dtype=np.int64)
    for h in (0, 1):
        m[:, h] = (R // 16) * 32 + h * 16 + (R % 16)
    return m


def natural_out_map(n_features, n_valid=None):
    """[n_blocks, 2, 16] -> output row (or -1 = zero padding)."""
    assert n_features % 32 == 0
    nb = n_features // 32
    m = np.empty((nb, 2, 16), dtype=np.int64)
    for b in range(nb):
        for h in (0, 1):
            m[b, h] = b * 32 + h * 16 + np.arange(16)
    if n_valid is not None:
        m[m >= n_valid] = -1
    return m


def small_in_map(n_features, base=0):
    """Map for a short feature group (n <= 32, padded to a multiple of 8): half h, register s
    holds feature h * (npad/2) + s.  Returns [npad/2, 2] with -1 for padding."""
    npad = ((n_features + 7) // 8) * 8
    half = npad // 2
    m = np.full((half, 2), -1, dtype=np.int64)
    for h in (0, 1):
        for s in range(half):
            f = h * half + s
            if f < n_features:
                m[s, h] = base + f
    return m


def concat_in_maps(*maps):
    out = np.concatenate(maps, axis=0)
    assert out.shape[0] % 4 == 0, "k-steps must be a multiple of 4 (pad the input groups)"
    return out


def qk_out_map(sub_heads, sub_ch):
    """Arrangement of the SH x SC (e.g. 14 x 18) query/key/lin_edge0 features so that the per-head
    reduction is register-local:
       blocks 0 .. SH/2-1 : half h of block b holds channels 0..15 of head 2b+h
       remaining blocks    : the SC-16 tail channels; tail channel c of head g sits in
                             block SH/2 + c//2, half c%2, register g
    Needs SH even, SH <= 16 and 16 <= SC."""
    assert sub_heads % 2 == 0 and sub_heads <= 16 and sub_ch >= 16
    tail = sub_ch - 16
    nb = sub_heads // 2 + (tail + 1) // 2
    m = np.full((nb, 2, 16), -1, dtype=np.int64)
    for b in range(sub_heads // 2):
        for h in (0, 1):
            g = 2 * b + h
            m[b, h] = g * sub_ch + np.arange(16)
    for c in range(tail):
        b = sub_heads // 2 + c // 2
        h = c % 2
        for g in range(sub_heads):
            m[b, h, g] = g * sub_ch + 16 + c
    return m


def qk_out_map_wide(sub_heads, sub_ch):
    """Width-generic arrangement (csrc/dgt_kernels_wide.h): head g owns the 32-row block g in natural
    order, rows >= SC are zero padding — the per-head reduction is 16 in-lane terms + one exchange for
    any SC <= 32 (SC = 27 at nf = 384)."""
    assert sub_ch <= 32
    m = np.full((sub_heads, 2, 16), -1, dtype=np.int64)
    for g in range(sub_heads):
        for h in (0, 1):
            for s in range(16):
                c = h * 16 + s
                if c < sub_ch:
                    m[g, h, s] = g * sub_ch + c
    return m


def out_row_of_lane(i):
    """MFMA output row i (0..31) of a block -> (half, register)."""
    return (i >> 2) & 1, (i & 3) + 4 * (i >> 3)


def pack_projection(weight, in_map, out_map):
    """weight [n_out, n_in] (PyTorch layout) -> float32 [nb, ksteps/4, 64, 4]."""
    W = np.asarray(weight, dtype=np.float32)
    ksteps = in_map.shape[0]
    assert ksteps % 4 == 0
    nb = out_map.shape[0]
    Wp = np.concatenate([W, np.zeros((1, W.shape[1]), np.float32)], axis=0)      # row -1 -> zeros
    Wp = np.concatenate([Wp, np.zeros((Wp.shape[0], 1), np.float32)], axis=1)    # col -1 -> zeros
    lane = np.arange(LANES)
    i = lane & 31
    kh = lane >> 5
    oh, os_ = (i >> 2) & 1, (i & 3) + 4 * (i >> 3)
    rows = out_map[:, oh, os_]                       # [nb, 64]
    cols = in_map[:, kh]                             # [ksteps, 64]
    packed = Wp[rows[:, None, :], cols[None, :, :]]  # [nb, ksteps, 64]
    packed = packed.reshape(nb, ksteps // 4, 4, LANES).transpose(0, 1, 3, 2)
    return np.ascontiguousarray(packed, dtype=np.float32)


def pack_vector(vec, out_map):
    """bias / per-feature vector [n_out] -> float32 [nb*32] in slot order (-1 -> 0)."""
    v = np.concatenate([np.asarray(vec, np.float32).reshape(-1), np.zeros(1, np.float32)])
    return np.ascontiguousarray(v[out_map.reshape(-1)], dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# numpy emulation of the device-side chain (used by CPU tests to validate the maps)
# ---------------------------------------------------------------------------------------------
def emulate_projection(packed, act):
    """packed [nb, ksteps/4, 64, 4]; act [ksteps, 2, 32] (register R, half h, item j).
    Returns acc [nb, 16, 2, 32] (register s, half h, item j), following the MFMA semantics
    D[i][j] = sum_{k in {0,1}} A[i][k] B[k][j] per step."""
    nb, kq = packed.shape[0], packed.shape[1]
    A = packed.transpose(0, 1, 3, 2).reshape(nb, kq * 4, LANES)      # [nb, ksteps, lane]
    A = A.reshape(nb, kq * 4, 2, 32)                                 # [nb, R, k, i]
    D = np.einsum('brki,rkj->bij', A.astype(np.float64), act.astype(np.float64))  # [nb, 32 rows, 32 items]
    acc = np.empty((nb, 16, 2, 32))
    for i in range(32):
        h, s = out_row_of_lane(i)
        acc[:, s, h, :] = D[:, i, :]
    return acc


def to_slots(x, in_map):
    """x [items<=32, n_features] -> act [ksteps, 2, 32] according to in_map (-1 -> 0)."""
    xs = np.zeros((32, x.shape[1] + 1), dtype=np.float64)
    xs[:x.shape[0], :-1] = x
    act = xs[:, in_map]                       # [32, ksteps, 2]
    return act.transpose(1, 2, 0)


def from_slots(acc, out_map, n_out):
    """acc [nb, 16, 2, 32] -> y [32, n_out] according to out_map."""
    y = np.zeros((32, n_out))
    nb = out_map.shape[0]
    for b in range(nb):
        for h in (0, 1):
            for s in range(16):
                f = out_map[b, h, s]
                if f >= 0:
                    y[:, f] = acc[b, s, h, :]
    return y


def acc_as_act(acc):
    """Accumulators of nb blocks -> activation registers of the next projection:
    register R = b*16 + s of half h (the zero-copy chaining the kernels rely on)."""
    nb = acc.shape[0]
    return acc.reshape(nb * 16, 2, 32)
