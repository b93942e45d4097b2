// Fused forward chains of the training step (see train_fused.h).  gfx950 only; strip model of dgt_device.h.
#include <hip/hip_runtime.h>
#include "dgt_kernels_attn.h"          // gbf_n (the Gaussian basis of the inference kernels), dgt_device.h
#include "train_fused.h"
#include "jodo_hip_internal.h"

namespace {
using namespace jd;

template <int D_>
struct FD {
    static constexpr int D = D_, De = D_ / 4, ND = D_ / 32, NE = D_ / 128, HD = D_ / 2, HE = D_ / 8;
    static constexpr int KQD = D_ / 8, KQE = D_ / 32;
    static constexpr int PG = (D_ % 256 == 0) ? 8 : 4;
};

// LayerNorm (no affine, eps 1e-6, biased variance) over 2 NR features of an item; x becomes the normalised row, returns rstd
template <int NR>
__device__ __forceinline__ float layer_norm_rs(float (&x)[NR]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += x[i];
    const float mean = pair_sum(s) * (1.f / (2 * NR));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) { x[i] -= mean; q = fmaf(x[i], x[i], q); }
    const float rstd = 1.f / sqrtf(pair_sum(q) * (1.f / (2 * NR)) + 1e-6f);
#pragma unroll
    for (int i = 0; i < NR; ++i) x[i] *= rstd;
    return rstd;
}

// dropout multipliers of 16 consecutive elements idx0 .. idx0 + 15 (idx0 a multiple of 4): the masks of train_common.h drop_mul,
// four elements per Philox call instead of one
__device__ __forceinline__ void drop16(const jt::Drop& d, unsigned long long idx0, float (&m)[16]) {
    if (d.p <= 0.f) {
#pragma unroll
        for (int s = 0; s < 16; ++s) m[s] = 1.f;
        return;
    }
    const float keep = 1.f / (1.f - d.p);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long g = (idx0 >> 2) + q;
        unsigned u[4];
        jt::philox4((unsigned)g, (unsigned)(g >> 32), d.site, 0x4a4f444fu, (unsigned)d.seed, (unsigned)(d.seed >> 32), u);
#pragma unroll
        for (int c = 0; c < 4; ++c) m[q * 4 + c] = ((float)(u[c] >> 8) * (1.0f / 16777216.0f)) < d.p ? 0.f : keep;
    }
}

struct Common {
    int R;
    const int *ea, *ec, *em;
    const float* packed;               // this block's packed operands
    unsigned oEE, oL0, oL1, oF3, oF4, oRO, oIN, oC0;      // byte offsets inside `packed`
    unsigned oC0T, oINT, oF4T, oF3T, oL1T, oL0T, oEET;    // transposed images (backward chains)
    const float* tab;                  // Gaussian table [3][De]
    int save;                          // 0: a forward no backward will follow (the no-grad self-conditioning call): activations only the
                                       // backward reads are not stored
};

struct ArgsA {
    Common c;
    const float *pos, *gm, *e_in, *emod, *ee_b;
    float *d2, *G, *xh, *rs, *et, *t0, *t1;
    int QK;
};

template <int D>
__global__ __launch_bounds__(64, 1) void k_chain_a(ArgsA A) {
    using X = FD<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const long r = (long)blockIdx.x * 32 + j;
    const bool valid = r < A.c.R;
    const long rc = valid ? r : (long)A.c.R - 1;
    const int a = A.c.ea[rc], c = A.c.ec[rc], mol = A.c.em[rc];
    const float dx = A.pos[a * 3] - A.pos[c * 3], dy = A.pos[a * 3 + 1] - A.pos[c * 3 + 1], dz = A.pos[a * 3 + 2] - A.pos[c * 3 + 2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    const bool keep = valid && A.c.save != 0;
    if (keep && half == 0) A.d2[r] = d2;
    const WSrc ws = make_wsrc(A.c.packed, lane);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, A.c.oEE);
    float G[X::HE], e[X::HE], x[X::HE];
    gbf_n<X::NE>(d2, A.gm[mol * 2], A.gm[mol * 2 + 1], A.c.tab, half, G);
    load_nat<X::NE>(A.e_in + rc * X::De, half, e);
    if (valid) store_nat<X::NE>(A.G + r * X::De, half, G);
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // edge_emb([G ; e])
        const unsigned cg = A.c.oEE + (unsigned)(b * 2 * X::KQE) * 1024, ce = cg + X::KQE * 1024;
        float bb[16];
        load16(A.ee_b + b * 32 + half * 16, bb);
        f32x16 acc = mfma_block_p<X::KQE>(wp, ws, cg, ce, G, zero16());
        acc = mfma_block_p<X::KQE>(wp, ws, ce, b + 1 < X::NE ? ce + X::KQE * 1024 : A.c.oL0, e, acc);
#pragma unroll
        for (int s = 0; s < 16; ++s) x[b * 16 + s] = acc[s] + bb[s];
    }
    const float rstd = layer_norm_rs<X::HE>(x);
    if (keep) {
        store_nat<X::NE>(A.xh + r * X::De, half, x);
        if (half == 0) A.rs[r] = rstd;
    }
    const float* mr = A.emod + (long)mol * 6 * X::De;
    modulate<X::NE>(x, mr, mr + X::De, half);
    if (keep) store_nat<X::NE>(A.et + r * X::De, half, x);
    const int nb0 = (A.QK + 31) / 32;
#pragma unroll 1
    for (int b = 0; b < nb0; ++b) {                                   // tanh(lin_edge0 et)
        const unsigned cur = A.c.oL0 + (unsigned)(b * X::KQE) * 1024;
        f32x16 acc = mfma_block_p<X::KQE>(wp, ws, cur, b + 1 < nb0 ? cur + X::KQE * 1024 : A.c.oL1, x, zero16());
        float T[16];
        tanh16(acc, T);
        if (valid) {
            float* o = A.t0 + r * A.QK + b * 32 + half * 16;          // rows of QK floats are 8-byte aligned (QK is even)
#pragma unroll
            for (int s = 0; s < 16; s += 2)
                if (b * 32 + half * 16 + s < A.QK) *reinterpret_cast<float2*>(o + s) = make_float2(T[s], T[s + 1]);
        }
    }
#pragma unroll 1
    for (int b = 0; b < X::ND; ++b) {                                 // tanh(lin_edge1 et)
        const unsigned cur = A.c.oL1 + (unsigned)(b * X::KQE) * 1024;
        f32x16 acc = mfma_block_p<X::KQE>(wp, ws, cur, b + 1 < X::ND ? cur + X::KQE * 1024 : A.c.oL1, x, zero16());
        float T[16];
        tanh16(acc, T);
        if (valid) store16(A.t1 + r * D + b * 32 + half * 16, T);
    }
}

struct ArgsB {
    Common c;
    const float *e_in, *n2e, *n2e_b, *emod, *b3, *b4, *bro;
    jt::Drop d3, d4;
    float *xh, *rs, *en, *f3, *a3, *f4, *e_out, *eh;
    int ld_eh, eh_col, ce;
};

template <int D, int RR>
__global__ __launch_bounds__(64, 1) void k_chain_b(ArgsB A) {
    using X = FD<D>;
    constexpr int HID = RR * X::De, NCH = HID / 64, KQ4 = HID / 8;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const long r = (long)blockIdx.x * 32 + j;
    const bool valid = r < A.c.R;
    const long rc = valid ? r : (long)A.c.R - 1;
    const int a = A.c.ea[rc], c = A.c.ec[rc], mol = A.c.em[rc];
    const float* mr = A.emod + (long)mol * 6 * X::De;
    const WSrc ws = make_wsrc(A.c.packed, lane);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, A.c.oF3);
    float x[X::HE];
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // x1 = e + g1 (n2e_a + n2e_c + bias)
        float e[16], ta[16], tc[16], g[16], bb[16];
        const int f0 = b * 32 + half * 16;
        load16(A.e_in + rc * X::De + f0, e);
        load16(A.n2e + (long)a * X::De + f0, ta);
        load16(A.n2e + (long)c * X::De + f0, tc);
        load16(mr + 2 * X::De + f0, g);
        load16(A.n2e_b + f0, bb);
#pragma unroll
        for (int s = 0; s < 16; ++s) x[b * 16 + s] = fmaf(g[s], ta[s] + tc[s] + bb[s], e[s]);
    }
    const bool keep = valid && A.c.save != 0;
    const float rstd = layer_norm_rs<X::HE>(x);
    if (keep) {
        store_nat<X::NE>(A.xh + r * X::De, half, x);
        if (half == 0) A.rs[r] = rstd;
    }
    modulate<X::NE>(x, mr + 3 * X::De, mr + 4 * X::De, half);
    if (keep) store_nat<X::NE>(A.en + r * X::De, half, x);
    f32x16 o[X::NE];
#pragma unroll
    for (int b = 0; b < X::NE; ++b) o[b] = zero16();
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        float hid[32];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int hb = ch * 2 + b2, f0 = hb * 32 + half * 16;
            const unsigned cur = A.c.oF3 + (unsigned)(hb * X::KQE) * 1024;
            const unsigned nxt = b2 == 0 ? cur + X::KQE * 1024 : A.c.oF4 + (unsigned)(ch * 8) * 1024;
            float bb[16], pre[16], act[16], m[16];
            load16(A.b3 + f0, bb);
            f32x16 acc = mfma_block_p<X::KQE>(wp, ws, cur, nxt, x, zero16());
            drop16(A.d3, (unsigned long long)rc * HID + f0, m);
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                const f32x2 p = pk2(acc[s], acc[s + 1]) + pk2(bb[s], bb[s + 1]);
                const f32x2 v = silu_f2(p) * pk2(m[s], m[s + 1]);
                pre[s] = p.x; pre[s + 1] = p.y; act[s] = v.x; act[s + 1] = v.y;
                hid[b2 * 16 + s] = v.x; hid[b2 * 16 + s + 1] = v.y;
            }
            if (keep) { store16(A.f3 + r * HID + f0, pre); store16(A.a3 + r * HID + f0, act); }
        }
#pragma unroll
        for (int ob = 0; ob < X::NE; ++ob) {
            const unsigned cur = A.c.oF4 + (unsigned)(ob * KQ4 + ch * 8) * 1024;
            const unsigned nxt = ob + 1 < X::NE ? A.c.oF4 + (unsigned)((ob + 1) * KQ4 + ch * 8) * 1024
                                                : (ch + 1 < NCH ? A.c.oF3 + (unsigned)((ch + 1) * 2 * X::KQE) * 1024 : A.c.oRO);
            o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, hid, o[ob]);
        }
    }
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // f4 (kept before its dropout), e' = en + g2 * dropout(f4)
        const int f0 = b * 32 + half * 16;
        float bb[16], g[16], m[16], f4[16];
        load16(A.b4 + f0, bb);
        load16(mr + 5 * X::De + f0, g);
        drop16(A.d4, (unsigned long long)rc * X::De + f0, m);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            f4[s] = o[b][s] + bb[s];
            x[b * 16 + s] = fmaf(g[s], f4[s] * m[s], x[b * 16 + s]);
        }
        if (keep) store16(A.f4 + r * X::De + f0, f4);
    }
    if (valid) store_nat<X::NE>(A.e_out + r * X::De, half, x);
    {   // readout edge_l(e') -> the head input, ce valid columns
        f32x16 acc = mfma_block_p<X::KQE>(wp, ws, A.c.oRO, A.c.oRO, x, zero16());
        if (valid) {
            float* o2 = A.eh + r * A.ld_eh + A.eh_col + half * 16;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (half * 16 + s < A.ce) o2[s] = acc[s] + A.bro[half * 16 + s];
        }
    }
}

struct ArgsC {
    Common c;
    const float *e_in, *G, *hr, *hc, *in_b, *qmod, *b0, *w2;
    float *xh, *rs, *u, *c0pre, *c0a, *inv;
};

#ifndef JODO_X_CHAINC_OCC                                              // experiment builds: waves per SIMD the compiler must leave room for
#define JODO_X_CHAINC_OCC 1
#endif
template <int D>
__global__ __launch_bounds__(64, JODO_X_CHAINC_OCC) void k_chain_c(ArgsC A) {
    using X = FD<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const long r = (long)blockIdx.x * 32 + j;
    const bool valid = r < A.c.R;
    const long rc = valid ? r : (long)A.c.R - 1;
    const int a = A.c.ea[rc], c = A.c.ec[rc], mol = A.c.em[rc];
    const WSrc ws = make_wsrc(A.c.packed, lane);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, A.c.oIN);
    float e[X::HE], G[X::HE], u[X::HD];
    load_nat<X::NE>(A.e_in + rc * X::De, half, e);
    load_nat<X::NE>(A.G + rc * X::De, half, G);
#pragma unroll
    for (int b = 0; b < X::ND; ++b) {                                 // pre = W_e e + W_g G + b + W_row h_a + W_col h_c
        const unsigned we = A.c.oIN + (unsigned)(b * 2 * X::KQE) * 1024, wg = we + X::KQE * 1024;
        const int f0 = b * 32 + half * 16;
        float bb[16], t1[16], t2[16];
        load16(A.in_b + f0, bb);
        load16(A.hr + (long)a * D + f0, t1);
        load16(A.hc + (long)c * D + f0, t2);
        f32x16 acc = mfma_block_p<X::KQE>(wp, ws, we, wg, e, zero16());
        acc = mfma_block_p<X::KQE>(wp, ws, wg, b + 1 < X::ND ? wg + X::KQE * 1024 : A.c.oC0, G, acc);
#pragma unroll
        for (int s = 0; s < 16; ++s) u[b * 16 + s] = (acc[s] + bb[s]) + (t1[s] + t2[s]);
    }
    const bool keep = valid && A.c.save != 0;
    const float rstd = layer_norm_rs<X::HD>(u);
    if (keep) {
        store_nat<X::ND>(A.xh + r * D, half, u);
        if (half == 0) A.rs[r] = rstd;
    }
    const float* mr = A.qmod + (long)mol * 2 * D;
    modulate<X::ND>(u, mr, mr + D, half);
    if (keep) store_nat<X::ND>(A.u + r * D, half, u);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 1
    for (int b = 0; b < X::ND; ++b) {                                 // coord_mlp.0 -> SiLU -> coord_mlp.2
        const unsigned cur = A.c.oC0 + (unsigned)(b * X::KQD) * 1024;
        const int f0 = b * 32 + half * 16;
        float bb[16], k0[16], k1[16], k2[16], pre[16], act[16];
        load16(A.b0 + f0, bb);
        load16(A.w2 + f0, k0);
        load16(A.w2 + D + f0, k1);
        load16(A.w2 + 2 * D + f0, k2);
        f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::ND ? cur + X::KQD * 1024 : A.c.oC0, u, zero16());
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            const f32x2 p = pk2(acc[s], acc[s + 1]) + pk2(bb[s], bb[s + 1]);
            const f32x2 v = silu_f2(p);
            pre[s] = p.x; pre[s + 1] = p.y; act[s] = v.x; act[s + 1] = v.y;
            c0 = fmaf(v.x, k0[s], c0); c0 = fmaf(v.y, k0[s + 1], c0);
            c1 = fmaf(v.x, k1[s], c1); c1 = fmaf(v.y, k1[s + 1], c1);
            c2 = fmaf(v.x, k2[s], c2); c2 = fmaf(v.y, k2[s + 1], c2);
        }
        if (keep) { store16(A.c0pre + r * D + f0, pre); store16(A.c0a + r * D + f0, act); }
    }
    c0 = tanh_f(pair_sum(c0)); c1 = tanh_f(pair_sum(c1)); c2 = tanh_f(pair_sum(c2));
    if (valid && half == 0) { A.inv[r * 3] = c0; A.inv[r * 3 + 1] = c1; A.inv[r * 3 + 2] = c2; }
}

// d/dx x sigmoid(x) = s (1 + x (1 - s)), element pairs on the packed pipe
__device__ __forceinline__ f32x2 dsilu_f2(f32x2 x) {
    const f32x2 t = x * -1.4426950408889634f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
    e = e + 1.f;
    f32x2 sg;
    sg.x = fast_rcp(e.x); sg.y = fast_rcp(e.y);
    return sg * __builtin_elementwise_fma(x, (f32x2)(1.f) - sg, (f32x2)(1.f));
}

// LayerNorm + modulate backward of one row held in registers: g = dy (1 + sc); dx = rstd (g - mean(g) - xhat mean(g xhat)).
// dy in / dx out in `v`; xh: the saved normalised row; sc: this lane's modulation scale row (natural order)
template <int NB, bool KEEP = true>       // KEEP: the saved row stays in registers between the two passes (short rows); otherwise it is read twice
__device__ __forceinline__ void ln_mod_bwd_regs(float (&v)[NB * 16], const float* __restrict__ xh_row, const float* __restrict__ sc, float rstd, int half) {
    float c1 = 0.f, c2 = 0.f;
    float xh[KEEP ? NB * 16 : 1];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float g[16], x[16];
        load16(sc + b * 32 + half * 16, g);
        load16(xh_row + b * 32 + half * 16, x);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float gv = v[b * 16 + s] * (1.f + g[s]);
            v[b * 16 + s] = gv;
            if constexpr (KEEP) xh[b * 16 + s] = x[s];
            c1 += gv; c2 = fmaf(gv, x[s], c2);
        }
    }
    c1 = pair_sum(c1) * (1.f / (NB * 32)); c2 = pair_sum(c2) * (1.f / (NB * 32));
    if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < NB * 16; ++i) v[i] = rstd * (v[i] - c1 - xh[i] * c2);
    } else {
        pipeline_fence();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float x[16];
            load16(xh_row + b * 32 + half * 16, x);
#pragma unroll
            for (int s = 0; s < 16; ++s) v[b * 16 + s] = rstd * (v[b * 16 + s] - c1 - x[s] * c2);
        }
    }
}

struct ArgsBC {
    Common c;
    const float *inv, *c0pre, *xh, *rs, *qmod, *w2;
    float *dinv, *dc0, *du, *dpre, *de, *dG;
};

template <int D>
__global__ __launch_bounds__(64, 1) void k_bwd_c(ArgsBC A) {
    using X = FD<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const long r = (long)blockIdx.x * 32 + j;
    const bool valid = r < A.c.R;
    const long rc = valid ? r : (long)A.c.R - 1;
    const int mol = A.c.em[rc];
    float d0, d1, d2;
    {
        const float i0 = A.inv[rc * 3], i1 = A.inv[rc * 3 + 1], i2 = A.inv[rc * 3 + 2];
        d0 = A.dinv[rc * 3] * (1.f - i0 * i0); d1 = A.dinv[rc * 3 + 1] * (1.f - i1 * i1); d2 = A.dinv[rc * 3 + 2] * (1.f - i2 * i2);
    }
    const WSrc ws = make_wsrc(A.c.packed, lane);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, A.c.oC0T);
    float v[X::HD];
#pragma unroll
    for (int b = 0; b < X::ND; ++b) {                                 // dc0 = (W2^T dinv') SiLU'(c0pre)
        const int f0 = b * 32 + half * 16;
        float k0[16], k1[16], k2[16], pre[16], o[16];
        load16(A.w2 + f0, k0); load16(A.w2 + D + f0, k1); load16(A.w2 + 2 * D + f0, k2);
        load16(A.c0pre + rc * D + f0, pre);
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            const f32x2 da = pk2(fmaf(d0, k0[s], fmaf(d1, k1[s], d2 * k2[s])), fmaf(d0, k0[s + 1], fmaf(d1, k1[s + 1], d2 * k2[s + 1])));
            const f32x2 g = da * dsilu_f2(pk2(pre[s], pre[s + 1]));
            o[s] = g.x; o[s + 1] = g.y; v[b * 16 + s] = g.x; v[b * 16 + s + 1] = g.y;
        }
        if (valid) store16(A.dc0 + r * D + f0, o);
    }
    __syncthreads();                                                  // (one wave: orders the in-place dinv update behind every lane's read)
    if (valid && half == 0) { A.dinv[r * 3] = d0; A.dinv[r * 3 + 1] = d1; A.dinv[r * 3 + 2] = d2; }
    float u[X::HD];
#pragma unroll
    for (int b = 0; b < X::ND; ++b) {                                 // du = W0^T dc0
        const unsigned cur = A.c.oC0T + (unsigned)(b * X::KQD) * 1024;
        f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::ND ? cur + X::KQD * 1024 : A.c.oINT, v, zero16());
        float o[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { o[s] = acc[s]; u[b * 16 + s] = acc[s]; }
        if (valid) store16(A.du + r * D + b * 32 + half * 16, o);
    }
    ln_mod_bwd_regs<X::ND, (D <= 256)>(u, A.xh + rc * D, A.qmod + (long)mol * 2 * D + D, A.rs[rc], half);
    if (valid) store_nat<X::ND>(A.dpre + r * D, half, u);
#pragma unroll
    for (int b = 0; b < 2 * X::NE; ++b) {                             // de += W_e^T dpre, dG = W_g^T dpre
        const unsigned cur = A.c.oINT + (unsigned)(b * X::KQD) * 1024;
        f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < 2 * X::NE ? cur + X::KQD * 1024 : A.c.oINT, u, zero16());
        float o[16];
        if (b < X::NE) {
            float* dst = A.de + rc * X::De + b * 32 + half * 16;
            load16(dst, o);
#pragma unroll
            for (int s = 0; s < 16; ++s) o[s] += acc[s];
            if (valid) store16(dst, o);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) o[s] = acc[s];
            if (valid) store16(A.dG + r * X::De + (b - X::NE) * 32 + half * 16, o);
        }
    }
}

struct ArgsBB {
    Common c;
    const float *de_out, *f4, *f3, *xh, *rs, *emod;
    jt::Drop d3, d4;
    float *f4d, *df4, *dhid, *den, *de_prev;
};

template <int D, int RR>
__global__ __launch_bounds__(64, 1) void k_bwd_b(ArgsBB A) {
    using X = FD<D>;
    constexpr int HID = RR * X::De, NHB = HID / 32, KQ4 = HID / 8;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const long r = (long)blockIdx.x * 32 + j;
    const bool valid = r < A.c.R;
    const long rc = valid ? r : (long)A.c.R - 1;
    const int mol = A.c.em[rc];
    const float* mr = A.emod + (long)mol * 6 * X::De;
    const WSrc ws = make_wsrc(A.c.packed, lane);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, A.c.oF4T);
    float dout[X::HE], df4[X::HE];
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // f4d = dropout(f4) (d g2 sums), df4 = g2 de_out mask
        const int f0 = b * 32 + half * 16;
        float de[16], f4[16], g[16], m[16], o1[16], o2[16];
        load16(A.de_out + rc * X::De + f0, de);
        load16(A.f4 + rc * X::De + f0, f4);
        load16(mr + 5 * X::De + f0, g);
        drop16(A.d4, (unsigned long long)rc * X::De + f0, m);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            o1[s] = f4[s] * m[s];
            o2[s] = g[s] * de[s] * m[s];
            dout[b * 16 + s] = de[s]; df4[b * 16 + s] = o2[s];
        }
        if (valid) { store16(A.f4d + r * X::De + f0, o1); store16(A.df4 + r * X::De + f0, o2); }
    }
    float dh[HID / 2];
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) {                                // dhid = (W4^T df4) mask3 SiLU'(f3)
        const int f0 = hb * 32 + half * 16;
        const unsigned cur = A.c.oF4T + (unsigned)(hb * X::KQE) * 1024;
        float pre[16], m[16], o[16];
        load16(A.f3 + rc * HID + f0, pre);
        f32x16 acc = mfma_block_p<X::KQE>(wp, ws, cur, hb + 1 < NHB ? cur + X::KQE * 1024 : A.c.oF3T, df4, zero16());
        drop16(A.d3, (unsigned long long)rc * HID + f0, m);
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            const f32x2 g = pk2(acc[s], acc[s + 1]) * pk2(m[s], m[s + 1]) * dsilu_f2(pk2(pre[s], pre[s + 1]));
            o[s] = g.x; o[s + 1] = g.y; dh[hb * 16 + s] = g.x; dh[hb * 16 + s + 1] = g.y;
        }
        if (valid) store16(A.dhid + r * HID + f0, o);
    }
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // den = de_out + W3^T dhid
        const unsigned cur = A.c.oF3T + (unsigned)(b * KQ4) * 1024;
        f32x16 acc = mfma_block_p<KQ4>(wp, ws, cur, b + 1 < X::NE ? cur + KQ4 * 1024 : A.c.oF3T, dh, zero16());
        float o[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { dout[b * 16 + s] += acc[s]; o[s] = dout[b * 16 + s]; }
        if (valid) store16(A.den + r * X::De + b * 32 + half * 16, o);
    }
    ln_mod_bwd_regs<X::NE>(dout, A.xh + rc * X::De, mr + 4 * X::De, A.rs[rc], half);
    if (valid) store_nat<X::NE>(A.de_prev + r * X::De, half, dout);
}

struct ArgsBA {
    Common c;
    const float *dt1, *dt0, *xh, *rs, *emod;
    float *det, *de1, *dG, *de_prev;
    int QK;
};

template <int D>
__global__ __launch_bounds__(64, 1) void k_bwd_a(ArgsBA A) {
    using X = FD<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const long r = (long)blockIdx.x * 32 + j;
    const bool valid = r < A.c.R;
    const long rc = valid ? r : (long)A.c.R - 1;
    const int mol = A.c.em[rc];
    const float* mr = A.emod + (long)mol * 6 * X::De;
    const WSrc ws = make_wsrc(A.c.packed, lane);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, A.c.oL1T);
    float x[X::HD];
    load_nat<X::ND>(A.dt1 + rc * D, half, x);
    f32x16 acc[X::NE];
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // det = lin_edge1^T dt1 ...
        const unsigned cur = A.c.oL1T + (unsigned)(b * X::KQD) * 1024;
        acc[b] = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::NE ? cur + X::KQD * 1024 : A.c.oL0T, x, zero16());
    }
    {
        const float* row = A.dt0 + rc * A.QK;                         // rows of QK floats: 8-byte aligned, the tail of the last block is zero
#pragma unroll
        for (int b = 0; b < X::ND; ++b)
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                const int f = b * 32 + half * 16 + s;
                float2 t = make_float2(0.f, 0.f);
                if (f < A.QK) t = *reinterpret_cast<const float2*>(row + f);
                x[b * 16 + s] = t.x; x[b * 16 + s + 1] = t.y;
            }
    }
    float v[X::HE];
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                 // ... + lin_edge0^T dt0
        const unsigned cur = A.c.oL0T + (unsigned)(b * X::KQD) * 1024;
        acc[b] = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::NE ? cur + X::KQD * 1024 : A.c.oEET, x, acc[b]);
        float o[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { o[s] = acc[b][s]; v[b * 16 + s] = acc[b][s]; }
        if (valid) store16(A.det + r * X::De + b * 32 + half * 16, o);
    }
    ln_mod_bwd_regs<X::NE>(v, A.xh + rc * X::De, mr + X::De, A.rs[rc], half);
    if (valid) store_nat<X::NE>(A.de1 + r * X::De, half, v);
#pragma unroll
    for (int b = 0; b < 2 * X::NE; ++b) {                             // dG += W_G^T de1, de_prev += W_e^T de1
        const unsigned cur = A.c.oEET + (unsigned)(b * X::KQE) * 1024;
        f32x16 a2 = mfma_block_p<X::KQE>(wp, ws, cur, b + 1 < 2 * X::NE ? cur + X::KQE * 1024 : A.c.oEET, v, zero16());
        float* dst = (b < X::NE ? A.dG + rc * X::De + b * 32 : A.de_prev + rc * X::De + (b - X::NE) * 32) + half * 16;
        float o[16];
        load16(dst, o);
#pragma unroll
        for (int s = 0; s < 16; ++s) o[s] += a2[s];
        if (valid) store16(dst, o);
    }
}

// ---- operand packing: PyTorch [out, in] (row stride ld, first column col0) -> [out block][quad][lane] float4 in the natural maps
// of dgt_pack.cpp (register R of half h = feature (R / 16) 32 + 16 h + R % 16 on both sides); rows >= n_out are zero
struct PackItem { const float* w; int ld, col0, n_out, nb, kq; unsigned dst; int trans, n_in; };   // dst in floats, nb output blocks, kq quads per block;
                                                                                                  // trans: element (row, col) = w[col * ld + col0 + row] (W^T), cols >= n_in are zero
struct PackArgs { PackItem it[8]; int first[9]; float* out; const float *means, *stds; int De; unsigned tab; };

__global__ __launch_bounds__(64) void k_pack(PackArgs P) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= P.first[8]) {                                            // the Gaussian table [3][De]: mu | 1 / sigma | 1 / (sqrt(2 * 3.14159) sigma)
        for (int k = lane; k < P.De; k += 64) {
            float mu = 0.f, is = 1.f, cf = 0.f;
            if (k > 0) {
                const float sd = fabsf(P.stds[k - 1]) + 1e-5f;
                mu = P.means[k - 1]; is = 1.f / sd; cf = 1.f / (2.5066272f * sd);
            }
            P.out[P.tab + k] = mu; P.out[P.tab + P.De + k] = is; P.out[P.tab + 2 * P.De + k] = cf;
        }
        return;
    }
    int m = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (q >= P.first[i]) m = i;
    const PackItem it = P.it[m];
    const int loc = q - P.first[m], b = loc / it.kq, kq = loc % it.kq;
    const int i = lane & 31, kh = lane >> 5, oh = (i >> 2) & 1, os = (i & 3) + 4 * (i >> 3);
    const int row = b * 32 + oh * 16 + os;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int R = kq * 4 + c, col = (R / 16) * 32 + kh * 16 + (R % 16);
        v[c] = (row < it.n_out && col < it.n_in) ? (it.trans ? it.w[(long)col * it.ld + it.col0 + row] : it.w[(long)row * it.ld + it.col0 + col]) : 0.f;
    }
    reinterpret_cast<float4*>(P.out + it.dst)[(long)loc * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

Common common_of(const jt::FusedDims& d, const jt::FusedTopo& t, const float* packed) {
    // (t.save: jt::FusedTopo carries the forward's save flag; the backward chains ignore it)
    const jt::FusedPackLayout L = jt::fused_pack_layout(d);
    Common c;
    c.R = t.R; c.ea = t.edge_a; c.ec = t.edge_c; c.em = t.edge_mol; c.packed = packed;
    c.oEE = (unsigned)(L.ee * 4); c.oL0 = (unsigned)(L.l0 * 4); c.oL1 = (unsigned)(L.l1 * 4); c.oF3 = (unsigned)(L.ff3 * 4);
    c.oF4 = (unsigned)(L.ff4 * 4); c.oRO = (unsigned)(L.ero * 4); c.oIN = (unsigned)(L.in_eg * 4); c.oC0 = (unsigned)(L.c0 * 4);
    c.oC0T = (unsigned)(L.c0t * 4); c.oINT = (unsigned)(L.int_eg * 4); c.oF4T = (unsigned)(L.ff4t * 4); c.oF3T = (unsigned)(L.ff3t * 4);
    c.oL1T = (unsigned)(L.l1t * 4); c.oL0T = (unsigned)(L.l0t * 4); c.oEET = (unsigned)(L.eet * 4);
    c.tab = packed + L.tab;
    c.save = t.save;
    return c;
}

}  // namespace

namespace jt {

FusedPackLayout fused_pack_layout(const FusedDims& d) {
    FusedPackLayout L;
    const size_t De = d.De, D = d.D, qkp = (size_t)(d.QK + 31) / 32 * 32;
    size_t o = 0;
    L.ee = o; o += De * 2 * De;
    L.l0 = o; o += qkp * De;
    L.l1 = o; o += D * De;
    L.ff3 = o; o += (size_t)d.r * De * De;
    L.ff4 = o; o += De * d.r * De;
    L.ero = o; o += 32 * De;
    L.in_eg = o; o += D * 2 * De;
    L.c0 = o; o += D * D;
    L.tab = o; o += 3 * De;
    L.total = (o + 63) / 64 * 64;
    o = L.total;
    L.c0t = o; o += D * D;                       // out D (inputs of coord_mlp.0), K = D
    L.int_eg = o; o += 2 * De * D;               // out 2 De ([e ; G] columns of input_lin), K = D
    L.ff4t = o; o += (size_t)d.r * De * De;      // out r De, K = De
    L.ff3t = o; o += De * d.r * De;              // out De, K = r De
    L.l1t = o; o += De * D;                      // out De, K = D
    L.l0t = o; o += De * D;                      // out De, K = D (lin_edge0 has QK <= D rows: the rest of K is zero)
    L.eet = o; o += 2 * De * De;                 // out 2 De ([G ; e] columns of edge_emb), K = De
    L.total_bwd = (o + 63) / 64 * 64;
    return L;
}

bool fused_available(const FusedDims& d) {
    return (d.D == 128 || d.D == 256 || d.D == 384) && (d.r == 2 || d.r == 4) && d.De == d.D / 4 && d.ce >= 1 && d.ce <= 32 && d.QK % 2 == 0 &&
           d.QK <= d.D && (d.r * d.De) % 64 == 0;
}

void fused_pack_block(hipStream_t s, const FusedDims& d, const FusedBlockParams& p, float* packed) {
    const FusedPackLayout L = fused_pack_layout(d);
    const int De = d.De, D = d.D;
    PackArgs P;
    auto item = [](const float* w, int ld, int col0, int n_out, int K, size_t dst) {
        PackItem it; it.w = w; it.ld = ld; it.col0 = col0; it.n_out = n_out; it.nb = (n_out + 31) / 32; it.kq = K / 8; it.dst = (unsigned)dst; it.trans = 0; it.n_in = K; return it;
    };
    P.it[0] = item(p.edge_emb_w, 2 * De, 0, De, 2 * De, L.ee);
    P.it[1] = item(p.le0, De, 0, d.QK, De, L.l0);
    P.it[2] = item(p.le1, De, 0, D, De, L.l1);
    P.it[3] = item(p.ff3_w, De, 0, d.r * De, De, L.ff3);
    P.it[4] = item(p.ff4_w, d.r * De, 0, De, d.r * De, L.ff4);
    P.it[5] = item(p.ero_w, De, 0, d.ce, De, L.ero);
    P.it[6] = item(p.in_w, 2 * D + 2 * De, 2 * D, D, 2 * De, L.in_eg);
    P.it[7] = item(p.c0_w, D, 0, D, D, L.c0);
    P.first[0] = 0;
    for (int i = 0; i < 8; ++i) P.first[i + 1] = P.first[i] + P.it[i].nb * P.it[i].kq;
    P.out = packed; P.means = p.gbf_means; P.stds = p.gbf_stds; P.De = De; P.tab = (unsigned)L.tab;
    hipLaunchKernelGGL(k_pack, dim3(P.first[8] + 1), dim3(64), 0, s, P);
}

void fused_chain_a(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* pos, const float* gm,
                   const float* e_in, const float* emod, float* d2, float* G, float* xh_e1, float* rs_e1, float* et, float* t0, float* t1) {
    ArgsA A;
    A.c = common_of(d, t, packed);
    A.pos = pos; A.gm = gm; A.e_in = e_in; A.emod = emod; A.ee_b = p.edge_emb_b;
    A.d2 = d2; A.G = G; A.xh = xh_e1; A.rs = rs_e1; A.et = et; A.t0 = t0; A.t1 = t1; A.QK = d.QK;
    const dim3 grid((unsigned)((t.R + 31) / 32));
    if (d.D == 128) hipLaunchKernelGGL(k_chain_a<128>, grid, dim3(64), 0, s, A);
    else if (d.D == 256) hipLaunchKernelGGL(k_chain_a<256>, grid, dim3(64), 0, s, A);
    else hipLaunchKernelGGL(k_chain_a<384>, grid, dim3(64), 0, s, A);
}

void fused_chain_b(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* e_in, const float* n2e,
                   const float* emod, Drop drop_a3, Drop drop_f4, float* xh_en, float* rs_en, float* en, float* f3, float* a3, float* f4, float* e_out,
                   float* eh, int ld_eh, int eh_col) {
    ArgsB A;
    A.c = common_of(d, t, packed);
    A.e_in = e_in; A.n2e = n2e; A.n2e_b = p.n2e_b; A.emod = emod; A.b3 = p.ff3_b; A.b4 = p.ff4_b; A.bro = p.ero_b;
    A.d3 = drop_a3; A.d4 = drop_f4;
    A.xh = xh_en; A.rs = rs_en; A.en = en; A.f3 = f3; A.a3 = a3; A.f4 = f4; A.e_out = e_out; A.eh = eh; A.ld_eh = ld_eh; A.eh_col = eh_col; A.ce = d.ce;
    const dim3 grid((unsigned)((t.R + 31) / 32));
#define JT_B(DD, RR) hipLaunchKernelGGL((k_chain_b<DD, RR>), grid, dim3(64), 0, s, A)
    if (d.D == 128) { if (d.r == 2) JT_B(128, 2); else JT_B(128, 4); }
    else if (d.D == 256) { if (d.r == 2) JT_B(256, 2); else JT_B(256, 4); }
    else { if (d.r == 2) JT_B(384, 2); else JT_B(384, 4); }
#undef JT_B
}

void fused_chain_c(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* e_in, const float* G,
                   const float* hr, const float* hc, const float* qmod, float* xh_pre, float* rs_pre, float* u, float* c0pre, float* c0a, float* inv) {
    ArgsC A;
    A.c = common_of(d, t, packed);
    A.e_in = e_in; A.G = G; A.hr = hr; A.hc = hc; A.in_b = p.in_b; A.qmod = qmod; A.b0 = p.c0_b; A.w2 = p.c2_w;
    A.xh = xh_pre; A.rs = rs_pre; A.u = u; A.c0pre = c0pre; A.c0a = c0a; A.inv = inv;
    const dim3 grid((unsigned)((t.R + 31) / 32));
    if (d.D == 128) hipLaunchKernelGGL(k_chain_c<128>, grid, dim3(64), 0, s, A);
    else if (d.D == 256) hipLaunchKernelGGL(k_chain_c<256>, grid, dim3(64), 0, s, A);
    else hipLaunchKernelGGL(k_chain_c<384>, grid, dim3(64), 0, s, A);
}

void fused_pack_block_bwd(hipStream_t s, const FusedDims& d, const FusedBlockParams& p, float* packed) {
    const FusedPackLayout L = fused_pack_layout(d);
    const int De = d.De, D = d.D;
    PackArgs P;
    auto item = [](const float* w, int ld, int col0, int n_out, int K, int n_in, size_t dst) {
        PackItem it; it.w = w; it.ld = ld; it.col0 = col0; it.n_out = n_out; it.nb = (n_out + 31) / 32; it.kq = K / 8; it.dst = (unsigned)dst; it.trans = 1; it.n_in = n_in; return it;
    };
    P.it[0] = item(p.c0_w, D, 0, D, D, D, L.c0t);
    P.it[1] = item(p.in_w, 2 * D + 2 * De, 2 * D, 2 * De, D, D, L.int_eg);
    P.it[2] = item(p.ff4_w, d.r * De, 0, d.r * De, De, De, L.ff4t);
    P.it[3] = item(p.ff3_w, De, 0, De, d.r * De, d.r * De, L.ff3t);
    P.it[4] = item(p.le1, De, 0, De, D, D, L.l1t);
    P.it[5] = item(p.le0, De, 0, De, D, d.QK, L.l0t);
    P.it[6] = item(p.edge_emb_w, 2 * De, 0, 2 * De, De, De, L.eet);
    P.it[7] = P.it[6]; P.it[7].nb = 0;
    P.first[0] = 0;
    for (int i = 0; i < 8; ++i) P.first[i + 1] = P.first[i] + P.it[i].nb * P.it[i].kq;
    P.out = packed; P.means = nullptr; P.stds = nullptr; P.De = 0; P.tab = 0;
    hipLaunchKernelGGL(k_pack, dim3(P.first[8]), dim3(64), 0, s, P);
}

void fused_bwd_c(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* inv, float* dinv,
                 const float* c0pre, const float* xh_pre, const float* rs_pre, const float* qmod, float* dc0, float* du, float* dpre, float* de, float* dG) {
    ArgsBC A;
    A.c = common_of(d, t, packed);
    A.inv = inv; A.c0pre = c0pre; A.xh = xh_pre; A.rs = rs_pre; A.qmod = qmod; A.w2 = p.c2_w;
    A.dinv = dinv; A.dc0 = dc0; A.du = du; A.dpre = dpre; A.de = de; A.dG = dG;
    const dim3 grid((unsigned)((t.R + 31) / 32));
    if (d.D == 128) hipLaunchKernelGGL(k_bwd_c<128>, grid, dim3(64), 0, s, A);
    else if (d.D == 256) hipLaunchKernelGGL(k_bwd_c<256>, grid, dim3(64), 0, s, A);
    else hipLaunchKernelGGL(k_bwd_c<384>, grid, dim3(64), 0, s, A);
}

void fused_bwd_b(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* de_out, const float* f4,
                 const float* f3, const float* xh_en, const float* rs_en, const float* emod, Drop drop_a3, Drop drop_f4, float* f4d, float* df4, float* dhid,
                 float* den, float* de_prev) {
    (void)p;
    ArgsBB A;
    A.c = common_of(d, t, packed);
    A.de_out = de_out; A.f4 = f4; A.f3 = f3; A.xh = xh_en; A.rs = rs_en; A.emod = emod; A.d3 = drop_a3; A.d4 = drop_f4;
    A.f4d = f4d; A.df4 = df4; A.dhid = dhid; A.den = den; A.de_prev = de_prev;
    const dim3 grid((unsigned)((t.R + 31) / 32));
#define JT_B(DD, RR) hipLaunchKernelGGL((k_bwd_b<DD, RR>), grid, dim3(64), 0, s, A)
    if (d.D == 128) { if (d.r == 2) JT_B(128, 2); else JT_B(128, 4); }
    else if (d.D == 256) { if (d.r == 2) JT_B(256, 2); else JT_B(256, 4); }
    else { if (d.r == 2) JT_B(384, 2); else JT_B(384, 4); }
#undef JT_B
}

void fused_bwd_a(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* dt1, const float* dt0,
                 const float* xh_e1, const float* rs_e1, const float* emod, float* det, float* de1, float* dG, float* de_prev) {
    (void)p;
    ArgsBA A;
    A.c = common_of(d, t, packed);
    A.dt1 = dt1; A.dt0 = dt0; A.xh = xh_e1; A.rs = rs_e1; A.emod = emod; A.det = det; A.de1 = de1; A.dG = dG; A.de_prev = de_prev; A.QK = d.QK;
    const dim3 grid((unsigned)((t.R + 31) / 32));
    if (d.D == 128) hipLaunchKernelGGL(k_bwd_a<128>, grid, dim3(64), 0, s, A);
    else if (d.D == 256) hipLaunchKernelGGL(k_bwd_a<256>, grid, dim3(64), 0, s, A);
    else hipLaunchKernelGGL(k_bwd_a<384>, grid, dim3(64), 0, s, A);
}

// ---- node rows: LayerNorm + modulate, one wave per row (round 5, second half) -------------------------------------------------------
// The op-by-op form is three launches per LayerNorm (eight partial sums per row, their combination, the normalisation) and one more
// for the gated residual in front of the second one — cooperation-free kernels for the host emulation, ~5 us each whatever their size.
// Here a wave owns a row (NV = F / 64 values per lane, consecutive lanes on consecutive features), the two row means are butterfly sums.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
//   x = RES ? a + g[mol] b : a;   xhat = (x - mean) rstd (kept);   y = xhat (1 + sc[mol]) + sh[mol]
template <int NV, bool RES>
__global__ __launch_bounds__(256) void k_node_ln_mod(long rows, const float* __restrict__ a, const float* __restrict__ b, const int* __restrict__ row_mol,
                                                     const float* __restrict__ mods, int ldm, int g_off, int sh_off, int sc_off, float* __restrict__ xhat,
                                                     float* __restrict__ rstd_out, float* __restrict__ y) {
    constexpr int F = NV * 64;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* m = mods + (long)row_mol[r] * ldm;
    float x[NV], s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = lane + 64 * j;
        x[j] = a[r * F + f];
        if (RES) x[j] += m[g_off + f] * b[r * F + f];
        s += x[j];
    }
    const float mean = wave_sum(s) * (1.f / F);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) { x[j] -= mean; q += x[j] * x[j]; }
    const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / F) + 1e-6f);
    if (lane == 0) rstd_out[r] = rstd;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = lane + 64 * j;
        const float xh = x[j] * rstd;
        xhat[r * F + f] = xh;
        y[r * F + f] = xh * (1.f + m[sc_off + f]) + m[sh_off + f];
    }
}
//   g = dy (1 + sc[mol]);   dx (+)= rstd (g - mean(g) - xhat mean(g xhat))
template <int NV>
__global__ __launch_bounds__(256) void k_node_ln_mod_bwd(long rows, const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                         const int* __restrict__ row_mol, const float* __restrict__ mods, int ldm, int sc_off, float* dx, int acc) {
    constexpr int F = NV * 64;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* m = mods + (long)row_mol[r] * ldm + sc_off;
    float g[NV], xh[NV], c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = lane + 64 * j;
        g[j] = dy[r * F + f] * (1.f + m[f]);
        xh[j] = xhat[r * F + f];
        c1 += g[j]; c2 = fmaf(g[j], xh[j], c2);
    }
    c1 = wave_sum(c1) * (1.f / F); c2 = wave_sum(c2) * (1.f / F);
    const float rs = rstd[r];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = lane + 64 * j;
        const float v = rs * (g[j] - c1 - xh[j] * c2);
        dx[r * F + f] = acc ? dx[r * F + f] + v : v;
    }
}

void fused_node_ln_mod(hipStream_t s, long rows, int F, const float* a, const float* res_b, const int* row_mol, const float* mods, int ldm, int g_off,
                       int sh_off, int sc_off, float* xhat, float* rstd, float* y) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define JT_N(NV) do { if (res_b) hipLaunchKernelGGL((k_node_ln_mod<NV, true>), grid, block, 0, s, rows, a, res_b, row_mol, mods, ldm, g_off, sh_off, sc_off, xhat, rstd, y); \
                      else hipLaunchKernelGGL((k_node_ln_mod<NV, false>), grid, block, 0, s, rows, a, res_b, row_mol, mods, ldm, g_off, sh_off, sc_off, xhat, rstd, y); } while (0)
    if (F == 128) JT_N(2); else if (F == 256) JT_N(4); else JT_N(6);
#undef JT_N
}
void fused_node_ln_mod_bwd(hipStream_t s, long rows, int F, const float* dy, const float* xhat, const float* rstd, const int* row_mol, const float* mods,
                           int ldm, int sc_off, float* dx, int acc) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (F == 128) hipLaunchKernelGGL(k_node_ln_mod_bwd<2>, grid, block, 0, s, rows, dy, xhat, rstd, row_mol, mods, ldm, sc_off, dx, acc);
    else if (F == 256) hipLaunchKernelGGL(k_node_ln_mod_bwd<4>, grid, block, 0, s, rows, dy, xhat, rstd, row_mol, mods, ldm, sc_off, dx, acc);
    else hipLaunchKernelGGL(k_node_ln_mod_bwd<6>, grid, block, 0, s, rows, dy, xhat, rstd, row_mol, mods, ldm, sc_off, dx, acc);
}

// ---- attention of a block (layers.py:131-186), forward in one launch and backward in two (round 5, second half) ----------------------
// The op-by-op form is scores | column softmax | messages (three launches, the [R, H] scores through memory twice) and six launches
// backwards; each walks the molecule's n x n tile with one thread per output.  Here a wave owns one TARGET atom c (forward, and the
// target-side half of the backward) or one SOURCE atom a (the other half): its rows (a, c) are read once per pass, the [n, H] scores /
// weights of the atom live in LDS, and every sum runs in the order of the op-by-op kernels — the results are bit-identical to them.
// One wave per workgroup (its __syncthreads() is a wave barrier), n <= ATT_NMAX.
constexpr int ATT_NMAX = 192;

__device__ __forceinline__ void att_ids(const AttnTopo& t, int node, int& b, int& n, int& i, long& e0, long& n0) {
    b = t.node_mol[node]; n = t.nn[b]; n0 = t.node_off[b]; i = node - (int)n0; e0 = t.edge_off[b];
}

// forward: S[(a, c), hd] -> alpha (kept) -> hhat[c, f] = sum_a v[a, f] t1[(a, c), f] alpha[(a, c), f / C].
// W waves per target: they share the sources in the score pass and the FEATURES in the message pass (NV / W of the NV 64-feature slices
// each), so no sum changes its order.
template <int NV, int W>
__global__ __launch_bounds__(64 * W) void k_attn_fwd(AttnTopo t, int H, int XH, int SC, float inv_sqrt_c, const float* __restrict__ q, const float* __restrict__ k,
                                                     const float* __restrict__ t0, const float* __restrict__ adj2d, const float* __restrict__ adjsp,
                                                     const float* __restrict__ v, const float* __restrict__ t1, float* __restrict__ alpha, float* __restrict__ hhat) {
    constexpr int D = NV * 64, NW = NV / W;
    static_assert(NV % W == 0, "the waves of a target share the 64-feature slices evenly");
    extern __shared__ float att_lds[];                            // [N][16] scores / weights of this target (N = the batch's largest molecule) | max, sum
    float (*A)[16] = reinterpret_cast<float (*)[16]>(att_lds);
    float (*MS)[16] = reinterpret_cast<float (*)[16]>(att_lds + (size_t)t.N * 16);
    const int node = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = tid >> 4, hd = tid & 15;      // g: 0 .. 4 W - 1
    int b, n, c; long e0, n0;
    att_ids(t, node, b, n, c, e0, n0);
    const int QK = (H - XH) * SC, C = D / H;
    // scores: thread (g, hd) takes head hd of the sources a = g, g + 4 W, ...
    if (hd < H) {
        for (int a = g; a < n; a += 4 * W) {
            const long r = e0 + (long)a * n + c;
            float s;
            if (hd < XH) s = ((hd == 0 ? adj2d[r] : adjsp[r]) > 0.f) ? 1.f : -1e10f;
            else {
                const float* qq = q + (long)node * t.ldqk + (hd - XH) * SC;
                const float* kk = k + (n0 + a) * t.ldqk + (hd - XH) * SC;
                const float* tt = t0 + r * QK + (hd - XH) * SC;
                float acc = 0.f;
#pragma unroll 6
                for (int j = 0; j < SC; ++j) acc += qq[j] * kk[j] * tt[j];
                s = acc * inv_sqrt_c;
            }
            A[a][hd] = s;
        }
    }
    __syncthreads();
    // column softmax over the sources a != c, one lane per head, sequentially like k_attn_softmax
    if (tid < H) {
        float m = -INFINITY;
        for (int a = 0; a < n; ++a) if (a != c) m = fmaxf(m, A[a][tid]);
        float sum = 0.f;
        for (int a = 0; a < n; ++a) if (a != c) sum += expf(A[a][tid] - m);
        MS[0][tid] = m; MS[1][tid] = sum;
    }
    __syncthreads();
    if (hd < H) {
        const float m = MS[0][hd], sum = MS[1][hd];
        for (int a = g; a < n; a += 4 * W) {
            const float wgt = a == c ? 0.f : expf(A[a][hd] - m) / (sum + 1e-16f);
            A[a][hd] = wgt;
            alpha[(e0 + (long)a * n + c) * H + hd] = wgt;
        }
    }
    __syncthreads();
    // messages: wave w takes the features lane + 64 j, j = w NW .. (w + 1) NW - 1
    float acc[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int a = 0; a < n; ++a) {
        const long r = e0 + (long)a * n + c;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int f = lane + 64 * (w * NW + j);
            acc[j] += v[(n0 + a) * t.ldv + f] * t1[r * D + f] * A[a][f / C];
        }
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) hhat[(long)node * D + lane + 64 * (w * NW + j)] = acc[j];
}

// backward, target side (W waves per target c): d alpha -> d S (softmax backward, kept for the source side) -> d t1, d q, d t0.
// W = 4 at small batches (a QM9 batch of 128 molecules has 2 260 atoms: two waves per SIMD cannot hide their own load latency): the
// waves share the sources (a = w, w + W, ...), the partial d q sums meet in LDS in the order w = 0 .. W - 1.
template <int NV, int W>
__global__ __launch_bounds__(64 * W) void k_attn_bwd_tgt(AttnTopo t, int H, int XH, int SC, float inv_sqrt_c, const float* __restrict__ dhhat,
                                                         const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         const float* __restrict__ t0, const float* __restrict__ t1, const float* __restrict__ alpha,
                                                         float* __restrict__ dS, float* __restrict__ dt1, float* __restrict__ dq, float* __restrict__ dt0) {
    constexpr int D = NV * 64;
    extern __shared__ float att_lds[];                            // alpha [N][16] | d alpha, then d S [N][16] | 16 dots | W partial rows of d q
    float (*A)[16] = reinterpret_cast<float (*)[16]>(att_lds);
    float (*G)[16] = reinterpret_cast<float (*)[16]>(att_lds + (size_t)t.N * 16);
    float* DOT = att_lds + (size_t)t.N * 32;
    float* PQ = DOT + 16;
    const int node = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = (tid >> 4), hd = tid & 15;      // g: 0 .. 4 W - 1
    int b, n, c; long e0, n0;
    att_ids(t, node, b, n, c, e0, n0);
    const int QK = (H - XH) * SC, C = D / H;
    if (hd < H) {
        const float* dh = dhhat + (long)node * D + hd * C;
        for (int a = g; a < n; a += 4 * W) {
            const long r = e0 + (long)a * n + c;
            const float* vv = v + (n0 + a) * t.ldv + hd * C;
            const float* tt = t1 + r * D + hd * C;
            float s = 0.f;
#pragma unroll 8
            for (int j = 0; j < C; ++j) s += dh[j] * vv[j] * tt[j];
            G[a][hd] = s;
            A[a][hd] = alpha[r * H + hd];
        }
    }
    __syncthreads();
    if (tid < H) {
        float dot = 0.f;
        for (int a = 0; a < n; ++a) dot += A[a][tid] * G[a][tid];
        DOT[tid] = dot;
    }
    __syncthreads();
    if (hd < H) {
        const float dot = DOT[hd];
        for (int a = g; a < n; a += 4 * W) {
            const float ds = A[a][hd] * (G[a][hd] - dot);
            G[a][hd] = ds;
            dS[(e0 + (long)a * n + c) * H + hd] = ds;
        }
    }
    __syncthreads();
    float dhv[NV], qv[NV], sq[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = lane + 64 * j;
        dhv[j] = dhhat[(long)node * D + f];
        qv[j] = f < QK ? q[(long)node * t.ldqk + f] : 0.f;
        sq[j] = 0.f;
    }
#pragma unroll 4
    for (int a = w; a < n; a += W) {
        const long r = e0 + (long)a * n + c;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = lane + 64 * j;
            const float tv = t1[r * D + f];
            dt1[r * D + f] = dhv[j] * v[(n0 + a) * t.ldv + f] * A[a][f / C] * (1.f - tv * tv);
            if (f < QK) {
                const float ds = G[a][XH + f / SC], kv = k[(n0 + a) * t.ldqk + f], t0v = t0[r * QK + f];
                sq[j] += ds * kv * t0v;
                dt0[r * QK + f] = ds * qv[j] * kv * inv_sqrt_c * (1.f - t0v * t0v);
            }
        }
    }
    if (W > 1) {
#pragma unroll
        for (int j = 0; j < NV; ++j) PQ[w * D + lane + 64 * j] = sq[j];
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float tsum = PQ[lane + 64 * j];
                for (int x = 1; x < W; ++x) tsum += PQ[x * D + lane + 64 * j];
                sq[j] = tsum;
            }
        }
    }
    if (w == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = lane + 64 * j;
            if (f < QK) dq[(long)node * t.ldd_qk + f] = sq[j] * inv_sqrt_c;
        }
    }
}

// backward, source side (W waves per source a): d v[a, f] = sum_c dhhat[c, f] t1[(a, c), f] alpha[(a, c), f / C];
// d k[a, j] = sum_c d S[(a, c), hd(j)] q[c, j] t0[(a, c), j] / sqrt(C)
template <int NV, int W>
__global__ __launch_bounds__(64 * W) void k_attn_bwd_src(AttnTopo t, int H, int XH, int SC, float inv_sqrt_c, const float* __restrict__ dhhat,
                                                         const float* __restrict__ q, const float* __restrict__ t0, const float* __restrict__ t1,
                                                         const float* __restrict__ alpha, const float* __restrict__ dS, float* __restrict__ dv,
                                                         float* __restrict__ dk) {
    constexpr int D = NV * 64;
    extern __shared__ float att_lds[];                            // W partial rows of d v | of d k
    const int node = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int b, n, a; long e0, n0;
    att_ids(t, node, b, n, a, e0, n0);
    const int QK = (H - XH) * SC, C = D / H;
    float sv[NV], sk[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { sv[j] = 0.f; sk[j] = 0.f; }
#pragma unroll 4
    for (int c = w; c < n; c += W) {
        const long r = e0 + (long)a * n + c;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = lane + 64 * j;
            sv[j] += dhhat[(n0 + c) * D + f] * t1[r * D + f] * alpha[r * H + f / C];
            if (f < QK) sk[j] += dS[r * H + XH + f / SC] * q[(n0 + c) * t.ldqk + f] * t0[r * QK + f];
        }
    }
    if (W > 1) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { att_lds[w * D + lane + 64 * j] = sv[j]; att_lds[(W + w) * D + lane + 64 * j] = sk[j]; }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float tv = att_lds[lane + 64 * j], tk = att_lds[W * D + lane + 64 * j];
                for (int x = 1; x < W; ++x) { tv += att_lds[x * D + lane + 64 * j]; tk += att_lds[(W + x) * D + lane + 64 * j]; }
                sv[j] = tv; sk[j] = tk;
            }
        }
    }
    if (w == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = lane + 64 * j;
            dv[(long)node * t.ldd_v + f] = sv[j];
            if (f < QK) dk[(long)node * t.ldd_qk + f] = sk[j] * inv_sqrt_c;
        }
    }
}

bool fused_attention_available(int D, int H, int N) { return (D == 128 || D == 256 || D == 384) && H <= 16 && N <= ATT_NMAX; }

void fused_attn_fwd(hipStream_t s, const AttnTopo& t, int D, int H, int XH, int SC, float inv_sqrt_c, const float* q, const float* k, const float* t0,
                    const float* adj2d, const float* adjsp, const float* v, const float* t1, float* alpha, float* hhat) {
    const dim3 grid((unsigned)t.Nn);
    const unsigned lds = (unsigned)(((size_t)t.N * 16 + 32) * sizeof(float));
    const bool wide = t.Nn < 16384 && !t.one_wave;                          // few atoms: several waves per target (same arithmetic)
#define JT_AF(NV, WW) hipLaunchKernelGGL((k_attn_fwd<NV, WW>), grid, dim3(64 * WW), lds, s, t, H, XH, SC, inv_sqrt_c, q, k, t0, adj2d, adjsp, v, t1, alpha, hhat)
    if (D == 128) { if (wide) JT_AF(2, 2); else JT_AF(2, 1); }
    else if (D == 256) { if (wide) JT_AF(4, 4); else JT_AF(4, 1); }
    else { if (wide) JT_AF(6, 3); else JT_AF(6, 1); }
#undef JT_AF
}
void fused_attn_bwd(hipStream_t s, const AttnTopo& t, int D, int H, int XH, int SC, float inv_sqrt_c, const float* dhhat, const float* q, const float* k,
                    const float* v, const float* t0, const float* t1, const float* alpha, float* dS, float* dt1, float* dt0, float* dq, float* dk, float* dv) {
    const dim3 grid((unsigned)t.Nn);
    // few atoms (the reference's training batches): four waves per atom; batches that fill the card on their own: one
    const int W = (t.Nn < 16384 && !t.one_wave) ? 4 : 1;
    const unsigned lds_t = (unsigned)(((size_t)t.N * 32 + 16 + (size_t)W * D) * sizeof(float)), lds_s = (unsigned)((size_t)2 * W * D * sizeof(float));
#define JT_AB(NV, WW) do { \
        hipLaunchKernelGGL((k_attn_bwd_tgt<NV, WW>), grid, dim3(64 * WW), lds_t, s, t, H, XH, SC, inv_sqrt_c, dhhat, q, k, v, t0, t1, alpha, dS, dt1, dq, dt0); \
        hipLaunchKernelGGL((k_attn_bwd_src<NV, WW>), grid, dim3(64 * WW), lds_s, s, t, H, XH, SC, inv_sqrt_c, dhhat, q, t0, t1, alpha, (const float*)dS, dv, dk); } while (0)
    if (W == 4) { if (D == 128) JT_AB(2, 4); else if (D == 256) JT_AB(4, 4); else JT_AB(6, 4); }
    else { if (D == 128) JT_AB(2, 1); else if (D == 256) JT_AB(4, 1); else JT_AB(6, 1); }
#undef JT_AB
}

// ---- Gaussian layer, backward (layers.py CondGaussianLayer; train_ops.h k_gbf_bwd_row + k_gbf_bwd_par) in one pass -------------------
// The op-by-op pair evaluates every Gaussian twice (once per row for d x', once per (chunk, k) for the parameter gradients) with one
// thread walking a row's 63 terms, resp. a chunk's 32 rows.  Here a wave owns a 32-row chunk, lane = Gaussian k (k + 64 for De = 96): the
// row's dG is one coalesced load, d x' is a butterfly sum over the lanes (in double, like the sequential sum it replaces), the
// parameter partials of the chunk accumulate per lane in double and land where k_gbf_bwd_par put them.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <int KV>
__global__ __launch_bounds__(256) void k_gbf_bwd_chunk(long rows, int De, const float* __restrict__ d2, const int* __restrict__ row_mol, const float* __restrict__ gm,
                                                       const float* __restrict__ means, const float* __restrict__ stds, const float* __restrict__ dG, int ldg,
                                                       int gcol, float* __restrict__ dxp, float* dd2, int acc, float* __restrict__ part_m,
                                                       float* __restrict__ part_s) {
    const int K = De - 1, lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nchunks = (rows + 31) / 32;
    if (c >= nchunks) return;
    const long r0 = c * 32, r1 = r0 + 32 < rows ? r0 + 32 : rows;
    float mk[KV], sd[KV], sg[KV];
    double dm[KV], ds[KV];
#pragma unroll
    for (int j = 0; j < KV; ++j) {
        const int k = lane + 64 * j;
        const float w = k < K ? stds[k] : 1.f;
        mk[j] = k < K ? means[k] : 0.f;
        sd[j] = fabsf(w) + 1e-5f; sg[j] = w < 0.f ? -1.f : (w > 0.f ? 1.f : 0.f);
        dm[j] = 0.0; ds[j] = 0.0;
    }
    for (long r = r0; r < r1; ++r) {
        const float* g = gm + (long)row_mol[r] * 2;
        const float g0 = g[0] + 1.f, x = d2[r] * g0 + g[1];
        const float* d = dG + r * ldg + gcol;
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < KV; ++j) {
            const int k = lane + 64 * j;
            if (k < K) {
                const float z = (x - mk[j]) / sd[j];
                const float gk = expf(-0.5f * (z * z)) / (2.5066272f * sd[j]);
                const float dv = d[1 + k];
                part += (double)(dv * gk * (-z / sd[j]));
                const float dg = dv * gk;
                dm[j] += (double)(dg * (z / sd[j]));
                ds[j] += (double)(dg * ((z * z - 1.f) / sd[j]) * sg[j]);
            }
        }
        const double sum = wave_sum_d(part) + (double)d[0];
        if (lane == 0) {
            const float sv = (float)sum;
            dxp[r] = sv;
            if (dd2) dd2[r] = (acc ? dd2[r] : 0.f) + sv * g0;
        }
    }
#pragma unroll
    for (int j = 0; j < KV; ++j) {
        const int k = lane + 64 * j;
        if (k < K) { part_m[c * K + k] = (float)dm[j]; part_s[c * K + k] = (float)ds[j]; }
    }
}
void fused_gbf_bwd(hipStream_t s, long rows, int De, const float* d2, const int* row_mol, const float* gm, const float* means, const float* stds, const float* dG,
                   int ldg, int gcol, float* dxp, float* dd2, int acc, float* part_m, float* part_s) {
    const long nchunks = (rows + 31) / 32;
    const dim3 grid((unsigned)((nchunks + 3) / 4)), block(256);
    if (De - 1 <= 64) hipLaunchKernelGGL(k_gbf_bwd_chunk<1>, grid, block, 0, s, rows, De, d2, row_mol, gm, means, stds, dG, ldg, gcol, dxp, dd2, acc, part_m, part_s);
    else hipLaunchKernelGGL(k_gbf_bwd_chunk<2>, grid, block, 0, s, rows, De, d2, row_mol, gm, means, stds, dG, ldg, gcol, dxp, dd2, acc, part_m, part_s);
}

}  // namespace jt
