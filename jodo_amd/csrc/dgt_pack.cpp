// jodo_dgt_pack_weights: the reference's state_dict (parameter names and shapes of DGT_concat / Cond_DGT_concat,
// /root/reference/models/mol_gnn.py:414-489, :601-684) -> the single fp32 blob + offset table jodo_dgt_forward reads.
//
// Every dense projection runs on v_mfma_f32_32x32x2_f32 in the transposed orientation
//     D[out_feature, item] += W[out_feature, k] * X[k, item]
// with the weights as A operand: lane l supplies W[row(l & 31)][k-slot l >> 5].  A packed projection is
// float [n_out_blocks][ksteps / 4][64 lanes][4] (one 16-byte load per lane feeds four MFMAs).  Which matrix row /
// column a (block, lane, k-step) reads is given by two index maps:
//     in_map  [ksteps][2]      input column of k-step R for half-lane h   (-1 = zero)
//     out_map [blocks][2][16]  output row held by register s of half h    (-1 = zero)
// "natural" maps put feature (R / 16) * 32 + h * 16 + R % 16 in register R of half h, so that memory stays in natural
// order and the accumulator of one projection is the B operand of the next.  Host-only code; no device work except
// the final upload in jodo_dgt_pack_weights.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "dgt_plan.h"
#include "jodo_hip_internal.h"

namespace {

typedef std::vector<int64_t> IMap;      // in_map  [ksteps][2] flattened
typedef std::vector<int64_t> OMap;      // out_map [nb][2][16] flattened

IMap nat_in(int n, int64_t base = 0) {
    IMap m((size_t)n);                   // n/2 k-steps x 2
    for (int R = 0; R < n / 2; ++R)
        for (int h = 0; h < 2; ++h) m[(size_t)R * 2 + h] = base + (R / 16) * 32 + h * 16 + (R % 16);
    return m;
}
OMap nat_out(int n, int n_valid = -1) {
    OMap m((size_t)n);
    for (int b = 0; b < n / 32; ++b)
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < 16; ++s) {
                const int64_t f = b * 32 + h * 16 + s;
                m[((size_t)b * 2 + h) * 16 + s] = (n_valid >= 0 && f >= n_valid) ? -1 : f;
            }
    return m;
}
// short feature group (n <= 32, padded to a multiple of 8): half h, register s holds feature h * (npad / 2) + s
IMap small_in(int n, int64_t base = 0) {
    const int npad = (n + 7) / 8 * 8, half = npad / 2;
    IMap m((size_t)npad, -1);
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < half; ++s) {
            const int f = h * half + s;
            if (f < n) m[(size_t)s * 2 + h] = base + f;
        }
    return m;
}
IMap cat(const IMap& a, const IMap& b) { IMap o(a); o.insert(o.end(), b.begin(), b.end()); return o; }
// tuned nf = 256 arrangement of the SH x SC score features (reduction per head stays register-local):
// blocks 0 .. SH/2-1: half h of block b = channels 0..15 of head 2b + h; tail channel c of head g in block SH/2 + c/2,
// half c % 2, register g
OMap qk_out(int SH, int SC) {
    const int tail = SC - 16, nb = SH / 2 + (tail + 1) / 2;
    OMap m((size_t)nb * 32, -1);
    for (int b = 0; b < SH / 2; ++b)
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < 16; ++s) m[((size_t)b * 2 + h) * 16 + s] = (int64_t)(2 * b + h) * SC + s;
    for (int c = 0; c < tail; ++c)
        for (int g = 0; g < SH; ++g) m[((size_t)(SH / 2 + c / 2) * 2 + (c % 2)) * 16 + g] = (int64_t)g * SC + 16 + c;
    return m;
}
// width-generic arrangement: head g owns block g in natural order, rows >= SC are padding
OMap qk_out_wide(int SH, int SC) {
    OMap m((size_t)SH * 32, -1);
    for (int g = 0; g < SH; ++g)
        for (int h = 0; h < 2; ++h)
            for (int s = 0; s < 16; ++s)
                if (h * 16 + s < SC) m[((size_t)g * 2 + h) * 16 + s] = (int64_t)g * SC + h * 16 + s;
    return m;
}
// first layer of a head MLP: padded activation layout [base | L x pad] -> true column base + l * cnt + c
IMap hid_in(int base, int L, int cnt, int pad) {
    const int width = base + L * pad;
    std::vector<int64_t> col((size_t)width, -1);
    for (int i = 0; i < base; ++i) col[i] = i;
    for (int l = 0; l < L; ++l)
        for (int c = 0; c < cnt; ++c) col[base + l * pad + c] = base + l * cnt + c;
    IMap nat = nat_in(width);
    for (auto& v : nat) v = col[(size_t)v];
    return nat;
}

// Rotated LayerNorm statistics of equi_update.  pre = W_in [h_a ; h_c ; e ; G] + b enters its LayerNorm only
// through P pre (P = I - 11^T / D).  With the Householder QR factorisation  Q (P W_eg) = [L ; 0]  of the centred [e ; G] columns
// (Q orthogonal D x D, L upper triangular KL x KL, KL = 2 De):  |P pre|^2 = |L z + (Q P R)[:KL] + (Q P C)[:KL]|^2 + |(Q P R + Q P C)[KL:]|^2,
// so the per-pair projection shrinks from D x KL to a triangular KL x KL one.  Everything in double; out: row-major float matrices.
struct RotStats {
    std::vector<float> rowq, colq, bq, lq, wec, qt;     // [D,D] [D,D] [D] [KL,KL] [D,KL] [D,D] (qt[f][j] = Q[j][f])
};
RotStats rot_stats(const float* Win, const float* bin, int D, int De) {
    const int KL = 2 * De, KIN = 2 * D + KL;
    std::vector<double> A((size_t)D * KL), Q((size_t)D * D, 0.0), v((size_t)D);
    for (int k = 0; k < KL; ++k) {                       // centred columns of the [e ; G] part
        double m = 0.0;
        for (int f = 0; f < D; ++f) m += Win[(size_t)f * KIN + 2 * D + k];
        m /= D;
        for (int f = 0; f < D; ++f) A[(size_t)f * KL + k] = (double)Win[(size_t)f * KIN + 2 * D + k] - m;
    }
    RotStats r;
    r.wec.resize((size_t)D * KL);
    for (size_t i = 0; i < r.wec.size(); ++i) r.wec[i] = (float)A[i];
    for (int i = 0; i < D; ++i) Q[(size_t)i * D + i] = 1.0;
    for (int k = 0; k < KL; ++k) {                       // Householder reflections H_k, Q = H_{KL-1} ... H_0
        double nrm = 0.0;
        for (int i = k; i < D; ++i) nrm += A[(size_t)i * KL + k] * A[(size_t)i * KL + k];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = A[(size_t)k * KL + k] > 0.0 ? -nrm : nrm;
        double vn = 0.0;
        for (int i = k; i < D; ++i) { v[i] = A[(size_t)i * KL + k]; if (i == k) v[i] -= alpha; vn += v[i] * v[i]; }
        if (vn == 0.0) continue;
        const double s2 = 2.0 / vn;
        for (int j = k; j < KL; ++j) {
            double s = 0.0;
            for (int i = k; i < D; ++i) s += v[i] * A[(size_t)i * KL + j];
            s *= s2;
            for (int i = k; i < D; ++i) A[(size_t)i * KL + j] -= s * v[i];
        }
        std::vector<double> sj((size_t)D, 0.0);
        for (int i = k; i < D; ++i) { const double vi = v[i]; const double* q = &Q[(size_t)i * D]; for (int j = 0; j < D; ++j) sj[j] += vi * q[j]; }
        for (int i = k; i < D; ++i) { const double vi = v[i] * s2; double* q = &Q[(size_t)i * D]; for (int j = 0; j < D; ++j) q[j] -= vi * sj[j]; }
    }
    r.lq.assign((size_t)KL * KL, 0.f);
    for (int i = 0; i < KL; ++i) for (int j = i; j < KL; ++j) r.lq[(size_t)i * KL + j] = (float)A[(size_t)i * KL + j];
    r.qt.resize((size_t)D * D);
    for (int f = 0; f < D; ++f) for (int j = 0; j < D; ++j) r.qt[(size_t)f * D + j] = (float)Q[(size_t)j * D + f];
    // Q P = Q - (Q 1) 1^T / D, then Q P W_row, Q P W_col, Q P b
    std::vector<double> QP(Q);
    for (int i = 0; i < D; ++i) {
        double m = 0.0;
        for (int f = 0; f < D; ++f) m += Q[(size_t)i * D + f];
        m /= D;
        for (int f = 0; f < D; ++f) QP[(size_t)i * D + f] -= m;
    }
    r.rowq.resize((size_t)D * D); r.colq.resize((size_t)D * D); r.bq.resize((size_t)D);
    std::vector<double> acc((size_t)2 * D);
    for (int i = 0; i < D; ++i) {
        std::fill(acc.begin(), acc.end(), 0.0);
        double b = 0.0;
        for (int f = 0; f < D; ++f) {
            const double qf = QP[(size_t)i * D + f];
            const float* w = Win + (size_t)f * KIN;      // columns 0..2D-1: h_row | h_col
            for (int c = 0; c < 2 * D; ++c) acc[c] += qf * (double)w[c];
            b += qf * (double)bin[f];
        }
        for (int c = 0; c < D; ++c) { r.rowq[(size_t)i * D + c] = (float)acc[c]; r.colq[(size_t)i * D + c] = (float)acc[D + c]; }
        r.bq[i] = (float)b;
    }
    return r;
}

struct Packer {
    std::vector<float> blob;
    std::vector<int64_t> offs;        // in put() order
    void put(const float* src, size_t n) {
        offs.push_back((int64_t)blob.size());
        blob.insert(blob.end(), src, src + n);
        blob.resize((blob.size() + 63) / 64 * 64, 0.f);     // every slot 256-byte aligned
    }
    void put(const std::vector<float>& v) { put(v.data(), v.size()); }
    void put_zero1() { const float z = 0.f; put(&z, 1); }
    // W [n_out][ld] row-major
    void put_proj(const float* W, int64_t ld, const IMap& in, const OMap& out) {
        const size_t ksteps = in.size() / 2, nb = out.size() / 32;
        std::vector<float> p(nb * ksteps * 64);
        for (size_t b = 0; b < nb; ++b)
            for (size_t kq = 0; kq < ksteps / 4; ++kq)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, kh = lane >> 5, oh = (i >> 2) & 1, os = (i & 3) + 4 * (i >> 3);
                    const int64_t row = out[(b * 2 + oh) * 16 + os];
                    for (int c = 0; c < 4; ++c) {
                        const int64_t col = in[(kq * 4 + c) * 2 + kh];
                        p[((b * (ksteps / 4) + kq) * 64 + lane) * 4 + c] = (row < 0 || col < 0) ? 0.f : W[row * ld + col];
                    }
                }
        put(p);
    }
    void put_vec(const float* v, const OMap& out) {
        std::vector<float> p(out.size());
        for (size_t i = 0; i < out.size(); ++i) p[i] = out[i] < 0 ? 0.f : v[out[i]];
        put(p);
    }
};

// ---- split-bf16 packing (opt-in bf16x3 form of a projection, dgt_split.h) ----
// float -> bf16 bits, round to nearest even (finite inputs)
inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_float(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// w = hi + mid + lo, each term rounded to nearest; the residuals w - hi and (w - hi) - mid are exact in fp32
inline void split3(float w, uint16_t (&t)[3]) {
    t[0] = bf16_rne(w);
    const float r1 = w - bf16_to_float(t[0]);
    t[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_to_float(t[1]);
    t[2] = bf16_rne(r2);
}
// W [n_out][ld] row-major -> [out block][K16 step][term][64 lanes][8 bf16]: lane l supplies row (l & 31) (the accumulator image of
// put_proj) and the input columns of the f32 k-steps 8 G + j, j = 0..7, of half l >> 5 — the same in_map / out_map as the f32 form.
std::vector<uint16_t> pack_proj_split(const float* W, int64_t ld, const IMap& in, const OMap& out) {
    const size_t ksteps = in.size() / 2, nb = out.size() / 32, ns = ksteps / 8;
    std::vector<uint16_t> p(nb * ns * 3 * 64 * 8, 0);
    for (size_t b = 0; b < nb; ++b)
        for (size_t g = 0; g < ns; ++g)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, kh = lane >> 5, oh = (i >> 2) & 1, os = (i & 3) + 4 * (i >> 3);
                const int64_t row = out[(b * 2 + oh) * 16 + os];
                for (int j = 0; j < 8; ++j) {
                    const int64_t col = in[(g * 8 + j) * 2 + kh];
                    uint16_t t[3] = {0, 0, 0};
                    if (row >= 0 && col >= 0) split3(W[row * ld + col], t);
                    for (int term = 0; term < 3; ++term) p[(((b * ns + g) * 3 + term) * 64 + lane) * 8 + j] = t[term];
                }
            }
    return p;
}

struct Lookup {
    std::unordered_map<std::string, const jodo_tensor*> m;
    std::string missing;
    const float* get(const std::string& name, int64_t numel) {
        auto it = m.find(name);
        if (it == m.end()) { if (missing.empty()) missing = name; return nullptr; }
        int64_t n = 1;
        for (int i = 0; i < it->second->ndim; ++i) n *= it->second->shape[i];
        if (n != numel) { if (missing.empty()) missing = name + " (wrong size)"; return nullptr; }
        return it->second->data;
    }
};

std::vector<float> gbf_table(Lookup& lk, const std::string& prefix, int De) {
    // [3][De]: mu, 1 / sigma, 1 / (sqrt(2 * 3.14159) * sigma), sigma = |w| + 1e-5 (layers.py:291-295, :328-334);
    // entry 0 of each row belongs to feature x' itself
    std::vector<float> tab((size_t)3 * De, 0.f);
    const float* mu = lk.get(prefix + ".means.weight", De - 1);
    const float* sd = lk.get(prefix + ".stds.weight", De - 1);
    if (!mu || !sd) return tab;
    const double a = std::pow(2 * 3.14159, 0.5);
    for (int k = 1; k < De; ++k) {
        const double sg = std::fabs((double)sd[k - 1]) + 1e-5;
        tab[k] = mu[k - 1];
        tab[De + k] = (float)(1.0 / sg);
        tab[2 * De + k] = (float)(1.0 / (a * sg));
    }
    tab[De] = 1.f;
    return tab;
}

int pack(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, Packer& P, std::vector<int64_t>& woff) {
    DgtDims d;
    int rc = dgt_dims_from_cfg(cfg, &d);
    if (rc != JODO_OK) return rc;
    const int D = d.D, De = d.De, T = d.T, L = d.L, nd = d.nd, ch = d.ch, r = d.r;
    const int cn = (2 * D) / L, ce = (2 * De) / L;
    Lookup lk;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].data) return jodo_set_error(JODO_ERR_ARG, "pack_weights: tensor %d has a null field", i);
        std::string nm(tensors[i].name);
        if (nm.rfind("module.", 0) == 0) nm = nm.substr(7);          // DataParallel-saved checkpoints (utils.py:23-30)
        lk.m[nm] = &tensors[i];
    }
    auto W = [&](const std::string& k, int64_t numel) { return lk.get(k, numel); };
    const std::vector<float> dummy((size_t)std::max<int64_t>(d.Mtot, (int64_t)T * 17), 0.f);
    auto S = [&](const float* p) { return p ? p : dummy.data(); };        // keep going after a miss; reported at the end

    // ---- time embedding ----
    P.put(S(W("time_mlp.0.weights", 8)), 8);
    P.put(S(W("time_mlp.1.weight", (int64_t)T * 17)), (size_t)T * 17);
    P.put(S(W("time_mlp.1.bias", T)), T);
    if (const float* w = W("time_mlp.3.weight", (int64_t)T * T)) P.put_proj(w, T, nat_in(T), nat_out(T)); else P.put_zero1();
    P.put(S(W("time_mlp.3.bias", T)), T);
    if (d.cond_ch > 0) {
        P.put(S(W("cond_mlp.0.weight", D)), D);
        P.put(S(W("cond_mlp.0.bias", D)), D);
        if (const float* w = W("cond_mlp.2.weight", (int64_t)D * D)) P.put_proj(w, D, nat_in(D), nat_out(D)); else P.put_zero1();
        P.put(S(W("cond_mlp.2.bias", D)), D);
        const int kc = d.cond_ch * D;
        if (kc % 64) return jodo_set_error(JODO_ERR_UNSUPPORTED, "cond_ch * nf = %d is not a multiple of 64", kc);
        if (const float* w = W("cond_lin.weight", (int64_t)T * kc)) P.put_proj(w, kc, nat_in(kc), nat_out(T)); else P.put_zero1();
        P.put(S(W("cond_lin.bias", T)), T);
    } else {
        for (int i = 0; i < 6; ++i) P.put_zero1();
    }
    // ---- all time-modulation linears fused into one [Mtot, T] projection ----
    {
        std::vector<float> Wm((size_t)d.Mtot * T, 0.f), bm((size_t)d.Mtot, 0.f);
        auto rows = [&](int64_t o, const std::string& base, int n) {
            const float* w = W(base + ".weight", (int64_t)n * T);
            const float* b = W(base + ".bias", n);
            if (w) std::memcpy(&Wm[(size_t)o * T], w, sizeof(float) * (size_t)n * T);
            if (b) std::memcpy(&bm[(size_t)o], b, sizeof(float) * (size_t)n);
        };
        rows(0, "dist_layer.time_mlp.1", 2);
        for (int l = 0; l < L; ++l) {
            const std::string b = "e_block_" + std::to_string(l);
            int64_t o = 32 + (int64_t)l * d.MB;
            rows(o, b + ".node_time_mlp.1", 6 * D); o += 6 * D;
            rows(o, b + ".edge_time_mlp.1", 6 * De); o += 6 * De;
            rows(o, b + ".equi_update.time_mlp.1", 2 * D); o += 2 * D;
            rows(o, b + ".dist_layer.time_mlp.1", 2); o += 32;
            // coord_mlp.0 pushed through the LayerNorm of equi_update (pair update kernel): W0 (1 + scale) and W0 shift + b0
            // are affine in SiLU(time_emb) like every other modulation output; composed in double, k ascending
            const float* W0 = W(b + ".equi_update.coord_mlp.0.weight", (int64_t)D * D);
            const float* b0 = W(b + ".equi_update.coord_mlp.0.bias", D);
            const float* Wt = W(b + ".equi_update.time_mlp.1.weight", (int64_t)2 * D * T);      // rows: shift [D] | scale [D]
            const float* bt = W(b + ".equi_update.time_mlp.1.bias", 2 * D);
            if (W0 && b0 && Wt && bt) {
                std::vector<double> acc((size_t)T);
                for (int part = 0; part < 2; ++part) {               // 0: scale rows -> W0 (1 + sc); 1: shift rows -> W0 sh + b0
                    const float* Wsrc = Wt + (size_t)(part == 0 ? D : 0) * T;
                    const float* bsrc = bt + (part == 0 ? D : 0);
                    for (int i = 0; i < D; ++i) {
                        std::fill(acc.begin(), acc.end(), 0.0);
                        double bacc = 0.0;
                        for (int k = 0; k < D; ++k) {
                            const double wk = W0[(size_t)i * D + k];
                            const float* row = Wsrc + (size_t)k * T;
                            for (int t = 0; t < T; ++t) acc[t] += wk * (double)row[t];
                            bacc += wk * ((part == 0 ? 1.0 : 0.0) + (double)bsrc[k]);
                        }
                        float* dst = &Wm[(size_t)(o + (int64_t)part * D + i) * T];
                        for (int t = 0; t < T; ++t) dst[t] = (float)acc[t];
                        bm[(size_t)(o + (int64_t)part * D + i)] = (float)(bacc + (part == 1 ? (double)b0[i] : 0.0));
                    }
                }
            }
        }
        P.put_proj(Wm.data(), T, nat_in(T), nat_out((int)d.Mtot));
        P.put(bm);
    }
    // ---- embeddings ----
    if (const float* w = W("node_emb.weight", (int64_t)D * 2 * nd)) P.put_proj(w, 2 * nd, small_in(2 * nd), nat_out(D)); else P.put_zero1();
    P.put(S(W("node_emb.bias", D)), D);
    if (const float* w = W("edge_emb.weight", (int64_t)De * (2 * ch + De)))
        P.put_proj(w, 2 * ch + De, cat(nat_in(De, 2 * ch), small_in(2 * ch)), nat_out(De));      // [G0 ; raw edge inputs]
    else P.put_zero1();
    P.put(S(W("edge_emb.bias", De)), De);
    P.put(gbf_table(lk, "dist_layer", De));
    // ---- heads ----
    const int catn = D + L * cn, cate = De + L * ce, h2 = De / 2;
    if (const float* w = W("node_pred_mlp.0.weight", (int64_t)D * catn)) P.put_proj(w, catn, hid_in(D, L, cn, d.cnp), nat_out(D)); else P.put_zero1();
    P.put(S(W("node_pred_mlp.0.bias", D)), D);
    if (const float* w = W("node_pred_mlp.2.weight", (int64_t)(D / 2) * D)) P.put_proj(w, D, nat_in(D), nat_out(D / 2)); else P.put_zero1();
    P.put(S(W("node_pred_mlp.2.bias", D / 2)), D / 2);
    {
        const OMap om = nat_out(32, nd);
        if (const float* w = W("node_pred_mlp.4.weight", (int64_t)nd * (D / 2))) P.put_proj(w, D / 2, nat_in(D / 2), om); else P.put_zero1();
        P.put_vec(S(W("node_pred_mlp.4.bias", nd)), om);
    }
    {   // exist | type heads side by side
        const float* w1a = W("edge_exist_mlp.0.weight", (int64_t)De * cate), *w1b = W("edge_type_mlp.0.weight", (int64_t)De * cate);
        std::vector<float> W1((size_t)2 * De * cate, 0.f), b1((size_t)2 * De, 0.f);
        if (w1a) std::memcpy(W1.data(), w1a, sizeof(float) * (size_t)De * cate);
        if (w1b) std::memcpy(W1.data() + (size_t)De * cate, w1b, sizeof(float) * (size_t)De * cate);
        P.put_proj(W1.data(), cate, hid_in(De, L, ce, d.cep), nat_out(2 * De));
        if (const float* b = W("edge_exist_mlp.0.bias", De)) std::memcpy(b1.data(), b, sizeof(float) * De);
        if (const float* b = W("edge_type_mlp.0.bias", De)) std::memcpy(b1.data() + De, b, sizeof(float) * De);
        P.put(b1);
        std::vector<float> W2((size_t)2 * h2 * 2 * De, 0.f), b2((size_t)2 * h2, 0.f);
        const float* w2a = W("edge_exist_mlp.2.weight", (int64_t)h2 * De), *w2b = W("edge_type_mlp.2.weight", (int64_t)h2 * De);
        for (int i = 0; i < h2; ++i)
            for (int k = 0; k < De; ++k) {
                if (w2a) W2[(size_t)i * 2 * De + k] = w2a[(size_t)i * De + k];
                if (w2b) W2[(size_t)(h2 + i) * 2 * De + De + k] = w2b[(size_t)i * De + k];
            }
        P.put_proj(W2.data(), 2 * De, nat_in(2 * De), nat_out(2 * h2));
        if (const float* b = W("edge_exist_mlp.2.bias", h2)) std::memcpy(b2.data(), b, sizeof(float) * h2);
        if (const float* b = W("edge_type_mlp.2.bias", h2)) std::memcpy(b2.data() + h2, b, sizeof(float) * h2);
        P.put(b2);
        std::vector<float> W3((size_t)ch * 2 * h2, 0.f), b3((size_t)ch, 0.f);
        const float* w3a = W("edge_exist_mlp.4.weight", h2), *w3b = W("edge_type_mlp.4.weight", (int64_t)(ch - 1) * h2);
        for (int k = 0; k < h2; ++k) {
            if (w3a) W3[k] = w3a[k];
            for (int c = 1; c < ch; ++c)
                if (w3b) W3[(size_t)c * 2 * h2 + h2 + k] = w3b[(size_t)(c - 1) * h2 + k];
        }
        const OMap om3 = nat_out(32, ch);
        P.put_proj(W3.data(), 2 * h2, nat_in(2 * h2), om3);
        if (const float* b = W("edge_exist_mlp.4.bias", 1)) b3[0] = b[0];
        if (const float* b = W("edge_type_mlp.4.bias", ch - 1)) for (int c = 1; c < ch; ++c) b3[c] = b[c - 1];
        P.put_vec(b3.data(), om3);
    }
    if ((int)P.offs.size() != JW_GLOBAL_COUNT) return jodo_set_error(JODO_ERR_ARG, "pack_weights: internal slot count %d", (int)P.offs.size());
    // ---- blocks ----
    const OMap qk = d.wide ? qk_out_wide(d.SH, d.SC) : qk_out(d.SH, d.SC);
    if ((int)qk.size() != d.QKP) return jodo_set_error(JODO_ERR_ARG, "pack_weights: q/k map width %d != %d", (int)qk.size(), d.QKP);
    const int QK = d.SH * d.SC, KIN = 2 * D + 2 * De;
    for (int l = 0; l < L; ++l) {
        const std::string b = "e_block_" + std::to_string(l), a = b + ".attn_mpnn";
        auto proj = [&](const std::string& key, int64_t n_out, int64_t ld, const IMap& in, const OMap& out) {
            if (const float* w = W(key, n_out * ld)) P.put_proj(w, ld, in, out); else P.put_zero1();
        };
        proj(a + ".lin_query.weight", QK, D, nat_in(D), qk);  P.put_vec(S(W(a + ".lin_query.bias", QK)), qk);
        proj(a + ".lin_key.weight", QK, D, nat_in(D), qk);    P.put_vec(S(W(a + ".lin_key.bias", QK)), qk);
        proj(a + ".lin_value.weight", D, D, nat_in(D), nat_out(D)); P.put(S(W(a + ".lin_value.bias", D)), D);
        proj(b + ".edge_emb.weight", De, 2 * De, cat(nat_in(De), nat_in(De, De)), nat_out(De)); P.put(S(W(b + ".edge_emb.bias", De)), De);
        proj(a + ".lin_edge0.weight", QK, De, nat_in(De), qk);
        proj(a + ".lin_edge1.weight", D, De, nat_in(De), nat_out(D));
        proj(b + ".node2edge_lin.weight", De, D, nat_in(D), nat_out(De)); P.put(S(W(b + ".node2edge_lin.bias", De)), De);
        proj(b + ".ff_linear1.weight", (int64_t)r * D, D, nat_in(D), nat_out(r * D)); P.put(S(W(b + ".ff_linear1.bias", r * D)), (size_t)r * D);
        proj(b + ".ff_linear2.weight", D, (int64_t)r * D, nat_in(r * D), nat_out(D)); P.put(S(W(b + ".ff_linear2.bias", D)), D);
        proj(b + ".ff_linear3.weight", (int64_t)r * De, De, nat_in(De), nat_out(r * De)); P.put(S(W(b + ".ff_linear3.bias", r * De)), (size_t)r * De);
        proj(b + ".ff_linear4.weight", De, (int64_t)r * De, nat_in(r * De), nat_out(De)); P.put(S(W(b + ".ff_linear4.bias", De)), De);
        // input_lin [D, 2D + De + De] = h_row | h_col | e | G: the kernels feed [e ; G] per edge and apply the h halves per node
        proj(b + ".equi_update.input_lin.weight", D, KIN, cat(nat_in(De, 2 * D), nat_in(De, 2 * D + De)), nat_out(D));
        proj(b + ".equi_update.input_lin.weight", D, KIN, nat_in(D, 0), nat_out(D));
        proj(b + ".equi_update.input_lin.weight", D, KIN, nat_in(D, D), nat_out(D));
        P.put(S(W(b + ".equi_update.input_lin.bias", D)), D);
        proj(b + ".equi_update.coord_mlp.0.weight", D, D, nat_in(D), nat_out(D)); P.put(S(W(b + ".equi_update.coord_mlp.0.bias", D)), D);
        P.put(S(W(b + ".equi_update.coord_mlp.2.weight", (int64_t)3 * D)), (size_t)3 * D);
        P.put(S(W(b + ".equi_update.coord_norm.scale", 1)), 1);
        const OMap omn = nat_out(d.cnp, cn), ome = nat_out(32, ce);
        proj("node_" + std::to_string(l) + ".weight", cn, D, nat_in(D), omn); P.put_vec(S(W("node_" + std::to_string(l) + ".bias", cn)), omn);
        proj("edge_" + std::to_string(l) + ".weight", ce, De, nat_in(De), ome); P.put_vec(S(W("edge_" + std::to_string(l) + ".bias", ce)), ome);
        P.put(gbf_table(lk, b + ".dist_layer", De));
        {   // rotated LayerNorm statistics (JB_ROWQ_W .. JB_QT_W)
            const float* win = W(b + ".equi_update.input_lin.weight", (int64_t)D * KIN);
            const float* bin = W(b + ".equi_update.input_lin.bias", D);
            if (win && bin) {
                const RotStats rs = rot_stats(win, bin, D, De);
                P.put_proj(rs.rowq.data(), D, nat_in(D), nat_out(D));
                P.put_proj(rs.colq.data(), D, nat_in(D), nat_out(D));
                P.put(rs.bq);
                P.put_proj(rs.lq.data(), 2 * De, cat(nat_in(De), nat_in(De, De)), nat_out(2 * De));
                P.put_proj(rs.wec.data(), 2 * De, cat(nat_in(De), nat_in(De, De)), nat_out(D));
                P.put_proj(rs.qt.data(), D, nat_in(D), nat_out(D));
            } else {
                for (int i = 0; i < 6; ++i) P.put_zero1();
            }
        }
    }
    if (!lk.missing.empty()) return jodo_set_error(JODO_ERR_ARG, "pack_weights: parameter '%s' missing from the tensor list", lk.missing.c_str());
    if ((int)P.offs.size() != JW_GLOBAL_COUNT + L * JB_BLOCK_COUNT)
        return jodo_set_error(JODO_ERR_ARG, "pack_weights: internal slot count %d", (int)P.offs.size());
    woff = P.offs;
    return JODO_OK;
}

}  // namespace

extern "C" int jodo_dgt_packed_size(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, size_t* n_floats, int* n_woff) {
    if (!cfg || !tensors || !n_floats || !n_woff) return jodo_set_error(JODO_ERR_ARG, "packed_size: null argument");
    Packer P;
    std::vector<int64_t> woff;
    const int rc = pack(cfg, tensors, n_tensors, P, woff);
    if (rc != JODO_OK) return rc;
    *n_floats = P.blob.size();
    *n_woff = (int)woff.size();
    return JODO_OK;
}

extern "C" int jodo_dgt_pack_weights_host(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, float* packed_host,
                                          size_t cap_floats, int64_t* woff_out, int n_woff) {
    if (!cfg || !tensors || !packed_host || !woff_out) return jodo_set_error(JODO_ERR_ARG, "pack_weights: null argument");
    Packer P;
    std::vector<int64_t> woff;
    const int rc = pack(cfg, tensors, n_tensors, P, woff);
    if (rc != JODO_OK) return rc;
    if (P.blob.size() > cap_floats || (int)woff.size() != n_woff)
        return jodo_set_error(JODO_ERR_ARG, "pack_weights: buffer holds %zu floats / %d slots, need %zu / %d", cap_floats, n_woff,
                              P.blob.size(), (int)woff.size());
    std::memcpy(packed_host, P.blob.data(), sizeof(float) * P.blob.size());
    std::memcpy(woff_out, woff.data(), sizeof(int64_t) * woff.size());
    return JODO_OK;
}

extern "C" int jodo_dgt_pack_weights(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, void* packed_dev,
                                     size_t cap_floats, int64_t* woff_out, int n_woff, void* stream) {
    if (!cfg || !tensors || !packed_dev || !woff_out) return jodo_set_error(JODO_ERR_ARG, "pack_weights: null argument");
    Packer P;
    std::vector<int64_t> woff;
    const int rc = pack(cfg, tensors, n_tensors, P, woff);
    if (rc != JODO_OK) return rc;
    if (P.blob.size() > cap_floats || (int)woff.size() != n_woff)
        return jodo_set_error(JODO_ERR_ARG, "pack_weights: buffer holds %zu floats / %d slots, need %zu / %d", cap_floats, n_woff,
                              P.blob.size(), (int)woff.size());
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(packed_dev, P.blob.data(), sizeof(float) * P.blob.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);       // the staging memory belongs to this call
    if (e != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "pack_weights: %s", hipGetErrorString(e));
    std::memcpy(woff_out, woff.data(), sizeof(int64_t) * woff.size());
    return JODO_OK;
}


// ---- split-bf16 weight TAPE of the pair update (JODO_OPT_SPLIT_BF16; csrc/dgt_kernels_split.h) ----
// The opt-in split form of k_edge_update_sym (folded, rotated statistics) reads its per-block static weights — edge FFN, readout,
// the triangular factor L of the rotated statistics — as ONE contiguous run of K16 steps (3 KiB each: hi | mid | lo terms) in exactly
// the order it consumes them, so that a workgroup can stream it through an LDS ring chunk by chunk:
//   for every hidden chunk c of 64:  ff_linear3 output blocks 2c, 2c + 1 (De / 16 steps each), then ff_linear4 output blocks
//                                    0 .. De / 32 - 1, their steps 4c .. 4c + 3 (the chunk's 64 hidden features)
//   the readout edge_l (De / 16 steps)
//   L blocks b = NB2 - 1 .. 0, steps 2b .. NSL - 1 each (upper block triangle; shortest block first, as the f32 kernel walks them)
// The folded coord_mlp.0 matrix follows from the workspace (k_fold_coord writes its split image per forward).
static int split_tape_steps(const DgtDims& d) {
    const int NCH = d.r * d.De / 64, NSE = d.De / 16, NE = d.De / 32, NB2 = 2 * NE;
    int lq = 0;
    for (int b = 0; b < NB2; ++b) lq += 2 * (NB2 - b);
    return NCH * (2 * NSE + NE * 4) + NSE + lq;
}

// The NODE tape (tuned nf = 256 kernel set only; k_node_post_split, dgt_kernels_split.h), per block, in the order k_node_post consumes its
// weights, every projection K = 256 -> 16 steps per output block:
//   node2edge_lin (2 blocks) | for every hidden chunk c of 64: ff_linear1 blocks 2c, 2c + 1, then ff_linear2 blocks 0 .. 7, their steps
//   4c .. 4c + 3 | for b = 0 .. 7: Q P W_row block b, Q P W_col block b (the rotated images, dgt_pack.cpp rot_stats) | node_l readout
//   (2 blocks) | the NEXT block's lin_query, lin_key, lin_value (8 blocks each, the tuned q / k arrangement; zero for the last block)
static int split_node_tape_steps(const DgtDims& d) {
    if (d.wide || d.D != 256) return 0;
    return 2 * 16 + d.r * 4 * (2 * 16 + 8 * 4) + 16 * 16 + 2 * 16 + 24 * 16;
}

// The ATTENTION tapes (tuned nf = 256 kernel set; k_edge_attn variants 4 and 5: the two launches that share an item by heads), per block and
// CYCLIC — a launch walks the same steps for every pair offset.  Every entry is one K = De block of 4 steps:
//   first launch (12 blocks):  edge_emb: the G halves of output blocks 0, 1, then their e halves | lin_edge0 blocks 0, 1, 2 and its tail block 7
//                              (tuned q / k arrangement) | lin_edge1 blocks 0 .. 3
//   second launch (13 blocks): edge_emb as above | lin_edge0 blocks 3 .. 6 and the tail block 7 | lin_edge1 blocks 4 .. 7
static int split_attn_tape_steps(const DgtDims& d) { return (d.wide || d.D != 256) ? 0 : (12 + 13) * 4; }

static int pack_split_tape(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, std::vector<uint16_t>& tape, size_t* block_elems,
                           size_t* node_block_elems, size_t* attn_block_elems) {
    DgtDims d;
    int rc = dgt_dims_from_cfg(cfg, &d);
    if (rc != JODO_OK) return rc;
    if ((d.D != 256 && d.D != 384) || d.cond_ch != 0) return jodo_set_error(JODO_ERR_UNSUPPORTED, "split-bf16 form: built for nf = 256 / 384 unconditional models (got nf %d, cond_ch %d)", d.D, d.cond_ch);
    const int D = d.D, De = d.De, L = d.L, r = d.r, ce = (2 * De) / L, KIN = 2 * D + 2 * De;
    Lookup lk;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].data) return jodo_set_error(JODO_ERR_ARG, "pack_split: tensor %d has a null field", i);
        std::string nm(tensors[i].name);
        if (nm.rfind("module.", 0) == 0) nm = nm.substr(7);
        lk.m[nm] = &tensors[i];
    }
    const int NCH = r * De / 64, NSE = De / 16, NE = De / 32, NB2 = 2 * NE, NS4 = r * De / 16, NSL = 2 * De / 16;
    constexpr size_t STEP = 3 * 64 * 8;                 // uint16 per (block, step)
    const size_t per_block = (size_t)split_tape_steps(d) * STEP;
    *block_elems = per_block;
    tape.clear();
    tape.reserve(per_block * L);
    auto slice = [&](const std::vector<uint16_t>& p, int ns, int blk, int s0, int s1) {      // steps [s0, s1) of output block blk
        const uint16_t* src = p.data() + ((size_t)blk * ns + s0) * STEP;
        tape.insert(tape.end(), src, src + (size_t)(s1 - s0) * STEP);
    };
    std::vector<RotStats> rsv((size_t)L);              // the QR factors of every block: shared by the pair tape (L) and the node tape (Q P W_row / W_col)
    for (int l = 0; l < L; ++l) {
        const std::string b = "e_block_" + std::to_string(l);
        const float* w3 = lk.get(b + ".ff_linear3.weight", (int64_t)r * De * De);
        const float* w4 = lk.get(b + ".ff_linear4.weight", (int64_t)De * r * De);
        const float* wro = lk.get("edge_" + std::to_string(l) + ".weight", (int64_t)ce * De);
        const float* win = lk.get(b + ".equi_update.input_lin.weight", (int64_t)D * KIN);
        const float* bin = lk.get(b + ".equi_update.input_lin.bias", D);
        if (!w3 || !w4 || !wro || !win || !bin) return jodo_set_error(JODO_ERR_ARG, "pack_split: missing or mis-sized tensor '%s'", lk.missing.c_str());
        const std::vector<uint16_t> p3 = pack_proj_split(w3, De, nat_in(De), nat_out(r * De));
        const std::vector<uint16_t> p4 = pack_proj_split(w4, (int64_t)r * De, nat_in(r * De), nat_out(De));
        const std::vector<uint16_t> pro = pack_proj_split(wro, De, nat_in(De), nat_out(32, ce));
        rsv[(size_t)l] = rot_stats(win, bin, D, De);
        const RotStats& rs = rsv[(size_t)l];
        const std::vector<uint16_t> plq = pack_proj_split(rs.lq.data(), 2 * De, cat(nat_in(De), nat_in(De, De)), nat_out(2 * De));
        for (int c = 0; c < NCH; ++c) {
            slice(p3, NSE, 2 * c, 0, NSE);
            slice(p3, NSE, 2 * c + 1, 0, NSE);
            for (int ob = 0; ob < NE; ++ob) slice(p4, NS4, ob, 4 * c, 4 * c + 4);
        }
        slice(pro, NSE, 0, 0, NSE);
        for (int k = 0; k < NB2; ++k) { const int blk = NB2 - 1 - k; slice(plq, NSL, blk, 2 * blk, NSL); }
    }
    if (tape.size() != per_block * L) return jodo_set_error(JODO_ERR_ARG, "pack_split: internal tape size");
    // ---- node tapes behind the pair tapes ----
    const size_t node_block = (size_t)split_node_tape_steps(d) * STEP;
    *node_block_elems = node_block;
    if (node_block > 0) {
        const int QK = d.SH * d.SC, cn = (2 * D) / L;
        const OMap qk = qk_out(d.SH, d.SC);
        if ((int)qk.size() != 256 || d.cnp != 64) return jodo_set_error(JODO_ERR_UNSUPPORTED, "pack_split: node tape expects the tuned nf 256 layouts");
        tape.reserve(tape.size() + node_block * L);
        for (int l = 0; l < L; ++l) {
            const std::string b = "e_block_" + std::to_string(l);
            const float* wn2e = lk.get(b + ".node2edge_lin.weight", (int64_t)De * D);
            const float* w1 = lk.get(b + ".ff_linear1.weight", (int64_t)r * D * D);
            const float* w2 = lk.get(b + ".ff_linear2.weight", (int64_t)D * r * D);
            const float* wnro = lk.get("node_" + std::to_string(l) + ".weight", (int64_t)cn * D);
            const float* win = lk.get(b + ".equi_update.input_lin.weight", (int64_t)D * KIN);
            const float* bin = lk.get(b + ".equi_update.input_lin.bias", D);
            if (!wn2e || !w1 || !w2 || !wnro || !win || !bin) return jodo_set_error(JODO_ERR_ARG, "pack_split: missing or mis-sized tensor '%s'", lk.missing.c_str());
            const size_t at0 = tape.size();
            const std::vector<uint16_t> pn2e = pack_proj_split(wn2e, D, nat_in(D), nat_out(De));
            const std::vector<uint16_t> p1 = pack_proj_split(w1, D, nat_in(D), nat_out(r * D));
            const std::vector<uint16_t> p2 = pack_proj_split(w2, (int64_t)r * D, nat_in(r * D), nat_out(D));
            const RotStats& rs = rsv[(size_t)l];
            const std::vector<uint16_t> prow = pack_proj_split(rs.rowq.data(), D, nat_in(D), nat_out(D));
            const std::vector<uint16_t> pcol = pack_proj_split(rs.colq.data(), D, nat_in(D), nat_out(D));
            const std::vector<uint16_t> pnro = pack_proj_split(wnro, D, nat_in(D), nat_out(d.cnp, cn));
            for (int blk = 0; blk < 2; ++blk) slice(pn2e, 16, blk, 0, 16);
            for (int c = 0; c < r * 4; ++c) {
                slice(p1, 16, 2 * c, 0, 16);
                slice(p1, 16, 2 * c + 1, 0, 16);
                for (int ob = 0; ob < 8; ++ob) slice(p2, r * 16, ob, 4 * c, 4 * c + 4);
            }
            for (int blk = 0; blk < 8; ++blk) { slice(prow, 16, blk, 0, 16); slice(pcol, 16, blk, 0, 16); }
            for (int blk = 0; blk < 2; ++blk) slice(pnro, 16, blk, 0, 16);
            if (l + 1 < L) {
                const std::string a = "e_block_" + std::to_string(l + 1) + ".attn_mpnn";
                const float* wq = lk.get(a + ".lin_query.weight", (int64_t)QK * D);
                const float* wk = lk.get(a + ".lin_key.weight", (int64_t)QK * D);
                const float* wv = lk.get(a + ".lin_value.weight", (int64_t)D * D);
                if (!wq || !wk || !wv) return jodo_set_error(JODO_ERR_ARG, "pack_split: missing or mis-sized tensor '%s'", lk.missing.c_str());
                const std::vector<uint16_t> pq = pack_proj_split(wq, D, nat_in(D), qk), pk = pack_proj_split(wk, D, nat_in(D), qk);
                const std::vector<uint16_t> pv = pack_proj_split(wv, D, nat_in(D), nat_out(D));
                for (int blk = 0; blk < 8; ++blk) slice(pq, 16, blk, 0, 16);
                for (int blk = 0; blk < 8; ++blk) slice(pk, 16, blk, 0, 16);
                for (int blk = 0; blk < 8; ++blk) slice(pv, 16, blk, 0, 16);
            } else {
                tape.resize(tape.size() + (size_t)24 * 16 * STEP, 0);
            }
            if (tape.size() - at0 != node_block) return jodo_set_error(JODO_ERR_ARG, "pack_split: internal node tape size");
        }
    }
    // ---- attention tapes behind the node tapes ----
    const size_t attn_block = (size_t)split_attn_tape_steps(d) * STEP;
    *attn_block_elems = attn_block;
    if (attn_block > 0) {
        const int QK = d.SH * d.SC;
        const OMap qk = qk_out(d.SH, d.SC);
        for (int l = 0; l < L; ++l) {
            const std::string b = "e_block_" + std::to_string(l), a = b + ".attn_mpnn";
            const float* wee = lk.get(b + ".edge_emb.weight", (int64_t)De * 2 * De);
            const float* wl0 = lk.get(a + ".lin_edge0.weight", (int64_t)QK * De);
            const float* wl1 = lk.get(a + ".lin_edge1.weight", (int64_t)D * De);
            if (!wee || !wl0 || !wl1) return jodo_set_error(JODO_ERR_ARG, "pack_split: missing or mis-sized tensor '%s'", lk.missing.c_str());
            const size_t at0 = tape.size();
            const std::vector<uint16_t> pee = pack_proj_split(wee, 2 * De, cat(nat_in(De), nat_in(De, De)), nat_out(De));
            const std::vector<uint16_t> p0 = pack_proj_split(wl0, De, nat_in(De), qk);
            const std::vector<uint16_t> p1 = pack_proj_split(wl1, De, nat_in(De), nat_out(D));
            for (int pass = 0; pass < 2; ++pass) {
                for (int blk = 0; blk < 2; ++blk) slice(pee, 8, blk, 0, 4);      // the G halves of both output blocks first, then the e halves:
                for (int blk = 0; blk < 2; ++blk) slice(pee, 8, blk, 4, 8);      // only one operand's split image is live at a time
                for (int blk = pass ? 3 : 0; blk < (pass ? 7 : 3); ++blk) slice(p0, 4, blk, 0, 4);
                slice(p0, 4, 7, 0, 4);
                for (int blk = pass * 4; blk < pass * 4 + 4; ++blk) slice(p1, 4, blk, 0, 4);
            }
            if (tape.size() - at0 != attn_block) return jodo_set_error(JODO_ERR_ARG, "pack_split: internal attention tape size");
        }
    }
    return JODO_OK;
}

// debug / gate experiments (csrc/dgt_split.hip): W [n_out, n_in] row-major, natural maps -> the f32 packing (n_out * n_in floats) and
// the split packing (n_out * n_in * 3 bf16) of the same projection, both into host buffers.
extern "C" int jodo_debug_pack_split(const float* W, int n_out, int n_in, float* f32_packed_host, void* split_packed_host) {
    if (!W || n_out <= 0 || n_in <= 0 || n_out % 32 || n_in % 32) return jodo_set_error(JODO_ERR_ARG, "debug_pack_split: sizes must be multiples of 32");
    if (f32_packed_host) {
        Packer P;
        P.put_proj(W, n_in, nat_in(n_in), nat_out(n_out));
        std::memcpy(f32_packed_host, P.blob.data(), sizeof(float) * (size_t)n_out * n_in);
    }
    if (split_packed_host) {
        const std::vector<uint16_t> s = pack_proj_split(W, n_in, nat_in(n_in), nat_out(n_out));
        std::memcpy(split_packed_host, s.data(), sizeof(uint16_t) * s.size());
    }
    return JODO_OK;
}

// The static weight tape of the opt-in split-bf16 pair update: sizes, then the tape itself into a host buffer (the caller uploads it
// and hands the device copy to jodo_plan_set_split_weights).
extern "C" int jodo_dgt_split_size(const jodo_cfg* cfg, size_t* total_bytes, size_t* pair_block_bytes, size_t* node_block_bytes, size_t* attn_block_bytes) {
    if (!cfg || !total_bytes || !pair_block_bytes || !node_block_bytes || !attn_block_bytes) return jodo_set_error(JODO_ERR_ARG, "split_size: null argument");
    DgtDims d;
    const int rc = dgt_dims_from_cfg(cfg, &d);
    if (rc != JODO_OK) return rc;
    if ((d.D != 256 && d.D != 384) || d.cond_ch != 0) return jodo_set_error(JODO_ERR_UNSUPPORTED, "split-bf16 form: built for nf = 256 / 384 unconditional models (got nf %d, cond_ch %d)", d.D, d.cond_ch);
    *pair_block_bytes = (size_t)split_tape_steps(d) * 3072;
    *node_block_bytes = (size_t)split_node_tape_steps(d) * 3072;
    *attn_block_bytes = (size_t)split_attn_tape_steps(d) * 3072;
    *total_bytes = (*pair_block_bytes + *node_block_bytes + *attn_block_bytes) * d.L;
    return JODO_OK;
}
extern "C" int jodo_dgt_pack_split_host(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, void* host, size_t cap_bytes) {
    if (!cfg || !tensors || !host) return jodo_set_error(JODO_ERR_ARG, "pack_split: null argument");
    std::vector<uint16_t> tape;
    size_t per_block = 0, node_block = 0, attn_block = 0;
    const int rc = pack_split_tape(cfg, tensors, n_tensors, tape, &per_block, &node_block, &attn_block);
    if (rc != JODO_OK) return rc;
    if (tape.size() * sizeof(uint16_t) > cap_bytes) return jodo_set_error(JODO_ERR_ARG, "pack_split: buffer of %zu bytes, need %zu", cap_bytes, tape.size() * sizeof(uint16_t));
    std::memcpy(host, tape.data(), tape.size() * sizeof(uint16_t));
    return JODO_OK;
}
