// Training step of the DGT behind the C ABI: jodo_train_forward evaluates the score network with every
// activation the backward needs kept in the caller's workspace (dropout active when p > 0, as under model.train()), and
// jodo_train_backward returns d loss / d parameter for ALL parameters of the state_dict given d loss / d outputs — what
// loss.backward() computes in /root/reference/losses.py:286-385 through
//   DGT_concat.forward / Cond_DGT_concat.forward   models/mol_gnn.py:491-594, :687-794
//   EquivariantMixBlock.forward                    models/mol_gnn.py:270-322
//   TransMixLayer                                  models/layers.py:131-186
//   MultiCondEquiUpdate                            models/mol_gnn.py:71-94
// Parameters stay in their PyTorch [out, in] layouts (no packing: the optimiser updates them every step); projections and both of
// their gradient products are jt::gemm (train_gemm.hip), everything else train_ops.h.  The forward here is the dense per-molecule
// formulation of SURVEY.md §3.2b on directed n x n tiles; the phases of the backward mirror it in reverse.
// Gradient buffers are fully written (zeroed, then accumulated in a fixed launch order: bit-deterministic).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <unordered_map>
#include <vector>

#include "../../include/jodo_hip.h"
#include "jodo_hip_internal.h"
#include "train_gemm.h"
#include "train_ops.h"
#include "train_fused.h"

using namespace jt;

namespace {

struct Lin { int w = -1, b = -1; };
struct BlkIx {
    Lin edge_emb, n2e, key, query, value, ff1, ff2, ff3, ff4, eq_time, eq_in, eq_c0, node_time, edge_time, gbf_time;
    int le0 = -1, le1 = -1, eq_c2 = -1, eq_scale = -1, gbf_means = -1, gbf_stds = -1;
    Lin node_ro, edge_ro;
};

struct Arena {
    char* base; size_t off;
    float* f(size_t n) { const size_t o = off; off += (n * 4 + 255) / 256 * 256; return base ? reinterpret_cast<float*>(base + o) : nullptr; }
    int* i(size_t n) { return reinterpret_cast<int*>(f(n)); }
};

struct BlkBuf {
    float *nmod, *emod, *qmod, *gm, *d2, *G, *xh_e1, *rs_e1, *et, *xh_h, *rs_h, *ht, *q, *k, *v, *t0, *t1, *alpha, *hhat, *n2e;
    float* qkv;                       // q | k | v side by side, [Nn, 2 QK + D]: what the merged projection writes and the wave-per-atom attention reads
    float *xh_hn, *rs_hn, *hn, *f1, *a1, *f2, *xh_en, *rs_en, *en, *f3, *a3, *f4, *xh_pre, *rs_pre, *u, *c0pre, *c0a, *inv;
};
struct Bufs {
    int* flags;
    float *feat, *t1pre, *t1a, *temb, *tau, *ctx, *cc0pre, *cc0a, *cc2, *gm_top;
    float *cpos, *nin, *ein, *adj2d, *adjsp, *d2c, *ah, *eh;
    std::vector<float*> h, e, pos;
    std::vector<BlkBuf> blk;
    float *nh1pre, *nh1, *nh2pre, *nh2, *atom, *x1pre[2], *x1[2], *x2pre[2], *x2[2], *Ep, *posf;
    // scratch shared by all phases
    float *tE_D[3], *tE_De[4], *tE_QK, *tE_rD, *tE_H, *tN_D[4], *tN_QK[2], *tN_rD, *tN_De, *tRow[3], *tE3[3], *tN3[4], *tcatn, *tcate;
    float *dtau, *dtemb, *tB_T[2], *tB_cD[2], *part, *part2, *rowpart, *splitk;
    float* fpack;                     // packed MFMA operands of the fused chains (forward + transposed images), one slice per block (train_fused.h)
    float* tE_De2[3];                 // scratch of the fused backward chains: df4 | den | de1
    float *Wall, *ball, *mods_all, *dmods_all, *dWall, *dball;      // batched modulation projections (train_ops.h ModTable)
    float *Wqkv, *bqkv, *dqkv, *dWqkv, *dbqkv;                // lin_query / lin_key / lin_value of a block as one product: [L][2 QK + D, D] gathered weights
    float *dwg;                                                     // split-K partial tiles of a grouped weight-gradient launch (gemm_dw_group)
    float *tN_D2[2];                                                // d h_row | d h_col of input_lin while their weight-gradient products are queued
    size_t fpack_block;
    size_t splitk_floats, part_floats, part2_floats, dwg_floats;
    std::vector<GemmJob> dwq;                                       // weight-gradient products waiting for the next Ctx::flush_dw
    float* finpart;                                                 // partial sums of the reductions whose later stages are queued (Ctx::flush_fin)
    size_t finpart_floats, fin_used;
    std::vector<FinJob> finq;
    bool defer_fin;                                                 // the backward queues; the forward (and everything outside it) launches at once
};

}  // namespace

struct jodo_train {
    jodo_cfg cfg;
    int B, N, Nn, R;
    int D, De, T, L, H, XH, SC, QK, C, r, nd, ch, cc, cn, ce, catn, cate, half;
    std::vector<int> tables;          // node_off | edge_off | nn | node_mol | edge_mol | edge_a | edge_c | ec_off | ec_mol_off
    size_t o_node_off, o_edge_off, o_nn, o_node_mol, o_edge_mol, o_edge_a, o_edge_c, o_ec_off, o_ec_mol_off;
    int NC;                           // chunks of edge rows (two-level per-molecule sums)
    int n_params;
    std::vector<size_t> numel;
    // parameter indices
    Lin node_emb, edge_emb, gbf_time, np0, np2, np4, et0, et2, et4, ee0, ee2, ee4, time1, time3, cond0, cond2, cond_lin;
    int gbf_means, gbf_stds, time_w;
    std::vector<BlkIx> blk;
    size_t ws_bytes;
    int fused;                        // 1: the three per-edge chains of a block run as fused strip kernels (train_fused.hip)
    int fused_bwd;                    // 1: their input-gradient sides too (the weight-gradient products stay GEMMs)
    int gbf_chunk;                    // option 5: the Gaussian layer's backward as one pass over 32-row chunks: 0 never, 1 (default) batches of >= 4 096 chunks, 2 always (tests)
    int fused_attn;                   // option 4: 1 = attention forward / backward as wave-per-atom kernels (train_fused.hip; bit-identical to the op-by-op ones)
    int group_dw;                     // option 3: 1 = the backward's weight-gradient products are queued and launched in groups (gemm_dw_group)
    int save_activations;             // option 2: 0 = the following forwards are not followed by a backward (no-grad self-conditioning call)
    int Mtot;                         // modulation floats per molecule: 2 (top-level GBF) + L (6 D + 6 De + 2 D + 2)
};

namespace {

void layout(const jodo_train& t, Arena& a, Bufs& b) {
    const size_t B = t.B, Nn = t.Nn, R = t.R, D = t.D, De = t.De, T = t.T, L = t.L, QK = t.QK, H = t.H, r = t.r, nd = t.nd, ch = t.ch;
    const size_t cc = t.cc > 0 ? t.cc : 1;
    b.flags = a.i(8);
    b.feat = a.f(B * (2 * t.half + 1)); b.t1pre = a.f(B * T); b.t1a = a.f(B * T); b.temb = a.f(B * T); b.tau = a.f(B * T);
    b.ctx = a.f(B * cc); b.cc0pre = a.f(B * cc * D); b.cc0a = a.f(B * cc * D); b.cc2 = a.f(B * cc * D); b.gm_top = a.f(B * 2);
    b.cpos = a.f(Nn * 3); b.nin = a.f(Nn * 2 * nd); b.ein = a.f(R * (2 * ch + De)); b.adj2d = a.f(R); b.adjsp = a.f(R); b.d2c = a.f(R);
    b.ah = a.f(Nn * t.catn); b.eh = a.f(R * t.cate);
    b.h.resize(L + 1); b.e.resize(L + 1); b.pos.resize(L + 1);
    for (size_t l = 0; l <= L; ++l) { b.h[l] = a.f(Nn * D); b.e[l] = a.f(R * De); b.pos[l] = a.f(Nn * 3); }
    b.blk.resize(L);
    for (size_t l = 0; l < L; ++l) {
        BlkBuf& k = b.blk[l];
        k.nmod = a.f(B * 6 * D); k.emod = a.f(B * 6 * De); k.qmod = a.f(B * 2 * D); k.gm = a.f(B * 2);
        k.d2 = a.f(R); k.G = a.f(R * De); k.xh_e1 = a.f(R * De); k.rs_e1 = a.f(R); k.et = a.f(R * De);
        k.xh_h = a.f(Nn * D); k.rs_h = a.f(Nn); k.ht = a.f(Nn * D); k.q = a.f(Nn * QK); k.k = a.f(Nn * QK); k.v = a.f(Nn * D);
        k.qkv = a.f(Nn * (2 * QK + D));
        k.t0 = a.f(R * QK); k.t1 = a.f(R * D); k.alpha = a.f(R * H); k.hhat = a.f(Nn * D); k.n2e = a.f(Nn * De);
        k.xh_hn = a.f(Nn * D); k.rs_hn = a.f(Nn); k.hn = a.f(Nn * D); k.f1 = a.f(Nn * r * D); k.a1 = a.f(Nn * r * D); k.f2 = a.f(Nn * D);
        k.xh_en = a.f(R * De); k.rs_en = a.f(R); k.en = a.f(R * De); k.f3 = a.f(R * r * De); k.a3 = a.f(R * r * De); k.f4 = a.f(R * De);
        k.xh_pre = a.f(R * D); k.rs_pre = a.f(R); k.u = a.f(R * D); k.c0pre = a.f(R * D); k.c0a = a.f(R * D); k.inv = a.f(R * 3);
    }
    b.nh1pre = a.f(Nn * D); b.nh1 = a.f(Nn * D); b.nh2pre = a.f(Nn * D / 2); b.nh2 = a.f(Nn * D / 2); b.atom = a.f(Nn * nd);
    for (int s = 0; s < 2; ++s) { b.x1pre[s] = a.f(R * De); b.x1[s] = a.f(R * De); b.x2pre[s] = a.f(R * De / 2); b.x2[s] = a.f(R * De / 2); }
    b.Ep = a.f(R * ch); b.posf = a.f(Nn * 3);
    for (int s = 0; s < 3; ++s) b.tE_D[s] = a.f(R * D);
    for (int s = 0; s < 4; ++s) b.tE_De[s] = a.f(R * De);
    b.tE_QK = a.f(R * QK); b.tE_rD = a.f(R * r * De); b.tE_H = a.f(R * H);
    for (int s = 0; s < 4; ++s) b.tN_D[s] = a.f(Nn * D);
    for (int s = 0; s < 2; ++s) b.tN_QK[s] = a.f(Nn * QK);
    b.tN_rD = a.f(Nn * r * D); b.tN_De = a.f(Nn * De);
    for (int s = 0; s < 3; ++s) b.tRow[s] = a.f(R > Nn ? R : Nn);
    for (int s = 0; s < 3; ++s) b.tE3[s] = a.f(R * 3);
    for (int s = 0; s < 4; ++s) b.tN3[s] = a.f(Nn * 3);
    b.tcatn = a.f(Nn * t.catn); b.tcate = a.f(R * t.cate);
    b.dtau = a.f(B * T); b.dtemb = a.f(B * T);       // (modulation gradients go straight into dmods_all)
    for (int s = 0; s < 2; ++s) { b.tB_T[s] = a.f(B * T); b.tB_cD[s] = a.f(B * cc * D); }
    const size_t rows = R > Nn ? R : Nn;
    const size_t maxF = std::max<size_t>({(size_t)6 * D, T, r * D});
    b.part_floats = ((rows + 31) / 32 + B + 1) * maxF;        // first-level partial sums: 32-row chunks x the widest reduced array
    b.part = a.f(b.part_floats);
    b.part2_floats = (b.part_floats / maxF / 32 + 2) * maxF;
    b.part2 = a.f(b.part2_floats);
    b.rowpart = a.f(rows * 16);                               // eight (sum, sum of squares) pairs per row
    b.splitk_floats = (size_t)32 << 20;      // 128 MiB of split-K partial tiles at most
    const size_t need = ((rows + 1023) / 1024 + 1) * (size_t)D * (2 * D + 2 * De);
    if (b.splitk_floats > need) b.splitk_floats = need;
    b.splitk = a.f(b.splitk_floats);
    b.dwg_floats = 6 * b.splitk_floats;      // the partial tiles of about a dozen queued products (a group is cut where they do not fit)
    b.dwg = a.f(b.dwg_floats);
    for (int s = 0; s < 2; ++s) b.tN_D2[s] = a.f(Nn * D);
    b.finpart_floats = (size_t)4 << 20;
    {   // one block's queued reductions: three two-sum passes (D, De, De), four plain ones (De, De, 1, 1) over NC chunks, column partials
        const size_t nc = (size_t)t.NC + 1, colp = (rows + 31) / 32 + 64;
        const size_t need = nc * (2 * D + 4 * De + 2 * De + 2) + colp * (De + 1 + 2 * De) + 65536;
        if (b.finpart_floats < need) b.finpart_floats = need;
    }
    b.finpart = a.f(b.finpart_floats); b.fin_used = 0; b.defer_fin = false;
    const FusedDims fd{t.D, t.De, t.r, t.QK, t.ce, t.L};
    b.fpack_block = fused_pack_layout(fd).total_bwd;
    b.fpack = a.f(b.fpack_block * L);
    for (int s = 0; s < 3; ++s) b.tE_De2[s] = a.f(R * De);
    const size_t Mt = (size_t)t.Mtot;
    b.Wall = a.f(Mt * T); b.ball = a.f(Mt); b.mods_all = a.f(B * Mt); b.dmods_all = a.f(B * Mt); b.dWall = a.f(Mt * T); b.dball = a.f(Mt);
    const size_t F3 = 2 * QK + D;
    b.Wqkv = a.f(L * F3 * D); b.bqkv = a.f(L * F3); b.dqkv = a.f(Nn * F3); b.dWqkv = a.f(L * F3 * D); b.dbqkv = a.f(L * F3);
}

struct Ctx {
    const jodo_train& t; Topo tp; const float* const* P; float* const* G; Bufs& b; hipStream_t s;
    const float* p(int i) const { return P[i]; }
    float* g(int i) const { return G[i]; }
    // Y[rows, N] (ldy) (+)= X[rows, K] (ldx) W[N, K]^T (ldw) + bias
    void lin(const float* X, int ldx, int rows, int K, const float* W, int ldw, int N, const float* bias, float* Y, int ldy, int acc) const {
        gemm(s, 0, 1, rows, N, K, X, ldx, W, ldw, Y, ldy, bias, acc, b.splitk, b.splitk_floats);
    }
    // Y = tanh(X W^T + bias)
    void lin_tanh(const float* X, int ldx, int rows, int K, const float* W, int ldw, int N, const float* bias, float* Y) const {
        GemmEpi e; e.act = 1; e.out2 = nullptr; e.drop = drop(0.f, 0, 0, 0); e.dbias = nullptr;
        gemm(s, 0, 1, rows, N, K, X, ldx, W, ldw, Y, N, bias, 0, b.splitk, b.splitk_floats, &e);
    }
    // pre = X W^T + bias (kept for the backward), act = SiLU(pre) * dropout; both dense [rows, N]
    void lin_silu(const float* X, int ldx, int rows, int K, const float* W, int ldw, int N, const float* bias, float* pre, float* act, Drop d) const {
        GemmEpi e; e.act = 2; e.out2 = act; e.drop = d; e.dbias = nullptr;
        gemm(s, 0, 1, rows, N, K, X, ldx, W, ldw, pre, N, bias, 0, b.splitk, b.splitk_floats, &e);
    }
    // dX[rows, K] (ldx) (+)= dY[rows, N] (ldy) W[N, K] (ldw)
    void lin_dx(const float* dY, int ldy, int rows, int N, const float* W, int ldw, int K, float* dX, int ldx, int acc) const {
        gemm(s, 0, 0, rows, K, N, dY, ldy, W, ldw, dX, ldx, nullptr, acc, b.splitk, b.splitk_floats);
    }
    // dW[N, K] (lddw) += dY[rows, N]^T X[rows, K]
    // db (optional): the bias gradient, db[N] += column sums of dY, computed by the same launch
    // With option 3 (default) the product is QUEUED: it runs, with every other product queued since, in the one launch of the next
    // flush_dw() — which the caller places before anything overwrites an operand (dY, X) of a queued product.
    void lin_dw(const float* dY, int ldy, int rows, int N, const float* X, int ldx, int K, float* dW, int lddw, float* db = nullptr) const {
        if (t.group_dw && t.fused_bwd) {                     // (the op-by-op backward reuses operands in place: launched at once)
            b.dwq.push_back(GemmJob{N, K, rows, dY, ldy, X, ldx, dW, lddw, db});
            return;
        }
        GemmEpi e; e.act = 0; e.out2 = nullptr; e.drop = drop(0.f, 0, 0, 0); e.dbias = db;
        gemm(s, 1, 0, N, K, rows, dY, ldy, X, ldx, dW, lddw, nullptr, 1, b.splitk, b.splitk_floats, &e);
    }
    void flush_dw() const {
        if (b.dwq.empty()) return;
        gemm_dw_group(s, b.dwq.data(), (int)b.dwq.size(), b.dwg, b.dwg_floats, b.splitk_floats);
        b.dwq.clear();
    }
    // out[F] += column sums of a[rows, F] (row stride lda), optionally of a * bb: 32-row partial sums, then partial sums of those
    // until at most 64 rows are left (every level a launch with rows x F / 32 threads; fixed order, no atomics)
    void colsum(const float* a, int lda, const float* bb, int ldb, long rows, int F, float* out) const {
        const int chunk = 32;
        if (b.defer_fin) {
            // first level now (it reads the caller's array), the rest queued: one more level down to at most 64 rows, then the final sum
            long n = (rows + chunk - 1) / chunk;
            const long c2 = n <= 2048 ? 32 : (n + 63) / 64, n2 = n > 64 ? (n + c2 - 1) / c2 : 0;
            // both levels' partial sums in ONE allocation: a second fin_alloc could flush the queue and hand out the first region again
            const size_t n1f = ((size_t)n * F + 63) / 64 * 64;
            float* p1 = fin_alloc(n1f + (size_t)n2 * F);
            JT_LAUNCH(k_colsum_part, n * F, s, rows, F, chunk, a, lda, bb, ldb, p1);
            if (n > 64) {
                float* p2 = p1 + n1f;
                push_fin(FinJob{p1, p2, nullptr, FIN_COLPART, F, (int)n, 0, 0, (int)c2, 0, 0});
                p1 = p2; n = n2;
            }
            push_fin(FinJob{p1, out, nullptr, FIN_COL, F, (int)n, 0, 0, 0, 1, 0});
            return;
        }
        float* dst = b.part; float* other = b.part2;
        long n = rows;
        while (true) {
            const long nch = (n + chunk - 1) / chunk;
            JT_LAUNCH(k_colsum_part, nch * F, s, n, F, chunk, a, lda, bb, ldb, dst);
            a = dst; lda = F; bb = nullptr; ldb = 0; n = nch;
            if (n <= 64) break;
            std::swap(dst, other);
        }
        JT_LAUNCH(k_colsum_fin, F, s, n, F, a, out, 1);
    }
    // per-molecule sums over EDGE rows (two levels over the plan's chunks), out[mol, ocol + f] written
    void seg_edge(int F, const float* a, const float* bb, float* out, int ldo, int ocol) const {
        float* part = b.defer_fin ? fin_alloc((size_t)tp.NC * F) : b.part;
        JT_LAUNCH(k_seg_part, (long)tp.NC * F, s, tp.NC, F, tp.ec_off, a, bb, part);
        if (b.defer_fin) push_fin(FinJob{part, out, tp.ec_mol_off, FIN_SEG, F, t.B, ldo, ocol, 0, 0, 0});
        else JT_LAUNCH(k_seg_fin, (long)t.B * F, s, t.B, F, tp.ec_mol_off, (const float*)part, out, ldo, ocol, 0);
    }
    // out[mol, ocol + f] = sum over the molecule's edge rows of a[(a, c), f] (p[a, f] + p[c, f] + bias[f])
    void seg_edge_ehat(int F, const float* a, const float* p, const float* bias, float* out, int ldo, int ocol) const {
        float* part = b.defer_fin ? fin_alloc((size_t)tp.NC * F) : b.part;
        JT_LAUNCH(k_seg_part_ehat, (long)tp.NC * F, s, tp, F, a, p, bias, part);
        if (b.defer_fin) push_fin(FinJob{part, out, tp.ec_mol_off, FIN_SEG, F, t.B, ldo, ocol, 0, 0, 0});
        else JT_LAUNCH(k_seg_fin, (long)t.B * F, s, t.B, F, tp.ec_mol_off, (const float*)part, out, ldo, ocol, 0);
    }
    // the two sums of a LayerNorm + modulate backward over edge rows: out[mol, c1 + f] = sum a, out[mol, c2 + f] = sum a bb
    void seg2_edge(int F, const float* a, const float* bb, float* out, int ldo, int c1, int c2) const {
        float* part = b.defer_fin ? fin_alloc((size_t)tp.NC * 2 * F) : b.part;
        JT_LAUNCH(k_seg_part2, (long)tp.NC * F, s, tp.NC, F, tp.ec_off, a, bb, part);
        if (b.defer_fin) push_fin(FinJob{part, out, tp.ec_mol_off, FIN_SEG2, F, t.B, ldo, c1, c2, 0, 0});
        else JT_LAUNCH(k_seg_fin2, (long)t.B * 2 * F, s, t.B, F, tp.ec_mol_off, (const float*)part, out, ldo, c1, c2);
    }
    // queued later stages of reductions (train_ops.h k_fin_group): a region of partial sums per queued job, all of them in one launch
    // (second-level column partials first, in a launch of their own) at flush_fin() — which the backward calls once per block
    float* fin_alloc(size_t n) const {
        n = (n + 63) / 64 * 64;
        if (b.fin_used + n > b.finpart_floats) flush_fin();
        if (n > b.finpart_floats) { fprintf(stderr, "jodo train: %zu floats of reduction scratch asked, %zu there\n", n, b.finpart_floats); abort(); }
        float* p = b.finpart + b.fin_used;
        b.fin_used += n;
        return p;
    }
    void push_fin(const FinJob& j) const { b.finq.push_back(j); }
    void flush_fin() const {
        for (int phase = 0; phase < 2; ++phase) {
            FinTable T; T.n = 0;
            int blocks = 0;
            auto launch = [&]() {
                if (T.n) hipLaunchKernelGGL(k_fin_group, dim3((unsigned)blocks), dim3(256), 0, s, T);
                T.n = 0; blocks = 0;
            };
            for (const FinJob& q : b.finq) {
                if ((q.kind == FIN_COLPART) != (phase == 0)) continue;
                if (T.n == FIN_MAX) launch();
                FinJob j = q;
                j.blk0 = blocks;
                blocks += (int)((fin_threads(j) + 255) / 256);
                T.j[T.n++] = j;
            }
            launch();
        }
        b.finq.clear();
        b.fin_used = 0;
    }
    void silu(long n, const float* x, float* y, Drop d) const { JT_LAUNCH(k_silu_fwd, n, s, n, x, y, d); }
    void silu_bwd(long n, const float* x, const float* dy, float* dx, Drop d) const { JT_LAUNCH(k_silu_bwd, n, s, n, x, dy, dx, d); }
    void stats(long rows, int F, const float* x, float* mean, float* rstd) const {
        JT_LAUNCH(k_row_part, rows * 8, s, rows, F, x, b.rowpart);
        JT_LAUNCH(k_row_stats, rows, s, rows, F, x, (const float*)b.rowpart, mean, rstd);
    }
    void ln_mod(long rows, int F, const float* x, const float* mean, const float* rstd, const int* row_mol, const float* mods, int ldm, int sh, int sc,
                float* xhat, float* y) const {
        JT_LAUNCH(k_ln_mod_fwd, rows * F, s, rows, F, x, mean, rstd, row_mol, mods, ldm, sh, sc, xhat, y);
    }
    // LayerNorm + modulate backward: modulation gradients into dmods[:, sh], [:, sc] (written), dx (acc)
    // (dmods: row stride ldd — the block's columns of the [B, Mtot] array of all modulation gradients)
    void ln_mod_bwd(long rows, int F, const float* dy, const float* xhat, const float* rstd, const int* row_mol, const int* seg_off, const float* mods,
                    int ldm, int sh, int sc, float* dmods, int ldd, float* dx, int acc) const {
        // d shift = sum dy, d scale = sum dy xhat per molecule: one pass (edge rows: two levels over the plan's chunks)
        if (seg_off == tp.edge_off) seg2_edge(F, dy, xhat, dmods, ldd, sh, sc);
        else JT_LAUNCH(k_seg_colsum2, (long)t.B * F, s, t.B, F, seg_off, dy, xhat, dmods, ldd, sh, sc);
        if (t.fused_bwd && seg_off != tp.edge_off) {             // node rows: the two row means and the result in one launch, a wave per row
            fused_node_ln_mod_bwd(s, rows, F, dy, xhat, rstd, row_mol, mods, ldm, sc, dx, acc);
            return;
        }
        JT_LAUNCH(k_ln_bwd_part, rows * 8, s, rows, F, dy, xhat, row_mol, mods, ldm, sc, b.rowpart);
        JT_LAUNCH(k_ln_bwd_stats, rows, s, rows, F, (const float*)b.rowpart, b.tRow[0], b.tRow[1]);
        JT_LAUNCH(k_ln_bwd_apply, rows * F, s, rows, F, dy, xhat, rstd, (const float*)b.tRow[0], (const float*)b.tRow[1], row_mol, mods,
                           ldm, sc, dx, acc);
    }
    // per-molecule sums: edge rows in two levels, node rows (at most 181 per molecule) directly
    void seg(int F, const int* off, const float* a, const float* bb, float* out, int ldo, int ocol) const {
        if (off == tp.edge_off) seg_edge(F, a, bb, out, ldo, ocol);
        else JT_LAUNCH(k_seg_colsum, (long)t.B * F, s, t.B, F, off, a, bb, out, ldo, ocol, 0, drop(0.f, 0, 0, 0));
    }
    // node rows: out[mol, ocol + f] = sum_rows a (bb * dropout)
    void seg_node_drop(int F, const float* a, const float* bb, Drop db, float* out, int ldo, int ocol) const {
        JT_LAUNCH(k_seg_colsum, (long)t.B * F, s, t.B, F, tp.node_off, a, bb, out, ldo, ocol, 0, db);
    }
    void copy2d(long rows, int F, const float* src, int lds, int scol, float* dst, int ldd, int dcol, int acc) const {
        JT_LAUNCH(k_copy2d, rows * F, s, rows, F, src, lds, scol, dst, ldd, dcol, acc);
    }
    Drop drop(float p, unsigned long long seed, int l, int site) const { Drop d; d.p = p; d.seed = seed; d.site = (unsigned)(l * 8 + site); return d; }
};

// Dropout sites of a block.  Only the four FFN sites are live (mol_gnn.py:308-317: `self.dropout` after the activation and after the
// second linear of the node and of the edge FFN).  SITE_ALPHA — F.dropout on the softmax weights, layers.py:179 — is an IDENTITY in the
// reference: EquivariantMixBlock builds its TransMixLayer without a dropout argument (mol_gnn.py:230-231), so TransMixLayer.dropout keeps
// its default 0.0 (layers.py:99) whatever config.model.dropout says.  The site is kept (with p = 0) so that the kernels' signatures and
// the (seed, site) numbering of the four live sites do not move.
enum { SITE_ALPHA = 0, SITE_A1, SITE_F2, SITE_A3, SITE_F4 };

// MLP head: Linear SiLU Linear SiLU Linear; saves the two pre-activations and activations
void head_fwd(const Ctx& c, const float* X, int ldx, long rows, int K, Lin l0, Lin l2, Lin l4, int H1, int H2, int NO, float* p1, float* a1, float* p2, float* a2,
              float* out, int ldo) {
    const Drop nod = c.drop(0.f, 0, 0, 0);
    c.lin_silu(X, ldx, rows, K, c.p(l0.w), K, H1, c.p(l0.b), p1, a1, nod);
    c.lin_silu(a1, H1, rows, H1, c.p(l2.w), H1, H2, c.p(l2.b), p2, a2, nod);
    c.lin(a2, H2, rows, H2, c.p(l4.w), H2, NO, c.p(l4.b), out, ldo, 0);
}
// dX (ldx, acc) from dOut (ldo); t1 [rows, H1], t2 [rows, H2] scratch
void head_bwd(const Ctx& c, const float* X, int ldx, long rows, int K, Lin l0, Lin l2, Lin l4, int H1, int H2, int NO, const float* p1, const float* a1,
              const float* p2, const float* a2, const float* dOut, int ldo, float* t1, float* t2, float* dX, int lddx, int acc) {
    const Drop nod = c.drop(0.f, 0, 0, 0);
    c.lin_dw(dOut, ldo, rows, NO, a2, H2, H2, c.g(l4.w), H2, c.g(l4.b));
    c.lin_dx(dOut, ldo, rows, NO, c.p(l4.w), H2, H2, t2, H2, 0);
    c.silu_bwd(rows * H2, p2, t2, t2, nod);
    c.lin_dw(t2, H2, rows, H2, a1, H1, H1, c.g(l2.w), H1, c.g(l2.b));
    c.lin_dx(t2, H2, rows, H2, c.p(l2.w), H1, H1, t1, H1, 0);
    c.silu_bwd(rows * H1, p1, t1, t1, nod);
    c.lin_dw(t1, H1, rows, H1, X, ldx, K, c.g(l0.w), K, c.g(l0.b));
    c.lin_dx(t1, H1, rows, H1, c.p(l0.w), K, K, dX, lddx, acc);
    c.flush_dw();                                            // t1 / t2 are the next head's scratch too
}

// the 4 L + 1 modulation projections in the order of their columns in [., Mtot]: top-level GBF | per block node, edge, equi, GBF
int mod_entries(const jodo_train& t, Lin* lin, int* F, int* col) {
    int n = 0, at = 0;
    auto add = [&](Lin l, int f) { lin[n] = l; F[n] = f; col[n] = at; at += f; ++n; };
    add(t.gbf_time, 2);
    for (int l = 0; l < t.L; ++l) {
        const BlkIx& ix = t.blk[l];
        add(ix.node_time, 6 * t.D); add(ix.edge_time, 6 * t.De); add(ix.eq_time, 2 * t.D); add(ix.gbf_time, 2);
    }
    return n;
}

void forward(const Ctx& c, const float* xh, const float* edge_x, const float* cond_x, const float* cond_edge_x, const float* nl, const float* context,
             float p_drop, unsigned long long seed, float* out_xh, float* out_edge) {
    const jodo_train& t = c.t; Bufs& b = c.b; const Topo& tp = c.tp; hipStream_t s = c.s;
    const int B = t.B, Nn = t.Nn, R = t.R, D = t.D, De = t.De, T = t.T, L = t.L, QK = t.QK, H = t.H, r = t.r, nd = t.nd, ch = t.ch;
    const int F17 = 2 * t.half + 1, ldin = 2 * ch + De;
    const Drop nod = c.drop(0.f, 0, 0, 0);
    (void)hipMemsetAsync(b.flags, 0, 8 * sizeof(int), s);
    JT_LAUNCH(k_pack_nodes, Nn, s, tp, nd, xh, cond_x, b.pos[0], b.cpos, b.nin);
    JT_LAUNCH(k_pack_edges, R, s, tp, ch, ldin, t.cfg.edge_quan_th, t.cfg.spatial_cut_off, edge_x, cond_edge_x, (const float*)b.cpos, b.ein,
                       b.adj2d, b.adjsp, b.d2c, b.flags + 3);
    // time embedding (mol_gnn.py:481-489) and, for the conditional model, cond_lin(cond_mlp(context)) (:728-734)
    JT_LAUNCH(k_time_feat, (long)B * F17, s, B, t.half, nl, c.p(t.time_w), b.feat);
    c.lin(b.feat, F17, B, F17, c.p(t.time1.w), F17, T, c.p(t.time1.b), b.t1pre, T, 0);
    JT_LAUNCH(k_gelu_fwd, (long)B * T, s, (long)B * T, (const float*)b.t1pre, b.t1a);
    c.lin(b.t1a, T, B, T, c.p(t.time3.w), T, T, c.p(t.time3.b), b.temb, T, 0);
    if (t.cc > 0) {
        (void)hipMemcpyAsync(b.ctx, context, (size_t)B * t.cc * 4, hipMemcpyDeviceToDevice, s);
        c.lin(b.ctx, 1, B * t.cc, 1, c.p(t.cond0.w), 1, D, c.p(t.cond0.b), b.cc0pre, D, 0);
        JT_LAUNCH(k_gelu_fwd, (long)B * t.cc * D, s, (long)B * t.cc * D, (const float*)b.cc0pre, b.cc0a);
        c.lin(b.cc0a, D, B * t.cc, D, c.p(t.cond2.w), D, D, c.p(t.cond2.b), b.cc2, D, 0);
        c.lin(b.cc2, t.cc * D, B, t.cc * D, c.p(t.cond_lin.w), t.cc * D, T, c.p(t.cond_lin.b), b.temb, T, 1);
    }
    c.silu((long)B * T, b.temb, b.tau, nod);
    // embeddings (:547-560); the first-step switch of :544 is the device flag [3]
    {   // every modulation row of every block in ONE product (train_ops.h ModTable): gather the weights, project, hand the rows out
        ModTable M;
        Lin lin[MOD_MAX]; int F[MOD_MAX], col[MOD_MAX];
        M.n = mod_entries(t, lin, F, col);
        int fmax = 0;
        for (int i = 0; i < M.n; ++i) {
            M.w[i] = c.p(lin[i].w); M.bias[i] = c.p(lin[i].b); M.F[i] = F[i]; M.col[i] = col[i];
            const int l = (i - 1) / 4, k = (i - 1) % 4;
            M.out[i] = i == 0 ? b.gm_top : (k == 0 ? b.blk[l].nmod : (k == 1 ? b.blk[l].emod : (k == 2 ? b.blk[l].qmod : b.blk[l].gm)));
            fmax = F[i] > fmax ? F[i] : fmax;
        }
        hipLaunchKernelGGL(k_mod_gather, dim3((unsigned)(((long)fmax * T + 255) / 256), (unsigned)M.n), dim3(256), 0, s, M, T, b.Wall, b.ball);
        c.lin(b.tau, T, B, T, b.Wall, T, t.Mtot, b.ball, b.mods_all, t.Mtot, 0);
        hipLaunchKernelGGL(k_mod_scatter, dim3((unsigned)(((long)B * fmax + 255) / 256), (unsigned)M.n), dim3(256), 0, s, M, B, t.Mtot, (const float*)b.mods_all);
    }
    const int F3 = 2 * QK + D;
    {   // lin_query, lin_key, lin_value of every block share their input (the modulated LayerNorm1 of h): their weights are gathered into
        // [L][2 QK + D, D] once per forward so that a block needs ONE product instead of three (and its backward two instead of six)
        ModTable M;
        M.n = 3 * L;
        for (int l = 0; l < L; ++l) {
            const BlkIx& ix = t.blk[l];
            const Lin ls[3] = {ix.query, ix.key, ix.value};
            for (int j = 0; j < 3; ++j) {
                const int i = 3 * l + j;
                M.w[i] = c.p(ls[j].w); M.bias[i] = c.p(ls[j].b); M.out[i] = nullptr; M.F[i] = j < 2 ? QK : D; M.col[i] = l * F3 + j * QK;
            }
        }
        hipLaunchKernelGGL(k_mod_gather, dim3((unsigned)(((long)D * D + 255) / 256), (unsigned)M.n), dim3(256), 0, s, M, D, b.Wqkv, b.bqkv);
    }
    JT_LAUNCH(k_gbf_fwd, (long)R * De, s, (long)R, De, (const float*)b.d2c, tp.edge_mol, (const float*)b.gm_top, c.p(t.gbf_means), c.p(t.gbf_stds),
                       (const int*)(b.flags + 3), b.ein, ldin, 2 * ch);
    c.lin(b.ein, ldin, R, ldin, c.p(t.edge_emb.w), ldin, De, c.p(t.edge_emb.b), b.e[0], De, 0);
    c.lin(b.nin, 2 * nd, Nn, 2 * nd, c.p(t.node_emb.w), 2 * nd, D, c.p(t.node_emb.b), b.h[0], D, 0);
    c.copy2d(Nn, D, b.h[0], D, 0, b.ah, t.catn, 0, 0);
    c.copy2d(R, De, b.e[0], De, 0, b.eh, t.cate, 0, 0);
    for (int l = 0; l < L; ++l) {
        const BlkIx& ix = t.blk[l]; BlkBuf& k = b.blk[l];
        // distances, Gaussian basis, edge_emb([G, e]) and the two modulated LayerNorms (:279-296)
        const FusedDims fd{D, De, r, QK, t.ce, L};
        FusedBlockParams fp;
        FusedTopo ft{R, tp.edge_a, tp.edge_c, tp.edge_mol, t.save_activations};
        float* fpk = b.fpack + (size_t)l * b.fpack_block;
        if (t.fused) {
            fp.edge_emb_w = c.p(ix.edge_emb.w); fp.edge_emb_b = c.p(ix.edge_emb.b); fp.le0 = c.p(ix.le0); fp.le1 = c.p(ix.le1);
            fp.ff3_w = c.p(ix.ff3.w); fp.ff3_b = c.p(ix.ff3.b); fp.ff4_w = c.p(ix.ff4.w); fp.ff4_b = c.p(ix.ff4.b);
            fp.ero_w = c.p(ix.edge_ro.w); fp.ero_b = c.p(ix.edge_ro.b); fp.in_w = c.p(ix.eq_in.w); fp.in_b = c.p(ix.eq_in.b);
            fp.c0_w = c.p(ix.eq_c0.w); fp.c0_b = c.p(ix.eq_c0.b); fp.c2_w = c.p(ix.eq_c2); fp.n2e_b = c.p(ix.n2e.b);
            fp.gbf_means = c.p(ix.gbf_means); fp.gbf_stds = c.p(ix.gbf_stds);
            fused_pack_block(s, fd, fp, fpk);
            // chain A: d2 -> G -> edge_emb -> LN1 -> modulate -> tanh(lin_edge0 / lin_edge1), one kernel (train_fused.hip)
            fused_chain_a(s, fd, ft, fp, fpk, b.pos[l], k.gm, b.e[l], k.emod, k.d2, k.G, k.xh_e1, k.rs_e1, k.et, k.t0, k.t1);
        } else {
            JT_LAUNCH(k_dist2, R, s, tp, (const float*)b.pos[l], k.d2);
            JT_LAUNCH(k_gbf_fwd, (long)R * De, s, (long)R, De, (const float*)k.d2, tp.edge_mol, (const float*)k.gm, c.p(ix.gbf_means), c.p(ix.gbf_stds),
                               (const int*)nullptr, k.G, De, 0);
            float* e1 = b.tE_De[0];
            c.lin(k.G, De, R, De, c.p(ix.edge_emb.w), 2 * De, De, c.p(ix.edge_emb.b), e1, De, 0);
            c.lin(b.e[l], De, R, De, c.p(ix.edge_emb.w) + De, 2 * De, De, nullptr, e1, De, 1);
            c.stats(R, De, e1, b.tRow[0], k.rs_e1);
            c.ln_mod(R, De, e1, b.tRow[0], k.rs_e1, tp.edge_mol, k.emod, 6 * De, 0, De, k.xh_e1, k.et);
        }
        if (t.fused) fused_node_ln_mod(s, Nn, D, b.h[l], nullptr, tp.node_mol, k.nmod, 6 * D, 0, 0, D, k.xh_h, k.rs_h, k.ht);
        else {
            c.stats(Nn, D, b.h[l], b.tRow[0], k.rs_h);
            c.ln_mod(Nn, D, b.h[l], b.tRow[0], k.rs_h, tp.node_mol, k.nmod, 6 * D, 0, D, k.xh_h, k.ht);
        }
        // attention (layers.py:131-186)
        {   // q | k | v in one product on the gathered weights, then handed out to their arrays
            c.lin(k.ht, D, Nn, D, b.Wqkv + (size_t)l * F3 * D, D, F3, b.bqkv + (size_t)l * F3, k.qkv, F3, 0);
            if (!t.fused_attn) {                               // (the op-by-op attention kernels read compact q, k, v)
                ModTable M;
                M.n = 3;
                float* outs[3] = {k.q, k.k, k.v};
                for (int j = 0; j < 3; ++j) { M.w[j] = nullptr; M.bias[j] = nullptr; M.out[j] = outs[j]; M.F[j] = j < 2 ? QK : D; M.col[j] = j * QK; }
                hipLaunchKernelGGL(k_mod_scatter, dim3((unsigned)(((long)Nn * D + 255) / 256), 3u), dim3(256), 0, s, M, Nn, F3, (const float*)k.qkv);
            }
        }
        if (!t.fused) {
            c.lin_tanh(k.et, De, R, De, c.p(ix.le0), De, QK, nullptr, k.t0);
            c.lin_tanh(k.et, De, R, De, c.p(ix.le1), De, D, nullptr, k.t1);
        }
        const AttnTopo at{Nn, t.N, tp.node_mol, tp.nn, tp.node_off, tp.edge_off, t.fused_attn == 2 ? 1 : 0, F3, F3, F3, F3};
        if (t.fused_attn) {
            // scores | column softmax | messages in one launch, a wave per target atom (train_fused.hip; bit-identical to the three below);
            // q, k, v read where the merged projection wrote them
            fused_attn_fwd(s, at, D, H, t.XH, t.SC, 1.f / sqrtf((float)t.C), k.qkv, k.qkv + QK, k.t0, b.adj2d, b.adjsp, k.qkv + 2 * QK, k.t1, k.alpha, k.hhat);
        } else {
            JT_LAUNCH(k_attn_scores, (long)R * H, s, tp, H, t.XH, t.SC, 1.f / sqrtf((float)t.C), (const float*)k.q, (const float*)k.k,
                               (const float*)k.t0, (const float*)b.adj2d, (const float*)b.adjsp, k.alpha);
            JT_LAUNCH(k_attn_softmax, (long)Nn * H, s, tp, H, k.alpha);
            JT_LAUNCH(k_attn_msg, (long)Nn * D, s, tp, D, H, (const float*)k.v, (const float*)k.t1, (const float*)k.alpha,
                               c.drop(0.f, seed, l, SITE_ALPHA), k.hhat);     // p = 0: see SITE_ALPHA
        }
        c.lin(k.hhat, D, Nn, D, c.p(ix.n2e.w), D, De, nullptr, k.n2e, De, 0);
        // edges: gated residual, LayerNorm2 + modulate, FFN (:313-317)
        if (t.fused) {
            // chain B: residual -> LN2 -> modulate -> ff_linear3 -> SiLU, dropout -> ff_linear4 -> dropout -> gate -> readout (:570), one kernel
            fused_chain_b(s, fd, ft, fp, fpk, b.e[l], k.n2e, k.emod, c.drop(p_drop, seed, l, SITE_A3), c.drop(p_drop, seed, l, SITE_F4), k.xh_en, k.rs_en,
                          k.en, k.f3, k.a3, k.f4, b.e[l + 1], b.eh, t.cate, De + l * t.ce);
        } else {
            float* x1e = b.tE_De[0];
            JT_LAUNCH(k_edge_bcast, (long)R * De, s, tp, De, (const float*)b.e[l], (const float*)k.n2e, (const float*)k.n2e, c.p(ix.n2e.b),
                               (const float*)k.emod, 6 * De, 2 * De, x1e);
            c.stats(R, De, x1e, b.tRow[0], k.rs_en);
            c.ln_mod(R, De, x1e, b.tRow[0], k.rs_en, tp.edge_mol, k.emod, 6 * De, 3 * De, 4 * De, k.xh_en, k.en);
            c.lin_silu(k.en, De, R, De, c.p(ix.ff3.w), De, r * De, c.p(ix.ff3.b), k.f3, k.a3, c.drop(p_drop, seed, l, SITE_A3));
            c.lin(k.a3, r * De, R, r * De, c.p(ix.ff4.w), r * De, De, c.p(ix.ff4.b), k.f4, De, 0);
            JT_LAUNCH(k_drop, (long)R * De, s, (long)R * De, (const float*)k.f4, b.tE_De[1], c.drop(p_drop, seed, l, SITE_F4));
            JT_LAUNCH(k_gate_add, (long)R * De, s, (long)R, De, (const float*)k.en, (const float*)b.tE_De[1], tp.edge_mol, (const float*)k.emod,
                               6 * De, 5 * De, b.e[l + 1]);
        }
        // nodes: gated residual, LayerNorm2 + modulate, FFN (:307-311)
        if (t.fused) {
            // gated residual, row statistics, LayerNorm2 + modulate: one launch (a wave per row)
            fused_node_ln_mod(s, Nn, D, b.h[l], k.hhat, tp.node_mol, k.nmod, 6 * D, 2 * D, 3 * D, 4 * D, k.xh_hn, k.rs_hn, k.hn);
        } else {
            float* x1n = b.tN_D[0];
            JT_LAUNCH(k_gate_add, (long)Nn * D, s, (long)Nn, D, (const float*)b.h[l], (const float*)k.hhat, tp.node_mol, (const float*)k.nmod,
                               6 * D, 2 * D, x1n);
            c.stats(Nn, D, x1n, b.tRow[0], k.rs_hn);
            c.ln_mod(Nn, D, x1n, b.tRow[0], k.rs_hn, tp.node_mol, k.nmod, 6 * D, 3 * D, 4 * D, k.xh_hn, k.hn);
        }
        c.lin_silu(k.hn, D, Nn, D, c.p(ix.ff1.w), D, r * D, c.p(ix.ff1.b), k.f1, k.a1, c.drop(p_drop, seed, l, SITE_A1));
        c.lin(k.a1, r * D, Nn, r * D, c.p(ix.ff2.w), r * D, D, c.p(ix.ff2.b), k.f2, D, 0);
        JT_LAUNCH(k_drop_gate_add, (long)Nn * D, s, (long)Nn, D, (const float*)k.hn, (const float*)k.f2, c.drop(p_drop, seed, l, SITE_F2), tp.node_mol,
                           (const float*)k.nmod, 6 * D, 5 * D, b.h[l + 1]);       // h' = hn + g2 dropout(ff2)
        // equivariant update (mol_gnn.py:71-94): input_lin([h_row, h_col, e, G]) factored per node / per edge
        const int ldw = 2 * D + 2 * De;
        const float* Win = c.p(ix.eq_in.w);
        float *hr = b.tN_D[0], *hc = b.tN_D[1], *pre = b.tE_D[0];
        c.lin(b.h[l + 1], D, Nn, D, Win, ldw, D, nullptr, hr, D, 0);
        c.lin(b.h[l + 1], D, Nn, D, Win + D, ldw, D, nullptr, hc, D, 0);
        if (t.fused) {
            // chain C: input_lin -> LN -> modulate -> coord_mlp.0 -> SiLU -> coord_mlp.2 -> tanh, one kernel
            fused_chain_c(s, fd, ft, fp, fpk, b.e[l + 1], k.G, hr, hc, k.qmod, k.xh_pre, k.rs_pre, k.u, k.c0pre, k.c0a, k.inv);
        } else {
            c.lin(b.e[l + 1], De, R, De, Win + 2 * D, ldw, D, c.p(ix.eq_in.b), pre, D, 0);
            c.lin(k.G, De, R, De, Win + 2 * D + De, ldw, D, nullptr, pre, D, 1);
            JT_LAUNCH(k_edge_bcast, (long)R * D, s, tp, D, (const float*)pre, (const float*)hr, (const float*)hc, (const float*)nullptr,
                               (const float*)nullptr, 0, 0, b.tE_D[1]);
            c.stats(R, D, b.tE_D[1], b.tRow[0], k.rs_pre);
            c.ln_mod(R, D, b.tE_D[1], b.tRow[0], k.rs_pre, tp.edge_mol, k.qmod, 2 * D, 0, D, k.xh_pre, k.u);
            c.lin_silu(k.u, D, R, D, c.p(ix.eq_c0.w), D, D, c.p(ix.eq_c0.b), k.c0pre, k.c0a, nod);
            c.lin_tanh(k.c0a, D, R, D, c.p(ix.eq_c2), D, 3, nullptr, k.inv);
        }
        JT_LAUNCH(k_coord_fwd, R, s, tp, (const float*)b.pos[l], (const float*)k.inv, (const float*)b.adj2d, (const float*)b.adjsp,
                           c.p(ix.eq_scale), b.tE3[0]);
        JT_LAUNCH(k_coord_sum, (long)Nn * 3, s, tp, (const float*)b.pos[l], (const float*)b.tE3[0], b.tN3[0]);
        JT_LAUNCH(k_center, (long)B * 3, s, tp, (const float*)b.tN3[0], (const int*)nullptr, b.pos[l + 1]);
        // readouts written into the head inputs in place (:569-570)
        c.lin(b.h[l + 1], D, Nn, D, c.p(ix.node_ro.w), D, t.cn, c.p(ix.node_ro.b), b.ah + D + l * t.cn, t.catn, 0);
        if (!t.fused) c.lin(b.e[l + 1], De, R, De, c.p(ix.edge_ro.w), De, t.ce, c.p(ix.edge_ro.b), b.eh + De + l * t.ce, t.cate, 0);
    }
    // heads and outputs (:572-594)
    head_fwd(c, b.ah, t.catn, Nn, t.catn, t.np0, t.np2, t.np4, D, D / 2, nd, b.nh1pre, b.nh1, b.nh2pre, b.nh2, b.atom, nd);
    head_fwd(c, b.eh, t.cate, R, t.cate, t.ee0, t.ee2, t.ee4, De, De / 2, 1, b.x1pre[0], b.x1[0], b.x2pre[0], b.x2[0], b.Ep, ch);
    head_fwd(c, b.eh, t.cate, R, t.cate, t.et0, t.et2, t.et4, De, De / 2, ch - 1, b.x1pre[1], b.x1[1], b.x2pre[1], b.x2[1], b.Ep + 1, ch);
    JT_LAUNCH(k_edge_out, (long)B * t.N * t.N * ch, s, tp, ch, (const float*)b.Ep, out_edge);
    JT_LAUNCH(k_nan_flag, (long)Nn * 3, s, (long)Nn * 3, (const float*)b.pos[L], b.flags);
    JT_LAUNCH(k_center, (long)B * 3, s, tp, (const float*)b.pos[L], (const int*)b.flags, b.posf);
    JT_LAUNCH(k_node_out, (long)B * t.N * (3 + nd), s, tp, nd, (const float*)b.posf, (const float*)b.atom, out_xh);
}

// Modulation gradients: every reduction that produces one writes straight into its columns of dmods_all [B, Mtot] (mod_entries order);
// the products (dW += dmod^T tau, db += column sums, dtau += dmod W) run once for all of them at the end of the backward (mod_bwd_all).
// Columns of block l: node | edge | equivariant | Gaussian layer; the top-level Gaussian layer's two are columns 0 .. 1.
struct ModCols { int node, edge, eq, gbf; };
ModCols mod_cols(const jodo_train& t, int l) {
    const int base = 2 + l * (6 * t.D + 6 * t.De + 2 * t.D + 2);
    return ModCols{base, base + 6 * t.D, base + 6 * t.D + 6 * t.De, base + 6 * t.D + 6 * t.De + 2 * t.D};
}
void mod_bwd_all(const Ctx& c) {
    const jodo_train& t = c.t; Bufs& b = c.b; hipStream_t s = c.s;
    (void)hipMemsetAsync(b.dWall, 0, (size_t)t.Mtot * t.T * 4, s);
    (void)hipMemsetAsync(b.dball, 0, (size_t)t.Mtot * 4, s);
    c.lin_dw(b.dmods_all, t.Mtot, t.B, t.Mtot, b.tau, t.T, t.T, b.dWall, t.T, b.dball);
    c.flush_dw();
    c.lin_dx(b.dmods_all, t.Mtot, t.B, t.Mtot, b.Wall, t.T, t.T, b.dtau, t.T, 1);
    ModGradTable M;
    Lin lin[MOD_MAX]; int F[MOD_MAX], col[MOD_MAX];
    M.n = mod_entries(t, lin, F, col);
    int fmax = 0;
    for (int i = 0; i < M.n; ++i) { M.gw[i] = c.g(lin[i].w); M.gb[i] = c.g(lin[i].b); M.F[i] = F[i]; M.col[i] = col[i]; fmax = F[i] > fmax ? F[i] : fmax; }
    hipLaunchKernelGGL(k_mod_scatter_grads, dim3((unsigned)(((long)fmax * t.T + 255) / 256), (unsigned)M.n), dim3(256), 0, s, M, t.T, (const float*)b.dWall, (const float*)b.dball);
}

void gbf_bwd(const Ctx& c, long rows, const float* d2, const float* gm, int means, int stds, float* dgm, int ldd, const float* dG, int ldg, int gcol, float* dd2) {
    const jodo_train& t = c.t; Bufs& b = c.b; hipStream_t s = c.s;
    const int De = t.De, K = De - 1;
    const int chunk = 32;
    const long nch = (rows + chunk - 1) / chunk;
    float *pm = b.tE_QK, *ps = b.tE_QK + nch * K;            // (the attention scratch is free here)
    if (t.fused_bwd && De <= 129 && (t.gbf_chunk == 2 || (t.gbf_chunk == 1 && nch >= 4096))) {
        // d x' per row and the chunk partials of d means / d stds in one pass: a wave per 32-row chunk, a lane per Gaussian (train_fused.hip).
        // Batches that fill the card only: batch 2 048 backward 90.5 -> 88.0 ms; at the reference's batch (1 344 chunks) a wave walking its
        // 32 rows is slower than 43 k independent threads (backward 8.98 -> 9.16 ms)
        fused_gbf_bwd(s, rows, De, d2, c.tp.edge_mol, gm, c.p(means), c.p(stds), dG, ldg, gcol, b.tRow[2], dd2, 0, pm, ps);
    } else {
        JT_LAUNCH(k_gbf_bwd_row, rows, s, rows, De, d2, c.tp.edge_mol, gm, c.p(means), c.p(stds), dG, ldg, gcol, b.tRow[2], dd2, 0);
        JT_LAUNCH(k_gbf_bwd_par, nch * K, s, rows, De, chunk, d2, c.tp.edge_mol, gm, c.p(means), c.p(stds), dG, ldg, gcol, pm, ps);
    }
    c.seg2_edge(1, b.tRow[2], d2, dgm, ldd, 1, 0);          // d shift = sum dx' (column 1), d scale = sum dx' d2 (column 0) per molecule, one pass
    c.colsum(pm, K, nullptr, 0, nch, K, c.g(means));
    c.colsum(ps, K, nullptr, 0, nch, K, c.g(stds));
}

void backward(const Ctx& c, const float* nl, const float* d_out_xh, const float* d_out_edge, float p_drop, unsigned long long seed) {
    const jodo_train& t = c.t; Bufs& b = c.b; const Topo& tp = c.tp; hipStream_t s = c.s;
    const int B = t.B, Nn = t.Nn, R = t.R, D = t.D, De = t.De, T = t.T, L = t.L, QK = t.QK, H = t.H, r = t.r, nd = t.nd, ch = t.ch;
    const int F17 = 2 * t.half + 1, ldin = 2 * ch + De;
    const Drop nod = c.drop(0.f, 0, 0, 0);
    // gradients are accumulated below: zero them first — adjacent buffers (a caller that carves all gradients out of one
    // allocation, as jodo_amd/train.py does) in one fill instead of one per tensor
    for (int i = 0; i < t.n_params;) {
        char* beg = reinterpret_cast<char*>(c.g(i));
        size_t bytes = t.numel[i] * 4;
        int j = i + 1;
        // (adjacent, or behind the alignment padding of jodo_amd/optim.py slice_offsets — slices start on 16-byte boundaries, so a gap
        // is under 16 bytes — which is filled along: the norm of the whole flat buffer is taken, padding included.  Contract stated in
        // include/jodo_hip.h at jodo_train_backward: bytes between two gradient buffers less than 16 bytes apart are zeroed too)
        while (j < t.n_params && reinterpret_cast<char*>(c.g(j)) >= beg + bytes && reinterpret_cast<char*>(c.g(j)) - (beg + bytes) < 16) {
            bytes = (size_t)(reinterpret_cast<char*>(c.g(j)) - beg) + t.numel[j] * 4;
            ++j;
        }
        (void)hipMemsetAsync(beg, 0, bytes, s);
        i = j;
    }
    b.defer_fin = true;                                      // later stages of the reductions are queued and run once per block (flush_fin)
    b.finq.clear(); b.fin_used = 0;
    const int Mt = t.Mtot;
    (void)hipMemsetAsync(b.dtau, 0, (size_t)B * T * 4, s);
    (void)hipMemsetAsync(b.dWqkv, 0, (size_t)L * (2 * QK + D) * D * 4, s);
    (void)hipMemsetAsync(b.dbqkv, 0, (size_t)L * (2 * QK + D) * 4, s);
    // outputs -> packed gradients; final centring (skipped, gradient zero, when the NaN guard fired)
    float *dposf = b.tN3[0], *datom = b.tN_De, *dEp = b.tE3[0];
    JT_LAUNCH(k_node_out_bwd, (long)Nn * (3 + nd), s, tp, nd, d_out_xh, dposf, datom);
    JT_LAUNCH(k_edge_out_bwd, (long)R * ch, s, tp, ch, d_out_edge, dEp);
    float *dpos = b.tN3[1], *dpos_prev = b.tN3[2];
    JT_LAUNCH(k_center, (long)B * 3, s, tp, (const float*)dposf, (const int*)b.flags, dpos);
    float *dah = b.tcatn, *deh = b.tcate;
    head_bwd(c, b.ah, t.catn, Nn, t.catn, t.np0, t.np2, t.np4, D, D / 2, nd, b.nh1pre, b.nh1, b.nh2pre, b.nh2, datom, nd, b.tN_D[0], b.tN_D[1], dah, t.catn, 0);
    head_bwd(c, b.eh, t.cate, R, t.cate, t.ee0, t.ee2, t.ee4, De, De / 2, 1, b.x1pre[0], b.x1[0], b.x2pre[0], b.x2[0], dEp, ch, b.tE_De[0], b.tE_De[1], deh, t.cate, 0);
    head_bwd(c, b.eh, t.cate, R, t.cate, t.et0, t.et2, t.et4, De, De / 2, ch - 1, b.x1pre[1], b.x1[1], b.x2pre[1], b.x2[1], dEp + 1, ch, b.tE_De[0], b.tE_De[1], deh,
             t.cate, 1);
    float *dh = b.tN_D[2], *dh_prev = b.tN_D[3], *de = b.tE_De[2], *de_prev = b.tE_De[3];
    (void)hipMemsetAsync(dh, 0, (size_t)Nn * D * 4, s);           // the head inputs hold h / e after the EMBEDDINGS in their first columns (:562),
    (void)hipMemsetAsync(de, 0, (size_t)R * De * 4, s);           // so that part of dah / deh joins d h[0] / d e[0] after the loop
    for (int l = L - 1; l >= 0; --l) {
        const BlkIx& ix = t.blk[l]; BlkBuf& k = b.blk[l];
        const ModCols mc = mod_cols(t, l);
        float *dnmod = b.dmods_all + mc.node, *demod = b.dmods_all + mc.edge, *dqmod = b.dmods_all + mc.eq, *dgm = b.dmods_all + mc.gbf;   // row stride Mt
        // readouts
        c.lin_dw(dah + D + l * t.cn, t.catn, Nn, t.cn, b.h[l + 1], D, D, c.g(ix.node_ro.w), D, c.g(ix.node_ro.b));
        c.lin_dx(dah + D + l * t.cn, t.catn, Nn, t.cn, c.p(ix.node_ro.w), D, D, dh, D, 1);
        c.lin_dw(deh + De + l * t.ce, t.cate, R, t.ce, b.e[l + 1], De, De, c.g(ix.edge_ro.w), De, c.g(ix.edge_ro.b));
        c.lin_dx(deh + De + l * t.ce, t.cate, R, t.ce, c.p(ix.edge_ro.w), De, De, de, De, 1);
        // ---- equivariant update, backwards: centring, position sums, CoorsNorm, tanh, coord_mlp, LayerNorm + modulate, input_lin
        float *dxp = b.tN3[3], *dinv = b.tE3[0], *ddiff = b.tE3[1];
        JT_LAUNCH(k_center, (long)B * 3, s, tp, (const float*)dpos, (const int*)nullptr, dxp);
        JT_LAUNCH(k_coord_bwd, R, s, tp, (const float*)b.pos[l], (const float*)k.inv, (const float*)b.adj2d, (const float*)b.adjsp,
                           c.p(ix.eq_scale), (const float*)dxp, dinv, ddiff, b.tRow[2]);
        c.colsum(b.tRow[2], 1, nullptr, 0, R, 1, c.g(ix.eq_scale));
        const FusedDims fd{D, De, r, QK, t.ce, L};
        FusedBlockParams fp;
        FusedTopo ft{R, tp.edge_a, tp.edge_c, tp.edge_mol};
        float* fpk = b.fpack + (size_t)l * b.fpack_block;
        float *dc0 = b.tE_D[0], *du = b.tE_D[1], *dpre = b.tE_D[2], *dG = b.tE_De[0];
        if (t.fused_bwd) {
            fp.edge_emb_w = c.p(ix.edge_emb.w); fp.edge_emb_b = c.p(ix.edge_emb.b); fp.le0 = c.p(ix.le0); fp.le1 = c.p(ix.le1);
            fp.ff3_w = c.p(ix.ff3.w); fp.ff3_b = c.p(ix.ff3.b); fp.ff4_w = c.p(ix.ff4.w); fp.ff4_b = c.p(ix.ff4.b);
            fp.ero_w = c.p(ix.edge_ro.w); fp.ero_b = c.p(ix.edge_ro.b); fp.in_w = c.p(ix.eq_in.w); fp.in_b = c.p(ix.eq_in.b);
            fp.c0_w = c.p(ix.eq_c0.w); fp.c0_b = c.p(ix.eq_c0.b); fp.c2_w = c.p(ix.eq_c2); fp.n2e_b = c.p(ix.n2e.b);
            fp.gbf_means = c.p(ix.gbf_means); fp.gbf_stds = c.p(ix.gbf_stds);
            fused_pack_block_bwd(s, fd, fp, fpk);
            // chain C': tanh', coord_mlp.2^T, SiLU', coord_mlp.0^T, LayerNorm + modulate backward, input_lin[e ; G]^T — one kernel; it leaves
            // dinv (in place), dc0, du, dpre for the weight-gradient products and the modulation sums below, adds into de and writes dG
            fused_bwd_c(s, fd, ft, fp, fpk, k.inv, dinv, k.c0pre, k.xh_pre, k.rs_pre, k.qmod, dc0, du, dpre, de, dG);
            c.lin_dw(dinv, 3, R, 3, k.c0a, D, D, c.g(ix.eq_c2), D);
            c.lin_dw(dc0, D, R, D, k.u, D, D, c.g(ix.eq_c0.w), D, c.g(ix.eq_c0.b));
            c.seg2_edge(D, du, k.xh_pre, dqmod, Mt, 0, D);
        } else {
            JT_LAUNCH(k_tanh_bwd, (long)R * 3, s, (long)R * 3, (const float*)k.inv, dinv);
            c.lin_dw(dinv, 3, R, 3, k.c0a, D, D, c.g(ix.eq_c2), D);
            c.lin_dx(dinv, 3, R, 3, c.p(ix.eq_c2), D, D, dc0, D, 0);
            c.silu_bwd((long)R * D, k.c0pre, dc0, dc0, nod);
            c.lin_dw(dc0, D, R, D, k.u, D, D, c.g(ix.eq_c0.w), D, c.g(ix.eq_c0.b));
            c.lin_dx(dc0, D, R, D, c.p(ix.eq_c0.w), D, D, du, D, 0);
            c.ln_mod_bwd(R, D, du, k.xh_pre, k.rs_pre, tp.edge_mol, tp.edge_off, k.qmod, 2 * D, 0, D, dqmod, Mt, dpre, 0);
        }
        const int ldw = 2 * D + 2 * De;
        const float* Win = c.p(ix.eq_in.w); float* dWin = c.g(ix.eq_in.w);
        float *dhr = b.tN_D2[0], *dhc = b.tN_D2[1];           // (not tN_D[0 .. 1]: d hhat / d tn below, while these products are queued)
        JT_LAUNCH(k_edge_to_node, (long)Nn * D, s, tp, D, (const float*)dpre, dhr, dhc, 0, (const float*)nullptr, 0, 0);
        c.lin_dw(dhr, D, Nn, D, b.h[l + 1], D, D, dWin, ldw);
        c.lin_dw(dhc, D, Nn, D, b.h[l + 1], D, D, dWin + D, ldw);
        c.lin_dw(dpre, D, R, D, b.e[l + 1], De, De, dWin + 2 * D, ldw, c.g(ix.eq_in.b));
        c.lin_dw(dpre, D, R, D, k.G, De, De, dWin + 2 * D + De, ldw);
        c.lin_dx(dhr, D, Nn, D, Win, ldw, D, dh, D, 1);
        c.lin_dx(dhc, D, Nn, D, Win + D, ldw, D, dh, D, 1);
        if (!t.fused_bwd) {
            c.lin_dx(dpre, D, R, D, Win + 2 * D, ldw, De, de, De, 1);
            c.lin_dx(dpre, D, R, D, Win + 2 * D + De, ldw, De, dG, De, 0);
        }
        // ---- edge FFN, LayerNorm2 + modulate, gated residual (phase D)
        float *dten = b.tE_De[1], *tE = b.tE_rD;
        if (t.fused_bwd) {
            // chain B': dropout / gate backward, ff_linear4^T, SiLU' x dropout, ff_linear3^T, LayerNorm2 + modulate backward — one kernel; it leaves
            // dropout(f4) in dten (d g2 sums), d f4 and d hidden for the weight-gradient products, d en for the modulation sums, and writes de_prev
            float *df4 = b.tE_De2[0], *den = b.tE_De2[1];
            fused_bwd_b(s, fd, ft, fp, fpk, de, k.f4, k.f3, k.xh_en, k.rs_en, k.emod, c.drop(p_drop, seed, l, SITE_A3), c.drop(p_drop, seed, l, SITE_F4),
                        dten, df4, tE, den, de_prev);
            c.seg(De, tp.edge_off, de, dten, demod, Mt, 5 * De);                                     // d eg2
            c.lin_dw(df4, De, R, De, k.a3, r * De, r * De, c.g(ix.ff4.w), r * De, c.g(ix.ff4.b));
            c.lin_dw(tE, r * De, R, r * De, k.en, De, De, c.g(ix.ff3.w), De, c.g(ix.ff3.b));
            c.seg2_edge(De, den, k.xh_en, demod, Mt, 3 * De, 4 * De);
        } else {
            JT_LAUNCH(k_drop, (long)R * De, s, (long)R * De, (const float*)k.f4, dten, c.drop(p_drop, seed, l, SITE_F4));
            c.seg(De, tp.edge_off, de, dten, demod, Mt, 5 * De);                                         // d eg2
            JT_LAUNCH(k_gate_bwd, (long)R * De, s, (long)R, De, (const float*)de, tp.edge_mol, (const float*)k.emod, 6 * De, 5 * De, dten, 0);
            JT_LAUNCH(k_drop, (long)R * De, s, (long)R * De, (const float*)dten, dten, c.drop(p_drop, seed, l, SITE_F4));
            c.lin_dw(dten, De, R, De, k.a3, r * De, r * De, c.g(ix.ff4.w), r * De, c.g(ix.ff4.b));
            c.lin_dx(dten, De, R, De, c.p(ix.ff4.w), r * De, r * De, tE, r * De, 0);
            c.silu_bwd((long)R * r * De, k.f3, tE, tE, c.drop(p_drop, seed, l, SITE_A3));
            c.lin_dw(tE, r * De, R, r * De, k.en, De, De, c.g(ix.ff3.w), De, c.g(ix.ff3.b));
            c.lin_dx(tE, r * De, R, r * De, c.p(ix.ff3.w), De, De, de, De, 1);                          // de is now d en
            c.ln_mod_bwd(R, De, de, k.xh_en, k.rs_en, tp.edge_mol, tp.edge_off, k.emod, 6 * De, 3 * De, 4 * De, demod, Mt, de_prev, 0);   // de_prev = d x1e = d e[l] (residual)
        }
        // d eg1 = sum over the molecule of d x1e * ehat, ehat = node2edge_lin(h_a) + node2edge_lin(h_c) + bias formed on the fly
        c.seg_edge_ehat(De, de_prev, k.n2e, c.p(ix.n2e.b), demod, Mt, 2 * De);
        // d ehat = g1 d x1e reaches node2edge_lin through both atoms of an edge: d n2e[i] = g1 (sum_c d x1e[(i, c)] + sum_a d x1e[(a, i)]) — the
        // gate is per molecule, so it multiplies the node sums (no [R, De] array of gated rows).  The bias saw every edge once, the node
        // sums see it twice: d bias = sum_edges d ehat = 0.5 sum_nodes d n2e, which rides on the weight-gradient product as its bias
        // gradient and is halved (exactly) for all blocks at the end of the backward.
        float* dn2e = b.tN_De;
        JT_LAUNCH(k_edge_to_node, (long)Nn * De, s, tp, De, (const float*)de_prev, dn2e, dn2e, 0, (const float*)k.emod, 6 * De, 2 * De);
        c.lin_dw(dn2e, De, Nn, De, k.hhat, D, D, c.g(ix.n2e.w), D, c.g(ix.n2e.b));
        float* dhhat = b.tN_D[0];
        c.lin_dx(dn2e, De, Nn, De, c.p(ix.n2e.w), D, D, dhhat, D, 0);
        // ---- node FFN, LayerNorm2 + modulate, gated residual
        float *dtn = b.tN_D[1], *tNr = b.tN_rD;
        c.seg_node_drop(D, dh, k.f2, c.drop(p_drop, seed, l, SITE_F2), dnmod, Mt, 5 * D);             // d ng2 = sum dh dropout(f2)
        JT_LAUNCH(k_gate_drop_bwd, (long)Nn * D, s, (long)Nn, D, (const float*)dh, tp.node_mol, (const float*)k.nmod, 6 * D, 5 * D, dtn,
                           c.drop(p_drop, seed, l, SITE_F2));                                        // d f2 = (g2 dh) mask
        c.lin_dw(dtn, D, Nn, D, k.a1, r * D, r * D, c.g(ix.ff2.w), r * D, c.g(ix.ff2.b));
        c.lin_dx(dtn, D, Nn, D, c.p(ix.ff2.w), r * D, r * D, tNr, r * D, 0);
        c.silu_bwd((long)Nn * r * D, k.f1, tNr, tNr, c.drop(p_drop, seed, l, SITE_A1));
        c.lin_dw(tNr, r * D, Nn, r * D, k.hn, D, D, c.g(ix.ff1.w), D, c.g(ix.ff1.b));
        c.lin_dx(tNr, r * D, Nn, r * D, c.p(ix.ff1.w), D, D, dh, D, 1);                              // dh is now d hn
        c.ln_mod_bwd(Nn, D, dh, k.xh_hn, k.rs_hn, tp.node_mol, tp.node_off, k.nmod, 6 * D, 3 * D, 4 * D, dnmod, Mt, dh_prev, 0);      // dh_prev = d x1n = d h[l] (residual)
        c.seg(D, tp.node_off, dh_prev, k.hhat, dnmod, Mt, 2 * D);                                    // d ng1
        JT_LAUNCH(k_gate_bwd, (long)Nn * D, s, (long)Nn, D, (const float*)dh_prev, tp.node_mol, (const float*)k.nmod, 6 * D, 2 * D, dhhat, 1);
        // ---- attention backwards
        c.flush_dw();                                        // d tn (tN_D[1]) and d c0 (tE_D[0]) are d v and d t1 from here on
        const float isc = 1.f / sqrtf((float)t.C);
        const Drop da = c.drop(0.f, seed, l, SITE_ALPHA);                    // p = 0: see SITE_ALPHA
        float *dv = b.tN_D[1], *dt1 = b.tE_D[0], *dS = b.tE_H, *dq = b.tN_QK[0], *dk = b.tN_QK[1], *dt0 = b.tE_QK;
        if (t.fused_attn) {
            // target side (d alpha, softmax backward, d t1, d q, d t0) and source side (d v, d k): two launches, bit-identical to the six below
            const int F3a = 2 * QK + D;
            const AttnTopo at{Nn, t.N, tp.node_mol, tp.nn, tp.node_off, tp.edge_off, t.fused_attn == 2 ? 1 : 0, F3a, F3a, F3a, F3a};
            // (d q | d k | d v land side by side in b.dqkv, where the merged projection's backward reads them)
            fused_attn_bwd(s, at, D, H, t.XH, t.SC, isc, dhhat, k.qkv, k.qkv + QK, k.qkv + 2 * QK, k.t0, k.t1, k.alpha, dS, dt1, dt0, b.dqkv, b.dqkv + QK,
                           b.dqkv + 2 * QK);
        } else {
            JT_LAUNCH(k_attn_bwd_v, (long)Nn * D, s, tp, D, H, (const float*)dhhat, (const float*)k.t1, (const float*)k.alpha, da, dv);
            JT_LAUNCH(k_attn_bwd_t1, (long)R * D, s, tp, D, H, (const float*)dhhat, (const float*)k.v, (const float*)k.t1, (const float*)k.alpha, da, dt1);
            JT_LAUNCH(k_attn_bwd_alpha, (long)R * H, s, tp, D, H, (const float*)dhhat, (const float*)k.v, (const float*)k.t1, da, dS);
            JT_LAUNCH(k_attn_bwd_softmax, (long)Nn * H, s, tp, H, (const float*)k.alpha, dS);
            JT_LAUNCH(k_attn_bwd_qk, (long)Nn * QK, s, tp, H, t.XH, t.SC, isc, (const float*)dS, (const float*)k.q, (const float*)k.k, (const float*)k.t0, dq, dk);
            JT_LAUNCH(k_attn_bwd_t0, (long)R * QK, s, tp, H, t.XH, t.SC, isc, (const float*)dS, (const float*)k.q, (const float*)k.k, (const float*)k.t0, dt0);
        }
        float* det = b.tE_De[1];
        c.lin_dw(dt1, D, R, D, k.et, De, De, c.g(ix.le1), De);
        c.lin_dw(dt0, QK, R, QK, k.et, De, De, c.g(ix.le0), De);
        float* de1 = t.fused_bwd ? b.tE_De2[2] : b.tE_De[1];
        if (t.fused_bwd) {
            // chain A': lin_edge1^T, lin_edge0^T, LayerNorm1 + modulate backward, edge_emb^T — one kernel; it leaves det for the modulation
            // sums and de1 for the weight-gradient products, and adds into dG and de_prev
            fused_bwd_a(s, fd, ft, fp, fpk, dt1, dt0, k.xh_e1, k.rs_e1, k.emod, det, de1, dG, de_prev);
        } else {
            c.lin_dx(dt1, D, R, D, c.p(ix.le1), De, De, det, De, 0);
            c.lin_dx(dt0, QK, R, QK, c.p(ix.le0), De, De, det, De, 1);
        }
        float* dht = b.tN_D[0];
        {   // d q | d k | d v side by side: one weight-gradient and one input-gradient product on the gathered weights of the forward
            const int F3 = 2 * QK + D;
            ModTable M;
            M.n = 3;
            float* srcs[3] = {dq, dk, dv};
            for (int j = 0; j < 3; ++j) { M.w[j] = nullptr; M.bias[j] = nullptr; M.out[j] = srcs[j]; M.F[j] = j < 2 ? QK : D; M.col[j] = j * QK; }
            if (!t.fused_attn) hipLaunchKernelGGL(k_mod_gather_cols, dim3((unsigned)(((long)Nn * D + 255) / 256), 3u), dim3(256), 0, s, M, Nn, F3, b.dqkv);
            c.lin_dw(b.dqkv, F3, Nn, F3, k.ht, D, D, b.dWqkv + (size_t)l * F3 * D, D, b.dbqkv + (size_t)l * F3);
            c.lin_dx(b.dqkv, F3, Nn, F3, b.Wqkv + (size_t)l * F3 * D, D, D, dht, D, 0);
        }
        // ---- the two modulated LayerNorms at the top of the block, edge_emb([G, e])
        if (t.fused_bwd) {
            c.seg2_edge(De, det, k.xh_e1, demod, Mt, 0, De);
        } else {
            c.ln_mod_bwd(R, De, det, k.xh_e1, k.rs_e1, tp.edge_mol, tp.edge_off, k.emod, 6 * De, 0, De, demod, Mt, de1, 0);      // in place: dx written after its own row was read
        }
        c.lin_dw(de1, De, R, De, k.G, De, De, c.g(ix.edge_emb.w), 2 * De, c.g(ix.edge_emb.b));
        c.lin_dw(de1, De, R, De, b.e[l], De, De, c.g(ix.edge_emb.w) + De, 2 * De);
        if (!t.fused_bwd) {
            c.lin_dx(de1, De, R, De, c.p(ix.edge_emb.w), 2 * De, De, dG, De, 1);
            c.lin_dx(de1, De, R, De, c.p(ix.edge_emb.w) + De, 2 * De, De, de_prev, De, 1);
        }
        c.ln_mod_bwd(Nn, D, dht, k.xh_h, k.rs_h, tp.node_mol, tp.node_off, k.nmod, 6 * D, 0, D, dnmod, Mt, dh_prev, 1);
        // ---- Gaussian basis and distances -> positions of the block input
        c.flush_dw();                                        // gbf_bwd's partial sums live in tE_QK (d t0), and the next block reuses everything
        float* dd2 = b.tRow[0];
        gbf_bwd(c, R, k.d2, k.gm, ix.gbf_means, ix.gbf_stds, dgm, Mt, dG, De, 0, dd2);
        JT_LAUNCH(k_dist2_bwd, (long)R * 3, s, tp, (const float*)b.pos[l], (const float*)dd2, ddiff);
        JT_LAUNCH(k_diff_to_node, (long)Nn * 3, s, tp, (const float*)ddiff, (const float*)dxp, dpos_prev);
        c.flush_fin();                                       // this block's queued reduction stages: two launches
        std::swap(dh, dh_prev); std::swap(de, de_prev); std::swap(dpos, dpos_prev);
    }
    // embeddings
    c.copy2d(Nn, D, dah, t.catn, 0, dh, D, 0, 1);
    c.copy2d(R, De, deh, t.cate, 0, de, De, 0, 1);
    c.lin_dw(dh, D, Nn, D, b.nin, 2 * nd, 2 * nd, c.g(t.node_emb.w), 2 * nd, c.g(t.node_emb.b));
    c.lin_dw(de, De, R, De, b.ein, ldin, ldin, c.g(t.edge_emb.w), ldin, c.g(t.edge_emb.b));
    c.flush_dw();
    float* dG0 = b.tE_De[0];
    c.lin_dx(de, De, R, De, c.p(t.edge_emb.w) + 2 * ch, ldin, De, dG0, De, 0);
    // the top-level Gaussian layer saw the self-conditioning distances, or nothing at all on a first step (flag [3] == 0: G0 = 0)
    JT_LAUNCH(k_scale_if_zero, (long)R * De, s, (long)R * De, dG0, (const int*)(b.flags + 3));
    gbf_bwd(c, R, b.d2c, b.gm_top, t.gbf_means, t.gbf_stds, b.dmods_all, Mt, dG0, De, 0, nullptr);
    c.flush_fin();
    {   // the gathered q | k | v weight gradients of every block back to their tensors
        const int F3 = 2 * QK + D;
        ModGradTable M;
        M.n = 3 * L;
        for (int l = 0; l < L; ++l) {
            const BlkIx& ix = t.blk[l];
            const Lin ls[3] = {ix.query, ix.key, ix.value};
            for (int j = 0; j < 3; ++j) { const int i = 3 * l + j; M.gw[i] = c.g(ls[j].w); M.gb[i] = c.g(ls[j].b); M.F[i] = j < 2 ? QK : D; M.col[i] = l * F3 + j * QK; }
        }
        hipLaunchKernelGGL(k_mod_scatter_grads, dim3((unsigned)(((long)D * D + 255) / 256), (unsigned)M.n), dim3(256), 0, s, M, D, (const float*)b.dWqkv, (const float*)b.dbqkv);
    }
    // every modulation projection at once, then the time embedding
    mod_bwd_all(c);
    c.silu_bwd((long)B * T, b.temb, b.dtau, b.dtemb, nod);
    if (t.cc > 0) {
        const int cD = t.cc * D;
        c.lin_dw(b.dtemb, T, B, T, b.cc2, cD, cD, c.g(t.cond_lin.w), cD, c.g(t.cond_lin.b));
        c.lin_dx(b.dtemb, T, B, T, c.p(t.cond_lin.w), cD, cD, b.tB_cD[0], cD, 0);
        c.lin_dw(b.tB_cD[0], D, B * t.cc, D, b.cc0a, D, D, c.g(t.cond2.w), D, c.g(t.cond2.b));
        c.lin_dx(b.tB_cD[0], D, B * t.cc, D, c.p(t.cond2.w), D, D, b.tB_cD[1], D, 0);
        JT_LAUNCH(k_gelu_bwd, (long)B * cD, s, (long)B * cD, (const float*)b.cc0pre, (const float*)b.tB_cD[1], b.tB_cD[1]);
        c.lin_dw(b.tB_cD[1], D, B * t.cc, D, b.ctx, 1, 1, c.g(t.cond0.w), 1, c.g(t.cond0.b));
    }
    c.lin_dw(b.dtemb, T, B, T, b.t1a, T, T, c.g(t.time3.w), T, c.g(t.time3.b));
    c.lin_dx(b.dtemb, T, B, T, c.p(t.time3.w), T, T, b.tB_T[0], T, 0);
    JT_LAUNCH(k_gelu_bwd, (long)B * T, s, (long)B * T, (const float*)b.t1pre, (const float*)b.tB_T[0], b.tB_T[0]);
    c.lin_dw(b.tB_T[0], T, B, T, b.feat, F17, F17, c.g(t.time1.w), F17, c.g(t.time1.b));
    c.lin_dx(b.tB_T[0], T, B, T, c.p(t.time1.w), F17, F17, b.tB_T[1], F17, 0);
    JT_LAUNCH(k_time_feat_bwd, t.half, s, B, t.half, nl, c.p(t.time_w), (const float*)b.tB_T[1], c.g(t.time_w));
    c.flush_dw();                                            // the time / context MLPs' products: their operands are final where they are queued
    c.flush_fin();
    b.defer_fin = false;
    for (int l0 = 0; l0 < L; l0 += 16) {                     // node2edge bias gradients: the node sums counted every edge twice (see the block loop)
        ScaleTable S;
        S.n_arrays = L - l0 < 16 ? L - l0 : 16;
        for (int i = 0; i < S.n_arrays; ++i) S.x[i] = c.g(t.blk[l0 + i].n2e.b);
        JT_LAUNCH(k_scale_arrays, (long)S.n_arrays * De, s, S, De, 0.5f);
    }
}

Topo make_topo(const jodo_train& t, const void* desc_dev) {
    const int* d = static_cast<const int*>(desc_dev);
    Topo tp;
    tp.B = t.B; tp.Nn = t.Nn; tp.R = t.R; tp.N = t.N;
    tp.node_off = d + t.o_node_off; tp.edge_off = d + t.o_edge_off; tp.nn = d + t.o_nn; tp.node_mol = d + t.o_node_mol;
    tp.edge_mol = d + t.o_edge_mol; tp.edge_a = d + t.o_edge_a; tp.edge_c = d + t.o_edge_c;
    tp.NC = t.NC; tp.ec_off = d + t.o_ec_off; tp.ec_mol_off = d + t.o_ec_mol_off;
    return tp;
}

}  // namespace

extern "C" {

int jodo_train_create(const jodo_cfg* cfg, int B, int N, const int32_t* n_nodes, const jodo_tensor* params, int n_params, jodo_train** out) {
    if (!cfg || !n_nodes || !params || !out || B <= 0 || N <= 0) return jodo_set_error(JODO_ERR_ARG, "jodo_train_create: null / non-positive argument");
    if (cfg->nf % cfg->n_heads || cfg->nf % 4 || cfg->n_heads <= cfg->n_extra || cfg->n_extra != 2)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "jodo_train_create: nf %d / n_heads %d / n_extra_heads %d", cfg->nf, cfg->n_heads, cfg->n_extra);
    // k_row_part / k_ln_bwd_part (train_ops.h) cut a LayerNorm row of De = nf / 4 features into eight equal parts, and the MLP heads
    // halve D and De: a width that is not a multiple of 32 would silently drop trailing features from the statistics
    if (cfg->n_layers < 1 || cfg->n_layers > 16)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "jodo_train_create: n_layers %d (1 .. 16: the batched modulation tables hold 4 x 16 + 1 projections)", cfg->n_layers);
    if (cfg->nf % 32)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "jodo_train_create: nf %d is not a multiple of 32 (LayerNorm rows of nf / 4 features are reduced in eight equal parts)", cfg->nf);
    jodo_train* t = new jodo_train();
    t->cfg = *cfg; t->B = B; t->N = N;
    t->D = cfg->nf; t->De = cfg->nf / 4; t->T = cfg->nf * 4; t->L = cfg->n_layers; t->H = cfg->n_heads; t->XH = cfg->n_extra;
    t->C = t->D / t->H; t->SC = (t->H * t->C) / (t->H - t->XH); t->QK = (t->H - t->XH) * t->SC; t->r = cfg->mlp_ratio; t->nd = cfg->in_node_dim;
    t->ch = cfg->edge_ch; t->cc = cfg->cond_ch; t->cn = (2 * t->D) / t->L; t->ce = (2 * t->De) / t->L; t->catn = t->D + t->L * t->cn;
    t->cate = t->De + t->L * t->ce;
    long Nn = 0, R = 0;
    for (int b = 0; b < B; ++b) {
        if (n_nodes[b] < 1 || n_nodes[b] > N) { delete t; return jodo_set_error(JODO_ERR_ARG, "jodo_train_create: n_nodes[%d] = %d outside 1..%d", b, n_nodes[b], N); }
        Nn += n_nodes[b]; R += (long)n_nodes[b] * n_nodes[b];
    }
    if (R > (1L << 30)) { delete t; return jodo_set_error(JODO_ERR_UNSUPPORTED, "jodo_train_create: %ld edge rows", R); }
    t->Nn = (int)Nn; t->R = (int)R;
    std::vector<int>& tb = t->tables;
    t->o_node_off = 0; t->o_edge_off = t->o_node_off + B + 1; t->o_nn = t->o_edge_off + B + 1; t->o_node_mol = t->o_nn + B;
    t->o_edge_mol = t->o_node_mol + Nn; t->o_edge_a = t->o_edge_mol + R; t->o_edge_c = t->o_edge_a + R;
    // chunks of a molecule's edge rows: at most 64 per molecule, at least 32 rows each
    std::vector<int> ec_off, ec_mol_off;
    {
        int eo2 = 0;
        for (int b = 0; b < B; ++b) {
            const int n2 = n_nodes[b] * n_nodes[b];
            const int ch = std::max(32, (n2 + 63) / 64);
            ec_mol_off.push_back((int)ec_off.size());
            for (int r0 = 0; r0 < n2; r0 += ch) ec_off.push_back(eo2 + r0);
            eo2 += n2;
        }
        ec_mol_off.push_back((int)ec_off.size());
        ec_off.push_back(eo2);
    }
    t->NC = (int)ec_off.size() - 1;
    t->o_ec_off = t->o_edge_c + R; t->o_ec_mol_off = t->o_ec_off + ec_off.size();
    tb.assign(t->o_ec_mol_off + ec_mol_off.size(), 0);
    std::copy(ec_off.begin(), ec_off.end(), tb.begin() + t->o_ec_off);
    std::copy(ec_mol_off.begin(), ec_mol_off.end(), tb.begin() + t->o_ec_mol_off);
    int no = 0, eo = 0;
    for (int b = 0; b < B; ++b) {
        const int n = n_nodes[b];
        tb[t->o_node_off + b] = no; tb[t->o_edge_off + b] = eo; tb[t->o_nn + b] = n;
        for (int i = 0; i < n; ++i) tb[t->o_node_mol + no + i] = b;
        for (int a = 0; a < n; ++a)
            for (int c = 0; c < n; ++c) {
                tb[t->o_edge_mol + eo + a * n + c] = b; tb[t->o_edge_a + eo + a * n + c] = no + a; tb[t->o_edge_c + eo + a * n + c] = no + c;
            }
        no += n; eo += n * n;
    }
    tb[t->o_node_off + B] = no; tb[t->o_edge_off + B] = eo;
    // parameters by name
    std::unordered_map<std::string, int> ix;
    t->n_params = n_params; t->numel.resize(n_params);
    for (int i = 0; i < n_params; ++i) {
        std::string nm = params[i].name ? params[i].name : "";
        if (nm.rfind("module.", 0) == 0) nm = nm.substr(7);
        size_t ne = 1;
        for (int d = 0; d < params[i].ndim; ++d) ne *= (size_t)params[i].shape[d];
        t->numel[i] = ne; ix[nm] = i;
    }
    std::string missing;
    auto find = [&](const std::string& nm, size_t numel) -> int {
        auto it = ix.find(nm);
        if (it == ix.end() || t->numel[it->second] != numel) { if (missing.empty()) missing = nm; return -1; }
        return it->second;
    };
    auto lin = [&](const std::string& nm, size_t out_f, size_t in_f) { Lin l; l.w = find(nm + ".weight", out_f * in_f); l.b = find(nm + ".bias", out_f); return l; };
    const size_t D = t->D, De = t->De, T = t->T, QK = t->QK, r = t->r, nd = t->nd, ch = t->ch;
    t->node_emb = lin("node_emb", D, 2 * nd); t->edge_emb = lin("edge_emb", De, 2 * ch + De);
    t->gbf_means = find("dist_layer.means.weight", De - 1); t->gbf_stds = find("dist_layer.stds.weight", De - 1); t->gbf_time = lin("dist_layer.time_mlp.1", 2, T);
    t->np0 = lin("node_pred_mlp.0", D, t->catn); t->np2 = lin("node_pred_mlp.2", D / 2, D); t->np4 = lin("node_pred_mlp.4", nd, D / 2);
    t->et0 = lin("edge_type_mlp.0", De, t->cate); t->et2 = lin("edge_type_mlp.2", De / 2, De); t->et4 = lin("edge_type_mlp.4", ch - 1, De / 2);
    t->ee0 = lin("edge_exist_mlp.0", De, t->cate); t->ee2 = lin("edge_exist_mlp.2", De / 2, De); t->ee4 = lin("edge_exist_mlp.4", 1, De / 2);
    {
        auto it = ix.find("time_mlp.0.weights");
        if (it == ix.end() || t->numel[it->second] < 1) { if (missing.empty()) missing = "time_mlp.0.weights"; t->time_w = -1; t->half = 8; }
        else { t->time_w = it->second; t->half = (int)t->numel[it->second]; }
    }
    t->time1 = lin("time_mlp.1", T, 2 * t->half + 1); t->time3 = lin("time_mlp.3", T, T);
    if (t->cc > 0) { t->cond0 = lin("cond_mlp.0", D, 1); t->cond2 = lin("cond_mlp.2", D, D); t->cond_lin = lin("cond_lin", T, (size_t)t->cc * D); }
    t->blk.resize(t->L);
    for (int l = 0; l < t->L; ++l) {
        const std::string p = "e_block_" + std::to_string(l) + ".";
        BlkIx& k = t->blk[l];
        k.edge_emb = lin(p + "edge_emb", De, 2 * De); k.n2e = lin(p + "node2edge_lin", De, D);
        k.key = lin(p + "attn_mpnn.lin_key", QK, D); k.query = lin(p + "attn_mpnn.lin_query", QK, D); k.value = lin(p + "attn_mpnn.lin_value", D, D);
        k.le0 = find(p + "attn_mpnn.lin_edge0.weight", QK * De); k.le1 = find(p + "attn_mpnn.lin_edge1.weight", D * De);
        k.ff1 = lin(p + "ff_linear1", r * D, D); k.ff2 = lin(p + "ff_linear2", D, r * D); k.ff3 = lin(p + "ff_linear3", r * De, De); k.ff4 = lin(p + "ff_linear4", De, r * De);
        k.eq_scale = find(p + "equi_update.coord_norm.scale", 1); k.eq_time = lin(p + "equi_update.time_mlp.1", 2 * D, T);
        k.eq_in = lin(p + "equi_update.input_lin", D, 2 * D + 2 * De); k.eq_c0 = lin(p + "equi_update.coord_mlp.0", D, D);
        k.eq_c2 = find(p + "equi_update.coord_mlp.2.weight", 3 * D);
        k.node_time = lin(p + "node_time_mlp.1", 6 * D, T); k.edge_time = lin(p + "edge_time_mlp.1", 6 * De, T);
        k.gbf_means = find(p + "dist_layer.means.weight", De - 1); k.gbf_stds = find(p + "dist_layer.stds.weight", De - 1);
        k.gbf_time = lin(p + "dist_layer.time_mlp.1", 2, T);
        k.node_ro = lin("node_" + std::to_string(l), t->cn, D); k.edge_ro = lin("edge_" + std::to_string(l), t->ce, De);
    }
    if (!missing.empty()) { delete t; return jodo_set_error(JODO_ERR_ARG, "jodo_train_create: parameter '%s' missing or mis-sized", missing.c_str()); }
    {
        const FusedDims fd{t->D, t->De, t->r, t->QK, t->ce, t->L};
        t->fused = fused_available(fd) ? 1 : 0;
        t->fused_bwd = t->fused;
        t->save_activations = 1;
        t->group_dw = 1;
        t->gbf_chunk = 1;
        t->fused_attn = t->fused && fused_attention_available(t->D, t->H, t->N) ? 1 : 0;
        t->Mtot = 2 + t->L * (6 * t->D + 6 * t->De + 2 * t->D + 2);
    }
    Arena a{nullptr, 0}; Bufs bufs;
    layout(*t, a, bufs);
    t->ws_bytes = a.off;
    *out = t;
    return JODO_OK;
}

void jodo_train_destroy(jodo_train* t) { delete t; }
size_t jodo_train_desc_bytes(const jodo_train* t) { return t ? t->tables.size() * sizeof(int) : 0; }
size_t jodo_train_workspace_bytes(const jodo_train* t) { return t ? t->ws_bytes : 0; }
// option 0: fused per-edge forward chains (train_fused.hip): 1 (default where the width is supported) / 0 (op-by-op, the reference form)
// option 1: the same for the input-gradient side of the backward
// option 2: 1 (default) every forward keeps what a backward needs; 0: the following forwards will not be differentiated (the no-grad
//           self-conditioning forward of a training step): the fused chains skip the stores only a backward reads
// option 5: the Gaussian layer's backward (d x' per row + chunk partials of d means / d stds) as one wave-per-chunk pass (train_fused.hip
//           k_gbf_bwd_chunk): 1 (default) for batches of at least 4 096 chunks of 32 edge rows, 0 never, 2 always (tests)
// option 4: 1 (default where built) attention forward in one launch and backward in two (train_fused.hip k_attn_fwd / k_attn_bwd_tgt / _src:
//           one to four waves per atom, the forward's sums in the op-by-op kernels' order — bit-identical); 0: scores | softmax | messages and six
//           backward kernels; 2: the one-wave-per-atom form that batches above 16 k atoms take (tests)
// option 3: 1 (default) the backward's weight-gradient products run in grouped launches (train_gemm.hip gemm_dw_group); 0: one launch
//           (+ one split-K sum) each — same plans, same arithmetic, bit-identical gradients
int jodo_train_set_option(jodo_train* t, int option, int value) {
    if (!t) return jodo_set_error(JODO_ERR_ARG, "jodo_train_set_option: null handle");
    if (option < 0 || option > 5 || (value != 0 && value != 1 && !(option >= 4 && value == 2))) return jodo_set_error(JODO_ERR_ARG, "jodo_train_set_option: option %d value %d", option, value);
    if (option == 2) { t->save_activations = value; return JODO_OK; }
    if (option == 3) { t->group_dw = value; return JODO_OK; }
    if (option == 5) { t->gbf_chunk = value; return JODO_OK; }
    if (option == 4) {
        if (value && !(fused_available(FusedDims{t->D, t->De, t->r, t->QK, t->ce, t->L}) && fused_attention_available(t->D, t->H, t->N)))
            return jodo_set_error(JODO_ERR_UNSUPPORTED, "jodo_train_set_option: the fused attention kernels are not built for this shape");
        t->fused_attn = value;
        return JODO_OK;
    }
    const FusedDims fd{t->D, t->De, t->r, t->QK, t->ce, t->L};
    if (value && !fused_available(fd)) return jodo_set_error(JODO_ERR_UNSUPPORTED, "jodo_train_set_option: fused chains are not built for this shape");
    if (option == 0) t->fused = value; else t->fused_bwd = value;
    return JODO_OK;
}
// tests: where a kept activation lives in the workspace.  what 0 = hhat (attention output [Nn, D]), 1 = alpha (softmax weights
// [R, H]) of block `layer`
int jodo_train_debug_locate(const jodo_train* t, int what, int layer, size_t* byte_offset, size_t* count) {
    if (!t || !byte_offset || !count) return jodo_set_error(JODO_ERR_ARG, "jodo_train_debug_locate: null argument");
    if (layer < 0 || layer >= t->L) return jodo_set_error(JODO_ERR_ARG, "jodo_train_debug_locate: layer %d of %d", layer, t->L);
    Arena a{reinterpret_cast<char*>(256), 0};              // any non-null base: only the differences are used
    Bufs b;
    layout(*t, a, b);
    const float* ptr = nullptr;
    size_t n = 0;
    if (what == 0) { ptr = b.blk[layer].hhat; n = (size_t)t->Nn * t->D; }
    else if (what == 1) { ptr = b.blk[layer].alpha; n = (size_t)t->R * t->H; }
    else return jodo_set_error(JODO_ERR_ARG, "jodo_train_debug_locate: unknown selector %d", what);
    *byte_offset = (size_t)(reinterpret_cast<const char*>(ptr) - reinterpret_cast<const char*>(256));
    *count = n;
    return JODO_OK;
}
// the host image of the index tables (jodo_train_desc_bytes() bytes, alive as long as the handle): a caller that stages it through
// pinned memory uploads without the stream synchronisation of jodo_train_upload (jodo_amd/train.py does)
const void* jodo_train_desc_host(const jodo_train* t) { return t ? t->tables.data() : nullptr; }
int jodo_train_upload(jodo_train* t, void* desc_dev, void* stream) {
    if (!t || !desc_dev) return jodo_set_error(JODO_ERR_ARG, "jodo_train_upload: null argument");
    (void)hipMemcpyAsync(desc_dev, t->tables.data(), t->tables.size() * sizeof(int), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));          // the host table may be freed with the handle
    return jodo_check_launch("jodo_train_upload");
}

int jodo_train_forward(jodo_train* t, const void* desc_dev, const float* const* params_dev, int n_params, const float* xh, const float* edge_x,
                       const float* cond_x, const float* cond_edge_x, const float* noise_level, const float* context, float dropout_p, uint64_t seed,
                       float* out_xh, float* out_edge, int32_t* flags_out, void* workspace, void* stream) {
    if (!t || !desc_dev || !params_dev || !xh || !edge_x || !noise_level || !out_xh || !out_edge || !workspace)
        return jodo_set_error(JODO_ERR_ARG, "jodo_train_forward: null argument");
    if (n_params != t->n_params) return jodo_set_error(JODO_ERR_ARG, "jodo_train_forward: %d parameters, handle was created with %d", n_params, t->n_params);
    if ((cond_x == nullptr) != (cond_edge_x == nullptr)) return jodo_set_error(JODO_ERR_ARG, "jodo_train_forward: cond_x and cond_edge_x go together");
    if (t->cc > 0 && !context) return jodo_set_error(JODO_ERR_ARG, "jodo_train_forward: the conditional model needs context");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return jodo_set_error(JODO_ERR_ARG, "jodo_train_forward: dropout %g", dropout_p);
    Arena a{static_cast<char*>(workspace), 0}; Bufs bufs;
    layout(*t, a, bufs);
    Ctx c{*t, make_topo(*t, desc_dev), params_dev, nullptr, bufs, static_cast<hipStream_t>(stream)};
    forward(c, xh, edge_x, cond_x, cond_edge_x, noise_level, context, dropout_p, seed, out_xh, out_edge);
    if (flags_out) (void)hipMemcpyAsync(flags_out, bufs.flags, 8 * sizeof(int), hipMemcpyDeviceToDevice, c.s);
    return jodo_check_launch("jodo_train_forward");
}

int jodo_train_backward(jodo_train* t, const void* desc_dev, const float* const* params_dev, float* const* grads_dev, int n_params,
                        const float* noise_level, const float* d_out_xh, const float* d_out_edge, float dropout_p, uint64_t seed, void* workspace,
                        void* stream) {
    if (!t || !desc_dev || !params_dev || !grads_dev || !noise_level || !d_out_xh || !d_out_edge || !workspace)
        return jodo_set_error(JODO_ERR_ARG, "jodo_train_backward: null argument");
    if (n_params != t->n_params) return jodo_set_error(JODO_ERR_ARG, "jodo_train_backward: %d parameters, handle was created with %d", n_params, t->n_params);
    Arena a{static_cast<char*>(workspace), 0}; Bufs bufs;
    layout(*t, a, bufs);
    Ctx c{*t, make_topo(*t, desc_dev), params_dev, grads_dev, bufs, static_cast<hipStream_t>(stream)};
    backward(c, noise_level, d_out_xh, d_out_edge, dropout_p, seed);
    return jodo_check_launch("jodo_train_backward");
}

int jodo_train_gemm(int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, int acc,
                    float* ws, size_t ws_floats, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return jodo_set_error(JODO_ERR_ARG, "jodo_train_gemm: bad argument");
    gemm(static_cast<hipStream_t>(stream), tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, acc, ws, ws_floats);
    return jodo_check_launch("jodo_train_gemm");
}

int jodo_train_gemm_ex(int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                       int act, float* out2, float* dbias, float* ws, size_t ws_floats, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0 || act < 0 || act > 2 || (act == 2 && !out2) || (dbias && (!tA || act)))
        return jodo_set_error(JODO_ERR_ARG, "jodo_train_gemm_ex: bad argument");
    GemmEpi e; e.act = act; e.out2 = out2; e.drop.p = 0.f; e.drop.seed = 0; e.drop.site = 0; e.dbias = dbias;
    gemm(static_cast<hipStream_t>(stream), tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, dbias ? 1 : 0, ws, ws_floats, &e);
    return jodo_check_launch("jodo_train_gemm_ex");
}

}  // extern "C"
