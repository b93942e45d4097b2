// jodo_dgt_forward: launch sequence of one score-network evaluation (see include/jodo_hip.h).
// Host code only builds the argument block and enqueues kernels on the caller's stream; no
// synchronisation, no allocation.
#include "dgt_kernels_pre.h"
#include "dgt_kernels_node.h"
#include "dgt_kernels_split_node.h"
#include "dgt_kernels_post.h"
#include "dgt_kernels_wide.h"
#include "jodo_hip_internal.h"
#include "dgt_launch.h"
#include <memory>
#include <utility>

using namespace jd;

namespace {

PlanDev make_plan_dev(const jodo_plan* p, const void* desc_dev) {
    const int* base = static_cast<const int*>(desc_dev);
    PlanDev d;
    d.node_b = base + p->off_node_b; d.node_i = base + p->off_node_i; d.node_n = base + p->off_node_n;
    d.node_noff = base + p->off_node_noff; d.node_eoff = base + p->off_node_eoff;
    d.orig_n = base + p->off_orig_n; d.orig_noff = base + p->off_orig_noff; d.orig_eoff = base + p->off_orig_eoff;
    d.item_strip = base + p->off_item_strip; d.item_t0 = base + p->off_item_t0; d.item_t1 = base + p->off_item_t1;
    d.item_part = base + p->off_item_part; d.strip_parts = base + p->off_strip_parts;
    d.pitem_strip = base + p->off_pitem_strip; d.pitem_t0 = base + p->off_pitem_t0; d.pitem_t1 = base + p->off_pitem_t1;
    d.n_pitems = p->n_pitems;
    d.ag_node = base + p->off_ag_node; d.ai_group = base + p->off_ai_group; d.ai_t0 = base + p->off_ai_t0; d.ai_t1 = base + p->off_ai_t1;
    d.ai_part = base + p->off_ai_part; d.ai_dir = base + p->off_ai_dir; d.ad_group = base + p->off_ad_group; d.ad_t0 = base + p->off_ad_t0; d.ad_t1 = base + p->off_ad_t1;
    d.ad_part = base + p->off_ad_part; d.ad_big = base + p->off_ad_big; d.anode_parts = base + p->off_anode_parts;
    d.aw_off = base + p->off_aw_off; d.a_persist = p->a_persist;
    d.n_agroups = p->n_agroups; d.n_aitems = p->n_aitems; d.n_aditems = p->n_aditems; d.amax_parts = p->amax_parts;
    d.Nn = p->Nn; d.Nn_pad = p->Nn_pad; d.n_strips = p->n_strips; d.n_items = p->n_items; d.B = p->B; d.N = p->N;
    d.max_parts = p->max_parts; d.rows = p->rows;
    d.ut_rows = base + p->off_ut_rows; d.n_ut_pad = p->n_ut_pad;
    d.gt_sa = base + p->off_gt_sa; d.gt_sc = base + p->off_gt_sc; d.n_gtiles = p->n_gtiles;
    return d;
}

template <typename T>
T* ws_ptr(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

void fill_ws(KArgs& A, const jodo_plan* p, void* ws) {
    const WsLayout& w = p->ws;
    A.hid1 = ws_ptr<float>(ws, w.hid1); A.temb = ws_ptr<float>(ws, w.temb); A.mods = ws_ptr<float>(ws, w.mods);
    A.condh = ws_ptr<float>(ws, w.condh); A.condh2 = ws_ptr<float>(ws, w.condh2);
    A.dpos = ws_ptr<float>(ws, w.dpos); A.cpos = ws_ptr<float>(ws, w.cpos); A.feat = ws_ptr<float>(ws, w.feat);
    A.h = ws_ptr<float>(ws, w.h); A.hhat = ws_ptr<float>(ws, w.hhat); A.astat = ws_ptr<float>(ws, w.astat); A.q = ws_ptr<float>(ws, w.q);
    A.k = ws_ptr<float>(ws, w.k); A.v = ws_ptr<float>(ws, w.v); A.n2e = ws_ptr<float>(ws, w.n2e);
    A.wrow = ws_ptr<float>(ws, w.wrow); A.wcol = ws_ptr<float>(ws, w.wcol); A.ua = ws_ptr<float>(ws, w.ua); A.ub = ws_ptr<float>(ws, w.ub);
    A.rmean = ws_ptr<float>(ws, w.rmean); A.mfold = ws_ptr<float>(ws, w.mfold); A.ffold = ws_ptr<float>(ws, w.ffold); A.ahid = ws_ptr<float>(ws, w.ahid);
    A.apred = ws_ptr<float>(ws, w.apred);
    A.eflag = ws_ptr<int>(ws, w.eflag); A.e = ws_ptr<float>(ws, w.e);
    A.ehid = ws_ptr<float>(ws, w.ehid); A.epred = ws_ptr<float>(ws, w.epred);
    A.e_out = ws_ptr<float>(ws, w.e2);
    A.dposE = ws_ptr<float>(ws, w.dposE); A.gramE = ws_ptr<float>(ws, w.gramE);
    A.pers_n = 0;
    A.wsplit = nullptr; A.wsplit_node = nullptr; A.wsplit_attn = nullptr; A.mfold_s = nullptr; A.ffold_s = nullptr;      // the opt-in split-bf16 kernels: decided per forward (jodo_dgt_forward)
}

int rowgemm(hipStream_t st, const float* X, int64_t ldx, float* Y, int64_t ldy, const float* Wp, const float* bias,
            int rows, int K, int NB, int in_act, int accumulate, const int* uniform_flag, float* Ysilu = nullptr) {
    if (K % 64 != 0) return jodo_set_error(JODO_ERR_ARG, "rowgemm: K=%d not a multiple of 64", K);
    // few rows (the sampling case: one shared time row): one output block per wave so that the launch
    // still has hundreds of waves; many rows: four blocks per wave to reuse the activation chunk
    const int nob = (rows <= 64 || uniform_flag) ? 1 : 4;
    const int gx = uniform_flag ? 1 : 0;
    RowGemmArgs G{X, ldx, Y, ldy, Wp, bias, rows, K, NB, in_act, accumulate, uniform_flag, nob, gx, Ysilu};
    dim3 grid((rows + 31) / 32, (NB + nob - 1) / nob);
    if (gx) grid = dim3(grid.y, grid.x);
    hipLaunchKernelGGL(k_rowgemm, grid, dim3(64), 0, st, G);
    return jodo_check_launch("k_rowgemm");
}

// event bracket for one launch class (no-op unless profiling is enabled on the plan)
struct ProfScope {
    jodo_plan* p; hipStream_t st; int cls; bool on;
    static hipEvent_t get(jodo_plan* p) {
        if (!p->prof_pool.empty()) { hipEvent_t e = (hipEvent_t)p->prof_pool.back(); p->prof_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    ProfScope(jodo_plan* p_, hipStream_t st_, int cls_)
        : p(p_), st(st_), cls(cls_), on(p_->prof_enabled == 1 || (p_->prof_enabled == 2 && cls_ == JODO_PROF_EDGE_UPDATE) || p_->prof_enabled == 16 + cls_) {
        if (on) { hipEvent_t e = get(p); (void)hipEventRecord(e, st); p->prof_ev.push_back(e); }
    }
    ~ProfScope() {
        if (on) { hipEvent_t e = get(p); (void)hipEventRecord(e, st); p->prof_ev.push_back(e); p->prof_cls.push_back(cls); }
    }
};

template <int D, int KQ>
int launch_embed_nodes(hipStream_t st, const KArgs& A) {
    LAUNCH((wide::k_embed_nodes<D, KQ>), A.pd.n_strips, 64, A);
    return JODO_OK;
}
// symmetric inputs (device flag): one evaluation per unordered pair, written to both rows; otherwise every dense row
// merged = 1: pinned symmetric inputs and JODO_OPT_HEADS_MIX — node head and pair edge head share one launch (k_heads_sym)
template <int D, int NBK>
int launch_edge_head(hipStream_t st, const KArgs& A, bool merged) {
    if (merged) {
        LAUNCH((wide::k_heads_sym<D, NBK>), (unsigned)(A.pd.n_strips + A.pd.n_ut_pad / 32), 64, A);
        return JODO_OK;
    }
    LAUNCH((wide::k_node_head<D>), A.pd.n_strips, 64, A);
    if (A.pd.n_ut_pad > 0 && A.pin_sym != 2) LAUNCH((wide::k_edge_head<D, NBK, true>), (unsigned)(A.pd.n_ut_pad / 32), 64, A);
    if (A.pin_sym != 1) LAUNCH((wide::k_edge_head<D, NBK, false>), (unsigned)((A.pd.rows + 31) / 32), 64, A);
    return JODO_OK;
}

// First block only: its q / k / v items and the edge embedding in ONE launch.  The embedding is bound by its HBM writes (edge
// state + head inputs of every pair), the projections by the matrix pipe, and neither depends on the other: the (strip, piece)
// items go first, the embedding's items fill the SIMDs their partial last round leaves idle and overlap their stores with the MFMAs.
__global__ __launch_bounds__(64, 1) void k_pre_embed(KArgs A) {
    const int npre = A.pd.n_strips * 3;
    if ((int)blockIdx.x < npre) node_pre_body<false>(A, (int)blockIdx.x);
    else wide::embed_edges_body<256>(A, (int)blockIdx.x - npre);
}


// Fewer than 1 024 strips (GEOM B = 512, conditional B = 1 250): k_node_post cannot also produce the next block's q / k / v (its
// single round is the critical path), and as launches of their own the 3 n_strips q / k / v items and the 2 n_strips k_node_ab items
// each end in a sparsely filled round (2 130 -> 3 rounds, 1 420 -> 2 rounds at 710 strips).  They are independent — the next block's
// q / k / v need h', k_node_ab needs W_row h' / W_col h', both written by k_node_post — so one launch carries both plus the Gram
// tiles: 3 550 items -> 4 rounds instead of 5.  A.ab1 - A.ab0 items of k_node_ab, A.g1 - A.g0 tiles, A.mix_nw = number of next-pre items.
__global__ __launch_bounds__(64, 1) void k_node_ab_pre(KArgs A) {
    const int npre = A.mix_nw, nab = A.ab1 - A.ab0;
    const int b = (int)blockIdx.x;
    if (b < npre) node_pre_body<true>(A, b);
    else if (b < npre + nab) wide::node_ab_body<256>(A, A.ab0 + b - npre);
    else wide::node_gram_body<256>(A, A.g0 + b - npre - nab);
}

// The same for the width-generic kernel set (nf 128 / 384, nf 256 'wide'), at any strip count: this set has no k_node_post that also
// produces the next block's q / k / v, so the 3 n_strips items of the following block always ride with the 2 n_strips k_node_ab items.
template <int D>
__global__ __launch_bounds__(64, 1) void k_node_ab_pre_w(KArgs A) {
    const int npre = A.mix_nw, nab = A.ab1 - A.ab0;
    const int b = (int)blockIdx.x;
    if (b < npre) wide::node_pre_body<D, true>(A, b);
    else if (b < npre + nab) wide::node_ab_body<D>(A, A.ab0 + b - npre);
    else wide::node_gram_body<D>(A, A.g0 + b - npre - nab);
}
// first block: its q / k / v items and the edge embedding in one launch (as k_pre_embed for the tuned set)
template <int D>
__global__ __launch_bounds__(64, 1) void k_pre_embed_w(KArgs A) {
    const int npre = A.pd.n_strips * 3;
    if ((int)blockIdx.x < npre) wide::node_pre_body<D, false>(A, (int)blockIdx.x);
    else wide::embed_edges_body<D>(A, (int)blockIdx.x - npre);
}

// One launch for a block's remainder strips (k_node_postw role: NW waves per strip, long items first) and for the fine-grained
// per-node items that do not depend on them — k_node_ab items and Gram tiles of the strips the preceding full-round k_node_post
// launch finished, NW per workgroup.  The cooperative remainder workgroups occupy 2 x 385 of 1024 SIMDs for 211 us at QM9
// B = 2500; the small items fill the rest instead of running as a launch of their own afterwards.
template <int R, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_node_mix(KArgs A) {
    __shared__ float4 part[NW * 8 * 4 * 64];
    const int wg = blockIdx.x, wave = threadIdx.x >> 6;
    if (wg < A.mix_nw) { node_postw_body<R, NW>(A, wg + A.strip0, part); return; }
    const int nab = (A.ab1 - A.ab0 + NW - 1) / NW;
    if (wg < A.mix_nw + nab) {
        const int it = A.ab0 + (wg - A.mix_nw) * NW + wave;
        if (it < A.ab1) wide::node_ab_body<256>(A, it);
        return;
    }
    const int t = A.g0 + (wg - A.mix_nw - nab) * NW + wave;
    if (t < A.g1) wide::node_gram_body<256>(A, t);
}

// nf = 256 node kernels (dgt_kernels_node.h): k_node_post with one wave per strip for every full round of 1024 strips (one
// per SIMD); the remainder r — which would otherwise occupy r SIMDs for a whole item while the rest idle — goes to a second
// launch in which a workgroup of 4 (r <= 256) or 2 (r <= 512) waves shares each strip.  Measured on MI355X: 177 strips
// 1.7 -> 0.75 ms/step with 4 waves; all 1409 strips with 2 / 4 waves 3.7 / 4.1 vs 3.6 ms/step with 1.
// with_ab: the pair path runs, so k_node_ab (and with A.rot the Gram tiles) follow — issued here, merged with the remainder launch
// where that pays (JODO_OPT_NODE_MIX)
int launch_node_post_256(jodo_plan* p, hipStream_t st, KArgs& A, bool with_ab, bool next_pre) {
    const DgtDims& d = p->dims;
    const int force = p->opt[JODO_OPT_NODE_POST_WAVES];                      // 0 = automatic
    const bool rem_only = force >= 10;                                       // 12 / 14: automatic split, 2 / 4 waves for the remainder
    const int full = (force && !rem_only) ? (force == 1 ? p->n_strips : 0) : (p->n_strips / 1024) * 1024;
    const int rem = p->n_strips - full;
    const int nw = rem_only ? force - 10 : (force ? force : (rem <= 256 ? 4 : (rem <= 512 ? 2 : 1)));
    // opt-in split-bf16 form (JODO_OPT_SPLIT_BF16; jodo_dgt_forward checked the preconditions): every strip through k_node_post_split, four
    // per workgroup; k_node_ab / the Gram tiles (exact fp32) follow as launches of their own
    const bool split_node = A.wsplit_node != nullptr && A.rot == 1;
    if (split_node) {
        A.strip0 = 0;
        const int wgs = (p->n_strips + 3) / 4;
        if (d.r == 2) { LAUNCH((split::k_node_post_split<2, 1>), wgs, split::SPLIT_WAVES * 64, A); LAUNCH((split::k_node_post_split<2, 2>), wgs, split::SPLIT_WAVES * 64, A); }
        else { LAUNCH((split::k_node_post_split<4, 1>), wgs, split::SPLIT_WAVES * 64, A); LAUNCH((split::k_node_post_split<4, 2>), wgs, split::SPLIT_WAVES * 64, A); }
    }
    // ... and the per-node coord_mlp.0 rows A' = F (Q P R), B' = F (Q P C) in the same form (four items per workgroup share F's tape) —
    // where they would be a launch of their own anyway (>= 1024 strips: the node kernel produced the next q / k / v).  Below that they ride
    // with the next block's q / k / v items in one fp32 launch that fills the chip better than two (GEOM B = 512: 16.27 against 16.45 ms/step).
    const bool split_ab = split_node && with_ab && A.ffold_s != nullptr && !next_pre;
    if (split_ab) {
        A.ab0 = 0; A.ab1 = 2 * p->n_strips;
        LAUNCH(split::k_node_ab_split, (2 * p->n_strips + split::SPLIT_WAVES - 1) / split::SPLIT_WAVES, split::SPLIT_WAVES * 64, A);
    }
    if (full > 0 && !split_node) {
        A.strip0 = 0;
        if (d.r == 2) LAUNCH(k_node_post<2>, full, 64, A); else LAUNCH(k_node_post<4>, full, 64, A);
    }
    A.ab0 = 0; A.ab1 = 2 * p->n_strips; A.g0 = 0; A.g1 = A.rot ? p->n_gtiles : 0; A.mix_nw = 0;
    const bool mix = !split_node && with_ab && full > 0 && rem > 0 && nw >= 2 && p->opt[JODO_OPT_NODE_MIX] != 0;
    if (mix) {
        if (p->gt_cache_full != full) {                      // Gram tiles among the strips of the full rounds: a prefix of the sorted list
            const int32_t* sa = p->desc.data() + p->off_gt_sa, *sc = p->desc.data() + p->off_gt_sc;
            int c = 0;
            while (c < p->n_gtiles && sa[c] < full && sc[c] < full) ++c;
            p->gt_cache_full = full; p->gt_cache_count = c;
        }
        const int gpre = A.rot ? p->gt_cache_count : 0;
        // remainder strips + what the full rounds already allow
        A.strip0 = full; A.mix_nw = rem; A.ab0 = 0; A.ab1 = 2 * full; A.g0 = 0; A.g1 = gpre;
        const int g1 = rem + (A.ab1 - A.ab0 + nw - 1) / nw + (A.g1 - A.g0 + nw - 1) / nw;
        if (nw == 2) { if (d.r == 2) LAUNCH((k_node_mix<2, 2>), g1, 128, A); else LAUNCH((k_node_mix<4, 2>), g1, 128, A); }
        else { if (d.r == 2) LAUNCH((k_node_mix<2, 4>), g1, 256, A); else LAUNCH((k_node_mix<4, 4>), g1, 256, A); }
        // the remainder strips' own items
        A.strip0 = 0; A.mix_nw = 0; A.ab0 = 2 * full; A.ab1 = 2 * p->n_strips; A.g0 = gpre; A.g1 = A.rot ? p->n_gtiles : 0;
        const int g2 = (A.ab1 - A.ab0 + 1) / 2 + (A.g1 - A.g0 + 1) / 2;
        if (d.r == 2) LAUNCH((k_node_mix<2, 2>), g2, 128, A); else LAUNCH((k_node_mix<4, 2>), g2, 128, A);
        A.ab0 = 0; A.g0 = 0;
        return JODO_OK;
    }
    if (rem > 0 && !split_node) {
        A.strip0 = full;
        if (nw == 1) { if (d.r == 2) LAUNCH(k_node_post<2>, rem, 64, A); else LAUNCH(k_node_post<4>, rem, 64, A); }
        else if (nw == 2) { if (d.r == 2) LAUNCH((k_node_postw<2, 2>), rem, 128, A); else LAUNCH((k_node_postw<4, 2>), rem, 128, A); }
        else { if (d.r == 2) LAUNCH((k_node_postw<2, 4>), rem, 256, A); else LAUNCH((k_node_postw<4, 4>), rem, 256, A); }
    }
    A.strip0 = 0;
    if (next_pre) {                                  // (with or without the k_node_ab items: the next block's q / k / v ride along)
        A.mix_nw = 3 * p->n_strips;
        if (!with_ab) { A.ab1 = A.ab0; A.g1 = A.g0; }
        LAUNCH(k_node_ab_pre, A.mix_nw + (A.ab1 - A.ab0) + (A.g1 - A.g0), 64, A);
        A.mix_nw = 0;
        return JODO_OK;
    }
    if (with_ab) {
        if (!split_ab) LAUNCH((wide::k_node_ab<256>), p->n_strips * 2, 64, A);
        if (A.rot && p->n_gtiles > 0) LAUNCH((wide::k_node_gram<256>), p->n_gtiles, 64, A);
    }
    return JODO_OK;
}

// Everything after the time / modulation prologue.  One kernel set for every width (dgt_kernels_wide.h, dgt_kernels_attn.h);
// TUNED (nf = 256 with the 8-block q / k arrangement) swaps in the nf = 256 node kernels of dgt_kernels_node.h and the
// LDS-resident-weight variant of the attention kernel.
template <int D, bool TUNED>
int forward_blocks(jodo_plan* p, hipStream_t st, KArgs& A, const int64_t* woff, float* const posbuf[2], std::unique_ptr<ProfScope>& pro) {
    const DgtDims& d = p->dims;
    int rc = JODO_OK;
    // ---- pack inputs, embeddings ----
    LAUNCH(k_pack_nodes, (p->Nn_pad + 255) / 256, 256, A);
    switch (d.ndp / 8) {
        case 1: rc = launch_embed_nodes<D, 1>(st, A); break;
        case 2: rc = launch_embed_nodes<D, 2>(st, A); break;
        case 3: rc = launch_embed_nodes<D, 3>(st, A); break;
        case 4: rc = launch_embed_nodes<D, 4>(st, A); break;
        case 5: rc = launch_embed_nodes<D, 5>(st, A); break;
        case 6: rc = launch_embed_nodes<D, 6>(st, A); break;
        case 7: rc = launch_embed_nodes<D, 7>(st, A); break;
        case 8: rc = launch_embed_nodes<D, 8>(st, A); break;
        default: return jodo_set_error(JODO_ERR_UNSUPPORTED, "node input width %d", d.ndp);
    }
    if (rc) return rc;
    // (nf = 256 tuned set: the edge embedding shares a launch with the first block's q / k / v items, k_pre_embed below)
    const bool embed_merged = p->n_items > 0 && p->opt[JODO_OPT_PRE_EMBED] != 0 && ((p->max_blocks >= 0 && p->max_blocks < d.L) ? p->max_blocks : d.L) > 0;
    if (p->n_items > 0 && !embed_merged) LAUNCH((wide::k_embed_edges<D>), p->n_items, 64, A);
    const bool pin_pair = p->opt[JODO_OPT_PIN_SYMMETRIC] == 1 && !p->force_directed, pin_dir = p->opt[JODO_OPT_PIN_SYMMETRIC] == 2 || p->force_directed;
    // shared modulation row + symmetric inputs (device flags; both can be pinned): folded coord_mlp.0 of every block, and with
    // JODO_OPT_ROT_STATS the LayerNorm statistics of equi_update in the rotated basis (A.rot tells the node kernels and k_node_ab)
    const bool can_fold = p->n_pitems > 0 && !pin_dir && p->opt[JODO_OPT_PIN_UNIFORM_T] != 2 && d.cond_ch == 0;
    A.rot = can_fold ? p->opt[JODO_OPT_ROT_STATS] : 0;      // 2: rotated with the uncentred Gram tiles (tests)
    if (can_fold) {
        if (d.L > 16) return jodo_set_error(JODO_ERR_UNSUPPORTED, "more than 16 blocks");
        FoldOffs F;
        for (int l = 0; l < 16; ++l) {
            F.c0[l] = l < d.L ? woff[JW_GLOBAL_COUNT + l * JB_BLOCK_COUNT + JB_C0_W] : 0;
            F.ine[l] = l < d.L ? woff[JW_GLOBAL_COUNT + l * JB_BLOCK_COUNT + (A.rot ? JB_INEC_W : JB_INE_W)] : 0;     // rot: centred factor
        }
        LAUNCH((wide::k_fold_coord<D, D / 16>), d.L * (D / 32) * (d.De / 4), 64, A, F, A.mfold, 0, A.mfold_s);
        if (A.rot) {                               // F_l = W0 diag(1 + sc_l) Q_l^T: what k_node_ab applies to the rotated rows
            for (int l = 0; l < d.L; ++l) F.ine[l] = woff[JW_GLOBAL_COUNT + l * JB_BLOCK_COUNT + JB_QT_W];
            LAUNCH((wide::k_fold_coord<D, D / 8>), d.L * (D / 32) * (D / 8), 64, A, F, A.ffold, 1, A.ffold_s);
        }
    }
    pro.reset();
    // ---- DGT blocks ----
    const int nblocks = (p->max_blocks >= 0 && p->max_blocks < d.L) ? p->max_blocks : d.L;
    // With at least one strip per SIMD, k_node_post of block l also produces block l + 1's q / k / v (it holds h' in
    // registers): drops a launch with its own latency-bound prologue and partial last round (QM9 B = 2500, 1409
    // strips: 23.31 -> 23.18 ms/step).  With fewer strips the separate kernel's 3x finer items fill the chip better
    // (GEOM B = 512, 710 strips: fused 35.1 vs 34.6 ms/step), so it stays separate there.
    const bool fuse_pre = TUNED && p->opt[JODO_OPT_FUSE_NEXT_QKV] != 0 && nblocks > 1 && p->n_strips >= 1024;
    // below 1 024 strips the next block's q / k / v items ride in the launch of this block's k_node_ab items (k_node_ab_pre)
    const bool ab_pre = !fuse_pre && p->opt[JODO_OPT_AB_PRE] != 0 && nblocks > 1 && (!TUNED || (p->n_strips < 1024 && p->opt[JODO_OPT_NODE_POST_WAVES] == 0));
    int cur = 0;                                   // posbuf[cur] holds the positions entering the block
    for (int l = 0; l < nblocks; ++l) {
        A.layer = l;
        A.mod_base = 32 + (int64_t)l * d.MB;
        for (int i = 0; i < JB_BLOCK_COUNT; ++i) A.wb[i] = woff[JW_GLOBAL_COUNT + l * JB_BLOCK_COUNT + i];
        // split-bf16 pair update (opt-in; A.mfold_s != NULL says jodo_dgt_forward found its preconditions met): this block's weight tape
        A.wsplit = nullptr; A.wsplit_node = nullptr; A.wsplit_attn = nullptr;
        if (A.mfold_s) {
            size_t total = 0, pair_block = 0, node_block = 0, attn_block = 0;
            (void)jodo_dgt_split_size(&p->cfg, &total, &pair_block, &node_block, &attn_block);
            const char* base = static_cast<const char*>(p->split_w);
            A.wsplit = reinterpret_cast<const unsigned short*>(base + (size_t)l * pair_block);
            if (node_block > 0) A.wsplit_node = reinterpret_cast<const unsigned short*>(base + (size_t)d.L * pair_block + (size_t)l * node_block);
            if (attn_block > 0 && p->opt[JODO_OPT_SPLIT_BF16] == 2) A.wsplit_attn = reinterpret_cast<const unsigned short*>(base + (size_t)d.L * (pair_block + node_block) + (size_t)l * attn_block);
        }
        A.pos_in = posbuf[cur]; A.pos_out = posbuf[cur ^ 1];
        {
            ProfScope ps(p, st, JODO_PROF_NODE_PRE);
            if constexpr (TUNED) {
                if (!fuse_pre && !ab_pre) {
                    A.pre_mode = 0;
                    if (l == 0 && embed_merged) LAUNCH(k_pre_embed, p->n_strips * 3 + p->n_items, 64, A);
                    else LAUNCH(k_node_pre, p->n_strips * 3, 64, A);
                } else {
                    // positions entering the block (needs the previous update); the q/k/v projections of this block were
                    // produced by the previous block's k_node_post: they only need h
                    LAUNCH(k_pos_final, (p->Nn_pad + 255) / 256, 256, A);
                    if (l == 0) {
                        A.pre_mode = 1;
                        if (embed_merged) LAUNCH(k_pre_embed, p->n_strips * 3 + p->n_items, 64, A);
                        else LAUNCH(k_node_pre, p->n_strips * 3, 64, A);
                    }
                }
            } else if (!ab_pre) {
                A.pre_mode = 0;
                if (l == 0 && embed_merged) LAUNCH((k_pre_embed_w<D>), p->n_strips * 3 + p->n_items, 64, A);
                else LAUNCH((wide::k_node_pre<D>), p->n_strips * 3, 64, A);
            } else {
                LAUNCH(k_pos_final, (p->Nn_pad + 255) / 256, 256, A);
                if (l == 0) {
                    A.pre_mode = 1;
                    if (embed_merged) LAUNCH((k_pre_embed_w<D>), p->n_strips * 3 + p->n_items, 64, A);
                    else LAUNCH((wide::k_node_pre<D>), p->n_strips * 3, 64, A);
                }
            }
        }
        cur ^= 1;                                  // the block's positions are in pos_out now
        {
            // fused attention edge phase (dgt_kernels_attn.h): pair-mode items do the work for symmetric inputs, directed-mode
            // items for asymmetric inputs and for molecules larger than a group (device flag; the other launch exits at once)
            ProfScope ps(p, st, JODO_PROF_EDGE_ATTN);
            rc = jd_launch_edge_attn(p, st, A, D, TUNED, pin_pair, pin_dir);
            if (rc) return rc;
        }
        {
            ProfScope ps(p, st, JODO_PROF_NODE_POST);
            if constexpr (TUNED) {
                A.fuse_next = (fuse_pre && l + 1 < nblocks) ? 1 : 0;
                const bool next_pre = ab_pre && l + 1 < nblocks;
                if (A.fuse_next || next_pre) {
                    static const int slots[6] = {JB_WQ, JB_BQ, JB_WK, JB_BK, JB_WV, JB_BV};
                    for (int i = 0; i < 6; ++i) A.wbn[i] = woff[JW_GLOBAL_COUNT + (l + 1) * JB_BLOCK_COUNT + slots[i]];
                    A.mod_base_next = 32 + (int64_t)(l + 1) * d.MB;
                }
                // (+ the per-node part of coord_mlp.0 pushed through the LayerNorm and the Gram tiles, dgt_kernels_wide.h)
                rc = launch_node_post_256(p, st, A, p->n_pitems > 0 && !pin_dir, next_pre);
                if (rc) return rc;
            } else {
                if (d.r == 2) LAUNCH((wide::k_node_post<D, 2>), p->n_strips, 64, A); else LAUNCH((wide::k_node_post<D, 4>), p->n_strips, 64, A);
                A.ab0 = 0; A.g0 = 0;
                const bool with_ab = p->n_pitems > 0 && !pin_dir;
                if (ab_pre && l + 1 < nblocks) {
                    static const int slots[6] = {JB_WQ, JB_BQ, JB_WK, JB_BK, JB_WV, JB_BV};
                    for (int i = 0; i < 6; ++i) A.wbn[i] = woff[JW_GLOBAL_COUNT + (l + 1) * JB_BLOCK_COUNT + slots[i]];
                    A.mod_base_next = 32 + (int64_t)(l + 1) * d.MB;
                    A.mix_nw = 3 * p->n_strips;
                    A.ab1 = with_ab ? 2 * p->n_strips : 0;
                    A.g1 = (with_ab && A.rot) ? p->n_gtiles : 0;
                    LAUNCH((k_node_ab_pre_w<D>), A.mix_nw + A.ab1 + A.g1, 64, A);
                    A.mix_nw = 0;
                } else {
                    if (with_ab) LAUNCH((wide::k_node_ab<D>), p->n_strips * 2, 64, A);
                    if (A.rot && p->n_gtiles > 0) LAUNCH((wide::k_node_gram<D>), p->n_gtiles, 64, A);
                }
            }
        }
        if (p->n_items > 0) {
            ProfScope ps(p, st, JODO_PROF_EDGE_UPDATE);          // exactly one of the two does the work (device flag)
            rc = jd_launch_edge_update(p, st, A, D, pin_pair, pin_dir);
            if (rc) return rc;
            std::swap(A.e, A.e_out);               // the state the next block reads is the one just written
            p->last_e_buf ^= 1;
        }
    }
    // ---- heads + outputs ----
    ProfScope epi(p, st, JODO_PROF_EPILOGUE);
    A.pos_in = posbuf[cur]; A.pos_out = posbuf[cur ^ 1];
    A.layer = nblocks;                             // k_pos_final adds the last block's partial updates if any ran
    LAUNCH(k_pos_final, (p->Nn_pad + 255) / 256, 256, A);
    p->last_pos_buf = cur ^ 1;
    const bool heads_mix = A.pin_sym == 1 && A.pd.n_ut_pad > 0 && p->opt[JODO_OPT_HEADS_MIX] != 0;
    switch (d.KEH / 32) {
        case 3: rc = launch_edge_head<D, 3>(st, A, heads_mix); break;
        case 4: rc = launch_edge_head<D, 4>(st, A, heads_mix); break;
        case 5: rc = launch_edge_head<D, 5>(st, A, heads_mix); break;
        case 6: rc = launch_edge_head<D, 6>(st, A, heads_mix); break;
        case 7: rc = launch_edge_head<D, 7>(st, A, heads_mix); break;
        case 8: rc = launch_edge_head<D, 8>(st, A, heads_mix); break;
        case 9: rc = launch_edge_head<D, 9>(st, A, heads_mix); break;
        case 10: rc = launch_edge_head<D, 10>(st, A, heads_mix); break;
        case 11: rc = launch_edge_head<D, 11>(st, A, heads_mix); break;
        case 12: rc = launch_edge_head<D, 12>(st, A, heads_mix); break;
        case 13: rc = launch_edge_head<D, 13>(st, A, heads_mix); break;
        case 15: rc = launch_edge_head<D, 15>(st, A, heads_mix); break;
        default: return jodo_set_error(JODO_ERR_UNSUPPORTED, "edge head width %d", d.KEH);
    }
    if (rc) return rc;
    LAUNCH(k_finalize_nodes, (p->B * p->N + 255) / 256, 256, A);
    {
        const size_t tot = (size_t)p->B * p->N * p->N;
        LAUNCH(k_finalize_edges, (unsigned)((tot + 255) / 256), 256, A);
    }
    return JODO_OK;
}

}  // namespace

extern "C" int jodo_dgt_forward(jodo_plan* p, const void* desc_dev, const float* packed_w, const int64_t* woff,
                                int n_woff, const float* xh, const float* edge_x, const float* cond_x,
                                const float* cond_edge_x, const float* noise_level, const float* context,
                                float* out_xh, float* out_edge, int32_t* flags_dev, void* workspace, void* stream) {
    if (!p || !desc_dev || !packed_w || !woff || !xh || !edge_x || !noise_level || !out_xh || !out_edge || !flags_dev ||
        !workspace)
        return jodo_set_error(JODO_ERR_ARG, "dgt_forward: null argument");
    const DgtDims& d = p->dims;
    if (n_woff != JW_GLOBAL_COUNT + d.L * JB_BLOCK_COUNT)
        return jodo_set_error(JODO_ERR_ARG, "dgt_forward: weight table has %d slots, expected %d", n_woff,
                              JW_GLOBAL_COUNT + d.L * JB_BLOCK_COUNT);
    if ((cond_x == nullptr) != (cond_edge_x == nullptr))
        return jodo_set_error(JODO_ERR_ARG, "dgt_forward: cond_x and cond_edge_x must both be given or both NULL");
    if (d.cond_ch > 0 && !context) return jodo_set_error(JODO_ERR_ARG, "dgt_forward: conditional model needs context");
    hipStream_t st = (hipStream_t)stream;

    KArgs A;
    A.pd = make_plan_dev(p, desc_dev);
    A.d = d;
    A.W = packed_w;
    for (int i = 0; i < JW_GLOBAL_COUNT; ++i) A.wg[i] = woff[i];
    for (int i = 0; i < JB_BLOCK_COUNT; ++i) A.wb[i] = 0;
    A.mod_base = 0; A.layer = 0; A.force_directed = p->force_directed; A.pin_sym = p->force_directed ? 0 : p->opt[JODO_OPT_PIN_SYMMETRIC];
    A.half_rows = (A.pin_sym == 1 && p->n_pitems > 0 && p->opt[JODO_OPT_HALF_ROWS] != 0) ? 1 : 0;
    A.pin_uni = p->opt[JODO_OPT_PIN_UNIFORM_T]; A.pre_mode = 0; A.strip0 = 0; A.item0 = 0; A.dir_split = 0; A.rot = 0; A.mix_nw = 0; A.ab0 = 0; A.ab1 = 0; A.g0 = 0; A.g1 = 0; A.fuse_next = 0; A.mod_base_next = 0;
    for (int i = 0; i < 6; ++i) A.wbn[i] = 0;
    fill_ws(A, p, workspace);
    // JODO_OPT_SPLIT_BF16 (opt-in): the folded pair update in the split-bf16 form.  Only where its preconditions hold — nf 256,
    // unconditional, both paths pinned (symmetric inputs, shared modulation row: what a sampler pins after its first self-conditioned
    // evaluation), rotated statistics in their default form, one circulant offset per item, the weight tape handed over — otherwise the
    // exact-fp32 kernels run as always.
    if (p->opt[JODO_OPT_SPLIT_BF16] >= 1 && p->split_w && (d.D == 256 || d.D == 384) && d.cond_ch == 0 && !p->force_directed && p->n_pitems > 0 && p->pitems_single &&
        p->opt[JODO_OPT_PIN_SYMMETRIC] == 1 && p->opt[JODO_OPT_PIN_UNIFORM_T] == 1 && p->opt[JODO_OPT_ROT_STATS] == 1) {
        size_t total = 0, per_block = 0, node_block = 0, attn_block = 0;
        if (jodo_dgt_split_size(&p->cfg, &total, &per_block, &node_block, &attn_block) == JODO_OK && total == p->split_bytes)
            A.mfold_s = ws_ptr<unsigned short>(workspace, p->ws.mfold_s);
        if (A.mfold_s && node_block > 0) A.ffold_s = ws_ptr<unsigned short>(workspace, p->ws.ffold_s);   // tuned nf 256 set: k_node_ab_split
    }
    float* posbuf[2] = {ws_ptr<float>(workspace, p->ws.pos0), ws_ptr<float>(workspace, p->ws.pos1)};
    A.pos_in = posbuf[0]; A.pos_out = posbuf[1];
    A.flags = flags_dev;
    A.dbgt = static_cast<unsigned long long*>(p->dbg_timing);
    A.xh = xh; A.edge_x = edge_x; A.cond_x = cond_x; A.cond_edge_x = cond_edge_x; A.noise = noise_level; A.context = context;
    A.out_xh = out_xh; A.out_edge = out_edge;
    const float* W = packed_w;
    int rc;

    // ---- time embedding -> modulation vectors ----
    p->last_e_buf = 0;
    std::unique_ptr<ProfScope> pro(new ProfScope(p, st, JODO_PROF_PROLOGUE));      // ends after the embeddings; released on every return path
    LAUNCH(k_flags_init, 1, 256, A);
    { const size_t tot = (size_t)p->B * p->N * p->N; LAUNCH(k_check_sym, (unsigned)((tot + 255) / 256), 256, A); }
    LAUNCH(k_time1, p->B, 256, A);
    // a conditional model never shares the modulation row (k_flags_init): no flag, and rowgemm then reuses each activation
    // chunk for four output blocks instead of one
    const int* uflag = d.cond_ch > 0 ? nullptr : flags_dev + FLAG_UNIFORM_T;
    // (the last writer of the time embedding also leaves SiLU(time_emb): the input of the fused modulation projection)
    float* tembs = ws_ptr<float>(workspace, p->ws.tembs);
    rc = rowgemm(st, A.hid1, d.T, A.temb, d.T, W + A.wg[JW_TIME_W3], W + A.wg[JW_TIME_B3], p->B, d.T, d.T / 32, 0, 0, uflag,
                 d.cond_ch > 0 ? nullptr : tembs);
    if (rc) return rc;
    if (d.cond_ch > 0) {
        LAUNCH(k_cond1, p->B * d.cond_ch, 256, A);
        rc = rowgemm(st, A.condh, d.D, A.condh2, d.D, W + A.wg[JW_COND_W2], W + A.wg[JW_COND_B2], p->B * d.cond_ch, d.D,
                     d.D / 32, 0, 0, nullptr);
        if (rc) return rc;
        rc = rowgemm(st, A.condh2, (int64_t)d.cond_ch * d.D, A.temb, d.T, W + A.wg[JW_COND_LIN_W], W + A.wg[JW_COND_LIN_B],
                     p->B, d.cond_ch * d.D, d.T / 32, 0, 1, nullptr, tembs);
        if (rc) return rc;
    }
    rc = rowgemm(st, tembs, d.T, A.mods, d.Mtot, W + A.wg[JW_MOD_W], W + A.wg[JW_MOD_B], p->B, d.T, (int)(d.Mtot / 32), 0, 0,
                 uflag);
    if (rc) return rc;
    if (!d.wide) return forward_blocks<256, true>(p, st, A, woff, posbuf, pro);
    if (d.D == 128) return forward_blocks<128, false>(p, st, A, woff, posbuf, pro);
    return d.D == 256 ? forward_blocks<256, false>(p, st, A, woff, posbuf, pro) : forward_blocks<384, false>(p, st, A, woff, posbuf, pro);
}

extern "C" int jodo_debug_fetch(jodo_plan* p, const void* workspace, int what, float* dst, int64_t* count, void* stream) {
    if (!p || !workspace || !dst || !count) return jodo_set_error(JODO_ERR_ARG, "debug_fetch: null");
    const char* ws = static_cast<const char*>(workspace);
    const void* src = nullptr;
    int64_t n = 0;
    switch (what) {
        case 0: src = ws + p->ws.h; n = (int64_t)p->Nn * p->dims.D; break;
        case 1: src = ws + (p->last_e_buf ? p->ws.e2 : p->ws.e); n = p->rows * p->dims.De; break;
        case 2: src = ws + (p->last_pos_buf ? p->ws.pos1 : p->ws.pos0); n = (int64_t)p->Nn * 4; break;
        case 3: src = ws + p->ws.hhat; n = (int64_t)p->Nn * p->amax_parts * p->dims.D; break;
        case 5: src = ws + p->ws.mods; n = p->dims.Mtot; break;
        case 6: src = ws + p->ws.q; n = (int64_t)p->Nn * p->dims.QKP; break;
        default: return jodo_set_error(JODO_ERR_ARG, "debug_fetch: unknown selector %d", what);
    }
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "debug_fetch: %s", hipGetErrorString(e));
    *count = n;
    return JODO_OK;
}
