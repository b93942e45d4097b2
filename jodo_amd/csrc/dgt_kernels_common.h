// Kernel argument block and small shared device helpers for the DGT forward kernels.
#pragma once
#include "dgt_device.h"
#include "dgt_plan.h"

// flags_dev layout (int32[8])
enum { FLAG_NAN = 0, FLAG_FIRST = 1, FLAG_UNIFORM_T = 2, FLAG_COND_NONZERO = 3, FLAG_ASYM = 4, FLAG_NAN_COUNT = 5, FLAG_PIN_VIOLATED = 6 };

struct KArgs {
    PlanDev pd;
    DgtDims d;
    const float* W;                       // packed weight blob
    int64_t wg[JW_GLOBAL_COUNT];          // global slot offsets (floats)
    int64_t wb[JB_BLOCK_COUNT];           // slot offsets of the current block (floats)
    int64_t mod_base;                     // offset of the current block inside a modulation vector
    int layer;
    int force_directed;                   // debug: never take the symmetric pair path
    int pin_sym, pin_uni;                 // JODO_OPT_PIN_*: variants the launcher left out (checked against the device flags in k_finalize_nodes)
    int strip0;                           // k_node_post*: first strip of this launch (a layer's strips may be split over two launches)
    int half_rows;                        // pinned symmetric inputs: of a molecule that fits an attention group only the edge row of a pair's EVALUATING lane
                                          // (pair_of: (i, i + d)) is ever read again, so e / ehid of the mirror row are not written
    int item0, dir_split;                 // pair update: first item of this launch; 1 = two workgroups per item, one direction each
    int pers_n;                           // pair update, persistent launches (-DJODO_X_UPD_PERS experiment builds): items of this launch
    int mix_nw, ab0, ab1, g0, g1;         // k_node_mix: workgroups in the k_node_postw role; k_node_ab items [ab0, ab1); Gram tiles [g0, g1)
    int rot;                              // JODO_OPT_ROT_STATS and a launch sequence that can use it: under FLAG_UNIFORM_T && !FLAG_ASYM the node
                                          // kernels write Q P (W_row h + b), Q P W_col h and the pair update takes its LayerNorm statistics from them
    // k_node_post*: when fuse_next != 0 the kernel also produces the NEXT block's q / k / v (LN1 + modulate of the
    // h it has just computed), with the next block's weights (wbn = its WQ, BQ, WK, BK, WV, BV slots) and modulation
    int fuse_next;
    int64_t wbn[6];
    int64_t mod_base_next;
    int pre_mode;                         // k_node_pre: 0 = also advance the positions, 1 = q/k/v only (k_pos_final did it)
    // workspace
    float *hid1, *temb, *mods, *condh, *condh2;
    float *pos_in, *pos_out, *dpos, *cpos, *feat, *h, *hhat, *astat, *q, *k, *v, *n2e, *wrow, *wcol, *ua, *ub, *rmean, *mfold, *ffold, *ahid, *apred;
    int* eflag;
    float *e, *ehid, *epred, *dposE, *gramE;
    float* e_out;                         // edge state written by the update kernels (ping-pong with e: never in place,
                                          // two workgroups of a direction-split item read the same input rows)
    // opt-in split-bf16 pair update (JODO_OPT_SPLIT_BF16, dgt_kernels_split.h): the static weight tape of the current block (NULL = off)
    // and the split image of the folded coord_mlp.0 matrices of all blocks (k_fold_coord writes it when mfold_s != NULL)
    const unsigned short* wsplit;
    const unsigned short* wsplit_attn;    // the cyclic tape of the fused attention kernel (k_edge_attn<., ., ., 4>), NULL = off
    const unsigned short* wsplit_node;    // the same for the node kernel of the tuned nf 256 set (k_node_post_split), NULL = off
    unsigned short* mfold_s;
    unsigned short* ffold_s;              // split image of ffold (the rotated per-node factor F of every block): k_node_ab_split
    int* flags;
    unsigned long long* dbgt;             // debug: per-phase cycle sums (builds with -DJODO_PHASE_TIMING only)
    // API tensors
    const float *xh, *edge_x, *cond_x, *cond_edge_x, *noise, *context;
    float *out_xh, *out_edge;
};

// k_fold_coord: per-layer slot offsets of coord_mlp.0 and of the [e ; G] part of input_lin (one launch covers every block)
struct FoldOffs { int64_t c0[16], ine[16]; };      // ine: the right-hand factor of the launch ([e ; G] part of input_lin, its centred copy, or Q^T)

namespace jd {

// atoms of an attention group (dgt_kernels_attn.h ATT_LANES, dgt_plan.cpp G): a molecule above it runs the directed attention items,
// which read every edge row of that molecule
constexpr int PAIR_GROUP_LANES = 128;

// the rotated-statistics path is taken by a call iff the launcher allows it and the call has a shared modulation row and symmetric inputs
__device__ __forceinline__ bool rot_active(const KArgs& A) { return A.rot && A.flags[FLAG_UNIFORM_T] && !A.flags[FLAG_ASYM]; }
}

namespace jd {

__device__ __forceinline__ const float4* wq(const KArgs& A, int64_t off, int lane) {
    return reinterpret_cast<const float4*>(A.W + off) + lane;
}

// modulation row of the molecule that owns packed node v
__device__ __forceinline__ const float* mod_row(const KArgs& A, int node_b) {
    const int trow = A.flags[FLAG_UNIFORM_T] ? 0 : node_b;
    return A.mods + (size_t)trow * A.d.Mtot;
}

struct LaneNode {      // per-lane view of "its" packed node
    int v, b, i, n, noff, eoff;
    bool valid;
};

__device__ __forceinline__ LaneNode lane_node(const KArgs& A, int strip, int j) {
    LaneNode L;
    L.v = strip * 32 + j;
    L.b = A.pd.node_b[L.v];
    L.i = A.pd.node_i[L.v];
    L.n = A.pd.node_n[L.v];
    L.noff = A.pd.node_noff[L.v];
    L.eoff = A.pd.node_eoff[L.v];
    L.valid = L.n > 0;
    return L;
}

// Positions entering a block = previous positions + the previous update's contributions (MultiCondEquiUpdate, mol_gnn.py:90-92:
// `pos + scatter(trans, row, reduce='add')`).  The contributions are summed AMONG THEMSELVES and added to the position once, as the
// reference does: rounds 1-4 added every neighbour's term into the running position, i.e. up to 180 roundings at ulp(|x| ~ 4) per
// block instead of one — the digit that the n > 128 tolerance regime of tests/helpers.py paid for (round-4 review, item 4).  The sum
// runs in double (fixed column order; k_pos_final is a few microseconds of loads, not arithmetic).
__device__ __forceinline__ float4 advance_position(const KArgs& A, float4 p, int v, int strip, int n, int i, int eoff) {
    double sx = 0., sy = 0., sz = 0.;
    if (A.flags[FLAG_ASYM]) {
        const int parts = A.pd.strip_parts[strip];
        for (int q = 0; q < parts; ++q) {
            const float4 dp = reinterpret_cast<const float4*>(A.dpos)[(size_t)v * A.pd.max_parts + q];
            sx += dp.x; sy += dp.y; sz += dp.z;
        }
    } else {                                               // pair path: one contribution per edge row (i, c)
        const float4* row = reinterpret_cast<const float4*>(A.dposE) + (size_t)eoff + (size_t)i * n;
        // eight, then four rows in flight per step (one load per iteration paid one exposed L2 round trip per neighbour: 15 us per
        // launch at QM9 B = 2500, nine launches per forward; GEOM molecules have up to 181 atoms)
        int c = 0;
        for (; c + 8 <= n; c += 8) {
            float4 d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = row[c + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + u != i) { sx += d[u].x; sy += d[u].y; sz += d[u].z; }
        }
        for (; c + 4 <= n; c += 4) {
            const float4 d0 = row[c], d1 = row[c + 1], d2 = row[c + 2], d3 = row[c + 3];
            if (c != i) { sx += d0.x; sy += d0.y; sz += d0.z; }
            if (c + 1 != i) { sx += d1.x; sy += d1.y; sz += d1.z; }
            if (c + 2 != i) { sx += d2.x; sy += d2.y; sz += d2.z; }
            if (c + 3 != i) { sx += d3.x; sy += d3.y; sz += d3.z; }
        }
        for (; c < n; ++c) {
            if (c == i) continue;
            const float4 dp = row[c];
            sx += dp.x; sy += dp.y; sz += dp.z;
        }
    }
    p.x += (float)sx; p.y += (float)sy; p.z += (float)sz;
    return p;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

}  // namespace jd
