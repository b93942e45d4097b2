// Fused forward chains of the training step on the strip model (train_fused.hip; round 5).
//
// The grad-enabled forward of round 4 evaluated every per-edge operation of a block as a launch of its own (a GEMM per projection,
// elementwise kernels between them: ~60 launches per block, every [R, 64] .. [R, 256] intermediate through HBM twice).  The three
// per-edge chains of EquivariantMixBlock.forward (mol_gnn.py:270-322) are now one kernel each, written like the inference kernels
// (dgt_device.h: a wave owns 32 edge rows, projections on v_mfma_f32_32x32x2_f32 with the accumulator of one projection as the B
// operand of the next) with an ACTIVATION-SAVE epilogue: everything jodo_train_backward reads is stored exactly where the op-by-op
// forward stored it, so the backward is unchanged.
//   chain A  d2 -> GBF -> edge_emb([G ; e]) -> LayerNorm1 -> modulate -> tanh(lin_edge0), tanh(lin_edge1)        (:279-296, layers.py:165-184)
//   chain B  e + g1 * node2edge -> LayerNorm2 -> modulate -> ff_linear3 -> SiLU, dropout -> ff_linear4 -> dropout -> gate -> readout (:313-317, :570)
//   chain C  input_lin([h_row ; h_col ; e ; G]) -> LayerNorm -> modulate -> coord_mlp.0 -> SiLU -> coord_mlp.2 -> tanh  (mol_gnn.py:71-84)
// The parameters change every optimiser step, so their MFMA operand images are packed by a small kernel per block and forward
// (fused_pack_block) into the workspace.  The host-emulation build of the CPU suite (tests/emul) has no matrix instructions: there
// fused_available() is false and dgt_train.hip runs the op-by-op sequence, which stays in the library as the reference form.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "train_common.h"

namespace jt {

struct FusedDims { int D, De, r, QK, ce, L; };

// floats of packed operands per block; offsets (floats) of the pieces inside a block's slice
struct FusedPackLayout { size_t ee, l0, l1, ff3, ff4, ero, in_eg, c0, tab, total;                      // forward images
                         size_t c0t, int_eg, ff4t, ff3t, l1t, l0t, eet, total_bwd; };   // transposed images of the backward chains (after `total`)
FusedPackLayout fused_pack_layout(const FusedDims& d);

bool fused_available(const FusedDims& d);

struct FusedBlockParams {       // device pointers of one block's parameters (PyTorch layouts)
    const float *edge_emb_w, *edge_emb_b, *le0, *le1, *ff3_w, *ff3_b, *ff4_w, *ff4_b, *ero_w, *ero_b, *in_w, *in_b, *c0_w, *c0_b, *c2_w, *n2e_b;
    const float *gbf_means, *gbf_stds;
};

struct FusedTopo { int R; const int *edge_a, *edge_c, *edge_mol; int save = 1; };   // save = 0: a forward without a backward (activations only the
                                                                                     // backward reads are not stored)

void fused_pack_block(hipStream_t s, const FusedDims& d, const FusedBlockParams& p, float* packed);
// transposed operand images for the backward chains (dX = W^T dY on the same MFMA orientation), into packed + layout offsets *t
void fused_pack_block_bwd(hipStream_t s, const FusedDims& d, const FusedBlockParams& p, float* packed);

// chain A.  pos [Nn, 3]; gm [B, 2] (scale, shift); e_in [R, De]; emod [B, 6 De] (shift at 0, scale at De)
void fused_chain_a(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* pos, const float* gm,
                   const float* e_in, const float* emod, float* d2, float* G, float* xh_e1, float* rs_e1, float* et, float* t0, float* t1);
// chain B.  n2e [Nn, De] (node2edge_lin(hhat), bias not yet added); emod: g1 at 2 De, shift2 at 3 De, scale2 at 4 De, g2 at 5 De
void fused_chain_b(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* e_in, const float* n2e,
                   const float* emod, Drop drop_a3, Drop drop_f4, float* xh_en, float* rs_en, float* en, float* f3, float* a3, float* f4, float* e_out,
                   float* eh, int ld_eh, int eh_col);
// chain C.  hr / hc [Nn, D] = W_row h, W_col h; qmod [B, 2 D] (shift at 0, scale at D)
void fused_chain_c(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* e_in, const float* G,
                   const float* hr, const float* hc, const float* qmod, float* xh_pre, float* rs_pre, float* u, float* c0pre, float* c0a, float* inv);


// ---- backward: the input-gradient side of the same three chains (round 5).  Each kernel walks one chain backwards for 32 edge rows per
// wave and stores exactly the arrays the remaining launches read: the dY operands of the weight-gradient products (which stay GEMMs: their
// contraction index is the row) and of the per-molecule modulation sums.
// chain C':  dinv (in place: x (1 - inv^2)) -> d coord_mlp.0 output dc0 [R, D] -> du = W0^T dc0 [R, D] -> LayerNorm + modulate backward ->
//            dpre [R, D] -> de += W_e^T dpre, dG = W_g^T dpre
void fused_bwd_c(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* inv, float* dinv,
                 const float* c0pre, const float* xh_pre, const float* rs_pre, const float* qmod, float* dc0, float* du, float* dpre, float* de, float* dG);
// chain B':  de_out -> f4d = dropout(f4) (for d g2), df4 = g2 de_out mask -> dhid = (W4^T df4) mask3 SiLU'(f3) [R, r De] -> den = de_out + W3^T dhid
//            -> LayerNorm2 + modulate backward -> de_prev (written)
void fused_bwd_b(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* de_out, const float* f4,
                 const float* f3, const float* xh_en, const float* rs_en, const float* emod, Drop drop_a3, Drop drop_f4, float* f4d, float* df4, float* dhid,
                 float* den, float* de_prev);
// chain A':  det = lin_edge1^T dt1 + lin_edge0^T dt0 [R, De] -> LayerNorm1 + modulate backward -> de1 -> dG += W_G^T de1, de_prev += W_e^T de1
void fused_bwd_a(hipStream_t s, const FusedDims& d, const FusedTopo& t, const FusedBlockParams& p, const float* packed, const float* dt1, const float* dt0,
                 const float* xh_e1, const float* rs_e1, const float* emod, float* det, float* de1, float* dG, float* de_prev);

// ---- node rows (F = D in {128, 256, 384}): LayerNorm + modulate with its row statistics in one launch, one wave per row.
// forward:  x = res_b ? a + mods[mol, g_off + f] res_b : a;  xhat, rstd kept;  y = xhat (1 + mods[mol, sc_off + f]) + mods[mol, sh_off + f]
void fused_node_ln_mod(hipStream_t s, long rows, int F, const float* a, const float* res_b, const int* row_mol, const float* mods, int ldm, int g_off,
                       int sh_off, int sc_off, float* xhat, float* rstd, float* y);
// backward: dx (+)= rstd (g - mean(g) - xhat mean(g xhat)),  g = dy (1 + mods[mol, sc_off + f])   (the modulation sums stay k_seg_colsum2)
void fused_node_ln_mod_bwd(hipStream_t s, long rows, int F, const float* dy, const float* xhat, const float* rstd, const int* row_mol, const float* mods,
                           int ldm, int sc_off, float* dx, int acc);

// ---- attention of a block: forward in one launch, backward in two (a wave per target / per source atom; bit-identical to the op-by-op
// kernels of train_ops.h: scores | softmax | messages and their six backward kernels).  No dropout on the weights (SITE_ALPHA is p = 0).
struct AttnTopo {
    int Nn, N;                                   // N: the largest molecule of the batch (LDS per atom)
    const int *node_mol, *nn, *node_off, *edge_off;
    int one_wave;                                // tests force the large-batch form
    int ldqk, ldv, ldd_qk, ldd_v;                // row strides of q / k, of v, of d q / d k, of d v: q | k | v of a block live side by side in one
                                                 // [Nn, 2 QK + D] array (the merged projection's output), and so do their gradients
};
bool fused_attention_available(int D, int H, int N);       // widths 128 / 256 / 384, at most 16 heads, molecules of at most 192 atoms
void fused_attn_fwd(hipStream_t s, const AttnTopo& t, int D, int H, int XH, int SC, float inv_sqrt_c, const float* q, const float* k, const float* t0,
                    const float* adj2d, const float* adjsp, const float* v, const float* t1, float* alpha, float* hhat);
void fused_attn_bwd(hipStream_t s, const AttnTopo& t, int D, int H, int XH, int SC, float inv_sqrt_c, const float* dhhat, const float* q, const float* k,
                    const float* v, const float* t0, const float* t1, const float* alpha, float* dS, float* dt1, float* dt0, float* dq, float* dk, float* dv);

// ---- Gaussian layer backward in one pass over 32-row chunks (a wave per chunk, a lane per Gaussian; De <= 129): d x' per row (dxp, and
// dd2 (+)= d x' (1 + scale)) and the chunk partials of d means / d stds, [ceil(rows / 32), De - 1] each, for the column sums
void fused_gbf_bwd(hipStream_t s, long rows, int De, const float* d2, const int* row_mol, const float* gm, const float* means, const float* stds, const float* dG,
                   int ldg, int gcol, float* dxp, float* dd2, int acc, float* part_m, float* part_s);

}  // namespace jt
