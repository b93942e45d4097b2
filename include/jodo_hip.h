/* libjodo_hip.so — C ABI of the MI355X-native JODO DGT denoising hot path.
 *
 * What this replaces in the reference (GRAPH-0/JODO, pure Python; there is no upstream FFI, so the
 * "reference interface" each entry point stands for is the Python call it makes redundant):
 *
 *   jodo_plan_*          <- the per-call dense->sparse conversion in DGT_concat.forward
 *                           (models/mol_gnn.py:512-514: edge_mask.nonzero(), dense_to_sparse) —
 *                           built once per (batch of atom counts) instead of once per step
 *   jodo_dgt_forward     <- DGT_concat.forward          models/mol_gnn.py:491-594
 *                           Cond_DGT_concat.forward     models/mol_gnn.py:687-794
 *                           (incl. EquivariantMixBlock.forward :270-322, TransMixLayer
 *                           models/layers.py:131-186, MultiCondEquiUpdate :71-94,
 *                           CondGaussianLayer models/layers.py:328-334, time_mlp :481-489)
 *   jodo_debug_mlp       <- (no reference counterpart) hardware self-test of the MFMA lane maps
 *
 * Conventions: every function returns 0 on success or a negative JODO_ERR_* code and records a
 * message retrievable with jodo_last_error() (thread-local).  The library never allocates or frees
 * device memory: the caller (PyTorch) owns every device buffer and passes raw pointers; sizes are
 * queried up front.  All launches go to the hipStream_t passed in (as void*); there are no hidden
 * synchronisations and no internal streams.  A plan handle is host memory owned by the library
 * (jodo_plan_destroy frees it); handles are independent and re-entrant, one handle is not
 * thread-safe.  Tensors are contiguous row-major fp32 in the reference's dense layouts.
 */
#ifndef JODO_HIP_H
#define JODO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JODO_OK 0
#define JODO_ERR_ARG (-1)
#define JODO_ERR_LAUNCH (-2)
#define JODO_ERR_UNSUPPORTED (-3)

/* Model hyper-parameters (config.model.* / config.data.* of the reference's configs). */
typedef struct {
    int32_t nf;          /* D: node hidden width: 128, 256 or 384 (README.md:150,162 the GEOM "Base" model `--config.model.nf 128`,
                            :168 `--config.model.nf 384`); anything else -> JODO_ERR_UNSUPPORTED */
    int32_t n_layers;    /* L in [1, 32] with: the per-block edge readout 2De // L at most one 32-row block wide (L >= 2 / 4 / 6 at
                            nf 128 / 256 / 384), and the edge head's input De + L * cep a multiple of 32, cep = that readout padded to
                            16 or 32 (at nf 128, and at nf 256 with L >= 8, this means even L); the Python module further asks for
                            L <= 16 and a head input of 96..480 except 448.  Tested: L = 6 (nf 128), 8 and 10 (nf 256), 10 (nf 384) */
    int32_t n_heads;     /* H (16)                                              */
    int32_t n_extra;     /* XH: adjacency heads (2)                             */
    int32_t mlp_ratio;   /* r                                                   */
    int32_t in_node_dim; /* nd = atom_types + include_fc_charge                 */
    int32_t edge_ch;     /* ch                                                  */
    int32_t cond_ch;     /* 0 = DGT_concat, >0 = cond_DGT_concat                */
    float spatial_cut_off;
    float edge_quan_th;
    int32_t layout;      /* q / k / lin_edge0 arrangement: 0 = automatic (nf 256: 8 blocks, two heads per block + tail block,
                            with the nf-256 node kernels and LDS-resident attention weights; otherwise one 32-row block per
                            head); 1 = one block per head even at nf 256 (tests: runs the width-generic instantiations) */
} jodo_cfg;

/* Slots of the weight-offset table handed to jodo_dgt_forward (offsets in floats into the packed
 * weight blob produced by jodo_dgt_pack_weights; tests/test_packing.py keeps this enum, the C packer and the
 * independent Python packer tests/py_packing_model.py in lock-step, blob for blob). */
enum jodo_wslot_global {
    JW_TIME_FREQ = 0, JW_TIME_W1, JW_TIME_B1, JW_TIME_W3, JW_TIME_B3,
    JW_COND_W0, JW_COND_B0, JW_COND_W2, JW_COND_B2, JW_COND_LIN_W, JW_COND_LIN_B,
    JW_MOD_W, JW_MOD_B,
    JW_NODE_EMB_W, JW_NODE_EMB_B, JW_EDGE_EMB_W, JW_EDGE_EMB_B, JW_GBF_TOP,
    JW_NH1_W, JW_NH1_B, JW_NH2_W, JW_NH2_B, JW_NH3_W, JW_NH3_B,
    JW_EH1_W, JW_EH1_B, JW_EH2_W, JW_EH2_B, JW_EH3_W, JW_EH3_B,
    JW_GLOBAL_COUNT
};
/* The last six block slots hold the rotated LayerNorm statistics of equi_update (pair update under a shared modulation row;
 * DESIGN.md 4a): with P the centring projection and Q the orthogonal factor of  Q (P W_in[:, e ; G]) = [L ; 0]:
 * ROWQ = Q P W_row, COLQ = Q P W_col, INQ_B = Q P b, LQ = L (block upper triangular, K = 2 De), INEC = P W_in[:, e ; G],
 * QT = Q^T packed as a D x D projection (right-hand factor of the per-forward fold W0 diag(1 + sc) Q^T). */
enum jodo_wslot_block {
    JB_WQ = 0, JB_BQ, JB_WK, JB_BK, JB_WV, JB_BV,
    JB_EE_W, JB_EE_B, JB_LE0_W, JB_LE1_W, JB_N2E_W, JB_N2E_B,
    JB_FF1_W, JB_FF1_B, JB_FF2_W, JB_FF2_B, JB_FF3_W, JB_FF3_B, JB_FF4_W, JB_FF4_B,
    JB_INE_W, JB_ROW_W, JB_COL_W, JB_IN_B, JB_C0_W, JB_C0_B, JB_C2_W, JB_CSCALE,
    JB_NRO_W, JB_NRO_B, JB_ERO_W, JB_ERO_B, JB_GBF,
    JB_ROWQ_W, JB_COLQ_W, JB_INQ_B, JB_LQ_W, JB_INEC_W, JB_QT_W,
    JB_BLOCK_COUNT
};
/* table length = JW_GLOBAL_COUNT + n_layers * JB_BLOCK_COUNT; block l slot s lives at
 * JW_GLOBAL_COUNT + l * JB_BLOCK_COUNT + s */

/* ---- weights --------------------------------------------------------------------------------------------------
 * jodo_dgt_pack_weights  <- the parameter container of DGT_concat / Cond_DGT_concat: the 351 (QM9) / 431 (GEOM L = 10) /
 *                           357 (conditional) tensors of its state_dict, names and shapes as registered by the
 *                           constructors models/mol_gnn.py:414-489 and :601-684 (a leading "module." — checkpoints
 *                           saved through nn.DataParallel, utils.py:23-30 — is accepted and ignored).
 * A host with nothing but a C FFI hands over the named fp32 tensors (HOST pointers, contiguous, PyTorch [out, in]
 * layout) and receives the packed blob in MFMA A-operand order plus the offset table jodo_dgt_forward takes.
 *   jodo_dgt_packed_size        number of floats of the blob and of offset-table slots for this configuration
 *   jodo_dgt_pack_weights_host  pack into caller-owned host memory (no device work at all)
 *   jodo_dgt_pack_weights       pack and upload into caller-owned DEVICE memory; this load-time call synchronises
 *                               `stream` before returning (its staging buffer lives only for the call)
 * A missing or mis-sized parameter is an error (JODO_ERR_ARG, the name is in jodo_last_error()). */
typedef struct {
    const char* name;       /* state_dict key, e.g. "e_block_3.attn_mpnn.lin_edge0.weight" */
    const float* data;      /* host pointer, contiguous fp32 */
    const int64_t* shape;
    int32_t ndim;
} jodo_tensor;
int jodo_dgt_packed_size(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, size_t* n_floats, int* n_woff);
int jodo_dgt_pack_weights_host(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, float* packed_host,
                               size_t cap_floats, int64_t* woff_out, int n_woff);
int jodo_dgt_pack_weights(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, void* packed_dev,
                          size_t cap_floats, int64_t* woff_out, int n_woff, void* stream);

typedef struct jodo_plan jodo_plan;

/* Build the execution plan for a batch: B molecules with n_nodes[b] atoms (host array), padded
 * width N of the dense API tensors.  Molecules are ordered by size internally; masks are implied
 * (prefix masks, diagonal excluded), which is what the reference's samplers produce
 * (sampling.py:193-201).  max_chunk packs three work-item sizes (0 in a field = automatic): bits 0-15 sources per
 * directed edge work item (embeddings, directed update; default 8), bits 16-23 pair offsets per item of the pair update
 * kernel (default 1), bits 24-31 the pair-mode decomposition of the fused attention kernel: 0 = automatic — 256 persistent
 * workgroups on a wrap-around schedule of equal-cost runs of pair offsets; 1..200 = that many pair offsets per
 * item, one workgroup per item; 255 = one workgroup per item with the chunk chosen by the plan's launch model. */
int jodo_plan_create(const jodo_cfg* cfg, int B, int N, const int32_t* n_nodes_host, int max_chunk,
                     jodo_plan** out);
void jodo_plan_destroy(jodo_plan* plan);
/* bytes of the device-side descriptor tables / of the scratch workspace for this plan */
size_t jodo_plan_desc_bytes(const jodo_plan* plan);
size_t jodo_plan_workspace_bytes(const jodo_plan* plan);
/* number of floats in the time-modulation vector of one molecule (for sizing/debug) */
int64_t jodo_plan_mod_len(const jodo_plan* plan);
/* copy the descriptor tables into caller-owned device memory (async on stream) */
int jodo_plan_upload(jodo_plan* plan, void* desc_dev, void* stream);
/* plan statistics: out[0]=packed nodes, [1]=dense edge rows (sum n^2), [2]=directed edges
 * (sum n(n-1)), [3]=node strips, [4]=edge work items, [5]=max parts */
int jodo_plan_stats(const jodo_plan* plan, int64_t* out6);

/* One evaluation of the score network.
 *   packed_w / woff: packed weight blob (device) and its offset table (host, see enums above)
 *   xh [B,N,3+nd], edge_x [B,N,N,ch]; cond_x / cond_edge_x same shapes or NULL (first step);
 *   noise_level [B]; context [B,cond_ch] or NULL
 *   out_xh [B,N,3+nd], out_edge [B,N,N,ch] (fully written, zeros on padding)
 *   flags_dev: int32[8] device scratch; after the call [0] = NaN guard fired (mol_gnn.py:587-589),
 *              [1] reserved (0), [2] = all molecules shared one noise level, [3] = the self-conditioning positions
 *              were not all equal (0 => the batch-global first-step branch of :544 was taken),
 *              [4] = edge inputs were not symmetric (directed kernels used), [5] = STICKY count of calls in which
 *              the NaN guard fired (only ever incremented by the library; the caller clears it when it reads it),
 *              [6] = STICKY pin violations (JODO_OPT_PIN_*: bit 0 symmetry, bit 1 shared row), [7] reserved
 *   workspace: jodo_plan_workspace_bytes() bytes of device scratch
 *   dbg: optional device buffer for intermediates (tests) or NULL */
int jodo_dgt_forward(jodo_plan* plan, const void* desc_dev, const float* packed_w, const int64_t* woff,
                     int n_woff, const float* xh, const float* edge_x, const float* cond_x,
                     const float* cond_edge_x, const float* noise_level, const float* context,
                     float* out_xh, float* out_edge, int32_t* flags_dev, void* workspace, void* stream);

/* Executed-work model of a plan (measurement): fp32 flops that the kernels of ONE jodo_dgt_forward issue on the matrix pipe
 * (v_mfma_f32_32x32x2_f32 = 4096 flop per instruction), per launch class (enum jodo_prof_class below; array of
 * JODO_PROF_COUNT doubles), counted from the plan's work-item lists and the kernels' loop structure, padded lanes included.
 * uniform_t / symmetric = flags [2] / ![4] of the call being priced (they select the kernel variants that do the work).
 * This is the numerator of bench.py's roofline fraction: work the hardware executes, not the reference formulation's. */
int jodo_plan_work(const jodo_plan* plan, int uniform_t, int symmetric, double* mfma_flops_per_class);

/* Work-decomposition options of a plan (defaults are the measured-best settings).  These replace the
 * environment switches of round 1: the library reads no environment variables. */
enum jodo_plan_option {
    JODO_OPT_FUSE_NEXT_QKV = 0,   /* 1 (default): with >= 1024 node strips k_node_post of block l also produces block l + 1's
                                     q / k / v; 0: always a separate k_node_pre launch */
    JODO_OPT_DIR_SPLIT = 1,       /* 1 (default): pair-update items of a launch's sparsely filled last round get two workgroups,
                                     one per direction; 0: one workgroup per item */
    JODO_OPT_NODE_POST_WAVES = 2, /* 0 (default): automatic; 1 / 2 / 4: waves per strip for every strip of k_node_post;
                                   * 12 / 14: automatic split, 2 / 4 waves per strip of the remainder launch */
    JODO_OPT_ATTN_VARIANT = 3,    /* nf 256 pair attention kernel, weight residency / hand-over granularity: 0 (default), 1, 2, 3 (dgt_kernels_attn.h); other values are rejected */
    /* Path pinning.  By default every forward launches every kernel variant and device flags decide which ones work (no host
     * sync); the idle variants cost a dispatch each (~5 us, 24 per forward at 8 blocks).  Inside one sampling round the inputs
     * stay symmetric and the noise level stays shared, so a caller that has read flags [2] / [4] of one call may pin them for the
     * following calls on this plan: only the working variants are then launched.  A call that violates a pin (e.g. asymmetric
     * inputs under a symmetric pin) is detected on the device: flags[6] gets bit 0 (symmetry pin) / bit 1 (shared-row pin), sticky,
     * and its outputs are unspecified — callers check flags[6] when they read the NaN counter. */
    JODO_OPT_PIN_SYMMETRIC = 4,   /* 0 (default): decided per call on the device; 1: inputs are symmetric (pair kernels only);
                                     2: asymmetric (directed kernels only) */
    JODO_OPT_PIN_UNIFORM_T = 5,   /* 0 (default): per call; 1: one shared modulation row (folded pair update only); 2: per-molecule rows */
    JODO_OPT_ROT_STATS = 6,       /* 1 (default): under a shared modulation row and symmetric inputs the LayerNorm statistics of
                                     equi_update are taken in the rotated basis (Q P W_row h, Q P W_col h per node, the triangular
                                     L [e ; G] per pair, a per-molecule Gram tile — taken around the molecule's first atom, so that a
                                     common component of the two rows is gone before anything is squared — for the rest);
                                     0: from S = W_in [e ; G] itself; 2 (tests): rotated with the uncentred Gram tiles of round 3 */
    JODO_OPT_NODE_MIX = 7,        /* 1 (default): when k_node_post runs as full rounds + a remainder launch, the remainder launch also carries
                                     the k_node_ab items and Gram tiles of the strips the full rounds finished (they fill the SIMDs the
                                     remainder's cooperative workgroups leave idle); 0: separate launches */
    JODO_OPT_HEADS_MIX = 8,       /* 1 (default): under a symmetric pin the node head and the pair form of the edge head share one launch
                                     (node strips first: the edge head's short items fill the SIMDs the node head's last round leaves idle) */
    JODO_OPT_HALF_ROWS = 9,       /* 1 (default): under a symmetric pin, for molecules that fit an attention group (n <= 128), the embedding and the pair
                                     update write the edge state and the head inputs only for the row a pair's evaluating lane reads back —
                                     (i, i + d) of the circulant walk — instead of both mirror rows (the workspace copy of e is then half
                                     stale: jodo_debug_fetch(what = 1) is for unpinned calls) */
    JODO_OPT_PRE_EMBED = 10,      /* 1 (default; nf 256 tuned kernel set): the edge embedding shares a launch with the first block's q / k / v
                                     items (k_pre_embed: HBM-write-bound items beside matrix-bound ones); 0: a launch of its own */
    JODO_OPT_AB_PRE = 11,         /* 1 (default; nf 256 tuned kernel set, fewer than 1024 node strips): the next block's q / k / v items share
                                     the launch of this block's k_node_ab items and Gram tiles (k_node_ab_pre); 0: a k_node_pre launch per block */
    JODO_OPT_Z_SPLIT = 12,        /* 1 (default): when the LAST round of the pair update's launch (n_pitems mod 1024) has at most 256 items, they run as
                                   * workgroups of 4 waves that share the per-pair coord_mlp.0 output blocks (k_edge_update_sym<.., ZW = 4>); 2: also
                                   * 2 waves for 257 .. 512 items (measured slower on MI355X, kept for A/B runs); 0: one wave per item throughout */
    JODO_OPT_SPLIT_BF16 = 13,     /* 0 (default): every projection runs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  1 (OPT-IN, nf 256 / 384
                                   * unconditional models, pinned symmetric + shared-row paths, rotated statistics on, weights handed over with
                                   * jodo_plan_set_split_weights): the folded pair update runs its projections in the split-bf16 form — every
                                   * operand as hi + mid + lo bf16 terms, six v_mfma_f32_32x32x16_bf16 products per K = 16 step, fp32
                                   * accumulation: fp32-equivalent arithmetic (dropped terms <= 3 * 2^-26 relative; profiles/r06_split_gate.txt)
                                   * at 6/16 of the matrix cycles (csrc/dgt_kernels_split.h); with the tuned nf 256 kernel set also the node kernels
                                   * (k_node_post_split, k_node_ab_split: csrc/dgt_kernels_split_node.h).  Results differ from the default path in the
                                   * last bits; the default and every headline number stay exact fp32.  Ignored when a precondition fails.
                                   * 2 (experiments): also the fused attention kernel (k_edge_attn variants 4 + 5: two launches that share every
                                   * item by heads; nf 256 tuned set, plans without molecules above an attention group) — parity-tested, measured
                                   * 15 % SLOWER than the fp32 kernel on MI355X (513 against 447 us per block at QM9 B = 2500), DESIGN.md 4i */
    JODO_OPT_COUNT
};
int jodo_plan_set_option(jodo_plan* plan, int option, int value);

/* debug / tests: the pair-mode attention schedule of a plan.  out8 = items, their total pair offsets, partials per atom (max),
 * persistent schedule (0 / 1), and for a persistent schedule: smallest / largest slot load (offsets), most items in a slot, idle slots.
 * Also checks the item lists themselves (every group's pair offsets tiled exactly once, partial indices 0 .. parts - 1, parts as its
 * atoms expect) and returns an error if they are inconsistent. */
int jodo_debug_attn_schedule(const jodo_plan* plan, int64_t* out8);

/* debug: copy an internal per-block intermediate out of the workspace after a forward.
 * what: 0 = h [Nn,D], 1 = e [rows,De], 2 = pos [Nn,4] (raw, not centred), 3 = the attention partials
 * [Nn, parts, D] (unnormalised) of the last block run, 5 = the modulation row, 6 = q.  dst must hold the full array; returns element count via *count. */
int jodo_debug_fetch(jodo_plan* plan, const void* workspace, int what, float* dst_dev, int64_t* count,
                     void* stream);
/* limit the number of DGT blocks executed by jodo_dgt_forward (tests; <0 = all) */
int jodo_debug_set_max_blocks(jodo_plan* plan, int max_blocks);
/* force the directed (per directed edge) kernels even for symmetric inputs (tests; default 0: the
 * symmetric pair kernels are chosen on the device whenever edge_x / cond_edge_x are symmetric) */
int jodo_debug_set_force_directed(jodo_plan* plan, int on);
/* phase-cycle instrumentation of k_edge_update_sym (only in builds with -DJODO_PHASE_TIMING): device
 * buffer of 16 uint64 sums, [15] = number of instrumented waves; NULL disables */
int jodo_debug_set_timing_buffer(jodo_plan* plan, void* dev16xu64);

/* Per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * enable = 1 makes every subsequent jodo_dgt_forward bracket every launch class with events (two event
 * packets per class boundary cost ~10 us each: ~0.5 ms per forward at 8 blocks); enable = 2 brackets only the
 * dominant class JODO_PROF_EDGE_UPDATE (~0.1 ms); enable = 16 + c brackets only class c (what the roofline leg needs: the class that was largest in
 * this run's warm-up); jodo_profile_read synchronises those events and returns, per class, the summed
 * milliseconds and launch counts since the last read (arrays of JODO_PROF_COUNT), then resets. */
enum jodo_prof_class {
    JODO_PROF_PROLOGUE = 0, JODO_PROF_NODE_PRE, JODO_PROF_EDGE_ATTN, JODO_PROF_RESERVED3, JODO_PROF_RESERVED4,
    JODO_PROF_NODE_POST, JODO_PROF_EDGE_UPDATE, JODO_PROF_EPILOGUE, JODO_PROF_COUNT
};
int jodo_profile_enable(jodo_plan* plan, int enable);
int jodo_profile_read(jodo_plan* plan, float* ms_sum, int32_t* launches);

/* ---- caller-side kernels of the sampling loop ---------------------------
 * jodo_sampler_step  <- the update of AncestralSampler.sampling, sampling.py:536-589, with the noise
 *                       construction of models/utils.py:67-99 folded in:
 *                         mean = c_x * x + c_pred * pred;  next = mean + sigma * eps   (nodes and edges)
 *                       eps_pos [B,N,3], eps_feat [B,N,node_feats-3], eps_edge [B,edge_ch,N,N] are RAW
 *                       N(0,1) draws in the reference's shapes and draw order; masking, centre-of-mass
 *                       removal (positions) and lower-triangle mirroring (edges) happen in the kernel.
 *                       c_x, c_pred, sigma: posterior coefficients of the step (host scalars).
 *                       n_nodes_dev: int32[B] atom counts (prefix masks).  All outputs fully written.
 * jodo_decode        <- post_process sampling.py:53-97 + inverse scaler utils.py:71-105 + the
 *                       per-molecule slicing inputs of mol_process :12-32: positions [B,N,3] f32,
 *                       atom type [B,N] u8 (argmax), formal charge [B,N] i8 (round), bond type [B,N,N] u8
 *                       (compress_edge thresholds, or argmax+1 where any channel > 0.5); zeros on padding. */
int jodo_sampler_step(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, float c_x, float c_pred,
                      float sigma, const float* x, const float* edge_x, const float* pred, const float* edge_pred,
                      const float* eps_pos, const float* eps_feat, const float* eps_edge, float* x_next,
                      float* edge_next, float* x_mean, float* edge_mean, void* stream);
/* Graph-replayable form of the same update: the per-step scalars come from a device table
 * coef_tab_dev [steps][4] = (c_x, c_pred, sigma, noise_level) at row *step_dev, so ONE captured hipGraph of a
 * sampling step (jodo_step_begin -> jodo_dgt_forward -> noise draws -> jodo_sampler_step_tab -> jodo_step_end)
 * serves every step.  jodo_step_begin writes noise_level[b] = tab[*step][3] for the forward; jodo_step_end
 * increments *step_dev. */
int jodo_sampler_step_tab(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef_tab_dev,
                          const int32_t* step_dev, const float* x, const float* edge_x, const float* pred,
                          const float* edge_pred, const float* eps_pos, const float* eps_feat, const float* eps_edge,
                          float* x_next, float* edge_next, float* x_mean, float* edge_mean, void* stream);
/* Device-noise form of the same update ("one jodo_sampler_step kernel with Philox"): the three
 * normal draws of models/utils.py:67-99 are generated inside the kernel by a counter-based generator (Philox4x32-10 +
 * Box-Muller) keyed by `seed`, counter = (element, draw index, stream); masking, centre-of-mass removal and lower-triangle
 * mirroring as above.  draw index = `draw` (host scalars; coef_tab_dev = step_dev = NULL) or `draw` + *step_dev with the
 * coefficients from row *step_dev of coef_tab_dev (captured hipGraph).  Same distribution as the reference's draws, a
 * different stream: parity runs keep the replayed-draw form above.  Callers give every rank its own seed. */
int jodo_sampler_step_rng(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, float c_x, float c_pred,
                          float sigma, const float* coef_tab_dev, const int32_t* step_dev, uint64_t seed, uint32_t draw,
                          const float* x, const float* edge_x, const float* pred, const float* edge_pred, float* x_next,
                          float* edge_next, float* x_mean, float* edge_mean, void* stream);
int jodo_step_begin(int B, const float* coef_tab_dev, const int32_t* step_dev, float* noise_level_out, void* stream);
int jodo_step_end(int32_t* step_dev, void* stream);
/* jodo_dpm_update  <- one update of the hybrid DPM-Solver++ sampler, mix_dpm_solver.py: positions take the ancestral step
 *                     of :44-59 (pos_out = cx * x_pos + cp * PP + sigma * eps, eps_pos [B,N,3] RAW N(0,1) draws: masking and
 *                     centre-of-mass removal happen in the kernel; sigma = 0 for the last update of a round, :56), every other
 *                     channel of the node tensor and the whole edge tensor take the data-prediction update of :61-265
 *                         out = a * base - b * P - c * (c2 * (DA - DB))
 *                     which covers first order (c = 0, :84-85), single-step second order (:124-125, :137-146: P = DB =
 *                     prediction at the start, DA = prediction at s1), third order (:199-219, c < 0) and second-order
 *                     multistep (:253-260: P = DA = newest prediction, DB = the one before, c2 = 1 / r0).  PP = the
 *                     prediction whose position channels drive the ancestral step.  Tensors: node [B,N,3+nd], edge
 *                     [B,N,N,ch]; e* = the edge counterparts.  Coefficients {cx, cp, sigma, a, b, c, c2, noise_level}: either 8
 *                     host floats, or row *step_dev of a device table (row stride tab_stride floats, first column tab_col) so
 *                     that a captured hipGraph of an outer step replays for every step.
 * jodo_step_begin_at  noise_level[b] = table[*step][tab_col + 7]  (the graph-replayable counterpart of the per-evaluation
 *                     noise level, mix_dpm_solver.py:118,131) */
int jodo_dpm_update(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef8_host,
                    const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col, const float* x_pos,
                    const float* x_base, const float* edge_base, const float* P, const float* eP, const float* DA, const float* eDA,
                    const float* DB, const float* eDB, const float* PP, const float* eps_pos, float* x_out, float* edge_out,
                    void* stream);
/* jodo_dpm_update with the position noise drawn in the kernel (see jodo_sampler_step_rng): draw index = draw (+ *step_dev *
 * draw_mul with a device table, so that the two updates of a captured outer step use draws 2k and 2k + 1). */
int jodo_dpm_update_rng(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef8_host,
                        const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col, uint64_t seed,
                        uint32_t draw, uint32_t draw_mul, const float* x_pos, const float* x_base, const float* edge_base,
                        const float* P, const float* eP, const float* DA, const float* eDA, const float* DB, const float* eDB,
                        const float* PP, float* x_out, float* edge_out, void* stream);
int jodo_step_begin_at(int B, const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col,
                       float* noise_level_out, void* stream);
int jodo_decode(int B, int N, int atom_types, int include_fc, int edge_ch, int compress_edge, int centered,
                float pos_norm, float atom_norm, float fc_norm, float edge_norm, const int32_t* n_nodes_dev,
                const float* xh, const float* edge_x, float* pos_out, uint8_t* atom_type_out, int8_t* fc_out,
                uint8_t* edge_type_out, void* stream);

/* ---- training step: the whole network ------------------------------------------------------------------
 * jodo_train_forward   <- the grad-enabled model call of get_sde_graph_loss_fn, losses.py:335-343 (DGT_concat.forward /
 *                         Cond_DGT_concat.forward under model.train(): dropout in the four FFN positions, mol_gnn.py:262-268; NOT on the
 *                         attention weights — layers.py:179 is an identity because the block builds its TransMixLayer with the default
 *                         dropout = 0, mol_gnn.py:230-231) with every activation the backward needs kept in `workspace`
 * jodo_train_backward  <- loss.backward(), losses.py:109: d loss / d parameter for EVERY tensor of the state_dict, given
 *                         d loss / d out_xh and d loss / d out_edge (the loss itself, losses.py:350-385, stays host code)
 * Parameters are NOT packed here: params_dev[i] / grads_dev[i] are DEVICE pointers to contiguous fp32 tensors in PyTorch layout,
 * in the order of the `params` array given to jodo_train_create (names = state_dict keys, a leading "module." is ignored; `data`
 * of that array is not read, shapes are checked).  grads are fully written (not accumulated).  Inputs need no gradient (the
 * self-conditioning inputs are detached, losses.py:339).  The handle fixes the batch: B molecules with n_nodes[b] atoms, padded
 * width N of the dense tensors (same layouts as jodo_dgt_forward).  dropout_p = config.model.dropout under model.train(), 0 under
 * model.eval(); masks come from a counter-based generator keyed by `seed` (same law as torch's, a different stream) and are
 * regenerated, not stored: pass the same (dropout_p, seed) to the backward.  flags_out (optional): int32[8], [0] = NaN guard
 * fired (mol_gnn.py:587-589: positions zeroed, their gradient is zero), [3] = self-conditioning distances present (0 => the
 * first-step branch of :544).  workspace: jodo_train_workspace_bytes() bytes, kept by the caller between forward and backward.
 * desc_dev: jodo_train_desc_bytes() bytes of device memory filled by jodo_train_upload (synchronises the stream once). */
typedef struct jodo_train jodo_train;
int jodo_train_create(const jodo_cfg* cfg, int B, int N, const int32_t* n_nodes_host, const jodo_tensor* params, int n_params,
                      jodo_train** out);
void jodo_train_destroy(jodo_train* t);
size_t jodo_train_desc_bytes(const jodo_train* t);
size_t jodo_train_workspace_bytes(const jodo_train* t);
/* option 0 = fused per-edge forward chains (csrc/train_fused.hip: edge_emb -> LN -> lin_edge0 / 1; edge FFN; input_lin -> LN -> coord_mlp as
 * one strip-model kernel each, activations saved where the backward expects them): 1 (default) / 0 (one launch per operation);
 * option 1 = the input-gradient side of the same chains in the backward (the weight-gradient products stay GEMMs): 1 (default) / 0;
 * option 2 = 1 (default): forwards keep every activation a backward reads; 0: the following forwards will not be differentiated (the
 *            no-grad self-conditioning forward of a training step, losses.py:335-339) and skip those stores — jodo_train_backward after
 *            such a forward is undefined;
 * option 3 = 1 (default): the backward's weight-gradient products are queued and run in grouped launches (csrc/train_gemm.hip
 *            gemm_dw_group: same plans and arithmetic as one launch each, bit-identical gradients); 0: one launch per product;
 * option 4 = 1 (default where built): the attention of a block as one forward and two backward launches, one to four waves per atom (forward
 *            bit-identical to the op-by-op kernels); 0: scores | softmax | messages and their six backward kernels; 2: the one-wave-per-atom
 *            form of batches above 16 k atoms whatever the batch (tests);
 * option 5 = the Gaussian layer's backward as one wave-per-chunk pass: 1 (default) for batches of >= 4 096 chunks of 32 edge rows (it loses
 *            below), 0 never, 2 always (tests) */
int jodo_train_set_option(jodo_train* t, int option, int value);
/* tests: byte offset and element count of a kept activation inside the workspace; what 0 = hhat [Nn, D] (TransMixLayer's output,
 * layers.py:153), 1 = alpha [R, H] (softmax weights, layers.py:178) of block `layer` */
int jodo_train_debug_locate(const jodo_train* t, int what, int layer, size_t* byte_offset, size_t* count);
int jodo_train_upload(jodo_train* t, void* desc_dev, void* stream);
/* the host image jodo_train_upload copies (jodo_train_desc_bytes() bytes, owned by the handle): callers that stage it through pinned
 * memory themselves upload it without jodo_train_upload's stream synchronisation */
const void* jodo_train_desc_host(const jodo_train* t);
int jodo_train_forward(jodo_train* t, const void* desc_dev, const float* const* params_dev, int n_params, const float* xh,
                       const float* edge_x, const float* cond_x, const float* cond_edge_x, const float* noise_level,
                       const float* context, float dropout_p, uint64_t seed, float* out_xh, float* out_edge, int32_t* flags_out,
                       void* workspace, void* stream);
/* jodo_train_backward writes (not accumulates) every gradient: the grads_dev buffers are zeroed first, and where two of them lie less
 * than 16 bytes apart (sub-allocations of one buffer with 16-byte aligned slices, jodo_amd/optim.py slice_offsets) the bytes BETWEEN them
 * are zeroed as well — do not keep live data in such a gap. */
int jodo_train_backward(jodo_train* t, const void* desc_dev, const float* const* params_dev, float* const* grads_dev, int n_params,
                        const float* noise_level, const float* d_out_xh, const float* d_out_edge, float dropout_p, uint64_t seed,
                        void* workspace, void* stream);
/* the training path's fp32 MFMA GEMM on its own (tests):  C[M, N] (+)= op(A) op(B) (+ bias[N]);  tA = 0: A[m lda + k], 1: A[k lda + m];
 * tB = 0: B[k ldb + n], 1: B[n ldb + k];  ws / ws_floats: device scratch for split-K partial tiles (NULL: never split) */
int jodo_train_gemm(int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, int acc, float* ws, size_t ws_floats, void* stream);
/* the same with the fused epilogues the training step uses (tests): act 1: C = tanh(.); act 2: C = pre-activation, out2 = SiLU(.) laid out
 * like C; dbias (tA = 1, act = 0; C is accumulated into): dbias[m] += sum_k A(k, m), the bias gradient riding on a weight-gradient product */
int jodo_train_gemm_ex(int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                       const float* bias, int act, float* out2, float* dbias, float* ws, size_t ws_floats, void* stream);

/* ---- optimiser side of the training step on flat buffers (csrc/train_step.hip; host mirror: jodo_amd/optim.py) ----------------------
 * jodo_adam_step  <- optimizer.step() of the optimisers get_optimizer builds, losses.py:14-26: torch.optim.AdamW(amsgrad=True,
 *   weight_decay=1e-12) (decoupled = 1) or torch.optim.Adam (decoupled = 0: L2 term added to the gradient), torch's single-tensor
 *   formulas, on n contiguous floats at once: every parameter of the module a slice of p, its gradient the same slice of g, m / v / vmax
 *   the moment buffers (vmax only with amsgrad).  step = 1 for the first update (bias corrections 1 - beta^step, formed in double).
 * jodo_gradnorm_clip  <- gradient_clipping, losses.py:29-50, with the history of recent gradient norms (Queue, :53-72) on the device:
 *   state_dev = double[52] (history[50], count, next slot; the caller writes {3000, 0, ..., count = 1, slot = 1} once, :77-78);
 *   norm_dev = the gradient's 2-norm; writes coef_dev = min(1, allowed / (norm + 1e-6)), allowed_dev = min(1.5 mean + 2 std, max_grad),
 *   and pushes min(norm, allowed) to the history.  No host synchronisation: the caller multiplies the flat gradient by *coef_dev. */
int jodo_adam_step(int64_t n, float* p, const float* g, float* m, float* v, float* vmax, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, int decoupled, int amsgrad, void* stream);
int jodo_gradnorm_clip(const float* norm_dev, double* state_dev, double max_grad, float* coef_dev, float* allowed_dev, void* stream);
/* jodo_kabsch_rotations  <- kabsch_batch, losses.py:424-434, behind the covariance einsum: A_dev [B, 3, 3] (A = P^T Q per molecule) ->
 *   R_dev [B, 3, 3] = U diag(1, 1, sign det A) V^T of A = U S V^T.  Replaces torch.linalg.svd + det + diag_embed + einsum, whose error
 *   check synchronises the host with the device (the last synchronisation of a training step). */
int jodo_kabsch_rotations(int B, const float* A_dev, float* R_dev, void* stream);

const char* jodo_last_error(void);

/* measurement helper (synchronises, default stream): fp32 MFMA throughput of the box in TFLOP/s from a
 * register-only kernel; chains = 8 independent accumulator chains per wave (pipe ceiling) or 1 dependent chain
 * (what one projection block is); waves_per_simd x 1024 waves are launched.  sink_dev: any device float. */
int jodo_debug_mfma_peak(int iters, int chains, int waves_per_simd, float* sink_dev, float* tflops_out);

/* same, one dependent chain per wave (waves_per_simd x 1024 waves) with nv independent v_fma (and nt transcendentals)
 * issued after every MFMA: tells whether vector work hides under the matrix pipe.  (nv, nt) in
 * {(0,0),(4,0),(8,0),(12,0),(16,0),(4,1),(4,2)}. */
int jodo_debug_mfma_valu(int iters, int nv, int nt, int waves_per_simd, float* sink_dev, float* tflops_out);

/* ---- opt-in split-bf16 pair update (JODO_OPT_SPLIT_BF16; no reference counterpart: an alternative arithmetic form of
 * MultiCondEquiUpdate + the edge FFN, models/mol_gnn.py:71-94, :313-317) ----
 * jodo_dgt_split_size: bytes of the static weight tapes for this configuration — all of them, one block's PAIR tape (edge FFN, readout,
 *   triangular factor of the rotated statistics: k_edge_update_sym_split), one block's NODE tape (node2edge, node FFN, rotated W_row /
 *   W_col, readout, the next block's q / k / v: k_node_post_split; 0 when the width-generic node kernels run, jodo_cfg.layout = 1), one
 *   block's ATTENTION tapes (two cyclic tapes, one per launch of k_edge_attn variants 4 / 5: edge_emb, that launch's blocks of lin_edge0 and
 *   lin_edge1; 0 outside the tuned nf 256 set);
 *   JODO_ERR_UNSUPPORTED unless nf is 256 or 384 and cond_ch = 0.  Layout of the buffer: L pair tapes, then L node tapes, then L attention tapes.
 * jodo_dgt_pack_split_host: the tapes (hi | mid | lo bf16 terms of every weight, in consumption order) from the same named fp32 tensors
 *   jodo_dgt_pack_weights takes, into a host buffer.
 * jodo_plan_set_split_weights: device copy of that tape for this plan (caller-owned, must outlive the plan's forwards; NULL clears). */
int jodo_dgt_split_size(const jodo_cfg* cfg, size_t* total_bytes, size_t* pair_block_bytes, size_t* node_block_bytes, size_t* attn_block_bytes);
int jodo_dgt_pack_split_host(const jodo_cfg* cfg, const jodo_tensor* tensors, int n_tensors, void* host, size_t cap_bytes);
int jodo_plan_set_split_weights(jodo_plan* plan, const void* tape_dev, size_t bytes);

/* ---- gate experiments of the opt-in split-bf16 (three-term, fp32-equivalent) MFMA form (csrc/dgt_split.{h,hip}; no reference
 * counterpart; measurement helpers, never on the data path) ----
 * jodo_debug_mfma_bf16_valu: `chains` (1 | 2) dependent v_mfma_f32_32x32x16_bf16 chains per wave with nv in {0,2,4,6,8,16} (one chain)
 *   or {0,4,8,16} (two chains) independent v_fma issued behind every MFMA; returns the matrix TFLOP/s (bf16 flops).
 * jodo_debug_pack_split: a row-major [n_out, n_in] matrix -> its f32 packing and its split packing (hi / mid / lo bf16 terms), host.
 * jodo_debug_chain: y [rows, 256] = W x for x [rows, K], K in {128, 256}, in the strip model with streamed weights; mode 0 exact fp32
 *   MFMA, 1 split form (one accumulator), 2 split form (corrections in their own accumulator); tiles = item tiles per wave (2: split
 *   modes, K = 128).  iters > 1 repeats the projection in registers (timing; y then holds a digest); ms_out != NULL times the launch
 *   with HIP events (synchronises). */
int jodo_debug_mfma_bf16_valu(int iters, int nv, int chains, int waves_per_simd, float* sink_dev, float* tflops_out);
int jodo_debug_pack_split(const float* W, int n_out, int n_in, float* f32_packed_host, void* split_packed_host);
int jodo_debug_chain(int mode, int K, int tiles, const float* x, int rows, const float* wf, const void* wsp, float* y, int iters,
                     float* ms_out, void* stream);

int jodo_debug_mlp(const float* x, int rows, const float* w1, const float* b1, const float* w2,
                   const float* b2, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif
