// Launch helpers shared by the two device translation units of the forward:
//   dgt_forward.hip  node kernels, embeddings, heads, prologue / epilogue (default code generation)
//   dgt_edge.hip     the two edge kernels of a block — fused attention and pair / directed update — built with
//                    -mllvm -amdgpu-mfma-vgpr-form: their accumulators are consumed by vector code right behind every MFMA
//                    block (tanh, LayerNorm statistics, SiLU tails), and with MFMA results in VGPRs the v_accvgpr_read per
//                    value disappears (measured on MI355X: attention 4.07 -> 3.96 ms/step at QM9 B = 2500, pair update at
//                    nf = 384 58.3 -> 57.1 ms/step); the node kernels and the heads measured slower with it (28 B of scratch in
//                    k_node_post), hence two units.
#pragma once
#include "dgt_kernels_common.h"
#include "jodo_hip_internal.h"

#define LAUNCH(kern, grid, block, ...)                                  \
    do {                                                                \
        auto kf_ = kern;                                                \
        hipLaunchKernelGGL(kf_, dim3(grid), dim3(block), 0, st, __VA_ARGS__); \
        int rc_ = jodo_check_launch(#kern);                             \
        if (rc_ != JODO_OK) return rc_;                                 \
    } while (0)

// fused attention of the current block (A.layer, A.wb set): pair-mode items for symmetric inputs, directed-mode items otherwise
// and for molecules larger than a group; pin_pair / pin_dir: JODO_OPT_PIN_SYMMETRIC
int jd_launch_edge_attn(jodo_plan* p, hipStream_t st, KArgs& A, int D, bool tuned, bool pin_pair, bool pin_dir);
// edge update of the current block: pair kernels (+ folded variant under a shared modulation row) and / or the directed kernel
int jd_launch_edge_update(jodo_plan* p, hipStream_t st, KArgs& A, int D, bool pin_pair, bool pin_dir);
