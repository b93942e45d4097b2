// Edge kernels of a DGT block: launch code + instantiations (see dgt_launch.h for why this is its own translation unit).
#include "dgt_kernels_attn.h"
#include "dgt_kernels_wide.h"
#include "dgt_launch.h"

using namespace jd;

namespace {

// Pair update: every full round of 1024 one-iteration items (one per SIMD) in one launch; the items of the last,
// sparsely filled round in a second launch with two workgroups per item, one direction each (both recompute the
// shared trunk: item time x 0.64).  At QM9 B = 2500 a launch has 12 666 items = 12 full rounds + 378.
template <int D>
int launch_update_sym(jodo_plan* p, hipStream_t st, KArgs& A) {
    const DgtDims& d = p->dims;
    const int full = (p->n_pitems / 1024) * 1024, rem = p->n_pitems - full;
    // (nf = 256: the directions share coord_mlp.0 and their tails are short vector work — nothing to split)
    const bool split = D != 256 && p->opt[JODO_OPT_DIR_SPLIT] != 0 && rem > 0 && rem <= 512;
    const int n1 = split ? full : p->n_pitems;
    A.item0 = 0; A.dir_split = 0;
    // JODO_OPT_PIN_UNIFORM_T: 1 = every call shares one modulation row (only the folded variant is launched), 2 = never
    const bool run_plain = p->opt[JODO_OPT_PIN_UNIFORM_T] != 1, run_fold = p->opt[JODO_OPT_PIN_UNIFORM_T] != 2 && d.cond_ch == 0;
    if (run_plain && n1 > 0) { if (d.r == 2) LAUNCH((wide::k_edge_update_sym<D, 2>), n1, 64, A); else LAUNCH((wide::k_edge_update_sym<D, 4>), n1, 64, A); }
    // shared modulation row (device flag): the variant with the folded coord_mlp.0 does the work instead (never split)
    // (A.rot: LayerNorm statistics in the rotated basis the node kernels of this forward wrote — decided by the launcher, not a flag)
    if (run_fold && A.rot) { if (d.r == 2) LAUNCH((wide::k_edge_update_sym<D, 2, true, true>), p->n_pitems, 64, A); else LAUNCH((wide::k_edge_update_sym<D, 4, true, true>), p->n_pitems, 64, A); }
    if (run_fold && !A.rot) { if (d.r == 2) LAUNCH((wide::k_edge_update_sym<D, 2, true>), p->n_pitems, 64, A); else LAUNCH((wide::k_edge_update_sym<D, 4, true>), p->n_pitems, 64, A); }
    if (run_plain && split) {
        A.item0 = full; A.dir_split = 1;
        if (d.r == 2) LAUNCH((wide::k_edge_update_sym<D, 2>), 2 * rem, 64, A); else LAUNCH((wide::k_edge_update_sym<D, 4>), 2 * rem, 64, A);
        A.item0 = 0; A.dir_split = 0;
    }
    return JODO_OK;
}


template <int D, bool TUNED>
int launch_attn(jodo_plan* p, hipStream_t st, KArgs& A, bool pin_pair, bool pin_dir) {
    if (p->n_aitems > 0 && !pin_dir) {
        const int var = TUNED ? p->opt[JODO_OPT_ATTN_VARIANT] : 0;
        const int grid = p->a_persist ? JODO_ATT_SLOTS : p->n_aitems;          // persistent: one workgroup per slot of the plan's schedule
        if (var == 1) LAUNCH((k_edge_attn<D, !TUNED, true, TUNED ? 1 : 0>), grid, ATT_WAVES * 64, A);
        else if (var == 2) LAUNCH((k_edge_attn<D, !TUNED, true, TUNED ? 2 : 0>), grid, ATT_WAVES * 64, A);
        else if (var == 3) LAUNCH((k_edge_attn<D, !TUNED, true, TUNED ? 3 : 0>), grid, ATT_WAVES * 64, A);
        else LAUNCH((k_edge_attn<D, !TUNED, true, 0>), grid, ATT_WAVES * 64, A);
    }
    if (p->n_aditems > 0 && (!pin_pair || p->has_big)) LAUNCH((k_edge_attn<D, !TUNED, false>), p->n_aditems, ATT_WAVES * 64, A);
    return JODO_OK;
}

template <int D>
int launch_update(jodo_plan* p, hipStream_t st, KArgs& A, bool pin_pair, bool pin_dir) {
    const DgtDims& d = p->dims;
    if (p->n_pitems > 0 && !pin_dir) {
        int rc = launch_update_sym<D>(p, st, A);
        if (rc) return rc;
    }
    if (!pin_pair) { if (d.r == 2) LAUNCH((wide::k_edge_update<D, 2>), p->n_items, 64, A); else LAUNCH((wide::k_edge_update<D, 4>), p->n_items, 64, A); }
    return JODO_OK;
}

}  // namespace

int jd_launch_edge_attn(jodo_plan* p, hipStream_t st, KArgs& A, int D, bool tuned, bool pin_pair, bool pin_dir) {
    if (D == 256) return tuned ? launch_attn<256, true>(p, st, A, pin_pair, pin_dir) : launch_attn<256, false>(p, st, A, pin_pair, pin_dir);
    if (D == 128) return launch_attn<128, false>(p, st, A, pin_pair, pin_dir);
    return launch_attn<384, false>(p, st, A, pin_pair, pin_dir);
}

int jd_launch_edge_update(jodo_plan* p, hipStream_t st, KArgs& A, int D, bool pin_pair, bool pin_dir) {
    if (D == 128) return launch_update<128>(p, st, A, pin_pair, pin_dir);
    return D == 256 ? launch_update<256>(p, st, A, pin_pair, pin_dir) : launch_update<384>(p, st, A, pin_pair, pin_dir);
}
