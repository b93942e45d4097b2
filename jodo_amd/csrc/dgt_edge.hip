// Edge kernels of a DGT block: launch code + instantiations (see dgt_launch.h for why this is its own translation unit).
#include "dgt_kernels_attn.h"
#include "dgt_kernels_wide.h"
#include "dgt_kernels_split.h"
#include "dgt_launch.h"

using namespace jd;

namespace {

// Pair update: one-iteration items, one wave each, every full round of 1024 (one per SIMD) in one launch; the items of the last,
// sparsely filled round in a second launch that gives each item several waves:
//   * hoisted forms (nf 256; any width under a shared modulation row): ZW = 2 / 4 waves share the output blocks of the per-pair
//     coord_mlp.0 projection (JODO_OPT_Z_SPLIT; item time x 0.73 / 0.60 folded, x 0.72 / 0.58 unfolded);
//   * nf 384 without a shared row: two workgroups per item, one direction each (JODO_OPT_DIR_SPLIT; item time x 0.64).
// QM9 B = 2500: 12 666 items = 12 rounds + 378; conditional B = 1250: 6 295 = 6 rounds + 151.
template <int D, int R, bool FOLD, bool ROT>
void launch_sym_variant(hipStream_t st, KArgs& A, int n_items, int zw) {
    if (!(D == 256 || FOLD)) zw = 1;                      // (the un-hoisted form has no per-pair Z to deal out)
    const int full = zw > 1 ? (n_items / 1024) * 1024 : n_items, rem = n_items - full;
    A.item0 = 0; A.dir_split = 0;
#if JODO_X_UPD_PERS
    if (FOLD && full > 0) {                               // persistent: one workgroup per SIMD walks the items block index + k * 1024
        A.pers_n = full;
        hipLaunchKernelGGL((wide::k_edge_update_sym<D, R, FOLD, ROT>), dim3(full < 1024 ? full : 1024), dim3(64), 0, st, A);
        A.pers_n = 0;
    } else
#endif
    if (full > 0) hipLaunchKernelGGL((wide::k_edge_update_sym<D, R, FOLD, ROT>), dim3(full), dim3(64), 0, st, A);
    if (rem > 0) {
        A.item0 = full;
        if constexpr (D == 256 || FOLD) {
            if (zw == 4) hipLaunchKernelGGL((wide::k_edge_update_sym<D, R, FOLD, ROT, 4>), dim3(rem), dim3(256), 0, st, A);
            else hipLaunchKernelGGL((wide::k_edge_update_sym<D, R, FOLD, ROT, 2>), dim3(rem), dim3(128), 0, st, A);
        }
        A.item0 = 0;
    }
}

template <int D>
int launch_update_sym(jodo_plan* p, hipStream_t st, KArgs& A) {
    const DgtDims& d = p->dims;
    const int full = (p->n_pitems / 1024) * 1024, rem = p->n_pitems - full;
    // waves per item of the last round (hoisted forms).  Measured on MI355X (round 5, profiles/r05b_ab_zsplit.txt): FOUR waves pay
    // when the last round has at most 256 items (conditional B = 1250, 6 295 items = 6 rounds + 151: pair update 4.34 -> 4.28
    // ms/step); TWO waves (257 .. 512 items) cost more than they save — QM9 B = 2500 (378 left) 4.69 -> 4.85, GEOM B = 512 (310)
    // 8.63 -> 8.82, nf 384 (423) 42.06 -> 42.55 ms/step: the dispatcher already overlaps the last "round" with the drift of the
    // earlier ones, and a second launch puts a barrier there.  Option value 2 keeps the two-wave form selectable for A/B runs.
    const int zopt = p->opt[JODO_OPT_Z_SPLIT];
    const int zw = (zopt != 0 && rem > 0) ? (rem <= 256 ? 4 : ((zopt == 2 && rem <= 512) ? 2 : 1)) : 1;
    // (nf = 256: the directions share coord_mlp.0 and their tails are short vector work — nothing to split by direction)
    const bool split = D != 256 && p->opt[JODO_OPT_DIR_SPLIT] != 0 && rem > 0 && rem <= 512;
    const int n1 = split ? full : p->n_pitems;
    A.item0 = 0; A.dir_split = 0;
    // JODO_OPT_PIN_UNIFORM_T: 1 = every call shares one modulation row (only the folded variant is launched), 2 = never
    const bool run_plain = p->opt[JODO_OPT_PIN_UNIFORM_T] != 1, run_fold = p->opt[JODO_OPT_PIN_UNIFORM_T] != 2 && d.cond_ch == 0;
    if (run_plain && n1 > 0) {
        if constexpr (D == 256) { if (d.r == 2) launch_sym_variant<D, 2, false, false>(st, A, n1, zw); else launch_sym_variant<D, 4, false, false>(st, A, n1, zw); }
        else { if (d.r == 2) LAUNCH((wide::k_edge_update_sym<D, 2>), n1, 64, A); else LAUNCH((wide::k_edge_update_sym<D, 4>), n1, 64, A); }
    }
    // shared modulation row (device flag): the variant with the folded coord_mlp.0 does the work instead
    // (A.rot: LayerNorm statistics in the rotated basis the node kernels of this forward wrote — decided by the launcher, not a flag)
    // opt-in split-bf16 form (JODO_OPT_SPLIT_BF16; jodo_dgt_forward checked the preconditions and set A.wsplit): four items per workgroup
    bool split_ran = false;
    if constexpr (D == 256 || D == 384) {
        if (run_fold && !run_plain && A.rot == 1 && A.wsplit && A.mfold_s) {
            const int wgs = (p->n_pitems + 31) / 32 * 8;
            if (d.r == 2) hipLaunchKernelGGL((split::k_edge_update_sym_split<D, 2>), dim3(wgs), dim3(split::SPLIT_WAVES * 64), 0, st, A);
            else hipLaunchKernelGGL((split::k_edge_update_sym_split<D, 4>), dim3(wgs), dim3(split::SPLIT_WAVES * 64), 0, st, A);
            split_ran = true;
        }
    }
    if (run_fold && A.rot && !split_ran) { if (d.r == 2) launch_sym_variant<D, 2, true, true>(st, A, p->n_pitems, zw); else launch_sym_variant<D, 4, true, true>(st, A, p->n_pitems, zw); }
    if (run_fold && !A.rot) { if (d.r == 2) launch_sym_variant<D, 2, true, false>(st, A, p->n_pitems, zw); else launch_sym_variant<D, 4, true, false>(st, A, p->n_pitems, zw); }
    if (run_plain && split) {
        A.item0 = full; A.dir_split = 1;
        if (d.r == 2) LAUNCH((wide::k_edge_update_sym<D, 2>), 2 * rem, 64, A); else LAUNCH((wide::k_edge_update_sym<D, 4>), 2 * rem, 64, A);
        A.item0 = 0; A.dir_split = 0;
    }
    return jodo_check_launch("k_edge_update_sym");
}


template <int D, bool TUNED>
int launch_attn(jodo_plan* p, hipStream_t st, KArgs& A, bool pin_pair, bool pin_dir) {
    if (p->n_aitems > 0 && !pin_dir) {
        const int var = TUNED ? p->opt[JODO_OPT_ATTN_VARIANT] : 0;
        const int grid = p->a_persist ? JODO_ATT_SLOTS : p->n_aitems;          // persistent: one workgroup per slot of the plan's schedule
        // opt-in split-bf16 form (JODO_OPT_SPLIT_BF16 = 2; jodo_dgt_forward checked the preconditions and set A.wsplit_attn): two launches,
        // each half of the heads of every item (VAR 4: message blocks 0 .. 3, VAR 5: 4 .. 7; disjoint halves of every partial)
        if (TUNED && A.wsplit_attn && pin_pair && p->n_ai_dir == 0) {
            if constexpr (TUNED) {
                LAUNCH((k_edge_attn<D, false, true, 4>), grid, ATT_WAVES * 64, A);
                LAUNCH((k_edge_attn<D, false, true, 5>), grid, ATT_WAVES * 64, A);
            }
        }
        else if (var == 1) LAUNCH((k_edge_attn<D, !TUNED, true, TUNED ? 1 : 0>), grid, ATT_WAVES * 64, A);
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
