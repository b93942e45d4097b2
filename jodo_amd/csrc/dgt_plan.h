// Host-side execution plan shared by dgt_plan.cpp (builder) and dgt_forward.hip (launcher).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/jodo_hip.h"

// workgroups of the persistent pair-mode attention launch: one per CU of an MI355X (each holds a CU: 4 waves of ~480 registers,
// 96 + 40 KiB of LDS); a device with fewer CUs runs them in rounds — slower, never wrong
constexpr int JODO_ATT_SLOTS = 256;

struct DgtDims {
    int D, De, T, L, H, XH, SH, SC, C, r, nd, ch, cond_ch;
    int QKP;        // padded q/k width in floats (slot order)
    int ndp;        // padded node input width (2*nd rounded up to 8)
    int einp;       // padded raw edge input width (2*ch rounded up to 8)
    int cnp, cep;   // padded per-block readout widths (node 64, edge 16)
    int KNH, KEH;   // head-MLP input widths  D + L*cnp,  De + L*cep
    int MB;         // modulation floats per block
    int64_t Mtot;   // modulation floats per molecule
    float cutoff, edge_th;
    int wide;       // 1 = width-generic kernel set (dgt_kernels_wide.h) and its weight layout
};

// device views into the descriptor buffer (all int32)
struct PlanDev {
    const int* node_b;     // packed node -> original batch index
    const int* node_i;     // index inside its molecule
    const int* node_n;     // atoms in its molecule (0 = padding lane)
    const int* node_noff;  // first packed node of its molecule
    const int* node_eoff;  // first dense edge row of its molecule
    const int* orig_n;     // [B] original b -> n
    const int* orig_noff;  // [B] original b -> first packed node
    const int* orig_eoff;  // [B] original b -> first dense edge row
    const int* item_strip; // edge work items
    const int* item_t0;
    const int* item_t1;
    const int* item_part;
    const int* strip_parts; // [n_strips] number of parts (work items) of each node strip
    const int* pitem_strip; // pair work items (symmetric path): strip, [d0, d1) over circulant offsets d = t + 1
    const int* pitem_t0;
    const int* pitem_t1;
    // fused attention kernel (k_edge_attn): groups of whole molecules, up to 128 lanes each, one 4-wave workgroup per item
    const int* ag_node;     // [n_agroups][128] packed node of lane ln (-1 = idle lane)
    const int* ai_group;    // pair-mode items: group, offsets d = t + 1 for t in [t0, t1), partial index
    const int* ai_t0;
    const int* ai_t1;
    const int* ai_part;
    const int* ai_dir;      // 1 = item of a molecule that spans several groups, carried by the pair launch in directed mode: sources t in [t0, t1)
    const int* ad_group;    // directed-mode items (asymmetric inputs; molecules larger than a group): sources t in [t0, t1)
    const int* ad_t0;
    const int* ad_t1;
    const int* ad_part;
    const int* ad_big;      // 1 = group of a molecule that spans several groups and the pair launch does not carry it (fixed-chunk plans): the directed launch works on it even for symmetric inputs
    const int* anode_parts; // [Nn_pad] number of attention partials of a node
    const int* aw_off;      // persistent pair-mode launch: items [aw_off[w], aw_off[w + 1]) belong to workgroup w (JODO_ATT_SLOTS + 1 entries)
    int a_persist;
    const int* ut_rows;     // [n_ut_pad][2] dense edge row (a, c) with a < c and its mirror (c, a); -1 = padding (edge head, symmetric inputs)
    int n_ut_pad;
    const int* gt_sa;       // Gram tiles of the rotated statistics (k_node_gram): strip of the row atoms a, strip of the column atoms c
    const int* gt_sc;
    int n_gtiles;
    int n_agroups, n_aitems, n_aditems, amax_parts;
    int Nn, Nn_pad, n_strips, n_items, n_pitems, B, N, max_parts;
    int64_t rows;
};

struct WsLayout {   // byte offsets into the workspace
    size_t hid1, temb, tembs, mods, condh, condh2;
    size_t pos0, pos1, dpos, cpos, feat, h, hhat, astat, q, k, v, n2e, wrow, wcol, ua, ub, rmean, mfold, ffold, ahid, apred;
    size_t eflag, e, e2, ehid, epred, dposE, gramE;
    size_t ffold_s;                  // split-bf16 image of ffold: [L][D / 32][D / 16][3][64][8] bf16
    size_t mfold_s;                  // split-bf16 image of mfold (JODO_OPT_SPLIT_BF16): [L][D / 32][2 De / 16][3][64][8] bf16
    size_t total;
};

struct jodo_plan {
    jodo_cfg cfg;
    DgtDims dims;
    int B, N, Nn, Nn_pad, n_strips, n_items, n_pitems, max_parts;
    int n_agroups, n_aitems, n_aditems, amax_parts, n_ut_pad, n_gtiles;
    int a_persist;                   // pair-mode attention items are scheduled onto JODO_ATT_SLOTS persistent workgroups (off_aw_off)
    size_t off_aw_off;
    int has_big;                     // some molecule spans several attention groups (n > 128) AND the directed launch must serve it for symmetric inputs (ad_big)
    size_t off_ag_node, off_ai_group, off_ai_t0, off_ai_t1, off_ai_part, off_ai_dir, off_ad_group, off_ad_t0, off_ad_t1, off_ad_part, off_ad_big, off_anode_parts, off_ut_rows, off_gt_sa, off_gt_sc;
    int64_t rows, dir_edges;
    std::vector<int32_t> desc;       // concatenated descriptor tables
    size_t off_node_b, off_node_i, off_node_n, off_node_noff, off_node_eoff, off_orig_n, off_orig_noff,
        off_orig_eoff, off_item_strip, off_item_t0, off_item_t1, off_item_part, off_strip_parts, off_pitem_strip, off_pitem_t0, off_pitem_t1;   // in int32 elements
    WsLayout ws;
    // profiling (jodo_profile_*): pairs of events per launch class, recorded on the launch stream
    int prof_enabled;
    std::vector<void*> prof_ev;      // hipEvent_t start/stop pairs in record order
    std::vector<int> prof_cls;       // class of each pair
    std::vector<void*> prof_pool;    // reusable events
    int force_directed;              // debug: always take the directed (non-pair) kernels
    void* dbg_timing;                // debug: device buffer of 16 x u64 phase-cycle sums (or null)
    int max_blocks;                  // debug: limit blocks executed (<0 = all)
    int last_pos_buf;                // debug: which pos buffer holds the latest positions
    int last_e_buf;                  // debug: which edge-state buffer holds the latest state
    int opt[JODO_OPT_COUNT];         // jodo_plan_set_option values
    const void* split_w;             // jodo_plan_set_split_weights: device tape of the split-bf16 pair update (caller-owned) and its size
    size_t split_bytes;
    int n_ai_dir;                    // directed-mode items carried by the pair-mode attention launch (molecules above an attention group)
    int pitems_single;               // every pair-update item is exactly one circulant offset (what the split kernel's lock-step workgroups need)
    int gt_cache_full, gt_cache_count;   // Gram tiles whose strips all lie below gt_cache_full (tiles are sorted by their larger strip)
};

int dgt_dims_from_cfg(const jodo_cfg* cfg, DgtDims* d);
