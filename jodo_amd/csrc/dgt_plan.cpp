// Plan builder: turns the batch's atom counts into the packed, size-sorted layout and the work-item
// lists the kernels iterate over.  Replaces the reference's per-forward dense->sparse conversion
// (models/mol_gnn.py:512-514) with a once-per-batch host computation.
#include <algorithm>
#include <numeric>
#include <utility>
#include <new>
#include "dgt_plan.h"
#include "jodo_hip_internal.h"

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// XCD-aware work-item order (MI355X: 8 XCDs with private L2s; workgroup w is observed to run on XCD
// w % 8).  All items of a node strip re-read the same few node rows (q/k/v, W_row h, W_col h of one or
// two molecules); keeping a strip's items on ONE XCD turns those gathers into L2 hits instead of
// HBM/MALL round trips (~2 us each).  Items are queued per class strip % 8 and dealt to workgroups
// round-robin, `group` items per workgroup; a drained class borrows from the next one (speed only —
// correctness never depends on placement).
static std::vector<int> xcd_order(const std::vector<int32_t>& strip, int group) {
    const int n = (int)strip.size();
    std::vector<std::vector<int>> q(8);
    for (int i = 0; i < n; ++i) q[strip[i] & 7].push_back(i);
    size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int> out;
    out.reserve(n);
    int wg = 0;
    while ((int)out.size() < n) {
        int c = wg & 7, tries = 0;
        while (pos[c] >= q[c].size() && tries < 8) { c = (c + 1) & 7; ++tries; }
        for (int k = 0; k < group && pos[c] < q[c].size(); ++k) out.push_back(q[c][pos[c]++]);
        ++wg;
    }
    return out;
}

static void permute(std::vector<int32_t>& a, const std::vector<int>& ord) {
    std::vector<int32_t> b(a.size());
    for (size_t i = 0; i < ord.size(); ++i) b[i] = a[ord[i]];
    a.swap(b);
}

int dgt_dims_from_cfg(const jodo_cfg* c, DgtDims* d) {
    if (c->nf != 128 && c->nf != 256 && c->nf != 384)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "nf=%d: kernels are built for nf=128, nf=256 and nf=384", c->nf);
    if (c->layout != 0 && c->layout != 1) return jodo_set_error(JODO_ERR_ARG, "layout=%d (0 or 1)", c->layout);
    if (c->n_heads != 16 || c->n_extra != 2)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "n_heads=%d n_extra_heads=%d: kernels are built for 16/2", c->n_heads, c->n_extra);
    if (c->mlp_ratio != 2 && c->mlp_ratio != 4)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "mlp_ratio=%d: supported 2, 4", c->mlp_ratio);
    if (c->n_layers < 1 || c->n_layers > 32) return jodo_set_error(JODO_ERR_ARG, "n_layers=%d", c->n_layers);
    if (c->in_node_dim < 1 || 2 * c->in_node_dim > 64) return jodo_set_error(JODO_ERR_ARG, "in_node_dim=%d", c->in_node_dim);
    if (c->edge_ch < 1 || c->edge_ch > 4) return jodo_set_error(JODO_ERR_ARG, "edge_ch=%d (1..4)", c->edge_ch);
    if (c->cond_ch < 0 || c->cond_ch > 8) return jodo_set_error(JODO_ERR_ARG, "cond_ch=%d", c->cond_ch);
    d->D = c->nf; d->De = c->nf / 4; d->T = c->nf * 4; d->L = c->n_layers; d->H = c->n_heads; d->XH = c->n_extra;
    d->SH = d->H - d->XH; d->C = d->D / d->H; d->SC = (d->H * d->C) / d->SH; d->r = c->mlp_ratio;
    d->nd = c->in_node_dim; d->ch = c->edge_ch; d->cond_ch = c->cond_ch;
    // per-block readout widths 2 D / L and 2 De / L (models/mol_gnn.py:452-455), padded to whole 32-row blocks / to 16 or 32
    const int cn = (2 * d->D) / d->L, ce = (2 * d->De) / d->L;
    d->cnp = std::max(d->D / 4, (cn + 31) / 32 * 32);
    d->cep = std::max((d->D / 16 + 15) / 16 * 16, (ce + 15) / 16 * 16);            // 64 / 16 at nf 256 (L >= 8), 96 / 32 at nf 384
    if (d->cep > 32) return jodo_set_error(JODO_ERR_UNSUPPORTED, "n_layers=%d at nf=%d: edge readout width %d beyond one 32-row block", d->L, d->D, ce);
    // the nf = 256 node kernels of dgt_kernels_node.h are written for the 64-wide node readout
    d->wide = (c->nf != 256 || c->layout == 1 || d->cnp != 64) ? 1 : 0;
    int tail = d->SC - 16;
    d->QKP = d->wide ? d->SH * 32 : (d->SH / 2 + (tail + 1) / 2) * 32;      // wide: one head per 32-row block
    d->ndp = (2 * d->nd + 7) / 8 * 8;
    d->einp = (2 * d->ch + 7) / 8 * 8;
    d->KNH = d->D + d->L * d->cnp; d->KEH = d->De + d->L * d->cep;
    if (d->KEH % 32 != 0) return jodo_set_error(JODO_ERR_UNSUPPORTED, "edge head width %d not a multiple of 32", d->KEH);
    d->MB = 6 * d->D + 6 * d->De + 2 * d->D + 32 + 2 * d->D;     // node | edge | equi (shift, scale) | gbf | W0 (1 + scale) | W0 shift + b0
    d->Mtot = 32 + (int64_t)d->L * d->MB;
    d->cutoff = c->spatial_cut_off; d->edge_th = c->edge_quan_th;
    return JODO_OK;
}

extern "C" int jodo_plan_create(const jodo_cfg* cfg, int B, int N, const int32_t* n_nodes, int max_chunk,
                                jodo_plan** out) {
    if (!cfg || !n_nodes || !out || B <= 0 || N <= 0) return jodo_set_error(JODO_ERR_ARG, "plan_create: bad argument");
    jodo_plan* p = new (std::nothrow) jodo_plan();
    if (!p) return jodo_set_error(JODO_ERR_ARG, "plan_create: out of host memory");
    p->cfg = *cfg;
    int rc = dgt_dims_from_cfg(cfg, &p->dims);
    if (rc != JODO_OK) { delete p; return rc; }
    for (int b = 0; b < B; ++b)
        if (n_nodes[b] < 1 || n_nodes[b] > N) {
            delete p;
            return jodo_set_error(JODO_ERR_ARG, "plan_create: n_nodes[%d]=%d outside [1,%d]", b, n_nodes[b], N);
        }
    // bits 0-15: sources per directed work item (default 8); bits 16-23: pair offsets per item of the pair UPDATE
    // kernel (default 1: long iterations, balance matters most); bits 24-31: pair offsets per item of the fused
    // attention kernel (default 6: its per-item cost — 96 KiB of weights staged into LDS, 1 KiB of partial per atom —
    // must be amortised, see below)
    int pair_chunk = (max_chunk >> 16) & 0xff, spair_chunk = (max_chunk >> 24) & 0xff;
    max_chunk &= 0xffff;
    if (max_chunk <= 0) {
        // automatic: 8 sources per item amortises the per-item prologue, but a small batch then has fewer
        // items than the chip has SIMDs (1024) and the attention kernels become latency-bound
        // (QM9 cond B = 313 on MI355X: k_edge_msgs 1.37 -> 0.52 ms/step going from 8 to 2); shrink the
        // chunk until there are >= 2 items per SIMD
        int64_t strip_nmax_sum = 0;                       // sum over strips of the largest molecule in the strip
        {
            std::vector<int> sorted(n_nodes, n_nodes + B);
            std::sort(sorted.begin(), sorted.end(), [](int a, int b) { return a > b; });
            int64_t v = 0;
            int strip = -1;
            for (int m = 0; m < B; ++m) {
                const int first = (int)(v / 32), last = (int)((v + sorted[m] - 1) / 32);
                for (int s = std::max(first, strip + 1); s <= last; ++s) strip_nmax_sum += sorted[m];   // sizes descend
                strip = std::max(strip, last);
                v += sorted[m];
            }
        }
        max_chunk = 8;
        while (max_chunk > 2 && strip_nmax_sum / max_chunk < 2048) --max_chunk;
    }
    if (pair_chunk <= 0) pair_chunk = 1;     // measured best on MI355X (QM9 B=2500: 26.7 ms/step vs 27.7 at 2)
    const bool spair_auto = spair_chunk <= 0;
    p->B = B; p->N = N; p->max_blocks = -1; p->last_pos_buf = 0; p->last_e_buf = 0;
    p->opt[JODO_OPT_FUSE_NEXT_QKV] = 1; p->opt[JODO_OPT_DIR_SPLIT] = 1; p->opt[JODO_OPT_NODE_POST_WAVES] = 0; p->opt[JODO_OPT_ATTN_VARIANT] = 0; p->opt[JODO_OPT_HEADS_MIX] = 1; p->opt[JODO_OPT_HALF_ROWS] = 1; p->opt[JODO_OPT_PRE_EMBED] = 1; p->opt[JODO_OPT_AB_PRE] = 1; p->opt[JODO_OPT_Z_SPLIT] = 1;
    p->opt[JODO_OPT_PIN_SYMMETRIC] = 0; p->opt[JODO_OPT_PIN_UNIFORM_T] = 0; p->opt[JODO_OPT_ROT_STATS] = 1; p->opt[JODO_OPT_NODE_MIX] = 1;
    p->opt[JODO_OPT_SPLIT_BF16] = 0; p->split_w = nullptr; p->split_bytes = 0;
    p->prof_enabled = 0; p->force_directed = 0; p->dbg_timing = nullptr;

    // molecules by descending size (stable): neighbouring lanes share n, big work first
    std::vector<int> order(B);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_nodes[a] > n_nodes[b]; });

    int64_t Nn = 0, rows = 0, dir = 0;
    for (int b = 0; b < B; ++b) { Nn += n_nodes[b]; rows += (int64_t)n_nodes[b] * n_nodes[b]; dir += (int64_t)n_nodes[b] * (n_nodes[b] - 1); }
    if (rows >= (int64_t)1 << 31) { delete p; return jodo_set_error(JODO_ERR_UNSUPPORTED, "batch too large: %lld dense edge rows", (long long)rows); }
    p->Nn = (int)Nn; p->Nn_pad = (int)align_up((size_t)Nn, 32); p->n_strips = p->Nn_pad / 32;
    p->rows = rows; p->dir_edges = dir;

    std::vector<int32_t> node_b(p->Nn_pad, 0), node_i(p->Nn_pad, 0), node_n(p->Nn_pad, 0), node_noff(p->Nn_pad, 0),
        node_eoff(p->Nn_pad, 0), orig_n(B), orig_noff(B), orig_eoff(B);
    int v = 0; int64_t eoff = 0;
    for (int m = 0; m < B; ++m) {
        int b = order[m], n = n_nodes[b];
        orig_n[b] = n; orig_noff[b] = v; orig_eoff[b] = (int32_t)eoff;
        for (int i = 0; i < n; ++i) {
            node_b[v + i] = b; node_i[v + i] = i; node_n[v + i] = n; node_noff[v + i] = v; node_eoff[v + i] = (int32_t)eoff;
        }
        v += n; eoff += (int64_t)n * n;
    }
    // edge work items: (strip, [t0,t1)) with balanced chunks
    std::vector<int32_t> it_strip, it_t0, it_t1, it_part, strip_parts(p->n_strips, 0);
    int max_parts = 1;
    for (int s = 0; s < p->n_strips; ++s) {
        int nmax = 0;
        for (int j = 0; j < 32; ++j) nmax = std::max(nmax, (int)node_n[s * 32 + j]);
        if (nmax == 0) continue;
        int parts = (nmax + max_chunk - 1) / max_chunk;
        int chunk = (nmax + parts - 1) / parts;
        parts = (nmax + chunk - 1) / chunk;
        max_parts = std::max(max_parts, parts);
        strip_parts[s] = parts;
        for (int q = 0; q < parts; ++q) {
            it_strip.push_back(s); it_t0.push_back(q * chunk); it_t1.push_back(std::min(nmax, (q + 1) * chunk)); it_part.push_back(q);
        }
    }
    p->n_items = (int)it_strip.size(); p->max_parts = max_parts;
    {   // directed items are consumed 4 per workgroup by the message kernel (8 by the directed scores kernel)
        const std::vector<int> ord = xcd_order(it_strip, 4);
        permute(it_strip, ord); permute(it_t0, ord); permute(it_t1, ord); permute(it_part, ord);
    }
    // pair work items for the symmetric path: lane i meets partner (i + d) mod n for d = 1 .. floor(n/2)
    // (circulant enumeration: every unordered pair exactly once, the same number of iterations per lane)
    std::vector<int32_t> pi_strip, pi_t0, pi_t1;
    for (int s = 0; s < p->n_strips; ++s) {
        int nmax = 0;
        for (int j = 0; j < 32; ++j) nmax = std::max(nmax, (int)node_n[s * 32 + j]);
        const int dmax = nmax / 2;
        if (dmax == 0) continue;
        int parts = (dmax + pair_chunk - 1) / pair_chunk;
        int chunk = (dmax + parts - 1) / parts;
        parts = (dmax + chunk - 1) / chunk;
        for (int q = 0; q < parts; ++q) { pi_strip.push_back(s); pi_t0.push_back(q * chunk); pi_t1.push_back(std::min(dmax, (q + 1) * chunk)); }
    }
    p->n_pitems = (int)pi_strip.size();
    p->pitems_single = 1;
    for (size_t i = 0; i < pi_strip.size(); ++i) if (pi_t1[i] - pi_t0[i] != 1) p->pitems_single = 0;
    {   // one pair-update item per workgroup
        const std::vector<int> ord = xcd_order(pi_strip, 1);
        permute(pi_strip, ord); permute(pi_t0, ord); permute(pi_t1, ord);
    }
    // ---- fused attention kernel: groups of whole molecules (<= 128 lanes), items = (group, range of pair offsets) ----
    // A pair (i, j) is evaluated once, by lane i; what it contributes to target j is handed to j's lane through LDS,
    // so both atoms of a pair must sit in the same 4-wave workgroup: a group holds whole molecules only.  Molecules
    // come in descending size; a group takes them in that order while they fit and then fills its gap with the
    // largest remaining molecules that still fit (lanes of a smaller molecule idle during the larger offsets, which
    // is better than idling throughout).  A molecule larger than a group (n > 128, GEOM's tail) gets groups of
    // consecutive atoms of its own and is walked in directed form (every lane visits all its sources, no hand-over).
    std::vector<int32_t> ag_node, ai_group, ai_t0, ai_t1, ai_part, ai_dir, ad_group, ad_t0, ad_t1, ad_part, ad_big, anode_parts(p->Nn_pad, 0), aw_off;
    int amax_parts = 1;
    {
        constexpr int G = 128;
        // spair_chunk: 0 = automatic: persistent workgroups with a wrap-around schedule (below); 1..200 = that many offsets per
        // item, one workgroup per item; 255 = the automatic choice of a per-item chunk (the round-3 rule before the schedule)
        const bool persist = spair_auto;
        int achunk = (spair_auto || spair_chunk == 255) ? 0 : spair_chunk;
        std::vector<int> mol_n(B), mol_noff(B);
        for (int m = 0; m < B; ++m) { mol_n[m] = n_nodes[order[m]]; mol_noff[m] = orig_noff[order[m]]; }
        std::vector<char> used(B, 0);
        std::vector<int> g_nmax, g_big;
        int head = 0;
        while (head < B) {
            if (used[head]) { ++head; continue; }
            if (mol_n[head] > G) {                              // spans several groups
                for (int o = 0; o < mol_n[head]; o += G) {
                    for (int k = 0; k < G; ++k) ag_node.push_back(o + k < mol_n[head] ? mol_noff[head] + o + k : -1);
                    g_nmax.push_back(mol_n[head]); g_big.push_back(1);
                }
                used[head++] = 1;
                continue;
            }
            int fill = 0, nmax = mol_n[head];
            std::vector<int32_t> lanes;
            for (int m = head; m < B && fill < G; ++m) {        // sizes descend: the first unused one that fits is the largest
                if (used[m] || mol_n[m] > G - fill) continue;
                for (int k = 0; k < mol_n[m]; ++k) lanes.push_back(mol_noff[m] + k);
                fill += mol_n[m]; used[m] = 1;
            }
            lanes.resize(G, -1);
            ag_node.insert(ag_node.end(), lanes.begin(), lanes.end());
            g_nmax.push_back(nmax); g_big.push_back(0);
        }
        const int ng = (int)g_nmax.size();
        p->n_agroups = ng;
        if (persist) achunk = 6;                                // (unused by the schedule below; keeps the fixed-chunk arithmetic defined)
        if (achunk <= 0 && p->dims.wide) {                      // streamed-weight kernels (nf 128 / 384): no staging cost per item;
            int64_t iters = 0;                                  // the model below, fitted to the LDS-resident kernel, picked coarser
            for (int g = 0; g < ng; ++g) iters += std::max(1, g_nmax[g] / 2);    // items and lost 27 % at nf 384 (39.5 -> 50.3 ms/step)
            achunk = 6;
            while (achunk > 1 && iters / achunk < 2 * 256) --achunk;
        }
        if (achunk <= 0) {
            // automatic: coarse items amortise the per-item cost (96 KiB of weights staged, 1 KiB of partial per atom), fine items
            // fill the chip evenly — and the kernel's time is the makespan of ~10^3 items on 256 one-workgroup CUs, which moves
            // by +-10 % with the decomposition (measured at QM9 B = 2500, ms/step of the kernel for 4 / 5 / 6 / 7 / 8 / 9 / 10 / 12
            // offsets per item: 4.12 / 4.32 / 3.91 / 4.06 / 4.08 / 3.79 / 4.23 / 4.47).  A two-parameter cost model — item time =
            // 0.8 + (number of offsets), in units of one offset, fitted to those eight points — reproduces the ranking, so the plan
            // simulates the launch (items longest first onto the least loaded CU, exactly what the dispatcher does) for every
            // chunk size and keeps the best; ties go to the coarser decomposition (fewer partials to merge).
            std::vector<double> slot(256);
            double best = 0.0;
            for (int c = 1; c <= 16; ++c) {
                std::vector<int> len;
                for (int g = 0; g < ng; ++g) {
                    if (g_big[g]) { len.push_back(g_nmax[g]); continue; }        // directed items: not modelled
                    const int dmax = g_nmax[g] / 2, parts = std::max(1, (dmax + c - 1) / c), cp = dmax > 0 ? (dmax + parts - 1) / parts : 0;
                    for (int q = 0; q < parts; ++q) len.push_back(std::min(dmax, (q + 1) * cp) - std::min(dmax, q * cp));
                }
                std::stable_sort(len.begin(), len.end(), [](int a, int b) { return a > b; });
                std::fill(slot.begin(), slot.end(), 0.0);
                for (int l : len) { auto it = std::min_element(slot.begin(), slot.end()); *it += 0.8 + (double)l; }
                const double mk = *std::max_element(slot.begin(), slot.end());
                if (achunk <= 0 || mk <= best) { best = mk; achunk = c; }
            }
        }
        struct It { int g, t0, t1, part, big; };
        std::vector<It> pit, dit;
        std::vector<int> g_parts(ng, 1), pit_dir;
        if (persist) {
            // Persistent workgroups (one per CU, k_edge_attn loops over its slot's items; the LDS-resident weights are staged once per
            // workgroup instead of once per item) with a wrap-around schedule: the pair offsets of all groups, laid end to end
            // with the per-item cost in front of every piece, are cut into JODO_ATT_SLOTS runs of equal cost.  A group that straddles
            // a cut becomes two items (two partials for its atoms) — at most one cut per slot, so the launch is balanced to within
            // one offset, where the dispatcher's longest-first greedy over whole items left 9 % (simulated and measured).
            // The groups of a molecule larger than a group (n > 128) take part with their sources instead of pair offsets: the
            // kernel walks those items in directed mode (round 5; they used to be a second, nearly empty launch serialised behind
            // this one: 710 us per block at nf 384 for two such molecules in 1 250).
            const double a_item = 0.3;                             // cost of starting an item, in pair offsets (0.3 / 0.6 / 0.9 / 1.3 measured: 3.58 / 3.62 / 3.62 / 3.76 ms/step)
            const double w_dir = 0.9;                              // a directed iteration in pair offsets: the same matrix work, one direction's vector work, no hand-over
            auto g_len = [&](int g) { return g_big[g] ? g_nmax[g] : g_nmax[g] / 2; };
            auto g_wgt = [&](int g) { return g_big[g] ? w_dir : 1.0; };
            double total = 0.0;
            for (int g = 0; g < ng; ++g) total += a_item + g_wgt(g) * (double)g_len(g);
            // every cut adds an item (and its cost) that `total` did not count: the slot capacity T grows until the walk ends
            // inside the last slot's capacity
            double T = std::max(total / JODO_ATT_SLOTS, a_item + 1.0);
            aw_off.assign(JODO_ATT_SLOTS + 1, 0);
            for (int attempt = 0; attempt < 400; ++attempt, T *= 1.004) {
                pit.clear();
                int slot = 0;
                double load = 0.0;
                auto next_slot = [&]() { if (slot + 1 < JODO_ATT_SLOTS) { ++slot; load = 0.0; } };
                for (int g = 0; g < ng; ++g) {
                    const int len = g_len(g);
                    const double wg = g_wgt(g);
                    int t = 0, part = 0;
                    do {
                        double room = T - load - a_item;
                        if (room < 0.75 && load > 0.0 && slot + 1 < JODO_ATT_SLOTS) { next_slot(); room = T - a_item; }
                        int take = std::min(len - t, std::max(1, (int)(room / wg + 0.5)));
                        take = std::max(take, 0);
                        pit.push_back({g, t, t + take, part++, slot});      // `big` holds the slot until the lists are written
                        load += a_item + wg * (double)take;
                        t += take;
                        if (load >= T - 0.5) next_slot();
                    } while (t < len);
                    g_parts[g] = part;
                }
                if (load <= T + 0.5) break;                                  // the last slot did not overflow
            }
            // items are already in slot order (slots ascend along the walk)
            for (const It& it : pit) aw_off[it.big + 1]++;
            for (int sl = 0; sl < JODO_ATT_SLOTS; ++sl) aw_off[sl + 1] += aw_off[sl];
            for (It& it : pit) { it.big = 0; pit_dir.push_back(g_big[it.g]); }
        }
        for (int g = 0; g < ng; ++g) {
            const int nmax = g_nmax[g], dmax = nmax / 2;
            int parts;
            if (persist) parts = g_parts[g];
            else if (g_big[g]) parts = std::max(1, (nmax + 2 * achunk - 1) / (2 * achunk));
            else parts = std::max(1, (dmax + achunk - 1) / achunk);
            amax_parts = std::max(amax_parts, parts);
            const int cp = dmax > 0 ? (dmax + parts - 1) / parts : 0, cd = (nmax + parts - 1) / parts;
            for (int q = 0; q < parts; ++q) {
                if (!g_big[g] && !persist) pit.push_back({g, std::min(dmax, q * cp), std::min(dmax, (q + 1) * cp), q, 0});
                // (the directed launch serves a big group for symmetric inputs only when the pair launch does not carry it)
                dit.push_back({g, std::min(nmax, q * cd), std::min(nmax, (q + 1) * cd), q, (g_big[g] && !persist) ? 1 : 0});
            }
            for (int k = 0; k < G; ++k) { const int v = ag_node[(size_t)g * G + k]; if (v >= 0) anode_parts[v] = parts; }
        }
        // longest items first (the tail of the launch is then filled with short ones); stable, so neighbours in the
        // queue stay neighbours in memory
        auto lpt = [](const It& a, const It& b) { return (a.t1 - a.t0) > (b.t1 - b.t0); };
        if (!persist) std::stable_sort(pit.begin(), pit.end(), lpt);
        std::stable_sort(dit.begin(), dit.end(), lpt);
        for (const It& it : pit) { ai_group.push_back(it.g); ai_t0.push_back(it.t0); ai_t1.push_back(it.t1); ai_part.push_back(it.part); }
        ai_dir.assign(pit.size(), 0);
        for (size_t i = 0; i < pit_dir.size(); ++i) ai_dir[i] = pit_dir[i];
        p->n_ai_dir = 0;
        for (size_t i = 0; i < ai_dir.size(); ++i) p->n_ai_dir += ai_dir[i] ? 1 : 0;
        for (const It& it : dit) { ad_group.push_back(it.g); ad_t0.push_back(it.t0); ad_t1.push_back(it.t1); ad_part.push_back(it.part); ad_big.push_back(it.big); }
        p->n_aitems = (int)pit.size(); p->n_aditems = (int)dit.size(); p->amax_parts = amax_parts;
        p->a_persist = persist ? 1 : 0;
        if (!persist) aw_off.assign(1, 0);
        p->has_big = 0;
        for (int g = 0; g < ng; ++g) p->has_big |= (g_big[g] && !persist) ? 1 : 0;
    }

    auto put = [&](const std::vector<int32_t>& a, size_t* off) {
        *off = p->desc.size();
        p->desc.insert(p->desc.end(), a.begin(), a.end());
        while (p->desc.size() % 64) p->desc.push_back(0);
    };
    put(node_b, &p->off_node_b); put(node_i, &p->off_node_i); put(node_n, &p->off_node_n);
    put(node_noff, &p->off_node_noff); put(node_eoff, &p->off_node_eoff);
    put(orig_n, &p->off_orig_n); put(orig_noff, &p->off_orig_noff); put(orig_eoff, &p->off_orig_eoff);
    put(it_strip, &p->off_item_strip); put(it_t0, &p->off_item_t0); put(it_t1, &p->off_item_t1); put(it_part, &p->off_item_part); put(strip_parts, &p->off_strip_parts);
    put(pi_strip, &p->off_pitem_strip); put(pi_t0, &p->off_pitem_t0); put(pi_t1, &p->off_pitem_t1);
    put(ag_node, &p->off_ag_node); put(ai_group, &p->off_ai_group); put(ai_t0, &p->off_ai_t0); put(ai_t1, &p->off_ai_t1); put(ai_part, &p->off_ai_part); put(ai_dir, &p->off_ai_dir);
    put(ad_group, &p->off_ad_group); put(ad_t0, &p->off_ad_t0); put(ad_t1, &p->off_ad_t1); put(ad_part, &p->off_ad_part); put(ad_big, &p->off_ad_big);
    put(anode_parts, &p->off_anode_parts); put(aw_off, &p->off_aw_off);
    {   // one row per unordered pair of every molecule's dense edge tile + its mirror: the edge head evaluates a symmetric pair once.
        // The row is the one the pair kernels' evaluating lane owns — (i, i + d mod n) of the circulant walk, pair_of() in
        // dgt_kernels_sym.h — so that under JODO_OPT_HALF_ROWS the head reads exactly the rows that were written
        std::vector<int32_t> ut;
        for (int b = 0; b < B; ++b) {
            const int n = orig_n[b];
            const int64_t e0 = orig_eoff[b];
            for (int i = 0; i < n; ++i)
                for (int d = 1; 2 * d <= n; ++d) {
                    if (2 * d == n && 2 * i >= n) continue;
                    const int j = (i + d) % n;
                    ut.push_back((int32_t)(e0 + (int64_t)i * n + j)); ut.push_back((int32_t)(e0 + (int64_t)j * n + i));
                }
        }
        while ((ut.size() / 2) % 32) { ut.push_back(-1); ut.push_back(-1); }
        p->n_ut_pad = (int)(ut.size() / 2);
        put(ut, &p->off_ut_rows);
    }
    {   // Gram tiles of the rotated statistics: every ordered pair of strips that share a molecule (sizes descend and molecules
        // are contiguous, so the strips of the molecules touching strip s form one range)
        std::vector<int32_t> gsa, gsc;
        for (int s = 0; s < p->n_strips; ++s) {
            int lo = s, hi = s;
            for (int j = 0; j < 32; ++j) {
                const int v0 = s * 32 + j;
                if (node_n[v0] <= 0) continue;
                lo = std::min(lo, (int)node_noff[v0] / 32);
                hi = std::max(hi, (int)(node_noff[v0] + node_n[v0] - 1) / 32);
            }
            for (int c = lo; c <= hi; ++c) { gsa.push_back(s); gsc.push_back(c); }
        }
        {   // by the larger of the two strips: the tiles among the first k strips are a prefix (k_node_mix)
            std::vector<int> ord(gsa.size());
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return std::max(gsa[a], gsc[a]) < std::max(gsa[b], gsc[b]); });
            permute(gsa, ord); permute(gsc, ord);
        }
        p->n_gtiles = (int)gsa.size(); p->gt_cache_full = -1; p->gt_cache_count = 0;
        put(gsa, &p->off_gt_sa); put(gsc, &p->off_gt_sc);
    }

    // workspace layout
    const DgtDims& d = p->dims;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t f = sizeof(float), NP = (size_t)p->Nn_pad, R = (size_t)rows + 32, Bp = align_up((size_t)B, 32);
    WsLayout& w = p->ws;
    w.hid1 = take(Bp * d.T * f); w.temb = take(Bp * d.T * f); w.tembs = take(Bp * d.T * f); w.mods = take(Bp * (size_t)d.Mtot * f);
    w.condh = take(Bp * (size_t)std::max(1, d.cond_ch) * d.D * f); w.condh2 = take(Bp * (size_t)std::max(1, d.cond_ch) * d.D * f);
    w.pos0 = take(NP * 4 * f); w.pos1 = take(NP * 4 * f); w.dpos = take(NP * max_parts * 4 * f); w.cpos = take(NP * 4 * f);
    w.feat = take(NP * d.ndp * f); w.h = take(NP * d.D * f); w.hhat = take(NP * amax_parts * d.D * f);
    w.astat = take(NP * amax_parts * 32 * f);
    w.q = take(NP * d.QKP * f); w.k = take(NP * d.QKP * f); w.v = take(NP * d.D * f); w.n2e = take(NP * d.De * f);
    w.wrow = take(NP * d.D * f); w.wcol = take(NP * d.D * f); w.ua = take(NP * d.D * f); w.ub = take(NP * d.D * f); w.rmean = take(NP * 2 * f); w.mfold = take((size_t)d.L * d.D * 2 * d.De * f); w.mfold_s = take((size_t)d.L * d.D * 2 * d.De * 6); w.ffold_s = take((size_t)d.L * d.D * d.D * 6); w.ffold = take((size_t)d.L * d.D * d.D * f); w.ahid = take(NP * d.KNH * f);
    w.apred = take(NP * 32 * f);
    w.eflag = take(R * sizeof(int32_t)); w.e = take(R * d.De * f); w.e2 = take(R * d.De * f);
    w.ehid = take(R * d.KEH * f); w.epred = take(R * 4 * f); w.dposE = take(R * 4 * f); w.gramE = take(R * f);
    w.total = o;
    *out = p;
    return JODO_OK;
}

extern "C" void jodo_plan_destroy(jodo_plan* p) {
    if (!p) return;
    for (void* e : p->prof_ev) (void)hipEventDestroy((hipEvent_t)e);
    for (void* e : p->prof_pool) (void)hipEventDestroy((hipEvent_t)e);
    delete p;
}
extern "C" int jodo_profile_enable(jodo_plan* p, int enable) {
    if (!p) return jodo_set_error(JODO_ERR_ARG, "null plan");
    p->prof_enabled = enable;
    return JODO_OK;
}
extern "C" int jodo_profile_read(jodo_plan* p, float* ms_sum, int32_t* launches) {
    if (!p || !ms_sum || !launches) return jodo_set_error(JODO_ERR_ARG, "profile_read: null");
    for (int c = 0; c < JODO_PROF_COUNT; ++c) { ms_sum[c] = 0.f; launches[c] = 0; }
    for (size_t i = 0; i + 1 < p->prof_ev.size(); i += 2) {
        hipEvent_t a = (hipEvent_t)p->prof_ev[i], b = (hipEvent_t)p->prof_ev[i + 1];
        hipError_t e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "profile_read: %s", hipGetErrorString(e));
        const int c = p->prof_cls[i / 2];
        ms_sum[c] += ms; launches[c] += 1;
        p->prof_pool.push_back(p->prof_ev[i]); p->prof_pool.push_back(p->prof_ev[i + 1]);
    }
    p->prof_ev.clear(); p->prof_cls.clear();
    return JODO_OK;
}
extern "C" size_t jodo_plan_desc_bytes(const jodo_plan* p) { return p ? p->desc.size() * sizeof(int32_t) : 0; }
extern "C" size_t jodo_plan_workspace_bytes(const jodo_plan* p) { return p ? p->ws.total : 0; }
extern "C" int64_t jodo_plan_mod_len(const jodo_plan* p) { return p ? p->dims.Mtot : 0; }
extern "C" int jodo_plan_stats(const jodo_plan* p, int64_t* o) {
    if (!p || !o) return jodo_set_error(JODO_ERR_ARG, "plan_stats: null");
    o[0] = p->Nn; o[1] = p->rows; o[2] = p->dir_edges; o[3] = p->n_strips; o[4] = p->n_items; o[5] = p->max_parts;
    return JODO_OK;
}

// ---- executed-work model (bench.py's roofline leg) ---------------------------------------------------------
// fp32 flops the kernels of ONE forward issue on the matrix pipe (v_mfma_f32_32x32x2_f32 = 4096 flop each), per launch class,
// counted from the plan's own work-item lists and the loop structure of the kernels: out-blocks of 32 x K / 2 k-steps per
// projection and 32-item wave iteration, padded lanes included — what SQ_INSTS_MFMA x 4096 measures (tools/pmc_work.py checks
// the two against each other).  uniform_t / symmetric: the device flags of the call (shared modulation row; pair path).
namespace {
inline double proj(int out, int K) { return (double)((out + 31) / 32) * (double)((K + 7) / 8 * 4); }   // MFMAs per wave iteration
}
extern "C" int jodo_plan_work(const jodo_plan* p, int uniform_t, int symmetric, double* mfma_flops) {
    if (!p || !mfma_flops) return jodo_set_error(JODO_ERR_ARG, "plan_work: null");
    const DgtDims& d = p->dims;
    const bool tuned = !d.wide, sym = symmetric && !p->force_directed && p->n_pitems > 0;
    const int D = d.D, De = d.De, L = d.L, r = d.r;
    const int32_t* dsc = p->desc.data();
    double cls[JODO_PROF_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double strips = p->n_strips;
    // prologue: time MLP second layer + fused modulation projection (row groups of 32 molecules; one group when the row is
    // shared), conditional path, embeddings (k_fold_coord is double-precision vector work)
    const double rowgroups = (uniform_t && d.cond_ch == 0) ? 1.0 : (double)((p->B + 31) / 32);
    cls[JODO_PROF_PROLOGUE] += rowgroups * (proj(d.T, d.T) + proj((int)d.Mtot, d.T));
    if (d.cond_ch > 0) cls[JODO_PROF_PROLOGUE] += (double)((p->B * d.cond_ch + 31) / 32) * proj(D, D) + (double)((p->B + 31) / 32) * proj(d.T, d.cond_ch * D);
    cls[JODO_PROF_PROLOGUE] += strips * proj(D, d.ndp);
    double dir_iters = 0, pair_iters = 0;                 // wave iterations of the directed / pair item lists
    for (int i = 0; i < p->n_items; ++i) dir_iters += dsc[p->off_item_t1 + i] - dsc[p->off_item_t0 + i];
    for (int i = 0; i < p->n_pitems; ++i) pair_iters += dsc[p->off_pitem_t1 + i] - dsc[p->off_pitem_t0 + i];
    // k_embed_edges: every dense row; under JODO_OPT_PRE_EMBED (tuned set) it runs inside the first block's node-pre launch
    cls[(p->opt[JODO_OPT_PRE_EMBED] != 0 && L > 0) ? JODO_PROF_NODE_PRE : JODO_PROF_PROLOGUE] += dir_iters * proj(De, d.einp + De);
    // per block
    const int nqb = tuned ? 8 : d.SH;                                                    // 32-row blocks of q / k / lin_edge0
    const double qkv = 2.0 * nqb * (D / 2) + proj(D, D);
    const bool fuse_pre = tuned && p->opt[JODO_OPT_FUSE_NEXT_QKV] != 0 && L > 1 && p->n_strips >= 1024;
    // (k_node_ab_pre: below 1024 strips the following block's q / k / v items run in the node-post bracket as well)
    const bool ab_pre = !fuse_pre && p->opt[JODO_OPT_AB_PRE] != 0 && L > 1 && (!tuned || (p->n_strips < 1024 && p->opt[JODO_OPT_NODE_POST_WAVES] == 0));
    cls[JODO_PROF_NODE_PRE] += strips * qkv * ((fuse_pre || ab_pre) ? 1 : L);
    if (fuse_pre || ab_pre) cls[JODO_PROF_NODE_POST] += strips * qkv * (L - 1);
    cls[JODO_PROF_NODE_POST] += L * strips * (proj(De, D) + proj(r * D, D) + proj(D, r * D) + 2 * proj(D, D) + proj(d.cnp, D));
    const bool hoist = sym && (D == 256 || uniform_t);                                  // coord_mlp.0 pushed through the LayerNorm
    if (p->n_pitems > 0 && (D == 256 || uniform_t)) cls[JODO_PROF_NODE_POST] += L * strips * 2 * proj(D, D);      // k_node_ab
    // fused attention: pair items (4 waves x offsets) for groups of whole molecules, directed items otherwise
    const double att_iter = proj(De, 2 * De) + nqb * (double)(De / 2) + proj(D, De);
    double att_iters = 0;
    for (int i = 0; i < p->n_aitems; ++i) if (sym) att_iters += 4.0 * (dsc[p->off_ai_t1 + i] - dsc[p->off_ai_t0 + i]);
    for (int i = 0; i < p->n_aditems; ++i)
        if (!sym || dsc[p->off_ad_big + i]) att_iters += 4.0 * (dsc[p->off_ad_t1 + i] - dsc[p->off_ad_t0 + i]);
    cls[JODO_PROF_EDGE_ATTN] += L * att_iters * att_iter;
    // edge update: trunk (edge FFN, readout, e + distance part S of input_lin) + coord_mlp.0 in its form of the call
    const double trunk = proj(r * De, De) + proj(De, r * De) + proj(d.cep, De) + proj(D, 2 * De);
    if (sym) {
        const bool fold = uniform_t && d.cond_ch == 0;
        const double c0 = fold ? proj(D, 2 * De) : (hoist ? proj(D, D) : 2 * proj(D, D));
        double tr = trunk;
        if (fold && p->opt[JODO_OPT_ROT_STATS]) {            // rotated statistics: triangular L [e ; G] instead of S, + the Gram tiles
            tr -= proj(D, 2 * De);
            for (int b = 0; b < 2 * De / 32; ++b) tr += (double)(2 * De - 32 * b) / 2;
            cls[JODO_PROF_NODE_POST] += L * (double)p->n_gtiles * (double)(D - 2 * De) / 2;
        }
        cls[JODO_PROF_EDGE_UPDATE] += L * pair_iters * (tr + c0);
        // JODO_OPT_Z_SPLIT: the items of the last round run with zw waves each; every wave repeats the trunk and the statistics,
        // the Z blocks (c0) are dealt out (launch_update_sym, dgt_edge.hip)
        if (p->opt[JODO_OPT_Z_SPLIT] != 0 && hoist && p->opt[JODO_OPT_PIN_UNIFORM_T] != (fold ? 2 : 1)) {
            const int rem = p->n_pitems % 1024;
            const int zw = rem > 0 ? (rem <= 256 ? 4 : ((p->opt[JODO_OPT_Z_SPLIT] == 2 && rem <= 512) ? 2 : 1)) : 1;
            cls[JODO_PROF_EDGE_UPDATE] += L * (double)rem * (zw - 1) * tr;
        }
    } else {
        cls[JODO_PROF_EDGE_UPDATE] += L * dir_iters * (trunk + proj(D, D));
    }
    // heads
    cls[JODO_PROF_EPILOGUE] += strips * (proj(D, d.KNH) + proj(D / 2, D) + proj(d.nd, D / 2));
    const double eh = 2 * proj(De, d.KEH) + proj(De, 2 * De) + proj(32, De);
    cls[JODO_PROF_EPILOGUE] += (sym ? (double)(p->n_ut_pad / 32) : (double)((p->rows + 31) / 32)) * eh;
    for (int c = 0; c < JODO_PROF_COUNT; ++c) mfma_flops[c] = cls[c] * 4096.0;
    return JODO_OK;
}

extern "C" int jodo_plan_upload(jodo_plan* p, void* desc_dev, void* stream) {
    if (!p || !desc_dev) return jodo_set_error(JODO_ERR_ARG, "plan_upload: null");
    hipError_t e = hipMemcpyAsync(desc_dev, p->desc.data(), p->desc.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "plan_upload: %s", hipGetErrorString(e));
    return JODO_OK;
}
extern "C" int jodo_debug_set_timing_buffer(jodo_plan* p, void* dev16xu64) {
    if (!p) return jodo_set_error(JODO_ERR_ARG, "null plan");
    p->dbg_timing = dev16xu64;
    return JODO_OK;
}
extern "C" int jodo_debug_set_force_directed(jodo_plan* p, int on) {
    if (!p) return jodo_set_error(JODO_ERR_ARG, "null plan");
    if (on && p->opt[JODO_OPT_PIN_SYMMETRIC] == 1)
        return jodo_set_error(JODO_ERR_ARG, "set_force_directed: the plan is pinned to the symmetric path");
    p->force_directed = on;
    return JODO_OK;
}
extern "C" int jodo_plan_set_option(jodo_plan* p, int option, int value) {
    if (!p) return jodo_set_error(JODO_ERR_ARG, "null plan");
    if (option < 0 || option >= JODO_OPT_COUNT) return jodo_set_error(JODO_ERR_ARG, "plan_set_option: unknown option %d", option);
    if (option == JODO_OPT_NODE_POST_WAVES && value != 0 && value != 1 && value != 2 && value != 4 && value != 12 && value != 14)
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: node-post waves per strip must be 0 (auto), 1, 2 or 4");
    if (option == JODO_OPT_SPLIT_BF16 && (value < 0 || value > 2))
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: the split-bf16 form is 0 (off), 1 (pair update + node kernel) or 2 (also the attention kernel: experiments, measured slower), got %d", value);
    if ((option == JODO_OPT_FUSE_NEXT_QKV || option == JODO_OPT_DIR_SPLIT || option == JODO_OPT_NODE_MIX || option == JODO_OPT_HEADS_MIX || option == JODO_OPT_HALF_ROWS || option == JODO_OPT_PRE_EMBED || option == JODO_OPT_AB_PRE) && value != 0 && value != 1)
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: option %d is a switch (0 or 1), got %d", option, value);
    if (option == JODO_OPT_Z_SPLIT && (value < 0 || value > 2))
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: the Z split is 0 (off), 1 (four waves for a last round of <= 256 items) or 2 (also two waves up to 512), got %d", value);
    if (option == JODO_OPT_ROT_STATS && (value < 0 || value > 2))
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: rotated statistics are 0 (off), 1 (on) or 2 (on, uncentred Gram tiles: tests), got %d", value);
    if (option == JODO_OPT_PIN_SYMMETRIC && value == 1 && p->force_directed)
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: the symmetric pin contradicts jodo_debug_set_force_directed");
    if ((option == JODO_OPT_PIN_SYMMETRIC || option == JODO_OPT_PIN_UNIFORM_T) && (value < 0 || value > 2))
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: a pin is 0 (none), 1 or 2, got %d", value);
    if (option == JODO_OPT_ATTN_VARIANT && (value < 0 || value > 3))
        return jodo_set_error(JODO_ERR_ARG, "plan_set_option: attention variant must be 0..3, got %d", value);
    p->opt[option] = value;
    return JODO_OK;
}
// out8: pair-mode attention items, their total offsets, partials per atom (max), persistent schedule (0 / 1), and for the
// persistent schedule the smallest / largest slot load in offsets, the largest item count of a slot and the number of idle slots
extern "C" int jodo_debug_attn_schedule(const jodo_plan* p, int64_t* o) {
    if (!p || !o) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: null");
    const int32_t* d = p->desc.data();
    int64_t iters = 0;
    for (int i = 0; i < p->n_aitems; ++i) iters += d[p->off_ai_t1 + i] - d[p->off_ai_t0 + i];
    o[0] = p->n_aitems; o[1] = iters; o[2] = p->amax_parts; o[3] = p->a_persist; o[4] = o[5] = o[6] = o[7] = 0;
    {   // self-check: the items of every group of whole molecules tile its pair offsets [0, nmax / 2) exactly once, and their
        // partial indices are 0 .. parts - 1 with parts = what the atoms of the group expect (anode_parts)
        std::vector<std::vector<std::pair<int, int>>> per((size_t)p->n_agroups);
        std::vector<std::vector<int>> parts((size_t)p->n_agroups);
        for (int i = 0; i < p->n_aitems; ++i) {
            const int g = d[p->off_ai_group + i];
            if (g < 0 || g >= p->n_agroups) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: item %d names group %d", i, g);
            per[g].push_back({d[p->off_ai_t0 + i], d[p->off_ai_t1 + i]});
            parts[g].push_back(d[p->off_ai_part + i]);
        }
        for (int g = 0; g < p->n_agroups; ++g) {
            int nmax = 0, want_parts = 0;
            for (int k = 0; k < 128; ++k) {
                const int v = d[p->off_ag_node + (size_t)g * 128 + k];
                if (v >= 0) { nmax = std::max(nmax, (int)d[p->off_node_n + v]); want_parts = d[p->off_anode_parts + v]; }
            }
            // a group of a molecule larger than a group: carried by the pair launch (persistent schedule) as directed-mode items that
            // tile its SOURCES [0, n), or not at all (fixed-chunk plans: the directed launch serves it)
            const bool bigg = nmax > 128;
            if (bigg && !p->a_persist) { if (!per[g].empty()) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: pair items for the big group %d", g); continue; }
            for (int i = 0; i < p->n_aitems; ++i)
                if (d[p->off_ai_group + i] == g && d[p->off_ai_dir + i] != (bigg ? 1 : 0)) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: item %d of group %d has the wrong mode", i, g);
            std::sort(per[g].begin(), per[g].end());
            std::sort(parts[g].begin(), parts[g].end());
            int at = 0;
            for (const auto& it : per[g]) {
                if (it.first != at || it.second < it.first) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: group %d offsets not tiled at %d", g, at);
                at = it.second;
            }
            const int want_len = bigg ? nmax : nmax / 2;
            if (at != want_len || per[g].empty()) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: group %d covers %d of %d offsets", g, at, want_len);
            for (size_t q = 0; q < parts[g].size(); ++q)
                if (parts[g][q] != (int)q) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: group %d partial indices are not 0..%d", g, (int)parts[g].size() - 1);
            if ((int)parts[g].size() != want_parts) return jodo_set_error(JODO_ERR_ARG, "attn_schedule: group %d has %d items, its atoms expect %d partials", g, (int)parts[g].size(), want_parts);
        }
    }
    if (p->a_persist) {
        int64_t lo = INT64_MAX, hi = 0, mi = 0, idle = 0;
        for (int s = 0; s < JODO_ATT_SLOTS; ++s) {
            const int k0 = d[p->off_aw_off + s], k1 = d[p->off_aw_off + s + 1];
            int64_t l = 0;
            for (int k = k0; k < k1; ++k) l += d[p->off_ai_t1 + k] - d[p->off_ai_t0 + k];
            lo = std::min(lo, l); hi = std::max(hi, l); mi = std::max<int64_t>(mi, k1 - k0); idle += (k1 == k0);
        }
        o[4] = lo; o[5] = hi; o[6] = mi; o[7] = idle;
    }
    return JODO_OK;
}
extern "C" int jodo_plan_set_split_weights(jodo_plan* p, const void* tape_dev, size_t bytes) {
    if (!p) return jodo_set_error(JODO_ERR_ARG, "null plan");
    if (tape_dev) {
        size_t total = 0, per_block = 0, node_block = 0, attn_block = 0;
        const int rc = jodo_dgt_split_size(&p->cfg, &total, &per_block, &node_block, &attn_block);
        if (rc != JODO_OK) return rc;
        if (bytes != total) return jodo_set_error(JODO_ERR_ARG, "set_split_weights: %zu bytes, this configuration's tape has %zu", bytes, total);
    }
    p->split_w = tape_dev; p->split_bytes = tape_dev ? bytes : 0;
    return JODO_OK;
}
extern "C" int jodo_debug_set_max_blocks(jodo_plan* p, int mb) {
    if (!p) return jodo_set_error(JODO_ERR_ARG, "null plan");
    p->max_blocks = mb;
    return JODO_OK;
}
