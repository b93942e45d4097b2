// OPT-IN split-bf16 form of the folded pair update (JODO_OPT_SPLIT_BF16; never the default, never the headline).
//
// Same work item, same formulas, same stores as wide::k_edge_update_sym<256, R, FOLD = true, ROT = true> (dgt_kernels_wide.h; the
// reference: MultiCondEquiUpdate models/mol_gnn.py:71-94, edge FFN + readout :313-317, :566-568) — what changes is how the dense
// projections are evaluated: every weight and every activation is the exact sum of three bf16 terms, a K = 16 step is six
// v_mfma_f32_32x32x16_bf16 products accumulated in fp32 (dgt_split.h: fp32-equivalent, dropped terms <= 3 * 2^-26; gate measurements in
// profiles/r06_split_gate.txt).  Per pair offset: 720 bf16 MFMAs x 32 cycles = 23 k matrix cycles against 960 x 64 = 61 k of the
// exact-fp32 form, and vector work issued beside a bf16 MFMA costs about half of what it costs beside an fp32 MFMA (same table).
//
// What that changes in the kernel's shape: the weights are 1.5 x the bytes for 0.375 x the matrix time, i.e. 4 x the weight bandwidth
// per wave — 64 B / clk / CU with four one-wave workgroups streaming the same weights from L2 through the CU's vector L1, which is that
// path's peak (the gate's streamed K = 128 -> 256 projection reached 1.5 - 1.67 x instead of 2.67 x that way).  So here a WORKGROUP of
// four waves (one pair item each, as before) shares ONE stream: the block's weights are a contiguous tape of K16 steps in consumption
// order (dgt_pack.cpp pack_split_tape; the folded coord_mlp.0 image from k_fold_coord follows it), cut into chunks of four steps
// (12 KiB); every wave fetches a quarter of a chunk (three 16-byte loads per lane), the chunk is assembled in a three-slot LDS ring and
// all four waves read their A operands from there (conflict-free ds_read_b128).  One workgroup barrier per chunk = per 24 MFMAs of
// every wave; the four waves execute the same instruction stream on items of equal length (one circulant offset each), so the
// lock-step costs little.  L1 traffic drops to a quarter, LDS reads run at half of the LDS peak.
//
// Ring protocol (chunk g is consumed during "period g"):   boundary(g):  [a] stage registers (chunk g + 1, requested one period ago)
// -> LDS slot (g + 1) % 3;  [b] __syncthreads();  [c] request chunk g + 2 into the stage registers.  Reads of chunk g follow the
// barrier of boundary(g - 1) that came after its write; slot (g + 1) % 3 last held chunk g - 2, whose reads every wave finished
// before it arrived at the barrier of boundary(g - 1).  The loads are requested AFTER the barrier so that no fence waits for them.
#pragma once
#include "dgt_kernels_wide.h"
#include "dgt_split.h"

namespace jd {
namespace split {

constexpr int CH_STEPS = 4;                          // K16 steps per tape chunk
constexpr int CH_BYTES = CH_STEPS * 3072;            // 12 KiB
constexpr int RING_SLOTS = 3;
constexpr int SPLIT_WAVES = 4;

struct Tape {
    __amdgpu_buffer_rsrc_t rs0, rs1;                 // static tape of this block | folded coord_mlp.0 image of this block
    int n0, ntot;                                    // chunks of the static part, chunks in all
    unsigned ld_off;                                 // this thread's byte offset inside a chunk: wave * 3072 + lane * 16 (+ i * 1024)
    unsigned rd_off;                                 // lane * 16
    char* ring;                                      // LDS ring base
    u32x4 stage[3];
};

__device__ __forceinline__ void tape_request(Tape& T, int g) {          // global -> stage registers
    if (g >= T.ntot) return;
    const bool st = g < T.n0;
    const unsigned base = (unsigned)(st ? g : g - T.n0) * CH_BYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        T.stage[i] = __builtin_amdgcn_raw_buffer_load_b128(st ? T.rs0 : T.rs1, T.ld_off + (unsigned)i * 1024u, base, 0);
}
__device__ __forceinline__ void tape_commit(Tape& T, int g) {           // stage registers -> LDS slot of chunk g
    if (g >= T.ntot) return;
    char* dst = T.ring + (g % RING_SLOTS) * CH_BYTES + T.ld_off;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(dst + i * 1024) = T.stage[i];
}
__device__ __forceinline__ void tape_boundary(Tape& T, int g) {
    tape_commit(T, g + 1);
    __syncthreads();
    tape_request(T, g + 2);
    // pinned HERE: left alone, hipcc sinks the three loads down to the next boundary's ds_write and waits for them there — one
    // exposed L2 round trip per chunk (seen in the ISA of the first build); behind the fence they have a whole chunk period to land
    pipeline_fence();
}
__device__ __forceinline__ void tape_start(Tape& T) {
    tape_request(T, 0);
    tape_commit(T, 0);
    tape_request(T, 1);
}

// NS consecutive tape steps: acc += W_steps * act.  The block starts W0 steps into chunk g (W0 is known at compile time everywhere: the
// tape sections are unrolled straight-line code, and the real loops advance by whole chunks); g moves on.
template <int NS, int W0>
__device__ __forceinline__ f32x16 tape_block(Tape& T, int& g, const Split8* act, f32x16 acc) {
    static_assert(W0 >= 0 && W0 < CH_STEPS, "offset inside a chunk");
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int w = (W0 + s) % CH_STEPS, gi = g + (W0 + s) / CH_STEPS;
        if (w == 0) tape_boundary(T, gi);
        const char* src = T.ring + (gi % RING_SLOTS) * CH_BYTES + w * 3072 + T.rd_off;
        const bf16x8 wh = as_bf16x8(*reinterpret_cast<const u32x4*>(src));
        const bf16x8 wm = as_bf16x8(*reinterpret_cast<const u32x4*>(src + 1024));
        const bf16x8 wl = as_bf16x8(*reinterpret_cast<const u32x4*>(src + 2048));
        acc = mfma_step_s(wh, wm, wl, act[s], acc);
    }
    g += (W0 + NS) / CH_STEPS;
    return acc;
}

// ---- the node kernel's ring (k_node_post_split: ONE workgroup per CU, so nothing covers what a chunk boundary costs) ----
// Measured on MI355X (SQ counters of tools/gpu_split_pmc.sh): with four-step chunks the waves of k_node_post_split sit parked 44 % of
// their cycles — about 900 cycles at each of a strip's 304 chunk boundaries (commit, LDS wait, barrier behind the slowest of four waves);
// reading fragments a step ahead and a second stage set (two periods of prefetch distance) moved that by 0 / 4 %: it is the boundary
// itself.  So this ring takes CHS = 8 steps per chunk (24 KiB, three slots = 72 KiB: the node kernel has the CU's LDS to itself) and
// halves their number; every block of the node tape is a whole number of such chunks except the four-step ff_linear2 pieces, which
// come in pairs (H = the piece's half of the chunk).
constexpr int N_CHS = 8;
constexpr int N_CH_BYTES = N_CHS * 3072;
constexpr int N_SLOTS = 3;
constexpr int N_LOADS = N_CH_BYTES / SPLIT_WAVES / 1024;      // 16-byte loads per lane and chunk (6)
struct Tape2 {
    __amdgpu_buffer_rsrc_t rs;
    int ntot;
    unsigned ld_off, rd_off;
    char* ring;
    u32x4 stage[N_LOADS];
    u32x4 cur[3];                                    // the NEXT step's three fragments, read one step ahead (one wave per SIMD: nothing else covers
                                                     // the LDS round trip — four waves ask for 12 KiB at once right behind every barrier)
};
__device__ __forceinline__ void tape2_request(Tape2& T, int g) {
    if (g >= T.ntot) return;
#pragma unroll
    for (int i = 0; i < N_LOADS; ++i)
        T.stage[i] = __builtin_amdgcn_raw_buffer_load_b128(T.rs, T.ld_off + (unsigned)i * 1024u, (unsigned)g * N_CH_BYTES, 0);
}
__device__ __forceinline__ void tape2_commit(Tape2& T, int g) {
    if (g >= T.ntot) return;
    char* dst = T.ring + (g % N_SLOTS) * N_CH_BYTES + T.ld_off;
#pragma unroll
    for (int i = 0; i < N_LOADS; ++i) *reinterpret_cast<u32x4*>(dst + i * 1024) = T.stage[i];
}
__device__ __forceinline__ void tape2_boundary(Tape2& T, int g) {
    tape2_commit(T, g + 1);
    __syncthreads();
    tape2_request(T, g + 2);
    pipeline_fence();
}
__device__ __forceinline__ void tape2_start(Tape2& T, int g0) {      // g0: first chunk this launch consumes
    tape2_request(T, g0);
    tape2_commit(T, g0);
    tape2_request(T, g0 + 1);
    pipeline_fence();
    __syncthreads();                                 // every wave's quarter of the first chunk is visible: read its first step
    const char* src = T.ring + (g0 % N_SLOTS) * N_CH_BYTES + T.rd_off;
    T.cur[0] = *reinterpret_cast<const u32x4*>(src);
    T.cur[1] = *reinterpret_cast<const u32x4*>(src + 1024);
    T.cur[2] = *reinterpret_cast<const u32x4*>(src + 2048);
}
// NS steps starting H steps into chunk g (H = 0 or, for the second of a pair of four-step pieces, 4); g moves on when the chunk is done
template <int NS, int H, bool TWO = false>            // TWO: two alternating accumulators (mfma_step_s2; measured: no gain, kept selectable)
__device__ __forceinline__ f32x16 tape2_block(Tape2& T, int& g, const Split8* act, f32x16 acc) {
    static_assert((H + NS) % N_CHS == 0 || (H == 0 && NS < N_CHS), "a block ends on a chunk boundary or is the first half of a chunk");
    f32x16 acc2 = zero16();                          // second accumulator (mfma_step_s2): neighbouring MFMAs independent
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int w = (H + s) % N_CHS, gi = g + (H + s) / N_CHS;
        if (w == 0) tape2_boundary(T, gi);
        // this step's fragments were read during the previous step (the tape is consumed strictly in order; chunk g + 1 is committed
        // before the barrier of boundary(g), so its first step may be read during the last step of chunk g); now the next step's
        const bf16x8 wh = as_bf16x8(T.cur[0]), wm = as_bf16x8(T.cur[1]), wl = as_bf16x8(T.cur[2]);
        const int wn = (H + s + 1) % N_CHS, gn = g + (H + s + 1) / N_CHS;
        const char* nx = T.ring + (gn % N_SLOTS) * N_CH_BYTES + wn * 3072 + T.rd_off;
        T.cur[0] = *reinterpret_cast<const u32x4*>(nx);
        T.cur[1] = *reinterpret_cast<const u32x4*>(nx + 1024);
        T.cur[2] = *reinterpret_cast<const u32x4*>(nx + 2048);
        pipeline_fence();
        if constexpr (TWO) mfma_step_s2(wh, wm, wl, act[s], acc, acc2);
        else acc = mfma_step_s(wh, wm, wl, act[s], acc);
    }
    g += (H + NS) / N_CHS;
    if constexpr (TWO) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
    }
    return acc;
}

template <int NR>                                    // NR fp32 registers -> NR / 8 Split8
__device__ __forceinline__ void split_regs(const float (&x)[NR], Split8* out) {
#pragma unroll
    for (int g = 0; g < NR / 8; ++g) out[g] = split8(&x[8 * g]);
}

// nf 256: 252 registers, two workgroups per CU (two waves per SIMD), no scratch.  nf 384: [e ; G] alone is 12 Split8 = 144 registers; held
// to 256 the kernel spills 296 B per lane and is still the faster form (GEOM nf 384 B = 1250, pair update per block: fp32 4 209 us, one wave
// per SIMD with 312 registers 3 283, two waves per SIMD 3 142: profiles/r06_split_ab_geom384.txt).
template <int D, int R>
__global__ __launch_bounds__(SPLIT_WAVES * 64, 2) void k_edge_update_sym_split(KArgs A) {
    static_assert(D == 256 || D == 384, "the split-bf16 pair update is built for nf = 256 and nf = 384");
    if (A.flags[FLAG_ASYM] || !A.flags[FLAG_UNIFORM_T]) return;      // (the launcher only runs this under both pins; a violated pin is
                                                                      // reported by k_finalize_nodes like for every pinned launch)
    using X = wide::Dim<D>;
    constexpr int NCH = R * X::De / 64;              // hidden chunks of the edge FFN
    constexpr int NSE = X::De / 16;                  // K16 steps of a K = De projection (4)
    constexpr int NB2 = 2 * X::NE;                   // blocks of the triangular factor L (4)
    constexpr int NSZ = 2 * NSE;                     // steps of a K = 2 De projection (8)
    constexpr int STATIC_STEPS = NCH * (2 * NSE + X::NE * 4) + NSE + NB2 * (NB2 + 1);
    constexpr int CHUNK_STEPS_FFN = 2 * NSE + X::NE * 4;   // steps of one hidden chunk: two ff_linear3 blocks + NE ff_linear4 pieces
    static_assert(STATIC_STEPS % CH_STEPS == 0 && NSZ % CH_STEPS == 0 && CHUNK_STEPS_FFN % CH_STEPS == 0 && (2 * NSE) % CH_STEPS == 0 && CH_STEPS == 4,
                  "tape sections end on chunk boundaries (K = De blocks of 6 steps at nf 384 start 0 or 2 steps into a chunk)");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jl = lane & 31, half = lane >> 5;
    // items: workgroup w lives on XCD w % 8; the plan's item order puts the items of XCD x at indices = x (mod 8) (xcd_order), so the
    // four waves take four of "their" XCD's items
    int it = ((int)blockIdx.x >> 3) * 32 + ((int)blockIdx.x & 7) + 8 * wave;
    const bool live = it < A.pd.n_pitems;            // an idle wave walks item 0 without stores: the ring needs all four waves
    if (!live) it = 0;
    const int strip = A.pd.pitem_strip[it], t = A.pd.pitem_t0[it];
    const LaneNode L = lane_node(A, strip, jl);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float* eg1 = mrow + X::M_EDGE + 2 * X::De;
    const float gscale = mrow[X::M_GBF + 0], gshift = mrow[X::M_GBF + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float cscale = A.W[A.wb[JB_CSCALE]];

    __shared__ u32x4 ring[RING_SLOTS * CH_BYTES / 16];
    __shared__ float4 w2s[3 * D / 4];
    Tape T;
    T.rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A.wsplit), 0, 0x7fffffff, 0x00020000);
    T.rs1 = __builtin_amdgcn_make_buffer_rsrc(A.mfold_s + (size_t)A.layer * D * 2 * X::De * 3, 0, 0x7fffffff, 0x00020000);
    T.n0 = STATIC_STEPS / CH_STEPS;
    T.ntot = T.n0 + X::ND * NSZ / CH_STEPS;
    T.ld_off = (unsigned)wave * 3072u + (unsigned)lane * 16u;
    T.rd_off = (unsigned)lane * 16u;
    T.ring = reinterpret_cast<char*>(ring);
    {
        const float4* src = reinterpret_cast<const float4*>(A.W + A.wb[JB_C2_W]);
        for (int i = threadIdx.x; i < 3 * D / 4; i += SPLIT_WAVES * 64) w2s[i] = src[i];
    }
    tape_start(T);                                    // (the first boundary's barrier also publishes w2s)
    int g = 0;                                        // tape chunk being consumed

    const PairLane P = pair_of(L, t + 1);
    const bool okw = P.ok && live;
    const float* eg1_ = launder(eg1);
    const float* es2_ = eg1_ + X::De, *ec2_ = es2_ + X::De, *eg2_ = ec2_ + X::De;
    const float* cst = launder(A.W);
    const float* n2bias_ = cst + A.wb[JB_N2E_B], *b3_ = cst + A.wb[JB_FF3_B], *b4_ = cst + A.wb[JB_FF4_B];
    const float* tab_ = cst + A.wb[JB_GBF], *bro_ = cst + A.wb[JB_ERO_B];
    const BRow wrow_i = brow(A.wrow, X::ND, L.v, half), wcol_i = brow(A.wcol, X::ND, L.v, half);
    const BRow wrow_j = brow(A.wrow, X::ND, P.u, half), wcol_j = brow(A.wcol, X::ND, P.u, half);
    const BRow ua_i = brow(A.ua, X::ND, L.v, half), ub_i = brow(A.ub, X::ND, L.v, half);
    const BRow ua_j = brow(A.ua, X::ND, P.u, half), ub_j = brow(A.ub, X::ND, P.u, half);
    const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[P.u];
    const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
    const float d2 = dx * dx + dy * dy + dz * dz;
    // ---- edge residual + LN2 + modulate (symmetric) ----
    float en[X::HE];
    {
        const TRow ra = trow(A.n2e, X::NE, L.v, half), rc = trow(A.n2e, X::NE, P.u, half);
#pragma unroll
        for (int b = 0; b < X::NE; ++b) {
            float e[16], ta[16], tc2[16], g[16], bb[16];
            load16(A.e + P.rij * X::De + b * 32 + half * 16, e);
            load16T(ra, b, ta);
            load16T(rc, b, tc2);
            load16(eg1_ + b * 32 + half * 16, g);
            load16(n2bias_ + b * 32 + half * 16, bb);
#pragma unroll
            for (int s = 0; s < 16; ++s) en[b * 16 + s] = fmaf(g[s], ta[s] + tc2[s] + bb[s], e[s]);
        }
    }
    layer_norm<X::HE>(en);
    modulate<X::NE>(en, es2_, ec2_, half);
    Split8 zs[NSZ];                                   // [en ; G] as split operands: steps 0 .. NSE - 1 = en, NSE .. = G (filled behind the FFN)
    split_regs<X::HE>(en, zs);
    // ---- edge FFN ----
    {
        f32x16 o[X::NE];
#pragma unroll
        for (int b = 0; b < X::NE; ++b) o[b] = zero16();
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            float hid[32];
            static_for<2>([&](auto bc) {
                constexpr int b2 = decltype(bc)::value;
                float bb[16];
                load16(b3_ + (c * 2 + b2) * 32 + half * 16, bb);
                const f32x16 acc = tape_block<NSE, (b2 * NSE) % CH_STEPS>(T, g, zs, zero16());
                silu_bias16(acc, bb, hid + b2 * 16);
            });
            Split8 hs[4];
            split_regs<32>(hid, hs);
#pragma unroll
            for (int ob = 0; ob < X::NE; ++ob) o[ob] = tape_block<4, 0>(T, g, hs, o[ob]);
        });
#pragma unroll
        for (int b = 0; b < X::NE; ++b) {
            float ob4[16], og2[16];
            load16(b4_ + b * 32 + half * 16, ob4);
            load16(eg2_ + b * 32 + half * 16, og2);
#pragma unroll
            for (int s = 0; s < 16; ++s) en[b * 16 + s] = fmaf(og2[s], o[b][s] + ob4[s], en[b * 16 + s]);
        }
    }
    if (okw) {
        store_nat<X::NE>(A.e_out + P.rij * X::De, half, en);
        if (!A.half_rows || L.n > PAIR_GROUP_LANES) store_nat<X::NE>(A.e_out + P.rji * X::De, half, en);
    }
    split_regs<X::HE>(en, zs);                        // the block's output edge state feeds the readout, L and Z
    {   // the Gaussian basis of this block's distances: only L and Z read it, so it is evaluated here, behind the FFN (48 registers
        // fewer during the FFN than the f32 kernel's order; the pairing of two waves per SIMD needs them)
        float G[X::HE];
        gbf_n<X::NE>(d2, gscale, gshift, tab_, half, G);
        split_regs<X::HE>(G, zs + NSE);
    }
    const float gr0 = A.gramE[P.rij], gr1 = A.gramE[P.rji];
    // ---- readout ----
    {
        float bb[16];
        load16(bro_ + half * 16, bb);
        const f32x16 acc = tape_block<NSE, 0>(T, g, zs, zero16());      // (the readout starts a chunk: NCH whole FFN chunks precede it)
        float rr[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) rr[s] = acc[s] + bb[s];
        if (okw && (half == 0 || A.d.cep == 32)) {
            store16(A.ehid + P.rij * A.d.KEH + X::De + A.layer * A.d.cep + half * 16, rr);
            if (!A.half_rows || L.n > PAIR_GROUP_LANES) store16(A.ehid + P.rji * A.d.KEH + X::De + A.layer * A.d.cep + half * 16, rr);
        }
    }
    // ---- rotated LayerNorm statistics: D var(pre) = |L z + Rq_a[:2De] + Cq_c[:2De]|^2 + Gram term (dgt_kernels_wide.h) ----
    f32x2 q02 = {0.f, 0.f}, q12 = {0.f, 0.f};
    // (two waves share a SIMD here: the partner's MFMAs cover this wave's row gathers, so the rows of a block are requested at its
    // start and consumed behind its MFMAs — the one-wave f32 kernel's block-ahead prefetch would cost 64 registers the pairing needs)
    static_for<NB2>([&](auto kc) {
        constexpr int k = decltype(kc)::value, b = NB2 - 1 - k;
        float n0[16], n1[16], n2[16], n3[16];
        bload16(wrow_i, b, n0); bload16(wcol_j, b, n1); bload16(wrow_j, b, n2); bload16(wcol_i, b, n3);
        pipeline_fence();
        const f32x16 acc = tape_block<2 * (NB2 - b), (NSE + k * (k + 1)) % CH_STEPS>(T, g, zs + 2 * b, zero16());      // blocks of 2, 4, 6, 8 (, 10, 12) steps behind the readout's NSE
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            const f32x2 sv = pk2(acc[s], acc[s + 1]);
            const f32x2 d0 = sv + (pk2(n0[s], n0[s + 1]) + pk2(n1[s], n1[s + 1])), d1 = sv + (pk2(n2[s], n2[s + 1]) + pk2(n3[s], n3[s + 1]));
            q02 = __builtin_elementwise_fma(d0, d0, q02);
            q12 = __builtin_elementwise_fma(d1, d1, q12);
        }
    });
    const float rstd0 = __builtin_amdgcn_rsqf(fmaxf((pair_sum(q02.x + q02.y) + gr0) * (1.f / D), 0.f) + 1e-6f);
    const float rstd1 = __builtin_amdgcn_rsqf(fmaxf((pair_sum(q12.x + q12.y) + gr1) * (1.f / D), 0.f) + 1e-6f);
    // ---- Z' = M [e ; G] block by block, SiLU / coord_mlp.2 tails of both directions ----
    const float* bs_v = launder(mrow + X::M_WG) + D;
    f32x2 c00 = {0.f, 0.f}, c01 = c00, c02 = c00, c10 = c00, c11 = c00, c12 = c00;
#pragma unroll 1
    for (int b = 0; b < X::ND; ++b) {
        float t0[16], t1[16], bsb[16];
        {
            float n0[16], n1[16], n2[16], n3[16];
            bload16(ua_i, b, n0); bload16(ub_j, b, n1); bload16(ua_j, b, n2); bload16(ub_i, b, n3);
            load16(bs_v + b * 32 + half * 16, bsb);
            pipeline_fence();
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                const f32x2 a = pk2(n0[s], n0[s + 1]) + pk2(n1[s], n1[s + 1]), c = pk2(n2[s], n2[s + 1]) + pk2(n3[s], n3[s + 1]);
                t0[s] = a.x; t0[s + 1] = a.y; t1[s] = c.x; t1[s + 1] = c.y;
            }
        }
        const f32x16 z = tape_block<NSZ, 0>(T, g, zs, zero16());
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            float k0[8], k1[8], k2[8];
            auto ld8 = [&](const float* p8, float (&r)[8]) {
                const float4 a = reinterpret_cast<const float4*>(p8)[0], c = reinterpret_cast<const float4*>(p8)[1];
                r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = c.x; r[5] = c.y; r[6] = c.z; r[7] = c.w;
            };
            const int fo = b * 32 + half * 16 + hq * 8;
            const float* w2l = reinterpret_cast<const float*>(w2s);
            ld8(w2l + fo, k0); ld8(w2l + D + fo, k1); ld8(w2l + 2 * D + fo, k2);
            pipeline_fence();
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                const f32x2 bs2 = pk2(bsb[hq * 8 + s], bsb[hq * 8 + s + 1]);
                const f32x2 pre = pk2(z[hq * 8 + s], z[hq * 8 + s + 1]) + pk2(t0[hq * 8 + s], t0[hq * 8 + s + 1]);
                const f32x2 ys0 = silu_f2(__builtin_elementwise_fma(pre, (f32x2)(rstd0), bs2));
                c00 = __builtin_elementwise_fma(ys0, pk2(k0[s], k0[s + 1]), c00);
                c01 = __builtin_elementwise_fma(ys0, pk2(k1[s], k1[s + 1]), c01);
                c02 = __builtin_elementwise_fma(ys0, pk2(k2[s], k2[s + 1]), c02);
            }
            pipeline_fence();
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                const f32x2 bs2 = pk2(bsb[hq * 8 + s], bsb[hq * 8 + s + 1]);
                const f32x2 pre = pk2(z[hq * 8 + s], z[hq * 8 + s + 1]) + pk2(t1[hq * 8 + s], t1[hq * 8 + s + 1]);
                const f32x2 ys1 = silu_f2(__builtin_elementwise_fma(pre, (f32x2)(rstd1), bs2));
                c10 = __builtin_elementwise_fma(ys1, pk2(k0[s], k0[s + 1]), c10);
                c11 = __builtin_elementwise_fma(ys1, pk2(k1[s], k1[s + 1]), c11);
                c12 = __builtin_elementwise_fma(ys1, pk2(k2[s], k2[s + 1]), c12);
            }
            pipeline_fence();
        }
    }
    const float nrm = fmaxf(sqrtf(d2), 1e-8f);
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        const float c0 = tanh_f(pair_sum(dir == 0 ? c00.x + c00.y : c10.x + c10.y));
        const float c1 = tanh_f(pair_sum(dir == 0 ? c01.x + c01.y : c11.x + c11.y));
        const float c2 = tanh_f(pair_sum(dir == 0 ? c02.x + c02.y : c12.x + c12.y));
        const size_t rr = dir == 0 ? P.rij : P.rji;
        const int fl = A.eflag[rr];
        const float iota = (c0 + ((fl & 1) ? c1 : 0.f) + ((fl & 2) ? c2 : 0.f)) * (1.f / 3.f);
        const float f = cscale * iota / nrm;
        const float sgn = dir == 0 ? 1.f : -1.f;
        if (okw && half == 0)
            reinterpret_cast<float4*>(A.dposE)[rr] = make_float4(sgn * dx * f, sgn * dy * f, sgn * dz * f, 0.f);
    }
}

}  // namespace split
}  // namespace jd
