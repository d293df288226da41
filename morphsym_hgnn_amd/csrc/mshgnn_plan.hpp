// Host-side plan compiler of the MS-HGNN engine: lowers (topology, model dims, symmetry masks, parameter
// offsets) to the integer tables the HIP kernels interpret.  No HIP calls in this file, so it is testable
// on a machine without a GPU (mshgnn_plan_create builds it, then uploads the tables).
//
// Path restated: GRF_HGNN_C2.forward hgnn_c2.py:133-182 / GRF_HGNN_K4.forward hgnn_k4.py:146-196 /
// GRF_HGNN.forward hgnn.py:57-62 with PyG-2.5.0 HeteroConv/GraphConv semantics:
//   H_d = sum_{r=(s,.,d)} [ W_rel^r Agg_r(X_s) + b_rel^r + W_root^r X_d ]
// lowered to its algorithmic minimum:  H_d[i] = (sum_r W_root^r) X_d[i] + sum_r b_rel^r
//                                              + sum_r sum_{j->i in r} W_rel^r X_s[j]
// Node types whose layer output cannot reach the decoder (e.g. base/joint in the last layer) are dead and
// skipped in forward and backward; their parameters get exact-zero gradients (the reference leaves them None).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mshgnn.h"

// Run-time switches.  The product library reads the ones README.md documents (what selects a route a test or the bench compares: MSHGNN_SPEC, _PRUNE, _SLAB,
// _STEP_KERNEL, _FUSED, _ENGINE, _STASH_NT, _STEP_CHUNK, _GEN_TILE, _GEN_MANY).  Everything that only ever served a measurement (grid targets, placement
// heuristics, alternative kernels kept for A/B runs) is read in TUNING builds only (make EXTRA=-DMSHGNN_TUNING): in the product build TUNE_ENV() is a constant,
// the branches behind it fold away and the kernels only they reach are not in the binary.
#ifdef MSHGNN_TUNING
#define TUNE_ENV(name) std::getenv(name)
#else
#define TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace mshgnn {

constexpr int H = 128;            // hidden width the kernels are built for
constexpr int GMAX = 12;          // accumulator slots (dst nodes) a workgroup keeps live at once
constexpr int TILE_ROWS = 16;      // windows per layer-kernel tile (one LDS node block = 16 x 128 elements)
constexpr int MAX_L = 16;
#ifndef LDS_LIMIT_KB
#define LDS_LIMIT_KB 160
#endif
constexpr int LDS_LIMIT = LDS_LIMIT_KB * 1024;
constexpr int SLAB_FLOATS = H * H + H;   // one split-K partial: 128x128 matrix + 128 column sums
constexpr int NWG_DEC = 512;             // workgroups of the decoder backward (each writes one small slab)
constexpr int DEC_SLAB_FLOATS = 8 * H + 16; // decoder partial: [out_channels<=8][128] + bias[8] + loss partial (+pad)
#ifndef GW_IPL_MAX
#define GW_IPL_MAX 2                     // measured with 4 and 8 (EXTRA=-DGW_IPL_MAX=..): the grid fills better (A1-C2 L=3: 704 -> 742 of 768 workgroups,
#endif                                   // MiniCheetah-K4 L=8: 560 -> 714) and the kernel gets SLOWER (110 -> 152 us, 333 -> 372 us): lanes of 3-4 interleaved items
                                         // reach a shared P / Q stream steps apart, and a stream survives in an XCD's L2 for about one step
constexpr int GW_IPL = GW_IPL_MAX;       // most items per weight-gradient lane; the plan picks gw_ipl in 1 .. GW_IPL: whichever fills most of
                                         // the GW_TARGET_WGS resident workgroups (lanes x window parts), the smallest on a tie (K4 L=8, 560 workgroups either
                                         // way: 1 item x 1 part 328 us, 2 items x 2 parts 340 us; split plan 646 vs 712).  The lean kernels
                                         // interleave a lane's items chunk by chunk (any number); the general bf16 kernel (MSHGNN_GRADW=general) at most two
#ifndef GW_TARGET_WGS
#define GW_TARGET_WGS 768.0              // workgroups of the weight-gradient kernel (lanes x window parts): 3 resident per CU x 256 CUs, so
                                         // that the whole grid runs as ONE wave of workgroups (1024: a third of them ran in a second,
                                         // mostly empty wave: 128 vs 112 us, and more slabs for k_finalize)
#endif

#ifndef GW_XCD_SLACK
#define GW_XCD_SLACK 1                   // lanes an XCD queue of the weight-gradient kernel may exceed the balanced length by (placement by operand sharing)
#endif

enum { OP_LOADW = 0, OP_MAC = 1 };
enum { KIND_RELU = 0, KIND_MLP = 1 };
enum { NK_DEAD = 0, NK_RELU = 1, NK_MLP = 2 };
enum { GF_RESIDUAL = 1, GF_ENC_MASK = 2, GF_STORE_MASK = 4, GF_LDS_EPI = 8 };

// layer program = [n_groups] + n_groups group headers (GH_SIZE ints each) + two WAVE PROGRAMS of WPROG_LEN ints (one per
// slot half) holding every group's MAC program back to back (group g starts at entry GH_PC0/GH_PC1 of its half).
// A wave loads its program into two VGPRs ONCE per kernel and interprets it with v_readlane (no memory latency in the
// MAC loop):   per group: [nseg, then per segment: pack, then per accumulator u (slot = 2u + half): count, src blocks...]
enum { GH_KIND = 0, GH_NSLOTS, GH_BIAS, GH_NSEG, GH_W1, GH_W2, GH_B1, GH_B2, GH_FLAGS, GH_PC0, GH_PC1, GH_PAD2,
       GH_NODES = 12, GH_MLPIDX = 12 + GMAX, GH_SCR = 12 + 2 * GMAX, GH_SIZE = 12 + 3 * GMAX };
constexpr int WPROG_LEN = 128;
// backward program header: [n_groups, n_mlp_live, w2pack, w1pack, 0,0,0,0, node_kind[64], (64 unused), mlp_nodes[GMAX]]
enum { BH_NGROUPS = 0, BH_NMLP, BH_W2, BH_W1, BH_KIND = 8, BH_MLPNODES = 8 + 128, BH_SIZE = 8 + 128 + GMAX };

// FUSED STACK kernels (bf16 plan): one workgroup keeps its 16-window tile of ALL nodes in LDS through every layer, so a
// layer's output never makes a round trip through HBM before the next layer reads it (the stashes for the backward
// pass are written on the side).  Every node owns one accumulator for the whole layer: node n -> wave half n & 1,
// accumulator n >> 1, so a layer is ONE group of <= 2 FS_HS slots and its epilogue runs once, after every wave has
// finished reading the previous activations (which it then overwrites in place).
constexpr int FS_HS = 10;                // accumulators per wave
constexpr int FS_MAXN = 2 * FS_HS;       // nodes per window the fused kernels support
constexpr int FPROG_LEN = 192;           // ints per wave program (three VGPRs)
// per-layer header (same layout forward / backward), then two wave programs (one per half) of FPROG_LEN ints:
//   ints [0, 64): pack id of segment s;  ints [64, 128): MAC counts of segment s, 3 bits per accumulator u (node 2u + half);
//   ints [128, 192): 256 byte entries, 4 per int, LSB first: [nseg, then the BLOCK stream: the source blocks of all MACs in
//   execution order (+ one pad entry, so the kernel can always fetch the block after the current one and prefetch it)]
enum { FH_NSEG = 0, FH_NMLP, FH_W1, FH_W2, FH_B1, FH_B2, FH_FLAGS, FH_MLP0,   // FH_MLP0: first node of the base_transform type
       FH_KIND = 8,                      // NK_* of node n in this layer
       FH_BIAS = 8 + FS_MAXN,            // fwd: bias row of node n
       FH_OUT = 8 + 2 * FS_MAXN,         // bwd: dX_l[n] is produced
       FH_RES = 8 + 3 * FS_MAXN,         // bwd: dX_{l+1}[n] flows into dX_l[n] through the residual
       FH_SLOTA = 8 + 4 * FS_MAXN,       // slab kernels: node of accumulator slot u of group A / group B (-1: unused)
       FH_SLOTB = FH_SLOTA + 16,
       FH_SIZE = FH_SLOTB + 16,
       FH_NEXT = FH_SLOTA };             // engine-driven kernels (their header copy has no slot arrays).  bwd: NK_* | residual << 2 of node n in layer l - 1;
                                         // fwd: [0, 4) bias row of type t in this layer, [4, 8) in the next layer, [8, 28) type of node n
static_assert(FH_NEXT + 8 + 20 <= FH_SIZE && MSHGNN_MAX_TYPES <= 4, "header: next-layer array");
// Slab variant of the stack kernels (4-wave workgroups, two per CU): wave wn owns columns [32 wn, 32 wn + 32) of EVERY node, so
// a weight pack goes through the CU's vector L1 once per tile instead of once per wave half (the stack kernels' MAC phase is
// bound by that path: DESIGN.md section 6).  To keep the accumulators in registers the destination nodes are processed in two
// groups, one after the other: A = the node type with the most nodes (<= SL_HA), B = all other nodes (<= SL_HB); group A's
// results wait, packed, while group B is multiplied.  Wave programs as above with 2 (group A) / 3 (group B) bits of MAC count per slot.
// Group B holds 6 slots (A1-C2, the MI graph, the COM graphs) or 8 (MiniCheetah-K4: 4 base + 4 foot nodes): the kernels are instantiated for both
// (template parameter HB), the plan says which (sl_hb).  Where the tile + base_transform scratch blocks do not fit twice into a CU's LDS but
// the tile alone does (K4: 20 + 4 blocks), the scratch blocks ALIAS the first group-A nodes (FF_SCR_ALIAS: one more barrier per layer with a
// live base_transform; group A's residual reads are over by then and its new rows wait in registers).
constexpr int SL_HA = 12, SL_HB = 6, SL_HB_MAX = 8, SL_THREADS = 256;
constexpr int SL_CBA = 2, SL_CBB = 3;    // bits of MAC count per slot and segment in group A / B programs (12 x 2, 6..8 x 3 bits)
enum { FF_RESIDUAL = 1, FF_ENC_MASK = 2, FF_SCR_ALIAS = 4, FF_A_EMPTY = 8, FF_B_EMPTY = 16 };      // FF_x_EMPTY (slab headers): group x has no work in this layer

// buffer ids used by weight-gradient items
enum { BUF_X = 0, BUF_DX = 17, BUF_DH = 34, BUF_HB = 50, BUF_T1 = 66, BUF_DU = 82, BUF_IN = 98, BUF_MASK = 102, BUF_COUNT = 118 };
// last item int: relu-bit buffer (BUF_MASK + l) when P = dX_{l+1}[node] . relu bits (dH of a relu node is recomputed
// by the weight-gradient kernel instead of being written by k_layer_bwd and read back), else -1
constexpr int ITEM_INTS = 10;   // p_buf p_nodes*H p_node q_buf (q_nodes*H | -1: raw input) q_node q_col0 q_ncols sign_off pad
constexpr int TGT_INTS = 8;     // lane_begin lane_end bias_flag pad..
constexpr int LANE_INTS = 4;    // item_begin item_end target bias_flag
constexpr int FIN_INTS = 8;     // dst_lo dst_hi rows cols dst_ld target kind extra (0, or the offset -- in ints, from the fin table's start -- of {n, dst_lo, dst_hi, ...}: further
                                // destinations of the same shape that receive the same slab sum: lin_root.weight / lin_rel.bias of every relation into one destination type)
enum { FIN_MATRIX = 0, FIN_BIAS = 1, FIN_ZERO = 2, FIN_DEC_W = 3, FIN_DEC_B = 4 };

struct PackDesc {       // one packed 128x128 B-operand image (weights as the MFMA wants them)
    int orient;         // 0: B[k][c] = W[c][k]  (forward: out = A W^T)   1: B[k][c] = W[k][c]  (backward: dA = dH W)
    int n_src;          // matrices summed into it (root-sum over relations)
    int64_t src[8];
    int ld;             // row pitch of the source matrices
    int col0;           // first source column (encoder K chunk)
    int ncols;          // valid source columns from col0 (zero beyond)
};
struct BiasDesc { int n_src; int64_t src[8]; };

struct HostPlan {
    mshgnn_desc d{};
    void* jit_prog = nullptr;      // selector of a program compiled for THIS plan after the library was built (mshgnn_plan_attach_program; same signature as the shards' selectors)
    std::vector<int32_t> rel_src, rel_dst, rel_mean, rel_edge_off, edges;
    std::vector<float> in_mask[MSHGNN_MAX_TYPES], out_mask;
    std::vector<int64_t> off_enc_w, off_enc_b, off_rel_w, off_rel_b, off_root_w;
    int L = 0, NT = 0, NR = 0, NN = 0;
    int type_base[MSHGNN_MAX_TYPES + 1]{};
    int node_type[64]{};
    int n_mlp = 0;
    int n_blk = 0;                 // LDS node blocks of the layer kernels (NN + scratch blocks)
    bool mlp_scratch = false;      // base_transform uses dedicated scratch blocks (else its own node blocks, and runs last)
    int rows = TILE_ROWS;          // windows per tile
    int blk_bytes = 8192;          // one LDS node block
    int gmax = GMAX;               // destination slots per group (12 fp32, 8 bf16: accumulator registers)
    int esize = 4;                 // bytes per stored element
    bool split = false;            // MSHGNN_BF16X3: every activation is stored as hi and lo bf16 halves; packs [0, n_img) = hi images, [n_img, 2 n_img) = lo
    int planes = 1;                // bf16 halves per stored value (2 on the split plan: rows of [hi 128 | lo 128])
    int n_img = 0;                 // weight images per plane (== packs.size())
    int lo_blk = 0;                // split plan: LDS block of the lo plane of node n is lo_blk + n
    int gw_target = (int)GW_TARGET_WGS;
    bool live[MAX_L][MSHGNN_MAX_TYPES]{};     // layer output of SOME node of type t reaches the decoder
    bool need_dx[MAX_L][MSHGNN_MAX_TYPES]{};  // dX_l of some node of type t must be produced
    // node-level liveness (round 4): live_n[l][n] -- X_{l+1}[n], the output of layer l at node n, can reach the decoder; need_n[l][n] -- X_l[n] is an input
    // of a live node of layer l (itself through the root weight / residual, or a destination it has an edge to), so dX_l[n] must be produced; need_n[0] =
    // the nodes the encoder has to compute.  A1-C2 at L = 3: the base nodes are four hops from the feet, so nothing of them is live (49 of 84 parameter
    // tensors get exact-zero gradients in the reference too).  MSHGNN_PRUNE=0: type-level liveness only (every node of a live type is live).
    bool live_n[MAX_L][64]{}, need_n[MAX_L][64]{};
    bool prune_nodes = true;
    int enc_live_nodes = 0;                   // nodes the encoder computes (need_n[0])
    // packed operands
    std::vector<PackDesc> packs;
    std::vector<BiasDesc> biases;
    std::vector<int> pack_root[2], pack_rel[2];   // [orient][l*NT+t], [orient][l*NR+r]
    int pack_mlp[2][2]{};                         // [orient][0|1]
    std::vector<int> pack_enc_base; std::vector<int> enc_nkc;   // per type
    std::vector<int> bias_layer;                  // [l*NT+t]
    int bias_mlp[2]{}; std::vector<int> bias_enc;
    // sign tables (uint8, [n_t][nkc*128]) concatenated; offset per type
    std::vector<uint8_t> signs; std::vector<int> sign_off;
    // programs
    std::vector<int32_t> tables;                  // everything below lives here (device copy = same layout)
    int fwd_prog_off[MAX_L]{}, bwd_prog_off[MAX_L]{};
    bool fused = false;                           // fused stack kernels available (bf16, <= FS_MAXN nodes, programs fit)
    int fs_fwd_off[MAX_L]{}, fs_bwd_off[MAX_L]{};
    bool slab = false;                            // slab variant available (two 4-wave workgroups per CU fit, groups fit)
    int sl_fwd_off[MAX_L]{}, sl_bwd_off[MAX_L]{}, sl_ta = -1, sl_hb = SL_HB, sl_blk = 0;      // sl_hb: group-B slots (6 / 8); sl_blk: LDS blocks of a slab workgroup
    bool sl_alias = false;                        // base_transform scratch aliases group-A node blocks
    int fs_blk = 0;                               // LDS blocks of the fused kernels (NN + base_transform scratch)
    bool x3_alias = false;                        // split plan: the base_transform scratch aliases the last n_mlp node blocks
    int ks_stack_fwd = -1, ks_stack_bwd = -1;
    int item_off = 0, n_items = 0, tgt_off = 0, n_targets = 0, lane_off = 0, n_lanes = 0, n_parts = 1, n_wg_gradw = 0;
    int lane_order_off = 0, n_lanes_pad = 0, gw_ipl = 1;
    // two-phase weight gradients (multi-GPU overlap): phase 0 = every lane but the encoder's (its gradients are all-reduced
    // while phase 1 = the encoder lanes runs); fin ops are stored phase 0 first
    int order_ph_off[2]{}, npad_ph[2]{}, n_fin_ph0 = 0;
    int64_t grad_split = -1;     // flat offset where the phase-0 parameters start ([0, grad_split) = encoder = phase 1); -1: no split
    int fin_off = 0, n_fin = 0;
    int enc_tile_mb = 4;
    int n_slabs = 0;
    std::vector<float> out_mask_f;
    std::vector<mshgnn_kernel_stat> kstats;   // one per kernel of a step, in launch order
    int ks_prep = 0, ks_enc = 0, ks_layer_fwd0 = 0, ks_dec_fwd = 0, ks_dec_bwd = 0, ks_layer_bwd0 = 0, ks_gradw = 0, ks_fin = 0, ks_stack_step = 0;
    mshgnn_info info{};
    std::string err;
};

inline bool fail(HostPlan& p, const std::string& m) { p.err = m; return false; }

inline int add_pack(HostPlan& p, int orient, const std::vector<int64_t>& src, int ld, int col0, int ncols) {
    PackDesc q{}; q.orient = orient; q.n_src = (int)src.size(); q.ld = ld; q.col0 = col0; q.ncols = ncols;
    for (size_t i = 0; i < src.size(); ++i) q.src[i] = src[i];
    p.packs.push_back(q); return (int)p.packs.size() - 1;
}
inline int add_bias(HostPlan& p, const std::vector<int64_t>& src) {
    BiasDesc b{}; b.n_src = (int)src.size(); for (size_t i = 0; i < src.size(); ++i) b.src[i] = src[i];
    p.biases.push_back(b); return (int)p.biases.size() - 1;
}

inline bool compile_plan(const mshgnn_desc* din, HostPlan& p) {
    if (!din) return fail(p, "null descriptor");
    p.d = *din;
    const mshgnn_desc& d = p.d;
    if (d.n_types < 1 || d.n_types > MSHGNN_MAX_TYPES) return fail(p, "n_types must be 1..4");
    if (d.hidden != H) return fail(p, "this build supports hidden_channels == 128 only");
    if (d.num_layers < 1 || d.num_layers > MAX_L) return fail(p, "num_layers must be 1..16");
    if (d.n_rel < 1 || d.n_rel > 64) return fail(p, "n_rel must be 1..64");
    if (d.dtype != MSHGNN_F32 && d.dtype != MSHGNN_BF16 && d.dtype != MSHGNN_BF16X3) return fail(p, "dtype must be MSHGNN_F32, MSHGNN_BF16 or MSHGNN_BF16X3");
    if (d.out_type < 0 || d.out_type >= d.n_types) return fail(p, "out_type out of range");
    if (d.out_channels < 1 || d.out_channels > 8) return fail(p, "out_channels must be 1..8");
    if (!d.rel_src || !d.rel_dst || !d.rel_mean || !d.rel_edge_off || !d.edges) return fail(p, "null relation arrays");
    if (!d.off_enc_w || !d.off_enc_b || !d.off_rel_w || !d.off_rel_b || !d.off_root_w) return fail(p, "null offset arrays");
    p.L = d.num_layers; p.NT = d.n_types; p.NR = d.n_rel;
    const int L = p.L, NT = p.NT, NR = p.NR;
    const bool has_mlp = (d.flags & MSHGNN_FLAG_BASE_MLP) != 0;
    const bool residual = (d.flags & MSHGNN_FLAG_RESIDUAL) != 0;
    if (has_mlp && (d.mlp_type < 0 || d.mlp_type >= NT)) return fail(p, "mlp_type out of range");
    p.type_base[0] = 0;
    {   // (before node_type[] is filled: a 129-node topology at hidden = 128 used to run past its 64 entries here instead of being handed to the generic engine)
        int64_t total = 0;
        for (int t = 0; t < NT; ++t) total += std::max(0, d.type_nodes[t]);
        if (total > 64) return fail(p, "topology has too many nodes per window for the LDS-resident layer kernel (max 20 fp32 / 40 bf16)");
    }
    for (int t = 0; t < NT; ++t) {
        if (d.type_nodes[t] < 1) return fail(p, "every node type needs >= 1 node");
        if (d.type_width[t] < 1) return fail(p, "every node type needs input width >= 1");
        p.type_base[t + 1] = p.type_base[t] + d.type_nodes[t];
        for (int i = 0; i < d.type_nodes[t]; ++i) p.node_type[p.type_base[t] + i] = t;
    }
    p.NN = p.type_base[NT];
    p.esize = d.dtype == MSHGNN_F32 ? 4 : 2;
    p.split = d.dtype == MSHGNN_BF16X3; p.planes = p.split ? 2 : 1;
#ifndef GWX3_TARGET_WGS
#define GWX3_TARGET_WGS 768
#endif
    p.gw_target = p.split ? GWX3_TARGET_WGS : (int)GW_TARGET_WGS;
    { const char* e = TUNE_ENV("MSHGNN_GW_TARGET"); if (e && std::atoi(e) >= 64) p.gw_target = std::atoi(e); }      // (measurements: workgroups of the weight-gradient launch)     // split plan: three resident workgroups per CU of its weight-gradient kernel (four 8 KB tiles per 32-window step)
    p.blk_bytes = TILE_ROWS * H * p.esize;
    if (!p.split && (int64_t)p.NN * p.blk_bytes > LDS_LIMIT)
        return fail(p, "topology has too many nodes per window for the LDS-resident layer kernel (max 20 fp32 / 40 bf16)");
    p.rows = TILE_ROWS;
    p.gmax = d.dtype == MSHGNN_F32 ? GMAX : 8;   // bf16: 4 accumulators per wave keep the layer kernels at 128 VGPRs (2 workgroups / CU)
    p.n_mlp = has_mlp ? d.type_nodes[d.mlp_type] : 0;
    if (p.n_mlp > p.gmax) return fail(p, "base_transform type has too many nodes for one accumulator group");

    // deep copies
    p.rel_src.assign(d.rel_src, d.rel_src + NR); p.rel_dst.assign(d.rel_dst, d.rel_dst + NR);
    p.rel_mean.assign(d.rel_mean, d.rel_mean + NR); p.rel_edge_off.assign(d.rel_edge_off, d.rel_edge_off + NR + 1);
    const int E = p.rel_edge_off[NR];
    if (p.rel_edge_off[0] != 0 || E < 0) return fail(p, "bad rel_edge_off");
    p.edges.assign(d.edges, d.edges + 2 * (size_t)E);
    p.off_enc_w.assign(d.off_enc_w, d.off_enc_w + NT); p.off_enc_b.assign(d.off_enc_b, d.off_enc_b + NT);
    p.off_rel_w.assign(d.off_rel_w, d.off_rel_w + (size_t)L * NR); p.off_rel_b.assign(d.off_rel_b, d.off_rel_b + (size_t)L * NR);
    p.off_root_w.assign(d.off_root_w, d.off_root_w + (size_t)L * NR);
    std::vector<bool> has_in(NT, false);
    for (int r = 0; r < NR; ++r) {
        const int s = p.rel_src[r], t = p.rel_dst[r];
        if (s < 0 || s >= NT || t < 0 || t >= NT) return fail(p, "relation type index out of range");
        if (p.rel_edge_off[r + 1] < p.rel_edge_off[r]) return fail(p, "rel_edge_off must be non-decreasing");
        has_in[t] = true;
        std::vector<int> deg(d.type_nodes[t], 0);
        for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) {
            const int j = p.edges[2 * e], i = p.edges[2 * e + 1];
            if (j < 0 || j >= d.type_nodes[s] || i < 0 || i >= d.type_nodes[t]) return fail(p, "edge endpoint out of range");
            deg[i]++;
        }
        if (p.rel_mean[r]) for (int i = 0; i < d.type_nodes[t]; ++i)
            if (deg[i] > 1) return fail(p, "mean aggregation with in-degree > 1 is not supported by this build "
                                           "(all reference topologies have degree 1 on mean relations)");
    }
    for (int t = 0; t < NT; ++t)
        if (!has_in[t]) return fail(p, "every node type must be the destination of at least one relation "
                                       "(HeteroConv drops types without incoming relations)");
    // masks must be +-1
    p.sign_off.assign(NT, 0); p.enc_nkc.assign(NT, 0);
    for (int t = 0; t < NT; ++t) {
        const int F = d.type_width[t], n = d.type_nodes[t];
        const int nkc = (F + H - 1) / H; p.enc_nkc[t] = nkc;
        p.sign_off[t] = (int)p.signs.size();
        p.signs.resize(p.signs.size() + (size_t)n * nkc * H, 0);
        if (d.in_mask[t]) {
            for (int i = 0; i < n; ++i) for (int k = 0; k < F; ++k) {
                const float m = d.in_mask[t][(size_t)i * F + k];
                if (m != 1.0f && m != -1.0f) return fail(p, "symmetry masks must be +1 or -1");
                p.signs[p.sign_off[t] + (size_t)i * nkc * H + k] = m < 0 ? 1 : 0;
            }
        }
    }
    const int n_out = d.type_nodes[d.out_type];
    p.out_mask_f.assign((size_t)n_out * d.out_channels, 1.0f);
    if (d.out_mask) for (size_t i = 0; i < p.out_mask_f.size(); ++i) {
        if (d.out_mask[i] != 1.0f && d.out_mask[i] != -1.0f) return fail(p, "output mask must be +1 or -1");
        p.out_mask_f[i] = d.out_mask[i];
    }

    // ---- liveness (node level) ------------------------------------------------------------------
    // X_{l+1}[n] is needed iff node n itself is live in layer l + 1 (root weight, residual) or it has an edge into a node that is.  Without
    // pruning (MSHGNN_PRUNE=0) a type is live as a whole, as in rounds 1-3.  The base_transform chain works on its type's nodes together: if one of
    // them is live in a layer, all are.
    { const char* e = std::getenv("MSHGNN_PRUNE"); p.prune_nodes = !(e && std::atoi(e) == 0); }
    auto widen = [&](bool (&v)[64]) {      // type-level closure: whole types (no pruning), or the base_transform type
        for (int t = 0; t < NT; ++t) {
            if (p.prune_nodes && !(has_mlp && t == d.mlp_type)) continue;
            bool any = false;
            for (int i = 0; i < d.type_nodes[t]; ++i) any = any || v[p.type_base[t] + i];
            if (any) for (int i = 0; i < d.type_nodes[t]; ++i) v[p.type_base[t] + i] = true;
        }
    };
    auto inputs_of = [&](const bool (&out)[64], bool (&in)[64]) {      // the nodes whose X feeds a live node of a layer
        for (int n = 0; n < p.NN; ++n) in[n] = out[n];
        for (int r = 0; r < NR; ++r)
            for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e)
                if (out[p.type_base[p.rel_dst[r]] + p.edges[2 * e + 1]]) in[p.type_base[p.rel_src[r]] + p.edges[2 * e]] = true;
        if (!p.prune_nodes)      // (type-level rule of rounds 1-3: a type is needed when a relation out of it enters a live type, edges or not)
            for (int r = 0; r < NR; ++r) {
                bool dl = false;
                for (int i = 0; i < d.type_nodes[p.rel_dst[r]]; ++i) dl = dl || out[p.type_base[p.rel_dst[r]] + i];
                if (dl) for (int j = 0; j < d.type_nodes[p.rel_src[r]]; ++j) in[p.type_base[p.rel_src[r]] + j] = true;
            }
    };
    for (int n = 0; n < p.NN; ++n) p.live_n[L - 1][n] = (p.node_type[n] == d.out_type);
    for (int l = L - 1; l >= 0; --l) {
        widen(p.live_n[l]);
        inputs_of(p.live_n[l], p.need_n[l]);
        if (l == 0) widen(p.need_n[0]);      // (the encoder computes whole types without pruning)
        if (l > 0) for (int n = 0; n < p.NN; ++n) p.live_n[l - 1][n] = p.need_n[l][n];
    }
    // what layer l needs of X_l is what layer l - 1 has to produce: after the closure of live_n[l - 1] the two must agree again
    for (int l = 1; l < L; ++l) for (int n = 0; n < p.NN; ++n) p.need_n[l][n] = p.live_n[l - 1][n];
    p.enc_live_nodes = 0;
    for (int n = 0; n < p.NN; ++n) p.enc_live_nodes += p.need_n[0][n] ? 1 : 0;
    for (int l = 0; l < L; ++l)
        for (int t = 0; t < NT; ++t) {
            p.live[l][t] = p.need_dx[l][t] = false;
            for (int i = 0; i < d.type_nodes[t]; ++i) { p.live[l][t] = p.live[l][t] || p.live_n[l][p.type_base[t] + i]; p.need_dx[l][t] = p.need_dx[l][t] || p.need_n[l][p.type_base[t] + i]; }
        }
    auto rel_live = [&](int l, int r) {      // relation r has an edge into a live node of layer l
        for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) if (p.live_n[l][p.type_base[p.rel_dst[r]] + p.edges[2 * e + 1]]) return true;
        return false; };

    // ---- packed operands --------------------------------------------------------------------------
    for (int o = 0; o < 2; ++o) { p.pack_root[o].assign((size_t)L * NT, -1); p.pack_rel[o].assign((size_t)L * NR, -1); }
    p.bias_layer.assign((size_t)L * NT, -1);
    for (int l = 0; l < L; ++l) {
        for (int t = 0; t < NT; ++t) {
            if (!p.live[l][t]) continue;
            std::vector<int64_t> ws, bs;
            for (int r = 0; r < NR; ++r) if (p.rel_dst[r] == t) { ws.push_back(p.off_root_w[l * NR + r]); bs.push_back(p.off_rel_b[l * NR + r]); }
            if (ws.size() > 8) return fail(p, "more than 8 relations into one node type");
            p.pack_root[0][l * NT + t] = add_pack(p, 0, ws, H, 0, H);
            p.pack_root[1][l * NT + t] = add_pack(p, 1, ws, H, 0, H);
            p.bias_layer[l * NT + t] = add_bias(p, bs);
        }
        for (int r = 0; r < NR; ++r) {
            if (!rel_live(l, r)) continue;
            p.pack_rel[0][l * NR + r] = add_pack(p, 0, {p.off_rel_w[l * NR + r]}, H, 0, H);
            p.pack_rel[1][l * NR + r] = add_pack(p, 1, {p.off_rel_w[l * NR + r]}, H, 0, H);
        }
    }
    if (has_mlp) {
        for (int o = 0; o < 2; ++o) {
            p.pack_mlp[o][0] = add_pack(p, o, {d.off_mlp[0]}, H, 0, H);
            p.pack_mlp[o][1] = add_pack(p, o, {d.off_mlp[2]}, H, 0, H);
        }
        p.bias_mlp[0] = add_bias(p, {d.off_mlp[1]});
        p.bias_mlp[1] = add_bias(p, {d.off_mlp[3]});
    }
    p.pack_enc_base.assign(NT, -1); p.bias_enc.assign(NT, -1);
    for (int t = 0; t < NT; ++t) {
        if (!p.need_dx[0][t]) continue;      // no node of this type feeds the output at this depth: its encoder is never run
        const int F = d.type_width[t];
        for (int kc = 0; kc < p.enc_nkc[t]; ++kc) {
            const int id = add_pack(p, 0, {p.off_enc_w[t]}, F, kc * H, std::min(H, F - kc * H));
            if (kc == 0) p.pack_enc_base[t] = id;
        }
        p.bias_enc[t] = add_bias(p, {p.off_enc_b[t]});
    }

    // ---- programs -----------------------------------------------------------------------------------
    std::vector<int32_t>& T = p.tables;
    T.clear();
    double exec_fwd = 0, exec_bwd = 0, alg_fwd = 0, alg_bwd = 0;
    std::vector<double> lf_alg(L, 0.0), lf_exec(L, 0.0), lb_alg(L, 0.0), lb_exec(L, 0.0);
    double gw_alg = 0, gw_exec = 0;
    const double NL = 2.0 * H * H;   // FLOPs of one node-linear (one window)
    p.mlp_scratch = has_mlp && (int64_t)(p.NN + p.n_mlp) * p.blk_bytes <= LDS_LIMIT;
    p.n_blk = p.NN + (p.mlp_scratch ? p.n_mlp : 0);

    struct Seg { int pack; std::vector<std::pair<int, int>> macs; };   // (slot, src block)
    bool prog_overflow = false;
    std::vector<int> wprog[2];   // the layer's wave programs under construction (half 0 / half 1)
    auto emit_segments = [&](const std::vector<Seg>& segs, int gh) {
        const int hs = p.gmax / 2;
        for (int half = 0; half < 2; ++half) {
            std::vector<int>& w = wprog[half];
            T[gh + GH_PC0 + half] = (int)w.size();
            w.push_back((int)segs.size());
            for (const Seg& sg : segs) {
                w.push_back(sg.pack);
                for (int u = 0; u < hs; ++u) {
                    const int slot = 2 * u + half;
                    const size_t pos = w.size(); w.push_back(0);
                    for (auto& m : sg.macs) if (m.first == slot) { w.push_back(m.second); w[pos]++; }
                }
            }
        }
    };
    auto flush_wprog = [&]() {
        for (int half = 0; half < 2; ++half) {
            if ((int)wprog[half].size() > WPROG_LEN) prog_overflow = true;
            wprog[half].resize(WPROG_LEN, 0);
            for (int v : wprog[half]) T.push_back(v);
            wprog[half].clear();
        }
    };
    struct GroupDef { int type, c0, ns; bool mlp; std::vector<int> idx; };      // idx: the group's nodes as indices inside their type (live nodes only)
    auto make_groups = [&](int t, const bool (&sel)[64], bool mlp, std::vector<GroupDef>& out) {
        std::vector<int> nodes;
        for (int i = 0; i < d.type_nodes[t]; ++i) if (sel[p.type_base[t] + i]) nodes.push_back(i);
        for (size_t c0 = 0; c0 < nodes.size(); c0 += p.gmax) {
            GroupDef g{t, (int)c0, (int)std::min<size_t>(p.gmax, nodes.size() - c0), mlp, {}};
            g.idx.assign(nodes.begin() + c0, nodes.begin() + c0 + g.ns);
            out.push_back(g);
        }
    };

    for (int l = 0; l < L; ++l) {
        const double ef0 = exec_fwd, af0 = alg_fwd, eb0 = exec_bwd, ab0 = alg_bwd;
        // ---------- forward program ----------
        // group order: relu groups by size (largest last: its epilogue goes through LDS with coalesced stores);
        // the base_transform group second-to-last when it has scratch blocks, else last (it overwrites its own blocks)
        std::vector<GroupDef> relu_g, mlp_g;
        for (int t = 0; t < NT; ++t) {
            if (!p.live[l][t]) continue;
            const bool mlp = has_mlp && t == d.mlp_type;
            make_groups(t, p.live_n[l], mlp, mlp ? mlp_g : relu_g);
        }
        std::stable_sort(relu_g.begin(), relu_g.end(), [](const GroupDef& a, const GroupDef& b) { return a.ns < b.ns; });
        std::vector<GroupDef> fgroups;
        if (p.mlp_scratch || mlp_g.empty()) {
            for (size_t i = 0; i + 1 < relu_g.size(); ++i) fgroups.push_back(relu_g[i]);
            for (auto& g : mlp_g) fgroups.push_back(g);
            if (!relu_g.empty()) fgroups.push_back(relu_g.back());
        } else {
            fgroups = relu_g; for (auto& g : mlp_g) fgroups.push_back(g);
        }
        p.fwd_prog_off[l] = (int)T.size();
        T.push_back((int)fgroups.size());
        for (size_t gi = 0; gi < fgroups.size(); ++gi) {
            const GroupDef& G = fgroups[gi];
            const int t = G.type, ns = G.ns;
            auto slot_of = [&](int i) { for (int u = 0; u < ns; ++u) if (G.idx[u] == i) return u; return -1; };
            const int gh = (int)T.size(); T.resize(T.size() + GH_SIZE, 0);
            T[gh + GH_KIND] = G.mlp ? KIND_MLP : KIND_RELU; T[gh + GH_NSLOTS] = ns;
            T[gh + GH_BIAS] = p.bias_layer[l * NT + t];
            const bool lds_epi = !G.mlp && gi + 1 == fgroups.size();
            T[gh + GH_FLAGS] = (residual ? GF_RESIDUAL : 0) | GF_STORE_MASK | (lds_epi ? GF_LDS_EPI : 0);
            if (G.mlp) { T[gh + GH_W1] = p.pack_mlp[0][0]; T[gh + GH_W2] = p.pack_mlp[0][1]; T[gh + GH_B1] = p.bias_mlp[0]; T[gh + GH_B2] = p.bias_mlp[1]; }
            for (int u = 0; u < ns; ++u) {
                T[gh + GH_NODES + u] = p.type_base[t] + G.idx[u]; T[gh + GH_MLPIDX + u] = G.idx[u];
                T[gh + GH_SCR + u] = (G.mlp && p.mlp_scratch) ? p.NN + G.idx[u] : p.type_base[t] + G.idx[u];
            }
            std::vector<Seg> segs;
            Seg root; root.pack = p.pack_root[0][l * NT + t];
            for (int u = 0; u < ns; ++u) { root.macs.push_back({u, p.type_base[t] + G.idx[u]}); exec_fwd += NL; alg_fwd += NL; }
            segs.push_back(root);
            for (int r = 0; r < NR; ++r) {
                if (p.rel_dst[r] != t || p.pack_rel[0][l * NR + r] < 0) continue;
                Seg sg; sg.pack = p.pack_rel[0][l * NR + r];
                std::vector<bool> hit(ns, false);
                for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) {
                    const int j = p.edges[2 * e], u = slot_of(p.edges[2 * e + 1]);
                    if (u < 0) continue;
                    sg.macs.push_back({u, p.type_base[p.rel_src[r]] + j}); exec_fwd += NL;
                    if (!hit[u]) { hit[u] = true; alg_fwd += NL; }
                }
                if (!sg.macs.empty()) segs.push_back(sg);
            }
            if (G.mlp) { exec_fwd += 2 * NL * ns; alg_fwd += 2 * NL * ns; }
            T[gh + GH_NSEG] = (int)segs.size();
            emit_segments(segs, gh);
        }
        flush_wprog();

        // ---------- backward program ----------
        p.bwd_prog_off[l] = (int)T.size();
        const int bh = (int)T.size(); T.resize(T.size() + BH_SIZE, 0);
        const bool mlp_live = has_mlp && p.live[l][d.mlp_type];
        T[bh + BH_NMLP] = mlp_live ? p.n_mlp : 0;
        if (mlp_live) { T[bh + BH_W2] = p.pack_mlp[1][1]; T[bh + BH_W1] = p.pack_mlp[1][0]; }
        for (int n = 0; n < p.NN; ++n) {
            const int t = p.node_type[n];
            T[bh + BH_KIND + n] = !p.live_n[l][n] ? NK_DEAD : ((has_mlp && t == d.mlp_type) ? NK_MLP : NK_RELU);
        }
        if (mlp_live) { for (int u = 0; u < p.n_mlp; ++u) T[bh + BH_MLPNODES + u] = p.type_base[d.mlp_type] + u; exec_bwd += 2 * NL * p.n_mlp; alg_bwd += 2 * NL * p.n_mlp; }
        std::vector<GroupDef> bgroups;
        // (a group is homogeneous in "live in this layer": the residual term dX_{l+1}[n] exists for live nodes only, and the kernel's flag is per group)
        bool need_live[64], need_dead[64];
        for (int n = 0; n < p.NN; ++n) { need_live[n] = p.need_n[l][n] && p.live_n[l][n]; need_dead[n] = p.need_n[l][n] && !p.live_n[l][n]; }
        for (int t = 0; t < NT; ++t) {
            if (!p.need_dx[l][t]) continue;
            make_groups(t, need_live, true, bgroups);       // (mlp field reused: the group's nodes are live in this layer)
            make_groups(t, need_dead, false, bgroups);
        }
        std::stable_sort(bgroups.begin(), bgroups.end(), [](const GroupDef& a, const GroupDef& b) { return a.ns < b.ns; });
        for (size_t gi = 0; gi < bgroups.size(); ++gi) {
            const GroupDef& G = bgroups[gi];
            const int t = G.type, ns = G.ns;
            auto slot_of = [&](int i) { for (int u = 0; u < ns; ++u) if (G.idx[u] == i) return u; return -1; };
            const int gh = (int)T.size(); T.resize(T.size() + GH_SIZE, 0);
            T[gh + GH_KIND] = KIND_RELU; T[gh + GH_NSLOTS] = ns; T[gh + GH_BIAS] = -1;
            // at layer 0 the kernel finishes dY_enc = relu'(X_0) . (G_1 + D_0) itself: GF_RESIDUAL = add G_1
            T[gh + GH_FLAGS] = ((residual && G.mlp) ? GF_RESIDUAL : 0) | (l == 0 ? GF_ENC_MASK : 0) |
                               (gi + 1 == bgroups.size() ? GF_LDS_EPI : 0);
            for (int u = 0; u < ns; ++u) T[gh + GH_NODES + u] = p.type_base[t] + G.idx[u];
            std::vector<Seg> segs;
            if (G.mlp) {      // dX_l[n] += dH_l[n] W_rootsum: the group's nodes are live themselves
                Seg root; root.pack = p.pack_root[1][l * NT + t];
                for (int u = 0; u < ns; ++u) { root.macs.push_back({u, p.type_base[t] + G.idx[u]}); exec_bwd += NL; alg_bwd += NL; }
                segs.push_back(root);
            }
            for (int r = 0; r < NR; ++r) {
                if (p.rel_src[r] != t || p.pack_rel[1][l * NR + r] < 0) continue;
                Seg sg; sg.pack = p.pack_rel[1][l * NR + r];
                for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) {
                    const int u = slot_of(p.edges[2 * e]), i = p.edges[2 * e + 1];
                    if (u < 0 || !p.live_n[l][p.type_base[p.rel_dst[r]] + i]) continue;
                    sg.macs.push_back({u, p.type_base[p.rel_dst[r]] + i}); exec_bwd += NL;
                }
                if (!sg.macs.empty()) segs.push_back(sg);
            }
            T[gh + GH_NSEG] = (int)segs.size();
            emit_segments(segs, gh);
        }
        flush_wprog();
        T[bh + BH_NGROUPS] = (int)bgroups.size();
        lf_alg[l] = alg_fwd - af0; lf_exec[l] = exec_fwd - ef0; lb_alg[l] = alg_bwd - ab0; lb_exec[l] = exec_bwd - eb0;
    }
    if (prog_overflow) return fail(p, "a layer's MAC program exceeds 128 entries per wave (too many relations/edges for this build)");

    // ---- fused stack programs (bf16 plan) ---------------------------------------------------------------
    // split plan: where the doubled tile + base_transform scratch does not fit (MiniCheetah-K4: 2 x 24 blocks) but the doubled tile alone does
    // (2 x 20 = 160 KB), the scratch blocks ALIAS the last n_mlp nodes' blocks: their owners read those nodes' residual octets before the chain
    // (x3_alias; needs an even first victim so that scratch block i and victim i belong to the same wave, and victims that are not
    // base_transform nodes themselves)
    p.x3_alias = p.split && has_mlp && (int64_t)p.planes * (p.NN + p.n_mlp) * p.blk_bytes > LDS_LIMIT && (int64_t)p.planes * p.NN * p.blk_bytes <= LDS_LIMIT &&
                 p.NN >= 2 * p.n_mlp && (p.NN - p.n_mlp) % 2 == 0;
    p.fs_blk = p.NN + (p.x3_alias ? 0 : p.n_mlp);
    p.fused = d.dtype != MSHGNN_F32 && p.NN <= FS_MAXN && (int64_t)p.planes * p.fs_blk * p.blk_bytes <= LDS_LIMIT &&
              (!has_mlp || (p.type_base[d.mlp_type] == 0 && p.n_mlp <= 4));   // base_transform nodes are accumulators 0..1 of each wave half
    p.lo_blk = p.fs_blk; p.n_img = (int)p.packs.size();
    if (p.split && !p.fused)
        return fail(p, "the split-bf16 parity plan is not supported for this topology (its LDS-resident tile holds 2 x (nodes + base_transform "
                       "nodes) <= 40 blocks and <= 20 nodes); use MSHGNN_F32");
    // split plan: X W = X_hi W_hi + X_lo W_hi + X_hi W_lo (the lo x lo term is below fp32 resolution).  The MAC loop of the stack kernels
    // is unchanged: every segment becomes two -- the hi image with the hi and the lo block of each source, the lo image with the hi block
    auto split_segs = [&](const std::vector<Seg>& in) {
        if (!p.split) return in;
        std::vector<Seg> out;
        for (const Seg& sg : in) {
            // at most 3 source blocks per accumulator and segment (the hi-image segment holds twice as many MACs, 3 count bits)
            std::vector<std::pair<int, int>> rest = sg.macs;
            while (!rest.empty()) {
                std::vector<std::pair<int, int>> take, keep; std::vector<int> cnt(64, 0);
                for (auto& m : rest) { if (cnt[m.first] < 3) { take.push_back(m); cnt[m.first]++; } else keep.push_back(m); }
                Seg a; a.pack = sg.pack; Seg b; b.pack = sg.pack + p.n_img;
                for (auto& m : take) { a.macs.push_back(m); a.macs.push_back({m.first, m.second + p.lo_blk}); b.macs.push_back(m); }
                out.push_back(a); out.push_back(b);
                rest.swap(keep);
            }
        }
        return out; };
    if (p.fused) {
        auto emit_fused = [&](const std::vector<Seg>& segs_in) {   // Seg.macs = (node, source block)
            const std::vector<Seg> segs = split_segs(segs_in);
            for (int half = 0; half < 2 && p.fused; ++half) {
                std::vector<int> w, blocks, counts;    // byte entries; packed per-segment counts
                w.push_back((int)segs.size());
                bool ok = segs.size() <= 64;
                for (const Seg& sg : segs) {
                    int cw = 0;
                    for (int u = 0; u < FS_HS; ++u) {
                        int c = 0;
                        for (auto& m : sg.macs) if (m.first == 2 * u + half) { blocks.push_back(m.second); ++c; }
                        if (c > 7) ok = false;
                        cw |= (c & 7) << (3 * u);
                    }
                    counts.push_back(cw);
                }
                blocks.push_back(blocks.empty() ? 0 : blocks.back());
                for (int b : blocks) w.push_back(b);
                if (!ok || w.size() > 256) { p.fused = false; break; }
                w.resize(256, 0); counts.resize(64, 0);
                for (int sgi = 0; sgi < 64; ++sgi) T.push_back(sgi < (int)segs.size() ? segs[sgi].pack : 0);
                for (int sgi = 0; sgi < 64; ++sgi) T.push_back(counts[sgi]);
                for (int i = 0; i < 64; ++i) T.push_back(w[4 * i] | (w[4 * i + 1] << 8) | (w[4 * i + 2] << 16) | (w[4 * i + 3] << 24));
            }
        };
        // slab programs: group A = the largest node type, group B = every other node in node order.  With node-level liveness a group may have nothing
        // to do in a layer (its pass is then skipped: FF_A_EMPTY / FF_B_EMPTY), and a pass costs ~5 k cycles of fixed latency whatever it holds -- so when
        // the nodes differ in how often they are computed (A1-C2 at 3 layers: thighs, knees and feet in every pass, the hips only in the backward's last
        // one, the base nodes never), group A takes the 12 busiest nodes instead: 7 group passes per tile instead of 11.  The base_transform nodes stay
        // the first slots of group B; a node with more than 3 in- or out-edges in one relation needs group B's 3-bit MAC counts.  Slots are the same in
        // every layer (the backward carries a slot's residual row in registers from layer to layer).
        int tA = 0; for (int t = 1; t < NT; ++t) if (d.type_nodes[t] > d.type_nodes[tA]) tA = t;
        std::vector<int> slotA, slotB;
        for (int n = 0; n < p.NN; ++n) (p.node_type[n] == tA ? slotA : slotB).push_back(n);
        p.sl_ta = tA;
        if (p.prune_nodes && !p.split) {
            std::vector<int> busy(p.NN, 0), deg(p.NN, 0);
            for (int l = 0; l < L; ++l) for (int n = 0; n < p.NN; ++n) busy[n] += (p.live_n[l][n] ? 1 : 0) + (p.need_n[l][n] ? 1 : 0);
            for (int r = 0; r < NR; ++r) {
                std::vector<int> din(d.type_nodes[p.rel_dst[r]], 0), dout(d.type_nodes[p.rel_src[r]], 0);
                for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) { ++dout[p.edges[2 * e]]; ++din[p.edges[2 * e + 1]]; }
                for (size_t i = 0; i < din.size(); ++i) deg[p.type_base[p.rel_dst[r]] + i] = std::max(deg[p.type_base[p.rel_dst[r]] + i], din[i]);
                for (size_t j = 0; j < dout.size(); ++j) deg[p.type_base[p.rel_src[r]] + j] = std::max(deg[p.type_base[p.rel_src[r]] + j], dout[j]);
            }
            auto passes = [&](const std::vector<int>& A, const std::vector<int>& Bv) {      // group passes with work, over all layers and both sweeps
                int c = 0;
                for (int l = 0; l < L; ++l)
                    for (int dir = 0; dir < 2; ++dir)
                        for (const std::vector<int>* g : {&A, &Bv}) {
                            bool any = false;
                            for (int n : *g) any = any || (dir == 0 ? p.live_n[l][n] : p.need_n[l][n]);
                            c += any ? 1 : 0;
                        }
                return c; };
            std::vector<int> cand, A2, B2;
            for (int n = 0; n < p.NN; ++n) {
                if (has_mlp && p.node_type[n] == d.mlp_type) B2.push_back(n);      // (first slots of group B, in node order)
                else cand.push_back(n);
            }
            std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) { return busy[x] > busy[y]; });
            for (int n : cand) { if ((int)A2.size() < SL_HA && deg[n] <= 3 && busy[n] > 0) A2.push_back(n); else B2.push_back(n); }
            std::sort(A2.begin(), A2.end());
            std::sort(B2.begin() + (has_mlp ? p.n_mlp : 0), B2.end());
            if ((int)B2.size() <= SL_HB_MAX && passes(A2, B2) < passes(slotA, slotB)) { slotA = A2; slotB = B2; p.sl_ta = -1; }
        }
        p.sl_hb = (int)slotB.size() <= SL_HB ? SL_HB : SL_HB_MAX;
        p.sl_alias = 2 * (int64_t)p.fs_blk * p.blk_bytes > LDS_LIMIT && p.n_mlp <= (int)slotA.size();
        p.sl_blk = p.sl_alias ? p.NN : p.fs_blk;
        p.slab = !p.split && 2 * (int64_t)p.sl_blk * p.blk_bytes <= LDS_LIMIT && (int)slotA.size() <= SL_HA && (int)slotB.size() <= SL_HB_MAX &&
                 (!has_mlp || ((p.sl_ta < 0 || d.mlp_type != tA) && p.n_mlp <= (int)slotB.size()));     // base_transform nodes = the first slots of group B
        if (has_mlp) for (int u = 0; u < p.n_mlp && p.slab; ++u) if (slotB[u] != p.type_base[d.mlp_type] + u) p.slab = false;
        auto emit_slab = [&](const std::vector<Seg>& segs, int hdr_src) {
            const int h = (int)T.size(); T.resize(T.size() + FH_SIZE, 0);
            for (int i = 0; i < FH_SLOTA; ++i) T[h + i] = T[hdr_src + i];
            for (int u = 0; u < 16; ++u) { T[h + FH_SLOTA + u] = u < (int)slotA.size() ? slotA[u] : -1; T[h + FH_SLOTB + u] = u < (int)slotB.size() ? slotB[u] : -1; }
            // per-node arrays re-indexed by slot q (group A: q = u, group B: q = SL_HA + u) so that the kernel reads them with constant lanes
            static_assert(SL_HA + SL_HB_MAX <= FS_MAXN, "slot arrays share the per-node header arrays");
            if (p.sl_alias) T[h + FH_FLAGS] |= FF_SCR_ALIAS;
            for (int q = 0; q < FS_MAXN; ++q) {
                const int n = q < SL_HA ? (q < (int)slotA.size() ? slotA[q] : -1) : (q - SL_HA < (int)slotB.size() ? slotB[q - SL_HA] : -1);
                for (int arr : {FH_KIND, FH_BIAS, FH_OUT, FH_RES}) T[h + arr + q] = n >= 0 ? T[hdr_src + arr + n] : 0;
            }
            for (int phase = 0; phase < 2 && p.slab; ++phase) {
                const std::vector<int>& slots = phase == 0 ? slotA : slotB;
                std::vector<int> w, blocks, counts, packs;
                bool ok = true;
                for (const Seg& sg : segs) {      // a segment's MACs go to the group that holds their destination slot (a segment may serve both groups)
                    int cw = 0, total = 0;
                    std::vector<int> blk;
                    for (size_t u = 0; u < slots.size(); ++u) {
                        int c = 0;
                        for (auto& m : sg.macs) if (m.first == slots[u]) { blk.push_back(m.second); ++c; }
                        const int cb = phase == 0 ? SL_CBA : SL_CBB;     // bits of MAC count per slot
                        if (c >= (1 << cb)) ok = false;
                        cw |= c << (cb * (int)u);
                        total += c;
                    }
                    if (total == 0) continue;
                    blocks.insert(blocks.end(), blk.begin(), blk.end());
                    counts.push_back(cw); packs.push_back(sg.pack);
                }
                {   // nothing to do for this group in this layer: no live / produced slot (the forward's base_transform chain counts as work)
                    bool any = false;
                    for (size_t u = 0; u < slots.size(); ++u) {
                        const int q = (phase == 0 ? 0 : SL_HA) + (int)u;
                        any = any || T[h + FH_KIND + q] != NK_DEAD || T[h + FH_OUT + q] != 0;
                    }
                    if (!any && packs.empty()) T[h + FH_FLAGS] |= (phase == 0 ? FF_A_EMPTY : FF_B_EMPTY);
                }
                w.push_back((int)packs.size());
                blocks.push_back(blocks.empty() ? 0 : blocks.back());
                for (int b : blocks) w.push_back(b);
                if (!ok || w.size() > 256 || packs.size() > 64) { p.slab = false; break; }
                w.resize(256, 0); counts.resize(64, 0); packs.resize(64, 0);
                for (int i = 0; i < 64; ++i) T.push_back(packs[i]);
                for (int i = 0; i < 64; ++i) T.push_back(counts[i]);
                for (int i = 0; i < 64; ++i) T.push_back(w[4 * i] | (w[4 * i + 1] << 8) | (w[4 * i + 2] << 16) | (w[4 * i + 3] << 24));
            }
            return h;
        };
        for (int l = 0; l < L && p.fused; ++l) {
            const bool mlp_live = has_mlp && p.live[l][d.mlp_type];
            // forward
            {
                const int fh = (int)T.size(); T.resize(T.size() + FH_SIZE, 0);
                p.fs_fwd_off[l] = fh;
                T[fh + FH_NMLP] = mlp_live ? p.n_mlp : 0; T[fh + FH_FLAGS] = residual ? FF_RESIDUAL : 0;
                T[fh + FH_MLP0] = has_mlp ? p.type_base[d.mlp_type] : 0;
                if (mlp_live) { T[fh + FH_W1] = p.pack_mlp[0][0]; T[fh + FH_W2] = p.pack_mlp[0][1]; T[fh + FH_B1] = p.bias_mlp[0]; T[fh + FH_B2] = p.bias_mlp[1]; }
                for (int n = 0; n < p.NN; ++n) {
                    const int t = p.node_type[n];
                    T[fh + FH_KIND + n] = !p.live_n[l][n] ? NK_DEAD : ((has_mlp && t == d.mlp_type) ? NK_MLP : NK_RELU);
                    T[fh + FH_BIAS + n] = p.live_n[l][n] ? p.bias_layer[l * NT + t] : 0;
                }
                std::vector<Seg> segs;
                for (int t = 0; t < NT; ++t) {
                    if (!p.live[l][t]) continue;
                    Seg root; root.pack = p.pack_root[0][l * NT + t];
                    for (int i = 0; i < d.type_nodes[t]; ++i) if (p.live_n[l][p.type_base[t] + i]) root.macs.push_back({p.type_base[t] + i, p.type_base[t] + i});
                    segs.push_back(root);
                    for (int r = 0; r < NR; ++r) {
                        if (p.rel_dst[r] != t || p.pack_rel[0][l * NR + r] < 0) continue;
                        Seg sg; sg.pack = p.pack_rel[0][l * NR + r];
                        for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e)
                            if (p.live_n[l][p.type_base[t] + p.edges[2 * e + 1]])
                                sg.macs.push_back({p.type_base[t] + p.edges[2 * e + 1], p.type_base[p.rel_src[r]] + p.edges[2 * e]});
                        if (!sg.macs.empty()) segs.push_back(sg);
                    }
                }
                T[fh + FH_NSEG] = (int)segs.size();
                emit_fused(segs);
                if (p.fused && p.slab) p.sl_fwd_off[l] = emit_slab(segs, fh);
            }
            if (!p.fused) break;
            // backward
            {
                const int bh2 = (int)T.size(); T.resize(T.size() + FH_SIZE, 0);
                p.fs_bwd_off[l] = bh2;
                T[bh2 + FH_NMLP] = mlp_live ? p.n_mlp : 0;
                T[bh2 + FH_FLAGS] = (residual ? FF_RESIDUAL : 0) | (l == 0 ? FF_ENC_MASK : 0);
                T[bh2 + FH_MLP0] = has_mlp ? p.type_base[d.mlp_type] : 0;
                if (mlp_live) { T[bh2 + FH_W2] = p.pack_mlp[1][1]; T[bh2 + FH_W1] = p.pack_mlp[1][0]; }
                for (int n = 0; n < p.NN; ++n) {
                    const int t = p.node_type[n];
                    T[bh2 + FH_KIND + n] = !p.live_n[l][n] ? NK_DEAD : ((has_mlp && t == d.mlp_type) ? NK_MLP : NK_RELU);
                    T[bh2 + FH_OUT + n] = p.need_n[l][n] ? 1 : 0;
                    T[bh2 + FH_RES + n] = (residual && p.live_n[l][n] && p.need_n[l][n]) ? 1 : 0;
                }
                std::vector<Seg> segs;
                for (int t = 0; t < NT; ++t) {
                    if (!p.need_dx[l][t]) continue;
                    if (p.live[l][t]) {
                        Seg root; root.pack = p.pack_root[1][l * NT + t];
                        for (int i = 0; i < d.type_nodes[t]; ++i)
                            if (p.live_n[l][p.type_base[t] + i] && p.need_n[l][p.type_base[t] + i]) root.macs.push_back({p.type_base[t] + i, p.type_base[t] + i});
                        if (!root.macs.empty()) segs.push_back(root);
                    }
                    for (int r = 0; r < NR; ++r) {
                        if (p.rel_src[r] != t || p.pack_rel[1][l * NR + r] < 0) continue;
                        Seg sg; sg.pack = p.pack_rel[1][l * NR + r];
                        for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e)
                            if (p.live_n[l][p.type_base[p.rel_dst[r]] + p.edges[2 * e + 1]])      // (its source is needed by definition)
                                sg.macs.push_back({p.type_base[t] + p.edges[2 * e], p.type_base[p.rel_dst[r]] + p.edges[2 * e + 1]});
                        if (!sg.macs.empty()) segs.push_back(sg);
                    }
                }
                T[bh2 + FH_NSEG] = (int)segs.size();
                emit_fused(segs);
                if (p.fused && p.slab) p.sl_bwd_off[l] = emit_slab(segs, bh2);
            }
        }
        if (!p.fused) p.slab = false;
    }
    if (p.split && !p.fused) return fail(p, "the split-bf16 parity plan is not supported for this topology (a layer's MAC program exceeds the "
                                            "stack kernels' program registers); use MSHGNN_F32");
    // algorithmic dX work: one node-linear per (relation, src node with >=1 out-edge into a live dst) -- count below
    for (int l = 0; l < L; ++l)
        for (int r = 0; r < NR; ++r) {
            if (p.pack_rel[0][l * NR + r] < 0) continue;
            std::vector<bool> hit(d.type_nodes[p.rel_src[r]], false);
            for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) if (p.live_n[l][p.type_base[p.rel_dst[r]] + p.edges[2 * e + 1]]) hit[p.edges[2 * e]] = true;
            for (bool b : hit) if (b) { alg_bwd += NL; lb_alg[l] += NL; }
        }
    const double gw_e0 = exec_bwd, gw_a0 = alg_bwd;

    // ---- weight-gradient targets / items / finalize ops -----------------------------------------------
    struct Tgt { std::vector<int> items; int bias_flag; };
    std::vector<std::vector<int32_t>> items;   // each ITEM_INTS
    std::vector<int> item_cluster;             // L2-sharing cluster of each item (same destination-node quarter / layer)
    std::vector<Tgt> tgts;
    auto add_item = [&](int pb, int ps, int po, int qb, int qs, int qo, int qc0, int qn, int so) {
        items.push_back({pb, ps, po, qb, qs, qo, qc0, qn, so, -1});
        // cluster key: which rows the item's P operand streams (buffer, node quarter): items with the same P share an XCD
        const int pnode = po, ptype_nodes = std::max(1, ps / H);
        item_cluster.push_back(pb * 4 + (pnode * 4) / ptype_nodes % 4);
        return (int)items.size() - 1; };
    const int SN = p.NN * H, SM = std::max(1, p.n_mlp) * H;
    std::vector<int> tgt_root((size_t)L * NT, -1), tgt_rel((size_t)L * NR, -1);
    int tgt_mlp[2] = {-1, -1};
    for (int l = 0; l < L; ++l) {
        for (int t = 0; t < NT; ++t) {
            if (!p.live[l][t]) continue;
            Tgt g; g.bias_flag = 1;
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                const int n = p.type_base[t] + i;
                if (!p.live_n[l][n]) continue;      // dH of a dead node is zero: no contribution
                g.items.push_back(add_item(BUF_DH + l, SN, n, BUF_X + l, SN, n, 0, H, -1));
                if (!(has_mlp && t == d.mlp_type)) { items.back()[0] = BUF_DX + l + 1; items.back()[9] = BUF_MASK + l; }
                exec_bwd += NL; alg_bwd += NL;
            }
            tgt_root[l * NT + t] = (int)tgts.size(); tgts.push_back(g);
        }
        for (int r = 0; r < NR; ++r) {
            if (p.pack_rel[0][l * NR + r] < 0) continue;
            Tgt g; g.bias_flag = 0;
            std::vector<bool> hit(d.type_nodes[p.rel_dst[r]], false);
            for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) {
                const int j = p.edges[2 * e], i = p.edges[2 * e + 1];
                if (!p.live_n[l][p.type_base[p.rel_dst[r]] + i]) continue;
                g.items.push_back(add_item(BUF_DH + l, SN, p.type_base[p.rel_dst[r]] + i,
                                           BUF_X + l, SN, p.type_base[p.rel_src[r]] + j, 0, H, -1));
                if (!(has_mlp && p.rel_dst[r] == d.mlp_type)) { items.back()[0] = BUF_DX + l + 1; items.back()[9] = BUF_MASK + l; }
                exec_bwd += NL; if (!hit[i]) { hit[i] = true; alg_bwd += NL; }
            }
            tgt_rel[l * NR + r] = (int)tgts.size(); tgts.push_back(g);
        }
    }
    if (has_mlp) {
        Tgt g1, g2; g1.bias_flag = g2.bias_flag = 1;
        for (int l = 0; l < L; ++l) {
            if (!p.live[l][d.mlp_type]) continue;
            for (int u = 0; u < p.n_mlp; ++u) {
                g1.items.push_back(add_item(BUF_DU + l, SM, u, BUF_HB + l, SM, u, 0, H, -1));
                g2.items.push_back(add_item(BUF_DX + l + 1, SN, p.type_base[d.mlp_type] + u, BUF_T1 + l, SM, u, 0, H, -1));
                exec_bwd += 2 * NL; alg_bwd += 2 * NL;
            }
        }
        if (!g1.items.empty()) { tgt_mlp[0] = (int)tgts.size(); tgts.push_back(g1); tgt_mlp[1] = (int)tgts.size(); tgts.push_back(g2); }
    }
    std::vector<std::vector<int>> tgt_enc(NT);
    for (int t = 0; t < NT; ++t) {
        if (!p.need_dx[0][t]) continue;
        const int F = d.type_width[t];
        for (int kc = 0; kc < p.enc_nkc[t]; ++kc) {
            Tgt g; g.bias_flag = (kc == 0);
            const int nc = std::min(H, F - kc * H);
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                if (!p.need_n[0][p.type_base[t] + i]) continue;      // the encoder does not compute this node: its rows feed nothing
                g.items.push_back(add_item(BUF_DX + 0, SN, p.type_base[t] + i, BUF_IN + t, -1, i, kc * H, nc,
                                           p.sign_off[t] + i * p.enc_nkc[t] * H + kc * H));
                exec_bwd += NL; alg_bwd += 2.0 * H * nc;
            }
            tgt_enc[t].push_back((int)tgts.size()); tgts.push_back(g);
        }
    }
    gw_alg = alg_bwd - gw_a0; gw_exec = exec_bwd - gw_e0;
    double enc_alg = 0, enc_exec = 0;
    std::vector<int> enc_n(NT, 0);      // nodes of each type the encoder computes
    for (int n = 0; n < p.NN; ++n) if (p.need_n[0][n]) ++enc_n[p.node_type[n]];
    for (int t = 0; t < NT; ++t) { enc_exec += (double)enc_n[t] * p.enc_nkc[t] * NL; enc_alg += (double)enc_n[t] * 2.0 * H * d.type_width[t]; }
    for (int t = 0; t < NT; ++t) {   // encoder forward work
        exec_fwd += (double)enc_n[t] * p.enc_nkc[t] * NL;
        alg_fwd += (double)enc_n[t] * 2.0 * H * d.type_width[t];
    }
    alg_fwd += 2.0 * n_out * d.out_channels * H; exec_fwd += 2.0 * n_out * d.out_channels * H;
    alg_bwd += 4.0 * n_out * d.out_channels * H; exec_bwd += 4.0 * n_out * d.out_channels * H;

    // work split of the split-K weight-gradient kernel: every target's items are cut into lanes of <= GW_IPL
    // items; workgroup (lane, part) runs its lane's items over window part `part`.  All workgroups do the same
    // work per window chunk, so they sweep the batch at the same pace and the P/Q rows that several items share
    // are re-read from L2 / Infinity Cache instead of HBM.
    p.item_off = (int)T.size(); p.n_items = 0;
    {
        double best = -1;
        for (int ipl = 1; ipl <= GW_IPL; ++ipl) {
            int nl = 0; for (auto& tg : tgts) nl += ((int)tg.items.size() + ipl - 1) / ipl;
            const double fill = (double)nl * std::max(1, std::min(16, p.gw_target / std::max(1, nl)));
            if (fill > best) { best = fill; p.gw_ipl = ipl; }
        }
    }
    std::vector<std::array<int, 4>> lanes;
    std::vector<int> tgt_lane_begin(tgts.size()), tgt_lane_end(tgts.size());
    for (size_t g = 0; g < tgts.size(); ++g) {
        tgt_lane_begin[g] = (int)lanes.size();
        const int first = p.n_items;
        for (int it : tgts[g].items) { for (int k = 0; k < ITEM_INTS; ++k) T.push_back(items[it][k]); ++p.n_items; }
        for (int i0 = first; i0 < p.n_items; i0 += p.gw_ipl)
            lanes.push_back({i0, std::min(p.n_items, i0 + p.gw_ipl), (int)g, tgts[g].bias_flag});
        tgt_lane_end[g] = (int)lanes.size();
    }
    p.n_lanes = (int)lanes.size();
    p.n_parts = std::max(1, std::min(16, p.gw_target / std::max(1, p.n_lanes)));    // never more workgroups than are resident at once
    // XCD placement: cluster -> least-loaded XCD queue (largest clusters first); block b runs queue[b % 8][b / 8]
    std::vector<int> lane_cluster(p.n_lanes), flat_item_cluster;
    for (size_t g = 0; g < tgts.size(); ++g) for (int it : tgts[g].items) flat_item_cluster.push_back(item_cluster[it]);
    for (int ln = 0; ln < p.n_lanes; ++ln) lane_cluster[ln] = flat_item_cluster[lanes[ln][0]];
    std::vector<int> cl_ids = lane_cluster; std::sort(cl_ids.begin(), cl_ids.end()); cl_ids.erase(std::unique(cl_ids.begin(), cl_ids.end()), cl_ids.end());
    std::vector<std::pair<int, int>> cl_size;   // (-size, id)
    for (int c : cl_ids) cl_size.push_back({-(int)std::count(lane_cluster.begin(), lane_cluster.end(), c), c});
    std::sort(cl_size.begin(), cl_size.end());
    // Operand streams of a lane: (buffer, node[, input column chunk]) of its P and Q rows, with their bytes per window.  Lanes on one XCD
    // that share a stream fetch it through that XCD's L2 once, so the placement minimises  sum over XCDs of the distinct stream bytes
    // (tools/gradw_sharing.py prints that sum next to the unique bytes) by local search from the cluster placement: single-lane moves and
    // pair swaps, queue lengths kept within GW_XCD_SLACK of the balanced length.  Deterministic (no randomness).
    std::vector<std::vector<std::pair<int, int>>> lane_streams(p.n_lanes);    // (stream id, bytes)
    int n_streams = 0;
    {
        std::vector<std::array<int, 4>> keys;
        auto sid = [&](std::array<int, 4> k) { for (size_t i = 0; i < keys.size(); ++i) if (keys[i] == k) return (int)i; keys.push_back(k); return (int)keys.size() - 1; };
        for (int ln = 0; ln < p.n_lanes; ++ln)
            for (int it = lanes[ln][0]; it < lanes[ln][1]; ++it) {
                const int32_t* im = &T[p.item_off + (size_t)it * ITEM_INTS];
                const int es = p.esize * p.planes;
                lane_streams[ln].push_back({sid({0, im[0], im[2], 0}), H * es + (im[9] >= 0 ? 16 : 0)});
                if (im[4] >= 0) lane_streams[ln].push_back({sid({1, im[3], im[5], 0}), H * es});
                else lane_streams[ln].push_back({sid({1, im[3], im[5], im[6]}), std::min(H, im[7]) * (p.split ? 4 : p.esize)});
            }
        n_streams = (int)keys.size();
    }
    auto refine_queues = [&](std::vector<std::vector<int>>& xq) {
        int n = 0; for (auto& q : xq) n += (int)q.size();
        if (n < 16) return;
        const int cap = (n + 7) / 8 + GW_XCD_SLACK;
        std::vector<int> where(p.n_lanes, -1), qsz(8, 0);
        std::vector<std::vector<int>> cnt(8, std::vector<int>(n_streams, 0));
        for (int x = 0; x < 8; ++x) for (int ln : xq[x]) { where[ln] = x; ++qsz[x]; for (auto& s : lane_streams[ln]) ++cnt[x][s.first]; }
        auto remove_gain = [&](int ln, int x) { int g = 0; for (auto& s : lane_streams[ln]) if (cnt[x][s.first] == 1) g += s.second; return g; };
        auto add_cost = [&](int ln, int x) { int c = 0; for (auto& s : lane_streams[ln]) if (cnt[x][s.first] == 0) c += s.second; return c; };
        auto take_out = [&](int ln, int x) { for (auto& s : lane_streams[ln]) --cnt[x][s.first]; --qsz[x]; };
        auto put_in = [&](int ln, int x) { for (auto& s : lane_streams[ln]) ++cnt[x][s.first]; ++qsz[x]; where[ln] = x; };
        for (int sweep = 0; sweep < 64; ++sweep) {
            bool improved = false;
            for (int a = 0; a < p.n_lanes; ++a) {
                if (where[a] < 0) continue;
                const int xa = where[a];
                // best single move
                int best_y = -1, best_d = 0;
                const int ga = remove_gain(a, xa);
                for (int y = 0; y < 8; ++y) {
                    if (y == xa || qsz[y] >= cap) continue;
                    const int dlt = add_cost(a, y) - ga;
                    if (dlt < best_d) { best_d = dlt; best_y = y; }
                }
                if (best_y >= 0) { take_out(a, xa); put_in(a, best_y); improved = true; continue; }
                // best swap with a lane of another queue
                int best_b = -1; best_d = 0;
                for (int b = a + 1; b < p.n_lanes; ++b) {
                    const int xb = where[b];
                    if (xb < 0 || xb == xa) continue;
                    take_out(a, xa); take_out(b, xb);
                    const int before_a = add_cost(a, xa), before_b = add_cost(b, xb);   // what the two lanes cost where they were
                    const int after = [&] { int c = add_cost(a, xb); put_in(a, xb); const int c2 = add_cost(b, xa); take_out(a, xb); return c + c2; }();
                    const int before = [&] { put_in(a, xa); const int c2 = add_cost(b, xb); take_out(a, xa); return before_a + c2; }();
                    (void)before_b;
                    put_in(a, xa); put_in(b, xb);
                    if (after - before < best_d) { best_d = after - before; best_b = b; }
                }
                if (best_b >= 0) { const int xb = where[best_b]; take_out(a, xa); take_out(best_b, xb); put_in(a, xb); put_in(best_b, xa); improved = true; }
            }
            if (!improved) break;
        }
        for (auto& q : xq) q.clear();
        for (int ln = 0; ln < p.n_lanes; ++ln) if (where[ln] >= 0) xq[where[ln]].push_back(ln);
    };
    auto queue_cost = [&](const std::vector<std::vector<int>>& xq) {
        long c = 0;
        for (auto& q : xq) { std::vector<char> seen(n_streams, 0); for (int ln : q) for (auto& s2 : lane_streams[ln]) if (!seen[s2.first]) { seen[s2.first] = 1; c += s2.second; } }
        return c; };
    auto build_queues = [&](const std::vector<bool>& take, std::vector<std::vector<int>>& xq) {
        xq.assign(8, {});
        for (auto& cs : cl_size) {
            int best = 0; for (int x = 1; x < 8; ++x) if (xq[x].size() < xq[best].size()) best = x;
            for (int ln = 0; ln < p.n_lanes; ++ln) if (take[ln] && lane_cluster[ln] == cs.second) xq[best].push_back(ln);
        }
        // rebalance: move lanes from the longest queue to the shortest while it shortens the maximum
        for (;;) {
            int lo = 0, hi = 0; for (int x = 1; x < 8; ++x) { if (xq[x].size() < xq[lo].size()) lo = x; if (xq[x].size() > xq[hi].size()) hi = x; }
            if (xq[hi].size() <= xq[lo].size() + 1) break;
            xq[lo].push_back(xq[hi].back()); xq[hi].pop_back();
        }
        refine_queues(xq);
        {   // alternative start: grow each queue along shared streams (seed, then always the lane that adds the fewest new bytes); keep the cheaper result
            std::vector<std::vector<int>> gq(8);
            std::vector<int> todo; for (int ln = 0; ln < p.n_lanes; ++ln) if (take[ln]) todo.push_back(ln);
            const int n = (int)todo.size();
            std::vector<bool> used(p.n_lanes, false);
            int left = n;
            for (int x = 0; x < 8 && left > 0; ++x) {
                const int want = (left + (8 - x) - 1) / (8 - x);
                std::vector<int> cnt(n_streams, 0);
                for (int k = 0; k < want; ++k) {
                    int best = -1, best_add = 0, best_share = 0;
                    for (int ln : todo) {
                        if (used[ln]) continue;
                        int add = 0, share = 0;
                        for (auto& s2 : lane_streams[ln]) { if (cnt[s2.first]) share += s2.second; else add += s2.second; }
                        if (best < 0 || (k > 0 && (add < best_add || (add == best_add && share > best_share)))) { best = ln; best_add = add; best_share = share; }
                        if (k == 0) break;      // seed: the first unplaced lane (target order)
                    }
                    used[best] = true; gq[x].push_back(best); --left;
                    for (auto& s2 : lane_streams[best]) ++cnt[s2.first];
                }
            }
            refine_queues(gq);
            if (queue_cost(gq) < queue_cost(xq)) xq = gq;
        }
        size_t m = 0; for (auto& q : xq) m = std::max(m, q.size());
        return m; };
    std::vector<std::vector<int>> xq;
    size_t qmax = build_queues(std::vector<bool>(p.n_lanes, true), xq);
    p.n_lanes_pad = (int)qmax * 8;
    p.n_wg_gradw = p.n_lanes * p.n_parts;
    p.tgt_off = (int)T.size(); p.n_targets = (int)tgts.size();
    for (size_t g = 0; g < tgts.size(); ++g) {
        T.push_back(tgt_lane_begin[g]); T.push_back(tgt_lane_end[g]); T.push_back(tgts[g].bias_flag);
        for (int k = 3; k < TGT_INTS; ++k) T.push_back(0);
    }
    p.lane_off = (int)T.size();
    for (auto& ln : lanes) for (int k = 0; k < LANE_INTS; ++k) T.push_back(ln[k]);
    p.lane_order_off = (int)T.size();
    for (int k = 0; k < (int)qmax; ++k) for (int x = 0; x < 8; ++x) T.push_back(k < (int)xq[x].size() ? xq[x][k] : -1);
    p.n_slabs = p.n_wg_gradw;
    // phase tables: the encoder targets are the last ones, so their lanes are the tail [lane_split, n_lanes)
    int lane_split = p.n_lanes;
    std::vector<bool> tgt_is_enc(tgts.size(), false);
    for (int t = 0; t < NT; ++t) for (int g : tgt_enc[t]) { tgt_is_enc[g] = true; lane_split = std::min(lane_split, tgt_lane_begin[g]); }
    for (int ph = 0; ph < 2; ++ph) {
        std::vector<bool> take(p.n_lanes);
        for (int ln = 0; ln < p.n_lanes; ++ln) take[ln] = (ln >= lane_split) == (ph == 1);
        std::vector<std::vector<int>> q2;
        const size_t m = build_queues(take, q2);
        p.order_ph_off[ph] = (int)T.size(); p.npad_ph[ph] = (int)m * 8;
        for (int k = 0; k < (int)m; ++k) for (int x = 0; x < 8; ++x) T.push_back(k < (int)q2[x].size() ? q2[x][k] : -1);
    }

    // finalize ops: every parameter is written exactly once
    struct FinRec { int64_t dst; int rows, cols, ld, target, kind; std::vector<int64_t> more; };
    std::vector<FinRec> fins;
    // Ops that share a slab sum (the pre-summed root weight and the bias of a destination type go to EVERY relation into that type) are ONE op with
    // several destinations: the slabs are read once (A1-C2 L=3: 75 -> 52 MB of slab reads per finalize launch).  Round 1 measured the merge slower --
    // with its coarse grid; the (ops x 64 row groups) grid of today has one output element per thread either way.
    auto add_fin = [&](int64_t dst, int rows, int cols, int ld, int target, int kind) {
        if (target >= 0 && (kind == FIN_MATRIX || kind == FIN_BIAS))
            for (auto& f : fins)
                if (f.target == target && f.kind == kind && f.rows == rows && f.cols == cols && f.ld == ld) { f.more.push_back(dst); return; }
        fins.push_back({dst, rows, cols, ld, target, kind, {}}); };
    for (int t = 0; t < NT; ++t) {
        const int F = d.type_width[t];
        for (int kc = 0; kc < p.enc_nkc[t]; ++kc) {
            const int nc = std::min(H, F - kc * H);
            const int tg = tgt_enc[t].empty() ? -1 : tgt_enc[t][kc];
            add_fin(p.off_enc_w[t] + kc * H, H, nc, F, tg, tg < 0 ? FIN_ZERO : FIN_MATRIX);
        }
        const int tg0 = tgt_enc[t].empty() ? -1 : tgt_enc[t][0];
        add_fin(p.off_enc_b[t], 1, H, H, tg0, tg0 < 0 ? FIN_ZERO : FIN_BIAS);
    }
    for (int l = 0; l < L; ++l)
        for (int r = 0; r < NR; ++r) {
            const int tr = tgt_rel[l * NR + r], to = tgt_root[l * NT + p.rel_dst[r]];
            add_fin(p.off_rel_w[l * NR + r], H, H, H, tr, tr < 0 ? FIN_ZERO : FIN_MATRIX);
            add_fin(p.off_rel_b[l * NR + r], 1, H, H, to, to < 0 ? FIN_ZERO : FIN_BIAS);
            add_fin(p.off_root_w[l * NR + r], H, H, H, to, to < 0 ? FIN_ZERO : FIN_MATRIX);
        }
    if (has_mlp) {
        for (int k = 0; k < 2; ++k) {
            add_fin(d.off_mlp[2 * k], H, H, H, tgt_mlp[k], tgt_mlp[k] < 0 ? FIN_ZERO : FIN_MATRIX);
            add_fin(d.off_mlp[2 * k + 1], 1, H, H, tgt_mlp[k], tgt_mlp[k] < 0 ? FIN_ZERO : FIN_BIAS);
        }
    }
    add_fin(d.off_dec_w, d.out_channels, H, H, -2, FIN_DEC_W);
    add_fin(d.off_dec_b, 1, d.out_channels, d.out_channels, -2, FIN_DEC_B);
    {   // alignment gaps of the flat buffer are zeroed, so grad_params really is fully overwritten
        std::vector<std::pair<int64_t, int64_t>> spans;    // (offset, length) of every parameter
        for (int t = 0; t < NT; ++t) { spans.push_back({p.off_enc_w[t], (int64_t)H * d.type_width[t]}); spans.push_back({p.off_enc_b[t], H}); }
        for (int i = 0; i < L * NR; ++i) { spans.push_back({p.off_rel_w[i], (int64_t)H * H}); spans.push_back({p.off_rel_b[i], H}); spans.push_back({p.off_root_w[i], (int64_t)H * H}); }
        if (has_mlp) for (int k = 0; k < 2; ++k) { spans.push_back({d.off_mlp[2 * k], (int64_t)H * H}); spans.push_back({d.off_mlp[2 * k + 1], H}); }
        spans.push_back({d.off_dec_w, (int64_t)d.out_channels * H}); spans.push_back({d.off_dec_b, d.out_channels});
        std::sort(spans.begin(), spans.end());
        int64_t pos = 0;
        for (auto& sp : spans) {
            if (sp.first > pos && sp.first - pos < 4096) add_fin(pos, 1, (int)(sp.first - pos), (int)(sp.first - pos), -1, FIN_ZERO);
            pos = std::max(pos, sp.first + sp.second);
        }
        if (d.n_flat > pos && d.n_flat - pos < 4096) add_fin(pos, 1, (int)(d.n_flat - pos), (int)(d.n_flat - pos), -1, FIN_ZERO);
    }
    {   // phase split of the flat buffer: [0, grad_split) = encoder parameters (phase 1), the rest phase 0 -- only when the
        // encoder really is a prefix of the buffer (state_dict order: encoder.lins.* first)
        int64_t enc_end = 0, rest_begin = d.n_flat;
        for (int t = 0; t < NT; ++t) { enc_end = std::max(enc_end, p.off_enc_w[t] + (int64_t)H * d.type_width[t]); enc_end = std::max(enc_end, p.off_enc_b[t] + (int64_t)H); }
        for (int i = 0; i < L * NR; ++i) { rest_begin = std::min(rest_begin, std::min(p.off_rel_w[i], std::min(p.off_rel_b[i], p.off_root_w[i]))); }
        if (has_mlp) for (int k = 0; k < 4; ++k) rest_begin = std::min<int64_t>(rest_begin, d.off_mlp[k]);
        rest_begin = std::min<int64_t>(rest_begin, std::min(d.off_dec_w, d.off_dec_b));
        p.grad_split = (enc_end <= rest_begin && lane_split < p.n_lanes) ? rest_begin : -1;
        auto is_ph1 = [&](const FinRec& f) { return p.grad_split >= 0 && f.dst < p.grad_split; };
        std::stable_sort(fins.begin(), fins.end(), [&](const FinRec& x, const FinRec& y) { return is_ph1(x) < is_ph1(y); });
        p.fin_off = (int)T.size(); p.n_fin = (int)fins.size(); p.n_fin_ph0 = 0;
        std::vector<int32_t> extra;      // destination lists behind the fixed-size records
        for (auto& f : fins) {
            if (!is_ph1(f)) ++p.n_fin_ph0;
            if (is_ph1(f) && f.target >= 0 && !tgt_is_enc[f.target]) p.grad_split = -1;      // (cannot happen with the layouts the host produces)
            for (int64_t m : f.more) if ((p.grad_split >= 0 && m < p.grad_split) != is_ph1(f)) p.grad_split = -1;      // (a merged op never straddles the phases)
            T.push_back((int32_t)(f.dst & 0xffffffff)); T.push_back((int32_t)(f.dst >> 32)); T.push_back(f.rows); T.push_back(f.cols);
            T.push_back(f.ld); T.push_back(f.target); T.push_back(f.kind);
            T.push_back(f.more.empty() ? 0 : (int32_t)(fins.size() * FIN_INTS + extra.size()));
            if (!f.more.empty()) {
                extra.push_back((int32_t)f.more.size());
                for (int64_t m : f.more) { extra.push_back((int32_t)(m & 0xffffffff)); extra.push_back((int32_t)(m >> 32)); }
            }
        }
        T.insert(T.end(), extra.begin(), extra.end());
        if (p.grad_split < 0) p.n_fin_ph0 = p.n_fin;
    }

    // ---- info ---------------------------------------------------------------------------------------
    p.info.rows_per_tile = p.rows; p.info.total_nodes = p.NN; p.info.lds_bytes = (int64_t)p.n_blk * p.blk_bytes;
    p.info.flops_fwd = alg_fwd; p.info.flops_bwd = alg_bwd; p.info.flops_exec_fwd = exec_fwd; p.info.flops_exec_bwd = exec_bwd;
    double bytes = 0, bytes_all = 0;     // split plan: fp32 inputs.  bytes: what the plan has to read (nodes whose inputs can reach the output); bytes_all: every node
    for (int t = 0; t < NT; ++t) { bytes += (double)enc_n[t] * d.type_width[t] * (p.split ? 4 : p.esize); bytes_all += (double)d.type_nodes[t] * d.type_width[t] * (p.split ? 4 : p.esize); }
    p.info.bytes_in = bytes_all; p.info.bytes_in_live = bytes; p.info.n_gradw_workgroups = p.n_wg_gradw;
    p.info.n_launches_fwd = 3 + L; p.info.n_launches_bwd = 3 + L;
    p.info.grad_split = p.grad_split;
    p.info.kernel_sets = (p.fused ? 1 : 0) | (p.slab ? 2 : 0);

    // ---- per-kernel work table (launch order of one fwd+bwd step) -----------------------------------------
    {
        const double es = p.esize * p.planes, act = (double)p.enc_live_nodes * H * es;      // X_0 rows that exist
        auto add = [&](const std::string& name, int bound, double fa, double fe, double by) {
            mshgnn_kernel_stat k{}; std::snprintf(k.name, sizeof(k.name), "%s", name.c_str());
            k.bound = bound; k.flops_per_window = fa; k.flops_exec_per_window = fe; k.bytes_per_window = by;
            p.kstats.push_back(k); return (int)p.kstats.size() - 1; };
        auto live_nodes = [&](int l) { int n = 0; for (int i = 0; i < p.NN; ++i) n += p.live_n[l][i] ? 1 : 0; return n; };
        auto need_nodes = [&](int l) { int n = 0; for (int i = 0; i < p.NN; ++i) n += p.need_n[l][i] ? 1 : 0; return n; };
        p.ks_prep = add("prep", MSHGNN_BOUND_HBM, 0, 0, 0);
        p.ks_enc = add("enc_fwd", d.dtype == MSHGNN_F32 ? MSHGNN_BOUND_MFMA : MSHGNN_BOUND_HBM, enc_alg, enc_exec, bytes + act);
        for (int l = 0; l < L; ++l) {
            const int idx = add("layer_fwd" + std::to_string(l), MSHGNN_BOUND_MFMA, lf_alg[l], lf_exec[l],
                                act + live_nodes(l) * (double)H * es + live_nodes(l) * 16.0);
            if (l == 0) p.ks_layer_fwd0 = idx;
        }
        p.ks_dec_fwd = add("dec_fwd", MSHGNN_BOUND_HBM, 2.0 * n_out * d.out_channels * H, 2.0 * n_out * d.out_channels * H, n_out * (double)H * es);
        p.ks_dec_bwd = add("dec_bwd", MSHGNN_BOUND_HBM, 4.0 * n_out * d.out_channels * H, 4.0 * n_out * d.out_channels * H, 2.0 * n_out * (double)H * es);
        for (int l = L - 1; l >= 0; --l) {
            const int idx = add("layer_bwd" + std::to_string(l), MSHGNN_BOUND_MFMA, lb_alg[l], lb_exec[l],
                                live_nodes(l) * (double)H * es * 2 + need_nodes(l) * (double)H * es + live_nodes(l) * 16.0);
            if (l == L - 1) p.ks_layer_bwd0 = idx;
        }
        // k_gradw reads every P / Q operand once at best: ~134 FLOP per algorithmic byte at bf16 -> below the ridge
        // (2.5 PF / 8 TB/s = 312 FLOP/B), so HBM is the bounding roofline (fp32: 157 TF / 8 TB/s = 20 FLOP/B -> MFMA-bound)
        p.ks_gradw = add("gradw", d.dtype == MSHGNN_F32 ? MSHGNN_BOUND_MFMA : MSHGNN_BOUND_HBM, gw_alg, gw_exec, bytes + (double)L * 2 * act + act);
        p.ks_fin = add("finalize", MSHGNN_BOUND_HBM, 0, 0, 0);
        if (p.fused) {   // the fused stack kernels replace layer_fwd* + dec_fwd and layer_bwd* of the step
            double fa = 0, fe = 0, fb = act, ba = 0, be = 0, bb = act;
            for (int l = 0; l < L; ++l) {
                fa += lf_alg[l]; fe += lf_exec[l]; fb += live_nodes(l) * (double)H * es + live_nodes(l) * 16.0;
                ba += lb_alg[l]; be += lb_exec[l]; bb += need_nodes(l) * (double)H * es + live_nodes(l) * 16.0;
            }
            fa += 2.0 * n_out * d.out_channels * H; fe += 2.0 * n_out * d.out_channels * H;
            p.ks_stack_fwd = add("stack_fwd", MSHGNN_BOUND_MFMA, fa, fe, fb);
            p.ks_stack_bwd = add("stack_bwd", MSHGNN_BOUND_MFMA, ba, be, bb);
            p.ks_stack_step = add("stack_step", MSHGNN_BOUND_MFMA, fa + ba, fe + be, fb + bb);      // one-call steps on the slab kernels: both sweeps in one launch
        }
    }
    return true;
}

// workspace layout ------------------------------------------------------------------------------------
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline void layout_workspace(const HostPlan& p, int64_t B, int training, mshgnn_ws_layout* o) {
    std::memset(o, 0, sizeof(*o));
    size_t off = 0;
    const size_t act = (size_t)B * p.NN * H * p.esize * p.planes, mlp = (size_t)B * std::max(1, p.n_mlp) * H * p.esize * p.planes;     // split plan: rows of [hi 128 | lo 128]
    auto take = [&](size_t bytes) { size_t r = off; off = align_up(off + bytes, 256); return r; };
    for (int l = 0; l <= p.L; ++l) o->x[l] = take(act);
    for (int l = 0; l < p.L; ++l) { o->mask[l] = take((size_t)((B + 15) / 16 * 16) * p.NN * 4 * 4); o->hb[l] = take(mlp); o->t1[l] = take(mlp); }
    if (training) {
        o->dd[0] = take((size_t)((B + 15) / 16 * 16) * p.NN * 4 * 4);     // relu bytes of the encoder activation X_0 (layout as mask[l])
        for (int l = 0; l <= p.L; ++l) o->dx[l] = take(act);
        for (int l = 0; l < p.L; ++l) { o->dh[l] = take(act); o->du[l] = take(mlp); }
        o->slabs = take((size_t)p.n_slabs * SLAB_FLOATS * 4);
        // decoder partials: NWG_DEC from k_dec_bwd, or one per 16-window tile when the fused forward produces them (mshgnn_step_mse)
        o->dec_slabs = take((size_t)std::max<int64_t>(NWG_DEC, (B + TILE_ROWS - 1) / TILE_ROWS) * DEC_SLAB_FLOATS * 4);
    }
    o->wpack = take(p.packs.size() * (size_t)H * H * p.esize * p.planes);
    o->bias = take(p.biases.size() * (size_t)H * 4);
    o->loss = take(64);
    o->total = off;
}

}  // namespace mshgnn
