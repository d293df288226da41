// Host-side plan compiler of the GENERIC-WIDTH MS-HGNN engine (mshgnn_gen.hip): any hidden width that is a multiple of 128, any
// number of nodes per window, any in-degree, 'add' and 'mean' aggregation -- what the LDS-resident stack kernels of mshgnn.hip
// (hidden == 128, <= 20 nodes) cannot hold, e.g. BASELINE.json configs[4] (synthetic 32-limb robot, h = 512, 6 layers).
//
// Same path and the same algebra as mshgnn_plan.hpp (GRF_HGNN.forward hgnn.py:57-62, GRF_HGNN_C2.forward hgnn_c2.py:133-182,
// GRF_HGNN_K4.forward hgnn_k4.py:146-196; PyG-2.5.0 HeteroConv / GraphConv semantics):
//     H_d[i] = (sum_r W_root^r) X_d[i] + sum_r b_rel^r + sum_r W_rel^r Agg_r({X_s[j] : j -> i in r})
// but lowered to a grouped GEMM instead of an LDS-resident tile: every (destination row, 64-window tile, 128-column tile) is a JOB whose
// K loop runs over TERMS = (weight pack, sources to aggregate).  Aggregation happens while a term's A tile is staged (sum / mean of
// the source rows in fp32 -- the reference's own order: aggregate, then lin_rel), so a node with in-degree 32 costs one block GEMM
// per relation, not 32.  The backward pass is the same job machinery on the transposed graph; weight gradients are split-K
// 128x128 tiles dW = P^T Q with Q aggregated the same way.  No HIP calls in this file.
#pragma once
#include "mshgnn_plan.hpp"
#include <climits>
#include <cstdlib>
#include <map>

namespace mshgnn {
namespace gen {

constexpr int TW = 128;                 // tile width: output-column tile of a job, K chunk of a term
constexpr int G_MAX_L = 16;
enum { JF_BIAS = 1, JF_RELU = 2, JF_RES = 4, JF_BITS_OUT = 8, JF_GATE_BITS = 16, JF_GATE_POS = 32, JF_DHM = 64, JF_DHM_ONLY = 128 };      // JF_DHM: also write out . relu bits (J_BITS_BUF) to J_DHM_BUF; JF_DHM_ONLY: and nobody reads the unmasked row (no residual): only that
// job:  out_buf out_node flags bias_idx | res_buf res_node bits_buf gate_buf | gate_node term_begin n_terms pad...
enum { J_OUT_BUF = 0, J_OUT_NODE, J_FLAGS, J_BIAS, J_RES_BUF, J_RES_NODE, J_BITS_BUF, J_GATE_BUF, J_GATE_NODE, J_TERM0, J_NTERMS, J_DHM_BUF, JOB_INTS = 12 };
// term: pack_base nkc kind(0 activation, 1 raw input) src_begin | n_src width sign_off pad     pack of (kc, ct) = pack_base + kc * NCT + ct
enum { T_PACK = 0, T_NKC, T_KIND, T_SRC0, T_NSRC, T_WIDTH, T_SIGN, TERM_INTS = 8 };
// source: buf (raw: node type) node mask_buf(-1: none) scale(float bits)
enum { S_BUF = 0, S_NODE, S_MASK, S_SCALE, SRC_INTS = 4 };
// weight-gradient unit = one 128x128 tile of one target over a chunk of its items: item_begin item_end p_col0 q_col0 | q_ncols bias_flag pad pad
enum { U_ITEM0 = 0, U_ITEM1, U_PCOL, U_QCOL, U_QN, U_BIAS, UNIT_INTS = 8 };
enum { SU_UNIT = 0, SU_ITEM0 = 4, SU_ITEM1, SU_PCOL, SU_QCOL, SU_QN, SU_FLAGS, SUNIT_INTS = 12 };      // SU_FLAGS: 1 every item has one plain source (lean streams), 2 bias sums read
// item: p_buf p_node p_mask_buf kind(0 activation sources, 1 raw input) | src_begin n_src pad pad
enum { I_PBUF = 0, I_PNODE, I_PMASK, I_KIND, I_SRC0, I_NSRC, GITEM_INTS = 8 };
// finalize op: dst_lo dst_hi rows cols | ld kind unit_begin n_units | src_row0 pad pad pad     (units of one target tile are consecutive)
enum { GF_DST_LO = 0, GF_DST_HI, GF_ROWS, GF_COLS, GF_LD, GF_KIND, GF_UNIT0, GF_NUNITS, GFIN_INTS = 12 };
constexpr int GBUF_MASK0 = BUF_COUNT;   // relu bytes of the encoder activation X_0 (workspace dd[0]); BUF_* ids as in mshgnn_plan.hpp
// Aggregates of MANY rows (more than G_MANY sources: the base node of an N-limb model sums N rows) are computed ONCE per layer and direction by their own small
// launch (k_gagg) into one-node-per-aggregate buffers, and the job / weight-gradient item that consumes them sees one plain row.  Gathered inside the job kernel
// they are one dependent round trip per source and staged row: at 32 sources the base node's workgroups run 263-347 us, as long as the whole layer launch
// (tools/timeline_gen.py), and pipelining that gather in place cost the plain path its registers (+50 us on every launch).  The crossover was measured on the
// 32-limb model: with the aggregate launches its layer launches take 250 -> 241 / 270 -> 252 us and the weight gradients lose their slowest super-units, but
// k_gagg itself takes 23 us per launch (32 MB gathered from 32 strided regions at 1.4 TB/s), a wash at 32 rows -- so the threshold sits above it.
constexpr int GBUF_AGGF = BUF_COUNT + 1, GBUF_AGGB = GBUF_AGGF + G_MAX_L;      // forward (sums of X_l rows) / backward (sums of masked dX_{l+1} rows) aggregates of layer l
constexpr int GBUF_COUNT = GBUF_AGGB + G_MAX_L;
constexpr int G_MANY = 32;
enum { AG_OUT_BUF = 0, AG_OUT_NODE, AG_SRC0, AG_NSRC, AGG_INTS = 4 };
constexpr int G_ITEMS_PER_UNIT = 8;     // items one weight-gradient workgroup sweeps (more units = more parallelism, more slabs to sum)

struct Launch { int job0, n_jobs, ks; int agg0 = 0, n_agg = 0; bool all_plain = false, any_mask = false, all_raw = false; int max_terms = 0; };      // all_raw: every term of every job is ONE raw input row (the encoder launch)      // agg0 / n_agg: the aggregates computed in front of the launch; all_plain: every term of every job is ONE activation row at scale 1 (k_gstep5's precondition)

struct GenPlan {
    mshgnn_desc d{};
    std::vector<int32_t> rel_src, rel_dst, rel_mean, rel_edge_off, edges;
    std::vector<float> out_mask_f;
    std::vector<int64_t> off_enc_w, off_enc_b, off_rel_w, off_rel_b, off_root_w;
    int L = 0, NT = 0, NR = 0, NN = 0, Hd = 0, NCT = 0, n_mlp = 0;
    bool split = false; int esize = 2, planes = 1;
    bool dhm = false;      // dH_l = dX_{l+1} . relu bits of layer l is WRITTEN by the producer of dX_{l+1} (into the dH stash of layer l) instead of masked by every reader
    int type_base[MSHGNN_MAX_TYPES + 1]{};
    std::vector<int> node_type;
    bool live[G_MAX_L][MSHGNN_MAX_TYPES]{}, need_dx[G_MAX_L][MSHGNN_MAX_TYPES]{};
    std::vector<PackDesc> packs; int n_img = 0;
    std::vector<BiasDesc> biases;
    std::vector<uint8_t> signs; std::vector<int> sign_off, enc_nkc;
    std::vector<int32_t> tables;            // jobs | terms | srcs | units | items | fins
    int agg_off = 0, n_aggbuf = 0;      // aggregate ops (AGG_INTS each); nodes per aggregate buffer
    int job_off = 0, term_off = 0, src_off = 0, unit_off = 0, sunit_off = 0, su_order_off = 0, n_sunits = 0, su_os = 1, item_off = 0, fin_off = 0, n_units = 0, n_parts = 1, n_fin = 0;
    std::vector<Launch> fwd, bwd;           // job launches in order
    int ks_prep = 0, ks_dec_fwd = 0, ks_dec_bwd = 0, ks_gradw = 0, ks_fin = 0, ks_agg = -1;
    std::vector<mshgnn_kernel_stat> kstats;
    mshgnn_info info{};
    std::string err;
};

inline bool gfail(GenPlan& p, const std::string& m) { p.err = m; return false; }

inline bool compile_gen_plan(const mshgnn_desc* din, GenPlan& p) {
    if (!din) return gfail(p, "null descriptor");
    p.d = *din;
    const mshgnn_desc& d = p.d;
    if (d.n_types < 1 || d.n_types > MSHGNN_MAX_TYPES) return gfail(p, "n_types must be 1..4");
    if (d.hidden < TW || d.hidden % TW || d.hidden > 2048) return gfail(p, "hidden_channels must be a multiple of 128 (128..2048): not supported by this build");
    if (d.num_layers < 1 || d.num_layers > G_MAX_L) return gfail(p, "num_layers must be 1..16");
    if (d.n_rel < 1 || d.n_rel > 64) return gfail(p, "n_rel must be 1..64");
    if (d.dtype != MSHGNN_BF16 && d.dtype != MSHGNN_BF16X3 && d.dtype != MSHGNN_F32)
        return gfail(p, "dtype must be MSHGNN_F32, MSHGNN_BF16 or MSHGNN_BF16X3");
    if (d.out_type < 0 || d.out_type >= d.n_types) return gfail(p, "out_type out of range");
    if (d.out_channels < 1 || d.out_channels > 8) return gfail(p, "out_channels must be 1..8");
    if (!d.rel_src || !d.rel_dst || !d.rel_mean || !d.rel_edge_off || !d.edges) return gfail(p, "null relation arrays");
    if (!d.off_enc_w || !d.off_enc_b || !d.off_rel_w || !d.off_rel_b || !d.off_root_w) return gfail(p, "null offset arrays");
    // the generic engine has two arithmetic modes: bf16, and the split-bf16 parity mode.  MSHGNN_F32 asks for parity: it is served by
    // the split mode (1e-4 relative) -- there is no fp32-MFMA variant of these kernels.
    p.split = d.dtype != MSHGNN_BF16; p.planes = p.split ? 2 : 1; p.esize = 2;
    p.L = d.num_layers; p.NT = d.n_types; p.NR = d.n_rel; p.Hd = d.hidden; p.NCT = d.hidden / TW;
    const int L = p.L, NT = p.NT, NR = p.NR, Hd = p.Hd, NCT = p.NCT;
    const bool has_mlp = (d.flags & MSHGNN_FLAG_BASE_MLP) != 0, residual = (d.flags & MSHGNN_FLAG_RESIDUAL) != 0;
    if (has_mlp && (d.mlp_type < 0 || d.mlp_type >= NT)) return gfail(p, "mlp_type out of range");
    p.type_base[0] = 0;
    for (int t = 0; t < NT; ++t) {
        if (d.type_nodes[t] < 1 || d.type_width[t] < 1) return gfail(p, "every node type needs >= 1 node and input width >= 1");
        p.type_base[t + 1] = p.type_base[t] + d.type_nodes[t];
        for (int i = 0; i < d.type_nodes[t]; ++i) p.node_type.push_back(t);
    }
    p.NN = p.type_base[NT];
    if (p.NN > 4096) return gfail(p, "more than 4096 nodes per window: not supported by this build");
    p.n_mlp = has_mlp ? d.type_nodes[d.mlp_type] : 0;
    p.rel_src.assign(d.rel_src, d.rel_src + NR); p.rel_dst.assign(d.rel_dst, d.rel_dst + NR);
    p.rel_mean.assign(d.rel_mean, d.rel_mean + NR); p.rel_edge_off.assign(d.rel_edge_off, d.rel_edge_off + NR + 1);
    const int E = p.rel_edge_off[NR];
    if (p.rel_edge_off[0] != 0 || E < 0) return gfail(p, "bad rel_edge_off");
    p.edges.assign(d.edges, d.edges + 2 * (size_t)E);
    p.off_enc_w.assign(d.off_enc_w, d.off_enc_w + NT); p.off_enc_b.assign(d.off_enc_b, d.off_enc_b + NT);
    p.off_rel_w.assign(d.off_rel_w, d.off_rel_w + (size_t)L * NR); p.off_rel_b.assign(d.off_rel_b, d.off_rel_b + (size_t)L * NR);
    p.off_root_w.assign(d.off_root_w, d.off_root_w + (size_t)L * NR);
    std::vector<bool> has_in(NT, false);
    // in-edges per (relation, dst node) and out-edges per (relation, src node); mean scale = 1 / in-degree of the dst node
    std::vector<std::vector<std::vector<int>>> in_src(NR), out_dst(NR);
    for (int r = 0; r < NR; ++r) {
        const int s = p.rel_src[r], t = p.rel_dst[r];
        if (s < 0 || s >= NT || t < 0 || t >= NT) return gfail(p, "relation type index out of range");
        if (p.rel_edge_off[r + 1] < p.rel_edge_off[r]) return gfail(p, "rel_edge_off must be non-decreasing");
        has_in[t] = true;
        in_src[r].assign(d.type_nodes[t], {}); out_dst[r].assign(d.type_nodes[s], {});
        for (int e = p.rel_edge_off[r]; e < p.rel_edge_off[r + 1]; ++e) {
            const int j = p.edges[2 * e], i = p.edges[2 * e + 1];
            if (j < 0 || j >= d.type_nodes[s] || i < 0 || i >= d.type_nodes[t]) return gfail(p, "edge endpoint out of range");
            in_src[r][i].push_back(j); out_dst[r][j].push_back(i);
        }
    }
    for (int t = 0; t < NT; ++t)
        if (!has_in[t]) return gfail(p, "every node type must be the destination of at least one relation (HeteroConv drops types without incoming relations)");
    auto scale_of = [&](int r, int i) { return p.rel_mean[r] ? 1.0f / (float)std::max<size_t>(1, in_src[r][i].size()) : 1.0f; };
    auto fbits = [](float f) { int32_t b; std::memcpy(&b, &f, 4); return b; };

    // symmetry masks (+-1) -> sign bytes per (node, K chunk)
    p.sign_off.assign(NT, 0); p.enc_nkc.assign(NT, 0);
    for (int t = 0; t < NT; ++t) {
        const int F = d.type_width[t], n = d.type_nodes[t];
        const int nkc = (F + TW - 1) / TW; p.enc_nkc[t] = nkc;
        p.sign_off[t] = (int)p.signs.size();
        p.signs.resize(p.signs.size() + (size_t)n * nkc * TW, 0);
        if (d.in_mask[t])
            for (int i = 0; i < n; ++i) for (int k = 0; k < F; ++k) {
                const float m = d.in_mask[t][(size_t)i * F + k];
                if (m != 1.0f && m != -1.0f) return gfail(p, "symmetry masks must be +1 or -1");
                p.signs[p.sign_off[t] + (size_t)i * nkc * TW + k] = m < 0 ? 1 : 0;
            }
    }
    const int n_out = d.type_nodes[d.out_type];
    p.out_mask_f.assign((size_t)n_out * d.out_channels, 1.0f);
    if (d.out_mask) for (size_t i = 0; i < p.out_mask_f.size(); ++i) {
        if (d.out_mask[i] != 1.0f && d.out_mask[i] != -1.0f) return gfail(p, "output mask must be +1 or -1");
        p.out_mask_f[i] = d.out_mask[i];
    }

    // ---- liveness, node level (as mshgnn_plan.hpp, round 4): live_n[l][n] -- the output of layer l at node n can reach the decoder; need_n[l][n] -- X_l[n]
    // feeds a live node of layer l (itself: root weight / residual; or a destination of one of its edges), so dX_l[n] is produced; need_n[0] = the nodes
    // the encoder computes.  MSHGNN_PRUNE=0: whole types, as in rounds 1-3.  The base_transform type is live as a whole.
    const bool prune = []() { const char* e = std::getenv("MSHGNN_PRUNE"); return !(e && std::atoi(e) == 0); }();
    std::vector<std::vector<char>> live_n(L, std::vector<char>(p.NN, 0)), need_n(L, std::vector<char>(p.NN, 0));
    {
        auto widen = [&](std::vector<char>& v) {
            for (int t = 0; t < NT; ++t) {
                if (prune && !(has_mlp && t == d.mlp_type)) continue;
                bool any = false;
                for (int i = 0; i < d.type_nodes[t]; ++i) any = any || v[p.type_base[t] + i];
                if (any) for (int i = 0; i < d.type_nodes[t]; ++i) v[p.type_base[t] + i] = 1;
            }
        };
        for (int n = 0; n < p.NN; ++n) live_n[L - 1][n] = p.node_type[n] == d.out_type;
        for (int l = L - 1; l >= 0; --l) {
            widen(live_n[l]);
            need_n[l] = live_n[l];
            for (int r = 0; r < NR; ++r) {
                bool dl = false;
                for (int i = 0; i < d.type_nodes[p.rel_dst[r]]; ++i) {
                    if (!live_n[l][p.type_base[p.rel_dst[r]] + i]) continue;
                    dl = true;
                    for (int j : in_src[r][i]) need_n[l][p.type_base[p.rel_src[r]] + j] = 1;
                }
                if (!prune && dl) for (int j = 0; j < d.type_nodes[p.rel_src[r]]; ++j) need_n[l][p.type_base[p.rel_src[r]] + j] = 1;
            }
            if (l == 0) widen(need_n[0]);
            if (l > 0) live_n[l - 1] = need_n[l];
        }
        for (int l = 1; l < L; ++l) need_n[l] = live_n[l - 1];
        for (int l = 0; l < L; ++l)
            for (int t = 0; t < NT; ++t) {
                p.live[l][t] = p.need_dx[l][t] = false;
                for (int i = 0; i < d.type_nodes[t]; ++i) { p.live[l][t] = p.live[l][t] || live_n[l][p.type_base[t] + i]; p.need_dx[l][t] = p.need_dx[l][t] || need_n[l][p.type_base[t] + i]; }
            }
    }
    auto rel_live = [&](int l, int r) {      // relation r has an edge into a live node of layer l
        for (int i = 0; i < d.type_nodes[p.rel_dst[r]]; ++i) if (live_n[l][p.type_base[p.rel_dst[r]] + i] && !in_src[r][i].empty()) return true;
        return false; };

    // ---- packs: a [rows x K] weight becomes ceil(K/128) x NCT(rows) images of 128x128, pack(kc, ct) = base + kc * nct + ct ------------
    // orient 0 (forward, out = A W^T):   B[k][c] = W[ct*128 + c][kc*128 + k]     orient 1 (backward, dA = dH W):   B[k][c] = W[kc*128 + k][ct*128 + c]
    auto add_packs = [&](int orient, const std::vector<int64_t>& src, int rows, int K) {
        const int nkc = orient == 0 ? (K + TW - 1) / TW : rows / TW, nct = orient == 0 ? rows / TW : (K + TW - 1) / TW;
        const int base = (int)p.packs.size();
        for (int kc = 0; kc < nkc; ++kc)
            for (int ct = 0; ct < nct; ++ct) {
                PackDesc q{}; q.orient = orient; q.n_src = (int)src.size(); q.ld = K;
                if (orient == 0) { q.col0 = kc * TW; q.ncols = std::min(TW, K - kc * TW); for (size_t i = 0; i < src.size(); ++i) q.src[i] = src[i] + (int64_t)ct * TW * K; }
                else { q.col0 = 0; q.ncols = TW; for (size_t i = 0; i < src.size(); ++i) q.src[i] = src[i] + (int64_t)kc * TW * K + (int64_t)ct * TW; }
                p.packs.push_back(q);
            }
        return base; };
    auto add_bias = [&](const std::vector<int64_t>& src) {      // an Hd-wide bias = NCT consecutive 128-wide bias rows
        const int base = (int)p.biases.size();
        for (int ct = 0; ct < NCT; ++ct) { BiasDesc b{}; b.n_src = (int)src.size(); for (size_t i = 0; i < src.size(); ++i) b.src[i] = src[i] + ct * TW; p.biases.push_back(b); }
        return base; };
    std::vector<int> pack_root[2], pack_rel[2], bias_layer((size_t)L * NT, -1), pack_enc(NT, -1), bias_enc(NT, -1);
    for (int o = 0; o < 2; ++o) { pack_root[o].assign((size_t)L * NT, -1); pack_rel[o].assign((size_t)L * NR, -1); }
    int pack_mlp[2][2] = {{-1, -1}, {-1, -1}}, bias_mlp[2] = {-1, -1};
    for (int l = 0; l < L; ++l) {
        for (int t = 0; t < NT; ++t) {
            if (!p.live[l][t]) continue;
            std::vector<int64_t> ws, bs;
            for (int r = 0; r < NR; ++r) if (p.rel_dst[r] == t) { ws.push_back(p.off_root_w[l * NR + r]); bs.push_back(p.off_rel_b[l * NR + r]); }
            if (ws.size() > 8) return gfail(p, "more than 8 relations into one node type");
            pack_root[0][l * NT + t] = add_packs(0, ws, Hd, Hd);
            pack_root[1][l * NT + t] = add_packs(1, ws, Hd, Hd);
            bias_layer[l * NT + t] = add_bias(bs);
        }
        for (int r = 0; r < NR; ++r) {
            if (!rel_live(l, r)) continue;
            pack_rel[0][l * NR + r] = add_packs(0, {p.off_rel_w[l * NR + r]}, Hd, Hd);
            pack_rel[1][l * NR + r] = add_packs(1, {p.off_rel_w[l * NR + r]}, Hd, Hd);
        }
    }
    if (has_mlp) {
        for (int o = 0; o < 2; ++o) { pack_mlp[o][0] = add_packs(o, {d.off_mlp[0]}, Hd, Hd); pack_mlp[o][1] = add_packs(o, {d.off_mlp[2]}, Hd, Hd); }
        bias_mlp[0] = add_bias({d.off_mlp[1]}); bias_mlp[1] = add_bias({d.off_mlp[3]});
    }
    for (int t = 0; t < NT; ++t) if (p.need_dx[0][t]) { pack_enc[t] = add_packs(0, {p.off_enc_w[t]}, Hd, d.type_width[t]); bias_enc[t] = add_bias({p.off_enc_b[t]}); }
    p.n_img = (int)p.packs.size();

    // ---- job tables ---------------------------------------------------------------------------------------------------
    std::vector<int32_t> jobs, terms, srcs;
    auto add_src = [&](int buf, int node, int mask, float scale) { srcs.insert(srcs.end(), {buf, node, mask, fbits(scale)}); return (int)(srcs.size() / SRC_INTS) - 1; };
    struct TermDef { int pack, nkc, kind, width, sign; std::vector<std::array<int, 4>> s; };
    auto add_job = [&](int out_buf, int out_node, int flags, int bias, int res_buf, int res_node, int bits_buf, int gate_buf, int gate_node,
                       const std::vector<TermDef>& tds, int dhm_buf = 0) {
        const int t0 = (int)(terms.size() / TERM_INTS);
        for (const TermDef& td : tds) {
            const int s0 = (int)(srcs.size() / SRC_INTS);
            for (auto& s : td.s) srcs.insert(srcs.end(), {s[0], s[1], s[2], s[3]});
            terms.insert(terms.end(), {td.pack, td.nkc, td.kind, s0, (int)td.s.size(), td.width, td.sign, 0});
        }
        jobs.insert(jobs.end(), {out_buf, out_node, flags, bias, res_buf, res_node, bits_buf, gate_buf, gate_node, t0, (int)tds.size(), dhm_buf});
        return (int)(jobs.size() / JOB_INTS) - 1; };
    auto one = [&](int buf, int node, int mask = -1, float scale = 1.0f) { return std::array<int, 4>{buf, node, mask, fbits(scale)}; };
    double alg_fwd = 0, alg_bwd = 0, exec_fwd = 0, exec_bwd = 0;
    const double NL = 2.0 * Hd * Hd;
    auto stat = [&](const std::string& name, int bound, double fa, double by) {
        mshgnn_kernel_stat k{}; std::snprintf(k.name, sizeof(k.name), "%s", name.c_str());
        k.bound = bound; k.flops_per_window = fa; k.flops_exec_per_window = fa; k.bytes_per_window = by;
        p.kstats.push_back(k); return (int)p.kstats.size() - 1; };
    const double es = (double)p.esize * p.planes, in_es = p.split ? 4.0 : 2.0;
    double bytes_in = 0, bytes_in_live = 0;
    for (int t = 0; t < NT; ++t) {
        bytes_in += (double)d.type_nodes[t] * d.type_width[t] * in_es;
        for (int i = 0; i < d.type_nodes[t]; ++i) if (need_n[0][p.type_base[t] + i]) bytes_in_live += (double)d.type_width[t] * in_es;
    }
    p.ks_prep = stat("prep", MSHGNN_BOUND_HBM, 0, 0);
    std::vector<int32_t> aggs;
    int agg_mark = 0;      // first aggregate op of the launch being built
    auto launch = [&](std::vector<Launch>& v, int j0, const std::string& name, double flops, double bytes) {
        const int n = (int)(jobs.size() / JOB_INTS) - j0, na = (int)(aggs.size() / AGG_INTS);
        if (n > 0) {
            Launch ln{j0, n, stat(name, MSHGNN_BOUND_MFMA, flops, bytes)}; ln.agg0 = agg_mark; ln.n_agg = na - agg_mark;
            ln.all_plain = true;
            for (int j = j0; j < j0 + n && ln.all_plain; ++j) {
                const int32_t* jb = &jobs[(size_t)j * JOB_INTS];
                ln.max_terms = std::max(ln.max_terms, (int)jb[J_NTERMS]);
                for (int ti = jb[J_TERM0]; ti < jb[J_TERM0] + jb[J_NTERMS]; ++ti) {
                    const int32_t* tm = &terms[(size_t)ti * TERM_INTS];
                    if (tm[T_NSRC] >= 1 && srcs[(size_t)tm[T_SRC0] * SRC_INTS + S_MASK] >= 0) ln.any_mask = true;
                    if (tm[T_KIND] != 0 || tm[T_NSRC] != 1 || srcs[(size_t)tm[T_SRC0] * SRC_INTS + S_SCALE] != fbits(1.0f) || tm[T_NKC] != NCT) { ln.all_plain = false; break; }
                }
            }
            ln.all_raw = true;
            for (int j = j0; j < j0 + n && ln.all_raw; ++j) {
                const int32_t* jb = &jobs[(size_t)j * JOB_INTS];
                if (jb[J_NTERMS] > 16) ln.all_raw = false;
                for (int ti = jb[J_TERM0]; ti < jb[J_TERM0] + jb[J_NTERMS]; ++ti) {
                    const int32_t* tm = &terms[(size_t)ti * TERM_INTS];
                    if (tm[T_KIND] != 1 || tm[T_NSRC] != 1) { ln.all_raw = false; break; }
                }
            }
            v.push_back(ln);
        }
        agg_mark = na; };
    // a term's / item's source list -> itself, or (more than G_MANY rows) one plain row of the layer's aggregate buffer + the op that fills it
    std::map<std::vector<std::array<int, 4>>, std::array<int, 4>> agg_known;      // (forward aggregates are looked up again by the weight-gradient items)
    int agg_count[2][G_MAX_L] = {};
    // Wide plans (hidden a multiple of 512, both arithmetics; of 256 in bf16 arithmetic: the pipelined job kernel k_gstep5 wants every term to be ONE plain row): a sum of TWO rows at scale 1 becomes two
    // terms on the same weights (W (a + b) = W a + W b: one more pass of MFMAs for that term, and the sum is no longer rounded to bf16 before the product), longer
    // sums go through the aggregate buffers from three rows on.  MSHGNN_GEN_SPLIT_SUMS=0 keeps the sums (A/B runs).
    const bool split_sums = (NCT % 4 == 0 || (!p.split && NCT % 2 == 0)) && []() { const char* e = TUNE_ENV("MSHGNN_GEN_SPLIT_SUMS"); return !(e && std::atoi(e) == 0); }();
    const int g_many = [&]() { const char* e = std::getenv("MSHGNN_GEN_MANY"); return e ? std::max(2, std::atoi(e)) : (split_sums ? 2 : G_MANY); }();      // (threshold override, read when the plan is compiled: tests, measurements)
    int n_split_terms = 0;
    auto push_term = [&](std::vector<TermDef>& tds, const TermDef& td) {
        bool unit = td.kind == 0 && td.s.size() == 2;
        for (auto& q : td.s) if (q[3] != fbits(1.0f)) unit = false;
        if (split_sums && unit) { for (auto& q : td.s) { TermDef t1 = td; t1.s = {q}; tds.push_back(t1); } ++n_split_terms; }
        else tds.push_back(td); };
    auto many = [&](const std::vector<std::array<int, 4>>& srcl, int dir, int l, bool create) -> std::vector<std::array<int, 4>> {
        if ((int)srcl.size() <= g_many) return srcl;
        auto it = agg_known.find(srcl);
        if (it != agg_known.end()) return {it->second};
        if (!create) return srcl;
        const int node = agg_count[dir][l]++;
        p.n_aggbuf = std::max(p.n_aggbuf, node + 1);
        const int s0 = (int)(srcs.size() / SRC_INTS);
        for (auto& q : srcl) srcs.insert(srcs.end(), {q[0], q[1], q[2], q[3]});
        aggs.insert(aggs.end(), {(dir ? GBUF_AGGB : GBUF_AGGF) + l, node, s0, (int)srcl.size()});
        const std::array<int, 4> row{(dir ? GBUF_AGGB : GBUF_AGGF) + l, node, -1, fbits(1.0f)};
        agg_known[srcl] = row;
        return {row}; };

    {   // encoder: X_0[n] = relu((mask . x) W_enc^T + b)      (hgnn_c2.py:143-147)
        const int j0 = (int)(jobs.size() / JOB_INTS); double fl = 0;
        for (int t = 0; t < NT; ++t)
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                if (!need_n[0][p.type_base[t] + i]) continue;      // its X_0 feeds nothing that reaches the decoder
                TermDef td{pack_enc[t], p.enc_nkc[t], 1, d.type_width[t], p.sign_off[t] + i * p.enc_nkc[t] * TW, {std::array<int, 4>{t, i, -1, fbits(1.0f)}}};
                add_job(BUF_X + 0, p.type_base[t] + i, JF_BIAS | JF_RELU | JF_BITS_OUT, bias_enc[t], -1, 0, GBUF_MASK0, -1, 0, {td});
                fl += 2.0 * Hd * d.type_width[t];
            }
        alg_fwd += fl; exec_fwd += fl;
        launch(p.fwd, j0, "enc_fwd", fl, bytes_in + (double)p.NN * Hd * es);
    }
    for (int l = 0; l < L; ++l) {
        // HeteroConv layer l: one job per live destination node (hgnn_c2.py:150-166)
        int j0 = (int)(jobs.size() / JOB_INTS); double fl = 0;
        const int ns0 = n_split_terms;      // (a split sum runs its weights twice: executed, not algorithmic, work)
        const bool mlp_live = has_mlp && p.live[l][d.mlp_type];
        for (int t = 0; t < NT; ++t) {
            if (!p.live[l][t]) continue;
            const bool mlp = has_mlp && t == d.mlp_type;
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                const int n = p.type_base[t] + i;
                if (!live_n[l][n]) continue;
                std::vector<TermDef> tds;
                tds.push_back({pack_root[0][l * NT + t], NCT, 0, Hd, 0, {one(BUF_X + l, n)}});
                for (int r = 0; r < NR; ++r) {
                    if (p.rel_dst[r] != t || in_src[r][i].empty()) continue;
                    TermDef td{pack_rel[0][l * NR + r], NCT, 0, Hd, 0, {}};
                    for (int j : in_src[r][i]) td.s.push_back(one(BUF_X + l, p.type_base[p.rel_src[r]] + j, -1, scale_of(r, i)));
                    td.s = many(td.s, 0, l, true);
                    push_term(tds, td);
                }
                fl += NL * tds.size();
                if (mlp) add_job(BUF_HB + l, i, JF_BIAS, bias_layer[l * NT + t], -1, 0, -1, -1, 0, tds);      // H -> base_transform
                else add_job(BUF_X + l + 1, n, JF_BIAS | JF_RELU | JF_BITS_OUT | (residual ? JF_RES : 0), bias_layer[l * NT + t], BUF_X + l, n, BUF_MASK + l, -1, 0, tds);
            }
        }
        alg_fwd += fl - NL * (n_split_terms - ns0); exec_fwd += fl;
        launch(p.fwd, j0, "layer_fwd" + std::to_string(l), fl - NL * (n_split_terms - ns0), 2.0 * p.NN * Hd * es);
        if (mlp_live) {   // base_transform: Y = W2 relu(W1 H + b1) + b2, X <- Y + X    (hgnn_c2.py:117-121,156,161-166)
            j0 = (int)(jobs.size() / JOB_INTS);
            for (int u = 0; u < p.n_mlp; ++u)
                add_job(BUF_T1 + l, u, JF_BIAS | JF_RELU, bias_mlp[0], -1, 0, -1, -1, 0, {TermDef{pack_mlp[0][0], NCT, 0, Hd, 0, {one(BUF_HB + l, u)}}});
            launch(p.fwd, j0, "mlp1_fwd" + std::to_string(l), NL * p.n_mlp, 2.0 * p.n_mlp * Hd * es);
            j0 = (int)(jobs.size() / JOB_INTS);
            for (int u = 0; u < p.n_mlp; ++u) {
                const int n = p.type_base[d.mlp_type] + u;
                add_job(BUF_X + l + 1, n, JF_BIAS | (residual ? JF_RES : 0), bias_mlp[1], BUF_X + l, n, -1, -1, 0, {TermDef{pack_mlp[0][1], NCT, 0, Hd, 0, {one(BUF_T1 + l, u)}}});
            }
            launch(p.fwd, j0, "mlp2_fwd" + std::to_string(l), NL * p.n_mlp, 3.0 * p.n_mlp * Hd * es);
            alg_fwd += 2 * NL * p.n_mlp; exec_fwd += 2 * NL * p.n_mlp;
        }
    }
    p.ks_dec_fwd = stat("dec_fwd", MSHGNN_BOUND_HBM, 2.0 * n_out * d.out_channels * Hd, n_out * (double)Hd * es);
    p.ks_dec_bwd = stat("dec_bwd", MSHGNN_BOUND_HBM, 4.0 * n_out * d.out_channels * Hd, 2.0 * n_out * (double)Hd * es);
    alg_fwd += 2.0 * n_out * d.out_channels * Hd; exec_fwd += 2.0 * n_out * d.out_channels * Hd;
    alg_bwd += 4.0 * n_out * d.out_channels * Hd; exec_bwd += 4.0 * n_out * d.out_channels * Hd;
    // dH of node n in layer l as a staging source: relu types = dX_{l+1}[n] . relu bits of layer l; base_transform type = the dH stash
    // dH of a relu type: every reader (the backward jobs of the layer below: 1 + ~2 terms; the weight-gradient items: ~3) used to request the relu bytes and mask
    // the row as it staged it.  p.dhm: the job (or the decoder's backward) that PRODUCES dX_{l+1}[n] also writes dX_{l+1}[n] . bits_l[n] into layer l's dH stash (one
    // more row written per node and layer); the readers then see plain rows -- every backward launch qualifies for the unmasked k_gstep5, and the weight-gradient
    // streams carry no relu bytes.  The same values, bit for bit (masking zeroes elements; it does not round).  MSHGNN_GEN_DHM=0: the readers mask.
    p.dhm = []() { const char* e = TUNE_ENV("MSHGNN_GEN_DHM"); return !(e && std::atoi(e) == 0); }();
    auto dh_src = [&](int l, int n, float scale = 1.0f) {
        const int t = p.node_type[n];
        if (has_mlp && t == d.mlp_type) return one(BUF_DH + l, n, -1, scale);
        if (p.dhm) return one(BUF_DH + l, n, -1, scale);
        return one(BUF_DX + l + 1, n, BUF_MASK + l, scale); };
    for (int l = L - 1; l >= 0; --l) {
        const bool mlp_live = has_mlp && p.live[l][d.mlp_type];
        if (mlp_live) {   // dT1 = dY W2 ; dU = dT1 . (T1 > 0) ; dH = dU W1
            int j0 = (int)(jobs.size() / JOB_INTS);
            for (int u = 0; u < p.n_mlp; ++u)
                add_job(BUF_DU + l, u, JF_GATE_POS, 0, -1, 0, -1, BUF_T1 + l, u, {TermDef{pack_mlp[1][1], NCT, 0, Hd, 0, {one(BUF_DX + l + 1, p.type_base[d.mlp_type] + u)}}});
            launch(p.bwd, j0, "mlp2_bwd" + std::to_string(l), NL * p.n_mlp, 3.0 * p.n_mlp * Hd * es);
            j0 = (int)(jobs.size() / JOB_INTS);
            for (int u = 0; u < p.n_mlp; ++u)
                add_job(BUF_DH + l, p.type_base[d.mlp_type] + u, 0, 0, -1, 0, -1, -1, 0, {TermDef{pack_mlp[1][0], NCT, 0, Hd, 0, {one(BUF_DU + l, u)}}});
            launch(p.bwd, j0, "mlp1_bwd" + std::to_string(l), NL * p.n_mlp, 2.0 * p.n_mlp * Hd * es);
            alg_bwd += 2 * NL * p.n_mlp; exec_bwd += 2 * NL * p.n_mlp;
        }
        // dX_l[j] = (residual) + dH_j W_rootsum + sum_r W_rel^r-transposed Agg of dH over the out-edges of j; layer 0 also applies relu'(X_0)
        const int j0 = (int)(jobs.size() / JOB_INTS); double fl = 0;
        const int ns0 = n_split_terms;
        for (int s = 0; s < NT; ++s) {
            if (!p.need_dx[l][s]) continue;
            for (int j = 0; j < d.type_nodes[s]; ++j) {
                const int n = p.type_base[s] + j;
                if (!need_n[l][n]) continue;
                std::vector<TermDef> tds;
                if (live_n[l][n]) tds.push_back({pack_root[1][l * NT + s], NCT, 0, Hd, 0, {dh_src(l, n)}});
                for (int r = 0; r < NR; ++r) {
                    if (p.rel_src[r] != s || pack_rel[1][l * NR + r] < 0 || out_dst[r][j].empty()) continue;
                    TermDef td{pack_rel[1][l * NR + r], NCT, 0, Hd, 0, {}};
                    for (int i : out_dst[r][j]) if (live_n[l][p.type_base[p.rel_dst[r]] + i]) td.s.push_back(dh_src(l, p.type_base[p.rel_dst[r]] + i, scale_of(r, i)));
                    if (td.s.empty()) continue;
                    td.s = many(td.s, 1, l, true);
                    push_term(tds, td);
                }
                fl += NL * tds.size();
                const bool res = residual && live_n[l][n];
                // (p.dhm) this job's dX_l[n] is also the dH of layer l - 1, once masked with that layer's relu bits -- where n is computed there and is a relu type
                const bool wr_dh = p.dhm && l >= 1 && live_n[l - 1][n] && !(has_mlp && s == d.mlp_type);
                add_job(BUF_DX + l, n, (res ? JF_RES : 0) | (l == 0 ? JF_GATE_BITS : 0) | (wr_dh ? JF_DHM | (residual ? 0 : JF_DHM_ONLY) : 0), 0, BUF_DX + l + 1, n, wr_dh ? BUF_MASK + l - 1 : -1, GBUF_MASK0, n, tds,
                        wr_dh ? BUF_DH + l - 1 : 0);
            }
        }
        alg_bwd += fl - NL * (n_split_terms - ns0); exec_bwd += fl;
        launch(p.bwd, j0, "layer_bwd" + std::to_string(l), fl - NL * (n_split_terms - ns0), 3.0 * p.NN * Hd * es);
    }

    // ---- weight-gradient targets / items / units / finalize ops -----------------------------------------------------------
    std::vector<int32_t> units, items, fins;
    struct Tgt { int rows, K; std::vector<std::array<int, 6>> items; bool bias; int unit0 = 0, chunks = 0; };     // item: p_buf p_node p_mask kind src0 nsrc
    std::vector<Tgt> tgts;
    auto new_item = [&](Tgt& g, std::array<int, 4> psrc, int kind, const std::vector<std::array<int, 4>>& qs) {
        const int s0 = (int)(srcs.size() / SRC_INTS);
        for (auto& s : qs) srcs.insert(srcs.end(), {s[0], s[1], s[2], s[3]});
        g.items.push_back({psrc[0], psrc[1], psrc[2], kind, s0, (int)qs.size()}); };
    std::vector<int> tgt_root((size_t)L * NT, -1), tgt_rel((size_t)L * NR, -1), tgt_enc(NT, -1);
    int tgt_mlp[2] = {-1, -1};
    double gw_fl = 0;
    for (int l = 0; l < L; ++l) {
        for (int t = 0; t < NT; ++t) {
            if (!p.live[l][t]) continue;
            Tgt g{Hd, Hd, {}, true};
            for (int i = 0; i < d.type_nodes[t]; ++i) { const int n = p.type_base[t] + i; if (!live_n[l][n]) continue; new_item(g, dh_src(l, n), 0, {one(BUF_X + l, n)}); gw_fl += NL; }
            tgt_root[l * NT + t] = (int)tgts.size(); tgts.push_back(g);
        }
        for (int r = 0; r < NR; ++r) {
            if (pack_rel[0][l * NR + r] < 0) continue;
            Tgt g{Hd, Hd, {}, false};
            const int t = p.rel_dst[r];
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                if (in_src[r][i].empty() || !live_n[l][p.type_base[t] + i]) continue;
                std::vector<std::array<int, 4>> qs;
                for (int j : in_src[r][i]) qs.push_back(one(BUF_X + l, p.type_base[p.rel_src[r]] + j, -1, scale_of(r, i)));
                new_item(g, dh_src(l, p.type_base[t] + i), 0, many(qs, 0, l, false)); gw_fl += NL;      // (the forward pass left the sum of many rows in its aggregate buffer)
            }
            tgt_rel[l * NR + r] = (int)tgts.size(); tgts.push_back(g);
        }
    }
    if (has_mlp) {
        Tgt g1{Hd, Hd, {}, true}, g2{Hd, Hd, {}, true};
        for (int l = 0; l < L; ++l) {
            if (!p.live[l][d.mlp_type]) continue;
            for (int u = 0; u < p.n_mlp; ++u) {
                new_item(g1, one(BUF_DU + l, u), 0, {one(BUF_HB + l, u)});
                new_item(g2, one(BUF_DX + l + 1, p.type_base[d.mlp_type] + u), 0, {one(BUF_T1 + l, u)});
                gw_fl += 2 * NL;
            }
        }
        if (!g1.items.empty()) { tgt_mlp[0] = (int)tgts.size(); tgts.push_back(g1); tgt_mlp[1] = (int)tgts.size(); tgts.push_back(g2); }
    }
    for (int t = 0; t < NT; ++t) {
        if (!p.need_dx[0][t]) continue;
        Tgt g{Hd, d.type_width[t], {}, true};
        for (int i = 0; i < d.type_nodes[t]; ++i) {
            if (!need_n[0][p.type_base[t] + i]) continue;
            new_item(g, one(BUF_DX + 0, p.type_base[t] + i), 1, {std::array<int, 4>{t, i, p.sign_off[t] + i * p.enc_nkc[t] * TW, 0}});
            gw_fl += 2.0 * Hd * d.type_width[t];
        }
        tgt_enc[t] = (int)tgts.size(); tgts.push_back(g);
    }
    alg_bwd += gw_fl; exec_bwd += gw_fl;
    auto item_plain = [&](const std::array<int, 6>& it) {      // one plain activation row at scale 1, or (bf16 arithmetic) the sum of two such rows
        if (it[3] != 0 || it[5] > (p.split ? 1 : 2)) return false;
        for (int k = 0; k < it[5]; ++k) {
            const int32_t* sp = &srcs[(size_t)(it[4] + k) * SRC_INTS];
            if (sp[S_SCALE] != fbits(1.0f) || sp[S_MASK] >= 0) return false;
        }
        return true; };
    int ipu = G_ITEMS_PER_UNIT;      // items per weight-gradient unit
    if (const char* e = TUNE_ENV("MSHGNN_GGW_ITEMS")) ipu = std::max(1, std::atoi(e));      // (measurements)
    // units: for every target tile (ot, kt) one unit per chunk of <= G_ITEMS_PER_UNIT items; the units of one tile are consecutive
    for (Tgt& g : tgts) {
        g.unit0 = (int)(units.size() / UNIT_INTS);
        const int i0 = (int)(items.size() / GITEM_INTS);
        // items with one plain activation source at scale 1 first: the weight-gradient kernel walks super-units that hold only such items with its lean
        // streams (a fixed order either way: the sums stay deterministic)
        std::stable_sort(g.items.begin(), g.items.end(), [&](const std::array<int, 6>& x, const std::array<int, 6>& y) { return item_plain(x) > item_plain(y); });
        for (auto& it : g.items) items.insert(items.end(), {it[0], it[1], it[2], it[3], it[4], it[5], 0, 0});
        const int n = (int)g.items.size();
        g.chunks = (n + ipu - 1) / ipu;
        const int nkt = (g.K + TW - 1) / TW;
        for (int ot = 0; ot < g.rows / TW; ++ot)
            for (int kt = 0; kt < nkt; ++kt)
                for (int c = 0; c < g.chunks; ++c)
                    units.insert(units.end(), {i0 + c * ipu, i0 + std::min(n, (c + 1) * ipu), ot * TW, kt * TW, std::min(TW, g.K - kt * TW),
                                               (g.bias && kt == 0) ? 1 : 0, 0, 0});
    }
    p.n_units = (int)(units.size() / UNIT_INTS);
    // SUPER-UNITS = what one weight-gradient workgroup computes: OS x 2 adjacent 128x128 tiles of a target over one chunk of its items (OS = 2
    // o-tiles with 16 waves on the bf16 plan, 1 with 8 waves on the split plan).  The P rows are staged once for the 2 k-tiles and the Q rows
    // once for the OS o-tiles (a quarter / three eighths of the operand traffic of one workgroup per tile); every 128x128 sub-tile keeps its own
    // slab (the unit's), so the finalize tables do not change.  super-unit: unit[o][k] x4 (-1: absent) | item_begin item_end p_col0 q_col0 | q_ncols pad..
    std::vector<int32_t> sunits;
#ifndef GGW_SPLIT_OS2
#define GGW_SPLIT_OS2 1      // split arithmetic on 256 x 256 super-units too, with 8 waves of 128 x 64 (253 registers, none spilled): 2.83 ms against 3.25 ms for 128 x 256 at h=512
#endif                       // (the 16-wave form of that: 128 registers with 164 B of scratch, 10.1 ms against 5.8 ms)
    p.su_os = ((!p.split || GGW_SPLIT_OS2) && NCT % 2 == 0) ? 2 : 1;
    for (Tgt& g : tgts) {
        const int n = (int)g.items.size(), nkt = (g.K + TW - 1) / TW, not_ = g.rows / TW, i0 = units[(size_t)g.unit0 * UNIT_INTS + U_ITEM0];
        for (int c = 0; c < g.chunks; ++c)
            for (int ot = 0; ot < not_; ot += p.su_os)
                for (int kt = 0; kt < nkt; kt += 2) {
                    int u[4] = {-1, -1, -1, -1};
                    for (int a = 0; a < p.su_os; ++a) for (int b = 0; b < 2; ++b)
                        if (ot + a < not_ && kt + b < nkt) u[a * 2 + b] = g.unit0 + ((ot + a) * nkt + kt + b) * g.chunks + c;
                    int flags = 1;
                    for (int i = c * ipu; i < std::min(n, (c + 1) * ipu); ++i) if (!item_plain(g.items[i])) flags = 0;
                    if (g.bias && kt == 0) flags |= 2;
                    sunits.insert(sunits.end(), {u[0], u[1], u[2], u[3], i0 + c * ipu, i0 + std::min(n, (c + 1) * ipu), ot * TW, kt * TW,
                                                 std::min(2 * TW, g.K - kt * TW), flags, 0, 0});
                }
    }
    p.n_sunits = (int)(sunits.size() / SUNIT_INTS);
    p.n_parts = std::max(1, std::min(16, 768 / std::max(1, p.n_sunits)));
    if (const char* e = TUNE_ENV("MSHGNN_GGW_PARTS")) p.n_parts = std::max(1, std::min(16, std::atoi(e)));      // (measurements)
    // launch order: the super-units of ONE (target, item chunk) read the same P and Q rows (each half of them twice at hidden = 512).  Workgroups b
    // and b + 8 run on the same XCD (round-robin dispatch), so they are placed 8 apart on one XCD, back to back, and the second reader hits that
    // XCD's L2 (speed only).  Super-units were emitted (target, chunk)-major, so a group is a run of consecutive indices.
    // The super-units on the general streams (aggregates of many rows, raw inputs: SU_FLAGS bit 0 clear) take about twice as long per step as the lean ones:
    // they are dispatched first, so that they do not form the tail of the launch.
    // Placement (round 4): a group goes to the XCD whose groups already read most of ITS rows -- the root-weight, relation and base_transform targets
    // of one layer want the same dH / X rows (a joint's row is the P operand of three to four targets and the Q operand of as many), and every
    // workgroup of the launch sweeps the batch at the same pace, so readers that sit on one XCD and start together meet in its L2 (the 32-limb model
    // moved 5.0 GB for 1.5 GB of distinct rows with the shortest-queue placement of round 2).  Greedy in target order: overlap in row-streams
    // (buffer, node) with the queue's groups first, queue length second, lengths kept within a slack of the balanced one; each queue is then ordered
    // by its groups' first P row, so that the groups of one layer and limb range are dispatched side by side.  Deterministic; speed only.
    std::vector<int32_t> su_order;
    const bool su_locality = []() { const char* e = TUNE_ENV("MSHGNN_GGW_PLACE"); return !(e && std::atoi(e) == 0); }();      // (=0: round 2's placement, A/B runs)
    for (int cls = 0; cls <= 1; ++cls) {
        struct Grp { int first, count; std::vector<int64_t> streams; int64_t key; int steps; };      // steps: items of the group's super-units (their length)
        std::vector<Grp> groups;
        int pos = 0;
        for (Tgt& g : tgts) {
            const int per = ((g.rows / TW + p.su_os - 1) / p.su_os) * (((g.K + TW - 1) / TW + 1) / 2);
            for (int c = 0; c < g.chunks; ++c) {
                if ((sunits[(size_t)pos * SUNIT_INTS + SU_FLAGS] & 1) == cls) {
                    Grp gr{pos, per, {}, INT64_MAX, 0};
                    const int ia = sunits[(size_t)pos * SUNIT_INTS + SU_ITEM0], ib = sunits[(size_t)pos * SUNIT_INTS + SU_ITEM1];
                    gr.steps = ib - ia;
                    for (int it = ia; it < ib; ++it) {
                        const int32_t* im = &items[(size_t)it * GITEM_INTS];
                        const int64_t ps = ((int64_t)im[I_PBUF] << 20) | (int64_t)im[I_PNODE];
                        gr.streams.push_back(ps); gr.key = std::min(gr.key, ps);
                        if (im[I_KIND] == 0)
                            for (int k = 0; k < im[I_NSRC]; ++k) {
                                const int32_t* sp = &srcs[(size_t)(im[I_SRC0] + k) * SRC_INTS];
                                gr.streams.push_back(((int64_t)sp[S_BUF] << 20) | (int64_t)sp[S_NODE]);
                            }
                    }
                    std::sort(gr.streams.begin(), gr.streams.end()); gr.streams.erase(std::unique(gr.streams.begin(), gr.streams.end()), gr.streams.end());
                    groups.push_back(std::move(gr));
                }
                pos += per;
            }
        }
        std::vector<std::vector<int>> xq(8);      // queues of group indices
        std::vector<int> qlen(8, 0);
        std::vector<std::map<int64_t, int>> seen(8);
        int total = 0; for (auto& gr : groups) total += gr.count;
        const int cap = (total + 7) / 8 + (su_locality ? std::max(4, total / 64) : 0);
        for (int gi = 0; gi < (int)groups.size(); ++gi) {
            const Grp& gr = groups[gi];
            int best = -1, best_ov = -1;
            for (int x = 0; x < 8; ++x) {
                if (su_locality && qlen[x] + gr.count > cap) continue;
                int ov = 0;
                if (su_locality) for (int64_t sid : gr.streams) if (seen[x].count(sid)) ++ov;
                if (best < 0 || ov > best_ov || (ov == best_ov && qlen[x] < qlen[best])) { best = x; best_ov = ov; }
            }
            if (best < 0) { best = 0; for (int x = 1; x < 8; ++x) if (qlen[x] < qlen[best]) best = x; }
            xq[best].push_back(gi); qlen[best] += gr.count;
            for (int64_t sid : gr.streams) ++seen[best][sid];
        }
        std::vector<std::vector<int>> xs(8);      // queues of super-units
        for (int x = 0; x < 8; ++x) {
            // longest groups first (the launch is a list schedule of ~3 rounds of workgroups: with the short ones last its tail is a short workgroup, not a long one);
            // groups of one length by their first P row, so that the groups of one layer and limb range are still dispatched side by side
            static const bool lpt = []() { const char* e = TUNE_ENV("MSHGNN_GGW_LPT"); return !(e && std::atoi(e) == 0); }();
            if (su_locality) std::stable_sort(xq[x].begin(), xq[x].end(), [&](int a, int b) {
                if (lpt && groups[a].steps != groups[b].steps) return groups[a].steps > groups[b].steps;
                return groups[a].key < groups[b].key; });
            for (int gi : xq[x]) for (int k = 0; k < groups[gi].count; ++k) xs[x].push_back(groups[gi].first + k);
        }
        std::vector<size_t> qpos(8, 0);
        for (int b = 0; b < total; ++b) {      // block b -> queue b % 8; an exhausted queue borrows from the longest remaining one
            int q = b % 8;
            if (qpos[q] >= xs[q].size()) { q = 0; for (int y = 1; y < 8; ++y) if (xs[y].size() - qpos[y] > xs[q].size() - qpos[q]) q = y; }
            su_order.push_back(xs[q][qpos[q]++]);
        }
    }
    auto add_fin = [&](int64_t dst, int rows, int cols, int ld, int kind, int unit0, int nunits, int row0) {
        fins.insert(fins.end(), {(int32_t)(dst & 0xffffffff), (int32_t)(dst >> 32), rows, cols, ld, kind, unit0, nunits, row0, 0, 0, 0}); };
    auto fin_matrix = [&](int64_t dst, int tg, int K) {      // one op per 128x128 tile of the destination matrix [Hd x K]
        const int nkt = (K + TW - 1) / TW;
        for (int ot = 0; ot < NCT; ++ot)
            for (int kt = 0; kt < nkt; ++kt) {
                const int cols = std::min(TW, K - kt * TW);
                if (tg < 0) add_fin(dst + (int64_t)ot * TW * K + kt * TW, TW, cols, K, FIN_ZERO, 0, 0, 0);
                else add_fin(dst + (int64_t)ot * TW * K + kt * TW, TW, cols, K, FIN_MATRIX, tgts[tg].unit0 + (ot * nkt + kt) * tgts[tg].chunks, tgts[tg].chunks, 0);
            } };
    auto fin_bias = [&](int64_t dst, int tg, int K) {        // bias sums ride in the slabs of the kt == 0 units
        const int nkt = (K + TW - 1) / TW;
        for (int ot = 0; ot < NCT; ++ot) {
            if (tg < 0) add_fin(dst + ot * TW, 1, TW, TW, FIN_ZERO, 0, 0, 0);
            else add_fin(dst + ot * TW, 1, TW, TW, FIN_BIAS, tgts[tg].unit0 + (ot * nkt) * tgts[tg].chunks, tgts[tg].chunks, 0);
        } };
    for (int t = 0; t < NT; ++t) { fin_matrix(p.off_enc_w[t], tgt_enc[t], d.type_width[t]); fin_bias(p.off_enc_b[t], tgt_enc[t], d.type_width[t]); }
    for (int l = 0; l < L; ++l)
        for (int r = 0; r < NR; ++r) {
            const int tr = tgt_rel[l * NR + r], to = tgt_root[l * NT + p.rel_dst[r]];
            fin_matrix(p.off_rel_w[l * NR + r], tr, Hd); fin_bias(p.off_rel_b[l * NR + r], to, Hd); fin_matrix(p.off_root_w[l * NR + r], to, Hd);
        }
    if (has_mlp) for (int k = 0; k < 2; ++k) { fin_matrix(d.off_mlp[2 * k], tgt_mlp[k], Hd); fin_bias(d.off_mlp[2 * k + 1], tgt_mlp[k], Hd); }
    for (int dd = 0; dd < d.out_channels; ++dd)      // decoder partials live in their own slabs ([8][Hd] + bias[8] + loss): one op per 128-column piece
        for (int cg = 0; cg < Hd; cg += TW) add_fin(d.off_dec_w + (int64_t)dd * Hd + cg, 1, TW, TW, FIN_DEC_W, 0, 0, dd * Hd + cg);
    add_fin(d.off_dec_b, 1, d.out_channels, d.out_channels, FIN_DEC_B, 0, 0, 8 * Hd);
    {   // alignment gaps of the flat buffer are zeroed, so grad_params really is fully overwritten
        std::vector<std::pair<int64_t, int64_t>> spans;
        for (int t = 0; t < NT; ++t) { spans.push_back({p.off_enc_w[t], (int64_t)Hd * d.type_width[t]}); spans.push_back({p.off_enc_b[t], Hd}); }
        for (int i = 0; i < L * NR; ++i) { spans.push_back({p.off_rel_w[i], (int64_t)Hd * Hd}); spans.push_back({p.off_rel_b[i], Hd}); spans.push_back({p.off_root_w[i], (int64_t)Hd * Hd}); }
        if (has_mlp) for (int k = 0; k < 2; ++k) { spans.push_back({d.off_mlp[2 * k], (int64_t)Hd * Hd}); spans.push_back({d.off_mlp[2 * k + 1], Hd}); }
        spans.push_back({d.off_dec_w, (int64_t)d.out_channels * Hd}); spans.push_back({d.off_dec_b, d.out_channels});
        std::sort(spans.begin(), spans.end());
        int64_t pos = 0;
        for (auto& sp : spans) {
            if (sp.first > pos && sp.first - pos < 4096) add_fin(pos, 1, (int)(sp.first - pos), (int)(sp.first - pos), FIN_ZERO, 0, 0, 0);
            pos = std::max(pos, sp.first + sp.second);
        }
        if (d.n_flat > pos && d.n_flat - pos < 4096) add_fin(pos, 1, (int)(d.n_flat - pos), (int)(d.n_flat - pos), FIN_ZERO, 0, 0, 0);
    }
    p.n_fin = (int)(fins.size() / GFIN_INTS);
    p.ks_gradw = stat("gradw", MSHGNN_BOUND_MFMA, gw_fl, bytes_in + (2.0 * L + 1) * p.NN * Hd * es);
    p.ks_fin = stat("finalize", MSHGNN_BOUND_HBM, 0, 0);
    if (!aggs.empty()) p.ks_agg = stat("aggregate", MSHGNN_BOUND_HBM, 0, 0);      // (k_gagg, in front of the launches that consume aggregates of many rows)

    std::vector<int32_t>& T = p.tables;
    p.job_off = 0; T.insert(T.end(), jobs.begin(), jobs.end());
    p.term_off = (int)T.size(); T.insert(T.end(), terms.begin(), terms.end());
    p.src_off = (int)T.size(); T.insert(T.end(), srcs.begin(), srcs.end());
    p.unit_off = (int)T.size(); T.insert(T.end(), units.begin(), units.end());
    p.sunit_off = (int)T.size(); T.insert(T.end(), sunits.begin(), sunits.end());
    p.su_order_off = (int)T.size(); T.insert(T.end(), su_order.begin(), su_order.end());
    p.item_off = (int)T.size(); T.insert(T.end(), items.begin(), items.end());
    p.fin_off = (int)T.size(); T.insert(T.end(), fins.begin(), fins.end());
    p.agg_off = (int)T.size(); T.insert(T.end(), aggs.begin(), aggs.end());

    p.info.rows_per_tile = 64; p.info.total_nodes = p.NN; p.info.lds_bytes = 4 * 4096 * p.planes;
    p.info.flops_fwd = alg_fwd; p.info.flops_bwd = alg_bwd; p.info.flops_exec_fwd = exec_fwd; p.info.flops_exec_bwd = exec_bwd;
    p.info.bytes_in = bytes_in; p.info.bytes_in_live = bytes_in_live; p.info.n_gradw_workgroups = p.n_sunits * p.n_parts;
    p.info.n_launches_fwd = 2 + (int)p.fwd.size(); p.info.n_launches_bwd = 3 + (int)p.bwd.size();
    p.info.kernel_sets = 4;      // bit 2: generic-width engine
    p.info.grad_split = -1;
    return true;
}

inline void layout_gen_workspace(const GenPlan& p, int64_t B, int training, mshgnn_ws_layout* o) {
    std::memset(o, 0, sizeof(*o));
    size_t off = 0;
    const size_t act = (size_t)B * p.NN * p.Hd * p.esize * p.planes, mlp = (size_t)B * std::max(1, p.n_mlp) * p.Hd * p.esize * p.planes;
    const size_t maskb = (size_t)((B + 15) / 16 * 16) * p.NN * (p.Hd / 32) * 4;      // one byte per (node, window, 8 features)
    auto take = [&](size_t bytes) { size_t r = off; off = align_up(off + bytes, 256); return r; };
    for (int l = 0; l <= p.L; ++l) o->x[l] = take(act);
    for (int l = 0; l < p.L; ++l) { o->mask[l] = take(maskb); o->hb[l] = take(mlp); o->t1[l] = take(mlp); }
    o->dd[0] = take(maskb);
    const size_t aggsz = align_up((size_t)B * p.n_aggbuf * p.Hd * p.esize * p.planes, 256);      // one aggregate buffer (a layer's, one direction)
    if (p.n_aggbuf) o->dd[1] = take(aggsz * p.L);      // dd[1] / dd[2]: forward / backward aggregates of many rows, L buffers of [n_aggbuf][B][h] each
    if (training) {
        for (int l = 0; l <= p.L; ++l) o->dx[l] = take(act);
        for (int l = 0; l < p.L; ++l) { o->dh[l] = take(act); o->du[l] = take(mlp); }
        if (p.n_aggbuf) o->dd[2] = take(aggsz * p.L);
        o->slabs = take((size_t)p.n_units * p.n_parts * SLAB_FLOATS * 4);
        o->dec_slabs = take((size_t)NWG_DEC * (8 * p.Hd + 16) * 4);
    }
    o->wpack = take(p.packs.size() * (size_t)TW * TW * p.esize * p.planes);
    o->bias = take(p.biases.size() * (size_t)TW * 4);
    o->loss = take(64);
    o->total = off;
}

}  // namespace gen
}  // namespace mshgnn
