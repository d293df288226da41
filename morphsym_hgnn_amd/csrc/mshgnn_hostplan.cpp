// Host-only build of the plan compilers (g++, no HIP): libmshgnn_hostplan.so.
//   * tools/gen_spec_tables.py dumps the slab tables of the BASELINE topologies through it -> mshgnn_spec_tables.inc (the compile-time programs of the
//     specialised step kernels);
//   * tests/test_plan_property.py drives it with hypothesis-generated topologies, also as an AddressSanitizer / UBSan build (`make asan-host`): the
//     index tables every kernel trusts are bounds-checked here, on the CPU.
// Not part of the drop-in boundary (include/mshgnn.h): test / build infrastructure around mshgnn_plan.hpp and mshgnn_gen_plan.hpp.
#include "mshgnn_plan.hpp"
#include "mshgnn_gen_plan.hpp"

using namespace mshgnn;

namespace {
thread_local std::string g_hp_err;

// every index a kernel reads out of the fused / slab tables must stay inside what it addresses; returns "" or the first violation
std::string check_tables(const HostPlan& p) {
    char buf[256];
    auto bad = [&](const char* what, int l, int v, int lim) { snprintf(buf, sizeof buf, "%s: layer %d value %d limit %d", what, l, v, lim); return std::string(buf); };
    const int n_packs = (int)p.packs.size() * (p.split ? 2 : 1), n_bias = (int)p.biases.size();
    auto check_prog = [&](int off, int hs, int cb, int nblk, const char* what, int l) -> std::string {
        const int32_t* pr = p.tables.data() + off;
        if (off < 0 || off + FPROG_LEN > (int)p.tables.size()) return bad("program offset", l, off, (int)p.tables.size());
        auto at = [&](int i) { return (pr[128 + (i >> 2)] >> ((i & 3) << 3)) & 0xff; };
        const int nseg = at(0);
        if (nseg > 64) return bad("segments", l, nseg, 64);
        int pb = 1;
        for (int s = 0; s < nseg; ++s) {
            if (pr[s] < 0 || pr[s] >= n_packs) return bad(what, l, pr[s], n_packs);
            for (int u = 0; u < hs; ++u) pb += (pr[64 + s] >> (cb * u)) & ((1 << cb) - 1);
            if (hs * cb < 32 && (pr[64 + s] >> (hs * cb)) != 0) return bad("count bits beyond the slots", l, pr[64 + s], 0);
        }
        if (pb + 1 > 256) return bad("block stream length", l, pb + 1, 256);
        for (int i = 1; i <= pb; ++i) if (at(i) >= nblk) return bad("source block", l, at(i), nblk);
        return "";
    };
    auto check_hdr = [&](int off, int l, bool slab) -> std::string {
        if (off < 0 || off + FH_SIZE > (int)p.tables.size()) return bad("header offset", l, off, (int)p.tables.size());
        const int32_t* h = p.tables.data() + off;
        if (h[FH_NMLP] < 0 || h[FH_NMLP] > 4) return bad("nmlp", l, h[FH_NMLP], 4);
        if (h[FH_NMLP] > 0) for (int k : {FH_W1, FH_W2}) if (h[k] < 0 || h[k] >= n_packs) return bad("mlp pack", l, h[k], n_packs);
        for (int q = 0; q < FS_MAXN; ++q) {
            if (h[FH_KIND + q] < NK_DEAD || h[FH_KIND + q] > NK_MLP) return bad("node kind", l, h[FH_KIND + q], NK_MLP);
            if (h[FH_KIND + q] != NK_DEAD && (h[FH_BIAS + q] < 0 || h[FH_BIAS + q] >= std::max(n_bias, 1))) return bad("bias row", l, h[FH_BIAS + q], n_bias);
        }
        if (slab) for (int u = 0; u < 16; ++u) for (int arr : {FH_SLOTA, FH_SLOTB}) if (h[arr + u] < -1 || h[arr + u] >= p.NN) return bad("slot node", l, h[arr + u], p.NN);
        return "";
    };
    std::string e;
    if (p.fused) for (int l = 0; l < p.L; ++l) for (int dir = 0; dir < 2; ++dir) {
        const int off = dir ? p.fs_bwd_off[l] : p.fs_fwd_off[l];
        if (!(e = check_hdr(off, l, false)).empty()) return "fused " + e;
        for (int half = 0; half < 2; ++half)
            if (!(e = check_prog(off + FH_SIZE + half * FPROG_LEN, FS_HS, 3, p.split ? 2 * p.lo_blk + 8 : p.fs_blk, "fused pack", l)).empty()) return "fused " + e;
    }
    if (p.slab) for (int l = 0; l < p.L; ++l) for (int dir = 0; dir < 2; ++dir) {
        const int off = dir ? p.sl_bwd_off[l] : p.sl_fwd_off[l];
        if (!(e = check_hdr(off, l, true)).empty()) return "slab " + e;
        if (!(e = check_prog(off + FH_SIZE, SL_HA, SL_CBA, p.sl_blk, "slab pack A", l)).empty()) return "slab " + e;
        if (!(e = check_prog(off + FH_SIZE + FPROG_LEN, p.sl_hb, SL_CBB, p.sl_blk, "slab pack B", l)).empty()) return "slab " + e;
    }
    // weight-gradient items / lanes / finalize ops
    if (p.item_off < 0 || p.item_off + p.n_items * ITEM_INTS > (int)p.tables.size()) return bad("item table", 0, p.item_off + p.n_items * ITEM_INTS, (int)p.tables.size());
    if (p.lane_off < 0 || p.lane_off + p.n_lanes * LANE_INTS > (int)p.tables.size()) return bad("lane table", 0, p.lane_off + p.n_lanes * LANE_INTS, (int)p.tables.size());
    for (int i = 0; i < p.n_lanes; ++i) {
        const int32_t* ln = p.tables.data() + p.lane_off + i * LANE_INTS;
        if (ln[0] < 0 || ln[1] < ln[0] || ln[1] > p.n_items) return bad("lane item range", i, ln[1], p.n_items);
        if (ln[2] < 0 || ln[2] >= std::max(p.n_targets, 1)) return bad("lane target", i, ln[2], p.n_targets);
    }
    for (int i = 0; i < p.n_items; ++i) {
        const int32_t* it = p.tables.data() + p.item_off + i * ITEM_INTS;
        if (it[0] < 0 || it[0] >= BUF_COUNT) return bad("item P buffer", i, it[0], BUF_COUNT);
        if (it[2] < 0 || it[2] >= 64) return bad("item P node", i, it[2], 64);
        if (it[3] < 0 || it[3] >= BUF_COUNT) return bad("item Q buffer", i, it[3], BUF_COUNT);
    }
    if (p.fin_off < 0 || p.fin_off + p.n_fin * FIN_INTS > (int)p.tables.size()) return bad("finalize table", 0, p.fin_off + p.n_fin * FIN_INTS, (int)p.tables.size());
    for (int i = 0; i < p.n_fin; ++i) {
        const int32_t* f = p.tables.data() + p.fin_off + i * FIN_INTS;
        const int64_t lo = ((int64_t)(uint32_t)f[0]) | ((int64_t)f[1] << 32);
        if (f[6] < FIN_MATRIX || f[6] > FIN_DEC_B) return bad("finalize kind", i, f[6], FIN_DEC_B);
        if (lo < 0 || lo >= std::max<int64_t>(p.d.n_flat, 1)) return bad("finalize destination", i, (int)lo, (int)p.d.n_flat);
    }
    return "";
}
// generic-width engine: jobs -> terms -> sources, weight-gradient units -> items -> sources, aggregates; every range inside its table
std::string check_gen_tables(const gen::GenPlan& p) {
    using namespace gen;
    char buf[256];
    auto bad = [&](const char* what, int i, long long v, long long lim) { snprintf(buf, sizeof buf, "%s: entry %d value %lld limit %lld", what, i, v, lim); return std::string(buf); };
    const int32_t* T = p.tables.data();
    const int n_jobs = (p.term_off - p.job_off) / JOB_INTS, n_terms = (p.src_off - p.term_off) / TERM_INTS, n_srcs = (p.unit_off - p.src_off) / SRC_INTS;
    const int n_units = (p.sunit_off - p.unit_off) / UNIT_INTS, n_sunits = (p.su_order_off - p.sunit_off) / SUNIT_INTS, n_items = (p.fin_off - p.item_off) / GITEM_INTS;
    const int n_packs = (int)p.packs.size() * (p.split ? 2 : 1), n_bias = (int)p.biases.size();
    if (!(p.job_off <= p.term_off && p.term_off <= p.src_off && p.src_off <= p.unit_off && p.unit_off <= p.sunit_off && p.sunit_off <= p.su_order_off &&
          p.su_order_off <= p.item_off && p.item_off <= p.fin_off && p.fin_off <= p.agg_off && p.agg_off <= (int)p.tables.size())) return "table offsets out of order";
    for (const auto* ls : {&p.fwd, &p.bwd}) for (const Launch& ln : *ls) {
        if (ln.job0 < 0 || ln.n_jobs < 0 || ln.job0 + ln.n_jobs > n_jobs) return bad("launch job range", ln.job0, ln.job0 + ln.n_jobs, n_jobs);
    }
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t* jb = T + p.job_off + j * JOB_INTS;
        if (jb[J_TERM0] < 0 || jb[J_NTERMS] < 0 || jb[J_TERM0] + jb[J_NTERMS] > n_terms) return bad("job term range", j, jb[J_TERM0] + jb[J_NTERMS], n_terms);
        if (jb[J_OUT_BUF] < 0 || jb[J_OUT_BUF] >= GBUF_COUNT) return bad("job out buffer", j, jb[J_OUT_BUF], GBUF_COUNT);
        if (jb[J_OUT_NODE] < 0 || jb[J_OUT_NODE] >= std::max(p.NN, p.n_aggbuf)) return bad("job out node", j, jb[J_OUT_NODE], std::max(p.NN, p.n_aggbuf));
        if ((jb[J_FLAGS] & JF_BIAS) && (jb[J_BIAS] < 0 || jb[J_BIAS] >= std::max(n_bias, 1))) return bad("job bias row", j, jb[J_BIAS], n_bias);
    }
    for (int t = 0; t < n_terms; ++t) {
        const int32_t* tm = T + p.term_off + t * TERM_INTS;
        if (tm[T_SRC0] < 0 || tm[T_NSRC] < 1 || tm[T_SRC0] + tm[T_NSRC] > n_srcs) return bad("term source range", t, tm[T_SRC0] + tm[T_NSRC], n_srcs);
        if (tm[T_NKC] < 1) return bad("term K chunks", t, tm[T_NKC], 1);
        if (tm[T_PACK] < 0 || tm[T_PACK] + tm[T_NKC] * p.NCT > std::max(n_packs, 1)) return bad("term pack range", t, tm[T_PACK] + (long long)tm[T_NKC] * p.NCT, n_packs);
    }
    // a source row is read as an activation row (buffer id, node, optional relu-byte buffer) or, from a raw-input term / item (kind 1), as an input row
    // (node type, node inside the type, offset of its sign bytes)
    auto check_srcs = [&](int s0, int ns, int kind, const char* who, int idx) -> std::string {
        for (int i = s0; i < s0 + ns; ++i) {
            const int32_t* sr = T + p.src_off + i * SRC_INTS;
            if (kind == 1) {
                if (sr[S_BUF] < 0 || sr[S_BUF] >= p.NT) return bad("raw source type", idx, sr[S_BUF], p.NT);
                if (sr[S_NODE] < 0 || sr[S_NODE] >= p.d.type_nodes[sr[S_BUF]]) return bad("raw source node", idx, sr[S_NODE], p.d.type_nodes[sr[S_BUF]]);
                if (sr[S_MASK] < -1 || sr[S_MASK] >= (int)p.signs.size()) return bad("raw source sign offset", idx, sr[S_MASK], (long long)p.signs.size());
            } else {
                if (sr[S_BUF] < 0 || sr[S_BUF] >= GBUF_COUNT) return bad("source buffer", idx, sr[S_BUF], GBUF_COUNT);
                if (sr[S_NODE] < 0 || sr[S_NODE] >= std::max(p.NN, p.n_aggbuf)) return bad("source node", idx, sr[S_NODE], std::max(p.NN, p.n_aggbuf));
                if (sr[S_MASK] < -1 || sr[S_MASK] >= GBUF_COUNT) return bad("source mask buffer", idx, sr[S_MASK], GBUF_COUNT);
            }
        }
        (void)who; return "";
    };
    for (int t = 0; t < n_terms; ++t) {
        const int32_t* tm = T + p.term_off + t * TERM_INTS;
        const std::string e = check_srcs(tm[T_SRC0], tm[T_NSRC], tm[T_KIND], "term", t);
        if (!e.empty()) return e;
    }
    for (int u = 0; u < n_units; ++u) {
        const int32_t* un = T + p.unit_off + u * UNIT_INTS;
        if (un[U_ITEM0] < 0 || un[U_ITEM1] < un[U_ITEM0] || un[U_ITEM1] > n_items) return bad("unit item range", u, un[U_ITEM1], n_items);
    }
    for (int u = 0; u < n_sunits; ++u) {
        const int32_t* su = T + p.sunit_off + u * SUNIT_INTS;
        if (su[SU_ITEM0] < 0 || su[SU_ITEM1] < su[SU_ITEM0] || su[SU_ITEM1] > n_items) return bad("super-unit item range", u, su[SU_ITEM1], n_items);
    }
    for (int i = 0; i < n_items; ++i) {
        const int32_t* it = T + p.item_off + i * GITEM_INTS;
        if (it[I_PBUF] < 0 || it[I_PBUF] >= GBUF_COUNT) return bad("item P buffer", i, it[I_PBUF], GBUF_COUNT);
        if (it[I_SRC0] < 0 || it[I_NSRC] < 1 || it[I_SRC0] + it[I_NSRC] > n_srcs) return bad("item source range", i, it[I_SRC0] + it[I_NSRC], n_srcs);
        { const std::string e = check_srcs(it[I_SRC0], it[I_NSRC], it[I_KIND], "item", i); if (!e.empty()) return e; }
    }
    return "";
}
}  // namespace

extern "C" {

const char* mshgnn_hostplan_last_error() { return g_hp_err.c_str(); }

enum { HPM_OK = 0, HPM_L, HPM_NN, HPM_NMLP, HPM_FUSED, HPM_SLAB, HPM_SL_HB, HPM_SL_BLK, HPM_FS_BLK, HPM_NTABLES, HPM_NPACKS, HPM_NBIASES, HPM_SL_ALIAS, HPM_NITEMS, HPM_NLANES, HPM_NFIN,
       HPM_SL_FWD = 16, HPM_SL_BWD = 32, HPM_FS_FWD = 48, HPM_FS_BWD = 64, HPM_LIVE = 80 /* [l] low / high 32 bits of the node bit mask: 2 ints per layer */, HPM_NEED = 112, HPM_X3_ALIAS = 144, HPM_SPLIT = 145, HPM_COUNT = 160 };

// Compile the LDS-resident plan of `desc` on the host.  tables (may be null): receives min(cap, n) table ints; meta: HPM_COUNT ints (layout above).
// Returns the number of table ints, or -1 (mshgnn_hostplan_last_error()).  check != 0: also bounds-check every index table (error text names the first violation).
int mshgnn_hostplan_compile(const mshgnn_desc* desc, int32_t* tables, int cap, int32_t* meta, int check) {
    HostPlan hp;
    if (!compile_plan(desc, hp)) { g_hp_err = hp.err; return -1; }
    if (check) { const std::string e = check_tables(hp); if (!e.empty()) { g_hp_err = "table check: " + e; return -2; } }
    if (meta) {
        std::memset(meta, 0, sizeof(int32_t) * HPM_COUNT);
        meta[HPM_OK] = 1; meta[HPM_L] = hp.L; meta[HPM_NN] = hp.NN; meta[HPM_NMLP] = hp.n_mlp; meta[HPM_FUSED] = hp.fused; meta[HPM_SLAB] = hp.slab; meta[HPM_SL_HB] = hp.sl_hb;
        meta[HPM_SL_BLK] = hp.sl_blk; meta[HPM_FS_BLK] = hp.fs_blk; meta[HPM_NTABLES] = (int)hp.tables.size(); meta[HPM_NPACKS] = (int)hp.packs.size(); meta[HPM_NBIASES] = (int)hp.biases.size();
        meta[HPM_X3_ALIAS] = hp.x3_alias; meta[HPM_SPLIT] = hp.split;
        meta[HPM_SL_ALIAS] = hp.sl_alias; meta[HPM_NITEMS] = hp.n_items; meta[HPM_NLANES] = hp.n_lanes; meta[HPM_NFIN] = hp.n_fin;
        for (int l = 0; l < hp.L; ++l) {
            meta[HPM_SL_FWD + l] = hp.sl_fwd_off[l]; meta[HPM_SL_BWD + l] = hp.sl_bwd_off[l]; meta[HPM_FS_FWD + l] = hp.fs_fwd_off[l]; meta[HPM_FS_BWD + l] = hp.fs_bwd_off[l];
            uint64_t lv = 0, nd = 0;
            for (int n = 0; n < hp.NN; ++n) { lv |= (uint64_t)(hp.live_n[l][n] ? 1 : 0) << n; nd |= (uint64_t)(hp.need_n[l][n] ? 1 : 0) << n; }
            meta[HPM_LIVE + 2 * l] = (int32_t)(uint32_t)lv; meta[HPM_LIVE + 2 * l + 1] = (int32_t)(uint32_t)(lv >> 32);
            meta[HPM_NEED + 2 * l] = (int32_t)(uint32_t)nd; meta[HPM_NEED + 2 * l + 1] = (int32_t)(uint32_t)(nd >> 32);
        }
    }
    if (tables) std::memcpy(tables, hp.tables.data(), sizeof(int32_t) * std::min<size_t>(cap > 0 ? cap : 0, hp.tables.size()));
    return (int)hp.tables.size();
}

// The generic-width plan of `desc` (any hidden multiple of 128, any node count).  meta (>= 16 ints): [0] ok, [1] L, [2] NN, [3] hidden, [4] n table ints, [5] packs, [6] jobs, [7] terms,
// [8] sources, [9] units, [10] items.  Returns the number of table ints, -1 on a descriptor the engine refuses, -2 when a table entry is out of range (check != 0).
int mshgnn_hostplan_compile_gen(const mshgnn_desc* desc, int32_t* meta, int check) {
    gen::GenPlan gp;
    if (!gen::compile_gen_plan(desc, gp)) { g_hp_err = gp.err; return -1; }
    if (check) { const std::string e = check_gen_tables(gp); if (!e.empty()) { g_hp_err = "generic table check: " + e; return -2; } }
    if (meta) {
        using namespace gen;
        std::memset(meta, 0, sizeof(int32_t) * 16);
        meta[0] = 1; meta[1] = gp.L; meta[2] = gp.NN; meta[3] = gp.Hd; meta[4] = (int)gp.tables.size(); meta[5] = (int)gp.packs.size();
        meta[6] = (gp.term_off - gp.job_off) / JOB_INTS; meta[7] = (gp.src_off - gp.term_off) / TERM_INTS; meta[8] = (gp.unit_off - gp.src_off) / SRC_INTS;
        meta[9] = gp.n_units; meta[10] = (gp.fin_off - gp.item_off) / GITEM_INTS;
    }
    return (int)gp.tables.size();
}

// The generic plan's tables themselves (tools/ggradw_traffic_model.py): `tables` receives min(cap, n) ints; meta (>= 16 ints): [0] sunit_off, [1] su_order_off, [2] item_off,
// [3] src_off, [4] n_sunits, [5] n_parts, [6] su_os, [7] n_units, [8] fin_off, [9] unit_off.  Returns the number of table ints or -1.
int mshgnn_hostplan_gen_tables(const mshgnn_desc* desc, int32_t* tables, int cap, int32_t* meta) {
    gen::GenPlan gp;
    if (!gen::compile_gen_plan(desc, gp)) { g_hp_err = gp.err; return -1; }
    if (meta) {
        meta[0] = gp.sunit_off; meta[1] = gp.su_order_off; meta[2] = gp.item_off; meta[3] = gp.src_off; meta[4] = gp.n_sunits; meta[5] = gp.n_parts; meta[6] = gp.su_os; meta[7] = gp.n_units;
        meta[8] = gp.fin_off; meta[9] = gp.unit_off;
    }
    if (tables) std::memcpy(tables, gp.tables.data(), sizeof(int32_t) * std::min<size_t>(cap > 0 ? cap : 0, gp.tables.size()));
    return (int)gp.tables.size();
}

}  // extern "C"
