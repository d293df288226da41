#!/usr/bin/env python3
"""Generates morphsym_hgnn_amd/csrc/mshgnn_wide_engine.inc: the MAC engine of the engine-driven stack kernels (mshgnn_wide.hip) as ONE hand-scheduled
gfx950 asm statement per layer -- threaded code over a jump table of fixed-size bodies.

Why asm: hipcc's code for the interpreted slot walk of the stack kernels (12-20 statically unrolled accumulator slots, each with a run-time MAC count)
pays ~47 cycles per skipped slot header and copies the operand registers around a `switch` (tools/probe/walk_cost*.hip); an MFMA gap of 16 cycles hides
one or two cheap instructions of the same wave (tools/probe/issue_cost.hip), so what is issued between the MFMAs has to be chosen by hand.  Here a
layer program is a flat stream of 16-bit entries {index of the NEXT body, LDS block of the MAC after next}; each body ends with s_setpc_b64 to the next
one, so nothing is skipped and nothing is decoded outside an MFMA gap:
  * MAC body (slot S, weight buffer b): the v_mfma_f32_16x16x32_bf16 of one (destination, source) pair on the slot's accumulators (NH 16-window halves
    x two 16-feature blocks x 4 K steps, the order of mac() in mshgnn_device.hpp), the ds_read_b128 that refill the window fragment for the NEXT MAC
    behind the MFMAs that consumed it, the address updates for the MAC after next, and the scalar decode of the next entry, all between the MFMAs;
  * segment switch: waits for the weight fragment of the segment that starts (NBUF rotating register buffers) and requests a later segment's;
  * exit.
Two geometries (class Geo):
  * wide  (NH = 2, NBUF = 3): 32-window tiles, one 4-wave workgroup per CU, 512 registers per wave;
  * slab2 (NH = 1, NBUF = 2): 16-window tiles, two 4-wave workgroups per CU, 256 registers per wave = 128 accumulator + 128 vector registers (the
    split hipcc gives a 256-register kernel): v0..v31 for the compiler, two weight buffers, the window fragment, slots 16 / 17 in v[112:127].
Accumulators live in FIXED registers for the whole kernel (slot s < 16: a[8 NH s ..]; slots 16..: v[VACC + 8 NH (s - 16) ..]).  The compiler tracks them as
values of 16 NH registers (two slots each) that every statement touching them names as operands PINNED to their registers ("+{a[0:31]}" ...), so it
never allocates anything else there; the instruction text addresses single registers literally.  C++ code reads / writes them through the wd_acc_*
accessors of mshgnn_wide.hip.  The engine's own vector registers (weight buffers + window fragment) and s84..s95 are clobbers.
"""
import os

NSLOT_T = 20            # table stride of the slot bodies per weight buffer
HDR_TYPE_LANE = 32      # FH_NEXT + 8 - 64 (mshgnn_plan.hpp): lane of the header's second register that holds the type of node 0
BODY = 512              # bytes per table entry


class Geo:
    def __init__(self, name, nh, nbuf, wbase, xbase, vacc, ns_list):
        self.name, self.nh, self.nbuf, self.ns_list = name, nh, nbuf, ns_list
        self.R = [wbase + 32 * b for b in range(nbuf)]      # weight buffers (32 VGPRs each)
        self.X = xbase                                        # window fragment: X[h][t] = v[X + 16 h + 4 t .. + 3]
        self.vacc = vacc                                      # first VGPR of the accumulators of slots 16.. (None: every slot in the accumulator half)
        self.clob_lo, self.clob_hi = wbase, xbase + 16 * nh   # clobbered vector registers [lo, hi)
        self.idx_sw_issue = NSLOT_T * nbuf
        self.idx_sw_noissue = self.idx_sw_issue + nbuf
        self.idx_sw_last = self.idx_sw_noissue + 1
        self.idx_exit = self.idx_sw_last + 1
        self.n_table = self.idx_exit + 1

    def acc(self, slot, h, fb):
        if self.vacc is None or slot < 16:
            b = 8 * self.nh * slot + 8 * h + 4 * fb
            return f"a[{b}:{b + 3}]"
        b = self.vacc + 8 * self.nh * (slot - 16) + 8 * h + 4 * fb
        return f"v[{b}:{b + 3}]"

    def tuples(self, ns):
        """The accumulator registers as pinned operands of 16 NH registers (two slots): (C++ name, register range)."""
        n = 16 * self.nh
        na = min(ns, 16) if self.vacc is not None else ns
        t = [(f"r.a[{k}]", f"a[{n * k}:{n * k + n - 1}]") for k in range((na + 1) // 2)]
        if self.vacc is not None and ns > 16:
            t += [(f"r.v[{k}]", f"v[{self.vacc + n * k}:{self.vacc + n * k + n - 1}]") for k in range((ns - 16 + 1) // 2)]
        return t


WIDE = Geo("wide", 2, 3, 64, 160, 192, (16, 18, 20))
SLAB2 = Geo("slab2", 1, 2, 32, 96, 112, (16, 18))      # 128 + 128 registers: hipcc splits a 256-register budget in halves, so slots 16, 17 live in v[112:127]


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def decode_ops():
    """Scalar work of every body: the jump to the next body, the LDS offset of the MAC after next, the fetch of the next entry (all from %[ent])."""
    return [
        "s_and_b32 %[t0], %[ent], 0x7f",              # 0: index of the next body
        "s_lshl_b32 %[t0], %[t0], 9",                 # 1
        "s_add_u32 %[jlo], %[tblo], %[t0]",           # 2
        "s_addc_u32 %[jhi], %[tbhi], 0",              # 3: -> %[j]
        "s_bfe_u32 %[off], %[ent], 0x50007",          # 4: LDS block of the MAC after next
        "s_lshl_b32 %[off], %[off], %[blksh]",        # 5: -> %[off] (needed by the first address update)
        "s_add_i32 %[i], %[i], 1",                    # 6: fetch entry i + 1: dword (i + 1) >> 1 of the program register, half (i + 1) & 1
        "s_lshr_b32 %[t1], %[i], 1",                  # 7
        "s_bitcmp1_b32 %[i], 0",                      # 8
        "s_cselect_b32 %[t0], 16, 0",                 # 9
        "v_readlane_b32 %[t1], %[prog], %[t1]",       # 10
        "s_lshr_b32 %[ent], %[t1], %[t0]",            # 11: -> %[ent] (nothing reads the old entry from here on)
    ]


def mac_body(g, slot, buf):
    w = g.R[buf]
    dec = decode_ops()
    out = []
    # %[off] before the first address update (end of K step 0), the jump target next, the fetch of the next entry last
    if g.nh == 2:
        per_step = {0: dec[4:6] + dec[0:4], 1: dec[6:10], 2: dec[10:12], 3: []}
    else:
        per_step = {0: dec[4:6] + dec[0:2], 1: dec[2:4] + dec[6:8], 2: dec[8:10], 3: dec[10:12]}
    for t in range(4):
        fill = list(per_step[t])
        out.append(f"s_waitcnt lgkmcnt({3 * g.nh})")      # all but the reads younger than this K step's (LDS returns in order)
        for h in range(g.nh):
            for fb in range(2):
                out.append(f"v_mfma_f32_16x16x32_bf16 {g.acc(slot, h, fb)}, {vr(w + 4 * (4 * fb + t))}, {vr(g.X + 16 * h + 4 * t)}, {g.acc(slot, h, fb)}")
                for _ in range(2):      # up to two cheap instructions per MFMA gap
                    if fill:
                        out.append(fill.pop(0))
        assert not fill
        for h in range(g.nh):
            out.append(f"ds_read_b128 {vr(g.X + 16 * h + 4 * t)}, %[va{t}]" + (f" offset:{4096 * h}" if h else ""))
        out.append(f"v_add_u32 %[va{t}], %[off], %[ao{t}]")
    out.append("s_setpc_b64 %[j]")
    return out


def frag_loads(g, buf):
    out = []
    for v in range(8):
        if v == 4:
            out += ["s_add_u32 %[plo], %[plo], 4096", "s_addc_u32 %[phi], %[phi], 0"]
        out.append(f"global_load_dwordx4 {vr(g.R[buf] + 4 * v)}, %[lane16], %[p] offset:{(v % 4) * 1024}")
    return out


def frag_ptr(seg_expr_lane):
    """%[p] = wave fragment base + pack(segment) * 32768; seg_expr_lane: SGPR or constant lane of %[pk]"""
    return [
        f"v_readlane_b32 %[t0], %[pk], {seg_expr_lane}",
        "s_lshl_b32 %[t0], %[t0], 15",
        "s_add_u32 %[plo], %[wlo], %[t0]",
        "s_addc_u32 %[phi], %[whi], 0",
    ]


def sw_body(g, kind, buf=0):
    """The fragments of the segments up to NBUF - 1 ahead are in flight: waiting for all but the (NBUF - 2) youngest fragments' loads leaves exactly
    the fragment of the segment that starts complete (loads return in order; nothing else of this wave is outstanding inside the engine)."""
    out = []
    keep = 8 * (g.nbuf - 2)
    if kind == "issue":
        out.append(f"s_waitcnt vmcnt({keep})")
        out += frag_ptr("%[seg]")
        out.append("s_add_i32 %[seg], %[seg], 1")
        out += frag_loads(g, buf)
    elif kind == "noissue":
        out.append(f"s_waitcnt vmcnt({keep})")
    else:
        out.append("s_waitcnt vmcnt(0)")
    out += decode_ops()
    out.append("s_setpc_b64 %[j]")
    return out


def engine_text(g, ns):
    L = []
    a = L.append
    # (no vmcnt wait here: the fragment loads below are requested behind whatever the epilogue before this statement still has in flight, and the first
    #  switch's counted wait covers both -- memory operations retire in order)
    a("s_waitcnt lgkmcnt(0)")
    a("s_getpc_b64 %[tb]")
    a("L_pc%=:")
    a("s_add_u32 %[tblo], %[tblo], L_table%=-L_pc%=")
    a("s_addc_u32 %[tbhi], %[tbhi], 0")
    # fragments of the first NBUF - 1 segments
    L += frag_ptr("0")
    L += frag_loads(g, 0)
    for s in range(1, g.nbuf - 1):
        a(f"s_cmp_lt_u32 %[nseg], {s + 1}")
        a("s_cbranch_scc1 L_pre%=")
        L += frag_ptr(str(s))
        L += frag_loads(g, s)
    a("L_pre%=:")
    a(f"s_mov_b32 %[seg], {g.nbuf - 1}")
    # forward layers (init != 0): the accumulators start at the bias row of their node's type, straight from this wave's rows in LDS into the
    # accumulator registers (lane HDR_TYPE_LANE + slot of the header's second register = type of the node; 128 bytes per type), under the fragment loads
    a("s_cmp_eq_u32 %[init], 0")
    a("s_cbranch_scc1 L_noinit%=")
    for u in range(ns):
        a(f"v_readlane_b32 %[t0], %[hdr1], {HDR_TYPE_LANE + u}")
        a("s_lshl_b32 %[t0], %[t0], 7")
        a("v_add_u32 %[va0], %[t0], %[bbase]")
        for h in range(g.nh):
            for fb in range(2):
                a(f"ds_read_b128 {g.acc(u, h, fb)}, %[va0]" + (" offset:16" if fb else ""))
    a("L_noinit%=:")
    # window fragment of MAC 0, addresses of MAC 1's
    for t in range(4):
        a(f"v_add_u32 %[va{t}], %[blk0], %[ao{t}]")
    for t in range(4):
        for h in range(g.nh):
            a(f"ds_read_b128 {vr(g.X + 16 * h + 4 * t)}, %[va{t}]" + (f" offset:{4096 * h}" if h else ""))
    for t in range(4):
        a(f"v_add_u32 %[va{t}], %[blk1], %[ao{t}]")
    # entry 0, jump to the first body
    a("s_mov_b32 %[i], 0")
    a("v_readlane_b32 %[ent], %[prog], 0")
    a("s_lshl_b32 %[t0], %[first], 9")
    a("s_add_u32 %[jlo], %[tblo], %[t0]")
    a("s_addc_u32 %[jhi], %[tbhi], 0")
    a("s_nop 7")                       # accumulators freshly written by v_accvgpr_write / v_mov in other statements
    a("s_setpc_b64 %[j]")
    a(".p2align 9")
    a("L_table%=:")
    bodies = {}
    for buf in range(g.nbuf):
        for s in range(ns):
            bodies[buf * NSLOT_T + s] = mac_body(g, s, buf)
    for buf in range(g.nbuf):
        bodies[g.idx_sw_issue + buf] = sw_body(g, "issue", buf)
    bodies[g.idx_sw_noissue] = sw_body(g, "noissue")
    bodies[g.idx_sw_last] = sw_body(g, "last")
    bodies[g.idx_exit] = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 7", "s_branch L_end%="]
    for idx in range(g.n_table):
        for ins in bodies.get(idx, ["s_trap 2"]):
            a(ins)
        a(".p2align 9")
    a("L_end%=:")
    return L


def clobbers(g):
    return [f"v{i}" for i in range(g.clob_lo, g.clob_hi)] + [f"s{i}" for i in range(84, 96)] + ["vcc", "scc", "memory"]


SGPR_ALIASES = {"%[tb]": "s[90:91]", "%[tblo]": "s90", "%[tbhi]": "s91", "%[j]": "s[92:93]", "%[jlo]": "s92", "%[jhi]": "s93",
                "%[p]": "s[94:95]", "%[plo]": "s94", "%[phi]": "s95",      # 64-bit scalars and their halves: literal registers (an operand prints only as a pair)
                "%[t0]": "s84", "%[t1]": "s85", "%[off]": "s86", "%[i]": "s87", "%[seg]": "s88", "%[ent]": "s89"}      # scalar temporaries (an asm statement takes 30 operands)


def emit(g, ns):
    text = engine_text(g, ns)
    for k, v in SGPR_ALIASES.items():
        text = [t.replace(k, v) for t in text]
    text = [t.replace("%[blksh]", str(12 + (g.nh - 1))) for t in text]      # LDS block of a node: NH x 4 KB
    s = []
    s.append(f"template <> __device__ __forceinline__ void wd_engine<{g.nh}, {ns}>(WRegs<{g.nh}>& r, int prog, int pk, int lane16, const int (&ao)[4], const void* wfrag, int blk0, int blk1, int nseg, int first, int hdr1, int bbase, int init) {{")
    s.append("    int va0, va1, va2, va3;")
    s.append("    const unsigned wlo = (unsigned)(unsigned long long)wfrag, whi = (unsigned)((unsigned long long)wfrag >> 32);")
    s.append("    asm volatile(")
    for ins in text:
        s.append(f'        "{ins}\\n\\t"')
    s.append('        : [va0] "=&v"(va0), [va1] "=&v"(va1), [va2] "=&v"(va2), [va3] "=&v"(va3),')
    s.append("          " + ", ".join(f'"+{{{reg}}}"({name})' for name, reg in g.tuples(ns)))
    s.append('        : [prog] "v"(prog), [pk] "v"(pk), [lane16] "v"(lane16), [ao0] "v"(ao[0]), [ao1] "v"(ao[1]), [ao2] "v"(ao[2]), [ao3] "v"(ao[3]), [wlo] "s"(wlo), [whi] "s"(whi),')
    s.append('          [blk0] "s"(blk0), [blk1] "s"(blk1), [nseg] "s"(nseg), [first] "s"(first), [hdr1] "v"(hdr1), [bbase] "v"(bbase), [init] "s"(init)')
    s.append("        : " + ", ".join(f'"{c}"' for c in clobbers(g)) + ");")
    s.append("}")
    return "\n".join(s)


HEADER = '''// GENERATED by tools/gen_wide_engine.py -- do not edit.  The MAC engine of the engine-driven stack kernels: see the generator's docstring.
#pragma once
typedef float wd_f32x32 __attribute__((ext_vector_type(32)));
typedef float wd_f32x16 __attribute__((ext_vector_type(16)));
// the accumulator registers as values the compiler tracks: a[k] is pinned to a[16 NH k : 16 NH k + 16 NH - 1] (slots 2 k, 2 k + 1), v[k] to
// v[VACC + 16 NH k ..] (slots 16 + 2 k, 17 + 2 k; VACC = 192 wide, 112 slab2)
template <int NH> struct WRegs;
template <> struct WRegs<2> { wd_f32x32 a[8]; wd_f32x32 v[2]; };
template <> struct WRegs<1> { wd_f32x16 a[8]; wd_f32x16 v[1]; };
template <int NH, int NS> __device__ __forceinline__ void wd_engine(WRegs<NH>& r, int prog, int pk, int lane16, const int (&ao)[4], const void* wfrag, int blk0, int blk1, int nseg, int first, int hdr1, int bbase, int init);
'''


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "morphsym_hgnn_amd", "csrc", "mshgnn_wide_engine.inc")
    parts = [HEADER]
    for g in (WIDE, SLAB2):
        parts.append(f"// geometry {g.name}: NH = {g.nh}, {g.nbuf} weight buffers; table: MAC body of (slot, buffer) = buffer * {NSLOT_T} + slot, switch + issue(b) = {g.idx_sw_issue} + b, "
                     f"switch = {g.idx_sw_noissue}, last switch = {g.idx_sw_last}, exit = {g.idx_exit}")
        for ns in g.ns_list:
            parts.append(emit(g, ns))
    txt = "\n".join(parts) + "\n"
    with open(out, "w") as f:
        f.write(txt)
    print("wrote", out, len(txt), "bytes")


if __name__ == "__main__":
    main()
