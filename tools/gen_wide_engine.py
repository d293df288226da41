#!/usr/bin/env python3
"""Generates morphsym_hgnn_amd/csrc/mshgnn_wide_engine.inc: the MAC engine of the wide stack kernels (mshgnn_wide.hip) as ONE hand-scheduled gfx950
asm statement per layer -- threaded code over a jump table of fixed-size bodies.

Why asm: one wave per SIMD issues strictly in order, so every instruction that is not hidden in an MFMA gap costs wall time, and hipcc's code for the
interpreted slot walk (18-20 statically unrolled accumulator slots, each with a run-time MAC count) pays ~47 cycles per skipped slot header and copies
the operand registers around a `switch` (tools/probe/walk_cost*.hip, tools/probe/issue_cost.hip: an MFMA gap of 16 cycles hides two cheap instructions).
Here a layer program is a flat stream of 16-bit entries {index of the NEXT body, LDS block of the MAC after next}; each body ends with s_setpc_b64 to
the next one, so nothing is skipped and nothing is decoded outside an MFMA gap:
  * MAC body (slot S, weight buffer b): 16 v_mfma_f32_16x16x32_bf16 on the slot's accumulators (two 16-window halves x two 16-feature blocks x 4 K
    steps, the order of mac() in mshgnn_device.hpp), the 8 ds_read_b128 that refill the window fragment for the NEXT MAC behind the MFMAs that consumed
    it, the 4 address updates for the MAC after next, and the scalar decode of the next entry, all between the MFMAs;
  * segment switch: waits for the weight fragment of the segment that starts (three rotating register buffers: two segments of prefetch) and requests
    the fragment of the segment after next;
  * exit.
Accumulators live in FIXED registers for the whole kernel: slots 0..15 in a[16 s .. 16 s + 15], slots 16..19 in v[192 + 16 (s - 16) ..].  The compiler
tracks them as ten 32-register values (WRegs) that every statement touching them names as operands PINNED to their registers ("+{a[0:31]}" ...), so
it never allocates anything else there; the instruction text addresses single registers literally.  C++ code reads / writes them through the
wd_acc_* accessors of mshgnn_wide.hip.  The engine's own registers (v64..v191: three weight buffers + the window fragment; s84..s95) are clobbers.
"""
import os
import sys

NSLOT_T = 20            # table stride of the slot bodies per weight buffer
BODY = 512              # bytes per table entry
R = [64, 96, 128]       # weight buffers (32 VGPRs each)
X = 160                 # window fragment: X[h][t] = v[X + 16 h + 4 t .. + 3]
VACC = 192              # accumulators of slots 16..19
IDX_SW_ISSUE = 60       # + buffer
IDX_SW_NOISSUE = 63
IDX_SW_LAST = 64
IDX_EXIT = 65
N_TABLE = 66


def acc(slot, h, fb):
    if slot < 16:
        b = 16 * slot + 8 * h + 4 * fb
        return f"a[{b}:{b + 3}]"
    b = VACC + 16 * (slot - 16) + 8 * h + 4 * fb
    return f"v[{b}:{b + 3}]"


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def decode_ops():
    """Scalar work of every body: the jump to the next body, the LDS offset of the MAC after next, the fetch of the next entry (all from %[ent])."""
    return [
        "s_and_b32 %[t0], %[ent], 0x7f",
        "s_lshl_b32 %[t0], %[t0], 9",
        "s_add_u32 %[jlo], %[tblo], %[t0]",
        "s_addc_u32 %[jhi], %[tbhi], 0",
        "s_bfe_u32 %[off], %[ent], 0x50007",
        "s_lshl_b32 %[off], %[off], 13",
        "s_add_i32 %[i], %[i], 1",
        "s_lshr_b32 %[t1], %[i], 1",
        "s_bitcmp1_b32 %[i], 0",
        "s_cselect_b32 %[t0], 16, 0",
        "v_readlane_b32 %[t1], %[prog], %[t1]",
        "s_lshr_b32 %[ent], %[t1], %[t0]",
    ]


def mac_body(slot, buf):
    w = R[buf]
    dec = decode_ops()
    out = []
    # the first six decode ops (jump target + offset) go into K step 0's gaps, the entry fetch into K steps 1-2
    sched = {0: dec[0:6], 1: dec[6:10], 2: dec[10:12], 3: []}
    for t in range(4):
        fill = list(sched[t])
        out.append("s_waitcnt lgkmcnt(6)")
        k = 0
        for h in range(2):
            for fb in range(2):
                out.append(f"v_mfma_f32_16x16x32_bf16 {acc(slot, h, fb)}, {vr(w + 4 * (4 * fb + t))}, {vr(X + 16 * h + 4 * t)}, {acc(slot, h, fb)}")
                for _ in range(2):      # up to two cheap instructions per MFMA gap
                    if fill:
                        out.append(fill.pop(0))
        out.append(f"ds_read_b128 {vr(X + 4 * t)}, %[va{t}]")
        out.append(f"ds_read_b128 {vr(X + 16 + 4 * t)}, %[va{t}] offset:4096")
        out.append(f"v_add_u32 %[va{t}], %[off], %[ao{t}]")
        assert not fill
    out.append("s_setpc_b64 %[j]")
    return out


def frag_loads(buf):
    out = []
    for v in range(8):
        if v == 4:
            out += ["s_add_u32 %[plo], %[plo], 4096", "s_addc_u32 %[phi], %[phi], 0"]
        out.append(f"global_load_dwordx4 {vr(R[buf] + 4 * v)}, %[lane16], %[p] offset:{(v % 4) * 1024}")
    return out


def frag_ptr(seg_expr_lane):
    """%[p] = wave fragment base + pack(segment) * 32768; seg_expr_lane: SGPR or constant lane of %[pk]"""
    return [
        f"v_readlane_b32 %[t0], %[pk], {seg_expr_lane}",
        "s_lshl_b32 %[t0], %[t0], 15",
        "s_add_u32 %[plo], %[wlo], %[t0]",
        "s_addc_u32 %[phi], %[whi], 0",
    ]


def sw_body(kind, buf=0):
    out = []
    if kind == "issue":
        out.append("s_waitcnt vmcnt(8)")
        out += frag_ptr("%[seg]")
        out.append("s_add_i32 %[seg], %[seg], 1")
        out += frag_loads(buf)
    elif kind == "noissue":
        out.append("s_waitcnt vmcnt(8)")
    else:
        out.append("s_waitcnt vmcnt(0)")
    out += decode_ops()
    out.append("s_setpc_b64 %[j]")
    return out


def engine_text(ns):
    L = []
    a = L.append
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_getpc_b64 %[tb]")
    a("L_pc%=:")
    a("s_add_u32 %[tblo], %[tblo], L_table%=-L_pc%=")
    a("s_addc_u32 %[tbhi], %[tbhi], 0")
    # fragments of segments 0 and 1
    L += frag_ptr("0")
    L += frag_loads(0)
    a("s_cmp_lt_u32 %[nseg], 2")
    a("s_cbranch_scc1 L_one%=")
    L += frag_ptr("1")
    L += frag_loads(1)
    a("L_one%=:")
    a("s_mov_b32 %[seg], 2")
    # window fragment of MAC 0, addresses of MAC 1's
    for t in range(4):
        a(f"v_add_u32 %[va{t}], %[blk0], %[ao{t}]")
    for t in range(4):
        a(f"ds_read_b128 {vr(X + 4 * t)}, %[va{t}]")
        a(f"ds_read_b128 {vr(X + 16 + 4 * t)}, %[va{t}] offset:4096")
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
    for buf in range(3):
        for s in range(ns):
            bodies[buf * NSLOT_T + s] = mac_body(s, buf)
    for buf in range(3):
        bodies[IDX_SW_ISSUE + buf] = sw_body("issue", buf)
    bodies[IDX_SW_NOISSUE] = sw_body("noissue")
    bodies[IDX_SW_LAST] = sw_body("last")
    bodies[IDX_EXIT] = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 7", "s_branch L_end%="]
    for idx in range(N_TABLE):
        for ins in bodies.get(idx, ["s_trap 2"]):
            a(ins)
        a(".p2align 9")
    a("L_end%=:")
    return L


def clobbers(ns):
    return [f"v{i}" for i in range(64, 192)] + [f"s{i}" for i in range(84, 96)] + ["vcc", "scc", "memory"]


def tuples(ns):
    """The accumulator registers as pinned 32-register operands (name, constraint register): the compiler sees them occupied for the whole kernel."""
    t = [(f"r.a[{k}]", f"a[{32 * k}:{32 * k + 31}]") for k in range(8)]
    if ns > 16:
        t.append(("r.v[0]", f"v[{VACC}:{VACC + 31}]"))
    if ns > 18:
        t.append(("r.v[1]", f"v[{VACC + 32}:{VACC + 63}]"))
    return t


SGPR_ALIASES = {"%[tb]": "s[90:91]", "%[tblo]": "s90", "%[tbhi]": "s91", "%[j]": "s[92:93]", "%[jlo]": "s92", "%[jhi]": "s93",
                "%[p]": "s[94:95]", "%[plo]": "s94", "%[phi]": "s95",      # 64-bit scalars and their halves: literal registers (an operand prints only as a pair)
                "%[t0]": "s84", "%[t1]": "s85", "%[off]": "s86", "%[i]": "s87", "%[seg]": "s88", "%[ent]": "s89"}      # scalar temporaries (an asm statement takes 30 operands)


def emit(ns):
    text = engine_text(ns)
    for k, v in SGPR_ALIASES.items():
        text = [t.replace(k, v) for t in text]
    s = []
    s.append(f"template <> __device__ __forceinline__ void wd_engine<{ns}>(WRegs& r, int prog, int pk, int lane16, const int (&ao)[4], const void* wfrag, int blk0, int blk1, int nseg, int first) {{")
    s.append("    int va0, va1, va2, va3;")
    s.append("    const unsigned wlo = (unsigned)(unsigned long long)wfrag, whi = (unsigned)((unsigned long long)wfrag >> 32);")
    s.append("    asm volatile(")
    for ins in text:
        s.append(f'        "{ins}\\n\\t"')
    s.append('        : [va0] "=&v"(va0), [va1] "=&v"(va1), [va2] "=&v"(va2), [va3] "=&v"(va3),')
    s.append("          " + ", ".join(f'"+{{{reg}}}"({name})' for name, reg in tuples(ns)))
    s.append('        : [prog] "v"(prog), [pk] "v"(pk), [lane16] "v"(lane16), [ao0] "v"(ao[0]), [ao1] "v"(ao[1]), [ao2] "v"(ao[2]), [ao3] "v"(ao[3]), [wlo] "s"(wlo), [whi] "s"(whi),')
    s.append('          [blk0] "s"(blk0), [blk1] "s"(blk1), [nseg] "s"(nseg), [first] "s"(first)')
    cl = clobbers(ns)
    s.append("        : " + ", ".join(f'"{c}"' for c in cl) + ");")
    s.append("}")
    return "\n".join(s)


HEADER = '''// GENERATED by tools/gen_wide_engine.py -- do not edit.  The MAC engine of the wide stack kernels: see the generator's docstring.
#pragma once
typedef float wd_f32x32 __attribute__((ext_vector_type(32)));
// the accumulator registers as values the compiler tracks: a[k] is pinned to a[32 k : 32 k + 31] (slots 2 k, 2 k + 1), v[k] to v[192 + 32 k : 223 + 32 k] (slots 16 + 2 k, ...)
struct WRegs { wd_f32x32 a[8]; wd_f32x32 v[2]; };
template <int NS> __device__ __forceinline__ void wd_engine(WRegs& r, int prog, int pk, int lane16, const int (&ao)[4], const void* wfrag, int blk0, int blk1, int nseg, int first);
'''


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "morphsym_hgnn_amd", "csrc", "mshgnn_wide_engine.inc")
    parts = [HEADER]
    for ns in (16, 18, 20):
        parts.append(emit(ns))
    txt = "\n".join(parts) + "\n"
    # lo / hi halves of the 64-bit scalar operands: the assembler takes s[n:n+1] only whole, so the halves are separate 32-bit operands tied to the pair
    with open(out, "w") as f:
        f.write(txt)
    print("wrote", out, len(txt), "bytes")


if __name__ == "__main__":
    main()
