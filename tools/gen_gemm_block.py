#!/usr/bin/env python3
"""Generates morphsym_hgnn_amd/csrc/mshgnn_gemm_block.inc: the MFMA block of the generic-width engine's job kernel k_gstep3 (mshgnn_gen.hip) as hand-scheduled
gfx950 asm -- ONE 128-wide K chunk of a wave's 128-window x 128-column output tile per statement: 256 v_mfma_f32_16x16x32_bf16 on 256 accumulator registers that
stay in a[0:255] for the whole kernel (the compiler tracks them as eight 32-register values pinned to their registers, as in mshgnn_wide_engine.inc).

Why asm: with 256 accumulators + operand buffers hipcc's schedule of this loop moves accumulators between the two register-file halves (1 456 moves) and
keeps one window fragment in flight; here the weight fragments of K step t + 1 (8 x global_load_dwordx4) and the window fragments of K step t + 1
(8 x ds_read_b128) are requested in the first MFMA gaps of K step t and have 64 MFMAs (~1 100 cycles) to land.

Layout (mshgnn_device.hpp): accumulator of (row block m < 8, column block cb = 2 slice + fb < 8) = a[32 m + 4 cb : +3]; D^T = W_frag (A operand) x X_frag (B operand).
Weight fragment (slice s, fb, K step t) of pack image p: p * 32768 + ((8 s + 4 fb + t) * 64 + lane) * 16 bytes.  Window fragment (row block m, K step t): LDS
abuf + 4096 m + (ao0 ^ 16 t).  Two geometries (class Geo): four waves x 128 columns (the whole register file per wave) and eight waves x 64 columns (two waves per SIMD, 128 + 128 registers).
"""
import os

class Geo:
    """ns: 32-column slices per wave (4: four waves per workgroup, 512 registers each; 2: eight waves, two per SIMD, 256 registers each = 128 + 128)."""
    def __init__(self, ns, w, x, raw, rmk, nr, rstep, mstep, vlo, vhi, slo):
        self.ns, self.W, self.X, self.RAW, self.RMK, self.nr, self.rstep, self.mstep = ns, w, x, raw, rmk, nr, rstep, mstep
        self.ncb = 2 * ns                 # 16-feature column blocks per wave
        self.tup = 4 * self.ncb           # accumulator registers per row block
        self.vlo, self.vhi, self.slo = vlo, vhi, slo

    def acc(self, m, cb):
        b = self.tup * m + 4 * cb
        return f"a[{b}:{b + 3}]"


# W: two weight-fragment buffers; X: ONE window-fragment buffer (row block m's fragment of the next K step is requested right behind the last MFMA that read
# it); RAW / RMK: the next chunk's rows and relu bytes (handed back to the caller); nr rows per thread, rstep rows apart
G4 = Geo(4, [64, 96], 128, 160, 192, 8, 16, 64, 96, 200, 60)
G2 = Geo(2, [44, 60], 76, 108, 124, 4, 32, 128, 60, 128, 60)


def vr(b):
    return f"v[{b}:{b + 3}]"


def chunk_text(g):
    L = []
    a = L.append
    ncb = g.ncb
    # scalar bases of the (slice, fb) fragment streams: %[wlo/whi] + (8 s + 4 fb) * 1024
    for cb in range(ncb):
        s, fb = cb >> 1, cb & 1
        a(f"s_add_u32 s{g.slo + 2 * cb}, %[wlo], {(8 * s + 4 * fb) * 1024}")
        a(f"s_addc_u32 s{g.slo + 1 + 2 * cb}, %[whi], 0")
    SR, SB, SM = g.slo + 2 * ncb, g.slo + 2 * ncb + 2, g.slo + 2 * ncb + 4      # row-offset temporary, row base pair, mask base pair
    a(f"s_mov_b32 s{SB}, %[rblo]")
    a(f"s_mov_b32 s{SB + 1}, %[rbhi]")
    a(f"s_mov_b32 s{SM}, %[mblo]")
    a(f"s_mov_b32 s{SM + 1}, %[mbhi]")
    a("v_add_u32 %[va], %[abuf], %[ao0]")
    for m in range(8):
        a(f"ds_read_b128 {vr(g.X + 4 * m)}, %[va] offset:{4096 * m}")
    nload = 2 * g.nr
    for t in range(4):
        wb = g.W[t & 1]
        nwb = g.W[(t + 1) & 1]
        # weight fragments: K step 1 waits for all but the next chunk's rows / relu bytes (requested behind them in K step 0); later K steps for everything
        a({0: None, 1: f"s_waitcnt vmcnt({nload})"}.get(t, "s_waitcnt vmcnt(0)")) if t else None
        pre = []
        if t == 3:
            # the NEXT chunk's K-step-0 fragments into the buffer the block was handed (free since K step 2): waited for before the block ends
            for cb in range(ncb):
                s_, fb = cb >> 1, cb & 1
                pre.append(f"s_add_u32 s{g.slo + 2 * cb}, %[nwlo], {(8 * s_ + 4 * fb) * 1024}")
                pre.append(f"s_addc_u32 s{g.slo + 1 + 2 * cb}, %[nwhi], 0")
                pre.append(f"global_load_dwordx4 {vr(g.W[0] + 4 * cb)}, %[lane16], s[{g.slo + 2 * cb}:{g.slo + 1 + 2 * cb}]")
        else:
            pre.append(f"v_xor_b32 %[vb], {16 * (t + 1)}, %[va]")      # (first: row block 0's next fragment is requested right behind its MFMAs)
            for cb in range(ncb):
                pre.append(f"global_load_dwordx4 {vr(nwb + 4 * cb)}, %[lane16], s[{g.slo + 2 * cb}:{g.slo + 1 + 2 * cb}] offset:{1024 * (t + 1)}")
        if t == 0:
            for i in range(g.nr):
                if i:
                    pre.append(f"s_add_u32 s{SR}, s{SR}, %[rs]")
                    pre.append(f"v_add_u32 %[vt], s{SR}, %[toff0]")
                else:
                    pre.append(f"s_mov_b32 s{SR}, 0")
                pre.append(f"global_load_dwordx4 v[{g.RAW + 4 * i}:{g.RAW + 4 * i + 3}], " + ("%[vt]" if i else "%[toff0]") + f", s[{SB}:{SB + 1}]")
                pre.append(f"global_load_ubyte v{g.RMK + i}, %[toffm0], s[{SM}:{SM + 1}] offset:{g.mstep * i}")
        for m in range(8):
            a(f"s_waitcnt lgkmcnt({7 if t < 3 else 7 - m})")      # row block m's fragment is the oldest outstanding read
            for cb in range(ncb):
                a(f"v_mfma_f32_16x16x32_bf16 {g.acc(m, cb)}, {vr(wb + 4 * cb)}, {vr(g.X + 4 * m)}, {g.acc(m, cb)}")
                if pre:
                    a(pre.pop(0))
            if t < 3:
                a(f"ds_read_b128 {vr(g.X + 4 * m)}, %[vb] offset:{4096 * m}")      # the next K step's fragment of this row block
        while pre:
            a(pre.pop(0))
    a("s_waitcnt vmcnt(0)")
    a("s_nop 15")
    a("s_nop 7")
    return [x for x in L if x]


def emit_chunk(g):
    n = g.ns
    s = [f"// one K chunk of a wave's 128 windows x {32 * n} columns: acc += W_chunk x A_tile.  w0: the weight fragments of K step 0 (in: this chunk's; out: the next chunk's);",
         "// wslice / wnext: address of slice 0, fragment 0 of this wave's part of the chunk's / the next chunk's pack image; abuf: LDS byte offset of the A tile; ao0: this",
         "// lane's first chunk offset; lane16 = lane * 16; rows / rmask: the NEXT chunk's staging data (16 bytes at rbase + toff0 + i * rstride; relu byte at mbase + toffm0 + MSTEP i)",
         f"__device__ __forceinline__ void g4_chunk(GAcc<{n}>& r, G4W<{n}>& w0, G4Rows<{n}>& rows, G4Mask<{n}>& rmask, const void* wslice, const void* wnext, const void* rbase, const void* mbase,",
         "                                         unsigned rstride, int toff0, int toffm0, unsigned abuf, int ao0, int lane16) {",
         "    const unsigned wlo = (unsigned)(unsigned long long)wslice, whi = (unsigned)((unsigned long long)wslice >> 32);",
         "    const unsigned nwlo = (unsigned)(unsigned long long)wnext, nwhi = (unsigned)((unsigned long long)wnext >> 32);",
         "    const unsigned rblo = (unsigned)(unsigned long long)rbase, rbhi = (unsigned)((unsigned long long)rbase >> 32);",
         "    const unsigned mblo = (unsigned)(unsigned long long)mbase, mbhi = (unsigned)((unsigned long long)mbase >> 32);",
         "    int va, vb, vt;",
         "    asm volatile("]
    for ins in chunk_text(g):
        s.append(f'        "{ins}\\n\\t"')
    wn = 4 * g.ncb
    s.append(f'        : [va] "=&v"(va), [vb] "=&v"(vb), [vt] "=&v"(vt), "+{{v[{g.W[0]}:{g.W[0] + wn - 1}]}}"(w0.v), "=&{{v[{g.RAW}:{g.RAW + 4 * g.nr - 1}]}}"(rows.v), "=&{{v[{g.RMK}:{g.RMK + g.nr - 1}]}}"(rmask.v), '
             + ", ".join(f'"+{{a[{g.tup * m}:{g.tup * m + g.tup - 1}]}}"(r.a[{m}])' for m in range(8)))
    s.append('        : [wlo] "s"(wlo), [whi] "s"(whi), [nwlo] "s"(nwlo), [nwhi] "s"(nwhi), [rblo] "s"(rblo), [rbhi] "s"(rbhi), [mblo] "s"(mblo), [mbhi] "s"(mbhi), [rs] "s"(rstride),')
    s.append('          [toff0] "v"(toff0), [toffm0] "v"(toffm0), [abuf] "s"(abuf), [ao0] "v"(ao0), [lane16] "v"(lane16)')
    clob_v = [i for i in range(g.vlo, g.vhi) if not (g.W[0] <= i < g.W[0] + wn or g.RAW <= i < g.RAW + 4 * g.nr or g.RMK <= i < g.RMK + g.nr)]
    s.append("        : " + ", ".join(f'"v{i}"' for i in clob_v) + ", " + ", ".join(f'"s{i}"' for i in range(g.slo, g.slo + 2 * g.ncb + 6)) + ', "memory");')
    s.append("}")
    return "\n".join(s)


def emit_init(g):
    n, tup = g.ns, g.tup
    s = [f"// accumulators of every row block = the bias values of the wave's column blocks: b[4 cb + j] = feature j of column block cb = 2 slice + fb for this lane",
         f"__device__ __forceinline__ void g4_init(GAcc<{n}>& r, const float (&b)[{tup}]) {{"]
    # statements of 16 inputs (an asm statement takes 30 operands)
    for part in range(tup // 16):
        s.append("    asm volatile(")
        for m in range(8):
            for k in range(16):
                s.append(f'        "v_accvgpr_write_b32 a[{tup * m + 16 * part + k}], %{8 + k}\\n\\t"')
        outs = ", ".join(f'"{"=" if part == 0 else "+"}{{a[{tup * m}:{tup * m + tup - 1}]}}"(r.a[{m}])' for m in range(8))
        ins = ", ".join(f'"v"(b[{16 * part + k}])' for k in range(16))
        s.append(f"        : {outs}")
        s.append(f"        : {ins});")
    s.append("}")
    return "\n".join(s)


def emit_read(g):
    n, tup = g.ns, g.tup
    out = [f"// the accumulators of row block M (4 floats per column block cb = 2 slice + fb)",
           f"template <int M> __device__ __forceinline__ void g4_read(GAcc<{n}>& r, float (&o)[{tup}]);"]
    for m in range(8):
        out.append(f"template <> __device__ __forceinline__ void g4_read<{m}>(GAcc<{n}>& r, float (&o)[{tup}]) {{")
        for part in range(tup // 16):
            out.append("    asm volatile(")
            for k in range(16):
                out.append(f'        "v_accvgpr_read_b32 %{k}, a[{tup * m + 16 * part + k}]\\n\\t"')
            outs = ", ".join(f'"=v"(o[{16 * part + k}])' for k in range(16))
            out.append(f"        : {outs}")
            out.append(f'        : "{{a[{tup * m}:{tup * m + tup - 1}]}}"(r.a[{m}]));')
        out.append("}")
    return "\n".join(out)


HEADER = '''// GENERATED by tools/gen_gemm_block.py -- do not edit.  The MFMA block of k_gstep3: see the generator's docstring.
#pragma once
typedef float g4_f32x32 __attribute__((ext_vector_type(32)));
typedef float g4_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned g4_u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned g4_u32x4 __attribute__((ext_vector_type(4)));
// NS = 32-column slices per wave.  GAcc: a[m] is pinned to a[8 NS m : 8 NS m + 8 NS - 1]: the accumulators of row block m (2 NS column blocks x 4);
// G4W: the weight fragments of one K step; G4Rows / G4Mask: a chunk's staging rows (16 bytes each) and relu bytes of one thread
template <int NS> struct GAcc;   template <> struct GAcc<4> { g4_f32x32 a[8]; };   template <> struct GAcc<2> { g4_f32x16 a[8]; };
template <int NS> struct G4W;    template <> struct G4W<4> { g4_f32x32 v; };       template <> struct G4W<2> { g4_f32x16 v; };
template <int NS> struct G4Rows; template <> struct G4Rows<4> { g4_f32x32 v; };    template <> struct G4Rows<2> { g4_f32x16 v; };
template <int NS> struct G4Mask; template <> struct G4Mask<4> { g4_u32x8 v; };     template <> struct G4Mask<2> { g4_u32x4 v; };
'''


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "morphsym_hgnn_amd", "csrc", "mshgnn_gemm_block.inc")
    parts = [HEADER]
    for g in (G4, G2):
        parts += [emit_init(g), emit_read(g), emit_chunk(g)]
    txt = "\n".join(parts) + "\n"
    with open(out, "w") as f:
        f.write(txt)
    print("wrote", out, len(txt), "bytes")


if __name__ == "__main__":
    main()
