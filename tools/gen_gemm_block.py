#!/usr/bin/env python3
"""Generates morphsym_hgnn_amd/csrc/mshgnn_gemm_block.inc: the MFMA block of the generic-width engine's job kernel k_gstep3 (mshgnn_gen.hip) as hand-scheduled
gfx950 asm -- ONE 128-wide K chunk of a wave's 128-window x 128-column output tile per statement: 256 v_mfma_f32_16x16x32_bf16 on 256 accumulator registers that
stay in a[0:255] for the whole kernel (the compiler tracks them as eight 32-register values pinned to their registers, as in mshgnn_wide_engine.inc).

Why asm: with 256 accumulators + operand buffers hipcc's schedule of this loop moves accumulators between the two register-file halves (1 456 moves) and
keeps one window fragment in flight; here the weight fragments of K step t + 1 (8 x global_load_dwordx4) and the window fragments of K step t + 1
(8 x ds_read_b128) are requested in the first MFMA gaps of K step t and have 64 MFMAs (~1 100 cycles) to land.

Layout (mshgnn_device.hpp): accumulator of (row block m < 8, column block cb = 2 slice + fb < 8) = a[32 m + 4 cb : +3]; D^T = W_frag (A operand) x X_frag (B operand).
Weight fragment (slice s, fb, K step t) of pack image p: p * 32768 + ((8 s + 4 fb + t) * 64 + lane) * 16 bytes.  Window fragment (row block m, K step t): LDS
abuf + 4096 m + (ao0 ^ 16 t).  Vector registers of the block: W buffers v[64:95] (passed in, K step 0) / v[96:127], window fragments v[128:159] / v[160:191].
"""
import os

W = [64, 96]
X = [128, 160]
RAW, RMK = 192, 224      # the next chunk's rows v[192:223] and relu bytes v[224:231] (handed back to the caller)


def acc(m, cb):
    b = 32 * m + 4 * cb
    return f"a[{b}:{b + 3}]"


def vr(b):
    return f"v[{b}:{b + 3}]"


def chunk_text():
    L = []
    a = L.append
    # scalar bases of the eight (slice, fb) fragment streams: %[wlo/whi] + (8 s + 4 fb) * 1024
    for cb in range(8):
        s, fb = cb >> 1, cb & 1
        a(f"s_add_u32 s{60 + 2 * cb}, %[wlo], {(8 * s + 4 * fb) * 1024}")
        a(f"s_addc_u32 s{61 + 2 * cb}, %[whi], 0")
    a("s_mov_b32 s78, %[rblo]")
    a("s_mov_b32 s79, %[rbhi]")
    a("s_mov_b32 s80, %[mblo]")
    a("s_mov_b32 s81, %[mbhi]")
    a("v_add_u32 %[va], %[abuf], %[ao0]")
    for m in range(8):
        a(f"ds_read_b128 {vr(X[0] + 4 * m)}, %[va] offset:{4096 * m}")
    for t in range(4):
        wb, xb = W[t & 1], X[t & 1]
        nwb, nxb = W[(t + 1) & 1], X[(t + 1) & 1]
        # K step 0: the window fragments just requested (the caller's own prefetch loads stay in flight); later steps: everything requested during the
        # previous K step has had 64 MFMAs to land
        a("s_waitcnt lgkmcnt(0)" if t == 0 else ("s_waitcnt vmcnt(16) lgkmcnt(0)" if t == 1 else "s_waitcnt vmcnt(0) lgkmcnt(0)"))
        pre = []
        if t == 3:
            # the NEXT chunk's K-step-0 fragments into the buffer the block was handed (free since K step 2): waited for before the block ends, so
            # the value the caller gets back is complete and it keeps no weight registers of its own across the block
            for cb in range(8):
                s_, fb = cb >> 1, cb & 1
                pre.append(f"s_add_u32 s{60 + 2 * cb}, %[nwlo], {(8 * s_ + 4 * fb) * 1024}")
                pre.append(f"s_addc_u32 s{61 + 2 * cb}, %[nwhi], 0")
                pre.append(f"global_load_dwordx4 {vr(W[0] + 4 * cb)}, %[lane16], s[{60 + 2 * cb}:{61 + 2 * cb}]")
        if t == 0:
            # the NEXT chunk's rows (8 x 16 bytes per lane: rows rr + 16 i of the tile) and their relu bytes, behind the K-step-1 fragments: K step 1 waits for
            # all but these sixteen, the later K steps find them landed
            post = []
            for i in range(8):
                if i:
                    post.append(f"s_add_u32 s76, s76, %[rs]")
                    post.append(f"v_add_u32 %[vt], s76, %[toff0]")
                else:
                    post.append("s_mov_b32 s76, 0")
                post.append(f"global_load_dwordx4 v[{RAW + 4 * i}:{RAW + 4 * i + 3}], " + ("%[vt]" if i else "%[toff0]") + ", s[78:79]")
                post.append(f"global_load_ubyte v{RMK + i}, %[toffm0], s[80:81] offset:{64 * i}")
        if t < 3:
            pre.append(f"v_xor_b32 %[vb], {16 * (t + 1)}, %[va]")
            for cb in range(8):
                pre.append(f"global_load_dwordx4 {vr(nwb + 4 * cb)}, %[lane16], s[{60 + 2 * cb}:{61 + 2 * cb}] offset:{1024 * (t + 1)}")
            for m in range(8):
                pre.append(f"ds_read_b128 {vr(nxb + 4 * m)}, %[vb] offset:{4096 * m}")
        if t == 0:
            pre += post
        for m in range(8):
            for cb in range(8):
                a(f"v_mfma_f32_16x16x32_bf16 {acc(m, cb)}, {vr(wb + 4 * cb)}, {vr(xb + 4 * m)}, {acc(m, cb)}")
                if pre:
                    a(pre.pop(0))
        assert not pre
    a("s_waitcnt vmcnt(0)")
    a("s_nop 15")
    a("s_nop 7")
    return L


def emit_chunk():
    s = ["// one K chunk: acc += W_chunk x A_tile.  w0: the eight weight fragments of K step 0 (requested by the caller, so that their latency hides under its staging);",
         "// wslice: address of this wave's slice 0, fragment 0 of the chunk's pack image; abuf: LDS byte offset of the A tile; ao0: this lane's first chunk offset;",
         "// lane16 = lane * 16",
         "// wnext: the same address for the NEXT chunk's pack image: its K-step-0 fragments are in w0 when the block returns",
         "// rows / rmask: the NEXT chunk's staging data (rows rr + 16 i of the tile, i < 8: 16 bytes at rbase + toff0 + i * rstride; relu byte at mbase + toffm0 + 64 i)",
         "__device__ __forceinline__ void g4_chunk(GAcc& r, g4_f32x32& w0, g4_f32x32& rows, g4_u32x8& rmask, const void* wslice, const void* wnext, const void* rbase, const void* mbase,",
         "                                         unsigned rstride, int toff0, int toffm0, unsigned abuf, int ao0, int lane16) {",
         "    const unsigned rblo = (unsigned)(unsigned long long)rbase, rbhi = (unsigned)((unsigned long long)rbase >> 32);",
         "    const unsigned mblo = (unsigned)(unsigned long long)mbase, mbhi = (unsigned)((unsigned long long)mbase >> 32);",
         "    const unsigned wlo = (unsigned)(unsigned long long)wslice, whi = (unsigned)((unsigned long long)wslice >> 32);",
         "    const unsigned nwlo = (unsigned)(unsigned long long)wnext, nwhi = (unsigned)((unsigned long long)wnext >> 32);",
         "    int va, vb, vt;",
         "    asm volatile("]
    for ins in chunk_text():
        s.append(f'        "{ins}\\n\\t"')
    s.append('        : [va] "=&v"(va), [vb] "=&v"(vb), [vt] "=&v"(vt), "+{v[64:95]}"(w0), "=&{v[192:223]}"(rows), "=&{v[224:231]}"(rmask), ' + ", ".join(f'"+{{a[{32 * m}:{32 * m + 31}]}}"(r.a[{m}])' for m in range(8)))
    s.append('        : [wlo] "s"(wlo), [whi] "s"(whi), [nwlo] "s"(nwlo), [nwhi] "s"(nwhi), [rblo] "s"(rblo), [rbhi] "s"(rbhi), [mblo] "s"(mblo), [mbhi] "s"(mbhi), [rs] "s"(rstride), [toff0] "v"(toff0), [toffm0] "v"(toffm0), [abuf] "s"(abuf), [ao0] "v"(ao0), [lane16] "v"(lane16)')
    s.append("        : " + ", ".join(f'"v{i}"' for i in range(96, 192)) + ", " + ", ".join(f'"s{i}"' for i in range(60, 82)) + ', "memory");')
    s.append("}")
    return "\n".join(s)


def emit_init():
    """All row blocks start at the bias values of their column blocks: b[cb] (4 floats per lane)."""
    s = ["// accumulators of every row block = the bias values of the wave's eight column blocks (b[2 s + fb]: this lane's four features)",
         "__device__ __forceinline__ void g4_init(GAcc& r, const g4_f32x4 (&b)[8]) {",
         "    asm volatile("]
    for m in range(8):
        for cb in range(8):
            for j in range(4):
                s.append(f'        "v_accvgpr_write_b32 a[{32 * m + 4 * cb + j}], %{8 + cb}\\n\\t"'.replace(f"%{8 + cb}", "%" + str(8 + cb) + f"_SUB{j}"))
    txt = "\n".join(s)
    # operands: 8 tuple outputs (%0..%7), then the 8 vectors; a 128-bit operand prints as v[n:n+3]: element j is addressed by splitting the vector into scalars instead
    return None


def emit_init_scalar():
    s = ["// accumulators of every row block = the bias values of the wave's eight column blocks: b[4 cb + j] = feature j of column block cb = 2 slice + fb for this lane",
         "__device__ __forceinline__ void g4_init(GAcc& r, const float (&b)[32]) {"]
    # two statements of 16 inputs each (an asm statement takes 30 operands): column blocks 0-3, then 4-7
    for half in range(2):
        s.append("    asm volatile(")
        for m in range(8):
            for cbl in range(4):
                cb = 4 * half + cbl
                for j in range(4):
                    s.append(f'        "v_accvgpr_write_b32 a[{32 * m + 4 * cb + j}], %{8 + 4 * cbl + j}\\n\\t"')
        outs = ", ".join(f'"{"=" if half == 0 else "+"}{{a[{32 * m}:{32 * m + 31}]}}"(r.a[{m}])' for m in range(8))
        ins = ", ".join(f'"v"(b[{16 * half + k}])' for k in range(16))
        s.append(f"        : {outs}")
        s.append(f"        : {ins});")
    s.append("}")
    return "\n".join(s)


def emit_read():
    s = ["// the 32 accumulators of row block M as eight 4-float groups (column block cb = 2 slice + fb)",
         "template <int M> __device__ __forceinline__ void g4_read(GAcc& r, float (&o)[32]) {"]
    for half in range(2):
        s.append("    asm volatile(")
        for k in range(16):
            s.append(f'        "v_accvgpr_read_b32 %{k}, a[%c17+{16 * half + k}]\\n\\t"')
        outs = ", ".join(f'"=v"(o[{16 * half + k}])' for k in range(16))
        s.append(f"        : {outs}")
        s.append('        : "{TUPLE}"(r.a[M]), "i"(32 * M));')
    s.append("}")
    txt = "\n".join(s)
    # the pinned tuple operand depends on M: one specialisation per row block
    out = ["template <int M> __device__ __forceinline__ void g4_read(GAcc& r, float (&o)[32]);"]
    for m in range(8):
        t = txt.replace("template <int M> __device__ __forceinline__ void g4_read(GAcc& r, float (&o)[32]) {", f"template <> __device__ __forceinline__ void g4_read<{m}>(GAcc& r, float (&o)[32]) {{")
        t = t.replace('"{TUPLE}"(r.a[M])', f'"{{a[{32 * m}:{32 * m + 31}]}}"(r.a[{m}])').replace('"i"(32 * M)', f'"i"({32 * m})')
        t = t.split("\n", 1)[1] if t.startswith("//") else t
        out.append(t)
    return "\n".join(out)


HEADER = '''// GENERATED by tools/gen_gemm_block.py -- do not edit.  The MFMA block of k_gstep3: see the generator's docstring.
#pragma once
typedef float g4_f32x32 __attribute__((ext_vector_type(32)));
typedef float g4_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned g4_u32x8 __attribute__((ext_vector_type(8)));
struct GAcc { g4_f32x32 a[8]; };      // a[m] is pinned to a[32 m : 32 m + 31]: the accumulators of row block m (8 column blocks x 4)
'''


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "morphsym_hgnn_amd", "csrc", "mshgnn_gemm_block.inc")
    txt = "\n".join([HEADER, emit_init_scalar(), emit_read(), emit_chunk()]) + "\n"
    with open(out, "w") as f:
        f.write(txt)
    print("wrote", out, len(txt), "bytes")


if __name__ == "__main__":
    main()
