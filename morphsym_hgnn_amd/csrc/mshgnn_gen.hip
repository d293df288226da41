// Generic-width MS-HGNN engine (any hidden multiple of 128, any node count / in-degree, 'add' and 'mean' aggregation): the path of
// mshgnn.hip (GRF_HGNN.forward hgnn.py:57-62, GRF_HGNN_C2.forward hgnn_c2.py:133-182, GRF_HGNN_K4.forward hgnn_k4.py:146-196 and
// their autograd backward) lowered to grouped block GEMMs by mshgnn_gen_plan.hpp, for the topologies / widths the LDS-resident
// stack kernels cannot hold -- BASELINE.json configs[4]: synthetic 32-limb robot, h = 512, 6 layers.
//
//   k_gstep    one JOB = (destination row, 64-window tile, 128-column tile): K loop over the job's TERMS; a term's A tile
//              [64 windows x 128] is the fp32 sum (mean: scaled) of its source rows, masked by relu bytes where the source is a
//              dH = dX . relu bits, staged into LDS; B = a packed 128x128 weight image straight from L2 into registers; bf16 MFMA
//              16x16x32, fp32 accumulate; bias / relu / relu-byte stash / residual / gating fused in the epilogue.  The encoder
//              (raw fp32 or bf16 input rows, symmetry sign XOR), every HeteroConv layer, the base_transform MLP and every backward
//              layer are launches of this one kernel with different job tables.
//   k_ggradw   split-K weight gradients, one 128x128 tile of one target per workgroup over a chunk of the target's items and a
//              part of the batch: dW = P^T Q with Q aggregated like a term's A tile; deterministic slab sums in k_gfinalize.
//   SPLIT      the split-bf16 parity arithmetic of mshgnn_x3.hip (hi + lo bf16 halves, three MFMA products per term, fp32
//              inputs, activation rows stored [hi Hd | lo Hd]): 1e-4 relative; the non-split instantiation is the bf16 plan.
#include <type_traits>
#include "mshgnn_device.hpp"
#include "mshgnn_gen_plan.hpp"

using namespace mshgnn::gen;
using T16 = __bf16;
using P16 = Prec<__bf16>;

struct mshgnn_gen_state {      // device side of a generic plan
    GenPlan gp;
    int* d_tables = nullptr; uint8_t* d_signs = nullptr; float* d_out_mask = nullptr; PackDesc* d_packs = nullptr; BiasDesc* d_biases = nullptr;
};

struct GArgs {
    char* ws; size_t buf_off[GBUF_COUNT];
    const void* x[MSHGNN_MAX_TYPES]; int64_t pitch[MSHGNN_MAX_TYPES]; int nodes[MSHGNN_MAX_TYPES]; int vb[MSHGNN_MAX_TYPES]; int aligned;
    const int* jobs; const int* terms; const int* srcs; const int* units; const int* items; const int* sunits; const int* su_order; const int* aggs;
    const void* wpack; const float* bias; const uint8_t* signs; float* slabs;
    int n_img, B, Hd, NCT, tiles, training, job0, n_units, n_parts, n_sunits, njt;      // njt: (job, tile) pairs of a k_gstep5 launch at 4 waves
#ifdef GGW_STAMPS
    long long* stamps;      // (phase clocks of k_ggradw, tools/stamps_ggradw.py)
#endif
#ifdef GEN_TIMELINE
    long long* tl;          // (start / end wall clock and hardware id of every workgroup of a job or weight-gradient launch, tools/timeline_gen.py)
#endif
};
#ifdef GEN_TIMELINE
#define GEN_TL(k) do { if (a.tl && threadIdx.x == 0) { a.tl[(size_t)blockIdx.x * 4 + (k)] = wall_clock64(); if ((k) == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); a.tl[(size_t)blockIdx.x * 4 + 2] = id; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); a.tl[(size_t)blockIdx.x * 4 + 3] = xcc; } } } while (0)
#else
#define GEN_TL(k) do { } while (0)
#endif

// element index of (window w, node, column 0) in an activation tensor: rows are Hd wide, or [hi Hd | lo Hd] on the split plan
template <bool SPLIT> __device__ __forceinline__ size_t g_row(int w, int node, int B, int Hd) { return ((size_t)node * B + w) * (SPLIT ? 2 * Hd : Hd); }
// relu bytes: one per (node, window, 8 features): [NN][Hd/32 column slices][ceil(B/16) tiles][4 groups][16 windows] (relu_byte of mshgnn_device.hpp at Hd = 128)
__device__ __forceinline__ size_t g_relu_byte(int n, int B, int Hd, int w, int f) {
    return ((((size_t)n * (Hd >> 5) + (f >> 5)) * ((B + 15) >> 4) + (w >> 4)) << 6) + (((f >> 3) & 3) << 4) + (w & 15);
}
__device__ __forceinline__ void acc8(float (&s)[8], u32x4 v, float scale) {      // s += scale * (8 bf16 of v)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[2 * e] += scale * __builtin_bit_cast(float, v[e] << 16);
        s[2 * e + 1] += scale * __builtin_bit_cast(float, v[e] & 0xffff0000u);
    }
}

// the 8 elements [col, col + 8) of the aggregated source row of window w (fp32): sum_s scale_s . mask_s . X_s[w]
template <bool SPLIT>
__device__ __forceinline__ void gather8(const GArgs& a, const int* src, int n_src, int w, int col, float (&s)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int k = 0; k < n_src; ++k, src += SRC_INTS) {
        const T16* base = reinterpret_cast<const T16*>(a.ws + a.buf_off[src[S_BUF]]) + g_row<SPLIT>(w, src[S_NODE], a.B, a.Hd) + col;
        u32x4 vh = *reinterpret_cast<const u32x4*>(base), vl = u32x4{0, 0, 0, 0};
        if constexpr (SPLIT) vl = *reinterpret_cast<const u32x4*>(base + a.Hd);
        if (src[S_MASK] >= 0) {      // dH = dX . relu bits
            const unsigned byte = reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[src[S_MASK]])[g_relu_byte(src[S_NODE], a.B, a.Hd, w, col)];
            vh = chunk_mask_bits<T16>(vh, byte);
            if constexpr (SPLIT) vl = chunk_mask_bits<T16>(vl, byte);
        }
        const float sc = __int_as_float(src[S_SCALE]);
        acc8(s, vh, sc);
        if constexpr (SPLIT) acc8(s, vl, sc);
    }
}
// the same for a raw input row (type t, node): fp32 (split plan) or bf16 elements [k0, k0 + 8) with pad columns zeroed and the symmetry sign applied
template <bool SPLIT>
__device__ __forceinline__ void raw8(const GArgs& a, int t, int node, int w, int k0, int F, const uint8_t* sg, float (&s)[8]) {
    const int nv = F - k0;
    if constexpr (SPLIT) {
        const float* p = reinterpret_cast<const float*>(a.x[t]) + ((size_t)w * a.nodes[t] + node) * a.pitch[t] + k0;
        const u32x4 fa = chunk_keep_first<float>(load_chunk<float>(p, nv, a.vb[t]), nv) ^ sign_xor<float>(sg);
        const u32x4 fb = chunk_keep_first<float>(load_chunk<float>(p + 4, nv - 4, a.vb[t]), nv - 4) ^ sign_xor<float>(sg + 4);
        const f32x4 va = __builtin_bit_cast(f32x4, fa), vb4 = __builtin_bit_cast(f32x4, fb);      // (whole-vector casts: hipcc 7.2 miscompiles __builtin_bit_cast of a single vector element)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] = va[e]; s[4 + e] = vb4[e]; }
    } else {
        const T16* p = reinterpret_cast<const T16*>(a.x[t]) + ((size_t)w * a.nodes[t] + node) * a.pitch[t] + k0;
        const u32x4 v = chunk_keep_first<T16>(load_chunk<T16>(p, nv, a.vb[t]), nv) ^ sign_xor<T16>(sg);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        acc8(s, v, 1.0f);
    }
}

// epilogue of a job tile: this lane owns 8 consecutive output features of one window per row block (bias already in the accumulators)
template <bool SPLIT, int MB>
__device__ __forceinline__ void gstep_epilogue(const GArgs& a, const int* job, P16::Acc (&acc)[MB], int ct, int wv, int lane, int w0) {
    using P = P16;
    const int B = a.B, Hd = a.Hd, flags = job[J_FLAGS];
    const int col = ct * TW + wv * 32 + c_oct(lane);
    T16* out = reinterpret_cast<T16*>(a.ws + a.buf_off[job[J_OUT_BUF]]);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int wb = w0 + m * P::ROWS, w = wb + c_win(lane);
        if (wb >= B) continue;
        const int wc = min(w, B - 1);
        if (flags & JF_GATE_POS) {      // dU = dT1 . (T1 > 0): the hi half carries the sign
            f32x4 g0, g1;
            load_oct(reinterpret_cast<const T16*>(a.ws + a.buf_off[job[J_GATE_BUF]]) + g_row<SPLIT>(wc, job[J_GATE_NODE], B, Hd) + col, g0, g1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[m].c[0][j] = g0[j] > 0.f ? acc[m].c[0][j] : 0.f; acc[m].c[1][j] = g1[j] > 0.f ? acc[m].c[1][j] : 0.f; }
        }
        if (flags & JF_RELU) {
            const unsigned bits = relu_with_bits<T16>(acc[m]);
            if ((flags & JF_BITS_OUT) && a.training)      // (rows past the batch inside the last 16-window block land in the buffer's padding)
                reinterpret_cast<uint8_t*>(a.ws + a.buf_off[job[J_BITS_BUF]])[g_relu_byte(job[J_OUT_NODE], B, Hd, w, col)] = (uint8_t)bits;
        }
        f32x4 y0 = acc[m].c[0], y1 = acc[m].c[1];
        if (flags & JF_RES) {
            const T16* r = reinterpret_cast<const T16*>(a.ws + a.buf_off[job[J_RES_BUF]]) + g_row<SPLIT>(wc, job[J_RES_NODE], B, Hd) + col;
            f32x4 r0v, r1v;
            if constexpr (SPLIT) join_oct(*reinterpret_cast<const u32x4*>(r), *reinterpret_cast<const u32x4*>(r + Hd), r0v, r1v);
            else load_oct(r, r0v, r1v);
            y0 += r0v; y1 += r1v;
        }
        if (flags & JF_GATE_BITS) {     // layer 0 of the backward pass: x relu'(X_0) from the encoder's relu bytes
            const unsigned xb = reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[job[J_GATE_BUF]])[g_relu_byte(job[J_GATE_NODE], B, Hd, wc, col)];
#pragma unroll
            for (int j = 0; j < 4; ++j) { y0[j] = ((xb >> j) & 1u) ? y0[j] : 0.f; y1[j] = ((xb >> (4 + j)) & 1u) ? y1[j] : 0.f; }
        }
        if (w < B) {
            T16* q = out + g_row<SPLIT>(w, job[J_OUT_NODE], B, Hd) + col;
            if constexpr (SPLIT) {
                u32x4 hi, lo;
                split_oct(y0, y1, hi, lo);
                *reinterpret_cast<u32x4*>(q) = hi;
                *reinterpret_cast<u32x4*>(q + Hd) = lo;
            } else store_oct(q, y0, y1);
        }
    }
}

// element i of a read-only int table at a UNIFORM address, as a scalar load whatever stores and barriers lie in between (constant address space)
__device__ __forceinline__ int ro_int(const int* p, int i) {
    typedef const int __attribute__((address_space(4))) cint;
    const uint64_t u = reinterpret_cast<uint64_t>(p + i);
    const uint64_t v = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    return *reinterpret_cast<cint*>(v);
}
// this lane's index, recomputed where it is used (two VALU operations): a thread index kept live across k_gstep5's loop is spilled, and a scratch reload waits for every request in flight
__device__ __forceinline__ int lane_now() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; }
// Two-phase epilogue of a job tile (bf16 arithmetic, NS 32-column slices per wave): EVERY request of the tile first, then the arithmetic and the stores.
// gstep_epilogue's request -> store per row block is one memory round trip per block (the store may alias the next block's rows, so hipcc keeps their order):
// 15-25 k clocks per slice of eight blocks, 20-45 k for k_gstep5's two -- a third of its workgroup's time.
// FS >= 0: the job's flags (without JF_BIAS, which the accumulators already hold) as a compile-time constant -- the flag tests fold away and the tile's requests are one
// straight run; with run-time flags every test is a branch, and at every join hipcc's wait insertion takes the stricter count: the "requests first" phase then waits
// request by request (7 300 lines of ISA and 12.7 k clocks for the two slices of k_gstep5).  FS < 0: run-time flags (any combination).
template <bool SPLIT, int MB, int NS, int FS>
__device__ __forceinline__ void gstep_epilogue2_impl(const GArgs& a, const int* job, P16::Acc (&acc)[NS][MB], int ct, int wv0, int lane, int w0, int rt_flags) {
    using P = P16;
    const int B = a.B, Hd = a.Hd, flags = FS >= 0 ? FS : rt_flags;
    // (the job record through the constant address space: behind the K loop's barriers plain loads of it are VECTOR loads, each awaited with vmcnt(0) -- one round trip per row block again)
    const int j_out_buf = ro_int(job, J_OUT_BUF), j_out_node = ro_int(job, J_OUT_NODE), j_res_buf = ro_int(job, J_RES_BUF), j_res_node = ro_int(job, J_RES_NODE),
              j_bits_buf = ro_int(job, J_BITS_BUF), j_gate_buf = ro_int(job, J_GATE_BUF), j_gate_node = ro_int(job, J_GATE_NODE), j_dhm_buf = ro_int(job, J_DHM_BUF);
    if ((flags & JF_GATE_POS) && (flags & JF_RES)) {      // (no job of the plans has both; they would share the request registers)
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) gstep_epilogue<SPLIT, MB>(a, job, acc[sl], ct, wv0 + sl, lane, w0);
        return;
    }
    u32x4 aux[NS][MB], auxl[SPLIT ? NS : 1][SPLIT ? MB : 1]; unsigned xb[NS][MB], hb[NS][MB];      // hb: the relu bytes of a JF_DHM job's second output
    const bool want_aux = (flags & (JF_GATE_POS | JF_RES)) != 0, want_lo = SPLIT && (flags & JF_RES) != 0;      // (the gate reads the hi half only: it carries the sign)
    const char* auxb = nullptr; const uint8_t* xbb = nullptr;
    if (flags & JF_GATE_POS) auxb = a.ws + a.buf_off[j_gate_buf] + g_row<SPLIT>(0, j_gate_node, B, Hd) * 2;
    if (flags & JF_RES) auxb = a.ws + a.buf_off[j_res_buf] + g_row<SPLIT>(0, j_res_node, B, Hd) * 2;
    if (flags & JF_GATE_BITS) xbb = reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[j_gate_buf]);
    const uint8_t* hbb = (flags & JF_DHM) ? reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[j_bits_buf]) : nullptr;
    constexpr int RS = SPLIT ? 2 : 1;      // row stride in units of Hd elements
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int wc = min(w0 + m * P::ROWS + c_win(lane), B - 1), col = ct * TW + (wv0 + sl) * 32 + c_oct(lane);
            if (want_aux) aux[sl][m] = *reinterpret_cast<const u32x4*>(auxb + ((size_t)wc * (RS * Hd) + col) * 2);
            if constexpr (SPLIT) { if (want_lo) auxl[sl][m] = *reinterpret_cast<const u32x4*>(auxb + ((size_t)wc * (RS * Hd) + Hd + col) * 2); }
            if (flags & JF_GATE_BITS) xb[sl][m] = xbb[g_relu_byte(j_gate_node, B, Hd, wc, col)];
            if (flags & JF_DHM) hb[sl][m] = hbb[g_relu_byte(j_out_node, B, Hd, wc, col)];
        }
    T16* out = reinterpret_cast<T16*>(a.ws + a.buf_off[j_out_buf]);
    T16* dhm = (flags & JF_DHM) ? reinterpret_cast<T16*>(a.ws + a.buf_off[j_dhm_buf]) : nullptr;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int wb = w0 + m * P::ROWS, w = wb + c_win(lane), col = ct * TW + (wv0 + sl) * 32 + c_oct(lane);
            if (wb >= B) continue;
            P::Acc& ac = acc[sl][m];
            f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (want_aux) {
                if constexpr (SPLIT) { if (want_lo) join_oct(aux[sl][m], auxl[sl][m], x0, x1); else unpack_oct(aux[sl][m], x0, x1); }
                else unpack_oct(aux[sl][m], x0, x1);
            }
            if (flags & JF_GATE_POS) {      // dU = dT1 . (T1 > 0)
#pragma unroll
                for (int q = 0; q < 4; ++q) { ac.c[0][q] = x0[q] > 0.f ? ac.c[0][q] : 0.f; ac.c[1][q] = x1[q] > 0.f ? ac.c[1][q] : 0.f; }
            }
            if (flags & JF_RELU) {
                const unsigned bits = relu_with_bits<T16>(ac);
                if ((flags & JF_BITS_OUT) && a.training)      // (rows past the batch inside the last 16-window block land in the buffer's padding)
                    reinterpret_cast<uint8_t*>(a.ws + a.buf_off[j_bits_buf])[g_relu_byte(j_out_node, B, Hd, w, col)] = (uint8_t)bits;
            }
            f32x4 y0 = ac.c[0], y1 = ac.c[1];
            if (flags & JF_RES) { y0 += x0; y1 += x1; }
            if (flags & JF_GATE_BITS) {     // layer 0 of the backward pass: x relu'(X_0) from the encoder's relu bytes
                const unsigned b8 = xb[sl][m];
#pragma unroll
                for (int q = 0; q < 4; ++q) { y0[q] = ((b8 >> q) & 1u) ? y0[q] : 0.f; y1[q] = ((b8 >> (4 + q)) & 1u) ? y1[q] : 0.f; }
            }
            if (w < B) {
                T16* q = out + g_row<SPLIT>(w, j_out_node, B, Hd) + col;
                if constexpr (SPLIT) {
                    u32x4 hi, lo;
                    split_oct(y0, y1, hi, lo);
                    if (!(flags & JF_DHM_ONLY)) {
                        *reinterpret_cast<u32x4*>(q) = hi;
                        *reinterpret_cast<u32x4*>(q + Hd) = lo;
                    }
                    if (flags & JF_DHM) {      // the same row with the relu bits of the layer below applied: that layer's dH
                        T16* q2 = dhm + g_row<SPLIT>(w, j_out_node, B, Hd) + col;
                        *reinterpret_cast<u32x4*>(q2) = chunk_mask_bits<T16>(hi, hb[sl][m]);
                        *reinterpret_cast<u32x4*>(q2 + Hd) = chunk_mask_bits<T16>(lo, hb[sl][m]);
                    }
                } else {
                    const u32x4 yv = pack_oct(y0, y1);
                    if (!(flags & JF_DHM_ONLY)) *reinterpret_cast<u32x4*>(q) = yv;
                    if (flags & JF_DHM) *reinterpret_cast<u32x4*>(dhm + g_row<SPLIT>(w, j_out_node, B, Hd) + col) = chunk_mask_bits<T16>(yv, hb[sl][m]);
                }
            }
        }
}

template <bool SPLIT, int MB, int NS, bool SPECIALISE = false>
__device__ __forceinline__ void gstep_epilogue2(const GArgs& a, const int* job, P16::Acc (&acc)[NS][MB], int ct, int wv0, int lane, int w0) {
    const int flags = ro_int(job, J_FLAGS) & ~JF_BIAS;
    if constexpr (SPECIALISE) {      // the flag sets of the plans' layer launches (k_gstep5); anything else takes the run-time form
        switch (flags) {
#define GS_EPI_CASE(F) case (F): gstep_epilogue2_impl<SPLIT, MB, NS, (F)>(a, job, acc, ct, wv0, lane, w0, flags); return;
            GS_EPI_CASE(JF_RELU | JF_BITS_OUT)
            GS_EPI_CASE(JF_RELU | JF_BITS_OUT | JF_RES)
            GS_EPI_CASE(JF_DHM | JF_DHM_ONLY)
            GS_EPI_CASE(JF_DHM | JF_RES)
            GS_EPI_CASE(JF_GATE_BITS)
            GS_EPI_CASE(JF_GATE_BITS | JF_RES)
            GS_EPI_CASE(0)
#undef GS_EPI_CASE
            default: break;
        }
    }
    gstep_epilogue2_impl<SPLIT, MB, NS, -1>(a, job, acc, ct, wv0, lane, w0, flags);
}

// ------------------------------------------------------------------------------------------------------
// k_gstep
// ------------------------------------------------------------------------------------------------------
// MB: 16-window row blocks per workgroup; NW: waves -- wave wv owns the 32-column slice (wv & 3) of the 128-column pack tile ctg * (NW / 4) + (wv >> 2),
// so a workgroup's output tile is MB * 16 windows x NW * 32 columns and the staged A tile is shared by all NW waves
// Measured and left off (round 2, h = 512, B = 1024, us per layer launch):  GSTEP_W_EARLY=1 (the chunk's weight fragment requested before its A rows
// are gathered instead of after the barrier) 297 vs 286 bf16, 829 vs 814 split;  a software-pipelined variant of this kernel (8 waves at two per SIMD
// and up to 256 VGPRs, two A buffers in LDS, the next chunk's weight fragment and source rows in flight under the MFMAs, one barrier per chunk;
// bit-identical results) 399 (128-window tiles) / 470 us bf16 against 287 for the 16-wave kernel below, 795 against 810 us split: with two waves per
// SIMD one MFMA phase (~0.45 us) is too short a prefetch distance for the L2 / HBM latency of the rows, and the occupancy it costs hid more.
#ifndef GSTEP_W_EARLY
#define GSTEP_W_EARLY 0
#endif
template <bool SPLIT, int MB, int NW> __global__ __launch_bounds__(64 * NW) void k_gstep(GArgs a) {
    using P = P16;
    constexpr int CPW = NW / 4;                          // 128-column pack tiles per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];      // blocks [0, MB): (hi) A tile; [MB, 2 MB): lo halves (split plan)
    const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nctg = a.NCT / CPW;
    const int ct = (blockIdx.x % nctg) * CPW + (wq >> 2), wv = wq & 3, tile = (blockIdx.x / nctg) % a.tiles;
    const int* job = a.jobs + (size_t)(a.job0 + blockIdx.x / (nctg * a.tiles)) * JOB_INTS;
    const int w0 = tile * MB * P::ROWS, B = a.B, Hd = a.Hd;
    const int flags = job[J_FLAGS];
    const T16* wpack = reinterpret_cast<const T16*>(a.wpack);

    P::Acc acc[MB];
    {
        const float* bias = (flags & JF_BIAS) ? a.bias + ((size_t)job[J_BIAS] + ct) * TW : nullptr;
#pragma unroll
        for (int m = 0; m < MB; ++m) acc_init_bias<T16>(acc[m], bias, wv, lane);
    }
    const int c = tid & 15, rr = tid >> 4;              // staging: thread = (row rr + 4 NW i of the tile, 8-element chunk c), i = 0 .. NPASS - 1
    constexpr int NPASS = MB * 16 / (4 * NW);
    static_assert(NPASS >= 1 && NPASS * 4 * NW == MB * 16, "row blocks must cover whole staging passes");
    const AOff<T16> ao(lane);
    P::BFrag bfh, bfl;
    P::AFrag af;
    const int* term = a.terms + (size_t)job[J_TERM0] * TERM_INTS;
    for (int ti = 0; ti < job[J_NTERMS]; ++ti, term += TERM_INTS) {
        const int nkc = term[T_NKC], kind = term[T_KIND], n_src = term[T_NSRC], F = term[T_WIDTH];
        const int* src = a.srcs + (size_t)term[T_SRC0] * SRC_INTS;
        for (int kc = 0; kc < nkc; ++kc) {
#if GSTEP_W_EARLY
            // the chunk's weight fragment first: it needs no LDS, and its L2 latency then runs under the A rows' instead of after them
            const int pack = term[T_PACK] + kc * a.NCT + ct;
            load_bfrag<T16>(bfh, wpack, pack, wv, lane);
            if constexpr (SPLIT) load_bfrag<T16>(bfl, wpack, a.n_img + pack, wv, lane);
#endif
            __syncthreads();   // the previous chunk's MFMAs are done reading LDS
#pragma unroll
            for (int i0 = 0; i0 < NPASS; i0 += 4) {      // four rows at a time: their loads are in flight together
                constexpr int NB = NPASS < 4 ? NPASS : 4;
                float s[NB][8];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int w = min(w0 + (i0 + i) * (4 * NW) + rr, B - 1);      // rows past the batch re-read the last window; they are never stored
                    if (kind == 0) gather8<SPLIT>(a, src, n_src, w, kc * TW + c * 8, s[i]);
                    else raw8<SPLIT>(a, src[S_BUF], src[S_NODE], w, kc * TW + c * 8, F, a.signs + term[T_SIGN] + kc * TW + c * 8, s[i]);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int grow = (i0 + i) * (4 * NW) + rr, m = grow >> 4, r0 = grow & 15;
                    const f32x4 lo4 = f32x4{s[i][0], s[i][1], s[i][2], s[i][3]}, hi4 = f32x4{s[i][4], s[i][5], s[i][6], s[i][7]};
                    if constexpr (SPLIT) {
                        u32x4 hi, lo;
                        split_oct(lo4, hi4, hi, lo);
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(m, r0, c)) = hi;
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(MB + m, r0, c)) = lo;
                    } else *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(m, r0, c)) = pack_oct(lo4, hi4);
                }
            }
            __syncthreads();
#if !GSTEP_W_EARLY
            const int pack = term[T_PACK] + kc * a.NCT + ct;
            load_bfrag<T16>(bfh, wpack, pack, wv, lane);
            if constexpr (SPLIT) load_bfrag<T16>(bfl, wpack, a.n_img + pack, wv, lane);
#endif
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (w0 + m * P::ROWS < B) {   // uniform
                    load_afrag<T16>(af, smem, m, ao);
                    mac(acc[m], af, bfh);
                    if constexpr (SPLIT) {
                        mac(acc[m], af, bfl);
                        load_afrag<T16>(af, smem, MB + m, ao);
                        mac(acc[m], af, bfh);
                    }
                }
            }
        }
    }
    gstep_epilogue2<SPLIT, MB, 1>(a, job, reinterpret_cast<P::Acc (&)[1][MB]>(acc), ct, wv, lane, w0);      // (requests of the whole tile first: see gstep_epilogue2)
}

// ------------------------------------------------------------------------------------------------------
// k_gstep4 (bf16 arithmetic): k_gstep's 16 waves on 128-window tiles
// ------------------------------------------------------------------------------------------------------
// k_gstep is bound by its weight stream (8 KB of fragments per wave and K chunk through L1 / L2: 2.2 GB per layer launch at h = 512, 7.6 TB/s); a weight
// fragment that feeds 8 row blocks instead of 4 halves it.  Round 2's 128-window variant of k_gstep spilled at the 128 registers 16 waves leave (389 us against
// 287); this kernel keeps the registers down instead: single-source chunks are staged as raw 16-byte chunks (mask applied on the packed value, no fp32 round trip),
// other chunks one row at a time, and the window fragments are read per K step (4 registers) instead of per row block (16).
// SPLIT (the split-bf16 arithmetic of section 4b: rows [hi Hd | lo Hd], three products per term) runs it on 64-window tiles (MB = 4): with the same register diet
// its 16-wave form fits the 128 registers that k_gstep<true, 4, 16> overflowed, so the A tile is staged once for 512 columns instead of twice.
template <bool SPLIT, int MB, int NW> __global__ __launch_bounds__(64 * NW) void k_gstep4(GArgs a) {
    using P = P16;
    constexpr int CPW = NW / 4, NPASS = MB * 16 / (4 * NW);      // staging passes: thread = (row rr + 4 NW i, chunk c)
    static_assert(NPASS >= 1 && NPASS * 4 * NW == MB * 16, "row blocks must cover whole staging passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // blocks [0, MB): (hi) A tile; [MB, 2 MB): lo halves (split arithmetic)
    const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nctg = a.NCT / CPW;
    const int ct = (blockIdx.x % nctg) * CPW + (wq >> 2), wv = wq & 3, tile = (blockIdx.x / nctg) % a.tiles;
    const int* job = a.jobs + (size_t)(a.job0 + blockIdx.x / (nctg * a.tiles)) * JOB_INTS;
    const int w0 = tile * MB * P::ROWS, B = a.B, Hd = a.Hd;
    const int flags = job[J_FLAGS];
    const T16* wpack = reinterpret_cast<const T16*>(a.wpack);
    GEN_TL(0);
#ifdef GGW_STAMPS
    long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64(); int nchunk = 0;      // (phase clocks, tools/stamps_gstep4.py)
#define GS_T(k) { const long long t1 = clock64(); tk[k] += t1 - t0; t0 = t1; }
#else
#define GS_T(k)
#endif

    P::Acc acc[MB];
    {
        const float* bias = (flags & JF_BIAS) ? a.bias + ((size_t)job[J_BIAS] + ct) * TW : nullptr;
#pragma unroll
        for (int m = 0; m < MB; ++m) acc_init_bias<T16>(acc[m], bias, wv, lane);
    }
    const int c = tid & 15, rr = tid >> 4;
    const int ao0 = lds_chunk<T16>(0, lane & 15, (lane >> 4) * P::NAV);
    const int one_bits = __float_as_int(1.0f);
    P::BFrag bfh, bfl;
    const int* term = a.terms + (size_t)job[J_TERM0] * TERM_INTS;
    for (int ti = 0; ti < job[J_NTERMS]; ++ti, term += TERM_INTS) {
        const int nkc = term[T_NKC], kind = term[T_KIND], n_src = term[T_NSRC], F = term[T_WIDTH];
        const int* src = a.srcs + (size_t)term[T_SRC0] * SRC_INTS;
        const bool plain = kind == 0 && n_src == 1 && src[S_SCALE] == one_bits;
        for (int kc = 0; kc < nkc; ++kc) {
            __syncthreads();   // the previous chunk's MFMAs are done reading LDS
            GS_T(0)
            if (plain) {
                const T16* base = reinterpret_cast<const T16*>(a.ws + a.buf_off[src[S_BUF]]);
                const bool msk = src[S_MASK] >= 0;
                const uint8_t* mb = reinterpret_cast<const uint8_t*>(a.ws + (msk ? a.buf_off[src[S_MASK]] : 0));
                const int col = kc * TW + c * 8;
                constexpr int NIF = NPASS < 2 ? NPASS : 2;      // rows in flight per thread (two: the registers 16 waves, or two workgroups of 8, leave)
#pragma unroll
                for (int i0 = 0; i0 < NPASS; i0 += NIF) {
                    u32x4 v[NIF], vl[SPLIT ? NIF : 1]; unsigned bm[NIF];
#pragma unroll
                    for (int i = 0; i < NIF; ++i) {
                        const int w = min(w0 + (i0 + i) * (4 * NW) + rr, B - 1);
                        const T16* rp = base + g_row<SPLIT>(w, src[S_NODE], B, Hd) + col;
                        v[i] = *reinterpret_cast<const u32x4*>(rp);
                        if constexpr (SPLIT) vl[i] = *reinterpret_cast<const u32x4*>(rp + Hd);
                        bm[i] = msk ? mb[g_relu_byte(src[S_NODE], B, Hd, w, col)] : 0xffu;
                    }
#ifdef GGW_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    GS_T(1)
#endif
#pragma unroll
                    for (int i = 0; i < NIF; ++i) {
                        const int grow = (i0 + i) * (4 * NW) + rr;
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(grow >> 4, grow & 15, c)) = chunk_mask_bits<T16>(v[i], bm[i]);
                        if constexpr (SPLIT) *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(MB + (grow >> 4), grow & 15, c)) = chunk_mask_bits<T16>(vl[i], bm[i]);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPASS; ++i) {
                    float s8[8];
                    const int w = min(w0 + i * (4 * NW) + rr, B - 1);
                    if (kind == 0) gather8<SPLIT>(a, src, n_src, w, kc * TW + c * 8, s8);
                    else raw8<SPLIT>(a, src[S_BUF], src[S_NODE], w, kc * TW + c * 8, F, a.signs + term[T_SIGN] + kc * TW + c * 8, s8);
                    const int grow = i * (4 * NW) + rr;
                    const f32x4 lo4 = f32x4{s8[0], s8[1], s8[2], s8[3]}, hi4 = f32x4{s8[4], s8[5], s8[6], s8[7]};
                    if constexpr (SPLIT) {
                        u32x4 hi, lo;
                        split_oct(lo4, hi4, hi, lo);
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(grow >> 4, grow & 15, c)) = hi;
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(MB + (grow >> 4), grow & 15, c)) = lo;
                    } else *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(grow >> 4, grow & 15, c)) = pack_oct(lo4, hi4);
                }
            }
            __syncthreads();
            GS_T(2)
            const int pack = term[T_PACK] + kc * a.NCT + ct;
            load_bfrag<T16>(bfh, wpack, pack, wv, lane);
            if constexpr (SPLIT) load_bfrag<T16>(bfl, wpack, a.n_img + pack, wv, lane);
#ifdef GGW_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GS_T(3)
            ++nchunk;
#endif
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                // (the order of mac() in k_gstep: hi x hi over the four K steps, then hi x lo-weights, then lo x hi)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + m * P::BLK + (ao0 ^ (16 * t))));
                    acc[m].c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh.v[t], xf, acc[m].c[0], 0, 0, 0);
                    acc[m].c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh.v[4 + t], xf, acc[m].c[1], 0, 0, 0);
                }
                if constexpr (SPLIT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + m * P::BLK + (ao0 ^ (16 * t))));
                        acc[m].c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl.v[t], xf, acc[m].c[0], 0, 0, 0);
                        acc[m].c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl.v[4 + t], xf, acc[m].c[1], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + (MB + m) * P::BLK + (ao0 ^ (16 * t))));
                        acc[m].c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh.v[t], xf, acc[m].c[0], 0, 0, 0);
                        acc[m].c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh.v[4 + t], xf, acc[m].c[1], 0, 0, 0);
                    }
                }
            }
#ifdef GGW_STAMPS
            { float d; asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(acc[MB - 1].c[1][3])); asm volatile("" :: "v"(d)); }
            GS_T(4)
#endif
        }
    }
    gstep_epilogue2<SPLIT, MB, 1>(a, job, reinterpret_cast<P::Acc (&)[1][MB]>(acc), ct, wv, lane, w0);      // (requests of the whole tile first; the JF_DHM second output)
#ifdef GGW_STAMPS
    GS_T(5)
    if (a.stamps && tid == 0) { for (int q = 0; q < 6; ++q) a.stamps[(size_t)blockIdx.x * 8 + q] = tk[q]; a.stamps[(size_t)blockIdx.x * 8 + 6] = nchunk; a.stamps[(size_t)blockIdx.x * 8 + 7] = job[J_NTERMS]; }
#endif
    GEN_TL(1);
}

// ------------------------------------------------------------------------------------------------------
// k_gstep5 (bf16 arithmetic): k_gstep4's tile (128 windows x 512 columns) as a software pipeline of 8 waves
// ------------------------------------------------------------------------------------------------------
// k_gstep4's K chunk is four phases in a row -- wait for the source rows (1.8-2.9 k clocks), LDS writes + barrier (1.2 k), wait for the weight fragment
// (1.0-1.9 k), MFMAs (2.5 k + 3.7 k at the next barrier while the SIMD's other waves finish theirs): 12.2 k clocks per chunk against 4.1 k of MFMA work
// (phase clocks, tools/stamps_gstep4.py).  Here the rows of chunk k + 1 are requested before the MFMAs of chunk k and written to the OTHER tile set after them
// (two sets in LDS, one barrier per chunk), and the weight fragment is streamed: the K loop is the outer one (K step t feeds all eight row blocks), so the two
// vectors of step t are dead after it and chunk k + 1's are requested into the same registers right there, three K steps ahead of their use.  No more registers
// than k_gstep4 + the 10 of the rows in flight.  Every accumulator still receives its products in k_gstep4's order (chunks in order, K steps 0..3): identical bits.
// Chunks that are not one plain row (sums of two rows, raw inputs) are gathered at staging time as before.
constexpr int GS5_MAX_TERMS = 16, GS5_TERM_BYTES = 32;      // k_gstep5's term table in LDS (behind the two tile sets)
#ifndef GS5_MB_SPLIT
#define GS5_MB_SPLIT 6      // 16-window row blocks per tile of the split arithmetic's k_gstep5 (96 windows: 253 registers, none spilled; 128: spills; 64: 7.67 vs 7.54 ms on the 32-limb model)
#endif
constexpr int gs5_lds_bytes(bool split, int mb) { return 2 * (split ? 2 : 1) * mb * P16::BLK + GS5_MAX_TERMS * GS5_TERM_BYTES; }
// SPLIT (the split-bf16 arithmetic of section 4b: rows [hi Hd | lo Hd], three products per term, hi x hi, lo-weights x hi, hi-weights x lo) runs it on 96-window tiles
// (MB = GS5_MB_SPLIT = 6): twice the planes in the weight ring, the rows in flight and the window fragments.
// RAW (bf16 arithmetic): every term is one RAW INPUT row of the caller's tensors (the encoder launch) -- its own row stride, its own number of K chunks, whole 16-byte
// chunks inside the row pitch only, columns past the feature width zeroed and the symmetry signs applied as the row is staged.
template <bool SPLIT, int MB, bool MASKED, int NW = 8, bool RAW = false> __global__ __launch_bounds__(64 * NW, 2) void k_gstep5(GArgs a) {
    static_assert(!RAW || (!SPLIT && !MASKED), "raw-input launches: bf16 arithmetic, no relu bits");      // MASKED: some term of the launch carries relu bits (the backward sweeps: dH = dX . relu bits)
    using P = P16;
    constexpr int NS = 2, PL = SPLIT ? 2 : 1;            // NW waves (8: one workgroup per CU and two waves per SIMD in step; 4: 256 columns per workgroup, TWO workgroups per CU, one wave per SIMD each, out of step)
    //            // 8 waves, each NS 32-column slices (64 columns) of the 512: two waves per SIMD, up to 256 registers; PL planes
    constexpr int SET = PL * MB * P16::BLK;                      // one tile set: blocks [0, MB) the (hi) rows, [MB, 2 MB) the lo halves
    constexpr int NPASS = MB * 16 / (4 * NW);                    // staging passes: thread = (row rr + 4 NW i, chunk c)
    static_assert(NPASS >= 1 && NPASS * 4 * NW == MB * 16, "row blocks must cover whole staging passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two tile sets of MB blocks, then the term table
    const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nctg = a.NCT * 2 / NW;      // column groups of NW * 64 columns
    int ctg, jt;
    if constexpr (NW == 4) {
        // the nctg workgroups of a (job, tile) read the same source rows: 8 blocks apart, i.e. on one XCD (round-robin dispatch) and dispatched back to back, so that
        // the later readers find the rows in that XCD's L2 (grid padded to whole groups of 8 nctg)
        ctg = (blockIdx.x >> 3) % nctg; jt = (blockIdx.x / (8 * nctg)) * 8 + (blockIdx.x & 7);
        if (jt >= a.njt) return;
    } else { ctg = blockIdx.x % nctg; jt = blockIdx.x / nctg; }
    const int tile = jt % a.tiles;
    const int* job = a.jobs + (size_t)(a.job0 + jt / a.tiles) * JOB_INTS;
    const int w0 = tile * MB * P::ROWS, B = a.B, Hd = a.Hd;
    const int flags = job[J_FLAGS], nterms = job[J_NTERMS], NCT = a.NCT;
    const int ct0 = ctg * (NW / 2) + (wq >> 1), wv0 = (wq & 1) * 2;      // this wave's slices 2 wq and 2 wq + 1 of the 16: 128-column pack tile ct0, wave slots wv0 and wv0 + 1 of it
    GEN_TL(0);
#ifdef GGW_STAMPS
    long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64(); int nchunk = 0;      // (phase clocks, tools/stamps_gstep4.py)
#define GS5_T(k) { const long long t1 = clock64(); tk[k] += t1 - t0; t0 = t1; }
#else
#define GS5_T(k)
#endif
    // term table: {row base, relu-byte base, pack, has relu bits} of each term, resolved ONCE (item -> source -> buffer offset is a chain of dependent loads) and read
    // back per term with one LDS broadcast -- behind the loop's barrier hipcc issues plain loads of the plan tables as VECTOR loads, whose vmcnt(0) would wait
    // for every row and weight request in flight
    char* ttab = smem + 2 * SET;
    if (tid < nterms) {
        const int* term = a.terms + (size_t)(job[J_TERM0] + tid) * TERM_INTS;
        const int* src = a.srcs + (size_t)term[T_SRC0] * SRC_INTS;
        const int node = src[S_NODE], mbuf = src[S_MASK];
        if constexpr (RAW) {      // {row 0 of the node, row stride in bytes, feature width | pack, K chunks, pitch in elements, first sign byte}
            const int t = src[S_BUF];
            const uint64_t base = reinterpret_cast<uint64_t>(a.x[t]) + (uint64_t)node * (uint64_t)a.pitch[t] * 2;
            *reinterpret_cast<u32x4*>(ttab + tid * GS5_TERM_BYTES) = u32x4{(unsigned)base, (unsigned)(base >> 32), (unsigned)(a.nodes[t] * a.pitch[t] * 2), (unsigned)term[T_WIDTH]};
            *reinterpret_cast<u32x4*>(ttab + tid * GS5_TERM_BYTES + 16) = u32x4{(unsigned)term[T_PACK], (unsigned)term[T_NKC], (unsigned)a.pitch[t], (unsigned)term[T_SIGN]};
        } else {
        const uint64_t base = reinterpret_cast<uint64_t>(a.ws + a.buf_off[src[S_BUF]]) + g_row<SPLIT>(0, node, B, Hd) * 2;
        const uint64_t mb = reinterpret_cast<uint64_t>(a.ws + a.buf_off[mbuf >= 0 ? mbuf : 0]) + g_relu_byte(node, B, Hd, 0, 0);      // (no relu bits: any mapped address; the bytes are requested and ignored)
        *reinterpret_cast<u32x4*>(ttab + tid * GS5_TERM_BYTES) = u32x4{(unsigned)base, (unsigned)(base >> 32), (unsigned)mb, (unsigned)(mb >> 32)};
        *reinterpret_cast<u32x2*>(ttab + tid * GS5_TERM_BYTES + 16) = u32x2{(unsigned)term[T_PACK], mbuf >= 0 ? 1u : 0u};
        }
    }

    P::Acc acc[NS][MB];
    {
        const float* bias = (flags & JF_BIAS) ? a.bias + ((size_t)job[J_BIAS] + ct0) * TW : nullptr;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc_init_bias<T16>(acc[sl][m], bias, wv0 + sl, lane);
    }
    const int ao0 = lds_chunk<T16>(0, lane & 15, (lane >> 4) * P::NAV);
    const int nb16 = (B + 15) >> 4;
    const char* wbase = reinterpret_cast<const char*>(a.wpack);
    const unsigned woff = ((unsigned)wv0 * P::NBV * 64 + lane) * 16;      // + pack * (H * H * 2) + slice * 8192 + vector * 1024
    // Per-thread offsets (32 bits, added to uniform bases) are RECOMPUTED per chunk from the lane index, a dozen VALU operations against 128 MFMAs: kept live across
    // the loop they -- not the rows in flight -- were what hipcc spilled, and every scratch reload waits for ALL outstanding requests (scratch shares vmcnt).
    struct Cur { int ti, kc, pack; bool msk; gchar* base; gchar* mb; int nkc, stride, F, pitchc, sign; };      // the chunk stream: (term, K chunk) in order (nkc .. sign: raw-input terms)
    auto open_term = [&](Cur& q) {
        const u32x4 e = *reinterpret_cast<const u32x4*>(ttab + q.ti * GS5_TERM_BYTES);
        const uint64_t b = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)e[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)e[0]);
        q.base = reinterpret_cast<gchar*>(b);
        if constexpr (RAW) {
            const u32x4 f = *reinterpret_cast<const u32x4*>(ttab + q.ti * GS5_TERM_BYTES + 16);
            q.stride = __builtin_amdgcn_readfirstlane((int)e[2]); q.F = __builtin_amdgcn_readfirstlane((int)e[3]);
            q.pack = __builtin_amdgcn_readfirstlane((int)f[0]); q.nkc = __builtin_amdgcn_readfirstlane((int)f[1]);
            q.pitchc = __builtin_amdgcn_readfirstlane((int)f[2]); q.sign = __builtin_amdgcn_readfirstlane((int)f[3]);
            q.msk = false; q.mb = nullptr;
        } else {
            const u32x2 f = *reinterpret_cast<const u32x2*>(ttab + q.ti * GS5_TERM_BYTES + 16);
            const uint64_t m = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)e[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)e[2]);
            q.mb = reinterpret_cast<gchar*>(m);
            q.pack = __builtin_amdgcn_readfirstlane((int)f[0]); q.msk = __builtin_amdgcn_readfirstlane((int)f[1]) != 0;
            q.nkc = NCT;      // (every term of an all-plain launch has NCT chunks)
        }
    };
    auto advance = [&](Cur& q) { if (++q.kc == q.nkc) { q.kc = 0; ++q.ti; open_term(q); } };
    u32x4 rv[PL][NPASS]; unsigned rm[MASKED ? NPASS : 1]; u32x2 rs;      // rs: the chunk's eight sign bytes (RAW)
    auto fetch = [&](const Cur& q) {      // request the rows of a chunk: thread = (window rr + 4 NW i of the tile, 16-byte chunk c); rows past the batch re-read the last window
        if constexpr (RAW) {
            const int tt = wq * 64 + lane_now(), c = tt & 15, rr = tt >> 4;
            const int col = q.kc * TW + c * 8;
            const unsigned coff = (col + 8 <= q.pitchc) ? (unsigned)(col * 2) : 0u;      // a chunk that is not wholly inside the row pitch is not requested: it re-reads the row's first chunk and is zeroed as it is staged (no branch)
            rs = *reinterpret_cast<const u32x2 __attribute__((address_space(1)))*>(uniform_ptr(reinterpret_cast<const char*>(a.signs) + q.sign + q.kc * TW) + (unsigned)(c * 8));
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const unsigned w = (unsigned)min(w0 + i * (4 * NW) + rr, B - 1);
                rv[0][i] = gload16(q.base, w * (unsigned)q.stride + coff);
            }
            return;
        }
        gchar* rb = q.base + q.kc * (TW * 2);
        gchar* mb = q.mb + (((size_t)(q.kc * 4) * nb16) << 6);
        const int tt = wq * 64 + lane_now(), c = tt & 15, rr = tt >> 4;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned w = (unsigned)min(w0 + i * (4 * NW) + rr, B - 1);
            rv[0][i] = gload16(rb, w * (unsigned)(PL * Hd * 2) + (unsigned)(c * 16));
            if constexpr (SPLIT) rv[1][i] = gload16(rb, w * (unsigned)(PL * Hd * 2) + (unsigned)(Hd * 2 + c * 16));
            if constexpr (MASKED) rm[i] = gload1(mb, ((((unsigned)(c >> 2)) * (unsigned)nb16 + (w >> 4)) << 6) + (unsigned)((c & 3) << 4) + (w & 15u));      // (requested whether or not THIS term has relu bits -- no branch; see stage)
        }
    };
    auto stage = [&](const Cur& q, char* set) {
        const int tt = wq * 64 + lane_now(), c = tt & 15, rr = tt >> 4;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int grow = i * (4 * NW) + rr;
            if constexpr (RAW) {      // columns past the feature width (and chunks outside the pitch: nv <= 0) zeroed, symmetry signs applied
                const int nv = q.F - (q.kc * TW + c * 8);
                const unsigned w0s = rs[0], w1s = rs[1];
                const u32x4 sx = u32x4{((w0s & 1u) << 15) | (((w0s >> 8) & 1u) << 31), (((w0s >> 16) & 1u) << 15) | (((w0s >> 24) & 1u) << 31),
                                       ((w1s & 1u) << 15) | (((w1s >> 8) & 1u) << 31), (((w1s >> 16) & 1u) << 15) | (((w1s >> 24) & 1u) << 31)};
                *reinterpret_cast<u32x4*>(set + lds_chunk<T16>(grow >> 4, grow & 15, c)) = chunk_keep_first<T16>(rv[0][i], nv) ^ sx;
                continue;
            }
            const unsigned bits = MASKED ? (q.msk ? rm[i] : 0xffu) : 0xffu;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
                *reinterpret_cast<u32x4*>(set + lds_chunk<T16>(pl * MB + (grow >> 4), grow & 15, c)) = MASKED ? chunk_mask_bits<T16>(rv[pl][i], bits) : rv[pl][i];
        }
    };
    // weight fragments: a ring of two K steps (slot t & 1 holds vectors t and 4 + t of both slices); after step t its slot takes step t + 2 -- of this chunk, or
    // steps 0 / 1 of the next one -- so a request has one whole K step (32 MFMAs per wave, two waves per SIMD: ~1 k clocks) to come back from L2
    bf16x8 wr[2][PL][NS][2];
    const size_t lo_img = (size_t)a.n_img * (H * H * 2);      // the lo halves of the packed weights: image a.n_img + pack
    auto wload = [&](int slot, const char* wp, int t) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const unsigned wo = (unsigned)opaque((int)woff);
                wr[slot][pl][sl][0] = __builtin_bit_cast(bf16x8, gload16(uniform_ptr(wp + pl * lo_img + sl * 8192 + t * 1024), wo));
                wr[slot][pl][sl][1] = __builtin_bit_cast(bf16x8, gload16(uniform_ptr(wp + pl * lo_img + sl * 8192 + (4 + t) * 1024), wo));
            }
    };
    auto mfma_step = [&](const char* cur, int t, auto) {      // K step t: every row block against the step's vectors of both slices; the next block's window fragment(s) are read under a block's MFMAs
        const char* xp = cur + (ao0 ^ (16 * t));
        bf16x8 x0[PL], x1[PL];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) x0[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xp + pl * MB * P::BLK));
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m + 1 < MB) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) x1[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xp + (pl * MB + m + 1) * P::BLK));
            }
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                acc[sl][m].c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t & 1][0][sl][0], x0[0], acc[sl][m].c[0], 0, 0, 0);
                acc[sl][m].c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t & 1][0][sl][1], x0[0], acc[sl][m].c[1], 0, 0, 0);
                if constexpr (SPLIT) {
                    acc[sl][m].c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t & 1][1][sl][0], x0[0], acc[sl][m].c[0], 0, 0, 0);
                    acc[sl][m].c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t & 1][1][sl][1], x0[0], acc[sl][m].c[1], 0, 0, 0);
                    acc[sl][m].c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t & 1][0][sl][0], x0[PL - 1], acc[sl][m].c[0], 0, 0, 0);
                    acc[sl][m].c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t & 1][0][sl][1], x0[PL - 1], acc[sl][m].c[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // (keeps the read of block m + 1 in front of block m's MFMAs: hipcc otherwise sinks it to its use, one read in flight)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) x0[pl] = x1[pl];
        }
    };
    int nchunks = nterms * NCT;
    if constexpr (RAW) { nchunks = 0; for (int ti = 0; ti < nterms; ++ti) nchunks += ro_int(a.terms + (size_t)(ro_int(job, J_TERM0) + ti) * TERM_INTS, T_NKC); }
    __syncthreads();      // the term table
    if (nchunks > 0) {
        Cur nx{0, 0, 0, false, nullptr, nullptr, 0, 0, 0, 0, 0};
        open_term(nx);
        fetch(nx);
        const char* wp_cur = wbase + (size_t)(nx.pack + nx.kc * NCT + ct0) * (H * H * 2);
        wload(0, wp_cur, 0); wload(1, wp_cur, 1);
        stage(nx, smem);
        if (nchunks > 1) advance(nx);
        __syncthreads();
        GS5_T(4)      // (the prologue: term table, bias, first rows and weight fragments)
        int buf = 0;
        // One chunk per trip (`nx` is the chunk after it), NO branch between the requests and their uses: with the requests inside `if (more)` blocks hipcc's wait
        // insertion lost count at the joins and put vmcnt(0) in front of the first MFMA.  The last trip has nothing left to prefetch; it requests its own chunk
        // again (valid addresses, L2 hits, results unused) rather than branch.  The rows of chunk k + 1 are requested in the MIDDLE of chunk k: at its top the
        // request queue of the CU is still draining the weight requests every wave issued at the end of the previous chunk (1.8 k clocks of stall there).
        // Measured and not kept (HISTORY.md): the two waves of a SIMD half a chunk apart (waves 4-7 request at the top and stage in mid-chunk, so that one
        // of the two always has MFMAs to issue while the other stages) -- as two copies of this loop hipcc allocates 256 registers and spills inside both
        // (each copy alone: 232, none), and every scratch reload is a vmcnt(0): 321 / 455 us per layer launch against 240 / 234.
        for (int k = 0; k < nchunks; ++k) {
            const char* cur = smem + buf * SET;
            const char* wp_nxt = wbase + (size_t)(nx.pack + nx.kc * NCT + ct0) * (H * H * 2);
            GS5_T(0)
#ifdef GS5_ABL_W      // (ablation builds, timing only: no weight requests inside the loop / no row requests and staging)
#define GS5_WLOAD(s, p, t)
#else
#define GS5_WLOAD(s, p, t) wload(s, p, t)
#endif
            mfma_step(cur, 0, std::true_type{}); GS5_WLOAD(0, wp_cur, 2);
            mfma_step(cur, 1, std::true_type{}); GS5_WLOAD(1, wp_cur, 3);
#ifndef GS5_ABL_ROWS
            fetch(nx);
#endif
            __builtin_amdgcn_sched_barrier(0);      // (nothing of the staging -- the relu-bit select on the requested bytes -- is to be scheduled up here: it would wait for the rows)
            mfma_step(cur, 2, std::true_type{}); GS5_WLOAD(0, wp_nxt, 0);
            mfma_step(cur, 3, std::true_type{}); GS5_WLOAD(1, wp_nxt, 1);
            wp_cur = wp_nxt;
#ifdef GGW_STAMPS
            { float d; asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(acc[NS - 1][MB - 1].c[1][3])); asm volatile("" :: "v"(d)); }
            ++nchunk;
#endif
            GS5_T(1)
            __builtin_amdgcn_sched_barrier(0);
#ifndef GS5_ABL_ROWS
            stage(nx, smem + (buf ^ 1) * SET);
#endif
            if (k + 2 < nchunks) advance(nx);
            GS5_T(2)
            __syncthreads();
            GS5_T(3)
            buf ^= 1;
        }
    }
    gstep_epilogue2<SPLIT, MB, NS, !MASKED>(a, job, acc, ct0, wv0, lane, w0);      // (the masked instantiations only run with MSHGNN_GEN_DHM=0: run-time flags there, for the size of the library)
#ifdef GGW_STAMPS
    GS5_T(5)
    if (a.stamps && tid == 0) { for (int q = 0; q < 6; ++q) a.stamps[(size_t)blockIdx.x * 8 + q] = tk[q]; a.stamps[(size_t)blockIdx.x * 8 + 6] = nchunk; a.stamps[(size_t)blockIdx.x * 8 + 7] = nterms; }
#endif
    GEN_TL(1);
}

// ------------------------------------------------------------------------------------------------------
// decoder forward / backward on the rows of the out type (hgnn_c2.py:176-189) + fused wrapper MSE / cross entropy
// thread = (row, 8-column chunk c of 16), looping over the Hd / 128 column groups
// ------------------------------------------------------------------------------------------------------
struct GDecArgs {
    const void* xl; void* dxl; void* dhl; const uint8_t* maskb; int dh_only; const float* params; const float* out_mask; float* out; const float* gout; float* slabs;
    int64_t off_w, off_b; int B, Hd, node0, n_out, dout;
    const float* y; const int32_t* labels; float inv_n;
};
template <bool SPLIT> __device__ __forceinline__ void g_load8(const T16* p, int Hd, float (&v)[8]) {
    load8<T16>(p, v);
    if constexpr (SPLIT) { float l[8]; load8<T16>(p + Hd, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += l[e]; }
}
template <bool SPLIT> __global__ __launch_bounds__(256) void k_gdec_fwd(GDecArgs a) {
    const int c = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4), rows = (int64_t)a.B * a.n_out;
    const bool ok = row < rows;
    const int w = ok ? (int)(row / a.n_out) : 0, f = ok ? (int)(row % a.n_out) : 0;
    const T16* xr = reinterpret_cast<const T16*>(a.xl) + g_row<SPLIT>(w, a.node0 + f, a.B, a.Hd);
    const float* W = a.params + a.off_w;
    float s[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) s[d] = 0.f;
    for (int cg = 0; cg < a.Hd; cg += TW) {
        float x[8];
        g_load8<SPLIT>(xr + cg + c * 8, a.Hd, x);
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (d < a.dout) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[d] += x[e] * W[(size_t)d * a.Hd + cg + c * 8 + e];
            }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if (d < a.dout) {
            float v = s[d];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (c == 0 && ok) a.out[row * a.dout + d] = (v + a.params[a.off_b + d]) * a.out_mask[f * a.dout + d];
        }
    }
}
template <bool SPLIT> __global__ __launch_bounds__(256) void k_gdec_bwd(GDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gred[];      // [16 row groups][8 x 128]: one column group at a time
    __shared__ float bred[16][8];
    const int SF = 8 * a.Hd + 16;
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t rows = (int64_t)a.B * a.n_out;
    const int64_t per = ((rows + gridDim.x - 1) / gridDim.x + 15) / 16 * 16;
    const int64_t r_begin = (int64_t)blockIdx.x * per, r_end = min(rows, r_begin + per);
    const float* W = a.params + a.off_w;
    float* slab = a.slabs + (size_t)blockIdx.x * SF;
    float accb[8], lsum = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) accb[d] = 0.f;
    // column group by column group: the decoder weight gradient accumulates in registers over this block's rows (fixed order)
    for (int cg = 0; cg < a.Hd; cg += TW) {
        float accw[8][8];
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 8; ++e) accw[d][e] = 0.f;
        for (int64_t r = r_begin + rg; r < r_end; r += 16) {
            const int w = (int)(r / a.n_out), f = (int)(r % a.n_out);
            const size_t idx = g_row<SPLIT>(w, a.node0 + f, a.B, a.Hd) + cg + c * 8;
            float x[8], dx[8];
            g_load8<SPLIT>(reinterpret_cast<const T16*>(a.xl) + idx, a.Hd, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) dx[e] = 0.f;
            float ce_g[2] = {0.f, 0.f};
            if (a.labels) {   // wrapper cross entropy (gnnLightning.py:640-648, mean over batch * feet): dL/dlogit = (p - onehot) / rows
                const float l0 = a.out[r * 2], l1 = a.out[r * 2 + 1];
                const float m = fmaxf(l0, l1), e0 = expf(l0 - m), e1 = expf(l1 - m), se = e0 + e1;
                const int lab = a.labels[r] != 0;
                ce_g[0] = (e0 / se - (lab ? 0.f : 1.f)) * a.inv_n; ce_g[1] = (e1 / se - (lab ? 1.f : 0.f)) * a.inv_n;
                if (c == 0 && cg == 0) lsum += (m + logf(se)) - (lab ? l1 : l0);
            }
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                if (d < a.dout) {
                    float go;
                    if (a.labels) go = ce_g[d & 1];
                    else if (a.y) {   // wrapper MSE (gnnLightning.py:633-639): dL/dout = 2 (out - y) / n
                        const float dlt = a.out[r * a.dout + d] - a.y[r * a.dout + d];
                        go = 2.0f * dlt * a.inv_n;
                        if (c == 0 && cg == 0) lsum += dlt * dlt;
                    } else go = a.gout[r * a.dout + d];
                    const float g = go * a.out_mask[f * a.dout + d];
                    if (cg == 0) accb[d] += g;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { accw[d][e] += g * x[e]; dx[e] += g * W[(size_t)d * a.Hd + cg + c * 8 + e]; }
                }
            }
            T16* q = reinterpret_cast<T16*>(a.dxl) + idx;
            const unsigned hbits = a.dhl ? a.maskb[g_relu_byte(a.node0 + f, a.B, a.Hd, w, cg + c * 8)] : 0u;      // (plans with dhm: dH of the last layer = dX_L . its relu bits, written here)
            if constexpr (SPLIT) {
                u32x4 hi, lo;
                split_oct(f32x4{dx[0], dx[1], dx[2], dx[3]}, f32x4{dx[4], dx[5], dx[6], dx[7]}, hi, lo);
                if (!a.dh_only) { *reinterpret_cast<u32x4*>(q) = hi; *reinterpret_cast<u32x4*>(q + a.Hd) = lo; }
                if (a.dhl) {
                    T16* q2 = reinterpret_cast<T16*>(a.dhl) + idx;
                    *reinterpret_cast<u32x4*>(q2) = chunk_mask_bits<T16>(hi, hbits); *reinterpret_cast<u32x4*>(q2 + a.Hd) = chunk_mask_bits<T16>(lo, hbits);
                }
            } else {
                const u32x4 yv = pack_oct(f32x4{dx[0], dx[1], dx[2], dx[3]}, f32x4{dx[4], dx[5], dx[6], dx[7]});
                if (!a.dh_only) *reinterpret_cast<u32x4*>(q) = yv;
                if (a.dhl) *reinterpret_cast<u32x4*>(reinterpret_cast<T16*>(a.dhl) + idx) = chunk_mask_bits<T16>(yv, hbits);
            }
        }
        __syncthreads();      // the previous column group's sums have been read
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 8; ++e) gred[rg * (8 * TW) + d * TW + c * 8 + e] = accw[d][e];
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * TW; i += 256) {
            float s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s2 += gred[r * (8 * TW) + i];
            slab[(i / TW) * a.Hd + cg + (i % TW)] = s2;
        }
    }
    if (c == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) bred[rg][d] = accb[d];
    }
    {   // per-block loss partial rides in the slab (summed in fixed order by k_gfinalize: no atomics, deterministic)
        __shared__ float lred[4];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m, 64);
        if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) slab[8 * a.Hd + 8] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
        if (threadIdx.x < 8) {
            float s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s2 += bred[r][threadIdx.x];
            slab[8 * a.Hd + threadIdx.x] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_ggradw: one SUPER-UNIT per workgroup = OS x 2 adjacent 128x128 tiles of one target's weight gradient, dW = P^T Q summed over a chunk of
// the target's items and a part of the batch.  4 OS x 4 waves, each a 64x64 piece; the P rows [32 windows x 128 OS] and the Q rows [32 x 256]
// of a step are staged ONCE for all of them.  Steps = (item, 32-window chunk) pairs; the global loads of the next step(s) run in registers,
// untouched, while a step is multiplied (P rows always; Q rows when the item has one source; aggregated Q rows are gathered at staging time).
// ------------------------------------------------------------------------------------------------------
// The wave grid is a parameter (NWV / 4 wave rows x 4 wave columns) because other shapes were measured on the bf16 plan -- 8 waves of 128x64 with double-buffered
// tiles: 12 % faster on the lean super-units, but a different block size than the general ones need, i.e. a second launch; 4 waves of 128x128 with the
// accumulators in the AGPRs: hipcc spills 0.7-1.5 KB per lane whichever way the accumulators are pinned -- the launches use NWV = 8 OS.
// Windows per step (= per barrier) of k_ggradw's lean streams; the general streams keep 32.  A 32-window step is 8 MFMAs per wave between barriers (the phase clocks,
// tools/stamps_ggradw.py, put 49 % of it at the barrier); 64 need the staging registers of two steps, which the bf16 kernel's 128 only have without the relu bytes
// (NOMASK).  Measured on the 32-limb model, B = 1024: with masks, 64 windows per step 1.80-1.99 ms against 1.16; 8 waves of 128 x 64 at 64 / 32: 1.40 / 1.47 ms.
#ifndef GGW_KW_LEAN_NOMASK
#define GGW_KW_LEAN_NOMASK 64
#endif
#ifndef GGW_NWV_BF16
#define GGW_NWV_BF16 16
#endif
#ifndef GGW_NWV_SPLIT2
#define GGW_NWV_SPLIT2 8      // waves of the split kernel on 256 x 256 super-units (GGW_SPLIT_OS2)
#endif
constexpr int ggw_kw_lean(bool split, int os, bool nomask) { return (!split && os == 2 && nomask) ? GGW_KW_LEAN_NOMASK : 32; }
constexpr int ggw_lds_bytes(bool split, int os, bool nomask = false) { return 2 * (split ? 2 : 1) * (os + 2) * ggw_kw_lean(split, os, nomask) * GWB_PITCH * 2; }      // k_ggradw's dynamic LDS: two tile sets per plane
// The kernel's body, instantiated once per stream kind (PATH 1: lean super-units, 0: general ones) so that each kind gets its own step size KW and its own registers.
// NOMASK (plans with dhm: no P operand carries relu bits): the relu-byte requests and the bit expansion are compiled out, which is what lets the lean streams of the
// bf16 kernel stage 64 windows per step inside 128 registers.
template <bool SPLIT, int OS, int NWV, int KW, int PATH, bool NOMASK> __device__ __forceinline__ void ggradw_body(const GArgs& a) {
    constexpr int NT = 64 * NWV, PC = 16 * OS;      // KW: windows per step; PC: 16-byte chunks per staged P row
    constexpr int NQ = KW * 32 / NT;                         // Q chunks per thread (1 with 1024 threads, 2 with 512)
    constexpr int NP = KW * PC / NT, PR = NT / PC;           // P chunks per thread; P rows per staging pass
    constexpr int WCOLS = 4;                                 // wave grid: NWV / WCOLS rows x WCOLS columns
    constexpr int RI = (128 * OS) / (NWV / WCOLS) / 32;      // 32-row blocks of the o range per wave
    constexpr int CJ = 256 / WCOLS / 32;                     // 32-column blocks of the k range per wave
#ifndef GGW_NST_SPLIT
#define GGW_NST_SPLIT 2
#endif
#ifndef GGW_NST_BF16
#define GGW_NST_BF16 1      // (two stages at 16 waves: 128 VGPRs with 60 B of scratch, 3.29 ms against 2.27 ms at h=512)
#endif
    constexpr int NST = OS == 2 ? 1 : (SPLIT ? GGW_NST_SPLIT : GGW_NST_BF16);      // register stages (16 waves: 128 VGPRs, one stage)
    constexpr int TSZ = KW * GWB_PITCH;                      // one staged 32 x 128 tile
    constexpr int BUFSZ = (OS + 2) * TSZ;
    // DB: two tile sets -- step s + 1 is staged into one while step s is multiplied from the other, ONE barrier per step instead
    // of two (the phase clocks showed 640 of a step's 3 360 clocks at the second barrier: the waves wait for the slowest stager, then again for the slowest
    // multiplier; with one barrier a wave that has staged goes straight on to its MFMAs)
    constexpr bool DB = true;
    extern __shared__ __attribute__((aligned(16))) char ggw_smem[];      // ggw_lds_bytes<SPLIT, OS>(): two tile sets of the hi plane, then (split arithmetic) two of the lo plane
    __bf16* tiles_h = reinterpret_cast<__bf16*>(ggw_smem);               // P sub-tiles [0, OS), Q sub-tiles [OS, OS + 2) of each set (the bias reduction reuses the first one)
    __bf16* tiles_l = tiles_h + (SPLIT ? 2 * BUFSZ : 0);
    int sbuf = 0, mbuf = 0;      // tile set the staging writes / the MFMAs read
    auto Ph = [&](int t) { return tiles_h + sbuf * BUFSZ + t * TSZ; };
    auto Qh = [&](int t) { return tiles_h + sbuf * BUFSZ + (OS + t) * TSZ; };
    auto Pl = [&](int t) { return tiles_l + sbuf * BUFSZ + t * TSZ; };
    auto Ql = [&](int t) { return tiles_l + sbuf * BUFSZ + (OS + t) * TSZ; };
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv / WCOLS, wc = wv % WCOLS;              // wave (wr, wc): rows [32 RI wr, +32 RI) of the o range, columns [32 CJ wc, +32 CJ) of the k range
    const int row0 = 32 * RI * wr, col0 = 32 * CJ * wc;
    GEN_TL(0);
    const int su_i = a.su_order[blockIdx.x % a.n_sunits], part = blockIdx.x / a.n_sunits;
    const int* su = a.sunits + (size_t)su_i * SUNIT_INTS;
    const int it0 = su[SU_ITEM0], pcol = su[SU_PCOL], qcol = su[SU_QCOL], qn = su[SU_QN];
    const int nchunks = (a.B + KW - 1) / KW;
    const int ch0 = (int)((int64_t)part * nchunks / a.n_parts), ch1 = (int)((int64_t)(part + 1) * nchunks / a.n_parts);
    const int nch = ch1 - ch0, nsteps = (su[SU_ITEM1] - it0) * nch;
    const int B = a.B, Hd = a.Hd;
    const int cp = tid % PC, rp = tid / PC;                  // P staging: (row rp < 32, chunk cp)
    const int cq = tid & 31, rq = tid >> 5;                  // Q staging: (row rq + (NT / 32) i, chunk cq), i < NQ
    const int one_bits = __float_as_int(1.0f);
    f32x16 acc[RI][CJ];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    struct Stage { u32x4 pa[NP], pb[NP], qa[NQ], qb[NQ], qc[SPLIT ? 1 : NQ]; unsigned mw[NP]; };      // qc: second source of a two-source sum (lean streams, bf16)      // a: bf16 chunk / hi half / first four fp32; b: lo half / next four fp32
    auto q_kind = [&](const int* im) {      // 1 raw input, 0 single activation source, 2 aggregate
        if (im[I_KIND] == 1) return 1;
        const int* src = a.srcs + (size_t)im[I_SRC0] * SRC_INTS;
        return (im[I_NSRC] == 1 && src[S_SCALE] == one_bits) ? 0 : 2; };
    // step cursors (item, chunk) of the fetch and of the staging stream: advanced by one step per call (no division per step)
    int f_it = it0, f_ch = ch0, s_it = it0, s_ch = ch0;
    // An item's descriptor, resolved ONCE per item and stream (not per step): item -> source -> buffer offset is a chain of five dependent scalar loads, ~1.5 k
    // cycles that every wave of the workgroup paid at the same moment in front of each step's row requests (a step's MFMAs are 256 cycles per wave).
    struct ItemC { const int* src; int n_src, qk, pmask, qt, qnode, sign; const T16* prow; const uint8_t* pmb; const T16* qrow; };
    auto resolve = [&](int it) {
        ItemC q;
        const int* im = a.items + (size_t)it * GITEM_INTS;
        q.src = a.srcs + (size_t)im[I_SRC0] * SRC_INTS; q.n_src = im[I_NSRC]; q.qk = q_kind(im); q.pmask = im[I_PMASK];
        const int pnode = im[I_PNODE];
        q.prow = reinterpret_cast<const T16*>(a.ws + a.buf_off[im[I_PBUF]]) + g_row<SPLIT>(0, pnode, B, Hd) + pcol;      // row 0 of the P operand, first column of this super-unit
        q.pmb = q.pmask >= 0 ? reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[q.pmask]) + g_relu_byte(pnode, B, Hd, 0, 0) : nullptr;
        q.qt = q.src[S_BUF]; q.qnode = q.src[S_NODE]; q.sign = q.src[S_MASK];
        q.qrow = q.qk == 0 ? reinterpret_cast<const T16*>(a.ws + a.buf_off[q.qt]) + g_row<SPLIT>(0, q.qnode, B, Hd) + qcol : nullptr;
        return q;
    };
    ItemC fi = resolve(min(it0, su[SU_ITEM1] - 1)), si = fi;
    // relu byte of (window w, column col) relative to the node's first byte (g_relu_byte without the node term)
    auto mask_off = [&](int w, int col) { return ((((size_t)(col >> 5)) * ((B + 15) >> 4) + (w >> 4)) << 6) + (((col >> 3) & 3) << 4) + (w & 15); };
    auto fetch = [&](Stage& st) {
        const int w0 = f_ch * KW, qk = fi.qk;
        const ItemC q = fi;
        if (++f_ch == ch1) { f_ch = ch0; ++f_it; if (f_it < su[SU_ITEM1]) fi = resolve(f_it); }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            st.pa[i] = u32x4{0, 0, 0, 0}; st.pb[i] = u32x4{0, 0, 0, 0}; st.mw[i] = 0xffu;
            const int w = w0 + rp + PR * i;
            if (w < B) {
                const T16* pr = q.prow + (size_t)w * (SPLIT ? 2 * Hd : Hd) + cp * 8;
                st.pa[i] = *reinterpret_cast<const u32x4*>(pr);
                if constexpr (SPLIT) st.pb[i] = *reinterpret_cast<const u32x4*>(pr + Hd);
                if constexpr (!NOMASK) { if (q.pmask >= 0) st.mw[i] = q.pmb[mask_off(w, pcol + cp * 8)]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int w = w0 + rq + (NT / 32) * i;
            st.qa[i] = u32x4{0, 0, 0, 0}; st.qb[i] = u32x4{0, 0, 0, 0};
            if (w < B && cq * 8 < qn) {
                if (qk == 0) {
                    const T16* qr = q.qrow + (size_t)w * (SPLIT ? 2 * Hd : Hd) + cq * 8;
                    st.qa[i] = *reinterpret_cast<const u32x4*>(qr);
                    if constexpr (SPLIT) st.qb[i] = *reinterpret_cast<const u32x4*>(qr + Hd);
                } else if (qk == 1) {
                    const int t = q.qt, nv = qn - cq * 8;
                    if constexpr (SPLIT) {
                        const float* qr = reinterpret_cast<const float*>(a.x[t]) + ((size_t)w * a.nodes[t] + q.qnode) * a.pitch[t] + qcol + cq * 8;
                        st.qa[i] = load_chunk<float>(qr, nv, a.vb[t]);
                        st.qb[i] = load_chunk<float>(qr + 4, nv - 4, a.vb[t]);
                    } else {
                        const T16* qr = reinterpret_cast<const T16*>(a.x[t]) + ((size_t)w * a.nodes[t] + q.qnode) * a.pitch[t] + qcol + cq * 8;
                        st.qa[i] = load_chunk<T16>(qr, nv, a.vb[t]);
                    }
                }
            }
        }
    };
    auto stage_to_lds = [&](const Stage& st) {
        const ItemC q = si;
        const int* src = q.src;
        const int w0 = s_ch * KW, qk = q.qk;
        int im[GITEM_INTS];      // (the fields the staging reads)
        im[I_PMASK] = q.pmask; im[I_NSRC] = q.n_src;
        if (++s_ch == ch1) { s_ch = ch0; ++s_it; if (s_it < su[SU_ITEM1]) si = resolve(s_it); }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            u32x4 ph = st.pa[i], pl = st.pb[i];
            if constexpr (!NOMASK) { if (im[I_PMASK] >= 0) { ph = chunk_mask_bits<T16>(ph, st.mw[i]); if constexpr (SPLIT) pl = chunk_mask_bits<T16>(pl, st.mw[i]); } }     // dH = dX . relu bits
            *reinterpret_cast<u32x4*>(Ph(cp >> 4) + gwb_elem(rp + PR * i, (cp & 15) * 8)) = ph;
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(Pl(cp >> 4) + gwb_elem(rp + PR * i, (cp & 15) * 8)) = pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {      // column sums of P (bias gradients of the kt == 0 units; cheap enough to keep unconditional)
                bsum[2 * e] += __builtin_bit_cast(float, ph[e] << 16) + (SPLIT ? __builtin_bit_cast(float, pl[e] << 16) : 0.f);
                bsum[2 * e + 1] += __builtin_bit_cast(float, ph[e] & 0xffff0000u) + (SPLIT ? __builtin_bit_cast(float, pl[e] & 0xffff0000u) : 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int row = rq + (NT / 32) * i, w = w0 + row;
            u32x4 qh = st.qa[i], ql = st.qb[i];
            if (qk == 1) {          // raw input: pad columns dropped, symmetry sign, (split plan) fp32 -> hi / lo
                const int nv = qn - cq * 8;
                const uint8_t* sg = a.signs + q.sign + qcol + cq * 8;
                if (nv > 0) {
                    if constexpr (SPLIT) {
                        const u32x4 fa = chunk_keep_first<float>(st.qa[i], nv) ^ sign_xor<float>(sg), fb = chunk_keep_first<float>(st.qb[i], nv - 4) ^ sign_xor<float>(sg + 4);
                        split_oct(__builtin_bit_cast(f32x4, fa), __builtin_bit_cast(f32x4, fb), qh, ql);
                    } else qh = chunk_keep_first<T16>(st.qa[i], nv) ^ sign_xor<T16>(sg);
                }
            } else if (qk == 2) {   // aggregate: fp32 sum (mean: scaled) of the source rows, gathered now
                float qs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) qs[e] = 0.f;
                if (w < B && cq * 8 < qn) gather8<SPLIT>(a, src, im[I_NSRC], w, qcol + cq * 8, qs);
                const f32x4 q0 = f32x4{qs[0], qs[1], qs[2], qs[3]}, q1 = f32x4{qs[4], qs[5], qs[6], qs[7]};
                if constexpr (SPLIT) split_oct(q0, q1, qh, ql); else qh = pack_oct(q0, q1);
            }
            *reinterpret_cast<u32x4*>(Qh(cq >> 4) + gwb_elem(row, (cq & 15) * 8)) = qh;
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(Ql(cq >> 4) + gwb_elem(row, (cq & 15) * 8)) = ql;
        }
    };
    // LEAN streams (super-units whose items all have ONE plain activation source at scale 1 -- or, on the bf16 plan, the sum of two --, SU_FLAGS bit 0 -- the plan sorts those items to the front of
    // their target): what changes per step is a uniform base address per operand (SGPRs, advanced by one 32-window chunk) -- the per-thread offsets are
    // constants of the launch, the item descriptor is read at item boundaries only, column sums of P are kept only where a bias gradient is read (bit 1).
    // The general streams above spend ~0.9 ms of a 2.0 ms launch at h = 512 on their own bookkeeping (timed with the loads and MFMAs compiled out).
    const int rowb = (SPLIT ? 2 * Hd : Hd) * 2;      // bytes per activation row
    const unsigned poff = (unsigned)rp * rowb + cp * 16, qoff = (unsigned)rq * rowb + cq * 16;
    const unsigned moff = (unsigned)mask_off(rp, pcol + cp * 8);
    const bool need_bias = (su[SU_FLAGS] & 2) != 0;
    struct LeanF { const char* pb; const uint8_t* mb; const char* qb; const char* qb2; int it, left, ch; bool msk, two; } LF{nullptr, nullptr, nullptr, nullptr, it0, 0, 0, false, false};
    struct LeanS { int it, left; bool msk, two; } LS{it0, 0, false, false};
    auto fetch_lean = [&](Stage& st) {
        if (LF.left == 0) {      // next item of the fetch stream
            const int* im = a.items + (size_t)LF.it * GITEM_INTS;
            const int* src = a.srcs + (size_t)im[I_SRC0] * SRC_INTS;
            const int pnode = im[I_PNODE], pm = im[I_PMASK];
            LF.ch = ch0;
            LF.pb = a.ws + a.buf_off[im[I_PBUF]] + (g_row<SPLIT>(LF.ch * KW, pnode, B, Hd) + pcol) * 2;
            LF.msk = pm >= 0;
            LF.mb = reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[pm >= 0 ? pm : 0]) + g_relu_byte(pnode, B, Hd, 0, 0) + ((size_t)((LF.ch * KW) >> 4) << 6);
            LF.qb = a.ws + a.buf_off[src[S_BUF]] + (g_row<SPLIT>(LF.ch * KW, src[S_NODE], B, Hd) + qcol) * 2;
            LF.two = !SPLIT && im[I_NSRC] == 2;      // Q = the sum of two plain rows (bf16 plan)
            if (LF.two) LF.qb2 = a.ws + a.buf_off[src[SRC_INTS + S_BUF]] + (g_row<SPLIT>(LF.ch * KW, src[SRC_INTS + S_NODE], B, Hd) + qcol) * 2;
            LF.left = nch; ++LF.it;
        }
        const int nrow = B - LF.ch * KW;      // rows of this chunk inside the batch
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            st.pa[i] = u32x4{0, 0, 0, 0}; st.pb[i] = u32x4{0, 0, 0, 0}; st.mw[i] = 0xffu;
            if (rp + PR * i < nrow) {
                const char* pr = LF.pb + (poff + (unsigned)(i * PR) * rowb);
                st.pa[i] = *reinterpret_cast<const u32x4*>(pr);
                if constexpr (SPLIT) st.pb[i] = *reinterpret_cast<const u32x4*>(pr + 2 * Hd);
                if constexpr (!NOMASK) { if (LF.msk) st.mw[i] = LF.mb[moff + (unsigned)((((PR * i) >> 4) << 6) + ((PR * i) & 15))]; }      // (rp < PR <= 16, or one pass)
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            st.qa[i] = u32x4{0, 0, 0, 0}; st.qb[i] = u32x4{0, 0, 0, 0};
            if constexpr (!SPLIT) st.qc[i] = u32x4{0, 0, 0, 0};
            if (rq + (NT / 32) * i < nrow && cq * 8 < qn) {
                const char* qr = LF.qb + (qoff + (unsigned)(i * (NT / 32)) * rowb);
                st.qa[i] = *reinterpret_cast<const u32x4*>(qr);
                if constexpr (SPLIT) st.qb[i] = *reinterpret_cast<const u32x4*>(qr + 2 * Hd);
                else if (LF.two) st.qc[i] = *reinterpret_cast<const u32x4*>(LF.qb2 + (qoff + (unsigned)(i * (NT / 32)) * rowb));
            }
        }
        LF.pb += (size_t)KW * rowb; LF.mb += (KW >> 4) << 6; LF.qb += (size_t)KW * rowb; LF.qb2 += (size_t)KW * rowb; ++LF.ch; --LF.left;
    };
    auto stage_lean = [&](const Stage& st) {
        if (LS.left == 0) {
            const int* im = a.items + (size_t)LS.it * GITEM_INTS;
            LS.msk = im[I_PMASK] >= 0; LS.two = !SPLIT && im[I_NSRC] == 2; LS.left = nch; ++LS.it;
        }
        --LS.left;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            u32x4 ph = st.pa[i], pl = st.pb[i];
            if constexpr (!NOMASK) { if (LS.msk) { ph = chunk_mask_bits<T16>(ph, st.mw[i]); if constexpr (SPLIT) pl = chunk_mask_bits<T16>(pl, st.mw[i]); } }     // dH = dX . relu bits
            *reinterpret_cast<u32x4*>(Ph(cp >> 4) + gwb_elem(rp + PR * i, (cp & 15) * 8)) = ph;
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(Pl(cp >> 4) + gwb_elem(rp + PR * i, (cp & 15) * 8)) = pl;
            if (need_bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __builtin_bit_cast(float, ph[e] << 16) + (SPLIT ? __builtin_bit_cast(float, pl[e] << 16) : 0.f);
                    bsum[2 * e + 1] += __builtin_bit_cast(float, ph[e] & 0xffff0000u) + (SPLIT ? __builtin_bit_cast(float, pl[e] & 0xffff0000u) : 0.f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int row = rq + (NT / 32) * i;
            u32x4 qh = st.qa[i];
            if constexpr (!SPLIT) {
                if (LS.two) {      // fp32 sum of the two rows, rounded once: what the general stream's gather computes
                    f32x4 a0, a1, b0, b1;
                    unpack_oct(st.qa[i], a0, a1); unpack_oct(st.qc[i], b0, b1);
                    qh = pack_oct(a0 + b0, a1 + b1);
                }
            }
            *reinterpret_cast<u32x4*>(Qh(cq >> 4) + gwb_elem(row, (cq & 15) * 8)) = qh;
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(Ql(cq >> 4) + gwb_elem(row, (cq & 15) * 8)) = st.qb[i];
        }
    };
    const __bf16* Pt0 = tiles_h + (row0 >> 7) * TSZ; const __bf16* Qt0 = tiles_h + (OS + (col0 >> 7)) * TSZ;
    const __bf16* Plt0 = tiles_l + (row0 >> 7) * TSZ; const __bf16* Qlt0 = tiles_l + (OS + (col0 >> 7)) * TSZ;
    const int my_unit = su[SU_UNIT + (row0 >> 7) * 2 + (col0 >> 7)];      // the 128x128 sub-tile this wave belongs to (-1: beyond the matrix edge)
    auto mfmas = [&]() {
        if (my_unit < 0) return;      // (wave-uniform)
        const __bf16* Pt = Pt0 + mbuf * BUFSZ; const __bf16* Qt = Qt0 + mbuf * BUFSZ;
        const __bf16* Plt = Plt0 + mbuf * BUFSZ; const __bf16* Qlt = Qlt0 + mbuf * BUFSZ;
#pragma unroll
        for (int ks = 0; ks < KW / 16; ++ks) {
            if constexpr (SPLIT) {      // Q fragments one column block at a time: 24 fragment registers live instead of 32
                bf16x8 afh[RI], afl[RI];
#pragma unroll
                for (int i = 0; i < RI; ++i) { afh[i] = tr_frag(Pt, ks * 16, (row0 & 127) + i * 32, lane); afl[i] = tr_frag(Plt, ks * 16, (row0 & 127) + i * 32, lane); }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 bqh = tr_frag(Qt, ks * 16, (col0 & 127) + j * 32, lane), bql = tr_frag(Qlt, ks * 16, (col0 & 127) + j * 32, lane);
#pragma unroll
                    for (int i = 0; i < RI; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bqh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bql, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[i], bqh, acc[i][j], 0, 0, 0);
                    }
                }
            } else {
                bf16x8 afh[RI], bqh[CJ];
#pragma unroll
                for (int i = 0; i < RI; ++i) afh[i] = tr_frag(Pt, ks * 16, (row0 & 127) + i * 32, lane);
#pragma unroll
                for (int i = 0; i < CJ; ++i) bqh[i] = tr_frag(Qt, ks * 16, (col0 & 127) + i * 32, lane);
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int j = 0; j < CJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bqh[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto run = [&](auto& fetchf, auto& stagef) {
        if constexpr (DB) {
            Stage sa;      // tile set s & 1 = step s; `sa` = step s + 1, requested under the MFMAs of step s - 1
            if (nsteps > 0) { fetchf(sa); sbuf = 0; stagef(sa); }
            if (nsteps > 1) fetchf(sa);
            __syncthreads();
#ifdef GGW_STAMPS
            long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64();
#define GGW_TD(k) { const long long t1 = clock64(); tk[k] += t1 - t0; t0 = t1; }
#else
#define GGW_TD(k)
#endif
            for (int s = 0; s < nsteps; ++s) {
                sbuf = (s + 1) & 1; mbuf = s & 1;
#ifdef GGW_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                GGW_TD(1)
#endif
                if (s + 1 < nsteps) stagef(sa);
                GGW_TD(2)
                if (s + 2 < nsteps) fetchf(sa);
                mfmas();
#ifdef GGW_STAMPS
                { float d; asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(acc[1][1][15])); asm volatile("" :: "v"(d)); }
#endif
                GGW_TD(4)
                __syncthreads();
                GGW_TD(0)
            }
            sbuf = 0;
#ifdef GGW_STAMPS
            if (a.stamps && tid == 0) { for (int k = 0; k < 5; ++k) a.stamps[(size_t)blockIdx.x * 8 + k] = tk[k]; a.stamps[(size_t)blockIdx.x * 8 + 5] = nsteps; a.stamps[(size_t)blockIdx.x * 8 + 6] = su[SU_FLAGS]; }
#endif
        } else if constexpr (NST == 2) {
            Stage sa, sb;
            if (nsteps > 0) fetchf(sa);
            if (nsteps > 1) fetchf(sb);
            for (int s = 0; s < nsteps; s += 2) {
                __syncthreads();      // the previous MFMA phase is done reading the tiles
                stagef(sa);
                __syncthreads();
                if (s + 2 < nsteps) fetchf(sa);
                mfmas();
                if (s + 1 < nsteps) {
                    __syncthreads();
                    stagef(sb);
                    __syncthreads();
                    if (s + 3 < nsteps) fetchf(sb);
                    mfmas();
                }
            }
        } else {
            Stage sa;
            if (nsteps > 0) fetchf(sa);
#ifdef GGW_STAMPS
            long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64();
#define GGW_T(k) { const long long t1 = clock64(); tk[k] += t1 - t0; t0 = t1; }
#else
#define GGW_T(k)
#endif
            for (int s = 0; s < nsteps; ++s) {
                __syncthreads();
                GGW_T(0)
#ifdef GGW_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                GGW_T(1)
#endif
                stagef(sa);
                GGW_T(2)
                __syncthreads();
                GGW_T(3)
                if (s + 1 < nsteps) fetchf(sa);
                mfmas();
#ifdef GGW_STAMPS
                { float d; asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(acc[1][1][15])); asm volatile("" :: "v"(d)); }
#endif
                GGW_T(4)
            }
#ifdef GGW_STAMPS
            if (a.stamps && tid == 0) { for (int k = 0; k < 5; ++k) a.stamps[(size_t)blockIdx.x * 8 + k] = tk[k]; a.stamps[(size_t)blockIdx.x * 8 + 5] = nsteps; a.stamps[(size_t)blockIdx.x * 8 + 6] = su[SU_FLAGS]; }
#endif
        }
    };
    if constexpr (PATH == 1) run(fetch_lean, stage_lean); else run(fetch, stage_to_lds);
    if (my_unit >= 0) {
        float* slab = a.slabs + ((size_t)part * a.n_units + my_unit) * SLAB_FLOATS;
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int o = (row0 & 127) + i * 32 + (q & 3) + ((q >> 2) << 3) + ((lane >> 5) << 2), k = (col0 & 127) + j * 32 + (lane & 31);
                    slab[o * H + k] = acc[i][j][q];
                }
    }
    GEN_TL(1);
    // bias gradients: column sums of the staged P rows, into the slab of the k-tile-0 unit of each o sub-tile (the only ones the finalize reads)
    if (qcol == 0) {      // (uniform: super-units that hold k tile 0)
        float* red = reinterpret_cast<float*>(tiles_h);      // [PR rows][128 OS] floats
        static_assert(sizeof(float) * PR * 128 * OS <= sizeof(__bf16) * 2 * (OS + 2) * TSZ, "bias reduction buffer fits the two tile sets");
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rp * (128 * OS) + cp * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128 * OS) {
            const int un = su[SU_UNIT + (tid >> 7) * 2];
            if (un >= 0) {
                float s2 = 0.f;
#pragma unroll
                for (int r = 0; r < PR; ++r) s2 += red[r * (128 * OS) + tid];
                a.slabs[((size_t)part * a.n_units + un) * SLAB_FLOATS + H * H + (tid & 127)] = s2;
            }
        }
    }
}

// k_gfinalize: fixed-order sums of the slabs of every destination tile -> flat gradient (every parameter written exactly once)
template <bool SPLIT, int OS, int NWV = 8 * OS, bool NOMASK = false> __global__ __launch_bounds__(64 * NWV) void k_ggradw(GArgs a) {
    const int su_i = a.su_order[blockIdx.x % a.n_sunits];
    if (a.sunits[(size_t)su_i * SUNIT_INTS + SU_FLAGS] & 1) ggradw_body<SPLIT, OS, NWV, ggw_kw_lean(SPLIT, OS, NOMASK), 1, NOMASK>(a);
    else ggradw_body<SPLIT, OS, NWV, 32, 0, NOMASK>(a);
}
struct GFinArgs { const int* fin; const float* slabs; const float* dec_slabs; float* grad; int n_units, n_parts, Hd; float* loss; float inv_n; };
__global__ __launch_bounds__(256) void k_gfinalize(GFinArgs a) {
    const int* f = a.fin + (size_t)blockIdx.x * GFIN_INTS;
    const int64_t dst = (int64_t)(unsigned)f[GF_DST_LO] | ((int64_t)f[GF_DST_HI] << 32);
    const int rows = f[GF_ROWS], cols = f[GF_COLS], ld = f[GF_LD], kind = f[GF_KIND], u0 = f[GF_UNIT0], nu = f[GF_NUNITS];
    const int SF = 8 * a.Hd + 16;
    if (a.loss && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {   // fused loss: sum the per-block partials of the decoder backward
        float l = 0.f;
        for (int b = threadIdx.x; b < NWG_DEC; b += 64) l += a.dec_slabs[(size_t)b * SF + 8 * a.Hd + 8];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m, 64);
        if (threadIdx.x == 0) *a.loss = l * a.inv_n;
    }
    if (kind == FIN_DEC_W || kind == FIN_DEC_B) {      // one decoder row piece: f[8] = its offset inside a decoder slab
        // one WAVE per element, elements dealt round-robin to the (gridDim.y x 4) waves of this op (four waves per op walked 32 elements each, one dependent round
        // trip per element: ~100 us, the tail of the whole launch)
        const int lane = threadIdx.x & 63, wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwaves = gridDim.y * 4;
        for (int e = wave; e < rows * cols; e += nwaves) {
            const int r = e / cols, cidx = e % cols;
            const int src = f[8] + cidx;
            float sum = 0.f;
            for (int b = lane; b < NWG_DEC; b += 64) sum += a.dec_slabs[(size_t)b * SF + src];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
            if (lane == 0) a.grad[dst + (int64_t)r * ld + cidx] = sum;
        }
        return;
    }
    // rows of the op are dealt to the gridDim.y workgroups of this op; per element a fixed-order sum over the (part, unit) slabs, 4 in flight
    const int rpb = (rows + gridDim.y - 1) / gridDim.y, r_lo = blockIdx.y * rpb, r_hi = min(rows, r_lo + rpb);
    if (kind == FIN_MATRIX && (cols & 3) == 0 && (ld & 3) == 0 && (dst & 3) == 0) {
        // whole 128-column tiles of an aligned matrix: four columns per thread as 16-byte loads, eight slabs in flight -- the same sums in the same order per element
        // (150 MB of slabs on the 32-limb model: 112 us with the element-wise loop below)
        const int c4 = cols >> 2, total = a.n_parts * nu;
        for (int i = threadIdx.x; i < (r_hi - r_lo) * c4; i += 256) {
            const int r = r_lo + i / c4, cidx = (i % c4) * 4, src = r * H + cidx;
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; t0 < total; t0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int t = t0 + q;
                    v[q] = t < total ? *reinterpret_cast<const f32x4*>(a.slabs + ((size_t)(t / nu) * a.n_units + u0 + t % nu) * SLAB_FLOATS + src) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                s += (v[0] + v[1]) + (v[2] + v[3]);
                if (t0 + 4 < total) s += (v[4] + v[5]) + (v[6] + v[7]);
            }
            *reinterpret_cast<f32x4*>(a.grad + dst + (int64_t)r * ld + cidx) = s;
        }
        return;
    }
    for (int i = threadIdx.x; i < (r_hi - r_lo) * cols; i += 256) {
        const int r = r_lo + i / cols, cidx = i % cols;
        float s = 0.f;
        if (kind == FIN_MATRIX || kind == FIN_BIAS) {
            const int src = kind == FIN_MATRIX ? r * H + cidx : H * H + cidx;
            const int total = a.n_parts * nu;      // slab t = part (t / nu), unit u0 + t % nu
            for (int t0 = 0; t0 < total; t0 += 4) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = t0 + q;
                    v[q] = t < total ? a.slabs[((size_t)(t / nu) * a.n_units + u0 + t % nu) * SLAB_FLOATS + src] : 0.f;
                }
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
        a.grad[dst + (int64_t)r * ld + cidx] = s;
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
int gen_create(mshgnn_plan* p, const mshgnn_desc* desc) {
    mshgnn_gen_state* g = new (std::nothrow) mshgnn_gen_state();
    if (!g) return set_err(MSHGNN_ENOMEM, "out of host memory");
    if (!compile_gen_plan(desc, g->gp)) {
        const std::string m = g->gp.err; delete g;
        return set_err(m.find("not supported") != std::string::npos ? MSHGNN_EUNSUPPORTED : MSHGNN_EINVAL, m);
    }
    p->gen = g;
    GenPlan& gp = g->gp;
    auto up = [&](void** dptr, const void* src, size_t bytes) -> int {
        HIPCHK(hipMalloc(dptr, std::max<size_t>(bytes, 16)));
        if (bytes) HIPCHK(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
        return MSHGNN_OK;
    };
    int rc;
    if ((rc = up((void**)&g->d_tables, gp.tables.data(), gp.tables.size() * 4)) != 0 ||
        (rc = up((void**)&g->d_signs, gp.signs.data(), gp.signs.size())) != 0 ||
        (rc = up((void**)&g->d_out_mask, gp.out_mask_f.data(), gp.out_mask_f.size() * 4)) != 0 ||
        (rc = up((void**)&g->d_packs, gp.packs.data(), gp.packs.size() * sizeof(PackDesc))) != 0 ||
        (rc = up((void**)&g->d_biases, gp.biases.data(), gp.biases.size() * sizeof(BiasDesc))) != 0) return rc;
    const int dec_lds = 16 * 8 * TW * 4;
    if ((rc = set_lds_attr(k_gdec_bwd<false>, dec_lds)) || (rc = set_lds_attr(k_gdec_bwd<true>, dec_lds)) ||
        (rc = set_lds_attr(k_gstep<true, 8, 8>, 16 * P16::BLK)) || (rc = set_lds_attr(k_ggradw<true, 1>, ggw_lds_bytes(true, 1))) ||
        (rc = set_lds_attr(k_ggradw<true, 2, GGW_NWV_SPLIT2>, ggw_lds_bytes(true, 2))) || (rc = set_lds_attr(k_ggradw<true, 2, GGW_NWV_SPLIT2, true>, ggw_lds_bytes(true, 2, true))) ||
        (rc = set_lds_attr(k_ggradw<true, 1, 8, true>, ggw_lds_bytes(true, 1, true))) || (rc = set_lds_attr(k_ggradw<false, 2, GGW_NWV_BF16, true>, ggw_lds_bytes(false, 2, true))) || (rc = set_lds_attr(k_ggradw<false, 2, GGW_NWV_BF16>, ggw_lds_bytes(false, 2))) || (rc = set_lds_attr(k_gstep5<false, 8, true>, gs5_lds_bytes(false, 8))) || (rc = set_lds_attr(k_gstep5<false, 8, false, 4, true>, gs5_lds_bytes(false, 8))) || (rc = set_lds_attr(k_gstep5<false, 8, false, 8, true>, gs5_lds_bytes(false, 8))) || (rc = set_lds_attr(k_gstep5<false, 8, true, 4>, gs5_lds_bytes(false, 8))) || (rc = set_lds_attr(k_gstep5<false, 8, false, 4>, gs5_lds_bytes(false, 8))) || (rc = set_lds_attr(k_gstep5<false, 8, false>, gs5_lds_bytes(false, 8))) ||
        (rc = set_lds_attr(k_gstep5<true, GS5_MB_SPLIT, true>, gs5_lds_bytes(true, GS5_MB_SPLIT))) || (rc = set_lds_attr(k_gstep5<true, GS5_MB_SPLIT, false>, gs5_lds_bytes(true, GS5_MB_SPLIT)))) return rc;
    return MSHGNN_OK;
}

void gen_destroy(mshgnn_plan* p) {
    mshgnn_gen_state* g = p->gen;
    if (!g) return;
    if (g->d_tables) (void)hipFree(g->d_tables);
    if (g->d_signs) (void)hipFree(g->d_signs);
    if (g->d_out_mask) (void)hipFree(g->d_out_mask);
    if (g->d_packs) (void)hipFree(g->d_packs);
    if (g->d_biases) (void)hipFree(g->d_biases);
    delete g;
    p->gen = nullptr;
}

const mshgnn_info* gen_info(const mshgnn_plan* p) { return &p->gen->gp.info; }
const std::vector<mshgnn_kernel_stat>* gen_kstats(const mshgnn_plan* p) { return &p->gen->gp.kstats; }
void gen_layout(const mshgnn_plan* p, int64_t batch, int training, mshgnn_ws_layout* out) { layout_gen_workspace(p->gen->gp, batch, training, out); }
int gen_host_compile(const mshgnn_desc* desc, mshgnn_info* info, int32_t* n_tables) {
    GenPlan gp;
    if (!compile_gen_plan(desc, gp)) return set_err(MSHGNN_EINVAL, gp.err);
    if (info) *info = gp.info;
    if (n_tables) *n_tables = (int32_t)gp.tables.size();
    return MSHGNN_OK;
}

// k_gagg: the aggregates of many rows of one launch (mshgnn_gen_plan.hpp, G_MANY): out[w] = round(sum_s scale_s . mask_s . X_s[w]) -- the fp32 sums of gather8 in the
// same order, rounded (or split) once, i.e. exactly the A-tile rows the job kernels staged for such a term.  A thread owns 16-byte chunks of the workgroup's GA_ROWS
// rows; GA_U sources are in flight per thread.
constexpr int GA_ROWS = 8, GA_U = 16, GA_THREADS = 512, GA_TAB = 64;      // (32 rows in flight: 24.7 us as with 16)
template <bool SPLIT> __global__ __launch_bounds__(GA_THREADS) void k_gagg(GArgs a, int agg0, int row_blocks) {
    // the sources' row / relu-byte addresses and scales, resolved by one thread each (source -> buffer -> offset is a chain of dependent loads: walked per source
    // by every thread through scalar loads it made the kernel 24 us for 32 MB)
    constexpr int U = GA_U;
    __shared__ unsigned long long t_row[GA_TAB], t_mask[GA_TAB];
    __shared__ float t_scale[GA_TAB];
    const int* op = a.aggs + (size_t)(agg0 + blockIdx.x / row_blocks) * AGG_INTS;
    const int w0 = (blockIdx.x % row_blocks) * GA_ROWS, B = a.B, Hd = a.Hd, chunks = Hd >> 3, n_src = op[AG_NSRC];
    const int* src0 = a.srcs + (size_t)op[AG_SRC0] * SRC_INTS;
    T16* out = reinterpret_cast<T16*>(a.ws + a.buf_off[op[AG_OUT_BUF]]);
    const int n_it = (GA_ROWS * chunks + GA_THREADS - 1) / GA_THREADS;
    for (int it = 0; it < n_it; ++it) {      // (uniform trip count: the table barriers are inside)
        const int idx = it * GA_THREADS + threadIdx.x;
        const bool live = idx < GA_ROWS * chunks && w0 + idx / chunks < B;
        const int w = min(w0 + idx / chunks, B - 1), col = (idx % chunks) * 8;
        const size_t roff = (size_t)w * (SPLIT ? 2 * Hd : Hd) + col;
        const int moff = (int)g_relu_byte(0, B, Hd, w, col);
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        for (int k0 = 0; k0 < n_src; k0 += GA_TAB) {
            const int nt = min(GA_TAB, n_src - k0);
            __syncthreads();
            if ((int)threadIdx.x < nt) {
                const int* sp = src0 + (size_t)(k0 + threadIdx.x) * SRC_INTS;
                const int node = sp[S_NODE], mb = sp[S_MASK];
                t_row[threadIdx.x] = reinterpret_cast<unsigned long long>(reinterpret_cast<const T16*>(a.ws + a.buf_off[sp[S_BUF]]) + g_row<SPLIT>(0, node, B, Hd));
                t_mask[threadIdx.x] = mb >= 0 ? reinterpret_cast<unsigned long long>(reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[mb]) + g_relu_byte(node, B, Hd, 0, 0)) : 0ull;
                t_scale[threadIdx.x] = __int_as_float(sp[S_SCALE]);
            }
            __syncthreads();
            for (int k = 0; k < nt; k += U) {
                u32x4 vh[U], vl[SPLIT ? U : 1]; unsigned bm[U]; float sc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u < nt) {      // (uniform)
                        const T16* rp = reinterpret_cast<const T16*>(t_row[k + u]) + roff;
                        const unsigned long long mk = t_mask[k + u];
                        vh[u] = *reinterpret_cast<const u32x4*>(rp);
                        if constexpr (SPLIT) vl[u] = *reinterpret_cast<const u32x4*>(rp + Hd);
                        bm[u] = mk ? reinterpret_cast<const uint8_t*>(mk)[moff] : 0xffu;
                        sc[u] = t_scale[k + u];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u < nt) {
                        acc8(s, chunk_mask_bits<T16>(vh[u], bm[u]), sc[u]);
                        if constexpr (SPLIT) acc8(s, chunk_mask_bits<T16>(vl[u], bm[u]), sc[u]);
                    }
                }
            }
        }
        if (live) {
            const f32x4 lo4 = f32x4{s[0], s[1], s[2], s[3]}, hi4 = f32x4{s[4], s[5], s[6], s[7]};
            T16* q = out + g_row<SPLIT>(w, op[AG_OUT_NODE], B, Hd) + col;
            if constexpr (SPLIT) {
                u32x4 hi, lo;
                split_oct(lo4, hi4, hi, lo);
                *reinterpret_cast<u32x4*>(q) = hi; *reinterpret_cast<u32x4*>(q + Hd) = lo;
            } else *reinterpret_cast<u32x4*>(q) = pack_oct(lo4, hi4);
        }
    }
}

static int g_fill(const mshgnn_plan* p, const mshgnn_ws_layout& lay, const void* const* x, const int64_t* x_pitch, char* ws, int B, int training, GArgs& a) {
    const mshgnn_gen_state* g = p->gen;
    const GenPlan& gp = g->gp;
    const mshgnn_desc& d = gp.d;
    a.ws = ws;
    for (int l = 0; l <= gp.L; ++l) { a.buf_off[BUF_X + l] = lay.x[l]; a.buf_off[BUF_DX + l] = lay.dx[l]; }
    for (int l = 0; l < gp.L; ++l) {
        a.buf_off[BUF_MASK + l] = lay.mask[l]; a.buf_off[BUF_DH + l] = lay.dh[l]; a.buf_off[BUF_HB + l] = lay.hb[l];
        a.buf_off[BUF_T1 + l] = lay.t1[l]; a.buf_off[BUF_DU + l] = lay.du[l];
    }
    a.buf_off[GBUF_MASK0] = lay.dd[0];
    {
        const size_t aggsz = align_up((size_t)B * gp.n_aggbuf * gp.Hd * gp.esize * gp.planes, 256);
        for (int l = 0; l < gp.L; ++l) { a.buf_off[GBUF_AGGF + l] = lay.dd[1] + l * aggsz; a.buf_off[GBUF_AGGB + l] = lay.dd[2] + l * aggsz; }
    }
    a.aggs = g->d_tables + gp.agg_off;
    const int in_es = gp.split ? 4 : 2;
    for (int t = 0; t < gp.NT; ++t) {
        a.x[t] = x[t]; a.pitch[t] = x_pitch ? x_pitch[t] : d.type_width[t]; a.nodes[t] = d.type_nodes[t];
        if (a.pitch[t] < d.type_width[t]) return set_err(MSHGNN_EINVAL, "x_pitch smaller than the feature width");
        a.vb[t] = vec_bytes(x[t], a.pitch[t], in_es);
    }
    a.jobs = g->d_tables + gp.job_off; a.terms = g->d_tables + gp.term_off; a.srcs = g->d_tables + gp.src_off;
    a.units = g->d_tables + gp.unit_off; a.items = g->d_tables + gp.item_off; a.sunits = g->d_tables + gp.sunit_off; a.su_order = g->d_tables + gp.su_order_off;
    a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias); a.signs = g->d_signs;
    a.slabs = reinterpret_cast<float*>(ws + lay.slabs);
    a.n_img = gp.n_img; a.B = B; a.Hd = gp.Hd; a.NCT = gp.NCT; a.tiles = (B + 63) / 64; a.training = training;
    a.n_units = gp.n_units; a.n_parts = gp.n_parts; a.n_sunits = gp.n_sunits;
    return MSHGNN_OK;
}

static bool g_raw_ok(const GenPlan& gp, const GArgs& a) {
    static const bool off = []() { const char* e = TUNE_ENV("MSHGNN_GEN_RAW5"); return e && atoi(e) == 0; }();      // (=0: the raw-input launch stays on k_gstep4)
    if (off) return false;
    for (int t = 0; t < gp.NT; ++t)
        if (a.vb[t] < 16 || (a.pitch[t] & 7) != 0 || (uint64_t)a.B * (uint64_t)a.nodes[t] * (uint64_t)a.pitch[t] * 2 >= (1ull << 32)) return false;
    return true;
}
static bool g_forced_tile() { const char* e = getenv("MSHGNN_GEN_TILE"); return e && atoi(e) >= 0; }
static int g_tile_blocks(int B, bool split) {
    const char* e = getenv("MSHGNN_GEN_TILE");      // (kernel experiments; read per launch so that a test can compare modes in one process)
    const int forced = e ? atoi(e) : -1;
    if (forced >= 0 && forced <= 9 && forced != 7) return forced;
    // bf16: 16 waves; on 128-window tiles (k_gstep5: the software pipeline of 8 waves; k_gstep4) once the batch has two of them -- half the weight stream per window, 250-258 / 252-314 us per layer launch
    // against 285-292 / 303-357 on the 32-limb model; split arithmetic: k_gstep4 at 16 waves on 64-window tiles (720-740 / 673-766 us against 803-813 / 814-925
    // for k_gstep at 8 waves, whose 16-wave form spills)
    return split ? 8 : (B >= 256 ? 8 : 3);      // (split, hidden % 512 != 0: the dispatch falls back to k_gstep at 8 / 4 waves; 8: k_gstep5 on the launches whose terms are all plain rows, k_gstep4 on the others)
}

#ifdef GEN_TIMELINE
static long long* gen_tl(const char* which) {      // MSHGNN_GEN_TL = "<launch name>:<hex device address>"
    const char* e = TUNE_ENV("MSHGNN_GEN_TL");
    if (!e) return nullptr;
    const char* c = strchr(e, ':');
    if (!c || strncmp(e, which, (size_t)(c - e)) != 0 || strlen(which) != (size_t)(c - e)) return nullptr;
    return reinterpret_cast<long long*>(strtoull(c + 1, nullptr, 16));
}
#endif
static void g_launch_jobs(const mshgnn_plan* p, const Launch& ln, GArgs a, hipStream_t st) {
    const GenPlan& gp = p->gen->gp;
    a.job0 = ln.job0;
    if (ln.n_agg > 0) {      // the launch's aggregates of many rows
        ProfScope ps(p, gp.ks_agg, st);
        const int row_blocks = (a.B + GA_ROWS - 1) / GA_ROWS;
        if (gp.split) hipLaunchKernelGGL(k_gagg<true>, dim3((unsigned)ln.n_agg * row_blocks), dim3(GA_THREADS), 0, st, a, ln.agg0, row_blocks);
        else hipLaunchKernelGGL(k_gagg<false>, dim3((unsigned)ln.n_agg * row_blocks), dim3(GA_THREADS), 0, st, a, ln.agg0, row_blocks);
    }
#ifdef GEN_TIMELINE
    a.tl = gen_tl(gp.kstats[ln.ks].name);
#endif
#ifdef GGW_STAMPS
    {   // MSHGNN_GS4_STAMPS = "<launch name>:<hex device address>" (tools/stamps_gstep4.py)
        const char* e = getenv("MSHGNN_GS4_STAMPS"); const char* c = e ? strchr(e, ':') : nullptr; const char* nm = gp.kstats[ln.ks].name;
        a.stamps = (c && strlen(nm) == (size_t)(c - e) && strncmp(e, nm, (size_t)(c - e)) == 0) ? reinterpret_cast<long long*>(strtoull(c + 1, nullptr, 16)) : nullptr;
    }
#endif
    // windows per workgroup: a packed weight fragment (8 KB per wave and K chunk, from L2) is reused for every 16-window row block of the tile
    // output tile of a workgroup: 64 windows x (32 NW) columns.  The staged A tile is shared by all NW waves, so wider tiles re-read the
    // activations fewer times (hidden / (32 NW) column groups per row of jobs): measured at h=512, B=1024: layer_fwd bf16 515 / 354 / 296 us at
    // 4 / 8 / 16 waves, split 870 / 799 / 1012 us (16 waves: 128 VGPRs, spills); 128-window tiles lose (fewer resident workgroups hide less
    // of the staging latency: 495 us at 8 waves, 389 us at 16 waves with 56 B of scratch)
    const int mode = g_tile_blocks(a.B, gp.split);      // 0: 4 waves; 1: 8 waves, 128 windows; 2: 8 waves; 3: 16 waves; 6: k_gstep4 (bf16, hidden % 512 == 0; the default from 256 windows)
    // k_gstep5 at 4 waves (256 columns per workgroup, two workgroups per CU) when the launch has more (job, tile) pairs than CUs: half-size workgroups halve the
    // launch's tail (1 032 pairs on 256 CUs: 209 -> 189 us; 1 024: equal; 256 pairs, a single round: 42 -> 46 us, so those keep 8 waves).  Same bits.  (=9 / =8 force one.)
    // The raw-input (encoder) launch on k_gstep5<RAW>: bf16 arithmetic, every input tensor 16-byte aligned with a pitch of whole chunks, row offsets inside 32 bits
    if ((mode == 8 || mode == 9) && !gp.split && gp.NCT % 2 == 0 && ln.all_raw && g_raw_ok(gp, a)) {
        a.tiles = (a.B + 127) / 128; a.njt = ln.n_jobs * a.tiles;
        ProfScope ps(p, ln.ks, st);
        if (gp.NCT % 4 != 0 || mode == 9 || (!g_forced_tile() && a.njt > p->n_cu)) {
            const int nctg = gp.NCT / 2;
            hipLaunchKernelGGL((k_gstep5<false, 8, false, 4, true>), dim3((unsigned)((a.njt + 7) / 8) * 8 * nctg), dim3(256), gs5_lds_bytes(false, 8), st, a);
        } else hipLaunchKernelGGL((k_gstep5<false, 8, false, 8, true>), dim3((unsigned)a.njt * (a.NCT / 4)), dim3(512), gs5_lds_bytes(false, 8), st, a);
        return;
    }
    // Widths that are a multiple of 256 but not of 512 (hidden = 256, 768: the reference's --hidden_size) only have the 4-wave form.
    const bool half_only = gp.NCT % 4 != 0;
    if ((mode == 9 || (mode == 8 && !g_forced_tile() && (half_only || ln.n_jobs * ((a.B + 127) / 128) > p->n_cu))) && !gp.split && gp.NCT % 2 == 0 && ln.all_plain && ln.max_terms <= GS5_MAX_TERMS) {
        a.tiles = (a.B + 127) / 128; a.njt = ln.n_jobs * a.tiles;
        const int nctg = gp.NCT / 2;
        const unsigned grid9 = (unsigned)((a.njt + 7) / 8) * 8 * nctg;
        ProfScope ps(p, ln.ks, st);
        if (ln.any_mask) hipLaunchKernelGGL((k_gstep5<false, 8, true, 4>), dim3(grid9), dim3(256), gs5_lds_bytes(false, 8), st, a);
        else hipLaunchKernelGGL((k_gstep5<false, 8, false, 4>), dim3(grid9), dim3(256), gs5_lds_bytes(false, 8), st, a);
        return;
    }
    if ((mode == 8 || mode == 9) && gp.NCT % 4 == 0 && ln.all_plain && ln.max_terms <= GS5_MAX_TERMS) {      // k_gstep5: k_gstep4's tile, software-pipelined (launches whose every term is one plain row)
        a.tiles = gp.split ? (a.B + 16 * GS5_MB_SPLIT - 1) / (16 * GS5_MB_SPLIT) : (a.B + 127) / 128;
        const unsigned grid4 = (unsigned)ln.n_jobs * a.tiles * (a.NCT / 4);
        ProfScope ps(p, ln.ks, st);
        if (gp.split) {
            if (ln.any_mask) hipLaunchKernelGGL((k_gstep5<true, GS5_MB_SPLIT, true>), dim3(grid4), dim3(512), gs5_lds_bytes(true, GS5_MB_SPLIT), st, a);
            else hipLaunchKernelGGL((k_gstep5<true, GS5_MB_SPLIT, false>), dim3(grid4), dim3(512), gs5_lds_bytes(true, GS5_MB_SPLIT), st, a);
        } else {
            if (ln.any_mask) hipLaunchKernelGGL((k_gstep5<false, 8, true>), dim3(grid4), dim3(512), gs5_lds_bytes(false, 8), st, a);
            else hipLaunchKernelGGL((k_gstep5<false, 8, false>), dim3(grid4), dim3(512), gs5_lds_bytes(false, 8), st, a);
        }
        return;
    }
    if (mode >= 6 && !gp.split && gp.NCT % 4 == 0) {      // k_gstep4: 16 waves on 128-window tiles (hidden a multiple of 512)
        a.tiles = (a.B + 127) / 128;
        const unsigned grid4 = (unsigned)ln.n_jobs * a.tiles * (a.NCT / 4);
        ProfScope ps(p, ln.ks, st);
        hipLaunchKernelGGL((k_gstep4<false, 8, 16>), dim3(grid4), dim3(1024), 8 * P16::BLK, st, a);
        return;
    }
    // (hidden = 256: the 8-wave form of k_gstep4 measured 1.51 against 1.47 ms/step for k_gstep on A1-C2, 8192 windows -- tools/time_h256.py; not dispatched)
    if (mode >= 6 && gp.split && gp.NCT % 4 == 0) {       // the split arithmetic: the same kernel at 16 waves on 64-window tiles
        a.tiles = (a.B + 63) / 64;
        const unsigned grid4 = (unsigned)ln.n_jobs * a.tiles * (a.NCT / 4);
        ProfScope ps(p, ln.ks, st);
        hipLaunchKernelGGL((k_gstep4<true, 4, 16>), dim3(grid4), dim3(1024), 2 * 4 * P16::BLK, st, a);
        return;
    }
    int nw = 4;
    if ((mode == 1 || mode == 2) && gp.NCT % 2 == 0) nw = 8;
    if (mode >= 3) nw = gp.NCT % 4 == 0 ? 16 : (gp.NCT % 2 == 0 ? 8 : 4);
    const int mb = (nw == 8 && mode == 1) ? 8 : 4;
    a.tiles = (a.B + mb * 16 - 1) / (mb * 16);
    const unsigned grid = (unsigned)ln.n_jobs * a.tiles * (a.NCT / (nw / 4));
    const int lds = mb * P16::BLK * (gp.split ? 2 : 1);
    ProfScope ps(p, ln.ks, st);
    if (gp.split) {
        if (nw == 16) hipLaunchKernelGGL((k_gstep<true, 4, 16>), dim3(grid), dim3(1024), lds, st, a);
        else if (nw == 8 && mb == 8) hipLaunchKernelGGL((k_gstep<true, 8, 8>), dim3(grid), dim3(512), lds, st, a);
        else if (nw == 8) hipLaunchKernelGGL((k_gstep<true, 4, 8>), dim3(grid), dim3(512), lds, st, a);
        else hipLaunchKernelGGL((k_gstep<true, 4, 4>), dim3(grid), dim3(256), lds, st, a);
    } else {
        if (nw == 16) hipLaunchKernelGGL((k_gstep<false, 4, 16>), dim3(grid), dim3(1024), lds, st, a);
        else if (nw == 8 && mb == 8) hipLaunchKernelGGL((k_gstep<false, 8, 8>), dim3(grid), dim3(512), lds, st, a);
        else if (nw == 8) hipLaunchKernelGGL((k_gstep<false, 4, 8>), dim3(grid), dim3(512), lds, st, a);
        else hipLaunchKernelGGL((k_gstep<false, 4, 4>), dim3(grid), dim3(256), lds, st, a);
    }
}

int gen_forward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, float* out, char* ws, int64_t batch,
                int training, hipStream_t st) {
    const mshgnn_gen_state* g = p->gen;
    const GenPlan& gp = g->gp;
    const mshgnn_desc& d = gp.d;
    mshgnn_ws_layout lay; layout_gen_workspace(gp, batch, training, &lay);
    const int B = (int)batch;
    {
        PrepArgs a{params, ws + lay.wpack, reinterpret_cast<float*>(ws + lay.bias), g->d_packs, g->d_biases, gp.n_img, (int)gp.biases.size()};
        ProfScope ps(p, gp.ks_prep, st);
        launch_prep(a, gp.split, st);
    }
    GArgs a{};
    int rc = g_fill(p, lay, x, x_pitch, ws, B, training, a);
    if (rc) return rc;
    for (const Launch& ln : gp.fwd) g_launch_jobs(p, ln, a, st);
    {
        GDecArgs da{};
        da.xl = ws + lay.x[gp.L]; da.params = params; da.out_mask = g->d_out_mask; da.out = out; da.off_w = d.off_dec_w; da.off_b = d.off_dec_b;
        da.B = B; da.Hd = gp.Hd; da.node0 = gp.type_base[d.out_type]; da.n_out = d.type_nodes[d.out_type]; da.dout = d.out_channels;
        const int64_t rows = (int64_t)B * da.n_out;
        ProfScope ps(p, gp.ks_dec_fwd, st);
        if (gp.split) hipLaunchKernelGGL(k_gdec_fwd<true>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, da);
        else hipLaunchKernelGGL(k_gdec_fwd<false>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, da);
    }
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

int gen_backward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* gout, float* gparams, char* ws,
                 int64_t batch, hipStream_t st, const float* out, const float* y, float* loss, const int32_t* labels) {
    const mshgnn_gen_state* g = p->gen;
    const GenPlan& gp = g->gp;
    const mshgnn_desc& d = gp.d;
    mshgnn_ws_layout lay; layout_gen_workspace(gp, batch, 1, &lay);
    const int B = (int)batch;
    const int n_out = d.type_nodes[d.out_type];
    {
        GDecArgs da{};
        da.xl = ws + lay.x[gp.L]; da.dxl = ws + lay.dx[gp.L]; da.params = params; da.out_mask = g->d_out_mask; da.gout = gout;
        if (gp.dhm && gp.L >= 1 && !((d.flags & MSHGNN_FLAG_BASE_MLP) && d.out_type == d.mlp_type)) { da.dhl = ws + lay.dh[gp.L - 1]; da.maskb = reinterpret_cast<const uint8_t*>(ws + lay.mask[gp.L - 1]); da.dh_only = (d.flags & MSHGNN_FLAG_RESIDUAL) ? 0 : 1; }
        da.slabs = reinterpret_cast<float*>(ws + lay.dec_slabs); da.off_w = d.off_dec_w; da.off_b = d.off_dec_b;
        da.B = B; da.Hd = gp.Hd; da.node0 = gp.type_base[d.out_type]; da.n_out = n_out; da.dout = d.out_channels;
        if (y) { da.y = y; da.out = const_cast<float*>(out); da.inv_n = 1.0f / (float)((int64_t)B * n_out * d.out_channels); }
        if (labels) { da.labels = labels; da.out = const_cast<float*>(out); da.inv_n = 1.0f / (float)((int64_t)B * n_out); }
        const int dec_lds = 16 * 8 * TW * 4;
        ProfScope ps(p, gp.ks_dec_bwd, st);
        if (gp.split) hipLaunchKernelGGL(k_gdec_bwd<true>, dim3(NWG_DEC), dim3(256), dec_lds, st, da);
        else hipLaunchKernelGGL(k_gdec_bwd<false>, dim3(NWG_DEC), dim3(256), dec_lds, st, da);
    }
    GArgs a{};
    int rc = g_fill(p, lay, x, x_pitch, ws, B, 1, a);
    if (rc) return rc;
    for (const Launch& ln : gp.bwd) g_launch_jobs(p, ln, a, st);
    {
        ProfScope ps(p, gp.ks_gradw, st);
        const unsigned grid = (unsigned)gp.n_sunits * gp.n_parts;
#ifdef GGW_STAMPS
        { const char* e = getenv("MSHGNN_GGW_STAMPS"); a.stamps = e ? reinterpret_cast<long long*>(strtoull(e, nullptr, 16)) : nullptr; }
#endif
#ifdef GEN_TIMELINE
        a.tl = gen_tl("gradw");
#endif
        // (gp.dhm: no P operand carries relu bits -- the NOMASK instantiations)
        if (gp.dhm) {
            if (gp.split && gp.su_os == 2) hipLaunchKernelGGL((k_ggradw<true, 2, GGW_NWV_SPLIT2, true>), dim3(grid), dim3(64 * GGW_NWV_SPLIT2), ggw_lds_bytes(true, 2, true), st, a);
            else if (gp.split) hipLaunchKernelGGL((k_ggradw<true, 1, 8, true>), dim3(grid), dim3(512), ggw_lds_bytes(true, 1, true), st, a);
            else if (gp.su_os == 2) hipLaunchKernelGGL((k_ggradw<false, 2, GGW_NWV_BF16, true>), dim3(grid), dim3(64 * GGW_NWV_BF16), ggw_lds_bytes(false, 2, true), st, a);
            else hipLaunchKernelGGL((k_ggradw<false, 1, 8, true>), dim3(grid), dim3(512), ggw_lds_bytes(false, 1, true), st, a);
        } else if (gp.split && gp.su_os == 2) hipLaunchKernelGGL((k_ggradw<true, 2, GGW_NWV_SPLIT2>), dim3(grid), dim3(64 * GGW_NWV_SPLIT2), ggw_lds_bytes(true, 2), st, a);
        else if (gp.split) hipLaunchKernelGGL((k_ggradw<true, 1>), dim3(grid), dim3(512), ggw_lds_bytes(true, 1), st, a);
        else if (gp.su_os == 2) hipLaunchKernelGGL((k_ggradw<false, 2, GGW_NWV_BF16>), dim3(grid), dim3(64 * GGW_NWV_BF16), ggw_lds_bytes(false, 2), st, a);
        else hipLaunchKernelGGL((k_ggradw<false, 1>), dim3(grid), dim3(512), ggw_lds_bytes(false, 1), st, a);
    }
    {
        GFinArgs fa{g->d_tables + gp.fin_off, reinterpret_cast<const float*>(ws + lay.slabs), reinterpret_cast<const float*>(ws + lay.dec_slabs), gparams,
                    gp.n_units, gp.n_parts, gp.Hd, (y || labels) ? loss : nullptr, 1.0f / (float)((int64_t)B * n_out * (labels ? 1 : d.out_channels))};
        ProfScope ps(p, gp.ks_fin, st);
        hipLaunchKernelGGL(k_gfinalize, dim3(gp.n_fin, 8), dim3(256), 0, st, fa);
    }
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}
