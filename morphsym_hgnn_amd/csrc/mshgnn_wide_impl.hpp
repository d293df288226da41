// ENGINE-DRIVEN stack kernels of the bf16 plan (hgnn_c2.py:150-166, the L x HeteroConv loop, + base_transform, residual, decoder): the MAC phase of
// a layer is a generated, hand-scheduled asm statement (mshgnn_wide_engine.inc, tools/gen_wide_engine.py: threaded code over a jump table of per-slot
// bodies, weight fragments in rotating register buffers), every node's accumulators stay in FIXED registers for the whole kernel, and a layer is ONE
// group: no packed results parked in registers while a second group is multiplied, the epilogue converts the accumulators in place.  Two geometries
// (template parameter NH = 16-window halves per tile):
//   * slab2 (NH = 1): 16-window tiles, two 4-wave workgroups per CU like the slab kernels of mshgnn.hip -- 128 accumulator registers (slots 0..15)
//     + 128 vector registers per wave: v0..v31 for the compiler, two weight buffers, the window fragment, slots 16 / 17.  LDS: 4 KB per node, 72 KB for
//     A1-C2 (the base_transform chain runs IN PLACE on the base nodes' own blocks: no scratch blocks), two workgroups per CU.
//   * wide  (NH = 2): 32-window tiles, ONE 4-wave workgroup per CU with the whole 512-entry register file per wave (three weight buffers): half the
//     weight-fragment bytes per window -- and every memory / instruction-fetch latency exposed, since nothing else runs on the SIMD.  Measured slower
//     than the slab kernels at 32 windows per CU; kept as an opt-in.
// The accumulators (slot s < 16: a[8 NH s ..], slots 16..: v[VACC + 8 NH (s - 16) ..]) are tracked by the compiler as values of 16 NH registers pinned to
// those registers in every statement that touches them (WRegs), so it allocates nothing else there, and C++ code reads / writes them through the
// wd_acc_* accessors (tools/audit_wide_isa.py lists compiler-generated instructions on accumulator registers).
// Arithmetic: the same v_mfma_f32_16x16x32_bf16 per (16 windows, 16 features, 32 K) in the same order as the slab / 8-wave kernels -> activations,
// stashes, relu bytes and dX rows are bit-identical to theirs (tests/test_engine_gpu.py).
#pragma once
#include "mshgnn_device.hpp"
#include "mshgnn_wide_engine.inc"

namespace {

using T = __bf16;
using P = Prec<__bf16>;
template <int NH> constexpr int WD_BLKB = NH * P::BLK;      // LDS bytes of one node: NH 16-window halves
// <= 18 nodes: a layer's bias rows (one per node type, this wave's 32 columns) wait in LDS behind the node blocks; the accumulators start from there
// without a global round trip, and the next layer's rows are requested right after the MAC engine, together with its header
constexpr int WD_LDSBIAS_MAXN = 18, WD_BIAS_LDS = 4 * 4 * 128;
#define WD_KERNEL(NH) __global__ __launch_bounds__(WD_THREADS, (NH) == 2 ? 1 : 2)

// ---- accumulator accessors: register (slot S, half H, element E of 8).  Every statement names the value that holds the slot as an operand pinned to
// its registers (the compiler then knows the registers are occupied and what each statement reads / writes); the instruction text addresses the single
// register literally.
#define WD_TUPLES2(X) X(0, "a[0:31]") X(1, "a[32:63]") X(2, "a[64:95]") X(3, "a[96:127]") X(4, "a[128:159]") X(5, "a[160:191]") X(6, "a[192:223]") \
                      X(7, "a[224:255]") X(8, "v[192:223]") X(9, "v[224:255]")
#define WD_TUPLES1(X) X(0, "a[0:15]") X(1, "a[16:31]") X(2, "a[32:47]") X(3, "a[48:63]") X(4, "a[64:79]") X(5, "a[80:95]") X(6, "a[96:111]") \
                      X(7, "a[112:127]") X(8, "v[112:127]")
template <int NH> constexpr int WD_VACC = NH == 2 ? 192 : 112;
template <int NH, int K> __device__ __forceinline__ auto& wd_tuple(WRegs<NH>& r) { if constexpr (K < 8) return r.a[K]; else return r.v[K - 8]; }
template <int NH, int S> constexpr int wd_tuple_of() { return S < 16 ? S / 2 : 8 + (S - 16) / 2; }
template <int NH, int S, int H, int E> constexpr int wd_reg_of() { return S < 16 ? 8 * NH * S + 8 * H + E : WD_VACC<NH> + 8 * NH * (S - 16) + 8 * H + E; }
template <int NH, int S, int H, int E> __device__ __forceinline__ float wd_acc_get1(WRegs<NH>& r) {
    float x;
    constexpr int K = wd_tuple_of<NH, S>(), IDX = wd_reg_of<NH, S, H, E>();
#define WD_X(KK, REG) if constexpr (K == KK) { \
        if constexpr (S < 16) asm("v_accvgpr_read_b32 %0, a[%c2]" : "=v"(x) : "{" REG "}"(wd_tuple<NH, KK>(r)), "i"(IDX)); \
        else asm("v_mov_b32 %0, v[%c2]" : "=v"(x) : "{" REG "}"(wd_tuple<NH, KK>(r)), "i"(IDX)); }
    if constexpr (NH == 2) { WD_TUPLES2(WD_X) } else { WD_TUPLES1(WD_X) }
#undef WD_X
    return x;
}
template <int NH, int S, int H, int E> __device__ __forceinline__ void wd_acc_set1(WRegs<NH>& r, float x) {
    constexpr int K = wd_tuple_of<NH, S>(), IDX = wd_reg_of<NH, S, H, E>();
#define WD_X(KK, REG) if constexpr (K == KK) { \
        if constexpr (S < 16) asm("v_accvgpr_write_b32 a[%c1], %2" : "+{" REG "}"(wd_tuple<NH, KK>(r)) : "i"(IDX), "v"(x)); \
        else asm("v_mov_b32 v[%c1], %2" : "+{" REG "}"(wd_tuple<NH, KK>(r)) : "i"(IDX), "v"(x)); }
    if constexpr (NH == 2) { WD_TUPLES2(WD_X) } else { WD_TUPLES1(WD_X) }
#undef WD_X
}
// first definition of the accumulator values a kernel with NS slots uses (their registers hold nothing yet)
template <int NH, int NS> __device__ __forceinline__ void wd_regs_init(WRegs<NH>& r) {
#define WD_X(KK, REG) if constexpr (KK < 8 ? 2 * KK < (NS < 16 ? NS : 16) : 16 + 2 * (KK - 8) < NS) asm volatile("" : "={" REG "}"(wd_tuple<NH, KK>(r)));
    if constexpr (NH == 2) { WD_TUPLES2(WD_X) } else { WD_TUPLES1(WD_X) }
#undef WD_X
}
template <int NH, int S, int H> __device__ __forceinline__ void wd_acc_get(WRegs<NH>& r, P::Acc& a) {
    a.c[0] = f32x4{wd_acc_get1<NH, S, H, 0>(r), wd_acc_get1<NH, S, H, 1>(r), wd_acc_get1<NH, S, H, 2>(r), wd_acc_get1<NH, S, H, 3>(r)};
    a.c[1] = f32x4{wd_acc_get1<NH, S, H, 4>(r), wd_acc_get1<NH, S, H, 5>(r), wd_acc_get1<NH, S, H, 6>(r), wd_acc_get1<NH, S, H, 7>(r)};
}
// (one statement for the eight registers of a (slot, half): element-wise statements leave hipcc's coalescer sixteen tied copies of the same
//  16 NH-register value per slot pair, and where it gives up it parks whole accumulator tuples in scratch)
template <int NH, int S, int H> __device__ __forceinline__ void wd_acc_set(WRegs<NH>& r, f32x4 c0, f32x4 c1) {
    constexpr int K = wd_tuple_of<NH, S>(), IDX = wd_reg_of<NH, S, H, 0>();
#define WD_X(KK, REG) if constexpr (K == KK) { \
        if constexpr (S < 16) asm("v_accvgpr_write_b32 a[%c1], %2\n v_accvgpr_write_b32 a[%c1+1], %3\n v_accvgpr_write_b32 a[%c1+2], %4\n v_accvgpr_write_b32 a[%c1+3], %5\n" \
                                  "v_accvgpr_write_b32 a[%c1+4], %6\n v_accvgpr_write_b32 a[%c1+5], %7\n v_accvgpr_write_b32 a[%c1+6], %8\n v_accvgpr_write_b32 a[%c1+7], %9" \
                                  : "+{" REG "}"(wd_tuple<NH, KK>(r)) : "i"(IDX), "v"(c0[0]), "v"(c0[1]), "v"(c0[2]), "v"(c0[3]), "v"(c1[0]), "v"(c1[1]), "v"(c1[2]), "v"(c1[3])); \
        else asm("v_mov_b32 v[%c1], %2\n v_mov_b32 v[%c1+1], %3\n v_mov_b32 v[%c1+2], %4\n v_mov_b32 v[%c1+3], %5\n v_mov_b32 v[%c1+4], %6\n v_mov_b32 v[%c1+5], %7\n" \
                 "v_mov_b32 v[%c1+6], %8\n v_mov_b32 v[%c1+7], %9" \
                 : "+{" REG "}"(wd_tuple<NH, KK>(r)) : "i"(IDX), "v"(c0[0]), "v"(c0[1]), "v"(c0[2]), "v"(c0[3]), "v"(c1[0]), "v"(c1[1]), "v"(c1[2]), "v"(c1[3])); }
    if constexpr (NH == 2) { WD_TUPLES2(WD_X) } else { WD_TUPLES1(WD_X) }
#undef WD_X
}
// all halves of a slot at once
template <int NH, int S> __device__ __forceinline__ void wd_acc_get_all(WRegs<NH>& r, P::Acc (&c)[NH]) {
    wd_acc_get<NH, S, 0>(r, c[0]);
    if constexpr (NH == 2) wd_acc_get<NH, S, 1>(r, c[1]);
}
template <int NH, int S> __device__ __forceinline__ void wd_acc_set_all(WRegs<NH>& r, const f32x4 (&c0)[NH], const f32x4 (&c1)[NH]) {
    wd_acc_set<NH, S, 0>(r, c0[0], c1[0]);
    if constexpr (NH == 2) wd_acc_set<NH, S, 1>(r, c0[1], c1[1]);
}
// compile-time loop over the accumulator slots
template <int U, int N, typename F> __device__ __forceinline__ void wd_for(F&& f) {
    if constexpr (U < N) { f(std::integral_constant<int, U>{}); wd_for<U + 1, N>(f); }
}

// layer program: entry stream, pack id per segment, {first block, second block, segments, first body} (mshgnn_plan.hpp, emit_wide)
struct WProg {
    int prog, pk, misc;
    __device__ __forceinline__ WProg() : prog(0), pk(0), misc(0) {}
    __device__ __forceinline__ WProg(const int* t, int lane) : prog(t[lane]), pk(t[64 + lane]), misc(t[128 + lane]) {}
    __device__ __forceinline__ void settle() { asm volatile("" : "+v"(prog), "+v"(pk), "+v"(misc)); }      // see FProg::settle
};
// hdr1 / bbase / init: forward layers whose bias rows wait in LDS (k_eng_fwd) let the engine start the accumulators from them
template <int NH, int NS> __device__ __forceinline__ void wd_run(WRegs<NH>& r, const WProg& wp, const char* smem, const T* wpack, int wn, int lane, const AOff<T>& ao,
                                                                 int hdr1 = 0, int bbase = 0, int init = 0) {
    static_assert(FH_NEXT + 8 - 64 == 32, "tools/gen_wide_engine.py: HDR_TYPE_LANE");
    const int m = __builtin_amdgcn_readlane(wp.misc, 0);
    wd_engine<NH, NS>(r, wp.prog, wp.pk, lane * 16, ao.o, reinterpret_cast<const char*>(wpack) + wn * (P::NBV * 64 * 16), (m & 0xff) * WD_BLKB<NH>,
                      ((m >> 8) & 0xff) * WD_BLKB<NH>, (m >> 16) & 0xff, (m >> 24) & 0xff, hdr1, bbase, init);
}

template <int NH> __device__ __forceinline__ int wd_chunk(int node, int h, int row16, int c) { return node * WD_BLKB<NH> + lds_chunk<T>(h, row16, c); }

// window operand of one MAC and the base_transform chain's GEMMs (compiler-scheduled, accumulators in VGPRs)
template <int NH> struct WAcc { P::Acc h[NH]; };
template <int NH> struct XFrag { bf16x8 v[NH][4]; };
template <int NH> __device__ __forceinline__ void wd_load_x(XFrag<NH>& x, const char* smem, int node, const AOff<T>& ao) {
    const int base = node * WD_BLKB<NH>;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + base + h * P::BLK + ao.o[t]);
            x.v[h][t] = __builtin_bit_cast(bf16x8, v);
        }
}
__device__ __forceinline__ void wd_mfma_v(f32x4& c, const bf16x8& w, const bf16x8& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
// hipcc pads nothing around an asm MFMA (guide 5.7): fresh value -> MFMA, MFMA -> reader
template <int NH> __device__ __forceinline__ void wd_fence_v(WAcc<NH>& a) {
    if constexpr (NH == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a.h[0].c[0]), "+v"(a.h[0].c[1]), "+v"(a.h[1].c[0]), "+v"(a.h[1].c[1]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a.h[0].c[0]), "+v"(a.h[0].c[1]));
}
// one base_transform GEMM in place: tm[u] += LDS[node u] . W for the first nm nodes (per accumulator the K steps run in the order of mac())
template <int NH, int NM>
__device__ __forceinline__ void wd_chain_gemm(WAcc<NH> (&tm)[NM], int nm, const char* smem, const P::BFrag& bf, const AOff<T>& ao) {
    XFrag<NH> x;
#pragma unroll
    for (int u = 0; u < NM; ++u) wd_fence_v<NH>(tm[u]);
#pragma unroll
    for (int u = 0; u < NM; ++u)
        if (u < nm) {
            wd_load_x<NH>(x, smem, u, ao);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    wd_mfma_v(tm[u].h[h].c[0], bf.v[t], x.v[h][t]);
                    wd_mfma_v(tm[u].h[h].c[1], bf.v[4 + t], x.v[h][t]);
                }
        }
#pragma unroll
    for (int u = 0; u < NM; ++u) wd_fence_v<NH>(tm[u]);
}

// ------------------------------------------------------------------------------------------------------
// decoder (+ fused wrapper loss and decoder backward) on the X_L tile in LDS: the tail of decoder_tail_impl (mshgnn_device.hpp) for a 32-window
// tile.  thread = (row16, 8-column chunk); a pass takes two out-type nodes x NH halves, every global load of the pass before its first store.
// ------------------------------------------------------------------------------------------------------
template <int NH, int DMAX>
__device__ __forceinline__ void wd_decoder_tail(const StackArgs& a, char* smem, int tid, int lane, int wv, int w0, int B) {
    const int c = tid & 15, row16 = (tid >> 4) & 15;
    const float* W = a.params + a.off_dec_w;
    const bool ce = a.labels != nullptr, fuse = a.y != nullptr || ce;
    T* dxl = reinterpret_cast<T*>(a.ws + a.dx_off[a.L]);
    float accw[DMAX][8], accb[DMAX], lsum = 0.f;
    if (fuse) {
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) { accb[dd] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) accw[dd][e] = 0.f; }
    }
    float Wv[DMAX][8], bv[DMAX];
#pragma unroll
    for (int dd = 0; dd < DMAX; ++dd) {
        const int dc = min(dd, a.dout - 1);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(W + dc * H + c * 8), wb = *reinterpret_cast<const f32x4*>(W + dc * H + c * 8 + 4);
        Wv[dd][0] = wa[0]; Wv[dd][1] = wa[1]; Wv[dd][2] = wa[2]; Wv[dd][3] = wa[3];
        Wv[dd][4] = wb[0]; Wv[dd][5] = wb[1]; Wv[dd][6] = wb[2]; Wv[dd][7] = wb[3];
        bv[dd] = a.params[a.off_dec_b + dc];
    }
    for (int f0 = 0; f0 < a.n_out; f0 += 2) {
        float ov[2 * NH][DMAX], dxv[2 * NH][8], mk[2][DMAX], yv[2 * NH][DMAX]; int labv[2 * NH];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = min(f0 + i, a.n_out - 1);
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) mk[i][dd] = a.out_mask[f * a.dout + min(dd, a.dout - 1)];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const size_t r = (size_t)min(w0 + 16 * h + row16, B - 1) * a.n_out + f;
                labv[NH * i + h] = ce ? a.labels[r] != 0 : 0;
#pragma unroll
                for (int dd = 0; dd < DMAX; ++dd) yv[NH * i + h][dd] = a.y ? a.y[r * a.dout + min(dd, a.dout - 1)] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < 2 * NH; ++it) {
            const int i = it / NH, h = it % NH, f = f0 + i;
            const bool live = f < a.n_out;
            f32x4 x0, x1;
            load_oct(reinterpret_cast<const T*>(smem + wd_chunk<NH>(a.node0 + (live ? f : f0), h, row16, c)), x0, x1);
            const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) dxv[it][e] = 0.f;
            const bool ok = live && w0 + 16 * h + row16 < B;
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) {
                ov[it][dd] = 0.f;
                if (dd < a.dout && live) {
                    float sum = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum += x[e] * Wv[dd][e];
                    sum = row16_sum(sum);
                    ov[it][dd] = (sum + bv[dd]) * mk[i][dd];
                }
            }
            if (fuse && ok) {
                float ce_g[2] = {0.f, 0.f};
                if (ce) {
                    const float l0 = ov[it][0], l1 = ov[it][1];
                    const float m = fmaxf(l0, l1), e0 = expf(l0 - m), e1 = expf(l1 - m), se = e0 + e1;
                    ce_g[0] = (e0 / se - (labv[it] ? 0.f : 1.f)) * a.inv_n; ce_g[1] = (e1 / se - (labv[it] ? 1.f : 0.f)) * a.inv_n;
                    if (c == 0) lsum += (m + logf(se)) - (labv[it] ? l1 : l0);
                }
#pragma unroll
                for (int dd = 0; dd < DMAX; ++dd) {
                    if (dd < a.dout) {
                        float g;
                        if (ce) g = ce_g[dd & 1] * mk[i][dd];
                        else {
                            const float dlt = ov[it][dd] - yv[it][dd];
                            g = 2.0f * dlt * a.inv_n * mk[i][dd];
                            if (c == 0) lsum += dlt * dlt;
                        }
                        accb[dd] += g;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { accw[dd][e] += g * x[e]; dxv[it][e] += g * Wv[dd][e]; }
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2 * NH; ++it) {
            const int i = it / NH, h = it % NH, f = f0 + i, w = w0 + 16 * h + row16;
            const bool ok = f < a.n_out && w < B;
            const size_t r = (size_t)w * a.n_out + f;
            if (c == 0 && ok) {
#pragma unroll
                for (int dd = 0; dd < DMAX; ++dd) if (dd < a.dout) a.out[r * a.dout + dd] = ov[it][dd];
            }
            if (fuse && ok) store8<T>(dxl + act_idx(w, a.node0 + f, B) + c * 8, dxv[it]);
        }
    }
    if (fuse) {
        // reduce over the 4 rows of the wave (lanes 16 apart), then over the 4 waves through LDS (the X tile is dead after the barrier)
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) {
            if (dd < a.dout) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { accw[dd][e] += __shfl_xor(accw[dd][e], 16, 64); accw[dd][e] += __shfl_xor(accw[dd][e], 32, 64); }
                accb[dd] += __shfl_xor(accb[dd], 16, 64); accb[dd] += __shfl_xor(accb[dd], 32, 64);
            }
        }
        lsum += __shfl_xor(lsum, 16, 64); lsum += __shfl_xor(lsum, 32, 64);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [4 waves][8 H + 16]
        if (lane < 16) {
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[wv * DEC_SLAB_FLOATS + dd * H + lane * 8 + e] = accw[dd][e];
                if (lane == 0) red[wv * DEC_SLAB_FLOATS + 8 * H + dd] = accb[dd];
            }
            if (lane == 0) red[wv * DEC_SLAB_FLOATS + 8 * H + 8] = lsum;
        }
        __syncthreads();
        float* slab = a.dec_slabs + (size_t)blockIdx.x * DEC_SLAB_FLOATS;
        for (int i = tid; i < 8 * H + 9; i += WD_THREADS) {
            if (i >= a.dout * H && i < 8 * H) continue;       // rows of unused output channels (k_finalize reads dout rows only)
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < WD_THREADS / 64; ++k) s2 += red[k * DEC_SLAB_FLOATS + i];
            slab[i] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
template <int NH, int NS, int NM, int DMAX> WD_KERNEL(NH) void k_eng_fwd(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BLKB = WD_BLKB<NH>;
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w0 = blockIdx.x * 16 * NH, B = a.B, NN = a.NN;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    const bool train = a.training != 0;
    WRegs<NH> R; wd_regs_init<NH, NS>(R);
    if constexpr (NH == 1) stack_stagger(a);
    FS_STAMP(0);

    // layer 0's header and program stream in under the tile load
    FHdr fhn(a.tables + a.prog_off[0], lane);
    WProg wpn(a.tables + a.prog_off[0] + FH_SIZE, lane);
    {   // X_0 tile -> LDS: thread = (row16, 16-byte chunk), six rows per pass
        const T* src = reinterpret_cast<const T*>(a.tile_in);
        const int row16 = tid >> 4, c = tid & 15;
        constexpr int NB = 6 / NH;
        for (int n0 = 0; n0 < NN; n0 += NB) {
            u32x4 v[NB][NH];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    v[i][h] = u32x4{0, 0, 0, 0};
                    const int w = w0 + 16 * h + row16;
                    if (n0 + i < NN && w < B) v[i][h] = *reinterpret_cast<const u32x4*>(src + act_idx(w, n0 + i, B) + c * P::EPC);
                }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    if (n0 + i < NN) *reinterpret_cast<u32x4*>(smem + wd_chunk<NH>(n0 + i, h, row16, c)) = v[i][h];
        }
    }
    constexpr bool LB = NS <= WD_LDSBIAS_MAXN;
    const int bsm = NN * BLKB + wn * 512;      // this wave's bias rows: [type][32 floats]
    // lanes 0..31: 16-byte chunk (lane & 7) of the row of type (lane >> 3); rows named by four consecutive header entries
    auto bias_fetch = [&](const FHdr& h, int base) {
        const int t = (lane >> 3) & 3, r0 = h[base], r1 = h[base + 1], r2 = h[base + 2], r3 = h[base + 3];
        const int row = t == 0 ? r0 : t == 1 ? r1 : t == 2 ? r2 : r3;
        return *reinterpret_cast<const f32x4*>(a.bias + (size_t)row * H + wn * 32 + (lane & 7) * 4);
    };
    auto bias_put = [&](f32x4 v) { if (lane < 32) *reinterpret_cast<f32x4*>(smem + bsm + (lane >> 3) * 128 + (lane & 7) * 16) = v; };
    fhn.settle(); wpn.settle();
    if constexpr (LB) bias_put(bias_fetch(fhn, FH_NEXT));
    __syncthreads();
    FS_STAMP(1);

    for (int l = 0; l < a.L; ++l) {
        const FHdr fh = fhn;
        const WProg wp = wpn;
        const int nmlp = fh[FH_NMLP];
        const bool residual = (fh[FH_FLAGS] & FF_RESIDUAL) != 0;
        if constexpr (!LB) {
        // accumulators start at the bias row of their node's type; the rows of six nodes are requested before the first of their accumulators is
            // written (three memory round trips per layer instead of one per node; dead nodes: their header entry names the plan's row of zeros, the select
            // below is kept because the straight-line version makes hipcc park finished accumulator tuples in scratch)
            wd_for<0, (NS + 5) / 6>([&](auto gc) {
                constexpr int G = decltype(gc)::value;
                f32x4 bv[6][2];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int u = 6 * G + i;
                    const bool live = u < NN && u < NS && fh[FH_KIND + (u < NS ? u : 0)] != NK_DEAD;
                    const float* bias = a.bias + (size_t)(live ? fh[FH_BIAS + (u < NS ? u : 0)] : 0) * H + wn * 32;
                    bv[i][0] = *reinterpret_cast<const f32x4*>(bias + c_feat(0, lane)); bv[i][1] = *reinterpret_cast<const f32x4*>(bias + c_feat(1, lane));
                }
                wd_for<0, 6>([&](auto ic) {
                    constexpr int I = decltype(ic)::value, U = 6 * G + I;
                    if constexpr (U < NS) {
                        const bool live = U < NN && fh[FH_KIND + U] != NK_DEAD;
                        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                        f32x4 b0[NH], b1[NH];
#pragma unroll
                        for (int h = 0; h < NH; ++h) { b0[h] = live ? bv[I][0] : z; b1[h] = live ? bv[I][1] : z; }
                        wd_acc_set_all<NH, U>(R, b0, b1);
                    }
                });
            });
        }
        FS_STAMP(20 + l);
        {   // (fragment offsets rebuilt where they are used: four registers that would otherwise be carried -- spilled -- across the engine)
            const AOff<T> ao(opaque(lane));
            // <= 18 nodes: the engine itself starts the accumulators from this wave's bias rows in LDS (row of type t at bsm + 128 t)
            if constexpr (LB) wd_run<NH, NS>(R, wp, smem, wpack, wn, lane, ao, fh.h1, bsm + c_oct(opaque(lane)) * 4, 1);      // (the dynamic LDS segment starts at address 0, as for the block offsets)
            else wd_run<NH, NS>(R, wp, smem, wpack, wn, lane, ao, wp.prog, wp.prog, 0);
        }
        if (l + 1 < a.L) {    // the next layer's header and program: requested here, so that they do not live (spilled, one serialised round trip each) across the
                              // MAC engine, and settled before the epilogue's first store
            fhn = FHdr(a.tables + a.prog_off[l + 1], lane);
            wpn = WProg(a.tables + a.prog_off[l + 1] + FH_SIZE, lane);
        }
        f32x4 brow = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (LB) brow = bias_fetch(fh, FH_NEXT + 4);      // the next layer's bias rows (this wave's copy is read only by this wave: no barrier)
        FS_STAMP(2 + 4 * l);
        __syncthreads();   // every wave is done reading X_l: the node blocks may be overwritten
        FS_STAMP(3 + 4 * l);

        // lane constants of the epilogue rebuilt per layer from an opaque copy of the lane id: the per-node addresses derived from them would
        // otherwise be hoisted out of the layer loop and live (spilled) across the MAC phase
        const int lq = opaque(lane);
        const int win = c_win(lq), col = wn * 32 + c_oct(lq);
        const int loff = lds_chunk<T>(0, win, col / P::EPC);
        T* xo = reinterpret_cast<T*>(a.ws + a.x_off[l + 1]);
        uint8_t* maskbytes = reinterpret_cast<uint8_t*>(a.ws + a.mask_off[l]);
        u32x4 resm[NM][NH], hpk[NM][NH];
        if (nmlp > 0) {
            // base_transform, in place on the nodes' own blocks: their residual rows wait in registers, H goes into the blocks
            wd_for<0, NM>([&](auto uc) {
                constexpr int U = decltype(uc)::value;
#pragma unroll
                for (int h = 0; h < NH; ++h) resm[U][h] = hpk[U][h] = u32x4{0, 0, 0, 0};
                if (U < nmlp) {
                    P::Acc c[NH]; wd_acc_get_all<NH, U>(R, c);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        resm[U][h] = *reinterpret_cast<const u32x4*>(smem + U * BLKB + h * P::BLK + loff);
                        hpk[U][h] = pack_oct(c[h].c[0], c[h].c[1]);
                    }
                }
            });
            wd_for<0, NM>([&](auto uc) {
                constexpr int U = decltype(uc)::value;
                if (U < nmlp) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) *reinterpret_cast<u32x4*>(smem + U * BLKB + h * P::BLK + loff) = hpk[U][h];
                }
            });
        }
        fhn.settle(); wpn.settle();
        if constexpr (LB) bias_put(brow);
        // X_{l+1}[n] = relu(H[n]) (+ X_l[n]) for the relu nodes, in place; stash + relu bytes on the side.  Six nodes at a time: their residual octets
        // are read before the first write of the group (the compiler cannot move an LDS read above an LDS write that may alias it)
        wd_for<0, (NS + 5) / 6>([&](auto gc) {
            constexpr int G = decltype(gc)::value;
            u32x4 rres[6][NH];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int u = 6 * G + i;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    rres[i][h] = u32x4{0, 0, 0, 0};
                    if (u < NN && u < NS && residual && fh[FH_KIND + (u < NS ? u : 0)] == NK_RELU)
                        rres[i][h] = *reinterpret_cast<const u32x4*>(smem + u * BLKB + h * P::BLK + loff);
                }
            }
            wd_for<0, 6>([&](auto ic) {
                constexpr int I = decltype(ic)::value, U = 6 * G + I;
                if constexpr (U < NS) {
                    if (U < NN && fh[FH_KIND + U] == NK_RELU) {
                        P::Acc c[NH]; wd_acc_get_all<NH, U>(R, c);
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            const unsigned bits = relu_with_bits<T>(c[h]);
                            f32x4 r0, r1; unpack_oct(rres[I][h], r0, r1);
                            const u32x4 pk = pack_oct(c[h].c[0] + r0, c[h].c[1] + r1);
                            *reinterpret_cast<u32x4*>(smem + U * BLKB + h * P::BLK + loff) = pk;
                            if (train) {
                                const int w = w0 + 16 * h + win;
                                if (w0 + 16 * h < B) maskbytes[relu_tile_base(U, B, NH * blockIdx.x + h, wn) + lane] = (uint8_t)bits;
                                if (w < B) *reinterpret_cast<u32x4*>(xo + act_idx(w, U, B) + col) = pk;
                            }
                        }
                    }
                }
            });
        });
        FS_STAMP(4 + 4 * l);
        if (nmlp > 0) {
            WAcc<NH> tm[NM];
            u32x4 tpk[NM][NH];
#pragma unroll
            for (int u = 0; u < NM; ++u) {
                acc_init_bias<T>(tm[u].h[0], a.bias + (size_t)fh[FH_B1] * H, wn, lane);
                if constexpr (NH == 2) tm[u].h[NH - 1] = tm[u].h[0];
            }
            const AOff<T> ao(opaque(lane));
            P::BFrag bf;      // (one fragment at a time, requested right before its GEMM: 64 registers held across the relu epilogue made it spill, and a
                              //  scratch reload next to pending stores is a full vmcnt(0) drain)
            load_bfrag<T>(bf, wpack, fh[FH_W1], wn, lane);
            __syncthreads();      // H of every wave is in the blocks
            wd_chain_gemm<NH, NM>(tm, nmlp, smem, bf, ao);
            load_bfrag<T>(bf, wpack, fh[FH_W2], wn, lane);
            __syncthreads();      // all reads of H done before T1 overwrites the blocks
#pragma unroll
            for (int u = 0; u < NM; ++u)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    tpk[u][h] = u32x4{0, 0, 0, 0};
                    if (u < nmlp) {
                        tpk[u][h] = pack_oct(relu4(tm[u].h[h].c[0]), relu4(tm[u].h[h].c[1]));
                        *reinterpret_cast<u32x4*>(smem + u * BLKB + h * P::BLK + loff) = tpk[u][h];
                    }
                }
#pragma unroll
            for (int u = 0; u < NM; ++u) {
                acc_init_bias<T>(tm[u].h[0], a.bias + (size_t)fh[FH_B2] * H, wn, lane);
                if constexpr (NH == 2) tm[u].h[NH - 1] = tm[u].h[0];
            }
            __syncthreads();
            wd_chain_gemm<NH, NM>(tm, nmlp, smem, bf, ao);
            __syncthreads();      // all reads of T1 done before X_{l+1} overwrites the blocks
            T* hb = reinterpret_cast<T*>(a.ws + a.hb_off[l]);
            T* t1 = reinterpret_cast<T*>(a.ws + a.t1_off[l]);
#pragma unroll
            for (int u = 0; u < NM; ++u)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if (u < nmlp) {
                        f32x4 y0 = tm[u].h[h].c[0], y1 = tm[u].h[h].c[1];
                        if (residual) { f32x4 r0, r1; unpack_oct(resm[u][h], r0, r1); y0 += r0; y1 += r1; }
                        const u32x4 pk = pack_oct(y0, y1);
                        *reinterpret_cast<u32x4*>(smem + u * BLKB + h * P::BLK + loff) = pk;
                        const int w = w0 + 16 * h + win;
                        if (train && w < B) {
                            *reinterpret_cast<u32x4*>(hb + act_idx(w, u, B) + col) = hpk[u][h];
                            *reinterpret_cast<u32x4*>(t1 + act_idx(w, u, B) + col) = tpk[u][h];
                            *reinterpret_cast<u32x4*>(xo + act_idx(w, u, B) + col) = pk;
                        }
                    }
                }
        }
        __syncthreads();
        FS_STAMP(5 + 4 * l);
    }
    wd_decoder_tail<NH, DMAX>(a, smem, tid, lane, wn, w0, B);
    FS_STAMP(30);
}

// ------------------------------------------------------------------------------------------------------
// backward: the L backward layers of a tile.  The accumulators carry the residual term dX_{l+1} from layer to layer (rounded to the stored bf16
// value, as the slab kernel's packed rows), and a layer's epilogue writes the NEXT layer's dH straight into LDS: relu nodes masked with the relu
// bytes of layer l - 1, base_transform nodes unmasked for the chain.
// ------------------------------------------------------------------------------------------------------
template <int NH, int NS, int NM> WD_KERNEL(NH) void k_eng_bwd(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BLKB = WD_BLKB<NH>;
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w0 = blockIdx.x * 16 * NH, B = a.B, NN = a.NN;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    WRegs<NH> R; wd_regs_init<NH, NS>(R);
    if constexpr (NH == 1) stack_stagger(a);
    FS_STAMP(0);

    FHdr bhn(a.tables + a.prog_off[a.L - 1], lane);
    WProg wpn(a.tables + a.prog_off[a.L - 1] + FH_SIZE, lane);
    {   // dX_L -> accumulators (residual) and, masked with the last layer's relu bytes, LDS: each lane loads the octets it owns
        const T* src = reinterpret_cast<const T*>(a.tile_in);
        const uint8_t* mbytes = reinterpret_cast<const uint8_t*>(a.ws + a.mask_off[a.L - 1]);
        const int lq = opaque(lane);
        const int win = c_win(lq), col = wn * 32 + c_oct(lq), loff = lds_chunk<T>(0, win, col / P::EPC);
        wd_for<0, NS>([&](auto uc) {
            constexpr int U = decltype(uc)::value;
            const int kind = U < NN ? bhn[FH_KIND + U] : NK_DEAD;
            u32x4 pk[NH]; unsigned mb[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) { pk[h] = u32x4{0, 0, 0, 0}; mb[h] = 0xffu; }
            if (kind != NK_DEAD) {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    pk[h] = *reinterpret_cast<const u32x4*>(src + act_idx(min(w0 + 16 * h + win, B - 1), U, B) + col);
                    if (kind == NK_RELU && w0 + 16 * h < B) mb[h] = mbytes[relu_tile_base(U, B, NH * blockIdx.x + h, wn) + lane];
                }
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    *reinterpret_cast<u32x4*>(smem + U * BLKB + h * P::BLK + loff) = kind == NK_RELU ? chunk_mask_bits<T>(pk[h], mb[h]) : pk[h];
            }
            const bool res = kind != NK_DEAD && bhn[FH_RES + U] != 0;
            f32x4 c0[NH], c1[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                c0[h] = c1[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (res) unpack_oct(pk[h], c0[h], c1[h]);
            }
            wd_acc_set_all<NH, U>(R, c0, c1);
        });
    }
    __syncthreads();
    bhn.settle(); wpn.settle();
    FS_STAMP(1);

    for (int l = a.L - 1; l >= 0; --l) {
        const FHdr bh = bhn;
        const WProg wp = wpn;
        FS_STAMP(2 + 5 * (a.L - 1 - l));
        const int nmlp = bh[FH_NMLP], flags = bh[FH_FLAGS];
        const bool enc_mask = (flags & FF_ENC_MASK) != 0;
        int lq = opaque(lane);
        int win = c_win(lq), col = wn * 32 + c_oct(lq), loff = lds_chunk<T>(0, win, col / P::EPC);

        if (nmlp > 0) {
            // dT1 = dY W2 ; dU = dT1 . (T1 > 0) ; dH = dU W1   (backward of base_transform, in place on nodes 0..nmlp-1)
            const T* t1 = reinterpret_cast<const T*>(a.ws + a.t1_off[l]);
            T* du = reinterpret_cast<T*>(a.ws + a.du_off[l]);
            T* dh = reinterpret_cast<T*>(a.ws + a.dh_off[l]);
            const AOff<T> ao(opaque(lane));
            P::BFrag bf;
            WAcc<NH> tm[NM];
            u32x4 traw[NM][NH], dupk[NM][NH];
            load_bfrag<T>(bf, wpack, bh[FH_W2], wn, lane);
#pragma unroll
            for (int u = 0; u < NM; ++u)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    traw[u][h] = u32x4{0, 0, 0, 0};
                    if (u < nmlp) traw[u][h] = *reinterpret_cast<const u32x4*>(t1 + act_idx(min(w0 + 16 * h + win, B - 1), u, B) + col);
                    acc_fill(tm[u].h[h], 0.f);
                }
            wd_chain_gemm<NH, NM>(tm, nmlp, smem, bf, ao);
            load_bfrag<T>(bf, wpack, bh[FH_W1], wn, lane);
            __syncthreads();   // all reads of the dY blocks done
#pragma unroll
            for (int u = 0; u < NM; ++u)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    dupk[u][h] = u32x4{0, 0, 0, 0};
                    if (u < nmlp) {
                        f32x4 t0, t1v, r0, r1; unpack_oct(traw[u][h], t0, t1v);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { r0[j] = t0[j] > 0.f ? tm[u].h[h].c[0][j] : 0.f; r1[j] = t1v[j] > 0.f ? tm[u].h[h].c[1][j] : 0.f; }
                        dupk[u][h] = pack_oct(r0, r1);
                        *reinterpret_cast<u32x4*>(smem + u * BLKB + h * P::BLK + loff) = dupk[u][h];
                    }
                    acc_fill(tm[u].h[h], 0.f);
                }
            __syncthreads();
            wd_chain_gemm<NH, NM>(tm, nmlp, smem, bf, ao);
            __syncthreads();   // all reads of the dU blocks done
#pragma unroll
            for (int u = 0; u < NM; ++u)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if (u < nmlp) {
                        const u32x4 hp = pack_oct(tm[u].h[h].c[0], tm[u].h[h].c[1]);
                        *reinterpret_cast<u32x4*>(smem + u * BLKB + h * P::BLK + loff) = hp;
                        const int w = w0 + 16 * h + win;
                        if (w < B) {
                            *reinterpret_cast<u32x4*>(du + act_idx(w, u, B) + col) = dupk[u][h];
                            *reinterpret_cast<u32x4*>(dh + act_idx(w, u, B) + col) = hp;
                        }
                    }
                }
            __syncthreads();
        }

        // dX_l[j] = (residual) + dH_j W_rootsum + sum_r sum_{j->i} dH_i W_rel^r
        FS_STAMP(3 + 5 * (a.L - 1 - l));
        {
            const AOff<T> ao(opaque(lane));
            wd_run<NH, NS>(R, wp, smem, wpack, wn, lane, ao, wp.prog, wp.prog, 0);      // (no bias start: any register serves as the two unused operands)
        }
        FS_STAMP(4 + 5 * (a.L - 1 - l));
        if (l > 0) {          // the next layer's header and program (not carried across the MAC engine; what this epilogue needs of the next layer is in bh[FH_NEXT])
            bhn = FHdr(a.tables + a.prog_off[l - 1], lane);
            wpn = WProg(a.tables + a.prog_off[l - 1] + FH_SIZE, lane);
        }
        __syncthreads();   // every wave is done reading dH_l
        FS_STAMP(5 + 5 * (a.L - 1 - l));
        lq = opaque(lane); win = c_win(lq); col = wn * 32 + c_oct(lq); loff = lds_chunk<T>(0, win, col / P::EPC);      // (rebuilt: nothing derived from them lives across the MAC phase)
        T* dxo = reinterpret_cast<T*>(a.ws + a.dx_off[l]);
        const uint8_t* mbytes = reinterpret_cast<const uint8_t*>(a.ws + (l > 0 ? a.mask_off[l - 1] : a.mask0_off));
        // the relu bytes of the next layer's mask (layer 0: the encoder activation's) of every node are requested before the first store: one round trip
        // per layer, which also brings the next header and program in
        unsigned mb[NS][NH];
        wd_for<0, NS>([&](auto uc) {
            constexpr int U = decltype(uc)::value;
            const bool outp = U < NN && bh[FH_OUT + U] != 0;
            const bool want = outp && (l > 0 ? (bh[FH_NEXT + U] & 3) == NK_RELU : enc_mask);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                mb[U][h] = 0xffu;
                if (want && w0 + 16 * h < B) mb[U][h] = mbytes[relu_tile_base(U, B, NH * blockIdx.x + h, wn) + lane];
            }
        });
        bhn.settle(); wpn.settle();
        wd_for<0, NS>([&](auto uc) {
            constexpr int U = decltype(uc)::value;
            const bool outp = U < NN && bh[FH_OUT + U] != 0;
            const int nx = (l > 0 && U < NN) ? bh[FH_NEXT + U] : 0;
            const int nkind = nx & 3;
            const bool nres = nkind != NK_DEAD && (nx & 4) != 0;
            f32x4 n0[NH], n1[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) n0[h] = n1[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (outp) {
                P::Acc c[NH]; wd_acc_get_all<NH, U>(R, c);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    u32x4 pk = pack_oct(c[h].c[0], c[h].c[1]);
                    if (l == 0 && enc_mask) pk = chunk_mask_bits<T>(pk, mb[U][h]);
                    const int w = w0 + 16 * h + win;
                    if (w < B) *reinterpret_cast<u32x4*>(dxo + act_idx(w, U, B) + col) = pk;
                    if (nkind != NK_DEAD)
                        *reinterpret_cast<u32x4*>(smem + U * BLKB + h * P::BLK + loff) = nkind == NK_RELU ? chunk_mask_bits<T>(pk, mb[U][h]) : pk;
                    if (nres) unpack_oct(pk, n0[h], n1[h]);
                }
            }
            if (l > 0) wd_acc_set_all<NH, U>(R, n0, n1);
        });
        __syncthreads();
        FS_STAMP(6 + 5 * (a.L - 1 - l));
    }
}

template <int NH, int NS, int NM, int DMAX> int launch_one(const StackArgs& a, bool bwd, int tiles, int lds, hipStream_t st, bool set_attr) {
    if (set_attr) {
        int rc;
        if ((rc = set_lds_attr(k_eng_fwd<NH, NS, NM, DMAX>, lds)) || (rc = set_lds_attr(k_eng_bwd<NH, NS, NM>, lds))) return rc;
        return MSHGNN_OK;
    }
    if (bwd) hipLaunchKernelGGL((k_eng_bwd<NH, NS, NM>), dim3(tiles), dim3(WD_THREADS), lds, st, a);
    else hipLaunchKernelGGL((k_eng_fwd<NH, NS, NM, DMAX>), dim3(tiles), dim3(WD_THREADS), lds, st, a);
    return MSHGNN_OK;
}
template <int NH, int NS, int NM> int launch_pair(const HostPlan& hp, const StackArgs& a, bool bwd, int tiles, int lds, hipStream_t st, bool set_attr) {
    return hp.d.out_channels <= 4 ? launch_one<NH, NS, NM, 4>(a, bwd, tiles, lds, st, set_attr) : launch_one<NH, NS, NM, 8>(a, bwd, tiles, lds, st, set_attr);
}
// The instantiations are spread over three translation units (compile time: the asm engines are 8-10 k lines each):
//   WD_PART 0: wide geometry, <= 18 nodes (+ the wide entry points)   1: wide geometry, 19-20 nodes   2: slab2 geometry (+ its entry points)
template <int NH, int NSLO, int NSHI> int dispatch(const HostPlan& hp, const StackArgs& a, bool bwd, int tiles, hipStream_t st, bool set_attr) {
    const int lds = hp.NN * WD_BLKB<NH> + (hp.NN <= WD_LDSBIAS_MAXN ? WD_BIAS_LDS : 0);
    const bool nm2 = hp.n_mlp <= 2;
#ifdef WD_ONLY18
    if constexpr (NSLO <= 18 && NSHI >= 18) return launch_one<NH, 18, 2, 4>(a, bwd, tiles, lds, st, set_attr);
    else return set_err(MSHGNN_EUNSUPPORTED, "experiment build: 18-node instantiation only");
#else
    if constexpr (NSLO <= 16) if (hp.NN <= 16) return nm2 ? launch_pair<NH, 16, 2>(hp, a, bwd, tiles, lds, st, set_attr) : launch_pair<NH, 16, 4>(hp, a, bwd, tiles, lds, st, set_attr);
    if constexpr (NSLO <= 18 && NSHI >= 18) if (hp.NN <= 18) return nm2 ? launch_pair<NH, 18, 2>(hp, a, bwd, tiles, lds, st, set_attr) : launch_pair<NH, 18, 4>(hp, a, bwd, tiles, lds, st, set_attr);
    if constexpr (NSHI >= 20) if (hp.NN <= 20) return nm2 ? launch_pair<NH, 20, 2>(hp, a, bwd, tiles, lds, st, set_attr) : launch_pair<NH, 20, 4>(hp, a, bwd, tiles, lds, st, set_attr);
    return set_err(MSHGNN_EUNSUPPORTED, "no engine-driven stack kernel for this many nodes per window");
#endif
}

}  // namespace

#if WD_PART == 0
int wide_dispatch_hi(const HostPlan& hp, const StackArgs& a, bool bwd, int tiles, hipStream_t st, bool set_attr);
static int wide_dispatch(const HostPlan& hp, const StackArgs& a, bool bwd, int tiles, hipStream_t st, bool set_attr) {
    return hp.NN <= 18 ? dispatch<2, 16, 18>(hp, a, bwd, tiles, st, set_attr) : wide_dispatch_hi(hp, a, bwd, tiles, st, set_attr);
}
int wide_set_attrs(const mshgnn_plan* p) { return wide_dispatch(p->hp, StackArgs{}, false, 0, nullptr, true); }
int wide_launch(const mshgnn_plan* p, const StackArgs& a, bool bwd, hipStream_t st) {
    return wide_dispatch(p->hp, a, bwd, (a.B + WD_ROWS - 1) / WD_ROWS, st, false);
}
#elif WD_PART == 1
int wide_dispatch_hi(const HostPlan& hp, const StackArgs& a, bool bwd, int tiles, hipStream_t st, bool set_attr) {
    return dispatch<2, 20, 20>(hp, a, bwd, tiles, st, set_attr);
}
#else
int slab2_set_attrs(const mshgnn_plan* p) { return dispatch<1, 16, 18>(p->hp, StackArgs{}, false, 0, nullptr, true); }
int slab2_launch(const mshgnn_plan* p, const StackArgs& a, bool bwd, hipStream_t st) {
    return dispatch<1, 16, 18>(p->hp, a, bwd, (a.B + TILE_ROWS - 1) / TILE_ROWS, st, false);
}
#endif
