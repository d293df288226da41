// Shared device helpers, argument structs and the plan object of libmshgnn (included by every .hip translation unit of the
// library: mshgnn.hip = fp32 / bf16 plans + C-ABI, mshgnn_x3.hip = split-bf16 parity plan, mshgnn_gen.hip = generic-width engine).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mshgnn_plan.hpp"

using namespace mshgnn;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// once-read streams (raw inputs): optional non-temporal loads, a build-time experiment switch (EXTRA=-DENC_NT=1 / -DGW_NT=1|2)
#ifndef ENC_NT
#define ENC_NT 0
#endif
#ifndef GW_NT
#define GW_NT 0
#endif
#ifndef ENC_ORDER
#define ENC_ORDER 0      // 1: encoder workgroups tile-major inside a type (all nodes of 64 windows run together)
#endif
template <int NT> __device__ __forceinline__ u32x4 ld16(const void* p) {
    if constexpr (NT != 0) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else return *reinterpret_cast<const u32x4*>(p);
}

// ------------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------------
inline thread_local std::string g_err;      // one per thread, shared by every translation unit of the library
inline int set_err(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return set_err(MSHGNN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)



// Timing ablations and in-kernel stamps exist only in instrumented builds (make EXTRA=-DMSHGNN_ABLATE, -DMSHGNN_FS_STAMPS, ...):
// in the product build every ABL() test is the constant false, so the branches fold away and no environment variable can
// change what the kernels compute.
#if defined(MSHGNN_GW_STAMPS) && MSHGNN_GW_STAMPS
#define MSHGNN_GW_STAMPS_BUILD 1
#else
#define MSHGNN_GW_STAMPS_BUILD 0
#endif
#ifdef MSHGNN_ABLATE
#define ABL(x) ((x) != 0)
#else
#define ABL(x) (false)
#endif
// ------------------------------------------------------------------------------------------------------
// precision traits
// ------------------------------------------------------------------------------------------------------
template <typename T> struct Prec;
// Both precisions use window tiles of 16 rows and 16-wide MFMA column blocks; a wave owns 2 column blocks (32 cols).
//   fp32: v_mfma_f32_16x16x4_f32   (lane group g = lane>>4 owns K range [32g, 32g+32): 8 chunks of 4 floats)
//   bf16: v_mfma_f32_16x16x32_bf16 (lane group g owns K range [32g, 32g+32): 4 chunks of 8 bf16)
template <> struct Prec<float> {
    static constexpr int ROWS = 16, EPC = 4, CPR = 32, RB = 512, NREG = 8, NBV = 16, NAV = 8, HS = 6, BLK = 8192, ENC_MB = 4, WPS = 2;   // WPS: waves/SIMD the layer kernels are built for
    using Vec = f32x4;
    struct Acc { f32x4 c[2]; };
    struct AFrag { f32x4 v[8]; };
    struct BFrag { f32x4 v[16]; };
};
template <> struct Prec<__bf16> {
    static constexpr int ROWS = 16, EPC = 8, CPR = 16, RB = 256, NREG = 8, NBV = 8, NAV = 4, HS = 4, BLK = 4096, ENC_MB = 4, WPS = 4;   // 2 workgroups of 8 waves per CU (80 KB LDS each)
    using Vec = bf16x8;
    struct Acc { f32x4 c[2]; };
    struct AFrag { bf16x8 v[4]; };
    struct BFrag { bf16x8 v[8]; };
};

// Make a lane-dependent value opaque so the compiler cannot hoist the address math derived from it out of the
// group loop (hoisted per-register epilogue addresses were being spilled to scratch -- guide, Appendix B pitfalls).
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
typedef const char __attribute__((address_space(1))) gchar;      // global address space, spelled out: a pointer rebuilt from integers is otherwise generic (flat requests, which count on BOTH vmcnt and lgkmcnt)
typedef char __attribute__((address_space(1))) gwchar;
__device__ __forceinline__ gchar* uniform_ptr(const char* p) {      // a global pointer the compiler must keep in SGPRs (request = SGPR base + 32-bit VGPR offset)
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<gchar*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu)));
}
__device__ __forceinline__ gwchar* uniform_wptr(char* p) { return const_cast<gwchar*>(uniform_ptr(p)); }
__device__ __forceinline__ u32x4 gload16(gchar* base, unsigned off) { return *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(base + off); }
__device__ __forceinline__ unsigned gload1(gchar* base, unsigned off) { return *reinterpret_cast<const uint8_t __attribute__((address_space(1)))*>(base + off); }
__device__ __forceinline__ void gstore16(gwchar* base, unsigned off, u32x4 v) { *reinterpret_cast<u32x4 __attribute__((address_space(1)))*>(base + off) = v; }
__device__ __forceinline__ void gstore16_nt(gwchar* base, unsigned off, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4 __attribute__((address_space(1)))*>(base + off)); }
__device__ __forceinline__ void gstore1(gwchar* base, unsigned off, unsigned v) { *reinterpret_cast<uint8_t __attribute__((address_space(1)))*>(base + off) = (uint8_t)v; }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// Accumulator layout.  The MFMAs are issued as  D^T = W_frag (A operand) x X_frag (B operand), so in the 16x16 C/D map
// (col = lane&15, row = 4*(lane>>4)+j -- cdna_hip_programming.md section 3) the COLUMN is the window and the ROWS are
// output features.  k_prep permutes the weight rows of the wave's two 16-row blocks so that MFMA row 4g+j of block fb is
// feature 8 g + 4 fb + j: lane (w = lane&15, g = lane>>4) then holds the 8 CONSECUTIVE features 8g..8g+7 of window w
// (c[0] = first four, c[1] = next four), the 4 lane groups of a window cover 32 contiguous features, and epilogue
// traffic is one 16-byte (bf16) / 32-byte (fp32) vector per lane and accumulator.
__device__ __forceinline__ int c_win(int lane) { return lane & 15; }
__device__ __forceinline__ int c_oct(int lane) { return (lane >> 4) << 3; }                            // within the wave's 32 columns
__device__ __forceinline__ int c_feat(int fb, int lane) { return c_oct(lane) + (fb << 2); }

template <typename A> __device__ __forceinline__ void acc_fill(A& a, float v) {
    a.c[0] = f32x4{v, v, v, v}; a.c[1] = f32x4{v, v, v, v};
}

// 4 consecutive elements of T <-> f32x4 (global or LDS; 16-byte aligned for fp32, 8-byte for bf16)
__device__ __forceinline__ f32x4 load_quad(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load_quad(const __bf16* p) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(p);
    return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
                 __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
}
__device__ __forceinline__ void store_quad(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store_quad(__bf16* p, f32x4 v) {
    union { u32x2 r; __bf16 e[4]; } u;
    u.e[0] = (__bf16)v[0]; u.e[1] = (__bf16)v[1]; u.e[2] = (__bf16)v[2]; u.e[3] = (__bf16)v[3];
    *reinterpret_cast<u32x2*>(p) = u.r;
}
// 8 consecutive elements of T <-> two f32x4 (global: 16-byte aligned for bf16, 32-byte for fp32)
__device__ __forceinline__ void load_oct(const float* p, f32x4& lo, f32x4& hi) {
    lo = reinterpret_cast<const f32x4*>(p)[0]; hi = reinterpret_cast<const f32x4*>(p)[1];
}
__device__ __forceinline__ void load_oct(const __bf16* p, f32x4& lo, f32x4& hi) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(p);
    lo = f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
               __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
    hi = f32x4{__builtin_bit_cast(float, r[2] << 16), __builtin_bit_cast(float, r[2] & 0xffff0000u),
               __builtin_bit_cast(float, r[3] << 16), __builtin_bit_cast(float, r[3] & 0xffff0000u)};
}
__device__ __forceinline__ void store_oct(float* p, f32x4 lo, f32x4 hi) {
    reinterpret_cast<f32x4*>(p)[0] = lo; reinterpret_cast<f32x4*>(p)[1] = hi;
}
__device__ __forceinline__ void store_oct(__bf16* p, f32x4 lo, f32x4 hi) {
    union { u32x4 r; __bf16 e[8]; } u;
#pragma unroll
    for (int j = 0; j < 4; ++j) { u.e[j] = (__bf16)lo[j]; u.e[4 + j] = (__bf16)hi[j]; }
    *reinterpret_cast<u32x4*>(p) = u.r;
}
__device__ __forceinline__ u32x4 pack_oct(f32x4 lo, f32x4 hi) {      // the 16 bytes store_oct(__bf16*) writes
    union { u32x4 r; __bf16 e[8]; } u;
#pragma unroll
    for (int j = 0; j < 4; ++j) { u.e[j] = (__bf16)lo[j]; u.e[4 + j] = (__bf16)hi[j]; }
    return u.r;
}
// sum over the 16 lanes of a DPP row, result in every lane: rotations by 8, 4, 2, 1 pair the same lanes as the xor butterfly
// (bit-identical sums) without the LDS round trips of ds_bpermute
template <int N> __device__ __forceinline__ float dpp_ror(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_ror<8>(x); x += dpp_ror<4>(x); x += dpp_ror<2>(x); x += dpp_ror<1>(x);
    return x;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
// value as it will read back after being stored as T (so LDS/global copies and the register copy agree)
template <typename T> __device__ __forceinline__ f32x4 round_as(f32x4 v) {
    if constexpr (sizeof(T) == 4) return v;
    else return f32x4{(float)(__bf16)v[0], (float)(__bf16)v[1], (float)(__bf16)v[2], (float)(__bf16)v[3]};
}

// Activation tensors (X_l, dX_l, dH_l, D_l, base_transform stash) are NODE-major in HBM: [node][window][128], so the
// streams of the encoder epilogue and of the weight-gradient kernel (one node, many windows) are contiguous and a layer
// tile reads one 16 x 128 block per node.
__device__ __forceinline__ size_t act_idx(int w, int node, int B) { return ((size_t)node * B + w) * H; }

// LDS node-block addressing: block = ROWS rows x 128 elements; 16-byte chunk c of row r lives at chunk slot
// c ^ swz(r), swz(r) = (r & 15) ^ ((r & 4) << 1).  The plain c ^ (r & 15) of guide T2 is 2-way on the 16x16x32 operand
// read (lane = row + 16 g reads chunk 4 g + t): ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, ..., and rows
// 0-3 of g collide with rows 4-7 of g + 1.  Folding row bit 2 into bit 3 makes those reads AND the 8-lane ds_write_b128
// groups of the octet stores conflict-free (exhaustive check over the GF(2)-linear maps: tools/lds_swizzle_search.py).
__device__ __forceinline__ int lds_swz(int row) { return (row & 15) ^ ((row & 4) << 1); }
template <typename T> __device__ __forceinline__ int lds_chunk(int blk, int row, int c) {
    return blk * Prec<T>::BLK + row * Prec<T>::RB + ((c ^ lds_swz(row)) << 4);
}
template <typename T> __device__ __forceinline__ int lds_elem(int blk, int row, int col) {
    return lds_chunk<T>(blk, row, col / Prec<T>::EPC) + (col % Prec<T>::EPC) * (int)sizeof(T);
}

// 8 consecutive features (col % 8 == 0) of one row of an LDS node block: one swizzled chunk (bf16) or two (fp32)
template <typename T> __device__ __forceinline__ void lds_load_oct(const char* smem, int blk, int row, int col, f32x4& lo, f32x4& hi) {
    if constexpr (sizeof(T) == 4) {
        lo = *reinterpret_cast<const f32x4*>(smem + lds_chunk<T>(blk, row, col / 4));
        hi = *reinterpret_cast<const f32x4*>(smem + lds_chunk<T>(blk, row, col / 4 + 1));
    } else load_oct(reinterpret_cast<const T*>(smem + lds_chunk<T>(blk, row, col / 8)), lo, hi);
}
template <typename T> __device__ __forceinline__ void lds_store_oct(char* smem, int blk, int row, int col, f32x4 lo, f32x4 hi) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(smem + lds_chunk<T>(blk, row, col / 4)) = lo;
        *reinterpret_cast<f32x4*>(smem + lds_chunk<T>(blk, row, col / 4 + 1)) = hi;
    } else store_oct(reinterpret_cast<T*>(smem + lds_chunk<T>(blk, row, col / 8)), lo, hi);
}

// A fragment: lane (row = lane % ROWS, group g = lane / ROWS) reads 8 consecutive chunks = its contiguous K range.
template <typename T> __device__ __forceinline__ void load_afrag(typename Prec<T>::AFrag& a, const char* smem, int blk, int lane_) {
    const int lane = opaque(lane_);
    const int row = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < Prec<T>::NAV; ++t) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(blk, row, g * Prec<T>::NAV + t));
        a.v[t] = __builtin_bit_cast(typename Prec<T>::Vec, v);
    }
}
// The same read with this lane's NAV chunk offsets computed once per kernel (stack kernels): per MAC only the block base
// is added, on address registers of their own, so each K-step's read can issue right behind the MFMAs that consumed it.
template <typename T> struct AOff {
    int o[Prec<T>::NAV];
    __device__ __forceinline__ explicit AOff(int lane) {
        const int row = lane & 15, g = lane >> 4;
#pragma unroll
        for (int t = 0; t < Prec<T>::NAV; ++t) o[t] = opaque(lds_chunk<T>(0, row, g * Prec<T>::NAV + t));
    }
};
template <typename T> __device__ __forceinline__ void load_afrag(typename Prec<T>::AFrag& a, const char* smem, int blk, const AOff<T>& ao) {
    const int base = blk * Prec<T>::BLK;
#pragma unroll
    for (int t = 0; t < Prec<T>::NAV; ++t) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + base + ao.o[t]);
        a.v[t] = __builtin_bit_cast(typename Prec<T>::Vec, v);
    }
}
// B fragment: packed by k_prep so that vector (wave, v, lane) is one contiguous 16-byte load.
template <typename T> __device__ __forceinline__ void load_bfrag(typename Prec<T>::BFrag& b, const T* wpack, int pack, int wv, int lane) {
    constexpr int NBV = Prec<T>::NBV;
    const u32x4* base = reinterpret_cast<const u32x4*>(wpack + (size_t)pack * H * H) + (size_t)wv * NBV * 64 + lane;
#pragma unroll
    for (int v = 0; v < NBV; ++v) {
        const u32x4 x = base[v * 64];
        b.v[v] = __builtin_bit_cast(typename Prec<T>::Vec, x);
    }
}

// the same fragment requested as SGPR base + this lane's 32-bit offset (16 lane): compile-time programs, whose pack ids are literals -- no 64-bit address
// arithmetic and no address register pair per fragment
template <typename T> __device__ __forceinline__ void load_bfrag_u(typename Prec<T>::BFrag& b, const T* wpack, int pack, int wv, unsigned lane16) {
    constexpr int NBV = Prec<T>::NBV;
    gchar* base = uniform_ptr(reinterpret_cast<const char*>(wpack) + ((size_t)pack * H * H * sizeof(T)) + (size_t)wv * NBV * 64 * 16);
#pragma unroll
    for (int v = 0; v < NBV; ++v) b.v[v] = __builtin_bit_cast(typename Prec<T>::Vec, gload16(base + v * 1024, lane16));
}

// acc^T += W_frag . X_frag  (A operand = packed weights, B operand = the window tile)
__device__ __forceinline__ void mac(Prec<float>::Acc& acc, const Prec<float>::AFrag& x, const Prec<float>::BFrag& w) {
#pragma unroll
    for (int t4 = 0; t4 < 8; ++t4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc.c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[t4][e], x.v[t4][e], acc.c[0], 0, 0, 0);
            acc.c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[8 + t4][e], x.v[t4][e], acc.c[1], 0, 0, 0);
        }
}
__device__ __forceinline__ void mac(Prec<__bf16>::Acc& acc, const Prec<__bf16>::AFrag& x, const Prec<__bf16>::BFrag& w) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc.c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v[t], x.v[t], acc.c[0], 0, 0, 0);
        acc.c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v[4 + t], x.v[t], acc.c[1], 0, 0, 0);
    }
}

// bias as the accumulator's initial value: this lane's 4 consecutive features of each 16-feature block
template <typename T> __device__ __forceinline__ void acc_init_bias(typename Prec<T>::Acc& a, const float* bias, int wv, int lane) {
    if (bias == nullptr) { acc_fill(a, 0.f); return; }
    a.c[0] = *reinterpret_cast<const f32x4*>(bias + wv * 32 + c_feat(0, lane));
    a.c[1] = *reinterpret_cast<const f32x4*>(bias + wv * 32 + c_feat(1, lane));
}
// (compile-time programs: scalar base of the row slice + this lane's byte offset)
template <typename T> __device__ __forceinline__ void acc_init_bias_u(typename Prec<T>::Acc& a, const float* bias_row, int wv, unsigned loff) {
    gchar* base = uniform_ptr(reinterpret_cast<const char*>(bias_row + wv * 32));
    a.c[0] = __builtin_bit_cast(f32x4, gload16(base, loff));
    a.c[1] = __builtin_bit_cast(f32x4, gload16(base + 16, loff));
}
// this lane's bias values, fetched at group start and ADDED in the epilogue so the load's latency hides under the MACs
struct BiasQ { f32x4 b[2]; };
__device__ __forceinline__ BiasQ load_bias(const float* bias, int wv, int lane) {
    BiasQ q;
    q.b[0] = *reinterpret_cast<const f32x4*>(bias + wv * 32 + c_feat(0, lane));
    q.b[1] = *reinterpret_cast<const f32x4*>(bias + wv * 32 + c_feat(1, lane));
    return q;
}

// layer kernels: 8 waves = 4 column slices (wn) x 2 slot halves (wh); wave (wn, wh) owns output columns
// [32 wn, 32 wn + 32) of the destination slots u with (u & 1) == wh  ->  HS = GMAX/2 accumulators per wave.
constexpr int LAYER_THREADS = 512;

// row-major work split of the 512 layer-kernel threads over node blocks: a block has VPB = ROWS*CPR 16-byte chunks
// (512 fp32 / 256 bf16), so NPB = 512/VPB blocks are covered per pass; thread -> (sub-block, row, chunk)
template <typename T> struct RowMap {
    static constexpr int VPB = Prec<T>::ROWS * Prec<T>::CPR, NPB = LAYER_THREADS / VPB;
    int sub, row, c;
    __device__ __forceinline__ explicit RowMap(int tid) : sub(tid / VPB), row((tid % VPB) / Prec<T>::CPR), c(tid % Prec<T>::CPR) {}
};

// stage node blocks [0, NN) of an activation tensor [B][NN][128] into LDS (zero rows beyond the batch)
template <typename T>
__device__ __forceinline__ void stage_nodes(char* smem, const T* src, int NN, int w0, int B, int tid) {
    constexpr int EPC = Prec<T>::EPC, NPB = RowMap<T>::NPB, BATCH = 6;
    const RowMap<T> m(tid);
    for (int nb = m.sub; nb < NN; nb += NPB * BATCH) {
        u32x4 v[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int n = nb + i * NPB;
            v[i] = u32x4{0, 0, 0, 0};
            if (n < NN && w0 + m.row < B) v[i] = *reinterpret_cast<const u32x4*>(src + act_idx(w0 + m.row, n, B) + m.c * EPC);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int n = nb + i * NPB;
            if (n < NN) *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, m.row, m.c)) = v[i];
        }
    }
}

// relu bits: one byte per (node, window, 8-feature group) -- exactly the 8 accumulator elements one lane of the layer / stack
// kernels owns, so the writer needs no cross-lane exchange: bytes [NN][4 column slices][ceil(B/16) tiles][4 groups][16 windows]
// (a wave's store is 64 contiguous bytes).  Bit j of the byte of (n, w, f0) <-> feature f0 + j, f0 a multiple of 8.
__device__ __forceinline__ size_t relu_byte(int n, int B, int w, int f) {
    return ((((size_t)n * 4 + (f >> 5)) * ((B + 15) >> 4) + (w >> 4)) << 6) + (((f >> 3) & 3) << 4) + (w & 15);
}
// relu of one accumulator in place + its 8 relu bits, on the integer pipe: for a float x (no NaNs), max_i32(bits(x), 0) is
// relu(x) (negative floats and -0 are negative integers), and y > 0 <=> y + 0x7fffffff has its top bit set; v_alignbit
// shifts that bit into the byte.  2 + 1 instructions per element, no compares / VCC hazards.
// base of the 64 relu bytes of (node n, column slice wn) in tile `tile`; the byte of lane (g, w & 15) is at + lane
__device__ __forceinline__ size_t relu_tile_base(int n, int B, int tile, int wn) { return (((size_t)n * 4 + wn) * ((B + 15) >> 4) + tile) << 6; }
__device__ __forceinline__ void unpack_oct(u32x4 r, f32x4& lo, f32x4& hi) {     // 8 bf16 -> two f32x4
    lo = f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
               __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
    hi = f32x4{__builtin_bit_cast(float, r[2] << 16), __builtin_bit_cast(float, r[2] & 0xffff0000u),
               __builtin_bit_cast(float, r[3] << 16), __builtin_bit_cast(float, r[3] & 0xffff0000u)};
}
template <typename T>
__device__ __forceinline__ unsigned relu_with_bits(typename Prec<T>::Acc& acc) {
    unsigned bits = 0;
#pragma unroll
    for (int fb = 1; fb >= 0; --fb)
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const float xf = acc.c[fb][j];      // (hipcc 7.2 miscompiles __builtin_bit_cast applied directly to a vector element)
            const int y = max(__float_as_int(xf), 0);
            acc.c[fb][j] = __int_as_float(y);
            bits = __builtin_amdgcn_alignbit(bits, (unsigned)y + 0x7fffffffu, 31);
        }
    return bits;
}
struct PrepArgs {
    const float* params; void* wpack; float* bias; const PackDesc* packs; const BiasDesc* biases; int n_packs; int n_biases;
    int pack0 = 0, pack_n = -1;      // per-thread kernel: the packs [pack0, pack0 + pack_n) of this launch (-1: all of them)
};

struct EncArgs {
    const void* x[MSHGNN_MAX_TYPES]; int64_t pitch[MSHGNN_MAX_TYPES]; int vb[MSHGNN_MAX_TYPES];
    int width[MSHGNN_MAX_TYPES], nodes[MSHGNN_MAX_TYPES], tbase[MSHGNN_MAX_TYPES + 1], nkc[MSHGNN_MAX_TYPES];      // nodes: of the launch (node_list); tbase[t + 1] - tbase[t]: of the type
    int pack0[MSHGNN_MAX_TYPES], bias_idx[MSHGNN_MAX_TYPES], sign_off[MSHGNN_MAX_TYPES], wg_prefix[MSHGNN_MAX_TYPES + 1];
    int n_types, tiles, B, NN;
    // the nodes this launch works on, per type: node_list[node_off[t] .. + nodes[t]) = indices inside the type (the plan's need_n[0]: nodes whose X_0 can
    // reach the output; every node when the launch also materialises window rows for the caller -- skip_mask then marks the nodes whose X_0 nobody
    // reads: bit (global node index) set = gather and write the window rows, no MACs, no X_0)
    unsigned char node_list[64]; int node_off[MSHGNN_MAX_TYPES]; unsigned long long skip_mask;
    int aligned;   // every input row starts 16-byte aligned and its pitch is a whole number of 16-byte chunks
    const void* wpack; const float* bias; const uint8_t* signs; void* x0;
    uint8_t* mask0;   // training: relu bytes of X_0 (one byte per lane, as the layer masks), read by the backward stack kernels at layer 0
    // bf16 plan, small plans: the LAYER weight packs are packed by extra workgroups behind the encoder's own (blockIdx >= wg_prefix[n_types]): they
    // run under the encoder's tail instead of in front of it (the encoder's own packs + the biases were packed by a short launch before)
    PrepArgs prep; int prep_vecs;      // prep_vecs: output vectors of those packs (0: none)
};

// The encoder's inputs gathered straight from a sequence's resident raw series (bf16 copies, column-major) instead of materialised windows --
// the fused window assembly of mshgnn_step_mse_series: feature k of node row (t, node) of window w is element starts[w] + k % T of the row's
// run k / T (runs of one length T, or the single constant-1 run; mshgnn_window_desc.fast_layout).
// Labels of a batch of windows (mshgnn_assemble_windows / mshgnn_step_*_series): y[b][k] = label series column label_cols[k] at the window's last
// step; label_rotate: the 3-D world-frame GRFs are taken into the body frame with the world->body quaternion of that step, R f per foot (the
// as_matrix() @ grfs_T branch of load_data_at_dataset_seq_3d); quat out = that quaternion (data.r_o, quadSDKDataset_Morph.py:365-367).
struct LabelArgs {
    const float* lab; int64_t lab_cs;            // the label series (column-major: element (row, c) at c * lab_cs + row)
    const float* quat_src; int64_t quat_cs;      // the quaternion series or nullptr
    const int64_t* starts; int64_t B; int T;
    const int* label_cols; int n_label, label_rotate;
    float* y; float* quat; int32_t* labels_int;  // labels_int (nullable): contact flags y != 0 for the fused cross entropy (mshgnn_step_ce_series)
};

__device__ __forceinline__ void window_labels_one(const LabelArgs& a, int64_t b) {
    const int64_t row = a.starts[b] + a.T - 1;
    const float* lab = a.lab + row;
    const int64_t lcs = a.lab_cs;
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    float qv[4] = {0.f, 0.f, 0.f, 1.f};
    if (a.quat_src) {
        const float* qp = a.quat_src + row;
        const int64_t qcs = a.quat_cs;
        const float q0 = qp[0], q1 = qp[qcs], q2 = qp[2 * qcs], q3 = qp[3 * qcs];
        qv[0] = q0; qv[1] = q1; qv[2] = q2; qv[3] = q3;
        if (a.label_rotate) {
            double x = q0, yq = q1, z = q2, s = q3;
            const double nrm = sqrt(x * x + yq * yq + z * z + s * s);
            x /= nrm; yq /= nrm; z /= nrm; s /= nrm;
            R[0][0] = 1 - 2 * (yq * yq + z * z); R[0][1] = 2 * (x * yq - z * s); R[0][2] = 2 * (x * z + yq * s);
            R[1][0] = 2 * (x * yq + z * s); R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (yq * z - x * s);
            R[2][0] = 2 * (x * z - yq * s); R[2][1] = 2 * (yq * z + x * s); R[2][2] = 1 - 2 * (x * x + yq * yq);
        }
    }
    // every label of the window is fetched before the first one is stored: the stores may alias the series as far as the compiler knows, and a
    // load -> store -> load chain costs one memory round trip per label (12 for the A1 GRFs: 18 us for 8192 windows, 13.6 with the loads batched)
    constexpr int LMAX = 24;
    for (int k0 = 0; k0 < a.n_label; k0 += LMAX) {
        float v[LMAX];
#pragma unroll
        for (int j = 0; j < LMAX; ++j) v[j] = k0 + j < a.n_label ? lab[a.label_cols[k0 + j] * lcs] : 0.f;
        if (a.label_rotate) {
#pragma unroll
            for (int j = 0; j + 2 < LMAX; j += 3) {
                if (k0 + j + 2 >= a.n_label) break;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    a.y[b * a.n_label + k0 + j + i] = (float)(R[i][0] * (double)v[j] + R[i][1] * (double)v[j + 1] + R[i][2] * (double)v[j + 2]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < LMAX; ++j) {
                if (k0 + j >= a.n_label) break;
                a.y[b * a.n_label + k0 + j] = v[j];
                if (a.labels_int) a.labels_int[b * a.n_label + k0 + j] = v[j] != 0.f;
            }
        }
    }
    if (a.quat_src && a.quat) { a.quat[b * 4] = qv[0]; a.quat[b * 4 + 1] = qv[1]; a.quat[b * 4 + 2] = qv[2]; a.quat[b * 4 + 3] = qv[3]; }
}

struct SeriesSrc {
    const unsigned long long* run_ptr;   // device: per run of the window descriptor, the address of its column's element 0 (0: the constant 1)
    const int* rows;                     // device: per node row, [first run, end run)
    const int64_t* starts;               // device: first series row of every window
    int row0[MSHGNN_MAX_TYPES];          // first node row of each type
    int T;
    LabelArgs lab;                       // the batch's labels: computed by extra workgroups of the fused-gather encoder launch (lab.B == 0: none)
};

// WIDE SOURCE ROWS (mshgnn_*_src entry points): the encoder reads the reference's own tensors -- fp64 (the reference's default dtype, gnnLightning.py:1183) or fp32,
// at their dense pitch -- converts in registers and writes the plan-dtype rows at the engine's pitch on the side (what the weight-gradient kernel reads
// later): the separate cast + re-pitch pass over the batch (472 MB read + 118 MB written + 118 MB re-read at A1-C2, 8192 windows) disappears.
// Loads are 8-byte units (one fp64 / two fp32 elements): a unit is either wholly inside its row or not requested at all (fp32: even widths), so nothing is
// ever read past the end of the caller's tensor; odd fp32 widths (the 1-wide foot rows) take guarded element loads.
struct WideSrc {
    int bytes;                               // element size of the source rows: 8 (fp64) or 4 (fp32); 0: none
    const void* p[MSHGNN_MAX_TYPES];         // per type: [B][n_t][pitch] source rows
    int64_t pitch[MSHGNN_MAX_TYPES];         // elements
};
inline thread_local const WideSrc* g_wide_src = nullptr;      // set by the _src entry points around the plain call they forward to (host side, same thread)
// One-call steps over at least twice MSHGNN_STEP_CHUNK windows (default 32 768) run as a sequence of equal sub-steps of at least that many windows over contiguous window ranges on the same workspace
// (mshgnn_step_mse / mshgnn_step_ce): every sub-step scales its loss terms by the WHOLE batch's element count and the finalize launches after the first add to the
// flat gradient and the loss instead of overwriting them.  Measured on Solo-12 K4 (BASELINE configs[3], 65 536 windows): the weight-gradient launch of one 65 536-window
// step costs 1.03x (bf16) / 1.21x (split plan) two 32 768-window ones, and the step's stash footprint halves.  Set by the entry point around its sub-steps (host side).
struct StepChunk { int64_t total_windows; int index; };
inline thread_local const StepChunk* g_step_chunk = nullptr;
inline int64_t loss_windows(int64_t B) { return g_step_chunk ? g_step_chunk->total_windows : B; }
inline int step_accumulates() { return g_step_chunk && g_step_chunk->index > 0 ? 1 : 0; }
// 8 source elements held as 8-byte units -> 8 fp32 values (two f32x4); units past `nvalid` elements were not requested: zero
template <int SB> __device__ __forceinline__ void wide_to_f32(const u32x2 (&u)[SB], int nvalid, f32x4& lo, f32x4& hi) {
    float f[8];
    if constexpr (SB == 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double d = __builtin_bit_cast(double, u[e]);
            float fe = (float)d;
            // the fp32 value must EXIST: torch rounds fp64 -> fp32 -> bf16 (two roundings, host and device alike); without this the compiler folds the two
            // conversions into one fp64 -> bf16 rounding, which differs on fp32 values that land exactly between two bf16 (seen: -0x1.22ffff259b41fp-1)
            asm volatile("" : "+v"(fe));
            f[e] = e < nvalid ? fe : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {      // (the whole unit is cast: hipcc 7.2 miscompiles __builtin_bit_cast applied to ONE element of a vector)
            const f32x2 v = __builtin_bit_cast(f32x2, u[i]);
            f[2 * i] = 2 * i < nvalid ? v[0] : 0.f;
            f[2 * i + 1] = 2 * i + 1 < nvalid ? v[1] : 0.f;
        }
    }
    lo = f32x4{f[0], f[1], f[2], f[3]}; hi = f32x4{f[4], f[5], f[6], f[7]};
}
// request this thread's 8-element chunk [k0, k0 + 8) of a source row (row = first byte of the row; F = row width).  unit_ok (uniform per type): every 8-byte
// unit is wholly valid or wholly invalid; invalid units re-read the row's first unit (never used).  Otherwise guarded element loads.
template <int SB> __device__ __forceinline__ void wide_fetch(u32x2 (&u)[SB], const char* row, int k0, int F, bool unit_ok, bool row_ok) {
    constexpr int EU = 8 / SB;      // elements per unit
    if (unit_ok) {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int e0 = k0 + i * EU;
            u[i] = *reinterpret_cast<const u32x2*>(row + (size_t)(e0 + EU <= F ? e0 : 0) * SB);
        }
    } else {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            u[i] = u32x2{0, 0};
#pragma unroll
            for (int j = 0; j < EU; ++j) {
                const int e = k0 + i * EU + j;
                if (row_ok && e < F) {
                    if constexpr (SB == 8) u[i] = *reinterpret_cast<const u32x2*>(row + (size_t)e * 8);
                    else u[i][j] = *reinterpret_cast<const unsigned*>(row + (size_t)e * 4);
                }
            }
        }
    }
}

// load up to EPC elements starting at p with the widest vector the alignment `vb` (bytes) allows
template <typename T> __device__ __forceinline__ u32x4 load_chunk(const T* p, int nvalid, int vb) {
    constexpr int EPC = Prec<T>::EPC;
    u32x4 r = u32x4{0, 0, 0, 0};
    if (nvalid <= 0) return r;
    if (nvalid >= EPC && vb >= 16) return *reinterpret_cast<const u32x4*>(p);
    if (nvalid >= EPC && vb == 8) {
        const u32x2 a = reinterpret_cast<const u32x2*>(p)[0], b = reinterpret_cast<const u32x2*>(p)[1];
        return u32x4{a[0], a[1], b[0], b[1]};
    }
    if (nvalid >= EPC && vb == 4) {
        const unsigned* q = reinterpret_cast<const unsigned*>(p);
        return u32x4{q[0], q[1], q[2], q[3]};
    }
    union { T e[EPC]; u32x4 v; } tmp;
#pragma unroll
    for (int e = 0; e < EPC; ++e) tmp.e[e] = e < nvalid ? p[e] : from_f32<T>(0.f);
    return tmp.v;
}
// zero every element of a 16-byte chunk beyond the first nv (nv >= EPC keeps all)
template <typename T> __device__ __forceinline__ u32x4 chunk_keep_first(u32x4 v, int nv) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e >= nv) v[e] = 0u;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned m = (2 * e < nv ? 0x0000ffffu : 0u) | (2 * e + 1 < nv ? 0xffff0000u : 0u);
            v[e] &= m;
        }
    }
    return v;
}
// XOR sign bits from EPC sign bytes (0/1) starting at s
template <typename T> __device__ __forceinline__ u32x4 sign_xor(const uint8_t* s) {
    if constexpr (sizeof(T) == 4) {
        const unsigned w = *reinterpret_cast<const unsigned*>(s);
        return u32x4{(w & 1u) << 31, ((w >> 8) & 1u) << 31, ((w >> 16) & 1u) << 31, ((w >> 24) & 1u) << 31};
    } else {
        const unsigned w0 = reinterpret_cast<const unsigned*>(s)[0], w1 = reinterpret_cast<const unsigned*>(s)[1];
        return u32x4{((w0 & 1u) << 15) | (((w0 >> 8) & 1u) << 31), (((w0 >> 16) & 1u) << 15) | (((w0 >> 24) & 1u) << 31),
                     ((w1 & 1u) << 15) | (((w1 >> 8) & 1u) << 31), (((w1 >> 16) & 1u) << 15) | (((w1 >> 24) & 1u) << 31)};
    }
}

// elementwise helpers on one 16-byte chunk of T
template <typename T> __device__ __forceinline__ u32x4 chunk_add(u32x4 a, u32x4 b) {
    union U { u32x4 v; T e[Prec<T>::EPC]; } x, y, r;
    x.v = a; y.v = b;
#pragma unroll
    for (int e = 0; e < Prec<T>::EPC; ++e) r.e[e] = from_f32<T>(to_f32(x.e[e]) + to_f32(y.e[e]));
    return r.v;
}
template <typename T> __device__ __forceinline__ u32x4 chunk_mask_pos(u32x4 v, u32x4 act) {   // keep v where act > 0
    union U { u32x4 v; T e[Prec<T>::EPC]; } x, y;
    x.v = v; y.v = act;
#pragma unroll
    for (int e = 0; e < Prec<T>::EPC; ++e) if (!(to_f32(y.e[e]) > 0.f)) x.e[e] = from_f32<T>(0.f);
    return x.v;
}
template <typename T> __device__ __forceinline__ u32x4 chunk_mask_bits(u32x4 v, unsigned bits) {   // EPC relu bits, LSB first
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!((bits >> e) & 1u)) v[e] = 0u;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {      // two sign-extending bit-field extracts + one bit-field insert per pair (the compare / select form costs 7 VALU per pair)
            const int lo = __builtin_amdgcn_sbfe((int)bits, 2 * e, 1), hi = __builtin_amdgcn_sbfe((int)bits, 2 * e + 1, 1);
            v[e] &= (unsigned)((lo & 0xffff) | (hi & (int)0xffff0000));
        }
    }
    return v;
}


// ------------------------------------------------------------------------------------------------------
// thread = (row, 8-column chunk): 16 lanes share a row, one wave covers 4 rows; whole rows are read / written coalesced
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __builtin_bit_cast(float, r[e] << 16); v[2 * e + 1] = __builtin_bit_cast(float, r[e] & 0xffff0000u); }
    }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<f32x4*>(p)[0] = f32x4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4*>(p)[1] = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        union { u32x4 r; __bf16 e[8]; } u;
#pragma unroll
        for (int e = 0; e < 8; ++e) u.e[e] = (__bf16)v[e];
        *reinterpret_cast<u32x4*>(p) = u.r;
    }
}


// split plan (MSHGNN_BF16X3): an fp32 value travels as two bf16 values, hi = bf16(x) and lo = bf16(x - hi): x = hi + lo to 16 mantissa
// bits (2^-17 relative), and a product is taken as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation
__device__ __forceinline__ void split_oct(f32x4 a, f32x4 b, u32x4& hi, u32x4& lo) {
    hi = pack_oct(a, b);
    f32x4 ha, hb; unpack_oct(hi, ha, hb);
    lo = pack_oct(a - ha, b - hb);
}
__device__ __forceinline__ void join_oct(u32x4 hi, u32x4 lo, f32x4& a, f32x4& b) {
    f32x4 la, lb; unpack_oct(hi, a, b); unpack_oct(lo, la, lb);
    a += la; b += lb;
}

// k_prep, tiled: one workgroup = one HALF (64 output columns) of one weight pack.  The source matrices are read row by row with 16-byte loads
// (root-sum of up to 8 matrices in registers, in source order), the summed tile goes through LDS, and every thread then assembles its MFMA
// B-fragment vectors from LDS -- the first version gathered every element with a scalar load from global memory (64 scattered 32-byte sectors per
// wave instruction).  Same sums in the same order: identical images.  It wins where there are many packs (>= PREP_TILED_MIN); with ~100 packs the
// per-thread kernel has more workgroups in flight and one memory round trip fewer.
//   orient 0 (forward images, B[k][c] = W[c][col0 + k], zero for k >= ncols): tile[a = c - 64 h][b = k], 64 x 128
//   orient 1 (backward images, B[k][c] = W[k][c]):                            tile[a = k][b = c - 64 h], 128 x 64
constexpr int PREP_PITCH0 = 132, PREP_PITCH1 = 68, PREP_TILE_FLOATS = 64 * PREP_PITCH0 > 128 * PREP_PITCH1 ? 64 * PREP_PITCH0 : 128 * PREP_PITCH1;
template <typename T, bool SPLIT_OUT>      // T: element type of the images (float / __bf16); SPLIT_OUT: hi image at wpack, lo image n_packs images further (split plan)
__device__ __forceinline__ void prep_pack_half(const PrepArgs& a, int pack, int h, float* tile, int tid) {
    constexpr int EPC = Prec<T>::EPC, NBV = Prec<T>::NBV;
    // (the descriptor's scalars are copied; its src[] array is read in place -- a local copy indexed by the loop counter lives in scratch: 96 B per lane)
    const PackDesc* pdp = a.packs + pack;
    struct { int orient, n_src, ld, col0, ncols; } pd{pdp->orient, pdp->n_src, pdp->ld, pdp->col0, pdp->ncols};
    const bool o0 = pd.orient == 0;
    const int pitch = o0 ? PREP_PITCH0 : PREP_PITCH1;
    // ---- summed source tile -> LDS
    {
        const int nb4 = o0 ? 32 : 16;                        // float4 columns of a tile row
        const int b4 = (tid % nb4) * 4, a0 = tid / nb4, apass = 256 / nb4, npass = (o0 ? 64 : 128) / apass;      // 8 passes either way
        f32x4 sum[8];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) sum[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < pd.n_src; ++i) {
            const int64_t src_i = pdp->src[i];
            const float* base = a.params + src_i;
            const bool al = ((src_i + (o0 ? pd.col0 : 0)) % 4 == 0) && pd.ld % 4 == 0;
            f32x4 g[8];
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int ar = a0 + ps * apass;
                const float* q = o0 ? base + (int64_t)(64 * h + ar) * pd.ld + pd.col0 + b4 : base + (int64_t)ar * pd.ld + 64 * h + b4;
                const int nvalid = o0 ? pd.ncols - b4 : 4;
                g[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ps < npass) {
                    if (al && nvalid >= 4) g[ps] = *reinterpret_cast<const f32x4*>(q);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (e < nvalid) g[ps][e] = q[e];
                    }
                }
            }
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) sum[ps] += g[ps];      // source order: the same value every step
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) if (ps < npass) *reinterpret_cast<f32x4*>(&tile[(a0 + ps * apass) * pitch + b4]) = sum[ps];
    }
    __syncthreads();
    // ---- fragment vectors of the two 32-column wave slices of this half
    constexpr int VPH = 2 * NBV * 64;          // vectors per half
    const size_t vec0 = (size_t)pack * (H * H / EPC) + (size_t)(2 * h) * NBV * 64;
    for (int r = tid; r < VPH; r += 256) {
        const int lane = r % 64, v = (r / 64) % NBV, wl = r / (64 * NBV);      // wl: wave slice inside the half
        const int i16 = lane & 15;
        int k, cl;                               // cl: column inside the half
        if constexpr (sizeof(T) == 4) { k = 32 * (lane >> 4) + 4 * (v & 7); cl = wl * 32 + 8 * (i16 >> 2) + 4 * (v >> 3) + (i16 & 3); }
        else { k = 32 * (lane >> 4) + 8 * (v & 3); cl = wl * 32 + 8 * (i16 >> 2) + 4 * (v >> 2) + (i16 & 3); }
        float s[EPC];
#pragma unroll
        for (int x = 0; x < EPC; ++x) s[x] = o0 ? tile[cl * PREP_PITCH0 + k + x] : tile[(k + x) * PREP_PITCH1 + cl];
        if constexpr (SPLIT_OUT) {
            u32x4 hi, lo;
            split_oct(f32x4{s[0], s[1], s[2], s[3]}, f32x4{s[4], s[5], s[6], s[7]}, hi, lo);
            u32x4* dst = reinterpret_cast<u32x4*>(a.wpack);
            dst[vec0 + r] = hi;
            dst[(size_t)a.n_packs * (H * H / EPC) + vec0 + r] = lo;
        } else {
            T* dst = reinterpret_cast<T*>(a.wpack) + (vec0 + r) * EPC;
            if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{s[0], s[1], s[2], s[3]};
            else *reinterpret_cast<u32x4*>(dst) = pack_oct(f32x4{s[0], s[1], s[2], s[3]}, f32x4{s[4], s[5], s[6], s[7]});
        }
    }
}
template <typename T, bool SPLIT_OUT> __global__ __launch_bounds__(256) void k_prep_tiled(PrepArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[PREP_TILE_FLOATS];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < 2 * a.n_packs) { prep_pack_half<T, SPLIT_OUT>(a, blockIdx.x >> 1, blockIdx.x & 1, tile, tid); return; }
    const int b = ((int)blockIdx.x - 2 * a.n_packs) * 256 + tid;
    if (b < a.n_biases * H) {
        const BiasDesc* bd = a.biases + b / H;
        const int ns = bd->n_src;
        float s = 0.f;
        for (int i = 0; i < ns; ++i) s += a.params[bd->src[i] + (b % H)];
        a.bias[b] = s;
    }
}
constexpr int PREP_TILED_MIN = 200;       // packs from which the tiled kernel wins
inline bool prep_use_tiled(int n_packs) {      // MSHGNN_PREP_TILED=0 / 1 forces either kernel (A/B runs)
    static const int forced = [] { const char* e = TUNE_ENV("MSHGNN_PREP_TILED"); return e ? atoi(e) : -1; }();
    return forced >= 0 ? forced != 0 : n_packs >= PREP_TILED_MIN;
}
inline unsigned prep_tiled_grid(int n_packs, int n_biases) { return (unsigned)(2 * n_packs + (n_biases * H + 255) / 256); }

struct StackArgs {
    static constexpr bool full = false;           // (StackView: every tile of the launch is complete)
    static constexpr int nt_mode = -1;            // (StackView: the stash store policy is a template parameter; -1: read stash_nt)
    const void* tile_in;                          // fwd: X_0                       bwd: dX_L
    char* ws;
    size_t x_off[MAX_L + 1], dx_off[MAX_L + 1];   // X_l stashes (fwd: written for l >= 1; bwd: X_0 read for the encoder mask), dX_l (bwd: written)
    size_t mask_off[MAX_L], hb_off[MAX_L], t1_off[MAX_L], dh_off[MAX_L], du_off[MAX_L];
    const void* wpack; const float* bias; const int* tables; int prog_off[MAX_L];
    int prog_off_b[MAX_L];                        // k_slab_step (forward + backward sweep of a tile in one launch): the backward layers' programs (prog_off: the forward's)
    int B, NN, L, training, dbg;
    const float* params; const float* out_mask; float* out; int64_t off_dec_w, off_dec_b; int node0, n_out, dout;   // fused decoder (fwd)
    // mshgnn_step_mse: the forward also takes the wrapper MSE and the decoder backward (dX_L rows, decoder partial gradients, loss partial)
    const float* y; float* dec_slabs; float inv_n;
    const int32_t* labels;   // mshgnn_step_ce: the classification wrappers' cross entropy instead of the MSE (labels [B][n_out] in {0, 1}, two logits per foot)
    size_t mask0_off;    // bwd: relu bytes of the encoder activation X_0 in the workspace (0: not available, X_0 rows are read)
    long long* stamps;   // timing experiments (MSHGNN_STAMPS): wave 0 of every workgroup records clock64() at phase boundaries
    // split plan: LDS block of the lo half of node n = lo_blk + n; the lo image of pack i is pack n_img + i
    int lo_blk, n_img;
    int scr0;            // split plan: first base_transform scratch block (NN, or NN - n_mlp when the scratch aliases the last nodes' blocks)
    int red_off;         // byte offset in LDS of the decoder tail's reduction scratch (0: the first blocks; one-launch steps whose out-type nodes come first: the blocks behind them)
    int stash_nt;        // stash rows (X_l, dX_l) are stored non-temporally: for stashes far beyond the Infinity Cache the weight-gradient pass that reads them next
                         // runs 11 % faster when the stack launch's writes do not allocate on their way out (stash_nt_for); a uniform choice per launch, same bits
    int stagger;         // two workgroups per CU: the second half of the grid starts this many cycles late, so that one workgroup's MAC phases
                         // (matrix pipe) run beside the other's epilogues (stores) instead of both competing for the same unit (0: off)
};
// The arguments as the kernel of a compile-time program sees them: the same member names (references into the kernel-argument segment), except the scalars
// that every node's stores read -- batch size, store policy, training flag -- which are COPIES pinned in SGPRs.  hipcc otherwise re-loads those from the
// kernel arguments wherever it runs short of scalar registers: a scalar-memory round trip (s_load + lgkmcnt(0)) per node in the store phases.
// full: the host launches these kernels only over batches of whole 16-window tiles (B % 16 == 0), so no store is predicated and the number of stores between a
// request and its use is the same on every path -- what lets the compiler wait for a load issued BEFORE a store phase with a counted vmcnt instead of a drain.
// NT: the stash store policy (StackArgs.stash_nt) as a template parameter, for the same reason (one arm, compiler-visible stores).
// FULL = false: the same view with every tile-edge predicate kept (rows past the batch are neither loaded nor stored, as in the interpreters): the kernels over compile-time
// programs that batches of other sizes take.  Measured (round 6): free at 3 layers (58 us either way), +22-27 % on the 8-layer programs (registers) -- hence both forms.
template <int NT, bool FULL = true> struct StackView {
    static constexpr bool full = FULL;
    static constexpr int nt_mode = NT;
    const StackArgs& s;
    const void* const& tile_in; char* const& ws;
    const size_t (&x_off)[MAX_L + 1]; const size_t (&dx_off)[MAX_L + 1];
    const size_t (&mask_off)[MAX_L]; const size_t (&hb_off)[MAX_L]; const size_t (&t1_off)[MAX_L]; const size_t (&dh_off)[MAX_L]; const size_t (&du_off)[MAX_L];
    const void* const& wpack; const float* const& bias; const int* const& tables; const int (&prog_off)[MAX_L]; const int (&prog_off_b)[MAX_L];
    const int& NN; const int& L; const int& dbg; const int& node0; const int& n_out; const size_t& mask0_off; long long* const& stamps; const int& stagger;
    int B, training, stash_nt;
    __device__ __forceinline__ StackView(const StackArgs& a, bool step)
        : s(a), tile_in(a.tile_in), ws(a.ws), x_off(a.x_off), dx_off(a.dx_off), mask_off(a.mask_off), hb_off(a.hb_off), t1_off(a.t1_off), dh_off(a.dh_off), du_off(a.du_off),
          wpack(a.wpack), bias(a.bias), tables(a.tables), prog_off(a.prog_off), prog_off_b(a.prog_off_b), NN(a.NN), L(a.L), dbg(a.dbg), node0(a.node0), n_out(a.n_out),
          mask0_off(a.mask0_off), stamps(a.stamps), stagger(a.stagger), B(a.B), training(step ? 1 : a.training), stash_nt(a.stash_nt) {
        asm volatile("" : "+s"(B));
    }
};
__device__ __forceinline__ const StackArgs& args_of(const StackArgs& a) { return a; }
template <int NT, bool FULL> __device__ __forceinline__ const StackArgs& args_of(const StackView<NT, FULL>& v) { return v.s; }

// one 16-byte stash store, plain or non-temporal (StackArgs.stash_nt: uniform)
__device__ __forceinline__ void stash_store(void* p, u32x4 v, bool nt) {
    // (two arms that differ only in the cache hint are merged by the optimiser into ONE plain store, and an opaque copy of the address turns it into a flat
    //  store: the non-temporal arm is therefore the instruction itself.  Nothing in a stack kernel loads what it stashed, and every MAC phase starts with an
    //  explicit vmcnt(0), so a store the compiler's wait counts do not know about is only ever waited for too long, never too short.)
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(p) = v;
}
// Stash rows one step writes for the weight-gradient pass: X_{l+1} of every node computed in layer l and dX_l of every node whose X_l was needed (plan liveness).  When they add
// up to more than MSHGNN_STASH_NT_MB (default 200, counted at 256 bytes per row on both plans: what the step launch pays grows with the number of stores, and the split plan's
// 512-byte rows at 3 layers -- 250 MB -- lose where the bf16 plan's 5 layers -- 230 MB -- win) the stores go out non-temporally -- the weight-gradient launch that reads them next then runs ~10 % faster
// (its shared rows keep meeting in L2 instead of being evicted by write-allocated stash lines) at no cost to the step launch.  Measured, 8192 windows (plain -> nt, ms/step):
// A1-C2 bf16 at 8 layers (640 MB) 0.6385 -> 0.6112, at 5 layers (~300 MB) 0.3738 -> 0.3649, MiniCheetah-K4 L = 8 0.7215 -> 0.6849, Solo K4 COM 5.16 -> 5.07, split plan at 8 layers
// 1.502 -> 1.458; below the limit the step launch pays for it instead (A1-C2 at 3 layers, 125 MB: step launch +3-5 us, weight gradients -3: -0.5 %; split plan, 250 MB: +1 %):
// plain stores stay there.  MSHGNN_STASH_NT=0 / 1 forces either; the bits are the same.
// the stash store of a kernel over argument type A (StackArgs: policy read at run time; StackView<NT>: compile-time, the non-temporal form through the builtin --
// with one arm only the optimiser has nothing to merge, and the store stays visible to the compiler's wait counting)
template <class A> __device__ __forceinline__ void stash_store_a(const A& a, void* p, u32x4 v) {
    if constexpr (A::nt_mode == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else if constexpr (A::nt_mode == 0) *reinterpret_cast<u32x4*>(p) = v;
    else stash_store(p, v, a.stash_nt != 0);
}
template <class A> __device__ __forceinline__ void stash_store_u(const A& a, gwchar* base, unsigned off, u32x4 v) {      // uniform base + lane offset (compile-time programs)
    static_assert(A::nt_mode >= 0, "compile-time store policy");
    if constexpr (A::nt_mode == 1) gstore16_nt(base, off, v); else gstore16(base, off, v);
}
inline int stash_nt_for(int64_t B, int stash_rows, int row_bytes) {
    static const int force = []() { const char* e = getenv("MSHGNN_STASH_NT"); return e ? atoi(e) : -1; }();
    static const int64_t limit_mb = []() { const char* e = TUNE_ENV("MSHGNN_STASH_NT_MB"); return (int64_t)(e ? atoi(e) : 200); }();
    if (force == 0 || force == 1) return force;
    return (int64_t)stash_rows * B * row_bytes > (limit_mb << 20) ? 1 : 0;
}
template <typename HP> inline int stash_rows_of(const HP& hp) {
    int rows = 0;
    for (int l = 0; l < hp.L; ++l)
        for (int n = 0; n < hp.NN; ++n) rows += (hp.live_n[l][n] && l + 1 < hp.L ? 1 : 0) + (hp.need_n[l][n] ? 1 : 0);
    return rows;
}
// start-up delay of the workgroups that share a CU with an earlier one (the first gridDim.x / 2 workgroups fill one slot per CU)
template <class A> __device__ __forceinline__ void stack_stagger(const A& a) {
    if (a.stagger > 0 && blockIdx.x >= (gridDim.x + 1) / 2)
        for (int i = 0; i < a.stagger; i += 1024) __builtin_amdgcn_s_sleep(16);      // (s_sleep n = 64 n cycles)
}
#ifdef MSHGNN_SEG_STAMPS
constexpr int FS_EXTRA_BLK = 6;     // LDS room for the per-segment clocks
#else
constexpr int FS_EXTRA_BLK = 0;
#endif
#if defined(MSHGNN_FS_STAMPS) || defined(MSHGNN_SEG_STAMPS)
#define FS_STAMP(k) do { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + (k)] = clock64(); } while (0)
#define FS_STAMP2(k) do { if (a.stamps && tid == 0) a.stamps[(size_t)gridDim.x * 32 + (size_t)blockIdx.x * 32 + (k)] = clock64(); } while (0)      // backward sweep of a one-launch step
#else
#define FS_STAMP(k) do { } while (0)     // the stamp stores are compiled out of the product build (they cost waits at phase boundaries)
#define FS_STAMP2(k) do { } while (0)
#endif

// Issue-sensitivity experiment (instrumented builds only: -DMSHGNN_PAD_VALU=n / -DMSHGNN_PAD_SALU=n): n extra independent vector instructions per node epilogue /
// n extra scalar instructions per MAC of the stack kernels.  If the launch were bound by instruction issue its time would follow the added count.
#ifdef MSHGNN_PAD_VALU
__device__ __forceinline__ void pad_valu() {
    int d0 = 0, d1 = 0;
#pragma unroll
    for (int i = 0; i < MSHGNN_PAD_VALU; i += 2) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(d0)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(d1)); }
}
#else
__device__ __forceinline__ void pad_valu() {}
#endif
#ifdef MSHGNN_PAD_SALU
__device__ __forceinline__ void pad_salu() {
    int d0 = 0;
#pragma unroll
    for (int i = 0; i < MSHGNN_PAD_SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(d0) :: "scc");
}
#else
__device__ __forceinline__ void pad_salu() {}
#endif

// wave program in three VGPRs, fetched with v_readlane: pk = pack id of segment `lane`; pcnt = its MAC counts (3 bits per
// accumulator); pb = 256 byte entries, 4 per lane (entry 0 = number of segments, then the block stream)
struct FProg {
    static constexpr bool is_static = false;
    int pk, pcnt, pb;
    __device__ __forceinline__ FProg() : pk(0), pcnt(0), pb(0) {}
    __device__ __forceinline__ FProg(const int* prog, int lane) : pk(prog[lane]), pcnt(prog[64 + lane]), pb(prog[128 + lane]) {}
    __device__ __forceinline__ int pack(int sgi) const { return __builtin_amdgcn_readlane(pk, sgi); }
    __device__ __forceinline__ int counts(int sgi) const { return __builtin_amdgcn_readlane(pcnt, sgi); }
    __device__ __forceinline__ int at(int i) const { return (__builtin_amdgcn_readlane(pb, i >> 2) >> ((i & 3) << 3)) & 0xff; }
    // the loads have landed from here on (the compiler waits for them HERE, with a counted vmcnt, and no longer tracks them as pending: a wait at the
    // first use -- the top of the next layer, behind this layer's stores -- would be a full drain)
    __device__ __forceinline__ void settle() { asm volatile("" : "+v"(pk), "+v"(pcnt), "+v"(pb)); }
};

// a layer header (FH_SIZE = 88 ints) held in two VGPRs and read with v_readlane: per-node flags cost no scalar-memory round
// trip (measured with in-kernel stamps: ~40 dependent s_loads of the header were 6.6k cycles before the first MAC of a layer)
struct FHdr {
    static constexpr bool is_static = false;
    int h0, h1;
    __device__ __forceinline__ FHdr() : h0(0), h1(0) {}
    __device__ __forceinline__ FHdr(const int* hdr, int lane) : h0(hdr[lane]), h1(lane < FH_SIZE - 64 ? hdr[64 + lane] : 0) {}
    __device__ __forceinline__ int operator[](int i) const { return i < 64 ? __builtin_amdgcn_readlane(h0, i & 63) : __builtin_amdgcn_readlane(h1, i & 63); }
    __device__ __forceinline__ void settle() { asm volatile("" : "+v"(h0), "+v"(h1)); }      // see FProg::settle
};

// The same header / wave program as COMPILE-TIME constants (specialised step kernels, mshgnn_spec_tables.inc): SP::fwd[l] / SP::bwd[l] are the ints the plan
// compiler emits for layer l of one (topology, depth) -- FH_SIZE header ints, then group A's and group B's wave programs.  Every accessor folds to a literal
// once the slot / segment loops are unrolled, so the slot walk (readlanes, count decoding, skipped slot headers, dead-node branches) leaves no instructions
// behind and LDS block addresses become immediate offsets.  DIR: 0 forward, 1 backward tables.
template <class SP, int DIR, int LL> struct SHdr {
    static constexpr bool is_static = true;
    __device__ __forceinline__ constexpr int operator[](int i) const { return DIR ? SP::bwd[LL][i] : SP::fwd[LL][i]; }
    __device__ __forceinline__ void settle() const {}
};
template <class SP, int DIR, int LL, int G> struct SProg {
    static constexpr bool is_static = true;
    static constexpr int BASE = FH_SIZE + G * FPROG_LEN;
    __device__ __forceinline__ constexpr int raw(int i) const { return DIR ? SP::bwd[LL][BASE + i] : SP::fwd[LL][BASE + i]; }
    __device__ __forceinline__ constexpr int pack(int sgi) const { return raw(sgi); }
    __device__ __forceinline__ constexpr int counts(int sgi) const { return raw(64 + sgi); }
    __device__ __forceinline__ constexpr int at(int i) const { return (raw(128 + (i >> 2)) >> ((i & 3) << 3)) & 0xff; }
    __device__ __forceinline__ void settle() const {}
};

// one segment: walk the accumulators in static order, each with its run-time MAC count; the source blocks come from the
// program's block stream in execution order, so the fragment of the NEXT MAC is read from LDS under this MAC's MFMAs
template <typename T, int HS = FS_HS, int CB = 3, class FP = FProg>      // HS accumulators, CB bits of MAC count per accumulator
__device__ __forceinline__ void fs_walk(const FP& wp, int sgi, int& pb, typename Prec<T>::AFrag& afn, typename Prec<T>::Acc (&acc)[HS],
                                        const typename Prec<T>::BFrag& bf, const char* smem, const AOff<T>& lane, int dbg = 0) {
    const int cw = wp.counts(sgi);      // one readlane per segment: 3 bits of MAC count per accumulator
#pragma unroll
    for (int u = 0; u < HS; ++u) {
        const int cnt = (cw >> (CB * u)) & ((1 << CB) - 1);
        for (int k = 0; k < cnt; ++k) {
            // the MFMAs of this MAC read afn as they issue; the fragment of the NEXT MAC is then read from LDS into the same
            // registers and lands while those MFMAs execute (no second buffer, no register copies)
#ifdef MSHGNN_ABLATE
            if (!ABL(dbg & 128)) mac(acc[u], afn, bf);
            if (!ABL(dbg & 256)) load_afrag<T>(afn, smem, wp.at(++pb), lane);
#else
            mac(acc[u], afn, bf);
            load_afrag<T>(afn, smem, wp.at(++pb), lane);
#endif
            pad_salu();
            if constexpr (sizeof(T) == 4) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * Prec<T>::NAV, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, Prec<T>::NAV, 0);
            } else {
                // K-step t's two MFMAs, then the read that refills x.v[t]: every read gets the rest of the MAC as cover
#pragma unroll
                for (int t = 0; t < Prec<T>::NAV; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
    }
}
// compile-time iteration: f(integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// position in the block stream of the first MAC of (segment sgi, slot u) of a compile-time program
template <class FP, int HS, int CB> constexpr int fs_static_pb(int sgi, int u) {
    constexpr FP cp{};
    int pb = 1;
    for (int sg = 0; sg <= sgi; ++sg)
        for (int v = 0; v < HS; ++v) {
            if (sg == sgi && v == u) return pb;
            pb += (cp.counts(sg) >> (CB * v)) & ((1 << CB) - 1);
        }
    return pb;
}
// fs_run over a compile-time program (SProg): the same MACs in the same order on the same accumulators (identical bits), as straight-line code -- no slot
// headers, no count decoding, no block-stream readlanes, LDS blocks as immediate offsets, exact counted waits (no branch between a request and its use).
// the first two weight fragments of a compile-time program's run, requested ahead of it (before the store phase that precedes the run: fs_run_static<PRE>)
template <typename T, class FP> __device__ __forceinline__ void fs_prefetch_static(typename Prec<T>::BFrag (&bf)[2], const T* wpack, int wn, int lane) {
    constexpr FP cp{};
    constexpr int nseg = cp.at(0);
    const unsigned lane16 = (unsigned)opaque(lane) * 16u;
    if constexpr (nseg > 0) load_bfrag_u<T>(bf[0], wpack, cp.pack(0), wn, lane16);
    if constexpr (nseg > 1) load_bfrag_u<T>(bf[1], wpack, cp.pack(1), wn, lane16);
}
// PRE: bf[0] / bf[1] were requested by fs_prefetch_static<FP> before the preceding store phase; the compiler waits for them with a counted vmcnt that leaves
// those stores in flight (same store count on every path: StackView), so the run starts under the drain instead of behind it
template <typename T, int HS, int CB, class FP, bool PRE = false>
__device__ __forceinline__ void fs_run_static(typename Prec<T>::Acc (&acc)[HS], const char* smem, const T* wpack, int wn, int lane, typename Prec<T>::BFrag (&bf)[2]) {
    constexpr FP cp{};
    constexpr int nseg = cp.at(0);
    if constexpr (nseg > 0) {
        constexpr int total = fs_static_pb<FP, HS, CB>(nseg - 1, HS) - 1;      // MACs of the run
        typename Prec<T>::AFrag afn;
        const AOff<T> ao(lane);
        const unsigned lane16 = (unsigned)opaque(lane) * 16u;
        if constexpr (!PRE) {
            __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));      // (as fs_run: the previous epilogue's stores are drained before the first request)
            load_bfrag_u<T>(bf[0], wpack, cp.pack(0), wn, lane16);
        }
        load_afrag<T>(afn, smem, cp.at(1), ao);
        static_for<0, nseg>([&](auto SG) {
            constexpr int sgi = decltype(SG)::value;
            if constexpr (sgi + 1 < nseg && !(PRE && sgi == 0)) load_bfrag_u<T>(bf[(sgi + 1) & 1], wpack, cp.pack(sgi + 1), wn, lane16);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, HS>([&](auto U) {
                constexpr int u = decltype(U)::value;
                constexpr int cnt = (cp.counts(sgi) >> (CB * u)) & ((1 << CB) - 1);
                constexpr int pb0 = fs_static_pb<FP, HS, CB>(sgi, u);
                static_for<0, cnt>([&](auto K) {
                    constexpr int pb = pb0 + decltype(K)::value;
                    mac(acc[u], afn, bf[sgi & 1]);
                    if constexpr (pb < total) load_afrag<T>(afn, smem, cp.at(pb + 1), ao);
#pragma unroll
                    for (int t = 0; t < Prec<T>::NAV; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        if (pb < total) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// all segments of a layer.  The next segment's weight fragment streams from L2 while the current one is multiplied
// (two register buffers).
template <typename T, int HS = FS_HS, int CB = 3, class FP = FProg>
__device__ __forceinline__ void fs_run(const FP& wp, typename Prec<T>::Acc (&acc)[HS], const char* smem, const T* wpack, int wn, int lane, int dbg = 0,
                                       long long* segclk = nullptr) {
    if constexpr (FP::is_static) { typename Prec<T>::BFrag bf[2]; fs_run_static<T, HS, CB, FP>(acc, smem, wpack, wn, lane, bf); return; }
    const int nseg = wp.at(0);
    int pb = 1;
    typename Prec<T>::BFrag bfa, bfb;
    typename Prec<T>::AFrag afn;
    const AOff<T> ao(lane);
    // drain the previous epilogue's memory operations first: with loads AND stores pending the compiler must assume
    // out-of-order completion and waits vmcnt(0) before every MAC, which would expose each prefetch
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
    if (nseg > 0) load_bfrag<T>(bfa, wpack, wp.pack(0), wn, lane);
    load_afrag<T>(afn, smem, wp.at(pb), ao);
    // a fragment load is ALWAYS in flight behind the one being multiplied (the last segment re-requests its own pack):
    // every path then has the same number of younger loads outstanding, so the compiler's in-order vmcnt waits inside
    // the MAC loops never have to drain the prefetch
#ifdef MSHGNN_ABLATE
    if ABL(dbg & 512) {     // both register buffers filled once, no weight streaming inside the layer (timing only, wrong results)
        load_bfrag<T>(bfb, wpack, wp.pack(0), wn, lane);
        for (int sgi = 0; sgi < nseg; sgi += 2) {
            fs_walk<T, HS, CB>(wp, sgi, pb, afn, acc, bfa, smem, ao, dbg);
            if (sgi + 1 < nseg) fs_walk<T, HS, CB>(wp, sgi + 1, pb, afn, acc, bfb, smem, ao, dbg);
        }
        return;
    }
#endif
#ifdef MSHGNN_SEG_STAMPS
#define FS_SEGCLK(i) do { if (segclk && lane == 0) segclk[i] = clock64(); } while (0)
#else
#define FS_SEGCLK(i) do { } while (0)
#endif
#ifdef MSHGNN_MAC_PRIO
    __builtin_amdgcn_s_setprio(MSHGNN_MAC_PRIO);      // (experiment: the MAC phase's wave wins issue arbitration against the co-resident workgroup's epilogue wave)
#endif
    for (int sgi = 0; sgi < nseg; sgi += 2) {
        load_bfrag<T>(bfb, wpack, wp.pack(min(sgi + 1, nseg - 1)), wn, lane);
        FS_SEGCLK(sgi);
        fs_walk<T, HS, CB>(wp, sgi, pb, afn, acc, bfa, smem, ao, dbg);
        if (sgi + 1 < nseg) {
            load_bfrag<T>(bfa, wpack, wp.pack(min(sgi + 2, nseg - 1)), wn, lane);
            FS_SEGCLK(sgi + 1);
            fs_walk<T, HS, CB>(wp, sgi + 1, pb, afn, acc, bfb, smem, ao, dbg);
        }
    }
    FS_SEGCLK(nseg);
#ifdef MSHGNN_MAC_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// decoder (+ fused wrapper MSE and decoder backward) on the X_L tile in LDS: shared tail of k_stack_fwd / k_slab_fwd
// TOLDS (k_slab_step): the dX_L rows also go into the out-type nodes' LDS blocks (rows past the batch: zeros), where the backward sweep of the same launch
// picks them up; the reduction scratch at the start of LDS must not reach those blocks (checked on the host)
// the tail's global operands of one pass (decoder weights / bias, output mask, labels of this thread's rows): requested in one go -- by the tail itself, or
// (compile-time programs) before the last layer's store phase, so that the tail starts with its operands in registers instead of behind that phase's drain
template <int DMAX, int NPP> struct DecOps { float Wv[DMAX][8], bv[DMAX], mk[NPP][DMAX], yv[NPP][DMAX]; int labv[NPP]; };
template <int DMAX, int NPP>
__device__ __forceinline__ void decoder_ops_load_w(const StackArgs& a, int tid, DecOps<DMAX, NPP>& o) {      // decoder weights / bias: once per tile
    const int c = tid & 15;
    const float* W = a.params + a.off_dec_w;
#pragma unroll
    for (int dd = 0; dd < DMAX; ++dd) {
        const int dc = min(dd, a.dout - 1);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(W + dc * H + c * 8), wb = *reinterpret_cast<const f32x4*>(W + dc * H + c * 8 + 4);
        o.Wv[dd][0] = wa[0]; o.Wv[dd][1] = wa[1]; o.Wv[dd][2] = wa[2]; o.Wv[dd][3] = wa[3];
        o.Wv[dd][4] = wb[0]; o.Wv[dd][5] = wb[1]; o.Wv[dd][6] = wb[2]; o.Wv[dd][7] = wb[3];
        o.bv[dd] = a.params[a.off_dec_b + dc];
    }
}
template <int THREADS, int DMAX, int NPP>
__device__ __forceinline__ void decoder_ops_load(const StackArgs& a, int tid, int w0, int B, int f0, bool with_w, DecOps<DMAX, NPP>& o) {      // one pass's output mask and labels (+ the weights)
    const int c = tid & 15, row = (tid >> 4) & 15;
    const bool ce = a.labels != nullptr;
    if (with_w) decoder_ops_load_w<DMAX, NPP>(a, tid, o);
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int f = min(f0 + i * (THREADS / 256), a.n_out - 1);
        const size_t r = (size_t)min(w0 + row, B - 1) * a.n_out + f;
        o.labv[i] = 0;
        if (ce) o.labv[i] = a.labels[r] != 0;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) {
            const int dc = min(dd, a.dout - 1);
            o.mk[i][dd] = a.out_mask[f * a.dout + dc];
            o.yv[i][dd] = a.y ? a.y[r * a.dout + dc] : 0.f;
        }
    }
}
template <typename T, int THREADS, int DMAX, bool SPLIT = false, bool TOLDS = false, int NPP = 2, bool PRE = false>      // DMAX: compile-time bound on the output channels (4 or 8): loops, loads and registers scale with it; NPP: nodes per pass; PRE: *pre holds pass 0's operands
__device__ __forceinline__ void decoder_tail_impl(const StackArgs& a, char* smem, int tid, int lane, int wv, int w0, int B, DecOps<DMAX, NPP>* pre = nullptr) {
    // decoder on the out-type rows of X_L (hgnn_c2.py:176-189): thread = (node, row, 8-column chunk).  With y (mshgnn_step_mse) the
    // same threads also take the wrapper MSE (gnnLightning.py:633-639) and the decoder backward: dX_L rows to global for
    // k_stack_bwd, and this tile's partial decoder gradients + loss partial into dec_slabs[tile] (summed by k_finalize).
    {
        const int c = tid & 15, row = (tid >> 4) & 15;
        const float* W = a.params + a.off_dec_w;
        const bool ce = a.labels != nullptr, fuse = a.y != nullptr || ce;      // ce: cross entropy over the two logits of a foot (gnnLightning.py:640-648)
        T* dxl = reinterpret_cast<T*>(a.ws + a.dx_off[a.L]);
        float accw[DMAX][8], accb[DMAX], lsum = 0.f;
        if (fuse) {
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) { accb[dd] = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) accw[dd][e] = 0.f; }
        }
        // two nodes per pass, every global load of a pass before its first store: a load waited for while stores are in flight
        // costs a full vmcnt(0) drain (loads and stores retire out of order with respect to each other)
        // (unconditional loads with clamped indices, so that they are issued back to back and waited for once)
        DecOps<DMAX, NPP> ops;
        if constexpr (PRE) ops = *pre; else decoder_ops_load_w<DMAX, NPP>(a, tid, ops);
        auto& Wv = ops.Wv; auto& bv = ops.bv; auto& mk = ops.mk; auto& yv = ops.yv; auto& labv = ops.labv;
        FS_STAMP(24);
        for (int f0 = tid >> 8; f0 < a.n_out; f0 += NPP * (THREADS / 256)) {
            float ov[NPP][DMAX], dxv[NPP][8];
            if (!(PRE && f0 == (tid >> 8))) decoder_ops_load<THREADS, DMAX, NPP>(a, tid, w0, B, f0, false, ops);
#pragma unroll
            for (int i = 0; i < NPP; ++i) {
                const int f = f0 + i * (THREADS / 256);
                const bool live = f < a.n_out;
                f32x4 x0, x1;
                lds_load_oct<T>(smem, a.node0 + (live ? f : f0), row, c * 8, x0, x1);
                if constexpr (SPLIT) {      // X_L = hi + lo
                    f32x4 l0, l1;
                    lds_load_oct<T>(smem, a.lo_blk + a.node0 + (live ? f : f0), row, c * 8, l0, l1);
                    x0 += l0; x1 += l1;
                }
                const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) dxv[i][e] = 0.f;
                const bool ok = live && w0 + row < B;
#pragma unroll
                for (int dd = 0; dd < DMAX; ++dd) {
                    ov[i][dd] = 0.f;
                    if (dd < a.dout && live) {
                        float sum = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum = __builtin_fmaf(x[e], Wv[dd][e], sum);      // (explicit fused multiply-adds in this tail: every instantiation -- one-launch
                        sum = row16_sum(sum);                                                     //  step or not, 4 / 8 waves, preloaded operands or not -- then rounds the same way)
                        ov[i][dd] = (sum + bv[dd]) * mk[i][dd];
                    }
                }
                if (fuse && ok) {
                    float ce_g[2] = {0.f, 0.f};
                    if (ce) {      // dL/dlogit = (softmax - onehot) / rows, the arithmetic of k_dec_bwd
                        const float l0 = ov[i][0], l1 = ov[i][1];
                        const float m = fmaxf(l0, l1), e0 = expf(l0 - m), e1 = expf(l1 - m), se = e0 + e1;
                        ce_g[0] = (e0 / se - (labv[i] ? 0.f : 1.f)) * a.inv_n; ce_g[1] = (e1 / se - (labv[i] ? 1.f : 0.f)) * a.inv_n;
                        if (c == 0) lsum += (m + logf(se)) - (labv[i] ? l1 : l0);
                    }
#pragma unroll
                    for (int dd = 0; dd < DMAX; ++dd) {
                        if (dd < a.dout) {
                            float g;
                            if (ce) g = ce_g[dd & 1] * mk[i][dd];
                            else {
                                const float dlt = ov[i][dd] - yv[i][dd];
                                g = 2.0f * dlt * a.inv_n * mk[i][dd];
                                if (c == 0) lsum = __builtin_fmaf(dlt, dlt, lsum);
                            }
                            accb[dd] += g;
#pragma unroll
                            for (int e = 0; e < 8; ++e) { accw[dd][e] = __builtin_fmaf(g, x[e], accw[dd][e]); dxv[i][e] = __builtin_fmaf(g, Wv[dd][e], dxv[i][e]); }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NPP; ++i) {
                const int f = f0 + i * (THREADS / 256);
                const bool ok = f < a.n_out && w0 + row < B;
                const size_t r = (size_t)(w0 + row) * a.n_out + f;
                if (c == 0 && ok) {
#pragma unroll
                    for (int dd = 0; dd < DMAX; ++dd) if (dd < a.dout) a.out[r * a.dout + dd] = ov[i][dd];
                }
                if (fuse && ok) {
                    if constexpr (SPLIT) {
                        u32x4 hi, lo;
                        split_oct(f32x4{dxv[i][0], dxv[i][1], dxv[i][2], dxv[i][3]}, f32x4{dxv[i][4], dxv[i][5], dxv[i][6], dxv[i][7]}, hi, lo);
                        T* q = dxl + 2 * act_idx(w0 + row, a.node0 + f, B) + c * 8;      // split plan rows: [hi 128 | lo 128]
                        *reinterpret_cast<u32x4*>(q) = hi; *reinterpret_cast<u32x4*>(q + H) = lo;
                    } else store8<T>(dxl + act_idx(w0 + row, a.node0 + f, B) + c * 8, dxv[i]);
                }
                if constexpr (TOLDS) {
                    if (f < a.n_out) {
                        if (!ok) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) dxv[i][e] = 0.f;
                        }
                        if constexpr (SPLIT) {      // hi plane in the node's block, lo plane in block lo_blk + node
                            u32x4 hi, lo;
                            split_oct(f32x4{dxv[i][0], dxv[i][1], dxv[i][2], dxv[i][3]}, f32x4{dxv[i][4], dxv[i][5], dxv[i][6], dxv[i][7]}, hi, lo);
                            *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(a.node0 + f, row, c)) = hi;
                            *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(a.lo_blk + a.node0 + f, row, c)) = lo;
                        } else store8<T>(reinterpret_cast<T*>(smem + lds_chunk<T>(a.node0 + f, row, c)), dxv[i]);
                    }
                }
            }
        }
        FS_STAMP(25);
        if (fuse) {
            // reduce over the 4 rows of the wave (lanes 16 apart), then over the 8 waves through LDS (the X tile is dead after the barrier)
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) {
                if (dd < a.dout) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { accw[dd][e] += __shfl_xor(accw[dd][e], 16, 64); accw[dd][e] += __shfl_xor(accw[dd][e], 32, 64); }
                    accb[dd] += __shfl_xor(accb[dd], 16, 64); accb[dd] += __shfl_xor(accb[dd], 32, 64);
                }
            }
            lsum += __shfl_xor(lsum, 16, 64); lsum += __shfl_xor(lsum, 32, 64);
            FS_STAMP(26);
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem + a.red_off);          // [waves][8 H + 16]
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
            FS_STAMP(27);
            float* slab = a.dec_slabs + (size_t)blockIdx.x * DEC_SLAB_FLOATS;
            for (int i = tid; i < 8 * H + 9; i += THREADS) {
                if (i >= a.dout * H && i < 8 * H) continue;       // rows of unused output channels (k_finalize reads dout rows only)
                float s2 = 0.f;
#pragma unroll
                for (int k = 0; k < THREADS / 64; ++k) s2 += red[k * DEC_SLAB_FLOATS + i];
                slab[i] = s2;
            }
        }
    }
}

template <typename T, int THREADS, bool SPLIT = false, bool TOLDS = false>
__device__ __forceinline__ void decoder_tail(const StackArgs& a, char* smem, int tid, int lane, int wv, int w0, int B) {
    if (a.dout <= 4) decoder_tail_impl<T, THREADS, 4, SPLIT, TOLDS>(a, smem, tid, lane, wv, w0, B);
    else decoder_tail_impl<T, THREADS, 8, SPLIT, TOLDS>(a, smem, tid, lane, wv, w0, B);
}
// compile-time programs: out channels bound DMAX from the program, all (<= 4 per 256 threads) output nodes in one pass
constexpr int DEC_NPP_STATIC = 4;

struct DecArgs {
    const void* xl; void* dxl; const float* params; const float* out_mask; float* out; const float* gout; float* slabs;
    int64_t off_w, off_b; int B, NN, node0, n_out, dout, slab0;
    const float* y; float* loss; float inv_n;   // fused MSE: gout = 2 (out - y) / n computed on the fly, loss accumulated
    const int32_t* labels;                      // fused cross entropy (dout == 2): gout = (softmax(out) - onehot) / rows
};


struct GradwArgs {
    const char* ws; size_t buf_off[BUF_COUNT];
    const void* x[MSHGNN_MAX_TYPES]; int64_t pitch[MSHGNN_MAX_TYPES]; int nodes[MSHGNN_MAX_TYPES]; int vb[MSHGNN_MAX_TYPES];
    const int* items; const int* lanes; const int* lane_order; const uint8_t* signs; float* slabs; int B, n_lanes, n_parts, n_pad;
    int aligned;   // all raw-input rows 16-byte aligned with whole-chunk pitch
    int dbg;   // timing ablations (MSHGNN_DBG_GW): 1 no global loads, 2 no LDS staging, 4 no MFMA phase, 8 no slab store
    long long* stamps;   // MSHGNN_STAMPS_GW: thread 0 of every workgroup accumulates clock64() deltas of the step phases
    SeriesSrc ser;       // raw-input operands gathered from the sequence's series (mshgnn_step_mse_series without materialised windows)
};

// bf16: P/Q staged row-major ([window][feature], pitch 160 elements = 320 B: 4 consecutive windows land on
// disjoint 16-bank ranges) and read as MFMA operands with the hardware-transposing ds_read_b64_tr_b16 (guide T10):
// both operands are K(=window)-strided in memory.  32x32x16 bf16 MFMA, wave = 64x64 of the 128x128 tile.
constexpr int GWB_KW = 64;
constexpr int GWB_PITCH = 128;           // no row padding: 16-byte chunk c of row r lives at chunk slot c ^ 4 (r & 3) instead (the four rows a
                                         // transposed read touches land in four different 64-byte bank windows, as they did with a 320-byte pitch)
__device__ __forceinline__ int gwb_elem(int row, int col) { return row * GWB_PITCH + ((((col >> 3) ^ ((row & 3) << 2)) << 3) | (col & 7)); }
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int w_base, int col_base, int lane) {
    // lane = 16 g + 4 q + p: supplies the address of row q, columns 4p..4p+3 of its group's 4x16 block;
    // receives column (lane & 15) of the 4 rows.  g&1 selects the 16-column half, g>>1 the K half (8 windows).
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const __bf16* p0 = tile + gwb_elem(w_base + 8 * (g >> 1) + q, col_base + 16 * (g & 1) + 4 * pp);     // (row + 4 has the same swizzle)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * GWB_PITCH));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}


struct FinArgs { const int* fin; const int* fin0; const int* targets; const float* slabs; const float* dec_slabs; float* grad; int n_lanes, n_parts; float* loss; float inv_n;
                 int n_dec; int accumulate; };   // decoder partial slabs: NWG_DEC (k_dec_bwd) or one per tile (fused forward)

struct mshgnn_gen_state;      // generic-width engine (mshgnn_gen.hip)
struct ProfRec { int slot; hipEvent_t a, b; };
struct mshgnn_plan {
    HostPlan hp;
    bool prof = false;
    std::vector<ProfRec> recs;          // recorded, not yet read
    std::vector<hipEvent_t> free_events;
    int* d_tables = nullptr; uint8_t* d_signs = nullptr; float* d_out_mask = nullptr;
    PackDesc* d_packs = nullptr; BiasDesc* d_biases = nullptr;
    bool attr_set = false;
    bool use_fused = false;             // bf16 plan: fused stack kernels (MSHGNN_FUSED=0 selects the per-layer kernels)
    bool use_slab = false;              // slab variant of the stack kernels (MSHGNN_SLAB=0 selects the 8-wave ones)
    bool slab_force = false; int n_cu = 256;
    int64_t step_chunk = 32768;         // one-call steps over at least twice as many windows run as sub-steps of at least this many (StepChunk; MSHGNN_STEP_CHUNK, 0 = never)
    bool use_step = false;              // one-call steps on the slab kernels: k_slab_step (MSHGNN_STEP_KERNEL=0: two launches)
    bool use_spec = true;               // ... on the specialised kernel where the plan has one (MSHGNN_SPEC=0 at plan creation: the interpreting kernel)
    const char* spec_name = "";         // the compile-time program this plan's tables equal ("" = none)
    std::string spec_name_buf;          // (owner of that name for a program attached after the build: mshgnn_plan_attach_program)
    int stagger = 0;                             // StackArgs.stagger of the two-workgroups-per-CU stack kernels (MSHGNN_STAGGER)
    int n_types = 0;
    mshgnn_gen_state* gen = nullptr;    // set: this plan runs on the generic-width engine (hidden != 128, many nodes, ...), hp is unused
    int dbg = 0, dbg_gw = 0;            // timing ablations (instrumented builds only: read once from MSHGNN_DBG / MSHGNN_DBG_GW at plan creation)
    // a slab workgroup has 4 waves for a whole tile (a lone one is ~12 % slower than the 8-wave kernels' workgroup: 69 against 62 us at 3 layers), two fit a CU:
    // it pays off from the first tile the 8-wave kernels would need a second round for (257 .. 383 tiles, 8-wave against slab launch: 113 / 77 us at 3 layers,
    // 425 / 295 at 8, MiniCheetah-K4 440 / 330 -- tools/slab_threshold_sweep.py; rounds 3-4 switched at 1.5 tiles per CU)
    bool slab_for(int tiles) const { return use_slab && (slab_force || tiles > n_cu); }
};

// in-kernel stamp buffers of the instrumented builds (tools/stamps_*.py pass a device pointer through the environment)
inline long long* stamp_ptr(const char* name) {
#if defined(MSHGNN_FS_STAMPS) || defined(MSHGNN_SEG_STAMPS) || MSHGNN_GW_STAMPS_BUILD
    const char* e = getenv(name);
    return e ? reinterpret_cast<long long*>((uintptr_t)strtoull(e, nullptr, 0)) : nullptr;
#else
    (void)name; return nullptr;
#endif
}

// bracket one kernel launch with events when profiling
struct ProfScope {
    mshgnn_plan* p; hipStream_t st; ProfRec r{}; bool on;
    ProfScope(const mshgnn_plan* pc, int slot, hipStream_t s) : p(const_cast<mshgnn_plan*>(pc)), st(s), on(pc->prof) {
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!p->free_events.empty()) { e = p->free_events.back(); p->free_events.pop_back(); } else (void)hipEventCreate(&e); return e; };
        r.slot = slot; r.a = get(); r.b = get();
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, st); p->recs.push_back(r); } }
};

template <typename K> inline int set_lds_attr(K kernel, int bytes) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return MSHGNN_OK;
}

inline int vec_bytes(const void* base, int64_t pitch_elems, int esize) {
    const uint64_t a = (uint64_t)(uintptr_t)base | (uint64_t)(pitch_elems * esize);
    if ((a & 15) == 0) return 16;
    if ((a & 7) == 0) return 8;
    if ((a & 3) == 0) return 4;
    return esize;
}



// Window parts of a weight-gradient launch for THIS batch: at most the plan's n_parts (the slab buffer is sized for it).  A launch lasts as long as its
// longest part, a part's steps slow down with the number of workgroups that stream beside it (more so once CUs hold a third workgroup), every part
// costs a slab per lane that the finalize launch reads back, and a grid that needs (nearly) three workgroups on EVERY CU runs into the placement's
// slack.  Measured (round 4, A1-C2 pruned plan, 56 lanes of 2 items, 8192 windows = 128 chunks; us): 13 parts = 728 workgroups 89.3, 12: 71.8,
// 11: 73.6, 10: 74.9, 9: 81.4, 8 (16 chunks each, 448 workgroups): 69.2, 7: 76.0, 6: 83.3, finalize 9 + 0.012 per workgroup; MiniCheetah-K4 L=3
// (128 lanes): 6 parts = 768 workgroups 92, 5 parts 76.5, 4 parts 81.8.  Model fitted to those, per step of one item:
//     t(W) = 1.1 + 0.0024 W us up to 480 workgroups (under two per CU), 1.1 + 0.0032 W beyond;   launch = ceil(chunks / q) ipl t(lanes q) + 0.012 lanes q
// with the grid kept at or below 2.8 workgroups per CU.  The plan's own count (as many parts as fit) stays unless the model sees > 3 % in another.
inline int gw_parts_for(int plan_parts, int lanes, int ipl, int64_t B, int chunk_windows, int n_cu) {
    static const int forced = [] { const char* e = TUNE_ENV("MSHGNN_GW_PARTS"); return e ? atoi(e) : 0; }();      // (measurements)
    if (forced > 0) return std::min(forced, plan_parts);
    const int64_t nchunks = (B + chunk_windows - 1) / chunk_windows;
    const double cu_scale = 256.0 / (double)std::max(1, n_cu);                               // (the fit is per CU: workgroup counts scaled to a 256-CU chip)
    const int cap_wg = 45 * std::max(1, n_cu) / 16;                                         // 720 on 256 CUs
    const int qmax = std::max(1, std::min(plan_parts, cap_wg / std::max(1, lanes)));
    auto cost = [&](int q) {
        const double W = (double)lanes * q * cu_scale;
        const double t = 1.1 + (W <= 480.0 ? 0.0024 : 0.0032) * W;
        return (double)((nchunks + q - 1) / q) * std::max(1, ipl) * t + 0.012 * W; };
    int best = qmax; double best_cost = cost(qmax);
    for (int q = 1; q < qmax; ++q) if (cost(q) < best_cost) { best_cost = cost(q); best = q; }
    return best_cost < 0.97 * cost(qmax) ? best : qmax;
}

// k_finalize launch of a step (mshgnn.hip): fixed-order slab sums -> flat gradient (+ fused loss)
int run_finalize(const mshgnn_plan* p, const mshgnn_ws_layout& lay, char* ws, float* gparams, int B, float* loss, bool is_ce, bool dec_done,
                 int gw_phase, hipStream_t st, int gw_parts);
// split-bf16 parity plan (mshgnn_x3.hip)
int x3_set_attrs(mshgnn_plan* p);
int x3_attach_program(mshgnn_plan* p, void* selector);
int x3_forward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, float* out, char* ws, int64_t batch,
               int training, hipStream_t st, const float* y_fused, const SeriesSrc* series = nullptr, bool* stack_step_done = nullptr,
               const int32_t* labels_fused = nullptr);
int x3_backward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* gout, float* gparams, char* ws,
                int64_t batch, hipStream_t st, const float* out, const float* y, float* loss, const int32_t* labels, bool dec_done, int gw_phase, bool stack_done = false);
int x3_launch_prep(const PrepArgs& a, hipStream_t st);
// weight-image packing for the other engines (mshgnn.hip): k_prep<__bf16> or the hi / lo images of k_prep_x3
int launch_prep(const PrepArgs& a, bool split, hipStream_t st);
// generic-width engine (mshgnn_gen.hip)
int gen_create(mshgnn_plan* p, const mshgnn_desc* desc);
void gen_destroy(mshgnn_plan* p);
const mshgnn_info* gen_info(const mshgnn_plan* p);
const std::vector<mshgnn_kernel_stat>* gen_kstats(const mshgnn_plan* p);
void gen_layout(const mshgnn_plan* p, int64_t batch, int training, mshgnn_ws_layout* out);
int gen_host_compile(const mshgnn_desc* desc, mshgnn_info* info, int32_t* n_tables);
int gen_forward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, float* out, char* ws, int64_t batch,
                int training, hipStream_t st);
int gen_backward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* gout, float* gparams, char* ws,
                 int64_t batch, hipStream_t st, const float* out, const float* y, float* loss, const int32_t* labels);
