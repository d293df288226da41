// Model of the wide stack kernel's MAC phase (one wave per SIMD, 18 accumulator slots x 2 halves, window fragments re-read from LDS per MAC, layer
// program in registers) without weight-fragment traffic: tunes the walk structure offline.
//   V=0: every slot header visited per segment (the first wide kernel)      V=1: range walk -- per segment a switch into the unrolled slot
//   sequence at its first slot, exit after its last; the next MAC's LDS base is computed one MAC ahead
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NS = 18, BLK = 8192;
struct Acc { f32x4 c[2][2]; };
struct XF { bf16x8 v[2][4]; };
struct Prog { int pk, c0, c1, rg, pb[2];      // rg: first slot | (last slot << 8) per segment
    __device__ int at(int i) const { const int v = i < 256 ? __builtin_amdgcn_readlane(pb[0], (i >> 2) & 63) : __builtin_amdgcn_readlane(pb[1], (i >> 2) & 63); return (v >> ((i & 3) << 3)) & 0xff; } };

template <bool AG> __device__ __forceinline__ void mfma(f32x4& c, const bf16x8& w, const bf16x8& x) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(x));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
// MAC + refill; V1: `va` holds the four LDS addresses of the NEXT fragment on entry and is advanced to the one after it from `nb2` at the end
template <bool AG, int V> __device__ __forceinline__ void mac_refill(Acc& a, XF& x, const bf16x8 (&w)[8], const char* smem, const int (&ao)[4], int (&va)[4], int nb) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { mfma<AG>(a.c[h][0], w[t], x.v[h][t]); mfma<AG>(a.c[h][1], w[4 + t], x.v[h][t]); }
        const int ad = V == 0 ? nb * BLK + ao[t] : va[t];
#pragma unroll
        for (int h = 0; h < 2; ++h) x.v[h][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + ad + h * 4096));
        __builtin_amdgcn_sched_barrier(0);
    }
}
#define SLOT(U) case U: if constexpr (U < NS) { \
        const int cnt = ((U < 10 ? c0 : c1) >> (3 * (U % 10))) & 7; \
        for (int kk = 0; kk < cnt; ++kk) { \
            const int nb2 = p.at(pb + 2); \
            if (U < 16) mac_refill<true, 1>(acc[U], x, w, smem, ao, va, 0); else mac_refill<false, 1>(acc[U], x, w, smem, ao, va, 0); \
            for (int t = 0; t < 4; ++t) va[t] = nb2 * BLK + ao[t]; \
            ++pb; __builtin_amdgcn_sched_barrier(0); \
        } \
        if (U == last) break; } [[fallthrough]];

template <int V> __global__ __launch_bounds__(256, 1) void k(const bf16x8* in, const int* prog, float* out, long long* cyc, int nseg, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 w[8];
    for (int i = 0; i < 8; ++i) w[i] = in[threadIdx.x + 256 * i];
    for (int i = threadIdx.x; i < NS * BLK / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c01u + i, 0x3c003c02u, 0x3c013c00u, 0x3c003c03u};
    Acc acc[NS];
    for (int u = 0; u < NS; ++u) for (int h = 0; h < 2; ++h) for (int j = 0; j < 2; ++j) acc[u].c[h][j] = f32x4{0, 0, 0, 0};
    Prog p; p.pk = prog[lane]; p.c0 = prog[64 + lane]; p.c1 = prog[128 + lane]; p.rg = prog[192 + lane]; p.pb[0] = prog[256 + lane]; p.pb[1] = prog[320 + lane];
    int ao[4];
    { const int row = lane & 15, g = lane >> 4; for (int t = 0; t < 4; ++t) { int c = g * 4 + t; ao[t] = row * 256 + ((c ^ ((row & 15) ^ ((row & 4) << 1))) << 4); asm volatile("" : "+v"(ao[t])); } }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        int pb = 1;
        XF x;
        { const int b = p.at(pb) * BLK; for (int t = 0; t < 4; ++t) for (int h = 0; h < 2; ++h) x.v[h][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + b + h * 4096 + ao[t])); }
        int va[4]; { const int b = p.at(pb + 1) * BLK; for (int t = 0; t < 4; ++t) va[t] = b + ao[t]; }
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const int c0 = __builtin_amdgcn_readlane(p.c0, sgi), c1 = __builtin_amdgcn_readlane(p.c1, sgi);
            if constexpr (V == 0) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    const int cnt = ((u < 10 ? c0 : c1) >> (3 * (u % 10))) & 7;
                    for (int kk = 0; kk < cnt; ++kk) { const int nb = p.at(++pb); if (u < 16) mac_refill<true, 0>(acc[u], x, w, smem, ao, va, nb); else mac_refill<false, 0>(acc[u], x, w, smem, ao, va, nb); }
                }
            } else {
                const int rg = __builtin_amdgcn_readlane(p.rg, sgi), first = rg & 0xff, last = rg >> 8;
                switch (first) {
                    SLOT(0) SLOT(1) SLOT(2) SLOT(3) SLOT(4) SLOT(5) SLOT(6) SLOT(7) SLOT(8) SLOT(9) SLOT(10) SLOT(11) SLOT(12) SLOT(13) SLOT(14) SLOT(15) SLOT(16) SLOT(17)
                    default: break;
                }
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int u = 0; u < NS; ++u) {
        if (u < 16) asm volatile("s_nop 15" : "+a"(acc[u].c[0][0]), "+a"(acc[u].c[0][1]), "+a"(acc[u].c[1][0]), "+a"(acc[u].c[1][1]));
        else asm volatile("s_nop 15" : "+v"(acc[u].c[0][0]), "+v"(acc[u].c[0][1]), "+v"(acc[u].c[1][0]), "+v"(acc[u].c[1][1]));
        for (int h = 0; h < 2; ++h) for (int j = 0; j < 2; ++j) s += acc[u].c[h][j][0];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    bf16x8* in; float* out; long long* cyc; int* prog;
    (void)hipMalloc(&in, 256 * 16 * 16); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&prog, 384 * 4);
    std::vector<unsigned short> h(256 * 16 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (i * 7919 % 251);
    (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * BLK);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * BLK);
    struct Pat { const char* name; std::vector<std::vector<std::pair<int, int>>> segs; };
    std::vector<Pat> pats;
    { Pat p{"A1-C2 layer 0, slots = nodes", {}};
      p.segs = {{{0,1},{1,1}}, {{0,1},{1,1}}, {{0,1},{1,1}}, {{0,1},{1,1}},
                {{2,1},{3,1},{4,1},{5,1},{6,1},{7,1},{8,1},{9,1},{10,1},{11,1},{12,1},{13,1}}, {{2,1},{8,1}}, {{5,1},{11,1}},
                {{2,1},{3,2},{4,1},{5,1},{6,2},{7,1},{8,1},{9,2},{10,1},{11,1},{12,2},{13,1}}, {{4,1},{7,1},{10,1},{13,1}},
                {{14,1},{15,1},{16,1},{17,1}}, {{14,1},{15,1},{16,1},{17,1}}};
      pats.push_back(p); }
    { Pat p{"A1-C2 layer 0, joints permuted (contiguous ranges)", {}};   // joint slots: [j0 j6 | j3 j9 | j2 j5 j8 j11 | j1 j4 j7 j10] = slots 2..13
      p.segs = {{{0,1},{1,1}}, {{0,1},{1,1}}, {{0,1},{1,1}}, {{0,1},{1,1}},
                {{2,1},{3,1},{4,1},{5,1},{6,1},{7,1},{8,1},{9,1},{10,1},{11,1},{12,1},{13,1}}, {{2,1},{3,1}}, {{4,1},{5,1}},
                {{2,1},{3,1},{4,1},{5,1},{6,1},{7,1},{8,1},{9,1},{10,2},{11,2},{12,2},{13,2}}, {{6,1},{7,1},{8,1},{9,1}},
                {{14,1},{15,1},{16,1},{17,1}}, {{14,1},{15,1},{16,1},{17,1}}};
      pats.push_back(p); }
    const int iters = 200;
    for (auto& p : pats) {
        std::vector<int> pr(384, 0); int macs = 0; std::vector<int> blocks;
        for (size_t s = 0; s < p.segs.size(); ++s) {
            int first = 99, last = 0;
            for (auto& e : p.segs[s]) { pr[64 + (e.first / 10) * 64 + s] |= e.second << (3 * (e.first % 10)); macs += e.second; first = std::min(first, e.first); last = std::max(last, e.first);
                                        for (int q = 0; q < e.second; ++q) blocks.push_back((e.first * 7 + q * 3) % NS); }
            pr[192 + s] = first | (last << 8);
        }
        std::vector<int> wb(512, 0); wb[0] = (int)p.segs.size(); for (size_t i = 0; i < blocks.size(); ++i) wb[1 + i] = blocks[i];
        for (int i = 0; i < 128; ++i) pr[256 + i] = wb[4 * i] | (wb[4 * i + 1] << 8) | (wb[4 * i + 2] << 16) | (wb[4 * i + 3] << 24);
        (void)hipMemcpy(prog, pr.data(), 384 * 4, hipMemcpyHostToDevice);
        for (int v = 0; v < 2; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), NS * BLK, 0, in, prog, out, cyc, (int)p.segs.size(), iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), NS * BLK, 0, in, prog, out, cyc, (int)p.segs.size(), iters);
                (void)hipDeviceSynchronize();
            }
            std::vector<long long> c(256); (void)hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto x : c) s += x;
            const double per_it = s / 256 / iters;
            printf("%-52s V%d %8.0f cycles per pass, %6.1f per MAC (ideal 256), %d MACs\n", p.name, v, per_it, per_it / macs, macs);
        }
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
