// What a lone wave per SIMD pays for other instructions between its MFMAs: 16 v_mfma_f32_16x16x32_bf16 (4 accumulators x 4 K steps) per block with
// N copies of instruction X behind every MFMA; cycles per block (ideal 256 when X hides in the MFMA gaps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MF(j, t) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(w[t + 4 * (j & 1)]), "v"(x[t + 4 * (j >> 1)]));
template <int X, int N> __device__ __forceinline__ void filler(u32x4& d0, u32x4& d1, int& vv, int& ss, int ad) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if constexpr (X == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(n & 1 ? d1 : d0) : "v"(ad));
        if constexpr (X == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vv) : "v"(ad));
        if constexpr (X == 3) asm volatile("s_waitcnt lgkmcnt(15)");
        if constexpr (X == 4) asm volatile("s_nop 0");
        if constexpr (X == 5) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(ss) : "v"(vv));
        if constexpr (X == 6) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<unsigned long long*>(&d0)) : "v"(ad));
    }
}
template <int X, int N> __global__ __launch_bounds__(256, 1) void k(const bf16x8* in, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 w[8], x[8];
    for (int i = 0; i < 8; ++i) { w[i] = in[threadIdx.x + 256 * i]; x[i] = in[threadIdx.x + 256 * (8 + i)]; }
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{1u, 2u, 3u, 4u};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    u32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}; int vv = lane, ss = 0;
    const int row = lane & 15, g = lane >> 4;
    const int ad = row * 256 + (((g * 4) ^ ((row & 15) ^ ((row & 4) << 1))) << 4);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { MF(j, t) filler<X, N>(d0, d1, vv, ss, ad); }
        if constexpr (X == 0 || X == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1));
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    const long long t1 = clock64();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + d0[0] + d1[1] + vv + ss;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int X, int N> void run(const char* name, const bf16x8* in, float* out, long long* cyc) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<X, N>), dim3(256), dim3(256), 65536, 0, in, out, cyc, iters); (void)hipDeviceSynchronize(); }
    std::vector<long long> c(256); (void)hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    printf("%-16s x%d per MFMA: %7.1f cycles per 16-MFMA block  (+%.1f per filler)\n", name, N, s / 256 / iters, N ? (s / 256 / iters - 272) / (16.0 * N) : 0.0);
}
int main() {
    bf16x8* in; float* out; long long* cyc;
    (void)hipMalloc(&in, 256 * 16 * 16); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    std::vector<unsigned short> h(256 * 16 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (i * 7919 % 251);
    (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<4, 0>("none", in, out, cyc);
    run<0, 1>("ds_read_b128", in, out, cyc); run<0, 2>("ds_read_b128", in, out, cyc);
    run<6, 1>("ds_read_b64", in, out, cyc); run<6, 2>("ds_read_b64", in, out, cyc);
    run<1, 1>("v_add_u32", in, out, cyc); run<1, 2>("v_add_u32", in, out, cyc); run<1, 4>("v_add_u32", in, out, cyc);
    run<3, 1>("s_waitcnt", in, out, cyc); run<3, 2>("s_waitcnt", in, out, cyc);
    run<4, 1>("s_nop 0", in, out, cyc); run<4, 2>("s_nop 0", in, out, cyc); run<4, 4>("s_nop 0", in, out, cyc);
    run<5, 1>("v_readlane", in, out, cyc); run<5, 2>("v_readlane", in, out, cyc);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
