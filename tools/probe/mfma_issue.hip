// Microbenchmark: cycles per v_mfma_f32_16x16x32_bf16 for one wave per SIMD, by how the accumulators are held and how the MFMAs are written
// (compiler builtin vs asm statements with "a" / "v" accumulator constraints), 4 independent accumulators x 4 dependent K steps per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE> __global__ __launch_bounds__(256, 1) void k(const bf16x8* in, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x;
    bf16x8 w[8], x[8];
    for (int i = 0; i < 8; ++i) { w[i] = in[lane + 256 * i]; x[i] = in[lane + 256 * (8 + i)]; }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t + 4 * (j & 1)], x[t + 4 * (j >> 1)], acc[j], 0, 0, 0);
                else if constexpr (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(w[t + 4 * (j & 1)]), "v"(x[t + 4 * (j >> 1)]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(w[t + 4 * (j & 1)]), "v"(x[t + 4 * (j >> 1)]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    const long long t1 = clock64();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + lane] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    bf16x8* in; float* out; long long* cyc;
    hipMalloc(&in, 256 * 16 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    std::vector<unsigned short> h(256 * 16 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (i * 7919 % 251);      // random-ish bf16 near 0.01
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int grid : {1, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(grid); hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : c) s += v;
            printf("grid %3d mode %d (%s): %.2f cycles per MFMA\n", grid, mode, mode == 0 ? "builtin" : mode == 1 ? "asm +a" : "asm +v", s / grid / (iters * 16.0));
        }
    }
    return 0;
}
