// Microbenchmark: HBM read rate of the encoder's access pattern (64 rows x 256-byte chunks at a 14.4 KB row stride, K chunks in
// sequence with a barrier between them) against other ways of sweeping the same 118 MB input.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: encoder pattern: WG = (node, tile of 64 windows); per step 64 rows x 256 B; nsteps chunks, barrier in between
// mode 1: same but every load issued up front (no barriers)
// mode 2: WG = 16 windows, wave-level loads of 1 KB contiguous per row (row = window's whole joint block, 10.8 KB), linear sweep
template <int MODE> __global__ __launch_bounds__(256) void k_read(const char* x, unsigned* out, int B, int nodes, size_t win_pitch, int row_bytes, int tiles) {
    const int tid = threadIdx.x;
    u32x4 acc = u32x4{0, 0, 0, 0};
    if constexpr (MODE == 0 || MODE == 1) {
        const int node = blockIdx.x / tiles, tile = blockIdx.x % tiles;
        const int c = tid & 15, r0 = tid >> 4;
        const int nsteps = (row_bytes + 255) / 256;
        __shared__ u32x4 lds[1024];
        for (int s = 0; s < nsteps; ++s) {
            u32x4 v[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int w = min(tile * 64 + m * 16 + r0, B - 1);
                int off = s * 256 + c * 16; if (off + 16 > row_bytes) off = s * 256;
                v[m] = *reinterpret_cast<const u32x4*>(x + (size_t)w * win_pitch + (size_t)node * row_bytes + off);
            }
            if (MODE == 0) __syncthreads();
#pragma unroll
            for (int m = 0; m < 4; ++m) { if (MODE == 0) lds[m * 256 + tid] = v[m]; acc ^= v[m]; }
            if (MODE == 0) { __syncthreads(); acc ^= lds[(tid * 7 + s) & 1023]; }
        }
    } else {
        // linear: WG b sweeps a contiguous range of the whole array in 4 KB pieces (256 threads x 16 B)
        const size_t total = (size_t)B * win_pitch, per = (total / gridDim.x) & ~(size_t)4095;
        const char* p = x + (size_t)blockIdx.x * per;
        for (size_t o = 0; o + 4096 * 4 <= per; o += 4096 * 4) {
            u32x4 v[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const u32x4*>(p + o + m * 4096 + tid * 16);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc ^= v[m];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192, nodes = 12, row_bytes = 900;
    const size_t win_pitch = 14416;      // bytes per window (all types), 16-byte aligned
    char* x; unsigned* out;
    CHK(hipMalloc(&x, (size_t)B * win_pitch + 4096)); CHK(hipMalloc(&out, 1 << 20));
    CHK(hipMemset(x, 1, (size_t)B * win_pitch + 4096));
    char* flush; CHK(hipMalloc(&flush, 1ull << 30));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int tiles = B / 64;
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f, sum = 0.f; int n = 0;
        for (int it = 0; it < 12; ++it) {
            CHK(hipMemsetAsync(flush, it, 1ull << 30));      // evict the input from L2 / Infinity Cache
            CHK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(nodes * tiles), dim3(256), 0, 0, x, out, B, nodes, win_pitch, row_bytes, tiles);
            if (mode == 1) hipLaunchKernelGGL(k_read<1>, dim3(nodes * tiles), dim3(256), 0, 0, x, out, B, nodes, win_pitch, row_bytes, tiles);
            if (mode == 2) hipLaunchKernelGGL(k_read<2>, dim3(B / 4), dim3(256), 0, 0, x, out, B, nodes, win_pitch, row_bytes, tiles);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) { best = ms < best ? ms : best; sum += ms; ++n; }
        }
        const double bytes = mode == 2 ? (double)B * win_pitch : (double)B * nodes * row_bytes;
        printf("mode %d: avg %.1f us  best %.1f us  %.2f TB/s (bytes %.1f MB)\n", mode, sum / n * 1e3, best * 1e3, bytes / (sum / n * 1e-3) / 1e12, bytes / 1e6);
    }
    return 0;
}
