// Microbenchmark of the stack kernels' interpreted MAC walk: 18 statically unrolled accumulator slots, each with a run-time MAC count read from a
// program register; no memory operations.  How much does a skipped slot (taken forward branch over its body) cost, against an executed MAC?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NS = 18;
struct Acc { f32x4 c[4]; };
__device__ __forceinline__ void mac(Acc& a, const bf16x8 (&w)[8], const bf16x8 (&x)[8]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(a.c[j]) : "v"(w[t + 4 * (j & 1)]), "v"(x[t + 4 * (j >> 1)]));
}
__global__ __launch_bounds__(256, 1) void k(const bf16x8* in, const int* prog, float* out, long long* cyc, int nseg, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 w[8], x[8];
    for (int i = 0; i < 8; ++i) { w[i] = in[threadIdx.x + 256 * i]; x[i] = in[threadIdx.x + 256 * (8 + i)]; }
    Acc acc[NS];
    for (int u = 0; u < NS; ++u) for (int j = 0; j < 4; ++j) acc[u].c[j] = f32x4{0, 0, 0, 0};
    const int c0r = prog[lane], c1r = prog[64 + lane];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const int c0 = __builtin_amdgcn_readlane(c0r, sgi), c1 = __builtin_amdgcn_readlane(c1r, sgi);
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int cnt = ((u < 10 ? c0 : c1) >> (3 * (u % 10))) & 7;
                for (int kk = 0; kk < cnt; ++kk) mac(acc[u], w, x);
            }
        }
    const long long t1 = clock64();
    float s = 0;
    for (int u = 0; u < NS; ++u) { asm volatile("s_nop 15" : "+a"(acc[u].c[0]), "+a"(acc[u].c[1]), "+a"(acc[u].c[2]), "+a"(acc[u].c[3])); for (int j = 0; j < 4; ++j) s += acc[u].c[j][0]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    bf16x8* in; float* out; long long* cyc; int* prog;
    (void)hipMalloc(&in, 256 * 16 * 16); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&prog, 128 * 4);
    std::vector<unsigned short> h(256 * 16 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (i * 7919 % 251);
    (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    // patterns: per segment a list of (slot, count)
    struct Pat { const char* name; std::vector<std::vector<std::pair<int, int>>> segs; };
    std::vector<Pat> pats;
    { Pat p{"dense: 4 segments x 18 slots x 1 MAC", {}}; for (int s = 0; s < 4; ++s) { std::vector<std::pair<int, int>> v; for (int u = 0; u < 18; ++u) v.push_back({u, 1}); p.segs.push_back(v); } pats.push_back(p); }
    { Pat p{"A1-C2 layer 0 (11 segments, 52 MACs)", {}};
      p.segs = {{{0,1},{1,1}}, {{0,1},{1,1}}, {{0,1},{1,1}}, {{0,1},{1,1}},
                {{2,1},{3,1},{4,1},{5,1},{6,1},{7,1},{8,1},{9,1},{10,1},{11,1},{12,1},{13,1}}, {{2,1},{8,1}}, {{5,1},{11,1}},
                {{2,1},{3,2},{4,1},{5,1},{6,2},{7,1},{8,1},{9,2},{10,1},{11,1},{12,2},{13,1}}, {{4,1},{7,1},{10,1},{13,1}},
                {{14,1},{15,1},{16,1},{17,1}}, {{14,1},{15,1},{16,1},{17,1}}};
      pats.push_back(p); }
    { Pat p{"sparse: 16 segments x 1 slot x 1 MAC", {}}; for (int s = 0; s < 16; ++s) p.segs.push_back({{(s * 5) % 18, 1}}); pats.push_back(p); }
    { Pat p{"one slot, 16 MACs in 8 segments x 2", {}}; for (int s = 0; s < 8; ++s) p.segs.push_back({{3, 2}}); pats.push_back(p); }
    const int iters = 200;
    for (auto& p : pats) {
        std::vector<int> pr(128, 0); int macs = 0;
        for (size_t s = 0; s < p.segs.size(); ++s) for (auto& e : p.segs[s]) { pr[(e.first / 10) * 64 + s] |= e.second << (3 * (e.first % 10)); macs += e.second; }
        (void)hipMemcpy(prog, pr.data(), 512, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, in, prog, out, cyc, (int)p.segs.size(), iters); (void)hipDeviceSynchronize(); }
        std::vector<long long> c(256); (void)hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += v;
        const double per_it = s / 256 / iters;
        printf("%-45s %8.0f cycles per pass, %6.1f per MAC (ideal 256), %d MACs, %zu segments, %zu slot headers\n", p.name, per_it, per_it / macs, macs, p.segs.size(), p.segs.size() * 18);
    }
    return 0;
}
