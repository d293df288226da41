// Does gfx950 serve 16-byte global loads at 2-byte aligned addresses (bf16 series gathered at arbitrary window starts), and at what rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* src, unsigned* out, int shift, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = u32x4{0, 0, 0, 0};
    for (int r = 0; r < n; ++r) {
        const unsigned short* p = src + ((i * 8 + (size_t)r * 8 * 256 * gridDim.x) % (64u << 20)) + shift;
        u32x4 v;
        __builtin_memcpy(&v, p, 16);            // the compiler may split this; the asm below forces one dwordx4
        asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        acc ^= v;
    }
    if (acc[0] == 0x12345u) out[i] = acc[1];
    if (i < 4 && blockIdx.x == 0) { out[i] = acc[i]; }
}
int main() {
    unsigned short* src; unsigned* out;
    hipMalloc(&src, (64u << 20) * 2 + 64); hipMalloc(&out, 1 << 24);
    unsigned short* h = (unsigned short*)malloc((64u << 20) * 2 + 64);
    for (size_t i = 0; i < (64u << 20) + 32; ++i) h[i] = (unsigned short)(i * 7 + 1);
    hipMemcpy(src, h, (64u << 20) * 2 + 64, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, src, out, shift, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, src, out, shift, 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned o[4]; hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
        hipError_t err = hipGetLastError();
        printf("shift %d elements: %s, %.1f us for 268 MB of 16-byte loads (%.2f TB/s)  sample %08x\n", shift, hipGetErrorString(err), ms * 1e3, 4096.0 * 256 * 16 * 16 / (ms * 1e-3) / 1e12, o[0]);
    }
    return 0;
}
