// Micro-benchmark: throughput of 64-bit global atomicMin without a result, as a visibility-buffer
// rasteriser would issue it: N lanes, each hitting `per` pixels near a pseudo-random screen position.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_atomics(unsigned long long* keys, uint32_t w, uint32_t h, uint32_t n, int per, int coherent)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x, y;
    if (coherent) { // neighbouring lanes land on neighbouring pixels, like a regular mesh
        uint32_t cols = w / 2; x = (i % cols) * 2; y = (i / cols) * 2 % h;
    } else {
        uint32_t hsh = i * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        x = hsh % (w - 2); y = (hsh >> 12) % (h - 2);
    }
    unsigned long long key = ((unsigned long long)(i & 0xFFFFFF) << 40) | i;
    for (int k = 0; k < per; ++k) atomicMin(&keys[(size_t)(y + (k >> 1)) * w + x + (k & 1)], key);
}

__global__ void k_stores(unsigned long long* keys, uint32_t w, uint32_t h, uint32_t n, int per)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t cols = w / 2, x = (i % cols) * 2, y = (i / cols) * 2 % h;
    unsigned long long key = ((unsigned long long)(i & 0xFFFFFF) << 40) | i;
    for (int k = 0; k < per; ++k) keys[(size_t)(y + (k >> 1)) * w + x + (k & 1)] = key;
}

int main()
{
    const uint32_t w = 1920, h = 1080, n = 1000000;
    unsigned long long* keys;
    CK(hipMalloc(&keys, (size_t)w * h * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int coherent = 0; coherent < 2; ++coherent) for (int per = 1; per <= 4; per *= 2) {
        CK(hipMemset(keys, 0xFF, (size_t)w * h * 8));
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_atomics, dim3((n + 255) / 256), dim3(256), 0, 0, keys, w, h, n, per, coherent);
        CK(hipEventRecord(a));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_atomics, dim3((n + 255) / 256), dim3(256), 0, 0, keys, w, h, n, per, coherent);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("atomicMin64 coherent=%d per_lane=%d: %.2f us per launch, %.1f G atomics/s\n", coherent, per, ms / 20 * 1e3, (double)n * per / (ms / 20 * 1e-3) / 1e9);
    }
    for (int per = 1; per <= 4; per *= 2) {
        CK(hipEventRecord(a));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_stores, dim3((n + 255) / 256), dim3(256), 0, 0, keys, w, h, n, per);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("plain store per_lane=%d: %.2f us per launch\n", per, ms / 20 * 1e3);
    }
    CK(hipFree(keys));
    return 0;
}
