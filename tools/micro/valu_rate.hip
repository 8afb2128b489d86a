// Vector-ALU issue rate of one SIMD as a function of the waves resident on it.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o build/valu_rate && build/valu_rate
// Every wave runs the same chain-free stream of v_fma_f32 (8 independent accumulators); a grid of
// 256 CUs x 4 SIMDs x W waves is timed for W = 1, 2, 4, 5, 8.  Prints wave-instructions per SIMD per cycle
// (clock taken from the device's reported rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int INT>
__global__ __launch_bounds__(64) void k_fma(float* out, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const float m = 1.0001f, c = 0.5f;
    unsigned i0 = threadIdx.x, i1 = threadIdx.x + 1, i2 = 3, i3 = 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (INT) {
                i0 = i0 + i1; i1 = i1 ^ i2; i2 = i2 + i3; i3 = i3 - i0;
                i0 = i0 + i1; i1 = i1 ^ i2; i2 = i2 + i3; i3 = i3 - i0;
            } else {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(i0 + i1 + i2 + i3) == 12345.678f) out[0] = 1.0f;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clock_hz = prop.clockRate * 1e3;
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, clock_hz / 1e6);
    for (int kind = 0; kind < 2; ++kind) {
        for (int w : {1, 2, 4, 5, 8}) {
            const int blocks = cus * 4 * w; /* one-wave blocks: the dispatcher spreads them over the SIMDs */
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
                else hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_wave = (double)iters * 8 * 8;
            const double per_simd = insts_per_wave * w; /* wave-instructions one SIMD executed */
            const double cycles = ms * 1e-3 * clock_hz;
            printf("%s waves/SIMD %d: %.3f ms, %.3f wave-instructions per SIMD per cycle (%.2f cycles each)\n",
                   kind ? "v_add/xor_u32" : "v_fma_f32    ", w, ms, per_simd / cycles, cycles / per_simd);
        }
    }
    return 0;
}
