// How many 256-thread workgroups does the chip keep resident, by registers per lane, scratch and LDS?
// Every workgroup adds itself to a counter, holds its slot for a fixed time, notes the largest count it saw and leaves.
//   hipcc -O3 --offload-arch=gfx950 -o build/residency tools/micro/residency.hip && build/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VGPRS, bool SCRATCH, int LDS_BYTES>
__global__ __launch_bounds__(256) void hold(unsigned* live, unsigned* peak, unsigned long long ticks, unsigned* sink)
{
    extern __shared__ unsigned char dyn[];
    __shared__ unsigned char lds[LDS_BYTES];
    if (threadIdx.x == 0) lds[0] = 1;
    // claim registers up to v(VGPRS-1)
    if constexpr (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if constexpr (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (VGPRS == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (VGPRS == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    volatile unsigned spill[SCRATCH ? 17 : 1];
    if constexpr (SCRATCH) { for (int i = 0; i < 17; ++i) spill[i] = threadIdx.x + i; }
    // one sample in the middle of the hold: generations of workgroups change at multiples of the hold time, when a
    // leaving workgroup's decrement may still be in flight while its successor has already counted itself
    unsigned seen = 0;
    if (threadIdx.x == 0) atomicAdd(live, 1u);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool sampled = false;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (!sampled && __builtin_amdgcn_s_memrealtime() - t0 >= ticks / 2) {
            sampled = true;
            if (threadIdx.x == 0) seen = __hip_atomic_load(live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (threadIdx.x == 0) { atomicMax(peak, seen); atomicSub(live, 1u); }
    if constexpr (SCRATCH) { if (spill[threadIdx.x % 17] == 0xFFFFFFFFu) sink[0] = lds[0]; }
    if (lds[0] == 77 && dyn[0] == 3) sink[1] = 1;
}

template <int VGPRS, bool SCRATCH, int LDS_BYTES>
int run(const char* what, unsigned* d, int grid)
{
    CHECK(hipMemset(d, 0, 16));
    hipLaunchKernelGGL((hold<VGPRS, SCRATCH, LDS_BYTES>), dim3(grid), dim3(256), 0, 0, d, d + 1, 2000ull /* 20 us */, d + 2);
    CHECK(hipDeviceSynchronize());
    unsigned h[4];
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("%-44s peak resident workgroups %u (%.2f per CU)\n", what, h[1], h[1] / 256.0);
    return 0;
}

int main()
{
    unsigned* d;
    CHECK(hipMalloc(&d, 16));
    const int grid = 8192;
    run<64, false, 1024>("64 VGPR, no scratch, 1 KB LDS", d, grid);
    run<80, false, 1024>("80 VGPR, no scratch, 1 KB LDS", d, grid);
    run<96, false, 1024>("96 VGPR, no scratch, 1 KB LDS", d, grid);
    run<96, true, 1024>("96 VGPR, 68 B scratch, 1 KB LDS", d, grid);
    run<96, false, 13072>("96 VGPR, no scratch, 13 KB LDS", d, grid);
    run<96, true, 13072>("96 VGPR, scratch, 13 KB LDS", d, grid);
    run<96, false, 32544>("96 VGPR, no scratch, 32.5 KB LDS", d, grid);
    run<128, false, 13072>("128 VGPR, no scratch, 13 KB LDS", d, grid);
    return 0;
}
