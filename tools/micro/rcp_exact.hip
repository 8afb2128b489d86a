// Exhaustive check: for which binary32 x does a short reciprocal sequence (hardware estimate plus
// Newton steps in fused multiply-adds) return exactly what the IEEE division 1.0f / x returns?
// Prints, per candidate, the number of mismatching x per binary exponent, so a safe range of
// exponents can be read off.  Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float cand(int which, float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    if (which == 0) return r;
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}

// bad[which][exponent] = number of x with that biased exponent (sign folded) whose candidate differs
__global__ __launch_bounds__(256) void k_check(unsigned long long* bad, uint32_t* example)
{
    const uint64_t total = 1ull << 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        const uint32_t ex = (bits >> 23) & 255u;
        for (int which = 0; which < 2; ++which) {
            const float got = cand(which, x);
            const bool same = __float_as_uint(got) == __float_as_uint(want) || (want != want && got != got);
            if (!same) {
                atomicAdd(&bad[which * 256 + ex], 1ull);
                if (ex > 40 && ex < 200) atomicMax(&example[which], bits & 0x7FFFFFFFu);
            }
        }
    }
}

int main()
{
    unsigned long long* bad; uint32_t* example;
    CK(hipMalloc(&bad, 512 * 8)); CK(hipMalloc(&example, 8));
    CK(hipMemset(bad, 0, 512 * 8)); CK(hipMemset(example, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(256 * 64), dim3(256), 0, 0, bad, example);
    CK(hipDeviceSynchronize());
    unsigned long long h[512]; uint32_t ex[2];
    CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(ex, example, sizeof(ex), hipMemcpyDeviceToHost));
    for (int which = 0; which < 2; ++which) {
        unsigned long long all = 0;
        int lo = -1, hi = -1;
        for (int e = 0; e < 256; ++e) { all += h[which * 256 + e]; }
        printf("candidate %d (%d Newton step%s): %llu mismatches of 2^32; exponents with any:", which, which + 1, which ? "s" : "", all);
        for (int e = 0; e < 256; ++e) if (h[which * 256 + e]) { if (lo < 0) lo = e; hi = e; printf(" %d:%llu", e, h[which * 256 + e]); }
        printf("\n  largest mismatching |x| with exponent in (40,200): 0x%08x\n", ex[which]);
        (void)lo; (void)hi;
    }
    CK(hipFree(bad)); CK(hipFree(example));
    return 0;
}
