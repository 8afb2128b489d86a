#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <typename Op>
__device__ __forceinline__ int wave_inclusive_scan(int v, int identity, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));
    return v;
}
__global__ void k(const int* in, int* out_add, int* out_max)
{
    const int v = in[threadIdx.x];
    out_add[threadIdx.x] = wave_inclusive_scan(v, 0, [](int a, int b) { return a + b; });
    out_max[threadIdx.x] = wave_inclusive_scan(v, -1, [](int a, int b) { return a > b ? a : b; });
}
int main()
{
    int *in, *oa, *om;
    hipMalloc(&in, 256); hipMalloc(&oa, 256); hipMalloc(&om, 256);
    int bad = 0;
    for (int trial = 0; trial < 50; ++trial) {
        std::vector<int> h(64), a(64), m(64);
        for (int i = 0; i < 64; ++i) h[i] = (rand() % 3 == 0) ? -1 : rand() % 64;
        if (trial == 0) for (int i = 0; i < 64; ++i) h[i] = 1;
        hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, oa, om);
        hipMemcpy(a.data(), oa, 256, hipMemcpyDeviceToHost); hipMemcpy(m.data(), om, 256, hipMemcpyDeviceToHost);
        int sa = 0, sm = -1;
        for (int i = 0; i < 64; ++i) {
            sa += h[i]; sm = h[i] > sm ? h[i] : sm;
            if (a[i] != sa || m[i] != sm) { if (bad < 10) printf("trial %d lane %d: add %d want %d, max %d want %d\n", trial, i, a[i], sa, m[i], sm); ++bad; }
        }
    }
    printf("mismatches %d\n", bad);
    return bad != 0;
}
