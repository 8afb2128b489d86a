// Where global_load_lds_dwordx3 puts a lane's twelve bytes: 64 lanes each copy 12 bytes from their own 16-byte record into one LDS
// piece; the piece is then dumped.  Expected (and what the tile kernel's corner staging assumes): lane l's bytes at piece + 12 l.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out, int mask_odd)
{
    __shared__ unsigned lds[64 * 4 + 64];
    for (int i = threadIdx.x; i < 64 * 4 + 64; i += 64) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
    const unsigned piece = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds[16];
    const unsigned* p = src + threadIdx.x * 4;
    unsigned keep;
    if (!mask_odd || !(threadIdx.x & 1))
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p), "s"(__builtin_amdgcn_readfirstlane(piece)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 + 64; i += 64) out[i] = lds[i];
}
int main()
{
    unsigned *src, *out;
    hipMalloc(&src, 64 * 16); hipMalloc(&out, (64 * 4 + 64) * 4);
    std::vector<unsigned> h(64 * 4), o(64 * 4 + 64);
    for (int l = 0; l < 64; ++l) for (int w = 0; w < 4; ++w) h[l * 4 + w] = (l << 8) | w;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mask = 0; mask < 2; ++mask) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, mask);
        if (hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int w = 0; w < 3; ++w) {
            const unsigned got = o[16 + l * 3 + w];
            const unsigned want = (mask && (l & 1)) ? 0xDEAD0000u + 16 + l * 3 + w : (unsigned)((l << 8) | w);
            if (got != want) { if (bad < 8) printf("mask %d lane %d word %d: got %08x want %08x\n", mask, l, w, got, want); ++bad; }
        }
        printf("every lane active: %s; lane l's 12 bytes at piece + 12 l: %s (%d words differ); words before / after the piece untouched: %s\n",
               mask ? "no (odd lanes off)" : "yes", bad ? "NO" : "yes", bad,
               (o[15] == 0xDEAD0000u + 15 && o[16 + 192] == 0xDEAD0000u + 16 + 192) ? "yes" : "NO");
        if (bad) { for (int i = 12; i < 40; ++i) printf("%08x ", o[i]); printf("\n"); }
    }
    return 0;
}
