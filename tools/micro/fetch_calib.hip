// What the L2's memory-side read counter (rocprofv3 FETCH_SIZE) tallies per byte actually moved, for the access shapes
// the frame's kernels use -- the shapes for which the factor of 2 measured on 16-byte-per-lane streaming is not known
// to hold.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o build/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -- build/fetch_calib         (tools/fetch_calib.sh runs it and folds the result)
//
// Every "dense" kernel reads each byte of a 1 GiB buffer exactly once (four times the 256 MiB of on-die cache, so a
// line's second fetch cannot be served on-die), and differs only in how a wave's lanes cover the lines:
//   dense_stream16   lane i reads 16 bytes at 16 i: a wave instruction covers 8 whole 128-byte lines
//   dense_stride4    lane i reads 4 bytes at 128 i + 4 k, k = 0..31: every instruction touches 64 lines, 4 bytes of each;
//                    the other 124 bytes of a line are read by the same lane in the following iterations (L1 / L2 hits)
//   dense_stride16   lane i reads 16 bytes at 128 i + 16 k, k = 0..7
//   dense_triple12   lane i reads 12 bytes (three words, as an index triple) at 12 * (64 * (3 * w) + i) ... the wave's 768
//                    bytes are contiguous: the shape of the setup kernel's index fetch
// The true traffic of each is the buffer's size; FETCH_SIZE / size is the factor for that shape.
// The "sparse" kernels read n = 2^22 records at pseudo-random places of the buffer (one record per 256 bytes on
// average, no line touched twice): what one isolated miss is tallied as.
//   sparse4 / sparse16 / sparse12    4-, 16- and 12-byte records (the 12-byte ones at multiples of 12: some straddle a line)
// Each kernel also reports its time: n isolated misses per second times the bytes of a miss cannot exceed what the
// memory delivers, which bounds the size of a miss from above.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dense_stream16(const U32x4* buf, size_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const U32x4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

/* thread t owns line t (128 bytes) of its workgroup's 32 KiB block and reads it 4 bytes at a time */
__global__ __launch_bounds__(256) void dense_stride4(const uint32_t* buf, size_t n_lines, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t line = (size_t)blockIdx.x * blockDim.x + threadIdx.x; line < n_lines; line += (size_t)gridDim.x * blockDim.x) {
        const uint32_t* p = buf + line * 32;
#pragma unroll 1
        for (int k = 0; k < 32; ++k) acc ^= p[k] * (uint32_t)(k + 1);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void dense_stride16(const U32x4* buf, size_t n_lines, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t line = (size_t)blockIdx.x * blockDim.x + threadIdx.x; line < n_lines; line += (size_t)gridDim.x * blockDim.x) {
        const U32x4* p = buf + line * 8;
#pragma unroll 1
        for (int k = 0; k < 8; ++k) { const U32x4 v = p[k]; acc ^= (v.x ^ v.y ^ v.z ^ v.w) * (uint32_t)(k + 1); }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

/* consecutive lanes read consecutive 12-byte triples */
__global__ __launch_bounds__(256) void dense_triple12(const uint32_t* buf, size_t n_triples, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_triples; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t* p = buf + i * 3;
        acc ^= p[0] ^ (p[1] * 3u) ^ (p[2] * 5u);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

/* record r of n lies in the r-th 256-byte cell of a permutation of the cells (an odd multiplier modulo a power of
 * two), at a pseudo-random offset inside the cell that keeps the record within it */
template <int BYTES>
__global__ __launch_bounds__(256) void sparse(const uint8_t* buf, uint32_t n_log2, uint32_t* sink)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = 1u << n_log2;
    if (r >= n) return;
    const uint32_t cell = (r * 2654435761u + 12345u) & (n - 1u);
    const uint32_t h = (r ^ (r >> 7)) * 0x9E3779B1u;
    const uint32_t step = BYTES == 12 ? 12u : (uint32_t)BYTES;
    const uint32_t slots = (256u - (uint32_t)BYTES) / step + 1u;
    const uint32_t off = ((h >> 8) % slots) * step;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(buf + (size_t)cell * 256u + off);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < BYTES / 4; ++k) acc ^= p[k] * (uint32_t)(k + 1);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = size_t(1) << 30;
    uint8_t* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const uint32_t n_log2 = 22;
    auto flush = [&]() { return hipMemset(buf, 1, bytes); }; /* a write pass over the buffer: nothing of it stays cached for the next read */
    struct Run { const char* name; double known_bytes; };
    float ms = 0;
#define TIMED(name, known, ...) do { CHECK(flush()); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); __VA_ARGS__; CHECK(hipEventRecord(e1)); \
        CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("%-16s known_bytes %.0f  ms %.4f  GB/s %.1f\n", name, (double)(known), ms, (double)(known) / ms * 1e-6); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        TIMED("dense_stream16", bytes, hipLaunchKernelGGL(dense_stream16, dim3(grid), dim3(256), 0, 0, (const U32x4*)buf, bytes / 16, sink));
        /* 64 workgroups: 16 384 lines = 2 MiB in use at any moment, a quarter MiB per XCD, so a line stays in the L2 until its owner has read all of it */
        TIMED("dense_stride4", bytes, hipLaunchKernelGGL(dense_stride4, dim3(64), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 128, sink));
        TIMED("dense_stride16", bytes, hipLaunchKernelGGL(dense_stride16, dim3(64), dim3(256), 0, 0, (const U32x4*)buf, bytes / 128, sink));
        TIMED("dense_triple12", (bytes / 12) * 12, hipLaunchKernelGGL(dense_triple12, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 12, sink));
        const uint32_t n = 1u << n_log2;
        TIMED("sparse4", (double)n * 4, hipLaunchKernelGGL(sparse<4>, dim3(n / 256), dim3(256), 0, 0, (const uint8_t*)buf, n_log2, sink));
        TIMED("sparse16", (double)n * 16, hipLaunchKernelGGL(sparse<16>, dim3(n / 256), dim3(256), 0, 0, (const uint8_t*)buf, n_log2, sink));
        TIMED("sparse12", (double)n * 12, hipLaunchKernelGGL(sparse<12>, dim3(n / 256), dim3(256), 0, 0, (const uint8_t*)buf, n_log2, sink));
    }
    printf("records per sparse kernel: %u\n", 1u << n_log2);
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
