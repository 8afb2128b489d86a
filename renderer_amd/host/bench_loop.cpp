/*
 * bench_loop.cpp -- the benchmark's timed loop as compiled host code.
 *
 * One frame of the reference's main loop is "fill the 320-byte uniform block, submit the command buffer
 * recorded for the next swapchain image" (base_obj_viewer_app.cpp:126-138, vk_app.cpp:433-467).  This file
 * runs exactly that loop against the C ABI, so the time it reports holds no interpreter work: the caller
 * (any language) creates the resources and records the command buffers through the same ABI, then hands the
 * handles over.  The three entry points are passed as function pointers so that the loop drives whichever
 * build of the library the caller has loaded.
 */
#include <chrono>
#include <cstdint>

extern "C" {

typedef int (*buffer_write_fn)(void* buf, uint64_t offset, const void* src, uint64_t bytes);
typedef int (*submit_fn)(void* dev, void* cmd, void* fence);
typedef int (*wait_idle_fn)(void* dev);

/* Frames first .. first+steps-1: frame i writes uniform block i % n_blocks (block_floats floats each) and submits
 * command buffer (first_image + i - first) % n_cmds.  Waits for the device at the end.  seconds_out[0]: the
 * whole loop including that wait; seconds_out[1]: the part spent queueing (before the wait).
 * Returns the first non-zero result of the library, or 0. */
int rnd_bench_frames(buffer_write_fn write, submit_fn submit, wait_idle_fn wait_idle, void* dev, void* const* cmds, int n_cmds,
                     int first_image, void* ubo, const float* blocks, int n_blocks, int block_floats, int first, int steps,
                     double* seconds_out)
{
    if (!write || !submit || !wait_idle || !dev || !cmds || n_cmds <= 0 || !ubo || !blocks || n_blocks <= 0 || steps < 0) return -1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) {
        const float* block = blocks + (size_t)((first + i) % n_blocks) * (size_t)block_floats;
        if (int rc = write(ubo, 0, block, (uint64_t)block_floats * 4u)) return rc;
        if (int rc = submit(dev, cmds[(first_image + i) % n_cmds], nullptr)) return rc;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (int rc = wait_idle(dev)) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    if (seconds_out) {
        seconds_out[0] = std::chrono::duration<double>(t2 - t0).count();
        seconds_out[1] = std::chrono::duration<double>(t1 - t0).count();
    }
    return 0;
}

} // extern "C"
