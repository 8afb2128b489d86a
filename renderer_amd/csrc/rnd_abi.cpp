/*
 * rnd_abi.cpp -- host side of the C ABI declared in include/renderer_amd.h.
 *
 * Owns device memory, records command buffers, and turns a recorded pass into
 * the kernel sequence of rnd_kernels.hip.  There is no CPU rendering path: if a
 * HIP call fails the entry point fails.
 */
#include "../../include/renderer_amd.h"
#include "rnd_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <map>
#include <tuple>
#include <string>
#include <utility>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

using namespace rnd;

namespace {

thread_local std::string g_last_error;
thread_local std::string g_sync_report; /* what a wait with a deadline found (drain_stream); added to the error of the HIP call that failed */

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? RND_ERROR_OUT_OF_DEVICE_MEMORY : RND_ERROR_DEVICE_LOST, \
                        std::string(#expr) + ": " + hipGetErrorString(e_) + (g_sync_report.empty() ? "" : " -- " + g_sync_report)); \
    } while (0)

double srgb_to_linear(double s) { return s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4); }

/* Host memory to device memory, finished on return.  On a named stream and waited for there: what the null stream's
 * hipMemcpy has done by the time it returns is less plainly said, and a command buffer's non-blocking stream does not
 * wait for the null stream. */
hipError_t upload_now(void* dst, const void* src, size_t bytes, hipStream_t s)
{
    if (!bytes) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}

template <typename T>
struct DeviceArray {
    T* ptr = nullptr;
    size_t count = 0;
    hipError_t resize(size_t n)
    {
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; }
        count = n;
        if (!n) return hipSuccess;
        return hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
    }
    hipError_t upload(const std::vector<T>& v, hipStream_t s)
    {
        hipError_t e = resize(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return upload_now(ptr, v.data(), v.size() * sizeof(T), s);
    }
    void release() { if (ptr) { (void)hipFree(ptr); ptr = nullptr; } count = 0; }
};

/* Zeroes device memory and returns when it is zero.  hipMemset is ordered on the null stream and may return before
 * the device has written; a command buffer's stream is non-blocking and does not wait for the null stream, so a kernel
 * launched on it soon afterwards could still read what the allocation held before (a freed image's words). */
hipError_t zero_now(void* p, size_t bytes, hipStream_t s)
{
    if (!bytes) return hipSuccess;
    const hipError_t e = hipMemsetAsync(p, 0, bytes, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}

/* Cross-stream order is an event recorded on the producer's tail and waited for by the consumer.  Every such edge takes
 * the next event of a small ring kept by the producer, so an event is recorded again only after EDGES other edges of
 * the same producer have been set up: a wait that is still queued never has its event recorded anew underneath it. */
struct EventRing {
    static constexpr unsigned EDGES = 16;
    hipEvent_t ev[EDGES] = {};
    unsigned next = 0;
    hipError_t create()
    {
        for (unsigned i = 0; i < EDGES; ++i) {
            const hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    void destroy() { for (unsigned i = 0; i < EDGES; ++i) if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; } }
    hipEvent_t take() { return ev[next++ % EDGES]; }
};

/* A stream command buffers execute on.  The device keeps a pool of them, created once and kept until the device is
 * destroyed; command buffers are dealt to the pool's streams (see rnd_cmd_create). */
struct CmdStream {
    hipStream_t stream = nullptr;
    EventRing done;            /* edges from this stream's tail to a stream that has to follow it */
    bool submitted = false;    /* work may still be running on the stream */
    uint64_t seen_seq = 0;     /* the device's upload_seq this stream has been ordered behind */
    int users = 0;             /* command buffers that execute on it */
    int index = 0;
    /* the last cross-stream wait queued on this stream, for the report of a wait that does not end */
    const char* last_wait = "none";
    uint64_t last_wait_seq = 0;
};

} // namespace

struct rnd_device_t {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cmd_stream_priority = 0; /* of the streams command buffers get, see rnd_device_create */
    uint64_t geometry_epoch = 0;   /* counts writes into vertex buffers: boxes computed before the last one are stale */
    ShadeTables* tables = nullptr; /* sRGB decode/encode tables in device memory */
    ShadeTables host_tables;       /* the same tables on the host */
    /* pinned ring for small host->device writes (the per-frame uniform block): the copy is
     * queued on the stream and the call returns without waiting for the device */
    static constexpr int STAGE_SLOTS = 64;
    static constexpr size_t STAGE_SLOT_BYTES = 4096;
    uint8_t* stage_host = nullptr;
    hipEvent_t stage_ev[STAGE_SLOTS] = {};
    bool stage_used[STAGE_SLOTS] = {};
    int stage_next = 0;
    uint32_t* status_host = nullptr; /* pinned word the tile kernel writes when a work list overflowed */
    uint32_t* status_dev = nullptr;  /* the same word as the device addresses it */
    uint8_t* scratch = nullptr;      /* this device's copy of the library's static scratch array (rnd_draw_indexed_noalloc) */
    /* The latest small write to a uniform buffer.  It never goes to memory by itself: every
     * submit from then on carries it in its kernel arguments and the vertex kernel reads the
     * buffer through it, so frames in flight each see the bytes that were current at their
     * submit and nothing on the GPU writes uniform memory.  It is stored to memory (flush_pending)
     * only when something else needs the buffer's bytes there. */
    UniformPatch pending = {};
    /* Frames in flight: every command buffer of a device that owns its stream executes on a
     * stream of its own, so consecutive frames overlap on the GPU like the reference's frames
     * in flight (vk_app.cpp:109-130).  What stays ordered across those streams:
     *  - work queued on the device's own stream (uploads, copies) precedes later frames (the device's ring of events)
     *    and follows the frames already submitted (their done events);
     *  - frames that render into the same target run one after the other (the target's event). */
    std::vector<rnd_cmd> cmds;
    std::vector<rnd_target> targets;
    /* The streams command buffers execute on: as many as the runtime has hardware queues in one priority pool
     * less one for the device's own stream (GPU_MAX_HW_QUEUES - 1, 3 unless the environment says otherwise;
     * RND_CMD_STREAMS overrides), made when the first
     * command buffers are created and kept for the device's lifetime.  Command buffers beyond that number share a
     * stream with an earlier one -- ordered by the stream itself -- instead of getting a stream that would share a
     * hardware queue with another stream behind the runtime's back, where one stream's barrier packets sit in front
     * of the other's work; and no stream or event is created or destroyed as command buffers come and go. */
    std::vector<CmdStream*> cmd_streams;
    int max_cmd_streams = 3;
    EventRing order;               /* edges from the device's own stream to command buffers' streams */
    const char* dev_last_wait = "none";
    /* the last cross-stream edges set up on this device (stream -1: the device's own), for the report of a wait that does not end */
    struct Edge { int waiter, producer; hipEvent_t ev; const char* why; uint64_t number; };
    static constexpr int EDGE_LOG = 48;
    Edge edges[EDGE_LOG] = {};
    uint64_t edge_count = 0;
    int debug_sync_ms = 0;         /* RND_DEBUG_SYNC=1: waits for the device poll every stream against a deadline and say which one is stuck */
    uint64_t upload_seq = 0;       /* counts the pieces of work queued on the device's own stream */
    int live_objects = 0;          /* buffers, textures, targets, pipelines, command buffers and fences not yet destroyed: each holds a
                                    * pointer to this object, so the device outlives them (rnd_device_destroy refuses while any is left) */
    uint64_t index_epoch = 0;      /* counts writes into index buffers: vertex counts scanned before the last one are stale */
    bool profiling = false;
    hipEvent_t ev[RND_STAGE_COUNT + 1] = {};
    float stage_ms[RND_STAGE_COUNT] = {};
    bool stage_valid = false;

    /* How many single-sample tile workgroups share a CU.  Five fill a CU's registers; with four (a tile workgroup made
     * to claim 2 KB more LDS) the vertex and setup waves of the next frame find room beside the tile kernel instead of
     * starting as tile waves retire.  Five is the setting: measured the same or faster on every benchmark workload at
     * one to four frames in flight, once each command buffer had a hardware queue of its own.  RND_TILE_WORKGROUPS=4
     * chooses four; both render the same bits. */
    bool four_tile_workgroups = false;
};

struct rnd_buffer_t {
    rnd_device dev;
    uint8_t* ptr;
    uint64_t bytes;
    uint32_t usage;
};

struct rnd_texture_t {
    rnd_device dev;
    uint32_t width, height, levels;
    uint32_t srgb;
    rnd_sampler_info sampler;
    uint8_t* level_ptr[16];
};

struct rnd_target_t {
    rnd_device dev;
    uint32_t width, height, samples;
    int32_t color_format;
    uint8_t* color;
    uint32_t* depth;
    EventRing order;           /* edges from last_stream's tail to another stream that renders into the target next */
    hipStream_t last_stream;   /* stream of the last frame rendered into, or copy taken from, this target */
    bool last_valid;
};

struct rnd_pipeline_t {
    rnd_device dev;
    rnd_pipeline_info info;
};

struct rnd_fence_t {
    rnd_device dev;
    hipEvent_t ev;
    bool pending;
};

namespace {

struct RecordedDraw {
    rnd_pipeline pipeline;
    rnd_buffer vb; uint64_t vb_offset;
    rnd_buffer ib; uint64_t ib_offset;
    rnd_buffer ubo; uint64_t ubo_offset;
    rnd_texture tex;
    float push[32];
    uint32_t index_count, first_index;
    int32_t vertex_offset;
};

enum CmdState { CMD_INITIAL, CMD_RECORDING, CMD_IN_PASS, CMD_EXECUTABLE };

} // namespace

struct rnd_cmd_t {
    rnd_device dev = nullptr;
    hipStream_t stream = nullptr; /* the device's stream, or one of the device's pool (cs) */
    CmdStream* cs = nullptr;      /* null: the command buffer executes on the device's own stream */
    CmdState state = CMD_INITIAL;
    /* pass */
    bool has_pass = false;
    rnd_target target = nullptr;
    float clear_color[4] = {0, 0, 0, 0};
    float clear_depth = 1.0f;
    int32_t area[4] = {0, 0, 0, 0};
    bool has_viewport = false, has_scissor = false;
    float viewport[6] = {0, 0, 0, 0, 0, 1};
    int32_t scissor[4] = {0, 0, 0, 0};
    /* bindings while recording */
    rnd_pipeline cur_pipeline = nullptr;
    rnd_buffer cur_vb = nullptr; uint64_t cur_vb_offset = 0;
    rnd_buffer cur_ib = nullptr; uint64_t cur_ib_offset = 0;
    rnd_buffer cur_ubo = nullptr; uint64_t cur_ubo_offset = 0;
    rnd_texture cur_tex = nullptr;
    float cur_push[32] = {};
    std::vector<RecordedDraw> draws;
    /* compiled */
    DeviceArray<DrawStatic> d_draws;
    DeviceArray<DrawDerived> d_derived;
    DeviceArray<Chunk> d_vchunks, d_tchunks, d_visible_vchunks, d_visible_tchunks;
    DeviceArray<TexDesc> d_textures;
    DeviceArray<SVtx> d_svtx;
    DeviceArray<SVary> d_svary;
    DeviceArray<PrimRec> d_prim_rec;
    DeviceArray<TileEntry> d_tile_lists;
    DeviceArray<LargeEntry> d_large_list;
    DeviceArray<ClipTri> d_clip_tris;
    DeviceArray<uint32_t> d_tile_count, d_counters;
    DeviceArray<unsigned long long> d_probe; /* RND_PROBES builds only */
    FrameParams fp = {};
    uint32_t shard_index = 0, shard_count = 1;
    uint64_t geometry_epoch = 0;   /* the device's when the draws' boxes were computed */
    uint64_t index_epoch = 0;      /* the device's when the draws' index ranges were scanned */
    uint64_t frames_submitted = 0; /* its parity picks the pair of visible-list lengths a frame counts into */

    void release_compiled()
    {
#ifdef RND_PROBES
        if (d_probe.ptr) {
            std::vector<unsigned long long> v(d_probe.count);
            if (hipMemcpy(v.data(), d_probe.ptr, v.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                double sum[16] = {};
                size_t rows = 0;
                for (size_t r = 0; r < v.size() / 16; ++r) {
                    if (!v[r * 16] || r >= fp.probe_kernel_row) continue;
                    ++rows;
                    for (int i = 0; i < 16; ++i) sum[i] += (double)v[r * 16 + i];
                }
                if (rows) {
                    fprintf(stderr, "tile kernel phases (cycles of wave 0, summed over frames, mean over %zu workgroups): ", rows);
                    for (int i = 0; i < 13; ++i) if (sum[i] > 0) fprintf(stderr, "%d:%.0f ", i, sum[i] / (double)rows);
                    /* the last frame: first start to last end, and the workgroups' own lifetimes */
                    unsigned long long t0 = ~0ull, t1 = 0; double life = 0;
                    std::vector<unsigned long long> starts;
                    for (size_t r = 0; r < v.size() / 16; ++r) {
                        if (!v[r * 16] || r >= fp.probe_kernel_row) continue;
                        t0 = std::min(t0, v[r * 16 + 14]); t1 = std::max(t1, v[r * 16 + 15]);
                        life += (double)(v[r * 16 + 15] - v[r * 16 + 14]);
                        starts.push_back(v[r * 16 + 14]);
                    }
                    if (const char* dump = getenv("RND_PROBE_DUMP")) {
                        /* appended: one block per command buffer, absolute 100 MHz ticks, so that the last frames of
                         * command buffers in flight together can be laid side by side */
                        if (FILE* f = fopen(dump, "a")) {
                            const size_t kr = (size_t)fp.probe_kernel_row * 16;
                            const size_t vb = (fp.n_vchunks + 3) / 4, sb = (fp.n_tchunks + 3) / 4;
                            unsigned long long vend = 0, send = 0;
                            for (size_t b = 0; b < vb && kr + 16 + b < v.size(); ++b) vend = std::max(vend, v[kr + 16 + b]);
                            for (size_t b = 0; b < sb + 8 * 64 && kr + 16 + vb + b < v.size(); ++b) send = std::max(send, v[kr + 16 + vb + b]);
                            fprintf(f, "# command buffer %p vertex %llu %llu setup %llu %llu\n", (void*)this, v[kr], vend, v[kr + 4], send);
                            for (size_t r = 0; r < v.size() / 16; ++r)
                                if (v[r * 16] && r < fp.probe_kernel_row) fprintf(f, "%zu %llu %llu %llu %llu\n", r, v[r * 16 + 14], v[r * 16 + 15], v[r * 16 + 13] >> 32, v[r * 16 + 13] & 0xFFFFFFFFull);
                            fclose(f);
                        }
                    }
                    std::sort(starts.begin(), starts.end());
                    fprintf(stderr, "\n  last frame (100 MHz ticks): span %llu, mean workgroup lifetime %.0f, workgroups resident on average %.0f; start of workgroup #1280 at +%llu, last start at +%llu\n",
                            t1 - t0, life / (double)rows, life / (double)(t1 - t0), starts.size() > 1280 ? starts[1280] - t0 : 0ull, starts.back() - t0);
                }
            }
        }
        d_probe.release();
#endif
        d_draws.release(); d_derived.release(); d_vchunks.release(); d_tchunks.release(); d_visible_vchunks.release(); d_visible_tchunks.release(); d_textures.release();
        d_svtx.release(); d_svary.release(); d_prim_rec.release(); d_tile_lists.release(); d_large_list.release(); d_clip_tris.release();
        d_tile_count.release(); d_counters.release();
    }
};

namespace {

int set_device(rnd_device dev)
{
    HIP_TRY(hipSetDevice(dev->ordinal));
    return RND_SUCCESS;
}

/* Waits for one stream.  With a deadline (RND_DEBUG_SYNC=1) the stream is polled instead of waited for, and a stream
 * that has not drained by then is named in the error with the last cross-stream wait queued on it, rather than the call
 * never returning. */
hipError_t drain_stream(rnd_device dev, hipStream_t s, const char* name, int index, const char* last_wait)
{
    if (dev->debug_sync_ms <= 0) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > dev->debug_sync_ms) {
            char msg[256];
            snprintf(msg, sizeof msg, "%s %d has not drained after %d ms; the last cross-stream wait queued on it: %s.  Streams now:", name, index,
                     dev->debug_sync_ms, last_wait);
            g_sync_report = msg;
            auto state = [](hipError_t q) { return q == hipSuccess ? "idle" : q == hipErrorNotReady ? "busy" : "error"; };
            g_sync_report += std::string(" device ") + state(hipStreamQuery(dev->stream));
            for (CmdStream* cs : dev->cmd_streams)
                g_sync_report += " cs" + std::to_string(cs->index) + " " + state(hipStreamQuery(cs->stream)) + (cs->submitted ? "(submitted)" : "");
            g_sync_report += ".  Edges (number waiter<-producer event-state why):";
            const uint64_t first = dev->edge_count > rnd_device_t::EDGE_LOG ? dev->edge_count - rnd_device_t::EDGE_LOG : 0;
            for (uint64_t k = first; k < dev->edge_count; ++k) {
                const rnd_device_t::Edge& ed = dev->edges[k % rnd_device_t::EDGE_LOG];
                snprintf(msg, sizeof msg, " [%llu %d<-%d %s %s]", (unsigned long long)ed.number, ed.waiter, ed.producer,
                         state(hipEventQuery(ed.ev)), ed.why);
                g_sync_report += msg;
            }
            return hipErrorNotReady;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

/* waits for everything queued on the device: its own stream and every command buffer's */
hipError_t sync_all(rnd_device dev)
{
    g_sync_report.clear();
    hipError_t e = drain_stream(dev, dev->stream, "the device's stream", 0, dev->dev_last_wait);
    for (CmdStream* cs : dev->cmd_streams) {
        if (e == hipErrorNotReady) break; /* a stuck stream has been named: the others are not waited for */
        const hipError_t e2 = drain_stream(dev, cs->stream, "command-buffer stream", cs->index, cs->last_wait);
        if (e == hipSuccess) e = e2;
    }
    if (e == hipSuccess) for (CmdStream* cs : dev->cmd_streams) { cs->submitted = false; cs->seen_seq = dev->upload_seq; }
    return e;
}

/* Cross-stream order is set up where it is needed, not on every submit: the producer's tail is
 * marked with an event of the producer's ring at that moment and the waiter waits for it. */
hipError_t wait_for_stream(rnd_device dev, hipStream_t waiter, int waiter_id, hipStream_t producer, int producer_id, EventRing& ring, const char* why)
{
    if (waiter == producer) return hipSuccess;
    const hipEvent_t ev = ring.take();
    rnd_device_t::Edge& log = dev->edges[dev->edge_count % rnd_device_t::EDGE_LOG];
    log = {waiter_id, producer_id, ev, why, dev->edge_count};
    ++dev->edge_count;
    const hipError_t e = hipEventRecord(ev, producer);
    return e != hipSuccess ? e : hipStreamWaitEvent(waiter, ev, 0);
}

/* the device's stream waits for every frame already submitted */
hipError_t follow_frames(rnd_device dev)
{
    for (CmdStream* cs : dev->cmd_streams) {
        if (!cs->submitted) continue;
        const hipError_t e = wait_for_stream(dev, dev->stream, -1, cs->stream, cs->index, cs->done, "follow_frames");
        if (e != hipSuccess) return e;
        dev->dev_last_wait = "the tail of a command-buffer stream (follow_frames)";
    }
    return hipSuccess;
}

/* stream-ordered small upload through the pinned ring: returns without waiting for the device */
int staged_write(rnd_device dev, uint8_t* dst, const void* src, size_t bytes)
{
    int slot = dev->stage_next;
    dev->stage_next = (slot + 1) % rnd_device_t::STAGE_SLOTS;
    if (dev->stage_used[slot]) HIP_TRY(hipEventSynchronize(dev->stage_ev[slot]));
    uint8_t* staging = dev->stage_host + (size_t)slot * rnd_device_t::STAGE_SLOT_BYTES;
    memcpy(staging, src, bytes);
    /* frames already queued read the old bytes: the copy follows them, on the GPU (the host does not wait) */
    HIP_TRY(follow_frames(dev));
    HIP_TRY(hipMemcpyAsync(dst, staging, bytes, hipMemcpyHostToDevice, dev->stream));
    HIP_TRY(hipEventRecord(dev->stage_ev[slot], dev->stream));
    dev->stage_used[slot] = true;
    ++dev->upload_seq;
    return RND_SUCCESS;
}

/* puts a held-back uniform write on the stream as an ordinary copy */
int flush_pending(rnd_device dev)
{
    if (!dev->pending.dst) return RND_SUCCESS;
    UniformPatch p = dev->pending;
    dev->pending.dst = nullptr;
    return staged_write(dev, p.dst, p.words, p.bytes);
}

int report_overflow(rnd_device dev)
{
    if (!dev->status_host) return RND_SUCCESS;
    /* the tile kernel writes the flags and, beside them, their complement, as one 64-bit word: read as one */
    const uint64_t pair = __atomic_load_n(reinterpret_cast<const uint64_t*>(dev->status_host), __ATOMIC_ACQUIRE);
    if (!pair) return RND_SUCCESS;
    const uint32_t ovf = (uint32_t)pair, check = (uint32_t)(pair >> 32);
    __atomic_store_n(reinterpret_cast<uint64_t*>(dev->status_host), (uint64_t)0, __ATOMIC_RELEASE);
    if (check == ~ovf && !(ovf & ~(OVF_LARGE | OVF_CLIP_TRIS)))
        return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, std::string("a work list overflowed while rendering (flags ") + std::to_string(ovf) +
                    "; pass-wide list=1, clipped triangles=4): the frame is incomplete");
    /* anything else is no report a kernel of this library makes: said as loudly, and with both words, so that whoever
     * meets it can tell a counter that was read before it was zeroed (the pair agrees) from a stray store (it does not) */
    char words[96];
    snprintf(words, sizeof words, "0x%08X / 0x%08X (%s)", ovf, check, check == ~ovf ? "a pair" : "not a pair");
    return fail(RND_ERROR_DEVICE_LOST, std::string("the status words hold ") + words + ", which is no overflow report: the frame may be incomplete");
}

/* a colour as the shading kernel stores it: linear -> sRGB code = number of thresholds not above
 * the value; plain UNORM = trunc(fma(c, 255, 0.5)) of the clamped value */
uint32_t host_pack_color(const ShadeTables& tb, const float* rgba, bool srgb, bool bgra)
{
    auto unorm = [](float c) -> uint32_t {
        if (!(c > 0.0f)) return 0u;
        c = fminf(c, 1.0f);
        return (uint32_t)fmaf(c, 255.0f, 0.5f);
    };
    auto encode = [&](float c) -> uint32_t {
        uint32_t k = 0;
        while (k < 255 && c >= tb.thr[k + 1]) ++k;
        return k;
    };
    const uint32_t r = srgb ? encode(rgba[0]) : unorm(rgba[0]), g = srgb ? encode(rgba[1]) : unorm(rgba[1]),
                   b = srgb ? encode(rgba[2]) : unorm(rgba[2]), a = unorm(rgba[3]);
    return bgra ? (b | (g << 8) | (r << 16) | (a << 24)) : (r | (g << 8) | (b << 16) | (a << 24));
}

bool format_is_srgb(int32_t f)
{
    return f == RND_FORMAT_R8_SRGB || f == RND_FORMAT_R8G8_SRGB || f == RND_FORMAT_R8G8B8_SRGB || f == RND_FORMAT_R8G8B8A8_SRGB ||
           f == RND_FORMAT_B8G8R8A8_SRGB;
}

/* channels of a texel format, 0 when unsupported (tools.cpp:469-512) */
int format_channels(int32_t f)
{
    switch (f) {
    case RND_FORMAT_R8_UNORM: case RND_FORMAT_R8_UINT: case RND_FORMAT_R8_SRGB: return 1;
    case RND_FORMAT_R8G8_UNORM: case RND_FORMAT_R8G8_UINT: case RND_FORMAT_R8G8_SRGB: return 2;
    case RND_FORMAT_R8G8B8_UNORM: case RND_FORMAT_R8G8B8_UINT: case RND_FORMAT_R8G8B8_SRGB: return 3;
    case RND_FORMAT_R8G8B8A8_UNORM: case RND_FORMAT_R8G8B8A8_UINT: case RND_FORMAT_R8G8B8A8_SRGB: return 4;
    default: return 0;
    }
}

/* Expand to RGBA8 with the component mapping of the reference's image views
 * (tools.cpp:474-509): R -> (r,r,r,1), RG -> (r,0,0,g), RGB -> (r,g,b,1). */
void expand_rgba(const uint8_t* src, int ch, size_t texels, std::vector<uint8_t>& out)
{
    out.resize(texels * 4);
    for (size_t i = 0; i < texels; ++i) {
        const uint8_t* p = src + i * (size_t)ch;
        uint8_t* o = &out[i * 4];
        switch (ch) {
        case 1: o[0] = p[0]; o[1] = p[0]; o[2] = p[0]; o[3] = 255; break;
        case 2: o[0] = p[0]; o[1] = 0; o[2] = 0; o[3] = p[1]; break;
        case 3: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 255; break;
        default: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; break;
        }
    }
}

uint32_t level_dim(uint32_t d, uint32_t l) { uint32_t v = d >> l; return v ? v : 1u; }

uint32_t host_depth24(float z)
{
    z = fminf(fmaxf(z, 0.0f), 1.0f);
    uint32_t d = (uint32_t)rintf(z * 16777215.0f); /* as the kernels convert: nearest integer, ties to even */
    return d > 0xFFFFFFu ? 0xFFFFFFu : d;
}

/* What a frame's kernels need to know about the pass itself -- the target, the render area and the tiles it touches,
 * viewport and scissor, the clear values and the depth state -- as opposed to its draws.  Shared by command recording
 * and by the entry point that allocates nothing. */
void pass_params(FrameParams& fp, rnd_device dev, rnd_target tg, const int32_t area[4], bool has_viewport_and_scissor, const float* vp,
                 const int32_t* scissor, const float* clear_color, float clear_depth, int depth_test, int depth_less, uint32_t tile_h)
{
    fp.tile_h = tile_h;
    fp.tile_h_magic = (uint32_t)((0x100000000ull + tile_h - 1) / tile_h); /* exact for rows below 2^16: row * (magic * tile_h - 2^32) stays below 2^32 */
    fp.status_host = dev->status_dev;
    fp.color = tg->color; fp.depth = tg->depth;
    fp.tables = dev->tables;
    fp.width = tg->width; fp.height = tg->height; fp.samples = tg->samples;
    fp.color_srgb = format_is_srgb(tg->color_format) ? 1u : 0u;
    fp.color_bgra = (tg->color_format == RND_FORMAT_B8G8R8A8_SRGB || tg->color_format == RND_FORMAT_B8G8R8A8_UNORM) ? 1u : 0u;
    fp.tiles_x = (tg->width + TILE_W - 1) / TILE_W; fp.tiles_y = (tg->height + tile_h - 1) / tile_h;
    fp.shard_index = 0; fp.shard_count = 1;
    fp.area_x0 = area[0]; fp.area_y0 = area[1];
    fp.area_x1 = area[0] + area[2] - 1; fp.area_y1 = area[1] + area[3] - 1;
    {
        /* tiles the render area touches */
        const int64_t ax0 = std::max<int64_t>(fp.area_x0, 0), ay0 = std::max<int64_t>(fp.area_y0, 0);
        const int64_t ax1 = std::min<int64_t>(fp.area_x1, (int64_t)tg->width - 1), ay1 = std::min<int64_t>(fp.area_y1, (int64_t)tg->height - 1);
        if (ax0 <= ax1 && ay0 <= ay1) {
            fp.grid_x0 = (uint32_t)(ax0 / TILE_W); fp.grid_y0 = (uint32_t)(ay0 / tile_h);
            fp.grid_w = (uint32_t)(ax1 / TILE_W) - fp.grid_x0 + 1; fp.grid_h = (uint32_t)(ay1 / tile_h) - fp.grid_y0 + 1;
        }
    }
    fp.cull_lo_x = fp.cull_lo_y = -INFINITY; fp.cull_hi_x = fp.cull_hi_y = INFINITY;
    if (has_viewport_and_scissor) {
        fp.half_w = vp[2] * 0.5f; fp.half_h = vp[3] * 0.5f;
        fp.ox = vp[0] + fp.half_w; fp.oy = vp[1] + fp.half_h;
        fp.pz = vp[5] - vp[4]; fp.zmin = vp[4];
        /* fragments land only on pixels whose centre is inside viewport, scissor and render area */
        int vx0 = (int)ceilf(vp[0] - 0.5f), vy0 = (int)ceilf(vp[1] - 0.5f);
        int vx1 = (int)ceilf(vp[0] + vp[2] - 0.5f) - 1, vy1 = (int)ceilf(vp[1] + vp[3] - 0.5f) - 1;
        fp.clamp_x0 = std::max(std::max(vx0, scissor[0]), std::max(area[0], 0));
        fp.clamp_y0 = std::max(std::max(vy0, scissor[1]), std::max(area[1], 0));
        int64_t sx1 = (int64_t)scissor[0] + scissor[2] - 1, sy1 = (int64_t)scissor[1] + scissor[3] - 1;
        fp.clamp_x1 = (int32_t)std::min<int64_t>(std::min<int64_t>(vx1, sx1), std::min<int64_t>(fp.area_x1, (int64_t)tg->width - 1));
        fp.clamp_y1 = (int32_t)std::min<int64_t>(std::min<int64_t>(vy1, sy1), std::min<int64_t>(fp.area_y1, (int64_t)tg->height - 1));
        /* the rectangle grown by a pixel, in normalised device coordinates (window = ndc * half + origin) */
        if (fp.half_w > 0.0f && fp.clamp_x0 <= fp.clamp_x1) {
            fp.cull_lo_x = ((float)fp.clamp_x0 - 1.0f - fp.ox) / fp.half_w;
            fp.cull_hi_x = ((float)fp.clamp_x1 + 2.0f - fp.ox) / fp.half_w;
        }
        if (fp.half_h > 0.0f && fp.clamp_y0 <= fp.clamp_y1) {
            fp.cull_lo_y = ((float)fp.clamp_y0 - 1.0f - fp.oy) / fp.half_h;
            fp.cull_hi_y = ((float)fp.clamp_y1 + 2.0f - fp.oy) / fp.half_h;
        }
    } else {
        fp.clamp_x0 = 0; fp.clamp_y0 = 0; fp.clamp_x1 = -1; fp.clamp_y1 = -1;
    }
    memcpy(fp.clear_color, clear_color, 16);
    fp.clear_packed = host_pack_color(dev->host_tables, clear_color, fp.color_srgb != 0, fp.color_bgra != 0);
    fp.clear_d24 = host_depth24(clear_depth);
    fp.depth_test = (uint32_t)depth_test;
    fp.depth_write = (uint32_t)depth_test;
    fp.depth_less = (uint32_t)depth_less;
}

int texture_alloc(rnd_device dev, uint32_t w, uint32_t h, int32_t format, uint32_t levels, const rnd_sampler_info* sampler,
                  rnd_texture* out)
{
    if (!w || !h || w > 16384 || h > 16384) return fail(RND_ERROR_INVALID_ARGUMENT, "texture extent out of range");
    if (levels < 1 || levels > (uint32_t)MAX_TEXTURE_LEVELS) return fail(RND_ERROR_INVALID_ARGUMENT, "texture level count out of range (a side of at most 16384 texels has at most 15 levels)");
    rnd_texture t = new rnd_texture_t();
    t->dev = dev; t->width = w; t->height = h; t->levels = levels; t->srgb = format_is_srgb(format) ? 1u : 0u;
    t->sampler = sampler ? *sampler : rnd_sampler_info{RND_FILTER_LINEAR, 0, 0.0f};
    memset(t->level_ptr, 0, sizeof(t->level_ptr));
    for (uint32_t l = 0; l < levels; ++l) {
        /* 16 bytes of slack behind every level: a bilinear sample reads its two columns as one 8-byte load,
         * which on a level one texel wide reaches four bytes past the last row */
        size_t bytes = (size_t)level_dim(w, l) * level_dim(h, l) * 4 + 16;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->level_ptr[l]), bytes);
        if (e != hipSuccess) {
            for (uint32_t k = 0; k < l; ++k) (void)hipFree(t->level_ptr[k]);
            delete t;
            return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "hipMalloc texture level failed");
        }
    }
    *out = t;
    ++dev->live_objects;
    return RND_SUCCESS;
}

} // namespace

extern "C" {

const char* rnd_last_error(void) { return g_last_error.c_str(); }

/* ------------------------------------------------------------------ device */

int rnd_device_create(int hip_ordinal, void* stream, rnd_device* out)
{
    if (!out) return fail(RND_ERROR_INVALID_ARGUMENT, "rnd_device_create: out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RND_ERROR_INITIALIZATION_FAILED, "no HIP device is visible");
    if (hip_ordinal < 0 || hip_ordinal >= n) return fail(RND_ERROR_INVALID_ARGUMENT, "HIP device ordinal out of range");
    HIP_TRY(hipSetDevice(hip_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, hip_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RND_ERROR_INITIALIZATION_FAILED, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    rnd_device dev = new rnd_device_t();
    dev->ordinal = hip_ordinal;
    /* The runtime keeps one pool of hardware queues per stream priority (GPU_MAX_HW_QUEUES of them each, 4 unless the
     * environment says otherwise) and lets streams share a queue once a pool is used up; two command buffers on one
     * queue run one after the other instead of side by side (measured: three frames in flight 53 us per frame with
     * a shared queue, 43 us without).  The default stream and whatever streams the application makes fill the pool of
     * normal priority, so this library's streams take the pool of the lowest priority, which nothing else in a process
     * normally uses, and a collective or copy issued at normal priority beside them is not kept waiting by a frame.
     *
     * ALL of the library's streams are of that one priority, the device's own included.  The streams wait for one
     * another on the GPU (a copy into a uniform buffer follows the frames in flight, later frames follow the copy), and
     * a hardware queue parked in such a wait still counts as ready work for the queue scheduler: with the device's
     * stream at normal priority and the command buffers' at the lowest, a wait of the former for a frame of the latter
     * kept that frame's queue from ever being served, and the wait never ended (with the command buffers at high
     * priority the other way round: their waits starved the copy they were waiting for).  Among queues of one
     * priority no wait starves what it waits for.  RND_STREAM_PRIORITY=high or =normal chooses another pool, for all
     * of them. */
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        const char* pr = getenv("RND_STREAM_PRIORITY");
        dev->cmd_stream_priority = !pr || !strcmp(pr, "low") ? least : !strcmp(pr, "high") ? greatest : 0;
    }
    if (stream) { dev->stream = static_cast<hipStream_t>(stream); dev->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithPriority(&dev->stream, hipStreamNonBlocking, dev->cmd_stream_priority);
        if (e != hipSuccess) { delete dev; return fail(RND_ERROR_DEVICE_LOST, "hipStreamCreate failed"); }
        dev->own_stream = true;
    }
    /* texel tables: value[0][k] = linear value of sRGB code k, value[1][k] = k / 255; thr[k] = linear value of code k-0.5;
     * lut[bucket] = number of thresholds <= the bucket's lower edge */
    ShadeTables& host_tables = dev->host_tables;
    memset(&host_tables, 0, sizeof(host_tables));
    for (int k = 0; k < 256; ++k) { host_tables.value[0][k] = (float)srgb_to_linear(k / 255.0); host_tables.value[1][k] = (float)k / 255.0f; }
    host_tables.thr[0] = -INFINITY;
    for (int k = 1; k < 256; ++k) host_tables.thr[k] = (float)srgb_to_linear((k - 0.5) / 255.0);
    host_tables.thr[256] = INFINITY;
    for (uint32_t i = 0; i < SRGB_LUT_SIZE; ++i) {
        uint32_t bits = (i + SRGB_LUT_FIRST) << SRGB_LUT_SHIFT, next_bits = (i + 1 + SRGB_LUT_FIRST) << SRGB_LUT_SHIFT;
        float c, next;
        memcpy(&c, &bits, 4);
        memcpy(&next, &next_bits, 4);
        uint32_t k = 0;
        while (k < 255 && host_tables.thr[k + 1] <= c) ++k;
        host_tables.lut[i] = (uint8_t)k;
        /* the encoder looks at one threshold after the table's answer: a bucket may hold only one */
        if (k < 254 && host_tables.thr[k + 2] < next) { rnd_device_destroy(dev); return fail(RND_ERROR_INITIALIZATION_FAILED, "sRGB bucket table: two thresholds in one bucket"); }
    }
    /* the encoder clamps a colour into the table's range [2^-13, 1) first: codes 0 and 255 must reach across its ends */
    if (!(host_tables.thr[1] > 1.220703125e-4f && host_tables.thr[255] < 0.99999994f)) {
        rnd_device_destroy(dev);
        return fail(RND_ERROR_INITIALIZATION_FAILED, "sRGB bucket table: the first or last threshold lies outside the table's range");
    }
    if (hipMalloc(reinterpret_cast<void**>(&dev->tables), sizeof(ShadeTables)) != hipSuccess) {
        rnd_device_destroy(dev);
        return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "hipMalloc tables failed");
    }
    /* a failure from here on releases what the device object already holds */
    hipError_t e = upload_now(dev->tables, &host_tables, sizeof(ShadeTables), dev->stream);
    for (int i = 0; i <= RND_STAGE_COUNT && e == hipSuccess; ++i) e = hipEventCreate(&dev->ev[i]);
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void**>(&dev->stage_host), rnd_device_t::STAGE_SLOTS * rnd_device_t::STAGE_SLOT_BYTES, hipHostMallocDefault);
    for (int i = 0; i < rnd_device_t::STAGE_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&dev->stage_ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&dev->status_host), 64, hipHostMallocDefault);
    if (e == hipSuccess) {
        memset(dev->status_host, 0, 64);
        void* dp = nullptr;
        e = hipHostGetDevicePointer(&dp, dev->status_host, 0);
        dev->status_dev = static_cast<uint32_t*>(dp);
    }
    if (e == hipSuccess) e = dev->order.create();
    if (const char* wg = getenv("RND_TILE_WORKGROUPS")) dev->four_tile_workgroups = atoi(wg) == 4;
    {
        /* as many command-buffer streams as one priority pool has hardware queues, less the one the device's own
         * stream takes (RND_CMD_STREAMS sets the number itself) */
        int n_cs = 3;
        if (const char* q = getenv("RND_CMD_STREAMS")) n_cs = atoi(q);
        else if (const char* hq = getenv("GPU_MAX_HW_QUEUES")) n_cs = atoi(hq) - 1;
        dev->max_cmd_streams = n_cs < 1 ? 1 : n_cs > 32 ? 32 : n_cs;
        if (const char* ds = getenv("RND_DEBUG_SYNC")) {
            if (atoi(ds) > 0) {
                const char* ms = getenv("RND_DEBUG_SYNC_MS");
                dev->debug_sync_ms = ms && atoi(ms) > 0 ? atoi(ms) : 20000;
            }
        }
    }
    /* every GPU holds its own copy of the static scratch array at its own address: asked for once per device, here,
     * with that device current */
    if (e == hipSuccess) { void* sp = nullptr; e = noalloc_scratch_address(&sp); dev->scratch = static_cast<uint8_t*>(sp); }
    if (e != hipSuccess) {
        rnd_device_destroy(dev);
        return fail(e == hipErrorOutOfMemory ? RND_ERROR_OUT_OF_DEVICE_MEMORY : RND_ERROR_DEVICE_LOST,
                    std::string("rnd_device_create: ") + hipGetErrorString(e));
    }
    *out = dev;
    return RND_SUCCESS;
}

int rnd_device_destroy(rnd_device dev)
{
    if (!dev) return RND_SUCCESS;
    /* every object created on the device points back at it and its destroy call follows that pointer: the device goes last,
     * as the reference destroys its context last (context.cpp:567-574) */
    if (dev->live_objects > 0)
        return fail(RND_ERROR_INVALID_ARGUMENT, "device_destroy: " + std::to_string(dev->live_objects) +
                    " object(s) created on this device are still alive; destroy them first (the device is left as it is)");
    (void)hipSetDevice(dev->ordinal);
    (void)sync_all(dev);
    dev->order.destroy();
    for (CmdStream* cs : dev->cmd_streams) {
        cs->done.destroy();
        if (cs->stream) (void)hipStreamDestroy(cs->stream);
        delete cs;
    }
    dev->cmd_streams.clear();
    if (dev->tables) (void)hipFree(dev->tables);
    for (int i = 0; i <= RND_STAGE_COUNT; ++i) if (dev->ev[i]) (void)hipEventDestroy(dev->ev[i]);
    for (int i = 0; i < rnd_device_t::STAGE_SLOTS; ++i) if (dev->stage_ev[i]) (void)hipEventDestroy(dev->stage_ev[i]);
    if (dev->stage_host) (void)hipHostFree(dev->stage_host);
    if (dev->status_host) (void)hipHostFree(dev->status_host);
    if (dev->own_stream && dev->stream) (void)hipStreamDestroy(dev->stream);
    delete dev;
    return RND_SUCCESS;
}

int rnd_device_wait_idle(rnd_device dev)
{
    if (!dev) return fail(RND_ERROR_INVALID_ARGUMENT, "device is null");
    if (int r = set_device(dev)) return r;
    HIP_TRY(sync_all(dev));
    return report_overflow(dev);
}

void* rnd_device_stream(rnd_device dev) { return dev ? static_cast<void*>(dev->stream) : nullptr; }

int rnd_device_set_profiling(rnd_device dev, int enable)
{
    if (!dev) return fail(RND_ERROR_INVALID_ARGUMENT, "device is null");
    dev->profiling = enable != 0;
    dev->stage_valid = false;
    return RND_SUCCESS;
}

int rnd_device_stage_times(rnd_device dev, float* ms, int count)
{
    if (!dev || !ms || count < RND_STAGE_COUNT) return fail(RND_ERROR_INVALID_ARGUMENT, "stage_times: bad arguments");
    if (!dev->stage_valid) return fail(RND_ERROR_INVALID_ARGUMENT, "no profiled submit has completed");
    if (int r = set_device(dev)) return r;
    HIP_TRY(hipEventSynchronize(dev->ev[RND_STAGE_COUNT]));
    for (int i = 0; i < RND_STAGE_COUNT; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], dev->ev[i], dev->ev[i + 1]));
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ buffers */

int rnd_buffer_create(rnd_device dev, uint64_t bytes, uint32_t usage, const void* init, rnd_buffer* out)
{
    if (!dev || !out || !bytes) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_create: bad arguments");
    if (int r = set_device(dev)) return r;
    uint8_t* p = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), (size_t)bytes));
    if (init) {
        hipError_t e = upload_now(p, init, (size_t)bytes, dev->stream);
        if (e != hipSuccess) { (void)hipFree(p); return fail(RND_ERROR_DEVICE_LOST, "buffer upload failed"); }
    }
    rnd_buffer b = new rnd_buffer_t{dev, p, bytes, usage};
    ++dev->live_objects;
    *out = b;
    return RND_SUCCESS;
}

int rnd_buffer_write(rnd_buffer buf, uint64_t offset, const void* src, uint64_t bytes)
{
    if (!buf || !src) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_write: bad arguments");
    if (offset > buf->bytes || bytes > buf->bytes - offset) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_write: range outside the buffer");
    if (int r = set_device(buf->dev)) return r;
    rnd_device dev = buf->dev;
    /* stream-ordered after earlier submits: a frame already queued still sees the old bytes */
    if (buf->usage == RND_BUFFER_USAGE_UNIFORM && bytes <= PATCH_MAX_BYTES && bytes % 4 == 0 && offset % 4 == 0) {
        /* the per-frame uniform block (base_obj_viewer_app.cpp:134-138): no copy is queued; the
         * submits that follow take the bytes along in their kernel arguments */
        uint8_t* dst = buf->ptr + offset;
        if (dev->pending.dst && !(dev->pending.dst == dst && dev->pending.bytes == (uint32_t)bytes))
            if (int r = flush_pending(dev)) return r;
        dev->pending.dst = dst;
        dev->pending.bytes = (uint32_t)bytes;
        memcpy(dev->pending.words, src, (size_t)bytes);
        return RND_SUCCESS;
    }
    if (int r = flush_pending(dev)) return r;
    if (buf->usage & RND_BUFFER_USAGE_VERTEX) ++dev->geometry_epoch;
    if (buf->usage & RND_BUFFER_USAGE_INDEX) ++dev->index_epoch;
    if (bytes <= rnd_device_t::STAGE_SLOT_BYTES) return staged_write(dev, buf->ptr + offset, src, (size_t)bytes);
    /* a large upload (geometry): frames in flight may still read the buffer, wait them out */
    HIP_TRY(sync_all(dev));
    HIP_TRY(hipMemcpyAsync(buf->ptr + offset, src, (size_t)bytes, hipMemcpyHostToDevice, dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    return RND_SUCCESS;
}

int rnd_buffer_write_device(rnd_buffer buf, uint64_t offset, const void* device_src, uint64_t bytes)
{
    if (!buf || !device_src) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_write_device: bad arguments");
    if (offset > buf->bytes || bytes > buf->bytes - offset) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_write_device: range outside the buffer");
    if (int r = set_device(buf->dev)) return r;
    if (int r = flush_pending(buf->dev)) return r;
    if (buf->usage & RND_BUFFER_USAGE_VERTEX) ++buf->dev->geometry_epoch;
    if (buf->usage & RND_BUFFER_USAGE_INDEX) ++buf->dev->index_epoch;
    /* queued on the device's stream behind every frame already submitted; later frames follow it */
    HIP_TRY(follow_frames(buf->dev));
    HIP_TRY(hipMemcpyAsync(buf->ptr + offset, device_src, (size_t)bytes, hipMemcpyDeviceToDevice, buf->dev->stream));
    ++buf->dev->upload_seq;
    return RND_SUCCESS;
}

int rnd_buffer_read(rnd_buffer buf, uint64_t offset, void* dst, uint64_t bytes)
{
    if (!buf || !dst) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_read: bad arguments");
    if (offset > buf->bytes || bytes > buf->bytes - offset) return fail(RND_ERROR_INVALID_ARGUMENT, "buffer_read: range outside the buffer");
    if (int r = set_device(buf->dev)) return r;
    if (int r = flush_pending(buf->dev)) return r;
    HIP_TRY(sync_all(buf->dev));
    HIP_TRY(hipMemcpy(dst, buf->ptr + offset, (size_t)bytes, hipMemcpyDeviceToHost));
    return RND_SUCCESS;
}

int rnd_buffer_destroy(rnd_buffer buf)
{
    if (!buf) return RND_SUCCESS;
    (void)hipSetDevice(buf->dev->ordinal);
    /* a held-back write into this buffer has nowhere to go any more */
    if (buf->dev->pending.dst >= buf->ptr && buf->dev->pending.dst < buf->ptr + buf->bytes) buf->dev->pending.dst = nullptr;
    (void)sync_all(buf->dev);
    (void)hipFree(buf->ptr);
    --buf->dev->live_objects;
    delete buf;
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ textures */

int rnd_texture_create_2d(rnd_device dev, uint32_t width, uint32_t height, int32_t format, int32_t gen_mips,
                          const void* texels, const rnd_sampler_info* sampler, rnd_texture* out)
{
    if (!dev || !out || !texels) return fail(RND_ERROR_INVALID_ARGUMENT, "texture_create_2d: bad arguments");
    int ch = format_channels(format);
    if (!ch) return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "unsupported pixels format."); /* tools.cpp:511 */
    if (int r = set_device(dev)) return r;
    /* tools.cpp:514: mip_levels = (uint32_t)log2(max(w,h)) -- one short of a full chain */
    uint32_t levels = 1;
    if (gen_mips) {
        levels = (uint32_t)log2((double)(width > height ? width : height));
        if (levels < 1) levels = 1;
    }
    rnd_texture t = nullptr;
    if (int r = texture_alloc(dev, width, height, format, levels, sampler, &t)) return r;
    std::vector<uint8_t> rgba;
    const uint8_t* src = static_cast<const uint8_t*>(texels);
    if (ch != 4) { expand_rgba(src, ch, (size_t)width * height, rgba); src = rgba.data(); }
    hipError_t e = upload_now(t->level_ptr[0], src, (size_t)width * height * 4, dev->stream);
    if (e != hipSuccess) { rnd_texture_destroy(t); return fail(RND_ERROR_DEVICE_LOST, "texture upload failed"); }
    for (uint32_t l = 1; l < levels; ++l) {
        launch_mip(t->level_ptr[l - 1], level_dim(width, l - 1), level_dim(height, l - 1), t->level_ptr[l], level_dim(width, l),
                   level_dim(height, l), (int)t->srgb, dev->tables, dev->stream);
    }
    e = hipStreamSynchronize(dev->stream); /* the reference waits on a fence here (tools.cpp:776-778) */
    if (e != hipSuccess) { rnd_texture_destroy(t); return fail(RND_ERROR_DEVICE_LOST, std::string("mip generation failed: ") + hipGetErrorString(e)); }
    *out = t;
    return RND_SUCCESS;
}

int rnd_texture_create_levels(rnd_device dev, uint32_t width, uint32_t height, int32_t format, uint32_t level_count,
                              const void* const* level_texels, const rnd_sampler_info* sampler, rnd_texture* out)
{
    if (!dev || !out || !level_texels) return fail(RND_ERROR_INVALID_ARGUMENT, "texture_create_levels: bad arguments");
    int ch = format_channels(format);
    if (!ch) return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "unsupported pixels format.");
    if (int r = set_device(dev)) return r;
    rnd_texture t = nullptr;
    if (int r = texture_alloc(dev, width, height, format, level_count, sampler, &t)) return r;
    for (uint32_t l = 0; l < level_count; ++l) {
        size_t texels = (size_t)level_dim(width, l) * level_dim(height, l);
        std::vector<uint8_t> rgba;
        const uint8_t* src = static_cast<const uint8_t*>(level_texels[l]);
        if (!src) { rnd_texture_destroy(t); return fail(RND_ERROR_INVALID_ARGUMENT, "texture level pointer is null"); }
        if (ch != 4) { expand_rgba(src, ch, texels, rgba); src = rgba.data(); }
        hipError_t e = upload_now(t->level_ptr[l], src, texels * 4, dev->stream);
        if (e != hipSuccess) { rnd_texture_destroy(t); return fail(RND_ERROR_DEVICE_LOST, "texture upload failed"); }
    }
    *out = t;
    return RND_SUCCESS;
}

int rnd_texture_level_count(rnd_texture tex, uint32_t* out)
{
    if (!tex || !out) return fail(RND_ERROR_INVALID_ARGUMENT, "texture_level_count: bad arguments");
    *out = tex->levels;
    return RND_SUCCESS;
}

int rnd_texture_read_level(rnd_texture tex, uint32_t level, void* rgba8_out)
{
    if (!tex || !rgba8_out || level >= tex->levels) return fail(RND_ERROR_INVALID_ARGUMENT, "texture_read_level: bad arguments");
    if (int r = set_device(tex->dev)) return r;
    HIP_TRY(sync_all(tex->dev));
    HIP_TRY(hipMemcpy(rgba8_out, tex->level_ptr[level], (size_t)level_dim(tex->width, level) * level_dim(tex->height, level) * 4,
                      hipMemcpyDeviceToHost));
    return RND_SUCCESS;
}

int rnd_texture_destroy(rnd_texture tex)
{
    if (!tex) return RND_SUCCESS;
    (void)hipSetDevice(tex->dev->ordinal);
    (void)sync_all(tex->dev);
    for (uint32_t l = 0; l < 16; ++l) if (tex->level_ptr[l]) (void)hipFree(tex->level_ptr[l]);
    --tex->dev->live_objects;
    delete tex;
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ targets */

int rnd_target_create(rnd_device dev, uint32_t width, uint32_t height, int32_t color_format, int32_t depth_format,
                      uint32_t samples, rnd_target* out)
{
    if (!dev || !out) return fail(RND_ERROR_INVALID_ARGUMENT, "target_create: bad arguments");
    if (!width || !height || width > TILE_W * MAX_TILE_AXIS || height > TILE_H * MAX_TILE_AXIS)
        return fail(RND_ERROR_INVALID_ARGUMENT, "target extent must be 1..8192 in each dimension");
    if (color_format != RND_FORMAT_R8G8B8A8_SRGB && color_format != RND_FORMAT_B8G8R8A8_SRGB &&
        color_format != RND_FORMAT_R8G8B8A8_UNORM && color_format != RND_FORMAT_B8G8R8A8_UNORM)
        return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "colour format must be R8G8B8A8 or B8G8R8A8, UNORM or SRGB");
    if (depth_format != RND_FORMAT_D24_UNORM_S8_UINT) return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "depth format must be D24_UNORM_S8_UINT");
    if (samples != 1 && samples != 4) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "sample count must be 1 or 4");
    if (int r = set_device(dev)) return r;
    rnd_target t = new rnd_target_t{dev, width, height, samples, color_format, nullptr, nullptr, EventRing(), nullptr, false};
    if (t->order.create() != hipSuccess) { t->order.destroy(); delete t; return fail(RND_ERROR_DEVICE_LOST, "hipEventCreate failed"); }
    ++dev->live_objects; /* from here on rnd_target_destroy releases it, on the failure paths below too */
    size_t px = (size_t)width * height;
    hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&t->color), px * 4);
    hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&t->depth), px * samples * 4);
    if (e1 != hipSuccess || e2 != hipSuccess) { rnd_target_destroy(t); return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "hipMalloc target failed"); }
    (void)zero_now(t->color, px * 4, dev->stream);
    (void)zero_now(t->depth, px * samples * 4, dev->stream);
    dev->targets.push_back(t);
    *out = t;
    return RND_SUCCESS;
}

int rnd_target_read(rnd_target target, void* color_out, uint32_t* depth_out)
{
    if (!target) return fail(RND_ERROR_INVALID_ARGUMENT, "target_read: target is null");
    if (int r = set_device(target->dev)) return r;
    HIP_TRY(sync_all(target->dev));
    size_t px = (size_t)target->width * target->height;
    if (color_out) HIP_TRY(hipMemcpy(color_out, target->color, px * 4, hipMemcpyDeviceToHost));
    if (depth_out) HIP_TRY(hipMemcpy(depth_out, target->depth, px * target->samples * 4, hipMemcpyDeviceToHost));
    return report_overflow(target->dev);
}

void* rnd_target_color_device_ptr(rnd_target target) { return target ? target->color : nullptr; }

int rnd_target_copy_color(rnd_target target, void* device_dst)
{
    if (!target || !device_dst) return fail(RND_ERROR_INVALID_ARGUMENT, "target_copy_color: bad arguments");
    if (int r = set_device(target->dev)) return r;
    /* queued behind the frame that last rendered into the target, on that frame's stream; a
     * later frame into the target follows the copy */
    hipStream_t s = target->last_valid ? target->last_stream : target->dev->stream;
    HIP_TRY(hipMemcpyAsync(device_dst, target->color, (size_t)target->width * target->height * 4, hipMemcpyDeviceToDevice, s));
    target->last_stream = s; target->last_valid = true;
    if (s == target->dev->stream) ++target->dev->upload_seq;
    return RND_SUCCESS;
}

int rnd_target_copy_color_rows(rnd_target target, uint32_t first_row, uint32_t row_count, void* device_dst)
{
    if (!target || !device_dst) return fail(RND_ERROR_INVALID_ARGUMENT, "target_copy_color_rows: bad arguments");
    if (first_row > target->height || row_count > target->height - first_row) return fail(RND_ERROR_INVALID_ARGUMENT, "target_copy_color_rows: rows outside the image");
    if (int r = set_device(target->dev)) return r;
    if (!row_count) return RND_SUCCESS;
    hipStream_t s = target->last_valid ? target->last_stream : target->dev->stream;
    const size_t pitch = (size_t)target->width * 4;
    HIP_TRY(hipMemcpyAsync(device_dst, target->color + (size_t)first_row * pitch, (size_t)row_count * pitch, hipMemcpyDeviceToDevice, s));
    target->last_stream = s; target->last_valid = true;
    if (s == target->dev->stream) ++target->dev->upload_seq;
    return RND_SUCCESS;
}

int rnd_target_destroy(rnd_target target)
{
    if (!target) return RND_SUCCESS;
    (void)hipSetDevice(target->dev->ordinal);
    (void)sync_all(target->dev);
    target->order.destroy();
    { auto& list = target->dev->targets; list.erase(std::remove(list.begin(), list.end(), target), list.end()); }
    if (target->color) (void)hipFree(target->color);
    if (target->depth) (void)hipFree(target->depth);
    --target->dev->live_objects;
    delete target;
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ pipelines */

int rnd_pipeline_create(rnd_device dev, const rnd_pipeline_info* info, rnd_pipeline* out)
{
    if (!dev || !info || !out) return fail(RND_ERROR_INVALID_ARGUMENT, "pipeline_create: bad arguments");
    if (info->shader < RND_SHADER_DUMMY || info->shader > RND_SHADER_PUSH_CONSTANTS) return fail(RND_ERROR_INVALID_ARGUMENT, "unknown shader id");
    if (info->topology != RND_TOPOLOGY_TRIANGLE_LIST) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "only TRIANGLE_LIST is supported");
    if (info->position.format != RND_FORMAT_R32G32B32_SFLOAT) return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "position must be R32G32B32_SFLOAT");
    if (info->normal.format != 0 && info->normal.format != RND_FORMAT_R32G32B32_SFLOAT) return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "normal must be R32G32B32_SFLOAT");
    if (info->uv.format != 0 && info->uv.format != RND_FORMAT_R32G32_SFLOAT) return fail(RND_ERROR_FORMAT_NOT_SUPPORTED, "uv must be R32G32_SFLOAT");
    if (!info->stride || (info->stride & 3u) || (info->position.offset & 3u) || (info->normal.offset & 3u) || (info->uv.offset & 3u))
        return fail(RND_ERROR_INVALID_ARGUMENT, "vertex stride and attribute offsets must be multiples of 4");
    if (info->position.offset + 12 > info->stride || (info->normal.format && info->normal.offset + 12 > info->stride) ||
        (info->uv.format && info->uv.offset + 8 > info->stride))
        return fail(RND_ERROR_INVALID_ARGUMENT, "vertex attribute reaches past the stride");
    if (info->cull_mode < 0 || info->cull_mode > 3 || info->front_face < 0 || info->front_face > 1) return fail(RND_ERROR_INVALID_ARGUMENT, "bad cull state");
    if (info->depth_test) {
        if (info->depth_compare != RND_COMPARE_LESS && info->depth_compare != RND_COMPARE_LESS_OR_EQUAL)
            return fail(RND_ERROR_FEATURE_NOT_PRESENT, "depth compare must be LESS or LESS_OR_EQUAL");
        if (!info->depth_write) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "depth test without depth write is not supported");
    }
    if (info->samples != 1 && info->samples != 4) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "sample count must be 1 or 4");
    if (info->color_write_mask != 0xFu) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "colour write mask must be RGBA");
    if (info->blend_enable) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "blending is not supported");
    *out = new rnd_pipeline_t{dev, *info};
    ++dev->live_objects;
    return RND_SUCCESS;
}

int rnd_pipeline_destroy(rnd_pipeline p)
{
    if (!p) return RND_SUCCESS;
    --p->dev->live_objects;
    delete p;
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ command recording */

int rnd_cmd_create(rnd_device dev, rnd_cmd* out)
{
    if (!dev || !out) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd_create: bad arguments");
    if (int r = set_device(dev)) return r;
    rnd_cmd c = new rnd_cmd_t();
    c->dev = dev;
    c->stream = dev->stream;
    if (dev->own_stream) {
        /* a caller-supplied stream keeps everything on that stream; otherwise frames of different command buffers
         * may overlap: the command buffer takes a stream of the device's pool that no other uses, a new one while
         * the pool is not full, else the one with the fewest users */
        CmdStream* pick = nullptr;
        for (CmdStream* cs : dev->cmd_streams) if (cs->users == 0) { pick = cs; break; }
        if (!pick && (int)dev->cmd_streams.size() < dev->max_cmd_streams) {
            pick = new CmdStream();
            pick->index = (int)dev->cmd_streams.size();
            hipError_t e = hipStreamCreateWithPriority(&pick->stream, hipStreamNonBlocking, dev->cmd_stream_priority);
            if (e == hipSuccess) e = pick->done.create();
            if (e != hipSuccess) {
                pick->done.destroy();
                if (pick->stream) (void)hipStreamDestroy(pick->stream);
                delete pick; delete c;
                return fail(RND_ERROR_DEVICE_LOST, std::string("cmd_create: making a stream failed: ") + hipGetErrorString(e));
            }
            pick->seen_seq = 0; /* nothing queued on the device's stream has been waited for yet */
            dev->cmd_streams.push_back(pick);
        }
        if (!pick) for (CmdStream* cs : dev->cmd_streams) if (!pick || cs->users < pick->users) pick = cs;
        ++pick->users;
        c->cs = pick;
        c->stream = pick->stream;
    }
    dev->cmds.push_back(c);
    ++dev->live_objects;
    *out = c;
    return RND_SUCCESS;
}

int rnd_cmd_begin(rnd_cmd cmd)
{
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd is null");
    if (cmd->state == CMD_RECORDING || cmd->state == CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "command buffer is already recording");
    if (int r = set_device(cmd->dev)) return r;
    (void)sync_all(cmd->dev);
    cmd->release_compiled();
    cmd->draws.clear();
    cmd->has_pass = false; cmd->has_viewport = false; cmd->has_scissor = false;
    cmd->cur_pipeline = nullptr; cmd->cur_vb = nullptr; cmd->cur_ib = nullptr; cmd->cur_ubo = nullptr; cmd->cur_tex = nullptr;
    memset(cmd->cur_push, 0, sizeof(cmd->cur_push));
    cmd->state = CMD_RECORDING;
    return RND_SUCCESS;
}

int rnd_cmd_begin_pass(rnd_cmd cmd, rnd_target target, const float clear_color[4], float clear_depth,
                       int32_t area_x, int32_t area_y, uint32_t area_w, uint32_t area_h)
{
    if (!cmd || !target || !clear_color) return fail(RND_ERROR_INVALID_ARGUMENT, "begin_pass: bad arguments");
    if (cmd->state != CMD_RECORDING) return fail(RND_ERROR_RECORDING, "begin_pass outside recording state");
    if (cmd->has_pass) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "one render pass per command buffer");
    if (target->dev != cmd->dev) return fail(RND_ERROR_INVALID_ARGUMENT, "target belongs to another device");
    if (area_x < 0 || area_y < 0 || (uint64_t)area_x + area_w > target->width || (uint64_t)area_y + area_h > target->height)
        return fail(RND_ERROR_INVALID_ARGUMENT, "render area outside the target");
    cmd->target = target;
    memcpy(cmd->clear_color, clear_color, 16);
    cmd->clear_depth = clear_depth;
    cmd->area[0] = area_x; cmd->area[1] = area_y; cmd->area[2] = (int32_t)area_w; cmd->area[3] = (int32_t)area_h;
    cmd->has_pass = true;
    cmd->state = CMD_IN_PASS;
    return RND_SUCCESS;
}

int rnd_cmd_set_viewport(rnd_cmd cmd, float x, float y, float w, float h, float min_depth, float max_depth)
{
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd is null");
    if (cmd->state != CMD_IN_PASS && cmd->state != CMD_RECORDING) return fail(RND_ERROR_RECORDING, "set_viewport outside recording state");
    if (!cmd->draws.empty()) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "one viewport per pass: set it before the first draw");
    if (!(w > 0.0f) || !(h > 0.0f) || !(min_depth >= 0.0f && min_depth <= 1.0f) || !(max_depth >= 0.0f && max_depth <= 1.0f))
        return fail(RND_ERROR_INVALID_ARGUMENT, "viewport extent or depth range invalid");
    /* window coordinates are held in 24.8 with the outcode in the low byte of x: the
     * guard band (2x the half extent) must stay inside +-32767 pixels */
    if (!(fabsf(x + 0.5f * w) + w + 1.0f < 32767.0f) || !(fabsf(y + 0.5f * h) + h + 1.0f < 32767.0f))
        return fail(RND_ERROR_FEATURE_NOT_PRESENT, "viewport too large or too far from the origin");
    cmd->viewport[0] = x; cmd->viewport[1] = y; cmd->viewport[2] = w; cmd->viewport[3] = h; cmd->viewport[4] = min_depth; cmd->viewport[5] = max_depth;
    cmd->has_viewport = true;
    return RND_SUCCESS;
}

int rnd_cmd_set_scissor(rnd_cmd cmd, int32_t x, int32_t y, uint32_t w, uint32_t h)
{
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd is null");
    if (cmd->state != CMD_IN_PASS && cmd->state != CMD_RECORDING) return fail(RND_ERROR_RECORDING, "set_scissor outside recording state");
    if (!cmd->draws.empty()) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "one scissor per pass: set it before the first draw");
    if (x < 0 || y < 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) return fail(RND_ERROR_INVALID_ARGUMENT, "scissor offset must be non-negative");
    cmd->scissor[0] = x; cmd->scissor[1] = y; cmd->scissor[2] = (int32_t)w; cmd->scissor[3] = (int32_t)h;
    cmd->has_scissor = true;
    return RND_SUCCESS;
}

int rnd_cmd_bind_pipeline(rnd_cmd cmd, rnd_pipeline p)
{
    if (!cmd || !p) return fail(RND_ERROR_INVALID_ARGUMENT, "bind_pipeline: bad arguments");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "bind_pipeline outside a render pass");
    cmd->cur_pipeline = p;
    return RND_SUCCESS;
}

int rnd_cmd_bind_vertex_buffer(rnd_cmd cmd, rnd_buffer buf, uint64_t byte_offset)
{
    if (!cmd || !buf) return fail(RND_ERROR_INVALID_ARGUMENT, "bind_vertex_buffer: bad arguments");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "bind_vertex_buffer outside a render pass");
    if (byte_offset > buf->bytes || (byte_offset & 3u)) return fail(RND_ERROR_INVALID_ARGUMENT, "vertex buffer offset outside the buffer or not a multiple of 4");
    cmd->cur_vb = buf; cmd->cur_vb_offset = byte_offset;
    return RND_SUCCESS;
}

int rnd_cmd_bind_index_buffer(rnd_cmd cmd, rnd_buffer buf, uint64_t byte_offset, int32_t index_type)
{
    if (!cmd || !buf) return fail(RND_ERROR_INVALID_ARGUMENT, "bind_index_buffer: bad arguments");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "bind_index_buffer outside a render pass");
    if (index_type != RND_INDEX_TYPE_UINT32) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "index type must be UINT32");
    if (byte_offset > buf->bytes || (byte_offset & 3u)) return fail(RND_ERROR_INVALID_ARGUMENT, "index buffer offset outside the buffer or not a multiple of 4");
    cmd->cur_ib = buf; cmd->cur_ib_offset = byte_offset;
    return RND_SUCCESS;
}

int rnd_cmd_bind_uniform(rnd_cmd cmd, uint32_t set, uint32_t binding, rnd_buffer buf, uint64_t offset, uint64_t range)
{
    if (!cmd || !buf) return fail(RND_ERROR_INVALID_ARGUMENT, "bind_uniform: bad arguments");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "bind_uniform outside a render pass");
    if (set != 0 || binding != 0) return fail(RND_ERROR_INVALID_ARGUMENT, "the uniform block is set 0 binding 0");
    if ((offset & 3u) || range < 320 || offset > buf->bytes || buf->bytes - offset < 320)
        return fail(RND_ERROR_INVALID_ARGUMENT, "the uniform block needs 320 bytes (five mat4)");
    cmd->cur_ubo = buf; cmd->cur_ubo_offset = offset;
    return RND_SUCCESS;
}

int rnd_cmd_bind_texture(rnd_cmd cmd, uint32_t set, uint32_t binding, rnd_texture tex)
{
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "bind_texture: bad arguments");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "bind_texture outside a render pass");
    if (!((set == 0 && binding == 1) || (set == 1 && binding == 0))) return fail(RND_ERROR_INVALID_ARGUMENT, "the sampler is set 0 binding 1 or set 1 binding 0");
    cmd->cur_tex = tex;
    return RND_SUCCESS;
}

int rnd_cmd_push_constants(rnd_cmd cmd, uint32_t offset, uint32_t size, const void* data)
{
    if (!cmd || !data) return fail(RND_ERROR_INVALID_ARGUMENT, "push_constants: bad arguments");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "push_constants outside a render pass");
    if ((offset & 3u) || (size & 3u) || offset + size > 128) return fail(RND_ERROR_INVALID_ARGUMENT, "push constant range must lie in 0..128 and be 4-byte aligned");
    memcpy(reinterpret_cast<uint8_t*>(cmd->cur_push) + offset, data, size);
    return RND_SUCCESS;
}

int rnd_cmd_draw_indexed(rnd_cmd cmd, uint32_t index_count, uint32_t instance_count, uint32_t first_index,
                         int32_t vertex_offset, uint32_t first_instance)
{
    (void)first_instance;
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd is null");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "draw_indexed outside a render pass");
    if (!cmd->cur_pipeline || !cmd->cur_vb || !cmd->cur_ib || !cmd->cur_ubo) return fail(RND_ERROR_RECORDING, "draw_indexed needs a pipeline, vertex, index and uniform buffer bound");
    if (vertex_offset < 0) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "negative vertexOffset is not supported");
    if (instance_count == 0 || index_count < 3) return RND_SUCCESS;
    /* the shaders never read gl_InstanceIndex: every instance draws the same fragments, so one is recorded */
    uint64_t ib_avail = (cmd->cur_ib->bytes - cmd->cur_ib_offset) / 4;
    if ((uint64_t)first_index + index_count > ib_avail) return fail(RND_ERROR_INVALID_ARGUMENT, "draw reads past the index buffer");
    RecordedDraw d;
    d.pipeline = cmd->cur_pipeline;
    d.vb = cmd->cur_vb; d.vb_offset = cmd->cur_vb_offset;
    d.ib = cmd->cur_ib; d.ib_offset = cmd->cur_ib_offset;
    d.ubo = cmd->cur_ubo; d.ubo_offset = cmd->cur_ubo_offset;
    d.tex = cmd->cur_tex;
    memcpy(d.push, cmd->cur_push, sizeof(d.push));
    d.index_count = index_count; d.first_index = first_index; d.vertex_offset = vertex_offset;
    cmd->draws.push_back(d);
    return RND_SUCCESS;
}

int rnd_cmd_end_pass(rnd_cmd cmd)
{
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd is null");
    if (cmd->state != CMD_IN_PASS) return fail(RND_ERROR_RECORDING, "end_pass outside a render pass");
    cmd->state = CMD_RECORDING;
    return RND_SUCCESS;
}

int rnd_cmd_set_tile_shard(rnd_cmd cmd, uint32_t shard_index, uint32_t shard_count)
{
    if (!cmd || !shard_count || shard_index >= shard_count) return fail(RND_ERROR_INVALID_ARGUMENT, "set_tile_shard: bad arguments");
    cmd->shard_index = shard_index; cmd->shard_count = shard_count;
    /* interleaved shards count tiles of 32 rows: a command buffer already compiled with taller tiles is compiled again */
    if (shard_count != 1 && cmd->state == CMD_EXECUTABLE && cmd->fp.tile_h != (uint32_t)TILE_H) {
        if (int r = set_device(cmd->dev)) return r;
        HIP_TRY(sync_all(cmd->dev));
        cmd->state = CMD_RECORDING;
        return rnd_cmd_end(cmd);
    }
    return RND_SUCCESS;
}

int rnd_cmd_visible_draws(rnd_cmd cmd, uint32_t* out_count)
{
    if (!cmd || !out_count) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd_visible_draws: bad arguments");
    if (cmd->state != CMD_EXECUTABLE) return fail(RND_ERROR_RECORDING, "command buffer is not in the executable state");
    if (int r = set_device(cmd->dev)) return r;
    HIP_TRY(sync_all(cmd->dev));
    std::vector<DrawDerived> dd(cmd->d_derived.count);
    if (!dd.empty()) HIP_TRY(hipMemcpy(dd.data(), cmd->d_derived.ptr, dd.size() * sizeof(DrawDerived), hipMemcpyDeviceToHost));
    uint32_t n = 0;
    for (const DrawDerived& d : dd) n += d.visible ? 1u : 0u;
    *out_count = n;
    return RND_SUCCESS;
}

/* Turns the recorded draws into device tables.  Runs once per recording, like the
 * reference's record_command_buffers (recorded once, submitted every frame). */
int rnd_cmd_end(rnd_cmd cmd)
{
    if (!cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "cmd is null");
    if (cmd->state != CMD_RECORDING) return fail(RND_ERROR_RECORDING, "cmd_end outside recording state (pass still open?)");
    if (!cmd->has_pass) return fail(RND_ERROR_RECORDING, "command buffer holds no render pass");
    if (!cmd->draws.empty() && (!cmd->has_viewport || !cmd->has_scissor)) return fail(RND_ERROR_RECORDING, "viewport and scissor are dynamic state: set both");
    rnd_device dev = cmd->dev;
    rnd_target tg = cmd->target;
    if (int r = set_device(dev)) return r;

    /* depth state is one per pass (all pipelines of the samples share it) */
    int depth_test = 1, depth_less = 0;
    for (size_t i = 0; i < cmd->draws.size(); ++i) {
        const rnd_pipeline_info& pi = cmd->draws[i].pipeline->info;
        if (pi.samples != tg->samples) return fail(RND_ERROR_INVALID_ARGUMENT, "pipeline sample count differs from the target's");
        int dt = pi.depth_test ? 1 : 0, dl = (pi.depth_test && pi.depth_compare == RND_COMPARE_LESS) ? 1 : 0;
        if (i == 0) { depth_test = dt; depth_less = dl; }
        else if (dt != depth_test || dl != depth_less) return fail(RND_ERROR_FEATURE_NOT_PRESENT, "all pipelines of a pass must share the depth state");
    }

    const size_t n = cmd->draws.size();
    std::vector<DrawStatic> ds(n);
    std::vector<rnd_texture> tex_list;
    std::vector<IndexRangeJob> jobs;
    std::vector<uint32_t> job_of(n);
    std::map<std::pair<const uint32_t*, uint32_t>, uint32_t> job_map;
    for (size_t i = 0; i < n; ++i) {
        const RecordedDraw& rd = cmd->draws[i];
        const rnd_pipeline_info& pi = rd.pipeline->info;
        DrawStatic& d = ds[i];
        memset(&d, 0, sizeof(d));
        uint64_t vb_first = rd.vb_offset + (uint64_t)rd.vertex_offset * pi.stride;
        uint64_t vb_avail = vb_first <= rd.vb->bytes ? rd.vb->bytes - vb_first : 0;
        d.vb = rd.vb->ptr + (vb_first <= rd.vb->bytes ? vb_first : 0);
        uint64_t cap = vb_avail / pi.stride;
        d.vb_capacity = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
        d.ib = reinterpret_cast<const uint32_t*>(rd.ib->ptr + rd.ib_offset) + rd.first_index;
        d.ubo = reinterpret_cast<const float*>(rd.ubo->ptr + rd.ubo_offset);
        d.stride = pi.stride;
        d.off_pos = (int32_t)pi.position.offset;
        d.off_nrm = pi.normal.format ? (int32_t)pi.normal.offset : -1;
        d.off_uv = pi.uv.format ? (int32_t)pi.uv.offset : -1;
        d.tri_count = rd.index_count / 3;
        d.shader = pi.shader;
        d.cull_mode = (uint32_t)pi.cull_mode;
        d.front_face = (uint32_t)pi.front_face;
        memcpy(d.push, rd.push, sizeof(d.push));
        d.texture = -1;
        if (rd.tex) {
            size_t k = 0;
            while (k < tex_list.size() && tex_list[k] != rd.tex) ++k;
            if (k == tex_list.size()) tex_list.push_back(rd.tex);
            d.texture = (int32_t)k;
        }
        auto key = std::make_pair(d.ib, d.tri_count * 3);
        auto it = job_map.find(key);
        if (it == job_map.end()) {
            it = job_map.emplace(key, (uint32_t)jobs.size()).first;
            jobs.push_back(IndexRangeJob{d.ib, d.tri_count * 3, 0});
        }
        job_of[i] = it->second;
    }

    /* how many vertices each draw references: largest index + 1, bounded by the buffer */
    std::vector<uint32_t> job_max(jobs.size(), 0);
    if (!jobs.empty()) {
        DeviceArray<IndexRangeJob> d_jobs;
        DeviceArray<uint32_t> d_max;
        HIP_TRY(d_jobs.upload(jobs, dev->stream));
        HIP_TRY(d_max.resize(jobs.size()));
        HIP_TRY(hipMemsetAsync(d_max.ptr, 0, jobs.size() * 4, dev->stream));
        for (size_t b = 0; b < jobs.size(); b += 32768) {
            uint32_t cnt = (uint32_t)std::min<size_t>(32768, jobs.size() - b);
            launch_index_range(d_jobs.ptr + b, d_max.ptr + b, cnt, dev->stream);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(dev->stream));
        HIP_TRY(hipMemcpy(job_max.data(), d_max.ptr, jobs.size() * 4, hipMemcpyDeviceToHost));
        d_jobs.release(); d_max.release();
    }

    for (size_t i = 0; i < n; ++i) {
        DrawStatic& d = ds[i];
        uint64_t want = (uint64_t)job_max[job_of[i]] + 1;
        d.vert_count = d.tri_count ? (uint32_t)std::min<uint64_t>(want, d.vb_capacity) : 0;
    }

    /* The box of every draw's vertices, for culling whole draws when the work executes.  A pass of one
     * draw gets none (that draw is the pass).  Draws over the same vertices share one job. */
    cmd->geometry_epoch = dev->geometry_epoch;
    cmd->index_epoch = dev->index_epoch;
    if (n >= 2) {
        std::vector<BoundsJob> bjobs;
        std::vector<uint32_t> bjob_of(n, 0xFFFFFFFFu);
        std::map<std::tuple<const uint8_t*, uint32_t, int32_t, uint32_t>, uint32_t> bjob_map;
        for (size_t i = 0; i < n; ++i) {
            const DrawStatic& d = ds[i];
            if (!d.vert_count) continue;
            auto key = std::make_tuple(d.vb, d.stride, d.off_pos, d.vert_count);
            auto it = bjob_map.find(key);
            if (it == bjob_map.end()) {
                it = bjob_map.emplace(key, (uint32_t)bjobs.size()).first;
                bjobs.push_back(BoundsJob{d.vb, d.stride, (uint32_t)d.off_pos, d.vert_count, 0});
            }
            bjob_of[i] = it->second;
        }
        if (!bjobs.empty()) {
            std::vector<uint32_t> boxes(bjobs.size() * 6);
            for (size_t j = 0; j < bjobs.size(); ++j)
                for (int k = 0; k < 6; ++k) boxes[j * 6 + k] = k < 3 ? 0xFFFFFFFFu : 0u;
            DeviceArray<BoundsJob> d_bjobs;
            DeviceArray<uint32_t> d_boxes;
            HIP_TRY(d_bjobs.upload(bjobs, dev->stream));
            HIP_TRY(d_boxes.upload(boxes, dev->stream));
            for (size_t b = 0; b < bjobs.size(); b += 32768) {
                uint32_t cnt = (uint32_t)std::min<size_t>(32768, bjobs.size() - b);
                launch_bounds(d_bjobs.ptr + b, d_boxes.ptr + b * 6, cnt, dev->stream);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(dev->stream));
            HIP_TRY(hipMemcpy(boxes.data(), d_boxes.ptr, boxes.size() * 4, hipMemcpyDeviceToHost));
            d_bjobs.release(); d_boxes.release();
            auto from_ordered = [](uint32_t o) -> float {
                uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
                float f; memcpy(&f, &b, 4); return f;
            };
            for (size_t i = 0; i < n; ++i) {
                if (bjob_of[i] == 0xFFFFFFFFu) continue;
                const uint32_t* bx = &boxes[(size_t)bjob_of[i] * 6];
                bool ok = true;
                for (int k = 0; k < 3; ++k) {
                    if (bx[k] == 0xFFFFFFFFu && bx[3 + k] == 0u) { ok = false; break; } /* no vertex without a NaN */
                    float lo = from_ordered(bx[k]), hi = from_ordered(bx[3 + k]);
                    if (!std::isfinite(lo) || !std::isfinite(hi)) { ok = false; break; }
                    ds[i].bounds_min[k] = lo; ds[i].bounds_max[k] = hi;
                }
                ds[i].has_bounds = ok ? 1u : 0u;
            }
        }
    }

    std::vector<Chunk> vchunks, tchunks;
    uint64_t svtx_total = 0, prim_total = 0;
    for (size_t i = 0; i < n; ++i) {
        DrawStatic& d = ds[i];
        if (svtx_total + d.vert_count > 0xFFFFFFFFull) return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "pass references more than 2^32 vertices");
        d.svtx_base = (uint32_t)svtx_total;
        svtx_total += d.vert_count;
        d.prim_base = (uint32_t)prim_total;
        d.first_tchunk = (uint32_t)tchunks.size();
        prim_total += d.tri_count;
        for (uint32_t v = 0; v < d.vert_count; v += CHUNK) vchunks.push_back(Chunk{(uint32_t)i, v});
        for (uint32_t t = 0; t < d.tri_count; t += CHUNK) tchunks.push_back(Chunk{(uint32_t)i, t});
    }
    if (tchunks.size() >= (1u << 26)) return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "pass holds more than 2^32 primitive slots");

    std::vector<TexDesc> td(tex_list.size());
    for (size_t k = 0; k < tex_list.size(); ++k) {
        rnd_texture t = tex_list[k];
        memset(&td[k], 0, sizeof(TexDesc));
        for (uint32_t l = 0; l < t->levels; ++l) td[k].level[l] = t->level_ptr[l];
        td[k].width = t->width; td[k].height = t->height; td[k].levels = t->levels;
        td[k].srgb = t->srgb; td[k].tiled = t->sampler.tiled ? 1u : 0u;
        td[k].filter_linear = t->sampler.filtering == RND_FILTER_NEAREST ? 0u : 1u;
    }

    /* Rows per tile.  Taller tiles mean fewer workgroups to start and a shorter second round of them, but 40 rows of
     * sample words leave LDS for four workgroups on a CU instead of five.  Measured, 32 rows against 40: the one-draw
     * mesh at 1920x1080 with a triangle per two pixels 41.1 -> 40.1 us; the 4K pass of 10 000 small draws 250 -> 258 us;
     * the mesh of five triangles per pixel 133.9 -> 139.7 us; 1280x720 13.3 -> 14.5 us.  So 40 rows only for a pass that
     * is one draw (a mesh: triangles of one size all over the target) on a target of 1080p or more with at most two
     * triangles per pixel; RND_TILE_HEIGHT=32|40 overrides.  A command buffer whose frame is divided among devices by
     * interleaved tiles keeps the 32 rows its peers' composition counts on. */
    uint32_t tile_h = TILE_H;
    {
        const uint64_t pixels = (uint64_t)tg->width * tg->height;
        const char* th = getenv("RND_TILE_HEIGHT");
        if (th && (atoi(th) == TILE_H || atoi(th) == TILE_H_TALL)) tile_h = (uint32_t)atoi(th);
        else if (n == 1 && pixels >= 1920ull * 1080ull && prim_total <= 2 * pixels) tile_h = TILE_H_TALL;
        if (tg->samples != 1 || cmd->shard_count != 1) tile_h = TILE_H;
    }
    const uint32_t tiles_x = (tg->width + TILE_W - 1) / TILE_W, tiles_y = (tg->height + tile_h - 1) / tile_h;
    const size_t n_tiles = (size_t)tiles_x * tiles_y;
    const size_t slots = tchunks.size() * CHUNK;
    if (slots > SLOT_MASK) return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "pass holds more than 2^28 primitive slots");
    if (n > SLOT_MASK) return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "pass holds more than 2^28 draws");
    /* Fixed-capacity tile lists: the largest power of two that keeps all lists within 2 GiB
     * (HBM is 288 GB; only the entries written are ever touched).  Spill goes to the pass-wide list. */
    size_t tile_cap = 16384;
    while (tile_cap > 256 && tile_cap * n_tiles * sizeof(TileEntry) > (size_t(2) << 30)) tile_cap >>= 1;
    while (tile_cap > 256 && tile_cap / 2 >= slots) tile_cap >>= 1;
    const size_t large_cap = 2 * slots + 65536;
    const size_t clip_tri_cap = std::min<size_t>(MAX_FAN * slots, 2 * slots + 65536) + 1;

    cmd->release_compiled();
    HIP_TRY(cmd->d_draws.upload(ds, cmd->stream));
    HIP_TRY(cmd->d_derived.resize(n));
    if (n) HIP_TRY(zero_now(cmd->d_derived.ptr, n * sizeof(DrawDerived), cmd->stream)); /* a draw without vertices is never marked visible */
    HIP_TRY(cmd->d_vchunks.upload(vchunks, cmd->stream));
    HIP_TRY(cmd->d_tchunks.upload(tchunks, cmd->stream));
    /* RND_COMPACT=0: a pass of many draws keeps one wave per chunk of every draw, visible or not (for A/B runs) */
    const char* compact_env = getenv("RND_COMPACT");
    const bool compact = n >= COMPACT_MIN_DRAWS && !(compact_env && atoi(compact_env) == 0);
    if (compact) {
        HIP_TRY(cmd->d_visible_vchunks.resize(vchunks.size()));
        HIP_TRY(cmd->d_visible_tchunks.resize(tchunks.size()));
    }
    HIP_TRY(cmd->d_textures.upload(td, cmd->stream));
    HIP_TRY(cmd->d_svtx.resize((size_t)svtx_total));
    HIP_TRY(cmd->d_svary.resize((size_t)svtx_total));
    HIP_TRY(cmd->d_prim_rec.resize(slots));
    HIP_TRY(cmd->d_tile_count.resize(n_tiles * TILE_COUNT_STRIDE));
    const size_t tile_stride = tile_cap + 16;
    HIP_TRY(cmd->d_tile_lists.resize(n_tiles * tile_stride));
    HIP_TRY(cmd->d_large_list.resize(large_cap));
    HIP_TRY(cmd->d_clip_tris.resize(clip_tri_cap));
    HIP_TRY(cmd->d_counters.resize(CTR_COUNT));
    /* every frame leaves its counters at zero for the next one; the first frame needs them so */
    HIP_TRY(zero_now(cmd->d_tile_count.ptr, n_tiles * TILE_COUNT_STRIDE * sizeof(uint32_t), cmd->stream));
    HIP_TRY(zero_now(cmd->d_counters.ptr, CTR_COUNT * sizeof(uint32_t), cmd->stream));
#ifdef RND_PROBES
    HIP_TRY(cmd->d_probe.resize(((size_t)tiles_x * (tiles_y + 8) + 64) * 16 + vchunks.size() / 4 + tchunks.size() / 4 + 64 + 8 * 64)); /* the tile kernel's grid is padded to whole groups of eight rows */
    HIP_TRY(zero_now(cmd->d_probe.ptr, cmd->d_probe.count * sizeof(unsigned long long), cmd->stream));
#endif

    FrameParams& fp = cmd->fp;
    memset(&fp, 0, sizeof(fp));
    fp.draws = cmd->d_draws.ptr; fp.derived = cmd->d_derived.ptr;
    fp.vchunks = cmd->d_vchunks.ptr; fp.tchunks = cmd->d_tchunks.ptr; fp.textures = cmd->d_textures.ptr;
    fp.visible_vchunks = cmd->d_visible_vchunks.ptr; fp.visible_tchunks = cmd->d_visible_tchunks.ptr;
    fp.compact = compact ? 1u : 0u;
    fp.svtx = cmd->d_svtx.ptr; fp.svary = cmd->d_svary.ptr; fp.prim_rec = cmd->d_prim_rec.ptr;
    fp.tile_count = cmd->d_tile_count.ptr; fp.tile_lists = cmd->d_tile_lists.ptr;
    fp.large_list = cmd->d_large_list.ptr; fp.clip_tris = cmd->d_clip_tris.ptr;
    fp.counters = cmd->d_counters.ptr;
    fp.probe_cycles = cmd->d_probe.ptr;
    fp.probe_kernel_row = tiles_x * (tiles_y + 8) + 8;
    fp.tile_capacity = (uint32_t)tile_cap; fp.tile_stride = (uint32_t)tile_stride; fp.large_capacity = (uint32_t)large_cap;
    fp.clip_tri_capacity = (uint32_t)clip_tri_cap;
    fp.n_draws = (uint32_t)n; fp.n_vchunks = (uint32_t)vchunks.size(); fp.n_tchunks = (uint32_t)tchunks.size();
    fp.n_textures = (uint32_t)td.size();
    pass_params(fp, dev, tg, cmd->area, cmd->has_viewport && cmd->has_scissor, cmd->viewport, cmd->scissor, cmd->clear_color, cmd->clear_depth,
                depth_test, depth_less, tile_h);
    /* an empty rectangle takes no fragment: the geometry kernels are not launched at all (the tile kernel
     * still clears the render area) */
    if (fp.clamp_x0 > fp.clamp_x1 || fp.clamp_y0 > fp.clamp_y1) fp.n_vchunks = fp.n_tchunks = 0;
    fp.direct_index = n == 1 ? 1u : 0u;
    if (n == 1) { fp.direct_ib = ds[0].ib; fp.direct_svtx_base = ds[0].svtx_base; }
    fp.pass_shader = n ? ds[0].shader : -1;
    fp.pass_texture = n ? ds[0].texture : -1;
    for (size_t i = 1; i < n; ++i) {
        if (ds[i].shader != fp.pass_shader) fp.pass_shader = -1;
        if (ds[i].texture != fp.pass_texture) fp.pass_texture = -2;
    }
    if (fp.pass_texture >= 0) fp.pass_tex = td[(size_t)fp.pass_texture];
    cmd->state = CMD_EXECUTABLE;
    return RND_SUCCESS;
}

int rnd_cmd_destroy(rnd_cmd cmd)
{
    if (!cmd) return RND_SUCCESS;
    (void)hipSetDevice(cmd->dev->ordinal);
    (void)sync_all(cmd->dev);
    if (cmd->cs) --cmd->cs->users; /* the stream stays in the device's pool */
    cmd->release_compiled();
    auto& list = cmd->dev->cmds;
    list.erase(std::remove(list.begin(), list.end(), cmd), list.end());
    --cmd->dev->live_objects;
    delete cmd;
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ submit */

int rnd_submit(rnd_device dev, rnd_cmd cmd, rnd_fence fence)
{
    if (!dev || !cmd) return fail(RND_ERROR_INVALID_ARGUMENT, "submit: bad arguments");
    if (cmd->dev != dev) return fail(RND_ERROR_INVALID_ARGUMENT, "command buffer belongs to another device");
    if (cmd->state != CMD_EXECUTABLE) return fail(RND_ERROR_RECORDING, "command buffer is not in the executable state");
    if (int r = set_device(dev)) return r;
    /* How many vertices a draw transforms is its largest index + 1, scanned when the command buffer was recorded.
     * Index data written since then may point past that: such a frame would silently drop primitives. */
    if (cmd->index_epoch != dev->index_epoch && !cmd->draws.empty())
        return fail(RND_ERROR_RECORDING, "an index buffer was written after this command buffer was recorded: record it again");
    FrameParams fp = cmd->fp;
    fp.shard_index = cmd->shard_index; fp.shard_count = cmd->shard_count;
    /* vertex data written since recording: the boxes no longer bound it, draw everything */
    fp.cull_draws = cmd->geometry_epoch == dev->geometry_epoch ? 1u : 0u;
    fp.list_parity = (uint32_t)(cmd->frames_submitted++ & 1u);
#ifdef RND_PROBES
    { const char* dbg = getenv("RND_DEBUG"); fp.debug_flags = dbg ? (uint32_t)atoi(dbg) : 0u; }
#else
    fp.debug_flags = 0u;
#endif
    hipStream_t s = cmd->stream;
    rnd_target tg = cmd->target;
    if (cmd->cs && cmd->cs->seen_seq != dev->upload_seq) {
        /* uploads and copies queued on the device's stream come first.  Each stream of the pool waits for what it
         * has not been ordered behind yet, not only the first to submit. */
        HIP_TRY(wait_for_stream(dev, s, cmd->cs->index, dev->stream, -1, dev->order, "uploads before a frame"));
        cmd->cs->seen_seq = dev->upload_seq;
        cmd->cs->last_wait = "the tail of the device's stream (uploads and copies before this frame)";
    }
    if (tg && tg->last_valid && tg->last_stream != s) {
        int producer_id = -1;
        for (CmdStream* cs : dev->cmd_streams) if (cs->stream == tg->last_stream) producer_id = cs->index;
        HIP_TRY(wait_for_stream(dev, s, cmd->cs ? cmd->cs->index : -1, tg->last_stream, producer_id, tg->order, "the target's last frame"));
        if (cmd->cs) cmd->cs->last_wait = "the last frame or copy of this frame's target, on another stream";
        else dev->dev_last_wait = "the last frame or copy of a frame's target, on another stream";
    }
    const bool prof = dev->profiling;
    if (prof) HIP_TRY(hipEventRecord(dev->ev[RND_STAGE_VERTEX], s));
    launch_vertex(fp, dev->pending, s); /* the pending uniform write stays pending: later frames carry it too */
    if (prof) HIP_TRY(hipEventRecord(dev->ev[RND_STAGE_SETUP], s));
    launch_setup(fp, s);
    if (prof) HIP_TRY(hipEventRecord(dev->ev[RND_STAGE_TILE], s));
    launch_tile(fp, s, dev->four_tile_workgroups && fp.samples == 1 ? TILE_LDS_PAD_FOUR : 0u);
    if (prof) { HIP_TRY(hipEventRecord(dev->ev[RND_STAGE_COUNT], s)); dev->stage_valid = true; }
    HIP_TRY(hipGetLastError());
    if (cmd->cs) {
        cmd->cs->submitted = true;
        if (tg) { tg->last_stream = s; tg->last_valid = true; }
    }
    if (fence) {
        HIP_TRY(hipEventRecord(fence->ev, s));
        fence->pending = true;
    }
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ one draw, nothing allocated */

int rnd_draw_indexed_noalloc(rnd_device dev, const rnd_noalloc_draw* dr, rnd_fence fence)
{
    if (!dev || !dr || !dr->target || !dr->pipeline || !dr->vertex_buffer || !dr->index_buffer || !dr->uniform_block)
        return fail(RND_ERROR_INVALID_ARGUMENT, "draw_indexed_noalloc: bad arguments");
    rnd_target tg = dr->target;
    const rnd_pipeline_info& pi = dr->pipeline->info;
    if (tg->dev != dev || dr->pipeline->dev != dev || dr->vertex_buffer->dev != dev || dr->index_buffer->dev != dev ||
        (dr->texture && dr->texture->dev != dev))
        return fail(RND_ERROR_INVALID_ARGUMENT, "draw_indexed_noalloc: an object belongs to another device");
    if (!dev->scratch) return fail(RND_ERROR_INITIALIZATION_FAILED, "the static scratch array has no address on this device");
    if (pi.samples != tg->samples) return fail(RND_ERROR_INVALID_ARGUMENT, "pipeline sample count differs from the target's");
    const uint32_t tiles_x = (tg->width + TILE_W - 1) / TILE_W, tiles_y = (tg->height + TILE_H - 1) / TILE_H;
    const uint32_t n_tiles = tiles_x * tiles_y;
    const uint32_t tri_count = dr->index_count / 3;
    if (n_tiles > NA_MAX_TILES)
        return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "draw_indexed_noalloc: the target has " + std::to_string(n_tiles) + " tiles of 32x32 pixels, the static scratch array holds lists for " + std::to_string(NA_MAX_TILES));
    if (dr->vertex_count > NA_MAX_VERTICES)
        return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "draw_indexed_noalloc: " + std::to_string(dr->vertex_count) + " vertices, the static scratch array holds " + std::to_string(NA_MAX_VERTICES));
    if (tri_count > NA_MAX_TRIANGLES)
        return fail(RND_ERROR_OUT_OF_DEVICE_MEMORY, "draw_indexed_noalloc: " + std::to_string(tri_count) + " triangles, the static scratch array holds " + std::to_string(NA_MAX_TRIANGLES));
    if ((dr->vertex_offset_bytes & 3u) || dr->vertex_offset_bytes > dr->vertex_buffer->bytes ||
        (dr->vertex_buffer->bytes - dr->vertex_offset_bytes) / pi.stride < dr->vertex_count)
        return fail(RND_ERROR_INVALID_ARGUMENT, "draw_indexed_noalloc: the vertices reach past the vertex buffer");
    if ((dr->index_offset_bytes & 3u) || dr->index_offset_bytes > dr->index_buffer->bytes ||
        (dr->index_buffer->bytes - dr->index_offset_bytes) / 4 < dr->index_count)
        return fail(RND_ERROR_INVALID_ARGUMENT, "draw_indexed_noalloc: the indices reach past the index buffer");
    const float* vp = dr->viewport;
    if (!(vp[2] > 0.0f) || !(vp[3] > 0.0f) || !(vp[4] >= 0.0f && vp[4] <= 1.0f) || !(vp[5] >= 0.0f && vp[5] <= 1.0f))
        return fail(RND_ERROR_INVALID_ARGUMENT, "viewport extent or depth range invalid");
    if (!(fabsf(vp[0] + 0.5f * vp[2]) + vp[2] + 1.0f < 32767.0f) || !(fabsf(vp[1] + 0.5f * vp[3]) + vp[3] + 1.0f < 32767.0f))
        return fail(RND_ERROR_FEATURE_NOT_PRESENT, "viewport too large or too far from the origin");
    if (dr->scissor[0] < 0 || dr->scissor[1] < 0 || dr->scissor[2] < 0 || dr->scissor[3] < 0)
        return fail(RND_ERROR_INVALID_ARGUMENT, "scissor offset and extent must be non-negative");
    if (int r = set_device(dev)) return r;

    uint8_t* base = dev->scratch;
    DrawStatic d;
    memset(&d, 0, sizeof(d));
    d.vb = dr->vertex_buffer->ptr + dr->vertex_offset_bytes;
    d.vb_capacity = dr->vertex_count;
    d.ib = reinterpret_cast<const uint32_t*>(dr->index_buffer->ptr + dr->index_offset_bytes);
    d.ubo = reinterpret_cast<const float*>(base + NoAllocLayout::ubo);
    d.stride = pi.stride;
    d.off_pos = (int32_t)pi.position.offset;
    d.off_nrm = pi.normal.format ? (int32_t)pi.normal.offset : -1;
    d.off_uv = pi.uv.format ? (int32_t)pi.uv.offset : -1;
    d.tri_count = tri_count;
    d.vert_count = tri_count ? dr->vertex_count : 0u;
    d.shader = pi.shader;
    d.texture = dr->texture ? 0 : -1;
    d.cull_mode = (uint32_t)pi.cull_mode;
    d.front_face = (uint32_t)pi.front_face;
    if (dr->push_constants) memcpy(d.push, dr->push_constants, sizeof(d.push));

    FrameParams fp;
    memset(&fp, 0, sizeof(fp));
    fp.draws = reinterpret_cast<const DrawStatic*>(base + NoAllocLayout::draw);
    fp.derived = reinterpret_cast<DrawDerived*>(base + NoAllocLayout::derived);
    fp.counters = reinterpret_cast<uint32_t*>(base + NoAllocLayout::counters);
    fp.tile_count = reinterpret_cast<uint32_t*>(base + NoAllocLayout::tile_count);
    fp.svtx = reinterpret_cast<SVtx*>(base + NoAllocLayout::svtx);
    fp.svary = reinterpret_cast<SVary*>(base + NoAllocLayout::svary);
    fp.prim_rec = reinterpret_cast<PrimRec*>(base + NoAllocLayout::prim_rec);
    fp.tile_lists = reinterpret_cast<TileEntry*>(base + NoAllocLayout::tile_lists);
    fp.large_list = reinterpret_cast<LargeEntry*>(base + NoAllocLayout::large_list);
    fp.clip_tris = reinterpret_cast<ClipTri*>(base + NoAllocLayout::clip_tris);
    fp.tile_capacity = NA_TILE_CAPACITY; fp.tile_stride = NA_TILE_CAPACITY + 16;
    fp.large_capacity = NA_LARGE_CAPACITY; fp.clip_tri_capacity = NA_CLIP_TRI_CAPACITY;
    /* one draw: its chunks are its vertices and triangles in order, no chunk tables (vchunks / tchunks stay null) */
    fp.n_draws = 1; fp.n_vchunks = (d.vert_count + CHUNK - 1) / CHUNK; fp.n_tchunks = (tri_count + CHUNK - 1) / CHUNK;
    fp.direct_index = 1u; fp.direct_ib = d.ib; fp.direct_svtx_base = d.svtx_base;
    const int depth_test = pi.depth_test ? 1 : 0, depth_less = (pi.depth_test && pi.depth_compare == RND_COMPARE_LESS) ? 1 : 0;
    const int32_t area[4] = {0, 0, (int32_t)tg->width, (int32_t)tg->height};
    pass_params(fp, dev, tg, area, true, vp, dr->scissor, dr->clear_color, dr->clear_depth, depth_test, depth_less, TILE_H);
    if (fp.clamp_x0 > fp.clamp_x1 || fp.clamp_y0 > fp.clamp_y1) fp.n_vchunks = fp.n_tchunks = 0;
    fp.pass_shader = d.shader;
    fp.pass_texture = d.texture;
    if (dr->texture) {
        rnd_texture t = dr->texture;
        for (uint32_t l = 0; l < t->levels; ++l) fp.pass_tex.level[l] = t->level_ptr[l];
        fp.pass_tex.width = t->width; fp.pass_tex.height = t->height; fp.pass_tex.levels = t->levels;
        fp.pass_tex.srgb = t->srgb; fp.pass_tex.tiled = t->sampler.tiled ? 1u : 0u;
        fp.pass_tex.filter_linear = t->sampler.filtering == RND_FILTER_NEAREST ? 0u : 1u;
    }
    /* the uniform block travels in the vertex kernel's arguments, exactly like a small rnd_buffer_write */
    UniformPatch patch;
    patch.dst = base + NoAllocLayout::ubo;
    patch.bytes = 320; patch.pad = 0;
    memcpy(patch.words, dr->uniform_block, 320);

    hipStream_t s = dev->stream;
    HIP_TRY(follow_frames(dev));              /* frames of command buffers already submitted come first */
    launch_noalloc_begin(d, base, n_tiles, s);
    launch_vertex(fp, patch, s);
    launch_setup(fp, s);
    launch_tile(fp, s);
    HIP_TRY(hipGetLastError());
    ++dev->upload_seq;                        /* and later frames of command buffers follow this one */
    tg->last_stream = s; tg->last_valid = true;
    if (fence) {
        HIP_TRY(hipEventRecord(fence->ev, s));
        fence->pending = true;
    }
    return RND_SUCCESS;
}

/* ------------------------------------------------------------------ fences */

int rnd_fence_create(rnd_device dev, rnd_fence* out)
{
    if (!dev || !out) return fail(RND_ERROR_INVALID_ARGUMENT, "fence_create: bad arguments");
    if (int r = set_device(dev)) return r;
    rnd_fence f = new rnd_fence_t{dev, nullptr, false};
    hipError_t e = hipEventCreateWithFlags(&f->ev, hipEventDisableTiming);
    if (e != hipSuccess) { delete f; return fail(RND_ERROR_DEVICE_LOST, "hipEventCreate failed"); }
    ++dev->live_objects;
    *out = f;
    return RND_SUCCESS;
}

int rnd_fence_wait(rnd_fence fence)
{
    if (!fence) return fail(RND_ERROR_INVALID_ARGUMENT, "fence is null");
    if (!fence->pending) return RND_SUCCESS; /* never submitted: nothing to wait for (the reference only waits on submitted fences, vk_app.cpp:435-444) */
    if (int r = set_device(fence->dev)) return r;
    HIP_TRY(hipEventSynchronize(fence->ev));
    fence->pending = false;
    return report_overflow(fence->dev);
}

int rnd_fence_destroy(rnd_fence fence)
{
    if (!fence) return RND_SUCCESS;
    (void)hipSetDevice(fence->dev->ordinal);
    (void)hipEventDestroy(fence->ev);
    --fence->dev->live_objects;
    delete fence;
    return RND_SUCCESS;
}

} // extern "C"
