"""Drives a Scene through the C ABI the way the reference's samples drive Vulkan:
create resources once (on_vulkan_initialized, test_ktx_app.cpp:46-56), record the
command buffer once (record_command_buffers, :627-746), then per frame write the
uniform block and submit (draw_frame, :66-73; vk_app.cpp:463-467).

`frames_in_flight` plays the part of the swapchain image count: the reference records one
command buffer per image (test_ktx_app.cpp:639-744) and keeps that many frames in flight
(vk_app.cpp:109-130); here each gets its own target, and consecutive submits go to
consecutive command buffers so their frames may overlap on the GPU."""
import numpy as np

from . import api


class SceneRunner:
    def __init__(self, scene, device=None, ordinal=0, stream=None, frames_in_flight=1):
        self.scene = scene
        self.frames_in_flight = max(1, int(frames_in_flight))
        self.next_image = 0
        self.own_device = device is None
        self.dev = device or api.Device(ordinal, stream)
        d = self.dev
        self.vbs = [api.Buffer(d, api.BUFFER_USAGE_VERTEX, v) for v in scene.vbs]
        self.ibs = [api.Buffer(d, api.BUFFER_USAGE_INDEX, i) for i in scene.ibs]
        self.ubos = [api.Buffer(d, api.BUFFER_USAGE_UNIFORM, np.asarray(u, dtype=np.float32)) for u in scene.ubos]
        self.textures = []
        for t in scene.textures:
            self.textures.append(api.Texture(d, t["texels"], fmt=t.get("fmt", api.FORMAT_R8G8B8A8_SRGB),
                                             gen_mips=t.get("gen_mips", True), filtering=t.get("filtering", 1),
                                             tiled=t.get("tiled", False), levels=t.get("levels")))
        self.pipelines = []
        for p in scene.pipelines:
            self.pipelines.append(api.Pipeline(d, p["shader"], stride=p["stride"], off_pos=p["off_pos"],
                                               off_nrm=p["off_nrm"], off_uv=p["off_uv"], cull_mode=p["cull_mode"],
                                               front_face=p["front_face"], depth_test=p["depth_test"],
                                               depth_write=p["depth_write"], depth_compare=p["depth_compare"],
                                               samples=p["samples"]))
        self.targets = [api.Target(d, scene.width, scene.height, scene.color_format, scene.samples)
                        for _ in range(self.frames_in_flight)]
        self.cmds = [api.CommandBuffer(d) for _ in range(self.frames_in_flight)]
        self.target, self.cmd = self.targets[0], self.cmds[0]
        self.fence = api.Fence(d)
        for cmd, target in zip(self.cmds, self.targets):
            self.record(cmd, target)

    def record(self, cmd=None, target=None):
        sc = self.scene
        cmd = cmd or self.cmd
        target = target or self.target
        cmd.begin()
        cmd.begin_pass(target, sc.clear_color, sc.clear_depth, sc.area)
        cmd.set_viewport(*sc.viewport)
        cmd.set_scissor(*sc.scissor)
        last_pipeline = None
        for dr in sc.draws:
            cmd.push_constants(np.asarray(dr["push"], dtype=np.float32))
            if dr["pipeline"] != last_pipeline:
                cmd.bind_pipeline(self.pipelines[dr["pipeline"]])
                last_pipeline = dr["pipeline"]
            cmd.bind_uniform(self.ubos[dr["ubo"]])
            tex = dr.get("texture", -1)
            cmd.bind_texture(self.textures[tex] if tex is not None and tex >= 0 else None)
            cmd.bind_vertex_buffer(self.vbs[dr["vb"]], dr.get("vb_offset", 0))
            cmd.bind_index_buffer(self.ibs[dr["ib"]], dr.get("ib_offset", 0))
            cmd.draw_indexed(dr["index_count"], 1, dr.get("first_index", 0), dr.get("vertex_offset", 0), 0)
        cmd.end_pass()
        cmd.end()

    def set_ubo(self, index, data):
        self.ubos[index].write(np.asarray(data, dtype=np.float32))

    def submit(self, fence=False):
        """Queues one frame on the next image's command buffer; returns that image's index."""
        image = self.next_image
        self.next_image = (image + 1) % self.frames_in_flight
        self.dev.submit(self.cmds[image], self.fence if fence else None)
        return image

    def render(self):
        """One frame, finished on return; returns (colour HxWx4 u8, depth HxWxS u32)."""
        image = self.submit(fence=True)
        self.fence.wait()
        return self.targets[image].read()

    def destroy(self):
        for h in self.cmds + [self.fence] + self.targets + self.pipelines + self.textures + self.ubos + self.ibs + self.vbs:
            h.destroy()
        if self.own_device:
            self.dev.destroy()
