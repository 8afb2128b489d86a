"""Drives a Scene through the C ABI the way the reference's samples drive Vulkan:
create resources once (on_vulkan_initialized, test_ktx_app.cpp:46-56), record the
command buffer once (record_command_buffers, :627-746), then per frame write the
uniform block and submit (draw_frame, :66-73; vk_app.cpp:463-467)."""
import numpy as np

from . import api


class SceneRunner:
    def __init__(self, scene, device=None, ordinal=0, stream=None):
        self.scene = scene
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
        self.target = api.Target(d, scene.width, scene.height, scene.color_format, scene.samples)
        self.cmd = api.CommandBuffer(d)
        self.fence = api.Fence(d)
        self.record()

    def record(self):
        sc, cmd = self.scene, self.cmd
        cmd.begin()
        cmd.begin_pass(self.target, sc.clear_color, sc.clear_depth, sc.area)
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
        self.dev.submit(self.cmd, self.fence if fence else None)

    def render(self):
        """One frame, finished on return; returns (colour HxWx4 u8, depth HxWxS u32)."""
        self.submit(fence=True)
        self.fence.wait()
        return self.target.read()

    def destroy(self):
        for h in [self.cmd, self.fence, self.target] + self.pipelines + self.textures + self.ubos + self.ibs + self.vbs:
            h.destroy()
        if self.own_device:
            self.dev.destroy()
