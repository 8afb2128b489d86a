"""The C++ host mirror (renderer_amd/host): its glm restatement agrees with the numpy one, both
error-transport flavours fail loudly without a GPU, and on a GPU the headless test_ktx sample
produces the frame the oracle computes for the same scene."""
import os
import subprocess

import numpy as np
import pytest

from renderer_amd import hostmath as hm, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = os.path.join(ROOT, "samples", "test_ktx_app_headless")
SAMPLE_NOEXC = os.path.join(ROOT, "samples", "test_ktx_app_headless_noexcept")
SAMPLE_PUSH = os.path.join(ROOT, "samples", "test_push_constants_app_headless")
SAMPLE_VK = os.path.join(ROOT, "samples", "test_ktx_app_vk_headless")  # the same sample through the reference's own Vulkan names (include/rnd_vk.h)


def dump_ubo(w, h):
    out = subprocess.run([SAMPLE, "--dump-ubo", "--width", str(w), "--height", str(h)], check=True,
                         capture_output=True, text=True).stdout.split()
    bits = np.array([int(x, 16) for x in out], dtype=np.uint32)
    assert bits.size == 80 + 160
    return bits[:80].view(np.float32), bits[80:].view(np.float32).reshape(10, 16)


def ulp_diff(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib).max()


@pytest.mark.parametrize("wh", [(800, 600), (1280, 720)])
def test_cpp_matrices_match_numpy(wh):
    ubo, ring = dump_ubo(*wh)
    ref = hm.global_ubo(hm.Camera(), hm.scale(hm.identity(), [0.12, 0.12, 0.12]), wh[0], wh[1])
    assert np.allclose(ubo, ref, rtol=1e-6, atol=1e-7)
    assert ulp_diff(ubo[:16], ref[:16]) <= 2  # projection: cos/sin may differ in the last place
    assert np.array_equal(ubo[32:64], ref[32:64])  # view and model have no trigonometry
    for j in range(10):
        assert np.allclose(ring[j], hm.ring_transform(j).reshape(16), rtol=1e-6, atol=2e-6)


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure path")
@pytest.mark.parametrize("exe", [SAMPLE, SAMPLE_NOEXC, SAMPLE_VK])
def test_sample_fails_loudly_without_gpu(exe):
    r = subprocess.run([exe, "--frames", "1"], capture_output=True, text=True)
    assert r.returncode == 1
    assert "no HIP device is visible" in r.stderr and ("rnd_device_create" in r.stderr or "context::init" in r.stderr)


def fnv1a(data):
    h = 1469598103934665603
    for b in bytes(data):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def cpp_checker(size):
    y, x = np.mgrid[0:size, 0:size]
    c = (((x // (size // 16)) + (y // (size // 16))) & 1).astype(bool)
    t = np.zeros((size, size, 4), np.uint8)
    t[..., 0] = np.where(c, 230, 60)
    t[..., 1] = 40 + 200 * x // size
    t[..., 2] = 40 + 200 * y // size
    t[..., 3] = 255
    return t


@pytest.mark.gpu
@pytest.mark.parametrize("exe", [SAMPLE, SAMPLE_NOEXC, SAMPLE_PUSH, SAMPLE_VK])
def test_sample_frame_matches_oracle(oracle, exe):
    w, h = 320, 180
    out = subprocess.run([exe, "--frames", "1", "--width", str(w), "--height", str(h)], check=True,
                         capture_output=True, text=True).stdout
    fields = dict(f.split("=") for f in out.split())
    ubo, ring = dump_ubo(w, h)
    # the push-constants sample: the same ring, lit by the push colour instead of a texture
    sc = scenes.config0(w, h, tex_size=16, shader=scenes.SHADER_PUSH_CONSTANTS if exe == SAMPLE_PUSH else scenes.SHADER_TEST_KTX)
    sc.textures[0]["texels"] = cpp_checker(1024)
    sc.ubos[0] = ubo
    for j, d in enumerate(sc.draws):
        d["push"] = scenes.push_block(hm.RING_COLORS[j], ring[j])
    color, depth, _ = oracle.render_scene(sc, threads=4)
    assert int(fields["covered"]) == int((depth != 0xFFFFFF).sum()) > 0
    assert int(fields["depth_fnv"], 16) == fnv1a(depth.astype(np.uint32).tobytes())
    assert int(fields["color_fnv"], 16) == fnv1a(color.tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["samples4", "push_constants", "push_constants_samples4"])
def test_vulkan_named_sample_variants_match_oracle(oracle, variant):
    """The sample written against the reference's Vulkan names with the obj viewer's multisampled render pass (three
    attachments, resolve into the swapchain image: dummy_obj_viewer_app.cpp:461-639) and with the push-constants shader pair
    and its descriptor set without a sampler (test_push_constants_app.cpp)."""
    w, h = 256, 144
    samples = 4 if "samples4" in variant else 1
    push = "push_constants" in variant
    args = [SAMPLE_VK, "--frames", "1", "--width", str(w), "--height", str(h), "--samples", str(samples)]
    if push:
        args += ["--shader", "push_constants"]
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
    fields = dict(f.split("=") for f in out.split())
    ubo, ring = dump_ubo(w, h)
    sc = scenes.config0(w, h, tex_size=16, samples=samples, shader=scenes.SHADER_PUSH_CONSTANTS if push else scenes.SHADER_TEST_KTX)
    sc.textures[0]["texels"] = cpp_checker(1024)
    sc.ubos[0] = ubo
    for j, d in enumerate(sc.draws):
        d["push"] = scenes.push_block(hm.RING_COLORS[j], ring[j])
    color, depth, _ = oracle.render_scene(sc, threads=4)
    assert int(fields["covered"]) == int((depth != 0xFFFFFF).sum()) > 0
    assert int(fields["depth_fnv"], 16) == fnv1a(depth.astype(np.uint32).tobytes())
    assert int(fields["color_fnv"], 16) == fnv1a(color.tobytes())
