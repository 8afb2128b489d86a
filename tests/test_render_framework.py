"""The reference's render_framework seam with the HIP backend (renderer_amd/host/render_framework.hpp):
what the builders derive from a vertex format and a parameter list, and -- on a GPU -- that the
test_ktx scene drawn through meshes, parameters lists, textures and a material is the frame the
test_ktx sample itself produces (which tests/test_host_sample.py checks against the oracle)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RF = os.path.join(ROOT, "samples", "render_framework_headless")
RF_NOEXC = os.path.join(ROOT, "samples", "render_framework_headless_noexcept")
KTX = os.path.join(ROOT, "samples", "test_ktx_app_headless")


@pytest.mark.parametrize("exe", [RF, RF_NOEXC])
def test_builders_layout(exe):
    out = subprocess.run([exe, "--layout"], check=True, capture_output=True, text=True).stdout.splitlines()
    # attribute i sits at the sum of the sizes before it (vk_mesh.cpp:221-252): pos3 | normal3 | uv2
    assert out[0] == "stride=32 position=0 normal=12 uv=24"
    # one slot per parameter, slots aligned to 16 bytes (vk_parameter.cpp:12, 226-248): int32, float32, vec4, mat4
    assert out[1].strip() == "list_size=112 slots=0:4 16:4 32:16 48:64"
    assert out[2] == "vec4_slot=1,2,3,4"           # load_value reaches the list through the listener
    assert out[3].startswith("error=bad elements count.")  # parameter.cpp: the builder's own check


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU failure path")
def test_fails_loudly_without_gpu():
    r = subprocess.run([RF], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device is visible" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("exe", [RF, RF_NOEXC])
def test_scene_through_the_framework_is_the_sample_frame(exe):
    args = ["--width", "640", "--height", "360"]
    a = subprocess.run([exe] + args, check=True, capture_output=True, text=True).stdout.split()
    b = subprocess.run([KTX, "--frames", "1"] + args, check=True, capture_output=True, text=True).stdout.split()
    fa, fb = dict(f.split("=") for f in a), dict(f.split("=") for f in b)
    assert int(fa["covered"]) > 10000
    assert fa == fb
