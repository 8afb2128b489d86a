"""The C ABI library loads without a GPU and exports every symbol include/renderer_amd.h
declares; with no device it fails loudly instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from renderer_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "renderer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rnd_[a-z0-9_]+)\s*\(", text)))



def _has_gpu():
    return os.path.exists("/dev/kfd")

def test_header_symbols_are_exported():
    names = declared_symbols()
    assert len(names) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert sorted(_lib.SYMBOLS) == names


def test_vulkan_named_entry_points_are_exported():
    """include/rnd_vk.h: every vk*, vma* and rndVk* function it declares is exported by librnd_vk.so, which loads without
    a GPU (its structure sizes are checked against the Vulkan ABI at compile time, renderer_amd/csrc/rnd_vk.cpp)."""
    text = open(os.path.join(ROOT, "include", "rnd_vk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b((?:vk|vma|rndVk)[A-Z][A-Za-z0-9]+)\s*\(", text)))
    assert len(names) >= 50 and "vkCmdDrawIndexed" in names and "vkQueueSubmit" in names and "vmaFlushAllocation" in names
    lib = ctypes.CDLL(os.path.join(ROOT, "renderer_amd", "librnd_vk.so"))
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    lib.rndVkCreateDevice.restype = ctypes.c_int
    dev = ctypes.c_void_p()
    if not _has_gpu():
        assert lib.rndVkCreateDevice(0, ctypes.byref(dev), None, None) == -3  # VK_ERROR_INITIALIZATION_FAILED
        lib.rndVkLastError.restype = ctypes.c_char_p
        assert b"HIP device" in lib.rndVkLastError()


def test_binding_loads_and_types_are_declared():
    lib = _lib.load()
    for n in _lib.SYMBOLS:
        assert getattr(lib, n).argtypes is not None
    assert ctypes.sizeof(_lib.PipelineInfo) == 4 + 3 * 8 + 4 * 10
    assert ctypes.sizeof(_lib.SamplerInfo) == 12


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure path")
def test_no_gpu_fails_loudly():
    with pytest.raises(api.RendererError) as e:
        api.Device(0)
    assert e.value.code == -3
    assert "HIP device" in e.value.msg


def test_null_handles_are_rejected_or_ignored():
    lib = _lib.load()
    assert lib.rnd_buffer_destroy(None) == 0 and lib.rnd_device_destroy(None) == 0   # destroy(NULL) is a no-op
    assert lib.rnd_cmd_destroy(None) == 0 and lib.rnd_texture_destroy(None) == 0
    assert lib.rnd_submit(None, None, None) == -1
    assert b"submit" in lib.rnd_last_error()
    assert lib.rnd_cmd_begin(None) == -1
    assert lib.rnd_fence_wait(None) == -1


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under renderer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "raster_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
