"""The step before the draw path (renderer_amd/host/obj_loader.hpp, ktx2.hpp, test_app.cpp): an OBJ
file becomes the interleaved vertex buffer, the local index buffer, the sub-geometry table and
the model transform the draw list binds (obj_loader.cpp:93-310); a KTX2 container becomes the
texture's levels (tools.cpp:1043-1334).  The reference's own assets are Git-LFS stubs, so the files
are written here.  CPU tests run the sample with --no-gpu; the GPU test renders the loaded model
through the C++ host and checks the frame against the oracle fed with the same buffers."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from renderer_amd import hostmath as hm, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "samples", "test_app_headless")
TOOL_NOEXC = os.path.join(ROOT, "samples", "test_app_headless_noexcept")
F = np.float32
KTX2_ID = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])


def fmt(x):
    return "%.9g" % float(x)  # nine significant digits round-trip a binary32


def write_obj(path, shapes, mtllib=None):
    """shapes: (name, rows[N,8] = pos3 nrm3 uv2, triangles as corner soup, material or None, style).
    Every face corner writes its own v / vn / vt lines, so equal corners appear many times and the
    loader has to merge them.  style 'abs': 1-based indices; 'rel': negative indices and one quad."""
    lines = []
    if mtllib:
        lines.append("mtllib " + mtllib)
    count = 0
    for name, rows, mat, style in shapes:
        lines.append("o " + name)
        if mat:
            lines.append("usemtl " + mat)
        if style == "abs":
            for tri in rows.reshape(-1, 3, 8):
                for r in tri:
                    lines.append("v " + " ".join(fmt(v) for v in r[0:3]))
                    lines.append("vn " + " ".join(fmt(v) for v in r[3:6]))
                    lines.append("vt " + " ".join(fmt(v) for v in r[6:8]))
                ids = [count + 1, count + 2, count + 3]
                count += 3
                lines.append("f " + " ".join("%d/%d/%d" % (i, i, i) for i in ids))
        else:  # four corners, one quad face addressed from the end of the lists
            assert rows.shape == (4, 8)
            for r in rows:
                lines.append("v " + " ".join(fmt(v) for v in r[0:3]))
                lines.append("vn " + " ".join(fmt(v) for v in r[3:6]))
                lines.append("vt " + " ".join(fmt(v) for v in r[6:8]))
            count += 4
            lines.append("f -4/-4/-4 -3/-3/-3 -2/-2/-2 -1/-1/-1")
    with open(path, "w") as f:
        f.write("# written by tests/test_loader.py\n" + "\n".join(lines) + "\n")


def merge(rows):
    """First-occurrence merge of equal rows: what init_obj_geometry's search does."""
    seen, out, idx = {}, [], []
    for r in rows:
        k = (r + F(0)).tobytes()
        if k not in seen:
            seen[k] = len(out)
            out.append(r)
        idx.append(seen[k])
    return np.array(out, dtype=F), np.array(idx, dtype=np.uint32)


def expected_transform(positions):
    """obj_loader.cpp:294-307 in binary32, with fabs."""
    mn, mx = positions.min(axis=0).astype(F), positions.max(axis=0).astype(F)
    offset = (mx - mn) * F(0.5)
    s = [max(abs(F(mn[k] - offset[k])), abs(F(mx[k] - offset[k]))) for k in range(3)]
    final = s[0] if s[0] > s[1] else s[1]
    final = F(1.0) / s[2] if s[2] > final else F(1.0) / final
    offset = (mn + (mx - mn) * F(0.5)) * final
    m = hm.translate(hm.identity(), offset)
    return hm.scale(m, [final, final, final])


def run_tool(args, exe=TOOL, check=True):
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    if check:
        assert r.returncode == 0, r.stderr
    return r


def two_shape_model(tmp_path, mtllib=None):
    v, i = scenes.icosphere(2, radius=0.8)
    soup = v[i]                                   # 960 corners, 162 distinct
    quad = np.array([[-1.5, -1.0, 0.5, 0, 0, -1, 0, 0], [1.5, -1.0, 0.5, 0, 0, -1, 1, 0],
                     [1.5, 1.2, 0.5, 0, 0, -1, 1, 1], [-1.5, 1.2, 0.5, 0, 0, -1, 0, 1]], dtype=F)
    path = str(tmp_path / "model.obj")
    write_obj(path, [("ball", soup, "paint" if mtllib else None, "abs"), ("wall", quad, None, "rel")], mtllib)
    return path, soup, quad


def load_dump(prefix):
    meta = json.load(open(prefix + ".json"))
    vb = np.fromfile(prefix + ".vb", dtype=F)
    ib = np.fromfile(prefix + ".ib", dtype=np.uint32)
    return meta, vb, ib


@pytest.mark.parametrize("exe", [TOOL, TOOL_NOEXC])
def test_obj_merge_layout_and_transform(tmp_path, exe):
    path, soup, quad = two_shape_model(tmp_path)
    prefix = str(tmp_path / "dump")
    run_tool(["--obj", path, "--no-gpu", "--dump", prefix], exe)
    meta, vb, ib = load_dump(prefix)
    ball_v, ball_i = merge(soup)
    assert ball_v.shape[0] == 162
    wall_i = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)  # the quad as a fan
    assert np.array_equal(vb, np.concatenate([ball_v.reshape(-1), quad.reshape(-1)]))
    assert np.array_equal(ib, np.concatenate([ball_i, wall_i]))
    sg = meta["sub_geometries"]
    assert [g["name"] for g in sg] == ["ball", "wall"]
    assert sg[0] == dict(name="ball", format=[106, 106, 103], indices_offset=0, indices_size=960, indices_bias=0,
                         vertices_offset=0, diffuse="")
    # vertices_offset is in BYTES, indices_offset in indices, indices stay local to the shape (obj_loader.cpp:173-205)
    assert sg[1] == dict(name="wall", format=[106, 106, 103], indices_offset=960, indices_size=6, indices_bias=162,
                         vertices_offset=162 * 32, diffuse="")
    got = np.array(meta["model_transform_bits"], dtype=np.uint32).view(F)
    want = expected_transform(np.concatenate([soup[:, 0:3], quad[:, 0:3]])).reshape(16)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    # the reference measures min and max against the half extent, not the centre (obj_loader.cpp:294-300):
    # x spans -1.5..1.5, its half extent is 1.5, |min - 1.5| = 3 is the largest of the three axes
    assert got[0] == got[5] == got[10] and abs(got[0] - 1.0 / 3.0) < 1e-6


def test_mtl_diffuse_texture_name(tmp_path):
    with open(tmp_path / "model.mtl", "w") as f:
        f.write("newmtl other\nmap_Kd other.png\nnewmtl paint\nKd 1 1 1\nmap_Kd -clamp on textures/diffuse.ktx2\n")
    path, _, _ = two_shape_model(tmp_path, mtllib="model.mtl")
    prefix = str(tmp_path / "dump")
    run_tool(["--obj", path, "--no-gpu", "--dump", prefix])
    sg = load_dump(prefix)[0]["sub_geometries"]
    # usemtl stays in force across "o" lines, as in the OBJ format: the wall is painted too
    assert sg[0]["diffuse"] == "textures/diffuse.ktx2" and sg[1]["diffuse"] == "textures/diffuse.ktx2"


@pytest.mark.parametrize("exe", [TOOL, TOOL_NOEXC])
@pytest.mark.parametrize("text,msg", [
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "bad face corner"),
    ("v 0 0 0\nv 1 0 0\nf 1 2\n", "fewer than three"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\n", "no faces"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3\n", "lacks an attribute"),
])
def test_obj_errors_are_reported(tmp_path, exe, text, msg):
    path = tmp_path / "bad.obj"
    path.write_text(text)
    r = run_tool(["--obj", str(path), "--no-gpu"], exe, check=False)
    assert r.returncode == 1 and msg in r.stderr


def write_ktx2(path, vk_format, width, height, levels, level_count=None, identifier=KTX2_ID, supercompression=0, truncate=0):
    """levels: list of uint8 arrays, level 0 first.  Header (80 bytes), level index, then the level
    data, smallest level first as the KTX 2.0 layout has it, each aligned to 4 bytes."""
    n = len(levels)
    index_at = 80
    data_at = index_at + 24 * n
    blobs, offsets, pos = [], [0] * n, data_at
    for l in reversed(range(n)):
        pos = (pos + 3) & ~3
        offsets[l] = pos
        blobs.append((pos, levels[l].tobytes()))
        pos += levels[l].nbytes
    out = bytearray(pos)
    out[0:12] = identifier
    struct.pack_into("<9I", out, 12, vk_format, 1, width, height, 0, 0, 1, n if level_count is None else level_count, supercompression)
    struct.pack_into("<4I2Q", out, 48, 0, 0, 0, 0, 0, 0)
    for l in range(n):
        struct.pack_into("<3Q", out, index_at + 24 * l, offsets[l], levels[l].nbytes, levels[l].nbytes)
    for at, b in blobs:
        out[at:at + len(b)] = b
    with open(path, "wb") as f:
        f.write(bytes(out[:len(out) - truncate]))
    return offsets


def rgba_levels(size, n, seed=3):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=(max(size >> l, 1), max(size >> l, 1), 4), dtype=np.uint8) for l in range(n)]


def tiny_obj(tmp_path):
    p = tmp_path / "tri.obj"
    p.write_text("v -1 -1 0\nv 1 -1 0\nv 0 1 0\nvn 0 0 -1\nvt 0 0\nvt 1 0\nvt 0.5 1\nf 1/1/1 2/2/1 3/3/1\n")
    return str(p)


def test_ktx2_stored_levels(tmp_path):
    levels = rgba_levels(32, 4)
    offsets = write_ktx2(tmp_path / "t.ktx2", 43, 32, 32, levels)
    prefix = str(tmp_path / "dump")
    run_tool(["--obj", tiny_obj(tmp_path), "--ktx", str(tmp_path / "t.ktx2"), "--no-gpu", "--dump", prefix])
    k = load_dump(prefix)[0]["ktx"]
    assert k == dict(vk_format=43, width=32, height=32, stored_levels=4, texture_levels=4, gen_mips=False, level_offsets=offsets)


def test_ktx2_level_count_zero_asks_for_generated_mips(tmp_path):
    write_ktx2(tmp_path / "t.ktx2", 37, 64, 32, [np.zeros((32, 64, 4), np.uint8)], level_count=0)
    prefix = str(tmp_path / "dump")
    run_tool(["--obj", tiny_obj(tmp_path), "--ktx", str(tmp_path / "t.ktx2"), "--no-gpu", "--dump", prefix])
    k = load_dump(prefix)[0]["ktx"]
    assert k["gen_mips"] is True and k["stored_levels"] == 1
    assert k["texture_levels"] == 6  # (uint32)log2(max(64, 32)), tools.cpp:1076


@pytest.mark.parametrize("kw,msg", [
    (dict(identifier=b"\xABKTX 11\xBB\r\n\x1A\n"), "bad ktx file."),
    (dict(vk_format=97), "unsupported vk format."),
    (dict(supercompression=2), "supercompressed"),
    (dict(truncate=100), "lies outside the file"),
])
def test_ktx2_errors_are_reported(tmp_path, kw, msg):
    args = dict(vk_format=43)
    args.update(kw)
    write_ktx2(tmp_path / "t.ktx2", args.pop("vk_format"), 16, 16, rgba_levels(16, 2), **args)
    r = run_tool(["--obj", tiny_obj(tmp_path), "--ktx", str(tmp_path / "t.ktx2"), "--no-gpu"], check=False)
    assert r.returncode == 1 and msg in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("samples,stored", [(4, True), (1, False)])
def test_loaded_model_frame_matches_oracle(tmp_path, oracle, samples, stored):
    """OBJ + KTX2 through the C++ host (the reference's test_app: dummy shaders, 4 samples), against
    the oracle drawing the dumped buffers with the dumped uniform block."""
    w, h = 320, 240
    path, _, _ = two_shape_model(tmp_path)
    levels = rgba_levels(64, 5, seed=9)
    if stored:
        write_ktx2(tmp_path / "t.ktx2", 43, 64, 64, levels)
    else:
        write_ktx2(tmp_path / "t.ktx2", 43, 64, 64, levels[:1], level_count=0)
    prefix = str(tmp_path / "dump")
    out = run_tool(["--obj", path, "--ktx", str(tmp_path / "t.ktx2"), "--width", str(w), "--height", str(h),
                    "--samples", str(samples), "--dump", prefix]).stdout
    assert "covered_samples=" in out and int(out.split("covered_samples=")[1]) > 500
    meta, vb, ib = load_dump(prefix)
    sc = scenes.Scene(w, h, samples=samples, clear_color=(0.0, 0.0, 0.0, 1.0), name="loaded")
    sc.vbs.append(vb.reshape(-1, 8))
    sc.ibs.append(ib)
    sc.ubos.append(np.array(meta["ubo_bits"], dtype=np.uint32).view(F))
    tex = dict(texels=levels[0], gen_mips=True, filtering=1, tiled=False, fmt=43)
    if stored:
        tex["levels"] = levels
    sc.textures.append(tex)
    sc.pipelines.append(scenes.default_pipeline(scenes.SHADER_DUMMY, samples=samples))
    for g in meta["sub_geometries"]:
        sc.draws.append(dict(pipeline=0, vb=0, vb_offset=g["vertices_offset"], ib=0, ib_offset=g["indices_offset"] * 4, ubo=0,
                             texture=0, push=scenes.push_block([1, 1, 1, 1], hm.identity()), index_count=g["indices_size"],
                             first_index=0, vertex_offset=0))
    ocolor, odepth, _ = oracle.render_scene(sc, threads=4)
    color = np.fromfile(prefix + ".color", dtype=np.uint8).reshape(h, w, 4)
    depth = np.fromfile(prefix + ".depth", dtype=np.uint32).reshape(odepth.shape)
    assert np.array_equal(depth, odepth)
    assert np.abs(color.astype(np.int32) - ocolor.astype(np.int32)).max() <= 1
    # the model transform the loader derived is the one in the uniform block's model matrix
    assert np.array_equal(np.array(meta["ubo_bits"], dtype=np.uint32)[48:64], np.array(meta["model_transform_bits"], dtype=np.uint32))
