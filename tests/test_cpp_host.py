"""The C++ host mirror (include/mi355rt.hpp) driven the way RayTracer.java drives the reference:
tests/cpp/host_mirror (built by __graft_entry__.build() / make -C tests/cpp)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "cpp", "host_mirror")


def _run(*args):
    if not os.path.exists(BINARY):
        pytest.fail("tests/cpp/host_mirror is not built: run __graft_entry__.build()")
    return subprocess.run([BINARY, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)


def test_cpp_mirror_builds_and_reports_errors():
    """No GPU needed: material defaults, a missing PLY raises E_IO, and without a device scene creation is refused
    (MI355RT_E_DEVICE) rather than served by some fallback."""
    r = _run("abi")
    assert r.returncode == 0, r.stdout
    assert "ABI_OK" in r.stdout and "E_IO" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_frame_matches_oracle(rt, orc, tmp_path):
    """TriangleMesh(ply) -> Scene -> Camera.createImage() in C++ against the oracle on the same PLY."""
    from helpers import compare_frames
    v, i = rt.meshgen.blob(12000, span=0.156, stretch=(1.0, 0.99, 0.78), seed=1)
    ply = str(tmp_path / "mesh.ply")
    rt.meshgen.write_ply(ply, v, i, binary=True)
    W, H, ms, seed, fov = 200, 150, 2, 77, 15.0
    out = str(tmp_path / "frame.bin")
    r = _run("render", ply, W, H, ms, seed, fov, out)
    assert r.returncode == 0 and "RENDER_OK" in r.stdout, r.stdout
    raw = np.fromfile(out, dtype=np.float32)
    header, body = raw[:8], raw[8:]
    rgb = body[: 3 * W * H].reshape(H, W, 3)
    hid = body[3 * W * H: 4 * W * H].view(np.int32).reshape(H, W)
    hw = body[4 * W * H: 5 * W * H].reshape(H, W)
    hobj = body[5 * W * H: 6 * W * H].view(np.int32).reshape(H, W)
    ov, oi, on = orc.ply_load(ply)
    osc = orc.OracleScene([(ov, oi, on, orc.material(diffuse=(0.9, 0.9, 0.9)))])
    ocam = osc.camera((0, 0, 0), fov, 0.01, W, H)
    np.testing.assert_array_equal(np.array(ocam.origin[:], dtype=np.float32), header[:3])
    assert np.float32(osc.epsilon()) == header[5]
    ost = orc.settings(epsilon=osc.epsilon(), ms=ms, lighting=1, seed=seed)
    o = osc.render(ocam, ost, n_threads=8)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(rgb, hid, hw, o, g_obj=hobj)
    assert ids_equal and w_equal and max_rel <= 1e-3, (n_diff, max_rel)
    assert (hid >= 0).mean() > 0.1
    osc.close()


@pytest.mark.gpu
def test_cpp_mirror_whitted_scene_matches_oracle(orc, tmp_path):
    """Plane / Sphere / CheckerboardMaterial / RefractionMaterial / ReflectionMaterial built in C++ the way
    RayTracer.getWhittedObjects does (RayTracer.java:155-167), against the oracle."""
    from helpers import compare_frames
    W, H, ms, seed = 240, 180, 2, 5
    out = str(tmp_path / "whitted.bin")
    r = _run("whitted", W, H, ms, seed, out)
    assert r.returncode == 0 and "RENDER_OK" in r.stdout, r.stdout
    raw = np.fromfile(out, dtype=np.float32)
    header, body = raw[:8], raw[8:]
    rgb = body[: 3 * W * H].reshape(H, W, 3)
    hid = body[3 * W * H: 4 * W * H].view(np.int32).reshape(H, W)
    hw = body[4 * W * H: 5 * W * H].reshape(H, W)
    hobj = body[5 * W * H: 6 * W * H].view(np.int32).reshape(H, W)
    floor = orc.plane([(-50, 0, -100), (-50, -40, 25), (50, -40, 25), (50, 0, -100)], orc.material(4, (0, 0, 0), 20.0),
                      orc.material_ext(0, (1, 1, 0), (1, 0, 0), (10, 10, 10)))
    glass = orc.sphere((0, -12, 0), 5, orc.material(3, (128 / 255.0,) * 3, 100.0), orc.material_ext(1.52))
    mirror = orc.sphere((5, -15, -10), 3, orc.material(2, (0, 0, 1), 100.0))
    osc = orc.OracleScene([floor, glass, mirror])
    ocam = osc.camera((0, 0, 0), 35.0, 0.01, W, H)
    np.testing.assert_array_equal(np.array(ocam.origin[:], dtype=np.float32), header[:3])
    assert np.float32(osc.epsilon()) == header[5] == np.float32(1e-3)
    ost = orc.settings(epsilon=osc.epsilon(), ms=ms, lighting=1, seed=seed)
    o = osc.render(ocam, ost, n_threads=8)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(rgb, hid, hw, o, g_obj=hobj)
    assert ids_equal and w_equal and max_rel <= 1e-3, (n_diff, max_rel)
    assert set(np.unique(o["hit_obj"])) == {-1, 0, 1, 2}
    osc.close()


@pytest.mark.gpu
def test_cpp_mirror_device_list_and_8bit_image(rt, tmp_path):
    """Scene::setDevices from C++ (GPU 0 named twice: two ranks, peer copies) renders the one-GPU frame bit for bit, and
    Camera.image (mi355rt_render_host's argb8 output, Camera.java:103-107) is the 8-bit pack of Camera.pixels."""
    v, i = rt.meshgen.blob(9000, span=0.156, seed=4)
    ply = str(tmp_path / "mesh.ply")
    rt.meshgen.write_ply(ply, v, i, binary=True)
    W, H, ms, seed, fov = 208, 152, 2, 11, 15.0
    frames = []
    for ranks in (0, 2):
        out = str(tmp_path / ("frame%d.bin" % ranks))
        r = _run("render", ply, W, H, ms, seed, fov, out, ranks)
        assert r.returncode == 0 and "RENDER_OK" in r.stdout, r.stdout
        if ranks:
            assert "ranks 2 transport 1" in r.stdout, r.stdout   # MI355RT_TRANSPORT_PEER
        assert "build seconds" in r.stdout
        frames.append(np.fromfile(out, dtype=np.uint32))
    assert frames[0].size == 8 + 7 * W * H
    assert np.array_equal(frames[0], frames[1])
    body = frames[0][8:]
    rgb = body[: 3 * W * H].view(np.float32).reshape(H, W, 3)
    image = body[6 * W * H: 7 * W * H].reshape(H, W)
    c = np.clip(rgb, np.float32(0.0), np.float32(1.0))   # Camera.java:103-107: new Color(bound(r), bound(g), bound(b)).getRGB()
    q = ((c * np.float32(255.0)).astype(np.float64) + 0.5).astype(np.int64).astype(np.uint32) & np.uint32(255)   # Color(float...): (int) (x * 255 + 0.5)
    expect = np.uint32(0xFF000000) | (q[..., 0] << np.uint32(16)) | (q[..., 1] << np.uint32(8)) | q[..., 2]
    assert np.array_equal(image, expect)
    assert (image != 0xFF000000).mean() > 0.05
