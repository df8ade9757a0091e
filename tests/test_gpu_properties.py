"""Full-size and size-independent checks of the HIP path (through the C ABI): the BASELINE configuration
against the oracle, tile sharding, multi-pass frames, determinism, counter identities, error behaviour."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rt, faces, name="dragon", color=None, extra=()):
    cfg = rt.meshgen.STANFORD[name]
    v, i = rt.meshgen.stanford_standin(name, faces)
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    meshes = [rt.TriangleMesh((v, i), color or rt.ColorInformation((0.9, 0.9, 0.9)), True)] + list(extra)
    scene = rt.Scene(meshes, light, cfg["orientation"], rt.PhongLightingModel(light, meshes), cfg["fov"])
    return scene, (v, i), cfg, light


def _device_frame(rt, cam, **kw):
    import torch
    W, H = cam.screenWidth, cam.screenHeight
    dev = torch.device("cuda", 0)
    rgb = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
    hid = torch.full((H, W), -5, dtype=torch.int32, device=dev)
    hw = torch.full((H, W), -1.0, dtype=torch.float32, device=dev)
    frame = rt._abi.Frame(rgb.data_ptr(), None, hid.data_ptr(), None, hw.data_ptr())
    return frame, rgb, hid, hw


def test_dragon_config_full_size_vs_oracle(rt, orc):
    """BASELINE configs[1]: Dragon-class mesh (871,414 faces), 1920x1080, 4 spp, Phong, primary + shadow --
    the whole frame against the oracle: ids and w bit-exact, radiance within 1e-3 relative."""
    from helpers import compare_frames
    scene, (v, i), cfg, light = _scene(rt, None)
    cam = rt.Camera(scene, 4, 1, rt.RTStatics.NEAR_PLANE, 1920, 1080, rng_seed=2024)
    cam.createImage()
    osc = orc.OracleScene([(v, i, None, orc.material(diffuse=(0.9, 0.9, 0.9)))])
    ocam = osc.camera(cfg["orientation"], cfg["fov"], rt.RTStatics.NEAR_PLANE, 1920, 1080)
    ost = orc.settings(epsilon=osc.epsilon(), ms=4, lighting=1, seed=2024, light_origin=light.origin,
                       light_emission=light.emission, light_ambient=light.ambient)
    o = osc.render(ocam, ost, n_threads=16)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(cam._rgb_rows, cam.hit_id.T, cam.hit_w.T, o, g_obj=cam.hit_obj.T)
    assert ids_equal, "%d pixels differ in hit id" % n_diff
    assert w_equal
    assert max_rel <= 1e-3, max_rel
    assert 0.2 < float((o["hit_id"] >= 0).mean()) < 0.9
    osc.close()
    scene.close()


def test_tile_shards_merge_to_the_whole_frame(rt):
    """Three ranks' tiles written into one frame equal the single-rank frame bit for bit; a rank writes
    nothing outside its tiles."""
    import torch
    from ray_tracer_amd import tiles
    scene, _, _, _ = _scene(rt, 30000)
    W, H = 200, 120
    cam = rt.Camera(scene, 2, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=7)
    f1, rgb1, id1, w1 = _device_frame(rt, cam)
    cam.render_device(f1)
    torch.cuda.synchronize()
    assert bool((id1 >= -1).all())
    f3, rgb3, id3, w3 = _device_frame(rt, cam)
    for world, tile in ((3, 16), (2, 32)):
        id3.fill_(-5)
        for r in range(world):
            before = id3.clone()
            cam.render_device(f3, tile_rank=r, tile_world=world, tile_size=tile)
            torch.cuda.synchronize()
            pix = tiles.tile_pixels(W, H, tiles.owned_tiles(W, H, r, world, tile), tile)
            own = torch.zeros(W * H, dtype=torch.bool)
            own[pix[pix >= 0]] = True
            own = own.reshape(H, W).to(id3.device)
            assert torch.equal(id3[~own], before[~own])          # other ranks' tiles untouched
        assert torch.equal(id3, id1) and torch.equal(w3, w1) and torch.equal(rgb3, rgb1)
    scene.close()


def test_compact_blocks_unpack_to_the_whole_frame(rt):
    """The multi-GPU data path on one GPU: every rank of an N-rank frame renders its tiles in tile-compact order into
    its block (settings.tile_compact), the blocks laid end to end are what the gather delivers, and
    mi355rt_tiles_unpack_device puts them back in image order -- equal to the single-rank frame bit for bit (ids,
    objects, w, radiance), and equal to the index-tensor restatement the CPU (gloo) tests use.  Mesh over a mirror
    floor, so hit objects differ; image sizes that are no multiple of the tile."""
    import torch
    from ray_tracer_amd import tiles
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    v, i = rt.meshgen.stanford_standin("dragon", 20000)
    blob = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.6, 0.3)), True)
    fv, fi = rt.meshgen.quad(float(v[:, 1].min()) - 0.005, 0.3)
    floor = rt.TriangleMesh((fv, fi), rt.ReflectionMaterial((0.8, 0.8, 0.8)), True)
    scene = rt.Scene([blob, floor], light, (0, -25, 0), rt.PhongLightingModel(light, [blob, floor]), 23.0)
    dev = torch.device("cuda", 0)
    for (W, H, world, tile, spp) in ((203, 121, 3, 16, 2), (96, 70, 8, 8, 1), (333, 200, 2, 32, 1)):
        cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=9)
        whole = [torch.full((H, W, 3), -1.0, device=dev), torch.full((H, W), -5, dtype=torch.int32, device=dev),
                 torch.full((H, W), -5, dtype=torch.int32, device=dev), torch.full((H, W), -1.0, device=dev)]
        cam.render_device(rt._abi.Frame(whole[0].data_ptr(), None, whole[1].data_ptr(), whole[2].data_ptr(), whole[3].data_ptr()))
        sharders = [tiles.TileSharder(W, H, r, world, tile, dev) for r in range(world)]
        for r, sh in enumerate(sharders):
            cam.render_device(sh.blocks[0].frame(), tile_rank=r, tile_world=world, tile_size=tile, tile_compact=True)
        torch.cuda.synchronize()
        blocks = torch.cat([sh.blocks[0].flat for sh in sharders])
        got = [torch.full_like(t, -9) for t in whole]
        sharders[0].unpack(blocks, *got)
        torch.cuda.synchronize()
        for a, b in zip(whole, got):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (W, H, world, tile)
        ref = [torch.full_like(t, -9).cpu() for t in whole]
        tiles.tiles_unpack_reference(blocks.cpu(), world, sharders[0].slots, tile, *ref)
        for a, b in zip(whole, ref):
            assert torch.equal(a.cpu().view(torch.int32), b.view(torch.int32))
        assert int((whole[2] == 1).sum()) > 0 and int((whole[2] == 0).sum()) > 0
    scene.close()


def test_multi_pass_frame(rt, orc, switch):
    """A frame of more sample slots than a pass holds runs as several passes of the pipeline (api.cpp pass_slots; the
    default pass, 2^27 slots, is forced down to 2^24 here)."""
    from helpers import compare_frames
    switch("PASS_SLOTS_LOG2", "24")
    scene, (v, i), cfg, light = _scene(rt, 5000, name="bunny")
    W, H, ms = 1920, 1088, 9       # 18.8 M samples -> 2 passes
    cam = rt.Camera(scene, ms, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=3)
    cam.createImage()
    osc = orc.OracleScene([(v, i, None, orc.material(diffuse=(0.9, 0.9, 0.9)))])
    ocam = osc.camera(cfg["orientation"], cfg["fov"], rt.RTStatics.NEAR_PLANE, W, H)
    ost = orc.settings(epsilon=osc.epsilon(), ms=ms, lighting=1, seed=3, light_origin=light.origin,
                       light_emission=light.emission, light_ambient=light.ambient)
    o = osc.render(ocam, ost, n_threads=16)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(cam._rgb_rows, cam.hit_id.T, cam.hit_w.T, o, g_obj=cam.hit_obj.T)
    assert ids_equal and w_equal and max_rel <= 1e-3, (n_diff, max_rel)
    osc.close()
    scene.close()


def test_frames_do_not_depend_on_the_pass_size(rt, switch):
    """How a frame is cut into passes is a matter of memory and speed only: a mirror over a floor rendered in passes of
    2^14, 2^17 and the default number of slots gives the same bits."""
    import torch
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    v, i = rt.meshgen.blob(60000, span=2.0, seed=12)
    fv, fi = rt.meshgen.quad(-1.2, 3.0)
    objs = [rt.TriangleMesh((v, i), rt.ReflectionMaterial((0.9, 0.9, 1.0)), True), rt.TriangleMesh((fv, fi), rt.ColorInformation((0.8, 0.3, 0.2)), True)]
    scene = rt.Scene(objs, light, (0, -25, 0), rt.PhongLightingModel(light, objs), 40.0)
    W, H, ms = 640, 400, 3
    cam = rt.Camera(scene, ms, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=8)
    frames = []
    for log2 in ("14", "17", None):
        if log2 is None:
            switch("PASS_SLOTS_LOG2", None)
        else:
            switch("PASS_SLOTS_LOG2", log2)
        f, rgb, ids, w = _device_frame(rt, cam)
        cam.render_device(f)
        torch.cuda.synchronize()
        frames.append((rgb.clone(), ids.clone(), w.clone()))
    assert int((frames[0][1] >= 0).sum()) > W * H // 20
    for other in frames[1:]:
        for a, b in zip(frames[0], other):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    scene.close()


@pytest.mark.parametrize("fused", [True, False])
def test_determinism_and_counter_identities(rt, switch, fused):
    """Two renders are bit-identical (whatever wave walked which ray or stole which subtree), and the
    counters obey: primary = W*H*spp, shadow = hits (one shadow ray per hit sample, RTStatics.java:617),
    per-kernel counts add up.  Both pipelines: shadow rays walked inside the primary launch (the default
    when any accepted face occludes) and in a launch of their own."""
    import torch
    switch("FUSE_SHADOW", "1" if fused else "0")
    scene, _, _, _ = _scene(rt, 100000)
    W, H, ms = 320, 200, 4
    cam = rt.Camera(scene, ms, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=11)
    fa, rgba, ida, wa = _device_frame(rt, cam)
    fb, rgbb, idb, wb = _device_frame(rt, cam)
    cam.render_device(fa, collect_stats=True, collect_times=True)
    torch.cuda.synchronize()
    st = cam.frame_stats()
    cam.render_device(fb)
    torch.cuda.synchronize()
    assert torch.equal(ida, idb) and torch.equal(wa, wb) and torch.equal(rgba, rgbb)
    assert st["primary_rays"] == W * H * ms
    assert st["shadow_rays"] == st["hits"] > 0
    assert st["secondary_rays"] == 0
    kp = cam.kernel_profile(reset=True)
    # the primary queue holds only the samples whose ray meets some object's box (the rest are misses at generation)
    assert st["hits"] <= kp["walk_primary"]["rays"] <= st["primary_rays"]
    assert kp["walk_shadow"]["rays"] == st["shadow_rays"]
    assert kp["walk_primary"]["tri_tests"] + kp["walk_shadow"]["tri_tests"] == st["tri_tests"]
    for name in ("generate", "walk_primary", "shade", "walk_shadow", "phong", "resolve"):
        if name in ("walk_shadow", "phong") and fused:   # shadow rays walked in the primary launch, Phong applied in shade
            assert kp[name]["launches"] == 0
        else:
            assert kp[name]["launches"] == 1 and kp[name]["ms"] > 0.0
    assert kp["walk_secondary"]["launches"] == 0
    assert cam.kernel_profile()["walk_primary"]["launches"] == 0      # reset
    scene.close()


def test_shadow_rays_inside_the_primary_launch_equal_the_separate_launch(rt, switch):
    """MODE_FUSED (each lane walks its hit's shadow ray before it takes another primary ray) gives the frame the
    two-launch pipeline gives, bit for bit -- on a scene with two meshes, one of them a mirror (the shadow ray leaves
    the PRIMARY hit whatever the bounces find), and on the BASELINE Dragon frame."""
    import torch
    v2, i2 = rt.meshgen.stanford_standin("bunny", 4000)
    v2 = (v2 * 0.6 + np.array([0.12, 0.05, 0.0], dtype=np.float32)).astype(np.float32)
    mirror = rt.TriangleMesh((v2, i2), rt.ReflectionMaterial((0.8, 0.8, 0.8)), True)
    for faces, extra, W, H, ms in ((20000, (mirror,), 320, 240, 2), (None, (), 1920, 1080, 4)):
        scene, _, _, _ = _scene(rt, faces, extra=extra)
        cam = rt.Camera(scene, ms, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=5)
        frames = {}
        for fused in ("1", "0"):
            switch("FUSE_SHADOW", fused)
            f, rgb, hid, hw = _device_frame(rt, cam)
            cam.render_device(f, collect_stats=True, collect_times=True)
            torch.cuda.synchronize()
            kp = cam.kernel_profile(reset=True)
            assert kp["walk_shadow"]["launches"] == (0 if fused == "1" else 1)
            assert kp["walk_shadow"]["rays"] == cam.frame_stats()["shadow_rays"] > 0
            frames[fused] = (rgb, hid, hw)
        for x, y in zip(frames["1"], frames["0"]):
            assert torch.equal(x, y)
        scene.close()


def test_pinned_output_buffers_give_the_same_frame(rt):
    """mi355rt_host_alloc: createImage() into one page-locked set of arrays (Camera.pinOutput) returns the frame it
    returns into fresh pageable arrays, and keeps writing into the same arrays."""
    scene, _, _, _ = _scene(rt, 20000)
    cam = rt.Camera(scene, 2, 1, rt.RTStatics.NEAR_PLANE, 320, 200, rng_seed=9)
    cam.createImage()
    ref = (cam._rgb_rows.copy(), cam.hit_id.copy(), cam.hit_w.copy())
    cam.pinOutput(True)
    cam.createImage()
    first = cam._rgb_rows
    assert np.array_equal(first, ref[0]) and np.array_equal(cam.hit_id, ref[1]) and np.array_equal(cam.hit_w, ref[2])
    first.fill(-7.0)
    cam.createImage()
    assert cam._rgb_rows is first and np.array_equal(first, ref[0])
    del first
    cam.pinOutput(False)
    cam.createImage()
    assert np.array_equal(cam._rgb_rows, ref[0])
    with pytest.raises(rt._abi.Mi355rtError):
        rt._abi.check(rt._abi.load().mi355rt_host_alloc(0, C.byref(C.c_void_p())))
    scene.close()


def test_mirror_bounces_between_two_mirrors(rt, orc):
    """Two facing mirrors: paths bounce until max_mirror_bounces cuts them (the reference would recurse
    without bound, bounds/KDTree.java:77-80; the oracle carries the same guard)."""
    from helpers import Pair
    v, i = rt.meshgen.blob(3000, span=0.2, seed=31)
    v2, i2 = rt.meshgen.blob(2000, span=0.15, seed=32, center=(0.2, 0.0, 0.0))
    fv, fi = rt.meshgen.quad(-0.12, 0.3, center=(0.1, 0.0))
    pair = Pair([(v, i, None, rt.ReflectionMaterial((0.9, 0.9, 1.0))), (v2, i2, None, rt.ReflectionMaterial((1.0, 0.8, 0.8))),
                 (fv, fi, None, rt.ColorInformation((0.3, 0.8, 0.2)))], orientation=(0, -15, 0), fov=25.0)
    from helpers import compare_frames
    cam, ocam, ost = pair.cameras(160, 100, ms=2, seed=17)
    cam.createImage()
    o = pair.oscene.render(ocam, ost, n_threads=8, want_stats=True)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(cam._rgb_rows, cam.hit_id.T, cam.hit_w.T, o, g_obj=cam.hit_obj.T)
    assert ids_equal and w_equal and max_rel <= 1e-3, (n_diff, max_rel)
    assert o["stats"]["secondary_rays"] > 1000
    pair.close()


def test_error_behaviour(rt):
    """Bad arguments come back as error codes with a message, never as a fault."""
    lib = rt._abi.load()
    scene, _, _, _ = _scene(rt, 2000, name="bunny")
    cam = rt.Camera(scene, 1, 1, rt.RTStatics.NEAR_PLANE, 32, 32)
    st = cam.settings()
    st.multi_samples = 0
    rc = lib.mi355rt_render(scene._handle, C.byref(cam._cam), C.byref(st), None, None, None, None)
    assert rc == rt._abi.E_ARG and b"multi_samples" in lib.mi355rt_last_error()
    st = cam.settings(tile_rank=2, tile_world=2)
    assert lib.mi355rt_render(scene._handle, C.byref(cam._cam), C.byref(st), None, None, None, None) == rt._abi.E_ARG
    st = cam.settings(tile_size=12)
    assert lib.mi355rt_render(scene._handle, C.byref(cam._cam), C.byref(st), None, None, None, None) == rt._abi.E_ARG
    # a BRDF material whose row is outside the loaded tables (8 rows)
    from helpers import synthetic_brdf_tables
    v, i = rt.meshgen.blob(500, span=0.2)
    rt.BRDFMaterial.setTables(*synthetic_brdf_tables())
    light = rt.Light((0, 100, 100), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    m = rt.TriangleMesh((v, i), rt.BRDFMaterial(50, (0.2, 0.8, 0.3)), True)
    s2 = rt.Scene([m], light, (0, 0, 0), rt.PhongLightingModel(light, [m]), 15.0)
    c2 = rt.Camera(s2, 1, 1, rt.RTStatics.NEAR_PLANE, 16, 16)
    with pytest.raises(rt._abi.Mi355rtError) as e:
        c2.createImage()
    assert e.value.code == rt._abi.E_ARG
    # an index out of range is refused at scene build
    bad = i.copy()
    bad[7, 1] = v.shape[0]
    with pytest.raises(rt._abi.Mi355rtError):
        rt.Scene([rt.TriangleMesh((v, bad), rt.ColorInformation((1, 1, 1)), True)], light, (0, 0, 0), rt.LightingModel(), 15.0)
    s2.close()
    scene.close()


def test_empty_and_degenerate_meshes(rt, orc):
    """An empty mesh beside a real one, and a mesh with zero-area and NaN-normal faces."""
    from helpers import Pair
    v, i = rt.meshgen.blob(1500, span=0.2, seed=41)
    i = i.copy()
    i[10] = (i[10, 0], i[10, 0], i[10, 1])         # zero-area face: NaN normals at its vertices (TriangleMesh.java:96-126)
    ev = np.zeros((3, 3), dtype=np.float32)
    ei = np.zeros((0, 3), dtype=np.int32)
    pair = Pair([(v, i, None, rt.ColorInformation((1, 1, 1))), (ev, ei, None, rt.ColorInformation((1, 0, 0)))], fov=15.0)
    cam, ocam, ost = pair.cameras(96, 96)
    cam.createImage()
    o = pair.oscene.render(ocam, ost, n_threads=4)
    assert np.array_equal(cam.hit_id.T, o["hit_id"])
    assert np.array_equal(cam.hit_w.T.view(np.uint32), o["hit_w"].view(np.uint32))
    both_nan = np.isnan(cam._rgb_rows) & np.isnan(o["rgb"])
    close = np.abs(cam._rgb_rows - o["rgb"]) <= 1e-3 * np.maximum(np.abs(o["rgb"]), 1e-3)
    assert (both_nan | close).all()
    pair.close()


def test_8bit_images_match_oracle(rt, orc):
    """Camera.image and Camera.getNormalizedImage() (Camera.java:103-107, 217-263) from a rendered frame, and from
    a synthetic frame with negative, > 1 and NaN-scrubbed (zero) pixels: every packed pixel identical."""
    import ctypes as C
    scene, _, _, _ = _scene(rt, 8000, name="bunny")
    cam = rt.Camera(scene, 2, 1, rt.RTStatics.NEAR_PLANE, 160, 120, rng_seed=4)
    cam.createImage()
    frames = [cam._rgb_rows]
    rs = np.random.RandomState(8)
    synth = (rs.standard_normal((90, 130, 3)) * 0.8 + 0.4).astype(np.float32)
    synth[5, 7] = 0.0
    frames.append(synth)
    for rgb in frames:
        H, W = rgb.shape[:2]
        cam._rgb_rows, cam.screenHeight, cam.screenWidth = np.ascontiguousarray(rgb), H, W
        for normalized in (False, True):
            got = cam.getImage(normalized)
            exp = np.zeros((H, W), dtype=np.uint32)
            orc.lib().orc_image_argb8(orc.P(np.ascontiguousarray(rgb)), W, H, 1 if normalized else 0, exp.ctypes.data_as(C.POINTER(C.c_uint32)))
            np.testing.assert_array_equal(got, exp)
            assert len(np.unique(got)) > 50
    scene.close()


def _compare_at_sampled_pixels(cam, o, rtol=1e-3):
    """The oracle rendered every k-th pixel (hit_id stays -2 elsewhere): ids and w bit-exact, radiance within rtol there."""
    m = o["hit_id"] != -2
    assert m.sum() > 1000
    assert np.array_equal(cam.hit_id.T[m], o["hit_id"][m]), int((cam.hit_id.T[m] != o["hit_id"][m]).sum())
    assert np.array_equal(cam.hit_obj.T[m], o["hit_obj"][m])
    assert np.array_equal(cam.hit_w.T[m].view(np.uint32), o["hit_w"][m].view(np.uint32))
    denom = np.maximum(np.abs(o["rgb"][m]), 1e-3)
    rel = np.abs(cam._rgb_rows[m] - o["rgb"][m]) / denom
    assert rel.max() <= rtol, float(rel.max())
    return float((o["hit_id"][m] >= 0).mean())


def test_buddha_config_brdf_full_size(rt, orc):
    """BASELINE configs[2]: Buddha-class mesh (1,087,716 faces), 1920x1080, measured-BRDF material under Phong,
    orientation (180,0,0), fov 10 (RayTracer.java:125-128) -- on the reference's own CUReT tables (abrdf.dat row 15, the
    material RayTracer.java:162 uses; tests/golden/curet_tables.npz).  The whole frame on the GPU; the oracle (its
    table search is slow on a CPU) checks every 7th pixel."""
    from helpers import curet_tables
    cfg = rt.meshgen.STANFORD["buddha"]
    v, i = rt.meshgen.stanford_standin("buddha")
    assert i.shape[0] == 1087716
    tables = curet_tables()
    rt.BRDFMaterial.setTables(*tables)
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    mesh = rt.TriangleMesh((v, i), rt.BRDFMaterial(15, (0.0, 1.0, 0.0)), True)
    scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"])
    cam = rt.Camera(scene, 1, 2, rt.RTStatics.NEAR_PLANE, 1920, 1080, rng_seed=5)
    cam.createImage()
    osc = orc.OracleScene([(v, i, None, orc.material(kind=1, diffuse=(0.0, 1.0, 0.0), brdf_id=15))])
    osc.set_brdf(*tables)
    ocam = osc.camera(cfg["orientation"], cfg["fov"], rt.RTStatics.NEAR_PLANE, 1920, 1080)
    ost = orc.settings(epsilon=osc.epsilon(), ms=1, brdf_samples=2, lighting=1, seed=5, light_origin=light.origin,
                       light_emission=light.emission, light_ambient=light.ambient)
    o = osc.render(ocam, ost, n_threads=16, pixel_stride=7)
    frac = _compare_at_sampled_pixels(cam, o)
    assert 0.05 < frac < 0.95
    osc.close()
    scene.close()


def test_thai_class_mirror_4k_two_passes(rt, orc, switch):
    """The shape of BASELINE configs[3]/[4] at a size the oracle can follow: a 2-million-face mesh of span 400
    (EPSILON = 1e-3, Scene.java:36) with ReflectionMaterial over a floor, 3840x2160, 3 spp = 24.9 M samples in two
    passes with mirror bounces; the oracle checks every 211th pixel."""
    from helpers import Pair
    switch("PASS_SLOTS_LOG2", "24")
    v, i = rt.meshgen.blob(2000000, span=400.0, stretch=(0.59, 1.0, 0.52), seed=4)
    fv, fi = rt.meshgen.quad(-205.0, 230.0)
    pair = Pair([(v, i, None, rt.ReflectionMaterial((0.9, 0.9, 1.0))), (fv, fi, None, rt.ColorInformation((0.8, 0.3, 0.2)))],
                orientation=(0, -20, 0), fov=30.0)
    assert np.float32(pair.eps) == np.float32(1e-3)
    cam, ocam, ost = pair.cameras(3840, 2160, ms=3, seed=21)
    cam.createImage()
    o = pair.oscene.render(ocam, ost, n_threads=16, pixel_stride=211, want_stats=True)
    frac = _compare_at_sampled_pixels(cam, o)
    assert frac > 0.2 and o["stats"]["secondary_rays"] > 10000
    pair.close()


def test_lucy_class_parity_4k(rt, orc):
    """BASELINE configs[4] at a size the oracle can follow: the Lucy stand-in at 3,000,000 faces -- span 1500 (EPSILON =
    1e-3, Scene.java:36), orientation (0,-90,0), fov 10, light (0,100,100) INSIDE the scene's box (RayTracer.java:106,129-132),
    so the shadow rays run RTStatics.shadowIntersects' own closest-hit rule (MODE_SHADOW) -- at 3840x2160, 2 spp.  The
    hierarchy hits the reference's depth limit of 20 (leaves of many faces), where the walker changes character.  The
    oracle checks every 199th pixel: faces, objects, w bit-exact, radiance within 1e-3."""
    from helpers import Pair
    cfg = rt.meshgen.STANFORD["lucy"]
    v, i = rt.meshgen.stanford_standin("lucy", 3000000)
    pair = Pair([(v, i, None, rt.ColorInformation((1.0, 1.0, 1.0)))], orientation=cfg["orientation"], fov=cfg["fov"])
    assert np.float32(pair.eps) == np.float32(1e-3)
    info = pair.scene.object_info(0)
    assert info["max_depth"] == 20 and info["n_leaf_slots"] > 5 * info["n_leaves"]      # depth-capped leaves
    cam, ocam, ost = pair.cameras(3840, 2160, ms=2, seed=31)
    cam.createImage()
    o = pair.oscene.render(ocam, ost, n_threads=16, pixel_stride=199, want_stats=True)
    frac = _compare_at_sampled_pixels(cam, o)
    assert frac > 0.05 and o["stats"]["shadow_rays"] > 1000
    # some sampled pixels are in shadow and some are lit: both outcomes of the exact rule occur
    m = o["hit_id"] >= 0
    lum = o["rgb"][m].sum(axis=1)
    assert lum.min() < 0.7 * lum.max()
    pair.close()


def test_lucy_config_full_size_properties(rt):
    """BASELINE configs[4] at full size -- 28,055,742 faces, 3840x2160, 16 spp (one pass of 2^27 slots), Phong, exact shadows -- through
    properties that need no oracle: every sample is a primary ray, every hit sample casts one shadow ray, two renders are
    bit-identical (whatever wave walked or stole what), and the first sample's hits agree with a 1-spp frame's."""
    import torch
    cfg = rt.meshgen.STANFORD["lucy"]
    v, i = rt.meshgen.stanford_standin("lucy")
    assert i.shape[0] == 28055742
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((1.0, 1.0, 1.0)), True)
    scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"])
    del v, i
    W, H, spp = 3840, 2160, 16
    cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)
    fa, rgba, ida, wa = _device_frame(rt, cam)
    fb, rgbb, idb, wb = _device_frame(rt, cam)
    cam.render_device(fa, collect_stats=True)
    torch.cuda.synchronize()
    st = cam.frame_stats()
    assert st["primary_rays"] == W * H * spp
    assert st["shadow_rays"] == st["hits"] > W * H * spp // 50
    cam.render_device(fb)
    torch.cuda.synchronize()
    assert torch.equal(ida, idb) and torch.equal(wa, wb) and torch.equal(rgba.view(torch.int32), rgbb.view(torch.int32))
    cam1 = rt.Camera(scene, 1, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)     # sample 0 is unjittered: same rays
    cam1.render_device(fb)
    torch.cuda.synchronize()
    assert torch.equal(ida, idb) and torch.equal(wa, wb)
    assert bool(torch.isfinite(rgba).all())
    scene.close()


def test_thai_config_full_size_16spp(rt, orc):
    """BASELINE configs[3] at full size: the Thai-statuette stand-in, 10,000,000 faces of span 400 (EPSILON = 1e-3) with
    ReflectionMaterial over a floor (a lone mirrored mesh reflects nothing: ReflectionMaterial.java:14-24 skips the object
    itself), orientation (0,0,0) tilted to show the floor, 3840x2160 at 16 spp = 132.7 M samples in one pass with mirror
    bounces.  The oracle checks every 499th pixel (all 16 samples, shadow and bounce rays of each)."""
    from helpers import Pair
    cfg = rt.meshgen.STANFORD["thai"]
    v, i = rt.meshgen.stanford_standin("thai")
    assert i.shape[0] == 10000000
    fv, fi = rt.meshgen.quad(float(v[:, 1].min()) - 5.0, 230.0)
    pair = Pair([(v, i, None, rt.ReflectionMaterial((0.9, 0.9, 1.0))), (fv, fi, None, rt.ColorInformation((0.8, 0.3, 0.2)))],
                orientation=(0, -20, 0), fov=30.0)
    assert np.float32(pair.eps) == np.float32(1e-3)
    cam, ocam, ost = pair.cameras(3840, 2160, ms=16, seed=23)
    cam.createImage()
    o = pair.oscene.render(ocam, ost, n_threads=16, pixel_stride=499, want_stats=True)
    frac = _compare_at_sampled_pixels(cam, o)
    assert frac > 0.2 and o["stats"]["secondary_rays"] > 10000
    assert o["stats"]["primary_rays"] == 16 * int((o["hit_id"] != -2).sum())
    pair.close()
