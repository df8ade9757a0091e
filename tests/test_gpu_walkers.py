"""The walker's other forms against its default form, frame for frame (bit-identical ids, objects, w and radiance):
the burst form for leaves of tens of faces and the two-sided form for small leaves, each forced on the other's scenes
(trace_kernels.hip, switch MI355RT_BIG_LEAVES=1/0), the bounce loop with a read-back after every bounce, launches trimmed
more or less than the default, queues dealt out in order or spread, and frames cut into small passes.  The switches are set
through mi355rt_debug_switch (the library reads the environment only once).  The oracle pins the default form elsewhere."""
import pytest

pytestmark = pytest.mark.gpu


def _frames(rt, scene, W, H, spp):
    import torch
    cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=5)
    out = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"), torch.zeros((H, W), dtype=torch.int32, device="cuda"),
           torch.zeros((H, W), dtype=torch.int32, device="cuda"), torch.zeros((H, W), dtype=torch.float32, device="cuda")]
    cam.render_device(rt._abi.Frame(out[0].data_ptr(), None, out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr()))
    torch.cuda.synchronize()
    return [t.clone() for t in out]


def _scenes(rt):
    far = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    near = rt.Light((0.3, 0.9, 0.8), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))        # inside the scene's box: exact shadows
    v, i = rt.meshgen.blob(120000, span=2.0, seed=9)
    fv, fi = rt.meshgen.quad(-1.2, 3.0)
    for name, light, mat, kd in (("far light", far, rt.ColorInformation((0.9, 0.8, 0.7)), True),
                                 ("light inside, mirror", near, rt.ReflectionMaterial((0.9, 0.9, 1.0)), True),
                                 ("no hierarchy (one leaf of many chunks)", near, rt.ColorInformation((0.9, 0.8, 0.7)), False)):
        n = 120000 if kd else 3000
        objs = [rt.TriangleMesh((v, i[:n]), mat, kd), rt.TriangleMesh((fv, fi), rt.ColorInformation((0.8, 0.3, 0.2)), True)]
        yield name, rt.Scene(objs, light, (0, -25, 0), rt.PhongLightingModel(light, objs), 40.0)


@pytest.mark.parametrize("env", [{"MI355RT_BIG_LEAVES": "1"}, {"MI355RT_BOUNCE_SYNC": "1"},   # (round 2's bounce loop: a read-back after every bounce)
                                 {"MI355RT_BIG_LEAVES": "1", "MI355RT_FUSE_SHADOW": "1"},
                                 {"MI355RT_BIG_LEAVES": "0"}, {"MI355RT_BIG_LEAVES": "0", "MI355RT_FUSE_SHADOW": "1"},   # the two-sided form on leaves of many chunks too
                                 {"MI355RT_TRIM": "0"}, {"MI355RT_TRIM_PRIMARY_X2": "400", "MI355RT_TRIM_SHADOW_X2": "400"},   # launches not trimmed / trimmed to the minimum
                                 {"MI355RT_SPREAD_PRIMARY_MAX": "0"}, {"MI355RT_SPREAD_PRIMARY_MAX": "100000000", "MI355RT_SPREAD_SL": "3", "MI355RT_SPREAD_BL": "10"},   # rays taken in queue order / dealt out spread
                                 {"MI355RT_SPREAD_PRIMARY_MAX": "100000000", "MI355RT_SPREAD_SL": "5", "MI355RT_SPREAD_BL": "5"}, {"MI355RT_PASS_SLOTS_LOG2": "15"}])
def test_other_walker_forms_render_the_same_frames(rt, switch, env):
    import torch
    for name, scene in _scenes(rt):
        W, H, spp = (1280, 800, 4) if "far" in name else (480, 300, 2)    # (the far-light frame is long enough for the fused launch)
        for k in env:
            switch(k, None)
        ref = _frames(rt, scene, W, H, spp)
        assert int((ref[1] >= 0).sum()) > W * H // 20
        for k, val in env.items():
            switch(k, val)
        got = _frames(rt, scene, W, H, spp)
        for a, b in zip(ref, got):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (name, env)
        scene.close()
