"""The C-ABI library loads and exports every symbol include/mi355rt.h declares; without a HIP device the
compute entry points fail loudly (there is no CPU path in the product)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355rt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("mi355rt_scene_create", "mi355rt_scene_destroy", "mi355rt_brdf_tables", "mi355rt_render", "mi355rt_render_device",
                 "mi355rt_camera_setup", "mi355rt_trace_rays", "mi355rt_last_error", "mi355rt_ply_load"):
        assert must in syms


def test_library_exports_every_declared_symbol(rt):
    lib = C.CDLL(rt._abi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    # and the ctypes view binds exactly the declared set
    assert sorted(rt._abi.PROTOTYPES) == declared_symbols()


def test_abi_version(rt):
    assert rt._abi.load().mi355rt_abi_version() == 8


def test_struct_sizes_match_the_header(rt):
    a = rt._abi
    assert C.sizeof(a.Material) == 4 + 12 * 4 + 4 + 4
    assert C.sizeof(a.BuildOpts) == 16
    assert C.sizeof(a.CameraDesc) == (3 + 9 + 3) * 4 + 8
    assert C.sizeof(a.Settings) == 96 and a.Settings.tile_compact.offset == 92 and a.Settings.rng_seed.offset == 32
    assert C.sizeof(a.MeshDesc) == 96 and a.MeshDesc.compute_kdtree.offset == 92
    assert C.sizeof(a.Stats) == 48
    assert C.sizeof(a.Frame) == 40


def test_no_cpu_fallback(rt):
    """Without a GPU the product must fail with MI355RT_E_DEVICE, not compute on the host."""
    lib = rt._abi.load()
    if lib.mi355rt_device_count() > 0:
        pytest.skip("a HIP device is present")
    v, i = rt.meshgen.blob(100)
    with pytest.raises(rt._abi.Mi355rtError) as e:
        rt.Scene([rt.TriangleMesh((v, i), rt.ColorInformation((1, 1, 1)))], rt.Light(), (0, 0, 0), rt.LightingModel(), 15.0)
    assert e.value.code == rt._abi.E_DEVICE
    out = [np.zeros(4, dtype=np.float32) for _ in range(4)]
    a = np.ones(4, dtype=np.float32)
    P = lambda x: x.ctypes.data_as(rt._abi.c_float_p)
    rc = lib.mi355rt_debug_arith(4, P(a), P(a), *[P(x) for x in out])
    assert rc == rt._abi.E_DEVICE


def test_argument_errors_are_reported(rt):
    lib = rt._abi.load()
    assert lib.mi355rt_scene_create(None, 0, None, None) == rt._abi.E_ARG
    assert b"null" in lib.mi355rt_last_error()
    assert lib.mi355rt_ply_load(b"/nonexistent/file.ply", *[None] * 6) == rt._abi.E_ARG


def test_debug_switch_knows_its_names(rt):
    """mi355rt_debug_switch (no GPU needed): every documented switch is accepted with and without the MI355RT_ prefix, a
    value of None restores the default, name None restores the environment's values, an unknown name is MI355RT_E_ARG."""
    for name in ("BUILDER", "BUILD_TIMING", "FUSE_SHADOW", "PASS_SLOTS_LOG2", "BIG_LEAVES", "BLOCKS_PER_CU", "BLOCKS_PER_CU_SHADOW", "TRIM",
                 "TRIM_PRIMARY_X2", "TRIM_SHADOW_X2", "SPREAD_SL", "SPREAD_BL", "SPREAD_PRIMARY_MAX", "TRANSPORT", "BOUNCE_SYNC", "GPU_GROUP"):
        rt.debug_switch(name, 1 if name not in ("BUILDER", "TRANSPORT") else ("host" if name == "BUILDER" else "peer"))
        rt.debug_switch("MI355RT_" + name, None)
    rt.debug_switch(None, None)
    with pytest.raises(rt._abi.Mi355rtError) as e:
        rt.debug_switch("NO_SUCH_SWITCH", 1)
    assert "unknown switch" in str(e.value)
