"""Host-side pieces that need no GPU: input meshes, PLY reader (library vs oracle vs expectation), normal
synthesis (library vs oracle), tile bookkeeping, BRDF table parsing rule."""
import os

import numpy as np
import pytest

f32 = np.float32


@pytest.mark.parametrize("n", [12, 13, 600, 601, 69451])
def test_meshgen_exact_face_count_and_closed(rt, n):
    v, i = rt.meshgen.blob(n, span=0.2, seed=3)
    assert i.shape == (n, 3) and i.min() >= 0 and i.max() < v.shape[0]
    assert np.isfinite(v).all()
    assert np.isclose((v.max(0) - v.min(0)).max(), 0.2, rtol=1e-6)
    assert np.bincount(i.ravel(), minlength=v.shape[0]).min() >= 1        # no unreferenced vertex
    assert np.bincount(i.ravel()).max() <= 12                              # no high-valence pole
    v2, i2 = rt.meshgen.blob(n, span=0.2, seed=3)
    assert np.array_equal(v, v2) and np.array_equal(i, i2)                 # seeded => reproducible


def test_stanford_standins(rt):
    v, i = rt.meshgen.stanford_standin("dragon")
    assert i.shape[0] == 871414
    assert np.isclose((v.max(0) - v.min(0))[0], 0.205, rtol=1e-5)


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("with_normals", [False, True])
def test_ply_readers_agree(rt, orc, tmp_path, binary, with_normals):
    v, i = rt.meshgen.blob(300, span=0.156, seed=5)
    n = rt.compute_normals(v, i) if with_normals else None
    path = str(tmp_path / "m.ply")
    rt.meshgen.write_ply(path, v, i, normals=n, binary=binary)
    pv, pi, pn = rt.load_ply(path)          # libmi355rt (C++)
    ov, oi, on = orc.ply_load(path)         # oracle (C)
    for a, b in ((pv, v), (ov, v)):
        np.testing.assert_array_equal(a.view(np.uint32), v.view(np.uint32))
    np.testing.assert_array_equal(pi, i)
    np.testing.assert_array_equal(oi, i)
    if with_normals:
        np.testing.assert_array_equal(pn.view(np.uint32), n.view(np.uint32))
        np.testing.assert_array_equal(on.view(np.uint32), n.view(np.uint32))
    else:
        assert pn is None and on is None


def test_ply_reference_quirks(rt, orc, tmp_path):
    """Slots count every non-list property whichever element it belongs to; slot names match by suffix; a
    CRLF header is accepted; faces read tokens 1..3 (TriangleMesh.java:41-70,146-151)."""
    text = ("ply\r\nformat ascii 1.0\r\nelement vertex 3\r\nproperty float confidence\r\nproperty float x\r\nproperty float y\r\n"
            "property float z\r\nelement face 1\r\nproperty list uchar int vertex_indices\r\nend_header\r\n"
            "0.5 0 0 0 \r\n0.5 1 0 0 \r\n0.5 0 1 0 \r\n3 0 1 2 \r\n")
    path = str(tmp_path / "q.ply")
    open(path, "w", newline="").write(text)
    for loader in (rt.load_ply, orc.ply_load):
        v, i, n = loader(path)
        np.testing.assert_array_equal(v, np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32))
        np.testing.assert_array_equal(i, [[0, 1, 2]])
        assert n is None
    # two spaces between fields produce an empty token: the reference's Float.parseFloat("") throws
    bad = str(tmp_path / "bad.ply")
    open(bad, "w").write(text.replace("\r\n", "\n").replace("0.5 1 0 0", "0.5  1 0 0"))
    with pytest.raises(Exception):
        rt.load_ply(bad)
    with pytest.raises(Exception):
        orc.ply_load(bad)
    with pytest.raises(rt._abi.Mi355rtError) as e:
        rt.load_ply(str(tmp_path / "missing.ply"))
    assert e.value.code == rt._abi.E_IO


@pytest.mark.parametrize("binary", [False, True])
def test_ply_counts_beyond_the_file_are_an_error_not_an_allocation(rt, tmp_path, binary):
    """A short file whose header declares two billion vertices / faces must come back as MI355RT_E_IO (the
    reference exits with an exception, TriangleMesh.java:90-93) -- never as a 48 GB request or an exception
    leaving the C ABI (which would terminate the host process)."""
    fmt = "binary_big_endian" if binary else "ascii"
    head = ("ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "element face %d\nproperty list uchar int vertex_indices\nend_header\n")
    for nv, nf in ((2000000000, 1), (3, 2000000000), (2147483647, 2147483647)):
        path = str(tmp_path / ("huge_%d_%d.ply" % (nv, nf)))
        with open(path, "wb") as f:
            f.write((head % (fmt, nv, nf)).encode())
            f.write(b"0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n" if not binary else b"\0" * 49)
        with pytest.raises(rt._abi.Mi355rtError) as e:
            rt.load_ply(path)
        assert e.value.code == rt._abi.E_IO
    # an ASCII file with fewer face lines than declared keeps zero rows, as the reference's int[numFaces][3] does
    path = str(tmp_path / "short_faces.ply")
    open(path, "w").write((head % ("ascii", 3, 3)) + "0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    v, i, n = rt.load_ply(path)
    np.testing.assert_array_equal(i, [[0, 1, 2], [0, 0, 0], [0, 0, 0]])


def test_compute_normals_library_vs_oracle(rt, orc):
    v, i = rt.meshgen.blob(2000, span=0.2, seed=8)
    got = rt.compute_normals(v, i)
    exp = np.empty_like(v)
    orc.lib().orc_compute_normals(orc.P(v), v.shape[0], orc.PI(i), i.shape[0], orc.P(exp))
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    assert (np.einsum("ij,ij->i", got, v) > 0).mean() > 0.9     # outward on a star-shaped blob
    # degenerate face -> NaN normal spreads to its vertices; unreferenced vertex -> 0/0 (TriangleMesh.java:96-126)
    vv = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [5, 5, 5]], dtype=np.float32)
    ii = np.array([[0, 1, 2]], dtype=np.int32)
    assert np.isnan(rt.compute_normals(vv, ii)).all()


def test_tile_bookkeeping():
    import torch
    from ray_tracer_amd import tiles
    W, H, T = 100, 70, 16
    tx, ty = tiles.tile_grid(W, H, T)
    assert (tx, ty) == (7, 5)
    seen = torch.zeros(W * H, dtype=torch.int32)
    for world in (1, 2, 3, 8):
        seen.zero_()
        for r in range(world):
            pix = tiles.tile_pixels(W, H, tiles.owned_tiles(W, H, r, world, T), T)
            valid = pix[pix >= 0]
            seen[valid] += 1
            assert pix.shape[0] <= tiles.max_owned(W, H, world, T)
        assert bool((seen == 1).all())                                    # every pixel owned exactly once
    # the compact block layout (what the library writes with settings.tile_compact) and back: every rank's tiles of
    # one frame -> blocks -> image order
    rgb = torch.rand(H, W, 3)
    w = torch.rand(H, W)
    ids = torch.randint(-1, 1000, (H, W), dtype=torch.int32)
    objs = torch.randint(-1, 5, (H, W), dtype=torch.int32)
    for world in (1, 3, 4):
        slots = tiles.max_owned(W, H, world, T)
        n = slots * T * T
        blocks = torch.full((world * tiles.PLANES * n,), -7.0)
        for r in range(world):
            tiles.tiles_pack_reference(rgb, ids, objs, w, r, world, T, tiles.Block(blocks[r * tiles.PLANES * n: (r + 1) * tiles.PLANES * n], n))
        out = (torch.zeros_like(rgb), torch.zeros_like(ids), torch.zeros_like(objs), torch.zeros_like(w))
        tiles.tiles_unpack_reference(blocks, world, slots, T, *out)
        assert torch.equal(out[0], rgb) and torch.equal(out[1], ids) and torch.equal(out[2], objs) and torch.equal(out[3], w)


def test_brdf_parse_rule(rt, tmp_path):
    """BRDFMaterial.parseBRDFFile (BRDFMaterial.java:109-116): drop token 0, keep tokens containing a digit."""
    p = tmp_path / "t.txt"
    p.write_text("1 0.5 -1.25 abc 3e-2 \n2  7 x\n")
    rows = rt.BRDFMaterial.parseBRDFFile(str(p))
    assert [list(map(float, r)) for r in rows] == [[0.5, -1.25, float(f32(3e-2))], [7.0]]


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources"), reason="reference data files not present (GPU box)")
def test_reference_brdf_tables_parse(rt, orc):
    """The reference's own data files (read-only, not copied): 205 rows x 4 angles, 61 rows x 205 weights; row 2
    of table.txt has phi_v = 0."""
    ang = rt.BRDFMaterial.parseBRDFFile("/root/reference/resources/table.txt")
    wts = rt.BRDFMaterial.parseBRDFFile("/root/reference/resources/abrdf.dat")
    assert len(ang) == 205 and all(len(r) == 4 for r in ang)
    assert len(wts) == 61 and all(len(r) == 205 for r in wts)
    ang = np.array(ang, dtype=np.float32)
    wts = np.array(wts, dtype=np.float32)
    assert np.isfinite(ang).all() and np.isfinite(wts).all()
    # the oracle accepts them and a luminance query lands in a sane range
    v, i = rt.meshgen.blob(100, span=0.2)
    s = orc.OracleScene([(v, i, None, orc.material(kind=1, diffuse=(0, 1, 0), brdf_id=15))])
    s.set_brdf(ang, wts)
    lum = orc.lib().orc_brdf_luminocity(s.h, 15, 1, orc.vec(0, 0, -1), orc.vec(0, 0, 0), orc.vec(0, 0, 1), orc.vec(0, 100, 100))
    assert 0.0 <= lum < 20.0
    s.close()
