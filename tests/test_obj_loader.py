"""CPU: Scene::load for OBJ+MTL (scene.rs:22-85, loader/obj.rs) -- BASELINE.json configs[0] goes OBJ -> Scene -> BVH."""
import os

import numpy as np
import pytest


def test_cornell_obj_roundtrip(rrt, tmp_path):
    from rust_ray_tracing_amd import synth
    path = synth.write_cornell_obj(str(tmp_path))
    sc = rrt.Scene.load(path)
    assert sc is not None and len(sc.tris) == 12
    tris, mats, texs, cam = synth.cornell_box()
    ref = rrt.Scene.from_arrays(tris, mats, texs)
    assert list(sc.materials.keys()) == ["white", "red", "green", "light"]
    assert sc.tris.tobytes() == ref.tris.tobytes()            # same expansion, same quad split, same BVH order
    assert sc.bvh_nodes.tobytes() == ref.bvh_nodes.tobytes()
    assert sc.materials_array().tobytes() == ref.materials_array().tobytes()
    m = sc.materials["light"]
    assert tuple(m["emission"]) == (4.0, 3.5, 3.0) and m["ior"] == np.float32(1.45) and m["base_color_tex_id"] == 0xFFFFFFFF


def test_faces_fan_flat_normals_defaults(rrt, tmp_path):
    p = tmp_path / "fan.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0.5 1.5 0\nv 0 1 0\nf 1 2 3 4 5\nf 1//1 2//1 3//1\n")
    sc = rrt.Scene.load(str(p))
    assert sc is not None and len(sc.tris) == 4               # 5-gon fan = 3 triangles (obj.rs:421-432) + 1
    assert list(sc.materials.keys()) == ["default_material"]  # no mtllib (obj.rs:44-50)
    assert sc.materials["default_material"]["base_color"][0] == np.float32(0.8)
    # no `vn` lines: flat normals synthesised per triangle (obj.rs:106-120); all faces lie in z = 0 -> (0,0,1)
    n = sc.tris["vertices"]["normal"]
    assert np.allclose(np.abs(n[..., 2]), 1.0) and np.allclose(n[..., :2], 0.0)
    # missing vt index -> tex coord 0 (scene.rs:55-60)
    assert np.all(sc.tris["vertices"]["tex_coord_x"] == 0)


def test_loader_error_conventions(rrt, tmp_path, capsys):
    assert rrt.Scene.load(str(tmp_path / "missing.obj")) is None           # scene.rs:23-26
    assert "Could not find scene" in capsys.readouterr().err
    q = tmp_path / "scene.gltf"
    q.write_text("{}")
    assert rrt.Scene.load(str(q)) is None                                   # scene.rs:31-34
    assert "Unsupported scene format" in capsys.readouterr().err
    neg = tmp_path / "neg.obj"
    neg.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -3 -2 -1\n")
    assert rrt.Scene.load(str(neg)) is None                                 # reference panics (obj.rs:356-362)
    nomtl = tmp_path / "nomtl.obj"
    nomtl.write_text("mtllib nothere.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    assert rrt.Scene.load(str(nomtl)) is None
    empty = tmp_path / "empty.obj"
    empty.write_text("v 0 0 0\n")
    assert rrt.Scene.load(str(empty)) is None                               # no faces: BVH::build would panic


def test_mtl_blank_line_terminates_material_and_ppm_texture(rrt, tmp_path):
    (tmp_path / "t.ppm").write_bytes(b"P6\n2 2\n255\n" + bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 9, 9, 9]))
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nmap_Kd t.ppm\nnewmtl ignored_without_blank_line\nKd 1 1 1\n\nnewmtl b\nKe 1 2 3\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl b\nf 1 2 3\n")
    sc = rrt.Scene.load(str(tmp_path / "m.obj"))
    assert sc is not None and list(sc.materials.keys()) == ["a", "b"]
    # the reference's inner loop swallows lines until a blank one (obj.rs:141-147): the second newmtl is
    # ignored and its Kd overwrites material a's
    assert tuple(sc.materials["a"]["base_color"]) == (1.0, 1.0, 1.0)
    assert sc.tris[0]["material_id"] == 1
    assert len(sc.textures) == 1 and sc.materials["a"]["base_color_tex_id"] == 0
    # Texture::load flips vertically (texture.rs:18): stored row 0 is the file's last row
    assert sc.textures[0][0, 0].tolist() == [0, 0, 255, 255] and sc.textures[0][1, 0].tolist() == [255, 0, 0, 255]


def _png(path, arr, ctype, filters=None, palette=None, trns=None, depth=8, chunk=7777):
    """Writes a PNG with Python's zlib (dynamic-Huffman deflate) and explicit per-row filter types."""
    import struct
    import zlib
    h, w = arr.shape[:2]
    rows = arr.reshape(h, -1).astype(np.uint8)
    bpp = rows.shape[1] // w
    raw = bytearray()
    prev = np.zeros(rows.shape[1], dtype=np.int32)
    for y in range(h):
        ft = (filters or [0])[y % len(filters or [0])]
        cur = rows[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0: enc = cur
        elif ft == 1: enc = cur - a
        elif ft == 2: enc = cur - prev
        elif ft == 3: enc = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            enc = cur - pred
        raw.append(ft)
        raw += (enc & 255).astype(np.uint8).tobytes()
        prev = cur

    def ch(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw), 9)
    out = b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        out += ch(b"PLTE", bytes(palette))
    if trns is not None:
        out += ch(b"tRNS", bytes(trns))
    for i in range(0, len(z), chunk):                      # several IDAT chunks
        out += ch(b"IDAT", z[i:i + chunk])
    out += ch(b"IEND", b"")
    open(path, "wb").write(out)


def test_png_textures_all_colour_types_and_filters(rrt, tmp_path):
    """map_Kd PNGs (texture.rs:13-31 via the built-in decoder): RGBA / RGB / grey / grey+alpha / palette+tRNS, all five
    scanline filters, multi-chunk IDAT; stored v-flipped like Texture::load."""
    rng = np.random.default_rng(9)
    w, h = 37, 21
    grad = (np.add.outer(np.arange(h) * 5, np.arange(w) * 3) % 256).astype(np.uint8)
    rgba = np.stack([grad, grad.T[:h, :w] if grad.T.shape == grad.shape else grad[::-1], rng.integers(0, 256, (h, w), dtype=np.uint8),
                     rng.integers(0, 256, (h, w), dtype=np.uint8)], axis=-1)
    idx = rng.integers(0, 5, (h, w), dtype=np.uint8)
    pal = rng.integers(0, 256, 15).astype(np.uint8)
    trns = [10, 200, 255]
    cases = {
        "rgba": (rgba, 6, rgba),
        "rgb": (rgba[..., :3], 2, np.concatenate([rgba[..., :3], np.full((h, w, 1), 255, np.uint8)], -1)),
        "grey": (rgba[..., 0], 0, np.stack([rgba[..., 0]] * 3 + [np.full((h, w), 255, np.uint8)], -1)),
        "ga": (rgba[..., [0, 3]], 4, np.stack([rgba[..., 0]] * 3 + [rgba[..., 3]], -1)),
        "pal": (idx, 3, np.concatenate([pal.reshape(5, 3)[idx], np.array(trns + [255, 255], np.uint8)[idx][..., None]], -1)),
    }
    mtl, obj = [], ["mtllib t.mtl", "v 0 0 0", "v 1 0 0", "v 0 1 0"]
    for name, (arr, ctype, _) in cases.items():
        _png(str(tmp_path / f"{name}.png"), arr, ctype, filters=[0, 1, 2, 3, 4], palette=pal if ctype == 3 else None,
             trns=trns if ctype == 3 else None, chunk=97)
        mtl.append(f"newmtl m_{name}\nmap_Kd {name}.png\n")
        obj += [f"usemtl m_{name}", "f 1 2 3"]
    (tmp_path / "t.mtl").write_text("\n".join(mtl))
    (tmp_path / "t.obj").write_text("\n".join(obj) + "\n")
    sc = rrt.Scene.load(str(tmp_path / "t.obj"))
    assert sc is not None and len(sc.textures) == len(cases)
    for name, (_, _, want) in cases.items():
        tid = int(sc.materials[f"m_{name}"]["base_color_tex_id"])
        assert tid != 0xFFFFFFFF
        assert np.array_equal(sc.textures[tid], want[::-1]), name          # flipv(): stored row 0 = bottom image row
    # a corrupt PNG is skipped like a missing texture (Texture::load -> None), the scene still loads
    (tmp_path / "rgba.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    sc2 = rrt.Scene.load(str(tmp_path / "t.obj"))
    assert sc2 is not None and int(sc2.materials["m_rgba"]["base_color_tex_id"]) == 0xFFFFFFFF


# ---- JPEG textures (jpeg_decode.cpp) -------------------------------------------------------------------------------
def _jpeg_test_image(h, w, seed=1):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos(x / 11.0 + y / 3.0), (x * 3 + y * 5) % 256], -1)
    return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)


def _pil_rgb_flipped(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))[::-1]


@pytest.mark.parametrize("size", [(64, 64), (37, 53), (1, 1), (8, 8), (17, 3), (3, 17), (2, 2), (5, 4), (100, 131)])
def test_jpeg_decoder_matches_libjpeg_bit_for_bit(rrt, tmp_path, size):
    """Texture::load on JPEG files (texture.rs:18) through the built-in decoder, against Pillow's libjpeg(-turbo) as the
    known-answer source: baseline / progressive (incl. successive-approximation refinement scans) / optimised tables,
    4:4:4 / 4:2:2 / 4:2:0 chroma with odd sizes (edge replication, the narrow-component upsampling rule), greyscale,
    restart intervals, custom quantisation tables.  Whether the `image` crate rounds identically is not pinned."""
    Image = pytest.importorskip("PIL.Image")
    h, w = size
    img = _jpeg_test_image(h, w)
    n = 0
    variants = [dict(), dict(progressive=True), dict(optimize=True), dict(quality=30), dict(quality=100), dict(progressive=True, quality=95),
                dict(restart_marker_blocks=1), dict(restart_marker_rows=1, progressive=True), dict(qtables=[list(range(1, 65)), [3] * 64])]
    for sub in (0, 1, 2):
        for kw in variants:
            p = str(tmp_path / f"t{n}.jpg"); n += 1
            try:
                Image.fromarray(img).save(p, "JPEG", subsampling=sub, **{"quality": 75, **kw})
            except (TypeError, ValueError):                    # an encoder option this Pillow does not know
                continue
            t = rrt.Texture.load(p)
            assert t is not None, (sub, kw)
            ref = _pil_rgb_flipped(p)
            assert (t.height, t.width) == ref.shape[:2]
            assert np.array_equal(t.pixel_data[..., :3], ref), (sub, kw)
            assert (t.pixel_data[..., 3] == 255).all()
    for kw in (dict(), dict(progressive=True)):
        p = str(tmp_path / f"g{n}.jpeg"); n += 1               # .jpeg extension, one component
        Image.fromarray(img[..., 0]).save(p, "JPEG", quality=80, **kw)
        t = rrt.Texture.load(p)
        assert t is not None and np.array_equal(t.pixel_data[..., :3], _pil_rgb_flipped(p))


def test_jpeg_real_files_when_present(rrt):
    """Camera-made JPEGs that ship with Python packages in this image (not with the reference): same bit-for-bit check."""
    pytest.importorskip("PIL.Image")
    import glob
    files = []
    for pat in ("/usr/local/lib/python3*/dist-packages/sklearn/datasets/images/*.jpg",
                "/usr/local/lib/python3*/dist-packages/matplotlib/mpl-data/sample_data/*.jpg"):
        files += sorted(glob.glob(pat))
    if not files:
        pytest.skip("no sample JPEGs on this machine")
    for f in files:
        t = rrt.Texture.load(f)
        assert t is not None, f
        assert np.array_equal(t.pixel_data[..., :3], _pil_rgb_flipped(f)), f


def test_jpeg_texture_through_obj_and_rejections(rrt, tmp_path, capfd):
    Image = pytest.importorskip("PIL.Image")
    img = _jpeg_test_image(24, 40, seed=3)
    Image.fromarray(img).save(str(tmp_path / "kd.JPG"), "JPEG", quality=90, subsampling=2)
    Image.fromarray(img).save(str(tmp_path / "same.jpg"), "JPEG", quality=90, subsampling=2)      # same pixels: de-duplicated by hash
    Image.fromarray(np.dstack([img, img[..., :1]])).convert("CMYK").save(str(tmp_path / "cmyk.jpg"), "JPEG")
    (tmp_path / "junk.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    (tmp_path / "t.mtl").write_text("newmtl a\nmap_Kd kd.JPG\n\nnewmtl b\nmap_Kd same.jpg\n\nnewmtl c\nmap_Kd cmyk.jpg\n\nnewmtl d\nmap_Kd junk.jpg\n")
    (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl b\nf 1 2 3\nusemtl c\nf 1 2 3\nusemtl d\nf 1 2 3\n")
    sc = rrt.Scene.load(str(tmp_path / "t.obj"))
    assert sc is not None and len(sc.textures) == 1
    ta, tb = int(sc.materials["a"]["base_color_tex_id"]), int(sc.materials["b"]["base_color_tex_id"])
    assert ta == tb == 0
    assert np.array_equal(sc.textures[0][..., :3], _pil_rgb_flipped(str(tmp_path / "kd.JPG")))
    assert int(sc.materials["c"]["base_color_tex_id"]) == 0xFFFFFFFF and int(sc.materials["d"]["base_color_tex_id"]) == 0xFFFFFFFF
    assert "CMYK" in capfd.readouterr().err
    # Texture::load on a missing file: the reference's log line, None
    assert rrt.Texture.load(str(tmp_path / "nope.jpg")) is None
    assert "Could not find texture at path" in capfd.readouterr().err
    # the hash the loader de-duplicates by is djb2 over every 4th pixel word (texture.rs:40-48)
    t = rrt.Texture.load(str(tmp_path / "kd.JPG"))
    words = t.pixel_data.reshape(-1, 4).copy().view("<u4").reshape(-1)[::4]
    hsh = 5381
    for wd in words.tolist():
        hsh = (hsh * 33 + wd) & 0xFFFFFFFF
    assert hsh == t.hash


# ---- PNG: every colour type / bit depth, Adam7, colour keys ----------------------------------------------------------
def _png_general(path, samples, ctype, depth, interlace=False, palette=None, trns=None):
    """samples: (h, w, channels) integers at `depth` bits.  Packs, (Adam7-)splits, filters (types cycle 0..4), deflates."""
    import struct
    import zlib
    h, w, ch_n = samples.shape

    def pack_rows(img):                                     # img (ph, pw, ch) -> list of packed byte rows
        ph, pw, _ = img.shape
        flat = img.reshape(ph, pw * ch_n).astype(np.uint32)
        if depth == 16:
            return [np.stack([r >> 8, r & 255], -1).astype(np.uint8).reshape(-1) for r in flat]
        if depth == 8:
            return [r.astype(np.uint8) for r in flat]
        per = 8 // depth
        out = []
        for r in flat:
            pad = (-len(r)) % per
            rr = np.concatenate([r, np.zeros(pad, np.uint32)]).reshape(-1, per)
            shifts = np.array([8 - depth * (k + 1) for k in range(per)], np.uint32)
            out.append((rr << shifts).sum(1).astype(np.uint8))
        return out

    bpp = max(1, ch_n * depth // 8)
    raw = bytearray()
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
    n = 0
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for row in pack_rows(sub):
            cur = row.astype(np.int32)
            pv = np.zeros_like(cur) if prev is None else prev
            a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            c = np.concatenate([np.zeros(bpp, np.int32), pv[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            ft = n % 5
            n += 1
            if ft == 0: enc = cur
            elif ft == 1: enc = cur - a
            elif ft == 2: enc = cur - pv
            elif ft == 3: enc = cur - ((a + pv) >> 1)
            else:
                p = a + pv - c
                pa, pb, pc = np.abs(p - a), np.abs(p - pv), np.abs(p - c)
                enc = cur - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, pv, c))
            raw.append(ft)
            raw += (enc & 255).astype(np.uint8).tobytes()
            prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b"")
    open(path, "wb").write(out)


@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_colour_type_and_depth(rrt, tmp_path, interlace):
    """All 15 colour-type / bit-depth combinations of the PNG standard, plain and Adam7-interlaced, odd sizes (so that some
    Adam7 passes are empty or one pixel wide), tRNS as a palette alpha table and as a grey / RGB colour key.  Depths <= 8
    are checked against Pillow's decoder, 16-bit against the (x + 128) / 257 narrowing of the `image` crate."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11 + interlace)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
    n = 0
    for (h, w) in [(13, 21), (1, 1), (2, 3), (5, 1), (9, 8)]:
        for ctype, depths in [(0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))]:
            for depth in depths:
                smp = rng.integers(0, 1 << depth, (h, w, ch[ctype]))
                pal = trns = None
                if ctype == 3:
                    pal = rng.integers(0, 256, 3 << depth).astype(np.uint8)
                    trns = rng.integers(0, 256, max(1, (1 << depth) // 2)).astype(np.uint8)
                elif ctype == 0:
                    k = int(smp[0, 0, 0]); trns = [k >> 8, k & 255]
                elif ctype == 2:
                    trns = [b for v in smp[0, 0] for b in (int(v) >> 8, int(v) & 255)]
                p = str(tmp_path / f"p{n}.png"); n += 1
                _png_general(p, smp, ctype, depth, interlace=interlace, palette=pal, trns=trns)
                t = rrt.Texture.load(p)
                assert t is not None, (h, w, ctype, depth)
                got = t.pixel_data[::-1]
                # expectation from the standard: low-depth grey scaled to the full range, 16 bit rounded, key compared at file depth
                if depth == 16:
                    s8 = ((smp.astype(np.uint32) + 128) // 257).astype(np.uint8)
                elif depth == 8 or ctype == 3:
                    s8 = smp.astype(np.uint8)
                else:
                    s8 = (smp * (255 // ((1 << depth) - 1))).astype(np.uint8)
                want = np.full((h, w, 4), 255, np.uint8)
                if ctype == 0:
                    want[..., :3] = s8[..., :1]; want[..., 3] = np.where(smp[..., 0] == smp[0, 0, 0], 0, 255)
                elif ctype == 2:
                    want[..., :3] = s8; want[..., 3] = np.where((smp == smp[0, 0]).all(-1), 0, 255)
                elif ctype == 3:
                    idx = smp[..., 0]
                    want[..., :3] = pal.reshape(-1, 3)[idx]
                    want[..., 3] = np.concatenate([trns, np.full((1 << depth) - len(trns), 255, np.uint8)])[idx]
                elif ctype == 4:
                    want[..., :3] = s8[..., :1]; want[..., 3] = s8[..., 1]
                else:
                    want = s8
                assert np.array_equal(got, want), (h, w, ctype, depth, interlace)
                # Pillow as a second opinion where its conversion follows the standard too (it narrows 16 bit differently
                # and compares a low-depth grey key after scaling)
                if depth == 8 or ctype == 3:
                    assert np.array_equal(got, np.asarray(Image.open(p).convert("RGBA"))), (h, w, ctype, depth, interlace)
                elif depth < 8:
                    assert np.array_equal(got[..., :3], np.asarray(Image.open(p).convert("RGB"))), (h, w, ctype, depth, interlace)


def test_png_decoder_on_the_reference_gallery_images(rrt):
    """The reference ships four 1920-wide gallery PNGs (README.md) -- real encoder output (adaptive filters, big deflate
    streams) rather than this file's own writer.  Read in place as data (skipped where /root/reference is absent), decoded
    by Texture::load's path and compared with Pillow's decoder."""
    Image = pytest.importorskip("PIL.Image")
    import glob
    files = sorted(glob.glob("/root/reference/*.png"))
    if not files:
        pytest.skip("/root/reference is not present on this machine")
    for f in files:
        t = rrt.Texture.load(f)
        assert t is not None, f
        want = np.asarray(Image.open(f).convert("RGBA"))[::-1]
        assert (t.height, t.width) == want.shape[:2] and np.array_equal(t.pixel_data, want), f


# ---- TGA / BMP textures (tga_bmp_decode.cpp) -------------------------------------------------------------------------
def test_tga_and_bmp_decoders_match_pillow(rrt, tmp_path):
    """Texture::load on the two other lossless formats OBJ exports carry: every variant Pillow can write (TGA grey / grey+alpha /
    RGB / RGBA / colour-mapped, raw and run-length, both vertical origins; BMP 1 / 8 / 24 / 32 bit incl. BITFIELDS with alpha)
    plus hand-built files it cannot (16-bit 5-5-5 TGA and BMP, 4-bit and top-down BMP, the 12-byte CORE header), all compared
    with Pillow's readers."""
    Image = pytest.importorskip("PIL.Image")
    import struct
    rng = np.random.default_rng(5)

    def check(p, want=None):
        t = rrt.Texture.load(p)
        assert t is not None, p
        if want is None:
            want = np.asarray(Image.open(p).convert("RGBA"))[::-1]
        assert t.pixel_data.shape == want.shape and np.array_equal(t.pixel_data, want), p

    def want555(v16, flip):                                 # 5-bit fields rounded to 8 bits (Pillow truncates instead)
        c = np.stack([(v16 >> 10) & 31, (v16 >> 5) & 31, v16 & 31], -1).astype(np.uint32)
        out = np.concatenate([((c * 255 + 15) // 31).astype(np.uint8), np.full(v16.shape + (1,), 255, np.uint8)], -1)
        return out if flip else out[::-1]

    n = 0
    for (h, w) in [(13, 21), (1, 1), (7, 5), (32, 33)]:
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        rgba[:, : w // 2] = rgba[:1, :1]                    # runs, so the run-length files hold both packet kinds
        for mode, arr in [("RGBA", rgba), ("RGB", rgba[..., :3]), ("L", rgba[..., 0]), ("LA", rgba[..., :2])]:
            im = Image.fromarray(arr, mode)
            for rle in (False, True):
                for orient in (1, -1):
                    p = str(tmp_path / f"t{n}.tga"); n += 1
                    im.save(p, compression="tga_rle" if rle else None, orientation=orient)
                    check(p)
            if mode != "LA":
                p = str(tmp_path / f"b{n}.bmp"); n += 1
                im.save(p); check(p)
        pim = Image.fromarray(rgba[..., :3], "RGB").convert("P", palette=Image.ADAPTIVE, colors=17)
        for ext, kw in (("tga", dict()), ("tga", dict(compression="tga_rle")), ("bmp", dict())):
            p = str(tmp_path / f"p{n}.{ext}"); n += 1
            pim.save(p, **kw); check(p)
        p = str(tmp_path / f"one{n}.bmp"); n += 1
        Image.fromarray((rgba[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1").save(p); check(p)
        # hand-built: 16-bit TGA, bottom-up
        v16 = rng.integers(0, 1 << 15, (h, w)).astype("<u2")
        p = str(tmp_path / f"h{n}.tga"); n += 1
        open(p, "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 16, 0) + v16.tobytes()); check(p, want555(v16, True))
        # hand-built BMPs: 16-bit BI_RGB, 4-bit palette, top-down 24-bit, CORE-header 24-bit
        def bmp(bits, rows_bytes, height_field, palette=b"", core=False):
            hdr = struct.pack("<IHHHH", 12, w, h, 1, bits) if core else struct.pack("<IiiHHIIiiII", 40, w, height_field, 1, bits, 0, 0, 2835, 2835, len(palette) // 4, 0)
            off = 14 + len(hdr) + len(palette)
            body = b"".join(r + b"\0" * ((-len(r)) % 4) for r in rows_bytes)
            return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + hdr + palette + body
        p = str(tmp_path / f"h{n}.bmp"); n += 1
        open(p, "wb").write(bmp(16, [v16[y].tobytes() for y in range(h)], h)); check(p, want555(v16, True))
        idx = rng.integers(0, 16, (h, w)).astype(np.uint8)
        packed = [bytes((int(r[i]) << 4) | (int(r[i + 1]) if i + 1 < w else 0) for i in range(0, w, 2)) for r in idx]
        p = str(tmp_path / f"h{n}.bmp"); n += 1
        open(p, "wb").write(bmp(4, packed, h, palette=rng.integers(0, 256, 64, dtype=np.uint8).tobytes())); check(p)
        p = str(tmp_path / f"h{n}.bmp"); n += 1
        open(p, "wb").write(bmp(24, [rgba[y, :, 2::-1].tobytes() for y in range(h)], -h)); check(p)
        p = str(tmp_path / f"h{n}.bmp"); n += 1
        open(p, "wb").write(bmp(24, [rgba[y, :, 2::-1].tobytes() for y in range(h)], h, core=True)); check(p)
    # rejected cleanly
    (tmp_path / "bad.tga").write_bytes(b"\0" * 10)
    (tmp_path / "bad.bmp").write_bytes(b"BM" + b"\0" * 40)
    assert rrt.Texture.load(str(tmp_path / "bad.tga")) is None and rrt.Texture.load(str(tmp_path / "bad.bmp")) is None


def test_png_hostile_sizes_are_rejected_before_allocation(rrt, tmp_path):
    """A ~100-byte PNG must not be able to make the loader allocate gigabytes: dimensions beyond the 2^27-pixel budget and
    zlib streams that inflate past the size IHDR implies are refused."""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    def png(w, h, raw):
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))
    huge = tmp_path / "huge.png"
    huge.write_bytes(png(60000, 60000, b"\0" * 64))
    assert rrt.Texture.load(str(huge)) is None
    bomb = tmp_path / "bomb.png"
    bomb.write_bytes(png(4, 4, b"\0" * (64 << 20)))                    # 64 MiB of zeros compress to ~64 KiB; the image needs 68 bytes
    assert len(bomb.read_bytes()) < 200_000
    assert rrt.Texture.load(str(bomb)) is None
    ok = tmp_path / "ok.png"
    ok.write_bytes(png(4, 4, b"".join(b"\0" + bytes(range(16 * y, 16 * y + 16)) for y in range(4))))
    t = rrt.Texture.load(str(ok))
    assert t is not None and t.width == 4 and t.height == 4


# ---- the chunked, multi-threaded parser (csrc/obj_loader.cpp) against a line-by-line restatement of obj.rs ------------------------
def _load_triangles(rrt, path):
    """mipt_obj_load_triangles -> (tris, material names) or the negative status"""
    import ctypes as C
    from rust_ray_tracing_amd import _lib as L
    import time
    lib = rrt.load()
    obj = C.c_void_p()
    t0 = time.time()
    rc = lib.mipt_obj_load_triangles(os.fsencode(path), C.byref(obj))
    _load_triangles.seconds = time.time() - t0                            # the C call alone (the copies below are the test's)
    if rc != 0:
        return rc, None
    try:
        desc = L.MiptSceneDesc()
        names = C.POINTER(C.c_char_p)()
        assert lib.mipt_obj_get(obj, C.byref(desc), C.byref(names)) == 0
        assert desc.n_nodes == 0 and not desc.nodes                       # no BVH::build in this entry
        tris = np.ctypeslib.as_array(C.cast(desc.tris, C.POINTER(C.c_uint8)), (desc.n_tris * 112,)).copy().view(L.TRIANGLE)
        return tris, [names[i].decode() for i in range(desc.n_materials)]
    finally:
        lib.mipt_obj_free(obj)


def _ref_obj_parse(text, material_names, has_mtl=True):
    """obj.rs:54-120 + scene.rs:44-85 read line by line (str::lines, split_whitespace, parse::<f32>, Triangle::from_str) -> Triangle array, or
    None where the reference panics."""
    from rust_ray_tracing_amd import TRIANGLE
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    lines = [l[:-1] if l.endswith("\r") else l for l in lines]
    pos, tex, nrm, tris = [], [], [], []
    active = 0

    def f32(tok):
        return np.float32(float(tok))

    def idx(tok):
        v = int(tok) - 1
        if v < 0:
            raise ValueError("negative index")
        return v

    def group(g):
        p = t = n = 0
        if "//" in g:
            parts = g.split("//")
            p = idx(parts[0])
            if len(parts) > 1:
                n = idx(parts[1])
        elif "/" in g:
            parts = g.split("/")
            if len(parts) == 2:
                p, t = idx(parts[0]), idx(parts[1])
            elif len(parts) == 3:
                p, t, n = idx(parts[0]), idx(parts[1]), idx(parts[2])
        else:
            p = idx(g)
        return p, t, n
    try:
        for line in lines:
            tok = line.split()
            if not tok:
                continue
            if tok[0] in ("v", "vn"):
                if len(tok) - 1 > 3:
                    return None
                d = [np.float32(0)] * 3
                for i, v in enumerate(tok[1:]):
                    d[i] = f32(v)
                (pos if tok[0] == "v" else nrm).append(d)
            elif tok[0] == "vt":
                if len(tok) - 1 > 2:
                    return None
                d = [np.float32(0)] * 2
                for i, v in enumerate(tok[1:]):
                    d[i] = f32(v)
                tex.append(d)
            elif tok[0] == "usemtl" and has_mtl:
                if not line.startswith("usemtl "):
                    return None
                name = line[7:]
                if name in material_names:
                    active = material_names.index(name)
            elif tok[0] == "f":
                if not line.startswith("f "):
                    return None
                g = line[2:].split()
                if len(g) == 3:
                    sets = [(0, 1, 2)]
                elif len(g) == 4:
                    sets = [(0, 1, 3), (1, 2, 3)]
                elif len(g) >= 5:
                    sets = [(0, i + 1, i + 2) for i in range(len(g) - 2)]
                else:
                    return None
                for s in sets:
                    tris.append(([group(g[k]) for k in s], active))
    except ValueError:
        return None
    if not tris:
        return None
    if not nrm:                                                             # flat normals (obj.rs:106-120); not used by the fuzz below
        raise NotImplementedError
    out = np.zeros(len(tris), dtype=TRIANGLE)
    for i, (gs, mat) in enumerate(tris):
        for k, (p, t, n) in enumerate(gs):
            if p < len(pos):
                out[i]["vertices"][k]["position"] = pos[p]
            if t < len(tex):
                out[i]["vertices"][k]["tex_coord_x"], out[i]["vertices"][k]["tex_coord_y"] = tex[t]
            if n < len(nrm):
                out[i]["vertices"][k]["normal"] = nrm[n]
        out[i]["material_id"] = mat
    return out


def _fuzz_obj(seed, n_lines, eol="\n"):
    """Random OBJ text: v / vt / vn in any interleaving, faces with every index form that point forwards and backwards (and past the
    end: missing -> zeros), quads, n-gons, usemtl switches incl. unknown names, comments, odd spacing, number spellings."""
    rng = np.random.default_rng(seed)
    spell = [lambda x: repr(float(np.float32(x))), lambda x: "%.9g" % x, lambda x: "%+.8e" % x, lambda x: "%.3f" % x, lambda x: str(int(x)), lambda x: "%.6E" % x]

    def num():
        return spell[rng.integers(len(spell))](float(rng.uniform(-100, 100)))
    out = ["mtllib f.mtl"]
    hi = max(3 * n_lines // 8, 4)
    for _ in range(n_lines):
        r = rng.integers(0, 100)
        sep = " " if r % 5 else "  \t "
        if r < 25:
            out.append("v" + sep + sep.join(num() for _ in range(rng.integers(1, 4))))       # missing components read as 0
        elif r < 40:
            out.append("vt" + sep + sep.join(num() for _ in range(rng.integers(0, 3))))
        elif r < 55:
            out.append(("  " if r % 2 else "") + "vn" + sep + sep.join(num() for _ in range(3)))
        elif r < 62:
            out.append("usemtl " + ["a", "b", "c c", "missing", "b "][rng.integers(5)])
        elif r < 66:
            out.append(["# a comment", "", "o thing", "s off", "g group1", "   "][rng.integers(6)])
        else:
            k = [3, 3, 3, 4, 4, 5, 7][rng.integers(7)]
            form = rng.integers(0, 5)
            gs = []
            for _ in range(k):
                a, b, c = (int(rng.integers(1, hi + hi // 8)) for _ in range(3))
                gs.append([f"{a}", f"{a}/{b}", f"{a}/{b}/{c}", f"{a}//{c}", f"+{a}/{b}/{c}"][form])
            out.append("f " + sep.join(gs) + ("  " if r % 3 == 0 else ""))
    return eol.join(out) + (eol if seed % 2 else "")


@pytest.mark.parametrize("seed,n_lines,eol", [(1, 200, "\n"), (2, 3000, "\r\n"), (3, 90000, "\n"), (4, 90000, "\r\n"), (5, 150000, "\n")])
def test_chunked_parser_equals_line_by_line_reading(rrt, tmp_path, seed, n_lines, eol):
    """Files from a few KB (one chunk) to several MB (up to 64 chunks on their own threads): the arrays equal a line-by-line reading of
    obj.rs -- vertex data concatenated in file order, absolute indices, the active material carried across chunk boundaries."""
    (tmp_path / "f.mtl").write_text("newmtl a\nKd 1 0 0\n\nnewmtl b\nKd 0 1 0\n\nnewmtl c c\nKd 0 0 1\n")
    text = _fuzz_obj(seed, n_lines, eol)
    p = tmp_path / "f.obj"
    p.write_bytes(text.encode())
    got, names = _load_triangles(rrt, str(p))
    assert names == ["a", "b", "c c"]
    want = _ref_obj_parse(text, names)
    assert want is not None and len(got) == len(want)
    if got.tobytes() != want.tobytes():
        bad = int(np.flatnonzero([a.tobytes() != b.tobytes() for a, b in zip(got, want)])[0])
        raise AssertionError(f"triangle {bad} differs:\n{got[bad]}\n{want[bad]}")
    assert len(set(got["material_id"].tolist())) == 3


def test_chunked_parser_reports_the_first_error_in_file_order(rrt, tmp_path):
    from rust_ray_tracing_amd import _lib as L
    (tmp_path / "f.mtl").write_text("newmtl a\n")
    text = _fuzz_obj(7, 120000)                                               # several MB: many chunks
    lines = text.split("\n")
    for where, bad_line, msg in ((len(lines) // 3, "v 1 2 3 4", b"more than 3 components"), (2 * len(lines) // 3, "f 1 2 -3", b"malformed face"),
                                 (len(lines) - 5, "vt 0.5 zzz", b"bad number")):
        lines[where] = bad_line
    p = tmp_path / "f.obj"
    p.write_text("\n".join(lines))
    rc, _ = _load_triangles(rrt, str(p))
    assert rc == L.ERR_IO and b"more than 3 components" in rrt.load().mipt_last_error()      # the earliest of the three
    lines[len(lines) // 3] = "v 1 2 3"
    p.write_text("\n".join(lines))
    rc, _ = _load_triangles(rrt, str(p))
    assert rc == L.ERR_IO and b"malformed face" in rrt.load().mipt_last_error()


def test_number_spellings_parse_as_strtof(rrt, tmp_path):
    """The fast decimal path (exact double arithmetic + float-midpoint detection) against the C library's strtof (what the line loader
    called; NOT float() rounded to f32, which rounds twice): halfway cases, 17+ digit mantissas, large / small exponents, signs, specials."""
    import ctypes
    import struct
    libc = ctypes.CDLL(None)
    libc.strtof.restype = ctypes.c_float
    libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    toks = ["1", "-0", "+0.0", ".5", "5.", "1e5", "1E-5", "+1.5e+3", "0.1", "16777217", "16777216.999999999", "0.000000000000000000001e21",
            "3.4028235e38", "3.4028236e38", "1e-45", "7e-46", "1.17549435e-38", "123456789012345678901234567890", "inf", "-inf", "nan", "infinity",
            "1.00000005960464477539062500", "1.0000001788139343", "8388608.5", "8388609.5", "0.3333333432674407958984375"]
    for _ in range(3000):                                                      # exact float midpoints and their neighbours, written in full
        f = np.float32(rng.uniform(-1, 1) * 10.0 ** rng.integers(-6, 7))
        g = np.nextafter(f, np.float32(np.inf))
        mid = (float(f) + float(g)) / 2.0                                      # exactly representable as a double
        toks.append("%.40g" % mid)
        toks.append("%.17g" % np.nextafter(mid, np.inf))
        toks.append("%.9g" % f)
    with open(tmp_path / "n.obj", "w") as fh:
        for t in toks:
            fh.write(f"v {t} 0 0\n")
        fh.write("vn 0 0 1\nf 1 2 3\n")
        fh.write("v 0x10 0 0\n")
    sc = rrt.Scene.load(str(tmp_path / "n.obj"))
    assert sc is not None
    # read the positions back through faces: triangle i = (v_{3i+1}, ...): simpler -- one face per vertex triple
    with open(tmp_path / "n2.obj", "w") as fh:
        for t in toks:
            fh.write(f"v {t} 0 0\n")
        fh.write("vn 0 0 1\n")
        for i in range(0, len(toks) - 2, 3):
            fh.write(f"f {i + 1}//1 {i + 2}//1 {i + 3}//1\n")
    got, _ = _load_triangles(rrt, str(tmp_path / "n2.obj"))
    xs = got["vertices"]["position"][:, :, 0].reshape(-1)
    for i, t in enumerate(toks[: len(xs)]):
        want = np.float32(libc.strtof(t.encode(), None))
        a, b = struct.pack("<f", xs[i]), struct.pack("<f", want)
        assert a == b or (np.isnan(xs[i]) and np.isnan(want)), (t, xs[i], want)


def test_obj_export_and_load_one_million_triangles(rrt, tmp_path):
    """The on-ramp at the size the configs name: the 1 M-triangle atrium written as .obj + .mtl + PNG textures
    (synth.write_obj) and read back -- triangles bit for bit, materials field for field, textures pixel for pixel -- with the
    parse rate of the loader on this machine's cores."""
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.make_scene("atrium", n_target=1_000_000, tex_size=64)
    path = synth.write_obj(str(tmp_path), "atrium1m", tris, mats, texs)
    size = os.path.getsize(path)
    got, names = _load_triangles(rrt, path)
    dt = _load_triangles.seconds
    rate = size / dt / 1e6
    print(f"\n1 M-triangle OBJ: {size / 1e6:.0f} MB, {len(got)} triangles, load {dt:.2f} s = {rate:.0f} MB/s on {os.cpu_count()} CPUs (incl. the PNG textures)")
    assert len(got) == len(tris) and got.tobytes() == np.ascontiguousarray(tris).tobytes()
    assert names == (list(mats.keys()) if isinstance(mats, dict) else [f"material_{i}" for i in range(len(mats))])
    assert rate > 300.0                                                       # measured here: 900 MB/s on 8 cores; round 3's loader: 19 MB/s
    # the full Scene::load (with BVH::build) agrees with the array path
    sc = rrt.Scene.load(path)
    ref = rrt.Scene.from_arrays(tris, mats, texs)
    assert sc.tris.tobytes() == ref.tris.tobytes() and sc.bvh_nodes.tobytes() == ref.bvh_nodes.tobytes()
    # materials: every field but the texture slots (the loader numbers textures in order of first use, texture.rs:40-48)
    ma, mb = sc.materials_array(), ref.materials_array()
    for f_ in ("base_color", "transmission", "specular_tint", "ior", "emission", "roughness", "metallic", "transparency"):
        assert np.array_equal(ma[f_], mb[f_]), f_
    for f_ in ("base_color_tex_id", "emission_tex_id"):
        for a, b in zip(ma[f_], mb[f_]):
            assert (a == 0xFFFFFFFF) == (b == 0xFFFFFFFF)
            if a != 0xFFFFFFFF:
                assert np.array_equal(sc.textures[a], np.asarray(texs[b], dtype=np.uint8))
