"""CPU-only checks of everything around the kernels: the pinned tables, the C
ABI surface, the host helpers against the oracle, error codes that need no GPU,
and the host-side scene mirror."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import raytracing_amd
from raytracing_amd import (Cube, Dielectric, HittableList, Lambertian, Metal, Plane, Scene, Sphere,
                            Triangle, _lib, image_height, load_tables, write_ppm)
from raytracing_amd.renderer import TABLE_FILES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


# ------------------------------------------------------------------ tables
def test_table_files_match_pinned_hashes():
    """tests/golden/tables.sha256 was written by tests/golden/make_tables.py from
    the reference's src/denoising/mod.rs:55,59,61 after checking token identity
    with its data/ text twins"""
    with open(os.path.join(ROOT, "tests", "golden", "tables.sha256")) as f:
        want = dict(reversed(line.split()) for line in f if line.strip())
    data = os.path.join(ROOT, "raytracing_amd", "data")
    for fname, _ in TABLE_FILES.values():
        with open(os.path.join(data, fname), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == want[fname], fname


def test_table_contents(tables):
    d = tables["disc"].reshape(64, 2)
    assert np.array_equal(d[:3], np.array([[0.478712, 0.875764], [-0.337956, -0.793959],
                                           [-0.955259, -0.028164]], dtype=f32))
    rad = np.hypot(d[:, 0].astype(np.float64), d[:, 1].astype(np.float64))
    assert 0.08 < rad.min() < 0.09 and 0.99 < rad.max() < 1.0          # a radius-1 disc, not a +-0.5 box
    s = tables["sphere"].reshape(256, 3).astype(np.float64)
    assert np.allclose(np.linalg.norm(s, axis=1), 1.0, atol=1e-6)       # ON the unit sphere
    n = tables["screen"]
    k = np.rint(n.astype(np.float64) * 255).astype(int)
    assert np.array_equal(n, (k / 255.0).astype(f32))                   # every value is k/255
    assert set(k.tolist()) == set(range(256))
    assert n[0] == f32(0.3686274509803922) and n[-1] == f32(0.18823529411764706)


def test_parse_floats_rounds_like_a_literal():
    lib = _lib.load()
    text = b"[(0.1, -2.5e-3), 16777217.0, 0.30000001192092896, 1e-46, 3.4028236e38]"
    out = np.zeros(8, dtype=f32)
    n = lib.rt_parse_floats(text, len(text), out.ctypes.data_as(C.POINTER(C.c_float)), 8)
    assert n == 6
    assert out[0] == f32(0.1) and out[1] == f32(-0.0025)
    assert out[2] == f32(16777216.0)            # ties-to-even directly from decimal
    assert out[3] == f32(0.3) and out[4] == 0.0 and np.isinf(out[5])
    assert lib.rt_parse_floats(text, len(text), None, 0) == 6      # counting only
    assert lib.rt_parse_floats(b"", 0, None, 0) == 0


# ----------------------------------------------------------------- C ABI
def _declared_functions():
    with open(os.path.join(ROOT, "include", "rt_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n + " is declared in rt_hip.h but missing from librt_hip.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_struct_layouts_match_header(tmp_path):
    """sizes and field offsets of the ctypes mirrors equal what a C compiler
    derives from include/rt_hip.h"""
    import subprocess
    structs = {"rt_scene": _lib.rt_scene, "rt_camera": _lib.rt_camera, "rt_params": _lib.rt_params}
    lines = []
    for name, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for field, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, field, name, field))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rt_hip.h"\nint main(void){\n'
                   + "\n".join(lines) + "\nreturn 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == C.sizeof(cls), name
        for field, _ in cls._fields_:
            assert int(got[name + "." + field]) == getattr(cls, field).offset, name + "." + field
    assert C.sizeof(_lib.rt_camera) == 19 * 4 and C.sizeof(_lib.rt_params) == 12 * 4


def test_errors_without_gpu_are_codes_not_crashes():
    lib = _lib.load()
    assert lib.rt_create(0, None) == _lib.RT_E_BADARG
    ctx = C.c_void_p()
    rc = lib.rt_create(10 ** 6, C.byref(ctx))
    assert rc in (_lib.RT_E_BADARG, _lib.RT_E_HIP) and not ctx.value
    assert lib.rt_last_error(None)
    for fn in ("rt_sync", "rt_pack"):
        assert getattr(lib, fn)(None) == _lib.RT_E_BADARG
    lib.rt_destroy(None)
    z3 = (C.c_float * 3)(0, 0, 0)
    assert lib.rt_camera_new(27.0, 1.7, 0.0, 1.0, z3, z3, z3, None) == _lib.RT_E_BADARG
    assert lib.rt_write_ppm(None, None, 0, 0) == _lib.RT_E_BADARG
    px = np.zeros(4, dtype=np.uint32)
    assert lib.rt_write_ppm(b"/nonexistent-dir/x.ppm", px.ctypes.data_as(C.POINTER(C.c_uint32)), 2, 2) == _lib.RT_E_IO


# ------------------------------------------------------- helpers vs the oracle
def test_camera_new_is_bitwise_the_oracle(oracle):
    rng = np.random.default_rng(7)
    for _ in range(200):
        o = rng.uniform(-5, 5, 3); la = rng.uniform(-5, 5, 3)
        up = (0.0, 1.0, 0.0) if rng.random() < 0.5 else tuple(rng.uniform(-1, 1, 3))
        fov = float(rng.uniform(5, 120)); asp = float(rng.uniform(0.5, 2.5))
        ap = float(rng.uniform(0, 0.5)); fd = float(rng.uniform(0.1, 20))
        mine = raytracing_amd.Camera.new(fov, asp, ap, fd, o, la, up).struct
        ref = oracle.camera_new(fov, asp, ap, fd, o, la, up)
        assert bytes(mine) == bytes(ref)
    for counter in (0, 1, 7, 99):
        for ap in (0.0, 0.1):
            Scene.COUNTER = counter
            try:
                assert bytes(Scene.get_camera(aperture=ap).struct) == bytes(oracle.scene_camera(counter, ap))
            finally:
                Scene.COUNTER = 0


def test_image_height_matches_oracle(oracle):
    a = float(f32(16.0) / f32(9.0))
    for w in list(range(2, 700, 13)) + [1080, 1920, 3840, 7680]:
        assert image_height(w) == oracle.image_height(w, a)
    assert [image_height(w) for w in (600, 1080, 1920, 3840)] == [337, 607, 1080, 2160]
    lib = _lib.load()
    assert lib.rt_image_height(100, float("nan")) == 0 and lib.rt_image_height(100, -1.0) == 0


def test_move_cam_wraps_at_100():
    """Scene::move_cam, src/scene/mod.rs:78-83"""
    Scene.COUNTER = 98
    try:
        Scene.move_cam(); assert Scene.COUNTER == 99
        Scene.move_cam(); assert Scene.COUNTER == 0
    finally:
        Scene.COUNTER = 0


def test_ppm_writer(tmp_path):
    """PPM::write_file src/image/mod.rs:61-67: header "P6 {w} {h} 255\\n" then RGB bytes"""
    px = np.array([[0x010203, 0xFFFEFD], [0x000000, 0x80FF00], [0x123456, 0x00ABCDEF]], dtype=np.uint32)
    path = tmp_path / "a.ppm"
    write_ppm(str(path), px)
    blob = path.read_bytes()
    assert blob.startswith(b"P6 2 3 255\n")
    assert blob[len(b"P6 2 3 255\n"):] == bytes([1, 2, 3, 255, 254, 253, 0, 0, 0, 128, 255, 0,
                                                 0x12, 0x34, 0x56, 0xCD, 0xEF, 0xAB][:15] + [0xAB, 0xCD, 0xEF])


# --------------------------------------------------------------- scene mirror
def test_png_writer(tmp_path):
    """signature, IHDR, one zlib stream of stored blocks, CRCs; pixels round-trip"""
    import struct
    import zlib
    from raytracing_amd import write_png
    rng = np.random.default_rng(5)
    img = (rng.integers(0, 1 << 24, size=(211, 173))).astype(np.uint32)      # > 65535 raw bytes: several blocks
    path = tmp_path / "x.png"
    write_png(str(path), img)
    raw = path.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + data) & 0xffffffff
        chunks.append((typ, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (173, 211, 8, 2, 0, 0, 0)
    pix = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(211, 173 * 3 + 1)
    assert (pix[:, 0] == 0).all()
    rgb = pix[:, 1:].reshape(211, 173, 3).astype(np.uint32)
    assert np.array_equal((rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2], img)


def test_default_scene_literals():
    """Scene::get_scene src/scene/mod.rs:15-57: order, kinds, materials"""
    flat = Scene.get_scene().flatten()
    assert flat.kinds_in_list_order() == ["sphere"] * 6 + ["cube", "plane"]
    a = flat.arrays
    assert np.array_equal(a["sph_id"], np.arange(6, dtype=np.uint32))
    assert np.array_equal(a["sph_r"], np.array([0.5, 0.25, 0.5, 0.1, 0.14, -0.13], dtype=f32))
    assert np.array_equal(a["sph_cx"], np.array([0.80, -0.10, -0.80, 0.25, -0.25, -0.25], dtype=f32))
    assert a["cb_id"][0] == 6 and a["pl_id"][0] == 7
    kinds = a["mat_kind"].tolist()
    assert kinds == [_lib.RT_METAL, _lib.RT_LAMBERTIAN, _lib.RT_METAL, _lib.RT_DIELECTRIC,
                     _lib.RT_DIELECTRIC, _lib.RT_DIELECTRIC, _lib.RT_LAMBERTIAN, _lib.RT_LAMBERTIAN]
    assert np.array_equal(a["mat_param"], np.array([0.25, 0, 0, 1.5, 1.5, 1.5, 0, 0], dtype=f32))
    assert (a["mat_r"][4], a["mat_g"][4], a["mat_b"][4]) == (f32(0.95), f32(0.95), f32(1.0))


def test_scene_floats_round_like_rust_literals():
    """every float the mirror hands over equals strtof of its decimal text"""
    lib = _lib.load()
    lits = ["0.80", "0.5", "-1.0", "0.8", "0.6", "0.2", "0.25", "-0.10", "0.7", "0.3", "-0.80", "0.75",
            "-0.5", "0.1", "1.5", "-0.25", "-0.42", "0.14", "0.95", "-0.13", "0.50", "1.30", "0.60", "27.0"]
    text = " ".join(lits).encode()
    out = np.zeros(len(lits), dtype=f32)
    assert lib.rt_parse_floats(text, len(text), out.ctypes.data_as(C.POINTER(C.c_float)), len(lits)) == len(lits)
    assert np.array_equal(out, np.array([float(x) for x in lits], dtype=f32))


def test_flatten_mixed_list_keeps_insertion_order():
    s = HittableList()
    m = Lambertian((0.5, 0.5, 0.5))
    s.add(Plane((0, 1, 0), (0, 0, 0), m)).add(Sphere((0, 0, 0), 1.0, Metal((1, 1, 1), 0.1)))
    s.add(Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), m)).add(Cube((0, 0, 0), 1, 2, 3, Dielectric((1, 1, 1), 1.3)))
    s.add(Sphere((1, 1, 1), -2.0, m))
    flat = s.flatten()
    assert flat.kinds_in_list_order() == ["plane", "sphere", "triangle", "cube", "sphere"]
    ids = np.concatenate([flat.arrays[k] for k in ("sph_id", "pl_id", "cb_id", "tri_id")])
    assert sorted(ids.tolist()) == list(range(5))
    assert flat.struct.n_spheres == 2 and flat.struct.n_tris == 1 and flat.struct.n_materials == 5
    with pytest.raises(TypeError):
        HittableList().add(object()).flatten()


def test_oracle_rejects_malformed_scenes(oracle, tables):
    from raytracing_amd import make_params, flat_scene_from_arrays
    cam = oracle.scene_camera(0, 0.0)
    bad_id = flat_scene_from_arrays(sph_cx=[0, 0], sph_cy=[0, 0], sph_cz=[0, 0], sph_r=[1, 1], sph_mat=[0, 0],
                                    sph_id=[0, 0], mat_kind=[1], mat_r=[1], mat_g=[1], mat_b=[1], mat_param=[0])
    with pytest.raises(RuntimeError):
        oracle.render(bad_id, cam, tables, make_params(16, 16))
    bad_mat = flat_scene_from_arrays(sph_cx=[0], sph_cy=[0], sph_cz=[0], sph_r=[1], sph_mat=[3], sph_id=[0],
                                     mat_kind=[1], mat_r=[1], mat_g=[1], mat_b=[1], mat_param=[0])
    with pytest.raises(RuntimeError):
        oracle.render(bad_mat, cam, tables, make_params(16, 16))


def test_library_build_id_matches_the_kernel_sources():
    """rt_build_id() is the SHA-256 (first 16 hex digits) of the sources that decide what a launch executes
    (kernels, scene layout and tree builder, launch code) as the library in the tree was built from them; bench.py uses it to tell whether committed counter passes belong to this build.  A
    library that is older than the sources next to it fails here."""
    import hashlib
    from raytracing_amd import _lib
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "csrc")
    h = hashlib.sha256()
    for name in ("rt_kernels.hip", "rt_device.h", "rt_scene_build.cpp", "rt_scene_build.h", "rt_api.cpp"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    assert _lib.load().rt_build_id().decode() == h.hexdigest()[:16]


def test_mesh_workload_comes_through_the_obj_reader(tmp_path):
    """c3's mesh is data/uv_sphere.obj read by load_obj: 960 triangles whose binary32 coordinates are
    those of the procedural UV sphere the file was written from, in the same order; writing the file
    again gives the same bytes"""
    from raytracing_amd import Lambertian, Scene, Triangle, workloads
    got = workloads.mesh_scene()
    scene = Scene.get_scene()
    mat = Lambertian((0.2, 0.25, 0.6))
    for a, b, c in workloads.uv_sphere_mesh():
        scene.add(Triangle(a, b, c, mat))
    want = scene.flatten()
    assert got.counts == want.counts and got.counts["n_tris"] == 960
    for key, arr in want.arrays.items():
        assert np.array_equal(np.asarray(got.arrays[key]).view(np.uint32), np.asarray(arr).view(np.uint32)), key
    again = tmp_path / "again.obj"
    assert workloads.write_mesh_obj(str(again)) == (482, 960)
    with open(workloads.MESH_OBJ, "rb") as f, open(again, "rb") as g:
        assert f.read() == g.read()


def test_bench_joins_counter_passes_only_with_their_own_build(monkeypatch):
    """bench.py takes the per-segment instruction counts of its roofline from the committed counter
    passes under profiles/, and only when they were taken on the library it is running (rt_build_id):
    with the library of the tree every configuration has its record; with any other build id the
    fraction comes back empty together with the reason, not with the old kernel's numbers"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for key in ("c1", "c2", "c3", "c4", "c5"):
        rec, why = bench.counters_from_profile(key)
        assert rec is not None and why is None, (key, why)
        assert rec["library_build"] == bench.library_build_id()
        assert bench.hbm_traffic_from_profile(key, rec["spp_per_launch"]) > 0
    flat = Scene.get_scene().flatten()
    roof = bench.valu_roofline("c2", "render_tiles<0, 16>", 2.9e8, 4.4, 64, 220, flat)
    assert 0.2 < roof["frac"] < 0.6 and roof["hbm_gbps"] > 0 and "frac_null_because" not in roof
    monkeypatch.setattr(bench, "library_build_id", lambda: "0123456789abcdef")
    rec, why = bench.counters_from_profile("c4")
    assert rec is None and "0123456789abcdef" in why
    roof = bench.valu_roofline("c4", "render_tiles<2, 10>", 6.4e9, 390.0, 256, 180094, flat)
    assert roof["frac"] is None and roof["traffic"] is None and "library build" in roof["frac_null_because"]
    cores, allowed, quota = bench.host_cores()
    assert 1 <= cores <= min(allowed, 256) and (quota is None or cores <= quota)
