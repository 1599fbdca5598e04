"""The Haskell binding (integration/Rendering/GpuRender.hs) cannot be compiled here (no GHC in the image), so what can be
checked mechanically is: every struct size and field offset it pokes at equals sizeof / offsetof of include/rtcore.h
(printed by tools/abi_layout.c, built with gcc); every C symbol it imports is exported by librtcore.so with the
arity the header declares; the ctypes mirror (raytracer_amd/_ffi.py) has the same layout; and the two patches apply to
the reference's files when the reference is present."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

from raytracer_amd import _ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "integration", "Rendering", "GpuRender.hs")


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("abi") / "abi_layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "abi_layout.c"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    return {k: int(v) for k, v in (line.split() for line in out.splitlines())}


def hs_layout():
    src = open(HS).read()
    block = src[src.index("-- BEGIN-ABI-LAYOUT"):src.index("-- END-ABI-LAYOUT")]
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"^((?:off|sizeof)_\w+|rt_abi_version_expected) = (\d+)\s*$", block, re.M)}


def test_haskell_offsets_equal_the_c_header(c_layout):
    hs = hs_layout()
    assert len(hs) >= 60
    assert hs.pop("rt_abi_version_expected") == c_layout["RT_ABI_VERSION"] == F.core().rt_abi_version()
    for name, value in hs.items():
        assert name in c_layout, f"{name}: not a field of include/rtcore.h"
        assert value == c_layout[name], f"{name}: GpuRender.hs says {value}, the C compiler says {c_layout[name]}"
    # every field of the structs the binding fills is covered (rt_stats is only cleared: its size is enough)
    for name in c_layout:
        if name.startswith(("off_", "sizeof_")) and not name.startswith("off_rt_stats"):
            assert name in hs, f"{name} is missing from GpuRender.hs"


def test_every_offset_constant_is_used_by_a_poke():
    src = open(HS).read()
    body = src[src.index("-- END-ABI-LAYOUT"):]
    for name in hs_layout():
        if name.startswith("off_") or name.startswith("sizeof_"):
            assert re.search(r"\b%s\b" % name, body), f"{name} is declared but never used"


def test_imported_symbols_exist_with_the_declared_arity():
    src = open(HS).read()
    imports = re.findall(r'foreign import ccall safe "(\w+)"\s+\w+\s*::\s*([^\n]+)', src)
    assert {s for s, _ in imports} == {"rt_init_devices", "rt_scene_create", "rt_render", "rt_scene_destroy", "rt_last_error", "rt_abi_version"}
    lib = F.core()
    for sym, sig in imports:
        assert sym in F.RTCORE_SYMBOLS and hasattr(lib, sym)
        n_args = len([p for p in sig.split("->")]) - 1           # the last element is the IO result
        assert n_args == len(F.RTCORE_SYMBOLS[sym][1]), f"{sym}: {n_args} Haskell arguments, {len(F.RTCORE_SYMBOLS[sym][1])} in the header"


def test_ctypes_mirror_has_the_same_layout(c_layout):
    for cls in (F.rt_material, F.rt_light, F.rt_scene_desc, F.rt_camera, F.rt_params, F.rt_output, F.rt_stats):
        assert C.sizeof(cls) == c_layout[f"sizeof_{cls.__name__}"], cls.__name__
        for fname, _ in cls._fields_:
            key = f"off_{cls.__name__}_{fname}"
            if key in c_layout:
                assert getattr(cls, fname).offset == c_layout[key], key


@pytest.mark.skipif(not os.path.isdir("/root/reference/app") or shutil.which("patch") is None, reason="needs the reference checkout and patch(1)")
def test_patches_apply_to_the_reference(tmp_path):
    os.makedirs(tmp_path / "app" / "Rendering")
    shutil.copy("/root/reference/app/Rendering/ImageGenerator.hs", tmp_path / "app" / "Rendering" / "ImageGenerator.hs")
    shutil.copy("/root/reference/raytracer.cabal", tmp_path / "raytracer.cabal")
    for p in ("ImageGenerator.patch", "raytracer.cabal.patch"):
        subprocess.check_call(["patch", "-p1", "--forward", "-s", "-i", os.path.join(ROOT, "integration", p)], cwd=tmp_path)
    patched = open(tmp_path / "app" / "Rendering" / "ImageGenerator.hs").read()
    assert "renderOnGpu config bvh materialMap skySphere cameraObj sceneLights" in patched and "forkIO $\n      worker" not in patched
    assert "Rendering.GpuRender" in open(tmp_path / "raytracer.cabal").read()
