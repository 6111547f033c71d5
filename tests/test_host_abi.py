"""Host-side checks that need no GPU: the C-ABI library loads and exports every declared symbol,
the scene presets equal the scenes the reference constructs, and save() bytes match."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import path_tracer_amd as pt
from _oracle import bodies_to_list, camera_to_list
from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "ptgpu.h")) as f:
        text = f.read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(pt_[a-z_0-9]+)\s*\(", text, flags=re.M))
    assert len(declared) >= 18
    L = pt.lib()
    for name in declared:
        assert hasattr(L, name), f"libptgpu.so does not export {name}"
    # the node library (several GPUs from one process): checked from its symbol table, without loading
    # it here (it links RCCL, and this process may already hold torch's own copy of that library)
    with open(os.path.join(ROOT, "include", "ptgpu_node.h")) as f:
        node_declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(pt_node_[a-z_0-9]+)\s*\(", f.read(), flags=re.M))
    assert len(node_declared) == 5
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "path_tracer_amd", "libptnode.so")],
                        capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if " T " in line}
    assert node_declared <= exported, node_declared - exported


@pytest.mark.parametrize("name", ["main32", "nee15"])
def test_presets_equal_reference_scenes(name):
    with open(os.path.join(GOLDEN, f"scene_{name}.json")) as f:
        g = json.load(f)
    d = pt.preset(name, 16, 16)
    assert d.body_count == len(g["bodies"])
    assert camera_to_list(d.camera) == g["camera"]
    stored = (pt.Body * d.body_count)()
    for i in range(d.body_count):
        stored[i] = d.bodies[i]
        assert pt.lib().pt_body_construct(C.byref(stored[i])) == 0
    got = bodies_to_list(stored, d.body_count)
    for i, (a, b) in enumerate(zip(got, g["bodies"])):
        assert a == b, (i, a, b)


def test_raytrace3_preset_equals_reference_scene():
    """The preset against the bodies raytrace3.cpp's main() builds (rectangle corners, and the normal
    its Rectangle constructor stores), tests/golden/raytrace3.npz."""
    import _oracle

    with open(os.path.join(GOLDEN, "raytrace3.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, "raytrace3.npz"))
    want, n = _oracle.bodies_from_bytes(z["scene_bodies"], meta["body_record"])
    d = pt.preset("raytrace3", 16, 16)
    assert d.body_count == n and d.estimator == pt.EST_PT and d.flags == pt.F_RAYTRACE3
    for i in range(n):
        b = pt.Body.from_buffer_copy(d.bodies[i])
        assert pt.lib().pt_body_construct(C.byref(b)) == 0
        w = want[i]
        assert (b.shape, b.material) == (w.shape, w.material), i
        for f in ("center", "normal", "color", "emission", "corners"):
            assert [float(v).hex() for v in getattr(b, f)] == [float(v).hex() for v in getattr(w, f)], (i, f)
        assert (b.radius, b.ior, b.reflection) == (w.radius, w.ior, w.reflection), i


def test_save_ppm_bytes(tmp_path):
    with open(os.path.join(GOLDEN, "save.json")) as f:
        g = json.load(f)
    buf = np.load(os.path.join(GOLDEN, "save_buffer.npy"))
    path = str(tmp_path / "o.ppm")
    pt.save_ppm(buf, g["samples"], path)
    with open(path) as f:
        assert f.read() == g["ppm"]["0"]["text"]


def test_cli_usage_line():
    # main.cpp:463-466: with fewer than four arguments print the usage line and return 0
    for prog in ("main", "next_event"):
        exe = os.path.join(ROOT, "path_tracer_amd", "bin", prog)
        r = subprocess.run([exe, "640", "480"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0
        assert r.stdout == f"{exe} <width> <height> <iterations> <thread count>\n"


def test_synthetic_scene_is_deterministic():
    a = pt.preset("synthetic10k", 8, 8)
    b = pt.preset("synthetic10k", 8, 8)
    assert a.body_count == 10000
    assert bodies_to_list(a.bodies, 10000)[9999] == bodies_to_list(b.bodies, 10000)[9999]
    assert pt.preset("synthetic10k", 8, 8).bodies[14].emission[0] == 1.9


def test_no_cpu_fallback_when_gpu_missing():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError):
        pt.Renderer(pt.preset("nee15", 8, 8))


def test_save_ppm_edge_values(tmp_path):
    """NaN, infinities, negatives and the int-truncation boundaries, as the reference's save() prints them."""
    with open(os.path.join(GOLDEN, "save_edge.json")) as f:
        g = json.load(f)
    buf = np.array([float.fromhex(h) for h in g["values_hex"]]).reshape(g["H"], g["W"], 3)
    path = str(tmp_path / "edge.ppm")
    pt.save_ppm(buf, g["samples"], path)
    with open(path, "rb") as f:
        data = f.read().decode()
    assert data == g["ppm"]["0"]["text"] == g["ppm"]["1"]["text"]


@pytest.mark.skipif(not os.path.exists("/root/reference/main.cpp"), reason="needs the reference checkout")
def test_reference_side_binding_stub_compiles():
    """The replacement body of worker() (main.cpp:436-458) that calls the C ABI compiles against the
    reference's own classes and include/ptgpu.h (oracle/binding_stub_check.cpp)."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "binding-check"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
