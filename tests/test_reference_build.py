"""The reference's own code, compiled here without Boost and without any stand-in for it
(oracle/Makefile: the programs minus the lines that mention `boost` or `threads`), reproduces the
committed golden vectors.  Runs only where that build exists (the build container); the golden files
themselves are what travels."""
import json
import os

import numpy as np
import pytest

import _oracle
from _oracle import MODE_PIXEL
from conftest import GOLDEN, ROOT, same_bits

pytestmark = pytest.mark.skipif(not (_oracle.Ref.available() and _oracle.Ref.available(hooked=True)),
                                reason="oracle/_ref is built only where the reference checkout is present")


def test_compiled_sources_hold_no_boost():
    src = os.path.join(ROOT, "oracle", "_ref", "noboost_src")
    if not os.path.isdir(src):
        pytest.skip("the filtered sources stay in the build container")
    names = sorted(os.listdir(src))
    assert len(names) == 9  # two parts of four programs, and the head of light_tracing.cpp
    for n in names:
        text = open(os.path.join(src, n)).read()
        assert "boost" not in text and "threads" not in text and "kdtree" not in text, n
    lt = os.path.join("/root/reference", "light_tracing.cpp")
    if os.path.exists(lt):
        # the head is the file's own lines up to `struct Scene`, minus the include lines and the mutex
        whole = open(lt).read().splitlines()
        cut = next(i for i, ln in enumerate(whole) if ln.startswith("struct Scene"))
        want = [ln for ln in whole[:cut] if "boost" not in ln and "kdtree" not in ln]
        assert open(os.path.join(src, "light_tracing.head.inc")).read().splitlines() == want
        assert cut - len(want) == 6
    # what was left out is the thread glue and nothing else: 10-12 lines per program
    for prog in ("main", "next_event", "raytrace3", "bidirectional"):
        ref = os.path.join("/root/reference", prog + ".cpp")
        if not os.path.exists(ref):
            pytest.skip("no reference checkout")
        kept = sum(len(open(os.path.join(src, f"{prog}.{p}.inc")).read().splitlines()) for p in ("A", "B"))
        dropped = [ln for ln in open(ref).read().splitlines() if "boost" in ln or "threads" in ln]
        assert kept + len(dropped) == len(open(ref).read().splitlines())
        assert all(("boost" in ln) or ("threads" in ln) for ln in dropped) and len(dropped) <= 20
    harness = open(os.path.join(ROOT, "oracle", "ref_harness.cpp")).read()
    assert "shim" not in harness.replace("no stand-in", "")


def test_reference_build_reproduces_golden_frames():
    with open(os.path.join(GOLDEN, "frames.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    refh = _oracle.Ref(hooked=True)
    prog = {"main32": 0, "nee15": 1}
    checked = 0
    for m in meta:
        if m["scene"] not in prog:
            continue
        frame, draws = refh.render(prog[m["scene"]], m["W"], m["H"], m["first"], m["n_it"], m["mode"], seed=m["seed"],
                                   script=m.get("script"))
        assert same_bits(frame, z[m["key"]]), m["key"]
        assert draws == m["draws"]
        checked += 1
    assert checked >= 2
    with open(os.path.join(GOLDEN, "bidir.json")) as f:
        bm = json.load(f)
    bz = np.load(os.path.join(GOLDEN, "bidir.npz"))
    c = bm["frames"][0]
    frame, draws = refh.render(3, c["W"], c["H"], c["first"], c["n_it"], MODE_PIXEL, seed=c["seed"])
    assert same_bits(frame, bz["frame_" + c["key"]]) and draws == c["draws"]
