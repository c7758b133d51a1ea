"""tests/stubs/opencv2/opencv.hpp is a stand-in for cv::imread / cv::resize used ONLY by the reference's
RMSError check of the resize pipeline (homo/fhe_resize.h:35-68).  It must be bit-faithful for that pin to
mean anything, so it is validated on its own here:
  * imread == Pillow's libjpeg-turbo decode, bit for bit, on generated baseline JPEGs
    (4:4:4 / 4:2:2 / 4:2:0, odd sizes, optimised tables, restart markers, greyscale) and on the
    reference's benchmark image;
  * RMS(resize(imread(boazbarak.jpg), 17x17, INTER_LINEAR) vs a black image) prints as 113.692 -- the value
    benchmark/results.txt holds for every resize run whose noise budget was exhausted (all pixels decode
    to garbage, saturate and clamp to 0), i.e. a reference-recorded number that involves no FHE at all."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("standin") / "standin_check")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-I" + os.path.join(ROOT, "tests", "stubs"),
                           os.path.join(ROOT, "tests", "stubs", "standin_check.cpp"), "-o", exe])
    return exe


def _run(exe, path, out, *extra):
    subprocess.check_call([exe, path, out] + [str(x) for x in extra])
    raw = open(out, "rb").read()
    W, H = np.frombuffer(raw[:8], dtype=np.uint32)
    return np.frombuffer(raw[8:], dtype=np.uint8).reshape(H, W, 3)[:, :, ::-1]      # BGR -> RGB


def test_imread_equals_libjpeg_turbo(checker, tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    out = str(tmp_path / "o.raw")
    golden = os.path.join(ROOT, "tests", "golden", "boazbarak.jpg")
    assert np.array_equal(_run(checker, golden, out), np.asarray(Image.open(golden).convert("RGB")))
    for (w, h) in [(48, 48), (33, 17), (64, 40), (7, 9), (100, 75)]:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 5 + yy * 3) % 256, (xx * xx + yy * 7) % 256, (yy * yy // 3 + xx * 11) % 256], -1)
        img = (img // 2 + rng.integers(0, 128, size=img.shape)).astype(np.uint8)
        p = str(tmp_path / "t.jpg")
        for sub in (0, 1, 2):
            for q in (50, 90, 100):
                for kw in ({}, {"optimize": True}, {"restart_marker_blocks": 3}):
                    Image.fromarray(img, "RGB").save(p, quality=q, subsampling=sub, **kw)
                    assert np.array_equal(_run(checker, p, out), np.asarray(Image.open(p).convert("RGB"))), (w, h, sub, q, kw)
        Image.fromarray(img[:, :, 0], "L").save(p, quality=85)
        assert np.array_equal(_run(checker, p, out), np.asarray(Image.open(p).convert("RGB"))), (w, h, "grey")


def test_resize_reproduces_the_references_black_image_rms(checker, tmp_path):
    golden = os.path.join(ROOT, "tests", "golden", "boazbarak.jpg")
    r = _run(checker, golden, str(tmp_path / "r.raw"), 17, 17, 1).astype(np.float64)
    assert r.shape == (17, 17, 3)
    assert "%.6g" % np.sqrt((r * r).mean()) == "113.692"          # std::cout default precision, homo/fhe_resize.h:67
