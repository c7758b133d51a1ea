"""The SEAL 2.3 cross-check harness (oracle/seal_crosscheck.cpp: a program written against the seal/seal.h API only -- the
reference's own seam, homo/fhe_image.h:13) prints the values of tests/golden/seal_crosscheck.json (made by the CPU oracle,
tests/golden/make_seal_crosscheck.py) on BOTH backends of this repository's facade: the oracle-backed C ABI (CPU, here) and
libfhe_hip.so (MI355X, -m gpu).  The same program builds against a real SEAL 2.3 (`make -C oracle seal23 SEAL_ROOT=...`); its
output is checked with `python tests/test_seal_crosscheck.py <file>` -- the day such a library exists, this is the test that
lifts the parity pin from plaintext level to ciphertext bits (pin 5).

The binaries include the reference's headers (Cubic, Linear, homomorphic_sin from homo/fhe_resize.h / fhe_decode.h), so they are
built where /root/reference exists (oracle/Makefile, into oracle/_ref/) and travel to the GPU box prebuilt.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "seal_crosscheck.json")


def parse(text):
    """{name: (sha256, [16 samples])} + the sin_value line's fields"""
    lines, sin = {}, None
    for ln in text.splitlines():
        f = ln.split()
        if not f or f[0].startswith("#"):
            continue
        if f[0] == "sin_value":
            sin = {"value": f[1], "size": int(f[f.index("size") + 1]), "budget": f[f.index("budget") + 1]}
        elif len(f) == 18 and len(f[1]) == 64:
            lines[f[0]] = (f[1], f[2:])
    return lines, sin


def compare(text):
    """list of human-readable mismatches (empty = the run reproduces the committed values)"""
    want = json.load(open(GOLDEN))["lines"]
    got, sin = parse(text)
    bad = []
    for name, w in want.items():
        if name == "sin_value":
            if sin is None:
                bad.append("sin_value: line missing")
            elif sin["value"] != w["value"] or sin["size"] != w["size"] or sin["budget"] != "positive":
                bad.append("sin_value: got %r, want %r" % (sin, w))
            continue
        if name not in got:
            bad.append("%s: line missing" % name)
        elif got[name][0] != w["sha256"] or got[name][1] != w["sample"]:
            first = next((j for j, (a, b) in enumerate(zip(got[name][1], w["sample"])) if a != b), None)
            bad.append("%s: sha256 %s.. != %s.. (first differing sample: %s)" % (name, got[name][0][:12], w["sha256"][:12], first))
    extra = sorted(set(got) - set(want))
    if extra:
        bad.append("lines without an expected value: %s" % extra)
    return bad


def _run(exe):
    path = os.path.join(ROOT, "oracle", "_ref", exe)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s not built (needs /root/reference: make -C oracle ref)" % exe)
    r = subprocess.run([path], capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return r.stdout


def test_expected_values_cover_every_group_of_lines():
    want = json.load(open(GOLDEN))["lines"]
    for name in ["input_A", "add", "sub", "negate", "add32", "sub23", "multiply22", "multiply32", "multiply43", "square2", "square3",
                 "relin16_3", "relin16_4", "relin30_3", "relin30_4", "cubic", "linear", "sin_plain", "sin_value"]:
        assert name in want, name
    assert sum(k.startswith("encode[") for k in want) == 24 and sum(k.startswith("multiply_plain[") for k in want) == 23      # encode(0.0) is not a multiplier
    assert sum(k.startswith("add_plain[") for k in want) == 24 and sum(k.startswith("sub_plain[") for k in want) == 24


def test_harness_uses_nothing_but_the_seal_api():
    """the property that makes it portable to a real SEAL: no facade hook macro, no C-ABI symbol, no oracle symbol in its source"""
    src = open(os.path.join(ROOT, "oracle", "seal_crosscheck.cpp")).read()
    code = "\n".join(ln.split("//")[0] for ln in src.splitlines() if not ln.startswith("#include"))     # the reference's headers are called fhe_*.h
    for forbidden in ("FHE_FACADE", "fhe_", "fo_", "detail::", "hip::", ".ptr()", ".buffer()", ".shape(", "node()"):
        assert forbidden not in code, forbidden
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    rule = mk[mk.index("$(REFDIR)/seal_crosscheck:"):mk.index("seal23:")]
    assert "FHE_FACADE_TEST_HOOKS" not in rule and "ref_hook" not in rule


def test_crosscheck_on_the_oracle_backed_abi():
    bad = compare(_run("seal_crosscheck_cpu"))
    assert not bad, "\n".join(bad)


@pytest.mark.gpu
def test_crosscheck_on_libfhe_hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    bad = compare(_run("seal_crosscheck"))
    assert not bad, "\n".join(bad)


if __name__ == "__main__":                    # python tests/test_seal_crosscheck.py <output of a seal_crosscheck build>
    problems = compare(open(sys.argv[1]).read())
    print("\n".join(problems) if problems else "every line equals tests/golden/seal_crosscheck.json")
    sys.exit(1 if problems else 0)
