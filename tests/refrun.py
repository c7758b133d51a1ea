"""Helpers for the tests that run the REFERENCE's own code (oracle/_ref/*, built by oracle/Makefile
target `ref` from /root/reference where it exists) through the SEAL facade.  Test infrastructure."""
import math
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
HDR = struct.Struct("<8sIIII")                      # seal/seal.h save_words: magic, polys, k, n, reserved


def ref_bin(name, gpu):
    """oracle/_ref/<name>[_cpu]; FHE_REF_VARIANT=asan (tests/test_sanitizers.py) selects the AddressSanitizer / UBSan build
    oracle/_san/<name>_cpu_asan of the CPU variants"""
    variant = os.environ.get("FHE_REF_VARIANT")
    if variant and not gpu:
        path = os.path.join(ROOT, "oracle", "_san", name + "_cpu_" + variant)
    else:
        path = os.path.join(REF_DIR, name + ("" if gpu else "_cpu"))
    return path if os.path.exists(path) else None


def write_record(f, ct):
    ct = np.ascontiguousarray(ct, dtype=np.uint64)
    f.write(HDR.pack(b"FHEHIP1\0", ct.shape[0], ct.shape[1], ct.shape[2], 0))
    f.write(ct.tobytes())


def read_records(path, size, k, n, count):
    rec = HDR.size + size * k * n * 8
    raw = open(path, "rb").read()
    assert len(raw) == rec * count, (len(raw), rec * count)
    out = np.empty((count, size, k, n), dtype=np.uint64)
    for i in range(count):
        magic, s, kk, nn, _ = HDR.unpack(raw[i * rec:i * rec + HDR.size])
        assert magic[:7] == b"FHEHIP1" and (s, kk, nn) == (size, k, n)
        out[i] = np.frombuffer(raw[i * rec + HDR.size:(i + 1) * rec], dtype=np.uint64).reshape(size, k, n)
    return out


def sample_origins(W, H, w, h):
    """(int(u), int(v)) per output pixel, float32 arithmetic as in homo/fhe_resize.h:351,382 and
    :226,229 / :258,264 (restated here independently of the product's circuits.resize_sample_plan)."""
    f32 = np.float32
    out = []
    for y in range(h):
        v = f32(f32(y) / f32(h - 1) * f32(H)) - f32(0.5)
        for x in range(w):
            u = f32(f32(x) / f32(w - 1) * f32(W)) - f32(0.5)
            out.append((int(u), int(v)))
    return out


def clamp(v, lo, hi):
    return lo if v < lo else hi if v > hi else v


def oracle_sample(orc, pix, W, H, xi, yi, ch, xf, yf, bicubic):
    """SampleBicubic / SampleLinear (homo/fhe_resize.h:222-305) for one output pixel and channel from
    the oracle's Cubic / Linear.  pix: [W*H, 3, 2, k, n]."""
    def P(dx, dy):
        return pix[clamp(yi + dy, 0, H - 1) * W + clamp(xi + dx, 0, W - 1), ch]
    if bicubic:
        cols = [orc.cubic(P(-1, j), P(0, j), P(1, j), P(2, j), xf) for j in (-1, 0, 1, 2)]
        return orc.cubic(cols[0], cols[1], cols[2], cols[3], yf)
    return orc.linear(orc.linear(P(0, 0), P(1, 0), xf), orc.linear(P(0, 1), P(1, 1), xf), yf)


def run_server_resize(workdir, orc, pix, W, H, w, h, bicubic, fracs, gpu, n_arg, env_extra=None):
    """homo/server_resize.cpp (unchanged) on a ciphertext stream; `fracs` [w*h*2, 2, k, n] are handed to the
    circuit's encrypt calls in order (xfract, yfract per output pixel).  Returns [w*h*3, size, k, n]."""
    exe = ref_bin("ref_server_resize", gpu)
    os.makedirs(os.path.join(workdir, "keys"), exist_ok=True)
    os.makedirs(os.path.join(workdir, "image"), exist_ok=True)
    with open(os.path.join(workdir, "keys", "params.txt"), "w") as f:
        f.write("%d %d 3 %d\n" % (W, H, orc.t))
    sk, pk = orc.keygen(3)
    with open(os.path.join(workdir, "keys", "pubkey.txt"), "wb") as f:
        write_record(f, pk)
    with open(os.path.join(workdir, "keys", "seckey.txt"), "wb") as f:
        write_record(f, sk[None])
    with open(os.path.join(workdir, "image", "in.ct"), "wb") as f:
        for p in range(W * H):
            for c in range(3):
                write_record(f, pix[p, c])
    hook = os.path.join(workdir, "hook.bin")
    np.ascontiguousarray(fracs, dtype=np.uint64).tofile(hook)
    env = dict(os.environ, FHE_ENCRYPT_HOOK_FILE=hook)
    env.update(env_extra or {})
    argv = [exe, "--width", str(w), "--height", str(h), "--cmod", str(n_arg), "--pmod", str(orc.t), "-f", "image/in.ct", "-o", "image/out.ct"]
    if bicubic:
        argv.append("--bicubic")
    r = subprocess.run(argv, cwd=workdir, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return read_records(os.path.join(workdir, "image", "out.ct"), 6 if bicubic else 4, orc.k, orc.n, w * h * 3)


def run_decode_circuit(workdir, orc, mode, inputs, hook_cts, gpu, n_arg, extra=(), env_extra=None, sizes=(11,)):
    """oracle/ref_decode_circuit_main.cpp: homo/fhe_decode.h's homomorphic_sin / homomorphic_cos /
    approximated_step unchanged.  Returns a list of arrays [size, k, n]."""
    exe = ref_bin("ref_decode_circuit", gpu)
    fin, fout, hook = (os.path.join(workdir, x) for x in ("in.bin", "out.bin", "hook.bin"))
    np.ascontiguousarray(inputs, dtype=np.uint64).tofile(fin)
    np.ascontiguousarray(hook_cts, dtype=np.uint64).tofile(hook)
    env = dict(os.environ, FHE_ENCRYPT_HOOK_FILE=hook)
    env.update(env_extra or {})
    r = subprocess.run([exe, str(n_arg), str(orc.t), mode, fin, fout] + [str(x) for x in extra], env=env, capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(fout, dtype=np.uint64)
    out, pos = [], 0
    for s in sizes:
        out.append(raw[pos:pos + s * orc.k * orc.n].reshape(s, orc.k, orc.n))
        pos += s * orc.k * orc.n
    assert pos == raw.size
    return out


def run_server_decode(workdir, orc, runs, pairs, width, height, hook_cts, gpu, n_arg, order, degree, delta, env_extra=None):
    """homo/server_decode.cpp's main (unchanged; its approximated_step call sent to the homomorphic overload by
    oracle/ref_server_decode_main.cpp) on a ciphertext stream.  runs: [sum(pairs), 2, 2, k, n] (elem, count per
    run, channel after channel); hook_cts: the server-side Enc(0)s in call order.  Returns the raw output file."""
    exe = ref_bin("ref_server_decode", gpu)
    os.makedirs(os.path.join(workdir, "keys"), exist_ok=True)
    os.makedirs(os.path.join(workdir, "image"), exist_ok=True)
    with open(os.path.join(workdir, "keys", "params.txt"), "w") as f:
        f.write("%d %d %d %d %d\n" % (width, height, pairs[0], pairs[1], pairs[2]))
    sk, pk = orc.keygen(3)
    with open(os.path.join(workdir, "keys", "pubkey.txt"), "wb") as f:
        write_record(f, pk)
    with open(os.path.join(workdir, "keys", "seckey.txt"), "wb") as f:
        write_record(f, sk[None])
    with open(os.path.join(workdir, "image", "in.ct"), "wb") as f:
        for r in range(runs.shape[0]):
            write_record(f, runs[r, 0])
            write_record(f, runs[r, 1])
    hook = os.path.join(workdir, "hook.bin")
    np.ascontiguousarray(hook_cts, dtype=np.uint64).tofile(hook)
    env = dict(os.environ, FHE_ENCRYPT_HOOK_FILE=hook)
    env.update(env_extra or {})
    argv = [exe, "--cmod", str(n_arg), "--pmod", str(orc.t), "--order", str(order), "--degree", str(degree), "--delta", repr(float(delta)),
            "-f", "image/in.ct", "-o", "image/out.ct"]
    r = subprocess.run(argv, cwd=workdir, env=env, capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return open(os.path.join(workdir, "image", "out.ct"), "rb").read()


def parse_stream(raw, k, n):
    """a ciphertext stream of records of any sizes -> list of arrays [size, k, n]"""
    out, pos = [], 0
    while pos < len(raw):
        magic, s, kk, nn, _ = HDR.unpack(raw[pos:pos + HDR.size])
        assert magic[:7] == b"FHEHIP1" and (kk, nn) == (k, n)
        pos += HDR.size
        out.append(np.frombuffer(raw[pos:pos + s * k * n * 8], dtype=np.uint64).reshape(s, k, n))
        pos += s * k * n * 8
    return out


def oracle_server_decode(orc, om, runs, pairs, width, height, hook_cts, order, degree, delta):
    """The driver loop of homo/server_decode.cpp:120-143 composed from the ORACLE's single operations (restated here,
    independently of the product's server.py / csrc/circuits.hip).  Returns res[channel][position]."""
    npos = width * height
    it = iter(range(hook_cts.shape[0]))
    nxt = lambda: hook_cts[next(it)]
    res, r = [], 0
    for ch in range(3):
        index = nxt()                                                  # :121
        channel = [nxt() for _ in range(npos)]                        # :124-128
        for _ in range(pairs[ch]):
            elem, count = runs[r, 0], runs[r, 1]                      # :131-132
            r += 1
            bank = {}
            for i in range(npos):                                      # the Enc(0)s of :231-232 in call order
                for j in range(1, degree + 1):
                    bank[(i, j, "sin")] = nxt()
                    bank[(i, j, "cos")] = nxt()
            run = om.oracle_approximated_step(orc, elem, index, count, order, degree, delta, width, height, lambda i, j, w: bank[(i, j, w)])   # :133
            channel = [orc.add(channel[k], run[k]) for k in range(npos)]   # :134-136
            index = orc.add(index, count)                             # :137
        res.append(channel)
    return res
