#!/usr/bin/env python3
"""One measurement per PUBLISHED row of the reference (BASELINE.md section 1 = benchmark/results.txt: mean milliseconds per call of
DCT, RGBYCC, Cubic, Linear, Encryption, Decryption at n = 2048 / 4096 / 8192 / 16384, single thread, unknown CPU), on one MI355X, at the
parameter sets the reference's mains get from coeff_modulus_128(n) (SEAL 2.3.1: 1, 2, 4, 8 primes of 54 / 55 bits) -- so that RESULTS.md
can set this repository's per-unit time beside the reference's per-call time, row by row.  Batched kernels have no "call": the unit is
what one call of the reference processes (one 8x8 block of one channel, one pixel, one Cubic, one Linear, one ciphertext), the figure is
device time of a resident batch / units (HIP events on the launch stream; inputs resident in HBM).

usage (GPU box): python tools/results_table.py > gpurun_out/r06_results_table.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe  # noqa: E402

# the reference's published means (with the benchmark/results.txt lines cited there)
PUBLISHED_MS = {"DCT": {2048: 55.7, 4096: 199.2, 8192: 762.6, 16384: 3093.0}, "RGBYCC": {2048: 1.90, 4096: 6.72, 8192: 24.97, 16384: 102.4},
                "Cubic": {2048: 9.08, 4096: 31.66, 8192: 122.4, 16384: 526.1}, "Linear": {2048: 3.06, 4096: 10.40, 8192: 39.42, 16384: 170.2},
                "Encryption": {2048: 1.76, 4096: 3.70, 8192: 8.26, 16384: 23.4}, "Decryption": {2048: 0.132, 4096: 0.429, 8192: 1.55, 16384: 6.8}}
PRESETS = {2048: "SEAL23_2048", 4096: "SEAL23_4096", 8192: "P8192", 16384: "SEAL23_16384"}
HBM = 8000.0


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3           # seconds per call of fn


def measure(n):
    ctx = fhe.SEALContext.preset(PRESETS[n])
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    ctb = 2 * ctx.k * ctx.n * 8                        # bytes of a size-2 ciphertext
    out = {"preset": PRESETS[n], "k": ctx.k, "ct_bytes": ctb}
    # DCT (+ quant: 64 more multiply_plain the reference's timer does not contain -- the repository's figure is for MORE work)
    B = max(16, min(512, (6 << 30) // (64 * ctb)))
    blocks = ctx.random_ct(B, 64, seed=fhe.SEED)
    res = torch.empty_like(blocks)
    plan = fhe.DctPlan(ctx, fhe.YQT)
    s = timed(lambda: ev.dct8x8_quant(plan, blocks, out=res))
    out["DCT"] = {"unit": "8x8 block of one channel (encrypted_dct + quantize_fhe)", "units": B, "us_per_unit": s / B * 1e6, "hbm_frac": B * 128 * ctb / s / 1e9 / HBM}
    del blocks, res
    # RGBYCC
    px = max(256, min(16384, (3 << 30) // (6 * ctb)))
    r, g, b = (ctx.random_ct(px, seed=fhe.SEED + i) for i in range(3))
    s = timed(lambda: ev.rgb_to_ycc(r, g, b))
    out["RGBYCC"] = {"unit": "pixel (rgb_to_ycc_fhe)", "units": px, "us_per_unit": s / px * 1e6, "hbm_frac": px * 6 * ctb / s / 1e9 / HBM}
    del r, g, b
    # Cubic: the reference's mix -- SampleBicubic = four Cubics on size-2 operands + one on size-4 operands (its timer averages them)
    W = H = 32
    w = h = 16
    pixels = ctx.random_ct(W * H, seed=fhe.SEED)
    taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    xf, yf = ctx.random_ct(w * h, seed=11), ctx.random_ct(w * h, seed=12)
    try:
        s = timed(lambda: fhe.circuits.sample_bicubic(ev, pc, pixels, taps, xf, yf), reps=2)
        alg = (16 + 2) * ctb + 3 * ctb                  # per pixel: 16 taps + 2 offsets read, one ct(6) written
        out["Cubic"] = {"unit": "Cubic (mean over the 4 row + 1 column Cubics of SampleBicubic)", "units": 5 * w * h, "us_per_unit": s / (5 * w * h) * 1e6,
                        "hbm_frac": w * h * alg / s / 1e9 / HBM}
        tl, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=False)
        s = timed(lambda: fhe.circuits.sample_linear(ev, pc, pixels, tl, xf, yf), reps=2)
        alg = (4 + 2) * ctb + 2 * ctb
        out["Linear"] = {"unit": "Linear (mean over the 2 row + 1 column Linears of SampleLinear)", "units": 3 * w * h, "us_per_unit": s / (3 * w * h) * 1e6,
                         "hbm_frac": w * h * alg / s / 1e9 / HBM}
    except Exception as exc:                             # n = 2048 with one 54-bit prime has no room for two levels of products in some builds: say so
        out["Cubic"] = out["Linear"] = {"error": str(exc)[:200]}
    del pixels, xf, yf
    # Encryption / Decryption (the reference's client side; the servers' own encryptions are the same call)
    kg = fhe.KeyGenerator(ctx, seed=1)
    der = fhe.DeviceEncryptor(ctx, kg.public_key())
    cnt = max(64, min(2048, (2 << 30) // ctb))
    vals = np.linspace(0.0, 255.0, cnt)
    s = timed(lambda: der.encrypt_values(vals))
    out["Encryption"] = {"unit": "encode + encrypt of one value (device batch)", "units": cnt, "us_per_unit": s / cnt * 1e6, "hbm_frac": cnt * ctb / s / 1e9 / HBM}
    cts = der.encrypt_values(vals)
    dec = fhe.Decryptor(ctx, kg.secret_key())
    scratch = {}

    def dec_device_only():
        L = fhe._lib.load()
        if "s" not in scratch:
            need = int(L.fhe_decrypt_scratch_bytes(ctx.h, 2, cnt))
            scratch["s"] = torch.empty((need + 7) // 8, dtype=torch.int64, device=ctx.device)
            scratch["p"] = torch.empty((cnt, ctx.n), dtype=torch.int64, device=ctx.device)
            scratch["b"] = torch.zeros(cnt, dtype=torch.int32, device=ctx.device)
        fhe._lib.call("fhe_decrypt_batch", ctx.h, dec._sk_ntt.data_ptr(), cts.data_ptr(), 2, cnt, scratch["p"].data_ptr(), scratch["b"].data_ptr(),
                      scratch["s"].data_ptr(), scratch["s"].numel() * 8, torch.cuda.current_stream().cuda_stream)
    s = timed(dec_device_only)
    out["Decryption"] = {"unit": "decrypt of one size-2 ciphertext incl. the noise budget (device batch, plaintexts left on the device)", "units": cnt,
                         "us_per_unit": s / cnt * 1e6, "hbm_frac": cnt * ctb / s / 1e9 / HBM}
    for name, rec in out.items():
        if isinstance(rec, dict) and "us_per_unit" in rec:
            rec["reference_ms_per_call"] = PUBLISHED_MS[name][n]
            rec["ratio_reference_over_this"] = PUBLISHED_MS[name][n] * 1e3 / rec["us_per_unit"]
    return out


if __name__ == "__main__":
    res = {"device": torch.cuda.get_device_name(0), "what": __doc__.split("\n\n")[0], "rows": {}}
    for n in (2048, 4096, 8192, 16384):
        res["rows"][str(n)] = measure(n)
        torch.cuda.empty_cache()
    # the BASELINE.json headline set (3 moduli of 36 / 37 bits at n = 4096) beside the SEAL 2.3.1 set of the same degree
    ctx = fhe.SEALContext.preset("P4096")
    ev = fhe.Evaluator(ctx)
    blocks = ctx.random_ct(1024, 64, seed=fhe.SEED)
    outb = torch.empty_like(blocks)
    plan = fhe.DctPlan(ctx, fhe.YQT)
    s = timed(lambda: ev.dct8x8_quant(plan, blocks, out=outb), reps=5)
    res["headline_P4096"] = {"us_per_block": s / 1024 * 1e6, "blocks_per_s": 1024 / s, "hbm_frac": 1024 * 25165824 / s / 1e9 / HBM}
    print(json.dumps(res, indent=1))
