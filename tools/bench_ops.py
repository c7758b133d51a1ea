#!/usr/bin/env python3
"""Per-primitive throughput of the Evaluator entry points on
resident batches: milliseconds, ciphertexts/s and algorithmic GB/s (bytes the op must read + write)
against the 8 TB/s HBM roofline.  Secondary measurement, one JSON line per op."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

preset = sys.argv[1] if len(sys.argv) > 1 else "P4096"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = fhe.SEALContext.preset(preset)
ev = fhe.Evaluator(ctx)
enc = fhe.FractionalEncoder(ctx)
ct_bytes = 2 * ctx.k * ctx.n * 8
a, b = ctx.random_ct(B, seed=1), ctx.random_ct(B, seed=2)
out = torch.empty_like(a)
pp = fhe.PreparedPlain(ctx, enc.encode(0.587))
plain = enc.encode(128.0)


def timed(fn, steps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def report(name, ms, nbytes, count=B):
    print(json.dumps({"preset": preset, "op": name, "batch": count, "ms": round(ms, 4), "ct_per_s": round(count / ms * 1e3),
                      "algorithmic_GB_per_s": round(nbytes / ms / 1e6, 1), "hbm_frac": round(nbytes / ms / 1e6 / 8000, 4)}), flush=True)


report("add", timed(lambda: ev.add(a, b, out=out)), 3 * B * ct_bytes)
report("sub", timed(lambda: ev.sub(a, b, out=out)), 3 * B * ct_bytes)
report("negate", timed(lambda: ev.negate(a, out=out)), 2 * B * ct_bytes)
report("add_plain", timed(lambda: ev.add_plain(a, plain)), 0.0)
report("multiply_plain", timed(lambda: ev.multiply_plain(a, pp, out=out)), 2 * B * ct_bytes)
report("ntt_forward", timed(lambda: ev.ntt_forward(a, out=out)), 2 * B * ct_bytes)
report("ntt_inverse", timed(lambda: ev.ntt_inverse(a, out=out)), 2 * B * ct_bytes)
report("dyadic_multiply", timed(lambda: ev.dyadic_multiply(a, b, out=out)), 3 * B * ct_bytes)
Bm = max(1, B // 8)
am, bm = a[:Bm].contiguous(), b[:Bm].contiguous()
report("multiply 2x2->3", timed(lambda: ev.multiply(am, bm), 20), (2 + 2 + 3) * Bm * ct_bytes / 2, Bm)
if preset == "P8192":      # counter traffic of the product's launches (tools/collect_traffic.py ctct), quoted only for the sources it was measured on
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_ctct.json")
    if os.path.exists(tpath):
        import collect_traffic
        tj = json.load(open(tpath))
        same = tj.get("kernel_source_hash") == collect_traffic.ctct_source_hash()
        print(json.dumps({"preset": preset, "op": "multiply 2x2->3 traffic", "hbm_bytes_per_product": round(tj["hbm_bytes_per_product"]) if same else None,
                          "algorithmic_bytes_per_product": tj["algorithmic_bytes_per_product"], "ratio_to_algorithmic": round(tj["ratio_to_algorithmic"], 2) if same else None,
                          "note": None if same else "profiles/pmc_traffic_ctct.json was measured on other kernel sources; re-run tools/collect_traffic.py ctct"}), flush=True)
report("square 2->3", timed(lambda: ev.square(am), 20), (2 + 3) * Bm * ct_bytes / 2, Bm)
dbc = 30
evk = fhe.KeyGenerator(ctx, seed=3).generate_evaluation_keys(dbc)
c3 = ctx.random_ct(Bm, size=3, seed=4)
report("relinearize 3->2 (dbc 30)", timed(lambda: ev.relinearize(c3, evk, dbc), 20), (3 + 2) * Bm * ct_bytes / 2, Bm)
