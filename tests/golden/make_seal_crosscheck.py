#!/usr/bin/env python3
"""Generate tests/golden/seal_crosscheck.json: the values oracle/seal_crosscheck.cpp must print, computed by the CPU oracle
(oracle/oracle.py over oracle/fhe_oracle.c) -- one entry per output line of the harness, same names, same order of evaluation.

    python tests/golden/make_seal_crosscheck.py

The harness itself touches nothing but the seal/seal.h API; this script is its independent restatement on the oracle's own
API (arrays in, arrays out).  Line format of the harness: `<name> <sha256 of the u64 words> <16 sampled words in hex>`; sampled
word j is word (j * 2654435761 + 7) mod len.
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as om  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seal_crosscheck.json")
CONSTS = [0.541196100, 0.765366865, -1.847759065, 1.175875602, 0.298631336, 2.053119869, 3.072711026,
          1.501321110, -0.899976223, -2.562915447, -1.961570560, -0.390180644, 0.125, 128.0, 3.0, 0.5,
          -0.168736, 1 / 16.0, 1 / 99.0, -4.71238898038469, 0.0, 1.0, -1.0, 255.0]


def entry(a):
    w = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    return {"sha256": hashlib.sha256(w.tobytes()).hexdigest(), "sample": ["%x" % int(w[(j * 2654435761 + 7) % w.size]) for j in range(16)]}


def main():
    om.build()
    p = om.PRESETS["SEAL23_4096"]
    orc = om.Oracle(p["n"], p["q"], p["t"])
    k, n = orc.k, orc.n
    ctw = 2 * k * n
    out = {}

    def fill(size, first):
        return orc.random_ct(1, size=size, seed=om.SEED, first_index=first)[0]

    A, B, C3, D4 = fill(2, 0), fill(2, ctw), fill(3, 2 * ctw), fill(4, 4 * ctw)
    out["input_A"], out["input_D4"] = entry(A), entry(D4)
    out["add"], out["sub"], out["negate"] = entry(orc.add(A, B)), entry(orc.sub(A, B)), entry(orc.negate(A))
    out["add32"], out["sub23"] = entry(orc.add(C3, A)), entry(orc.sub(A, C3))
    for i, c in enumerate(CONSTS):
        plain = orc.encode(c)
        out["encode[%d]" % i] = entry(plain)
        out["add_plain[%d]" % i] = entry(orc.add_plain(A, plain))
        out["sub_plain[%d]" % i] = entry(orc.sub_plain(A, plain))
        if c != 0.0:
            out["multiply_plain[%d]" % i] = entry(orc.multiply_plain(A, plain))
    P3 = orc.multiply(A, B)
    out["multiply22"], out["multiply32"], out["multiply43"] = entry(P3), entry(orc.multiply(C3, A)), entry(orc.multiply(D4, C3))
    out["square2"], out["square3"] = entry(orc.square(A)), entry(orc.square(C3))
    for dbc in (16, 30):
        nd = int(orc.L.fo_evk_digits(orc.h, dbc))
        evks = np.zeros((2, k, nd, 2, k, n), dtype=np.uint64)
        first = (100 + dbc) * ctw
        for j in range(2):
            for l in range(k * nd):                       # key l = prime * digits + digit: a size-2 ciphertext, coefficient form -> NTT form
                key = fill(2, first)
                first += ctw
                for poly in range(2):
                    for i in range(k):
                        evks[j, l // nd, l % nd, poly, i] = orc.ntt_fwd(key[poly, i], i)
        out["relin%d_3" % dbc] = entry(orc.relinearize_n(P3, evks, dbc))
        out["relin%d_4" % dbc] = entry(orc.relinearize_n(D4, evks, dbc))
    t, E2 = fill(2, 9 * ctw), fill(2, 10 * ctw)
    out["cubic"] = entry(om.oracle_cubic_calls(orc, A, B, E2, t, t))
    out["linear"] = entry(om.oracle_linear_calls(orc, A, B, t))
    # homomorphic_sin on a real encryption at n = 8192: the decrypted plaintext polynomial does not depend on keys or noise
    p8 = om.PRESETS["P8192"]
    o8 = om.Oracle(p8["n"], p8["q"], p8["t"])
    sk, pk = o8.keygen(5)
    x, zero = o8.encrypt(pk, o8.encode(4.0), seed=1), o8.encrypt(pk, o8.encode(0.0), seed=2)
    res = om.oracle_homomorphic_sin(o8, x, zero)
    plain, budget = o8.decrypt(sk, res)
    assert budget > 0, budget
    out["sin_plain"] = entry(plain)
    out["sin_value"] = {"value": "%.9f" % o8.decode(plain), "size": int(res.shape[0]), "true_sin": "%.9f" % math.sin(4.0)}
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/make_seal_crosscheck.py (oracle/fhe_oracle.c through oracle/oracle.py)",
                   "parameters": {"n": n, "q": [hex(q) for q in p["q"]], "t": p["t"], "encoder": [100, 100, 2], "seed": hex(om.SEED),
                                  "sin": {"n": p8["n"], "q": [hex(q) for q in p8["q"]]}},
                   "lines": out}, f, indent=1)
    print("wrote %s: %d lines" % (OUT, len(out)))


if __name__ == "__main__":
    main()
