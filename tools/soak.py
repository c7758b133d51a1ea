#!/usr/bin/env python3
"""Soak test: repeat the fused kernels on the same inputs and compare output digests every time
(would expose a rare ordering hazard in the barrier-free in-wave LDS exchanges or the packed
intermediate, the hand-written LDS-DMA waits of the row kernel, or the lazy ranges of the u64 kernels).
usage: python tools/soak.py [iterations] [preset]     (preset P4096 by default; SEAL23_4096 / P8192 run the u64 kernels,
and every preset also repeats a ct x ct product)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
preset = sys.argv[2] if len(sys.argv) > 2 else "P4096"
ctx = fhe.SEALContext.preset(preset)
ev = fhe.Evaluator(ctx)
plan = fhe.DctPlan(ctx, fhe.YQT)
blocks = ctx.random_ct(1024 if ctx.n <= 4096 else 256, 64, seed=fhe.SEED)
out = torch.empty_like(blocks)
ev.dct8x8_quant(plan, blocks, out=out)
ref = ctx.digest(out.view(-1))
r0, g0, b0 = (ctx.random_ct(4096, seed=11 + i) for i in range(3))
r, g, b = r0.clone(), g0.clone(), b0.clone()
ev.rgb_to_ycc(r, g, b)
ref_rgb = ctx.digest(torch.cat([r, g, b]).view(-1))
a = ctx.random_ct(4096, seed=5)
ref_ntt = ctx.digest(ev.ntt_inverse(ev.ntt_forward(a)).view(-1))
assert ref_ntt == ctx.digest(a.view(-1))
ma, mb = ctx.random_ct(256, seed=21), ctx.random_ct(256, seed=22)
ref_mul = ctx.digest(ev.multiply(ma, mb).view(-1))
# the library circuits (csrc/circuits.hip): index arrays through the page-locked staging ring, scratch arena reuse, back to back without a
# host synchronisation in between -- a stale index slot or an arena overlap would change a digest
pc = fhe.circuits.PlainCache(ctx)
W, H, w, h = 12, 10, 7, 6
pix = ctx.random_ct(W * H, seed=31)
taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
xf, yf = ctx.random_ct(w * h, seed=32), ctx.random_ct(w * h, seed=33)
half = (w * h) // 2
ref_bic = [ctx.digest(fhe.circuits.sample_bicubic(ev, pc, pix, taps[s:e], xf[s:e].contiguous(), yf[s:e].contiguous()).view(-1)) for s, e in ((0, half), (half, w * h))]
# approximated_step: the offset chain through the staging ring, the in-place forward transforms of the plaintext sums and the harmonic sum
# that accumulates in place (k_mulplain_fwd_pm / k_sum_inv_pm on pseudo-Mersenne bases, the separate products elsewhere)
s_amp, s_idx, s_cnt = (ctx.random_ct(1, size=2, seed=41 + i) for i in range(3))
s_zeros = ctx.random_ct(3 * 3 * 2, size=2, seed=44).reshape(3, 3, 2, 2, ctx.k, ctx.n)


def step_digest():
    return ctx.digest(torch.cat(fhe.circuits.approximated_step(ev, pc, s_amp, s_idx, s_cnt, order=64, degree=3, delta=0.5, width=3, height=1, zeros=s_zeros)).view(-1))


ref_step = step_digest()
# the relinearised mode (fhe_circuits_create_relin): the same circuits with a key switch after every product -- the wave-local
# transposes of the digit transforms, the relinearisation scratch inside the arena, sizes 2 throughout
relin = (fhe.KeyGenerator(ctx, seed=1).generate_evaluation_keys(30).contiguous(), 30)


def relin_digests():
    o = fhe.circuits.sample_bicubic(ev, pc, pix, taps[:half], xf[:half].contiguous(), yf[:half].contiguous(), relin=relin)
    st = torch.cat(fhe.circuits.approximated_step(ev, pc, s_amp, s_idx, s_cnt, order=64, degree=3, delta=0.5, width=3, height=1, zeros=s_zeros, relin=relin))
    return [ctx.digest(o.view(-1)), ctx.digest(st.view(-1))]


ref_relin = relin_digests()
# round 6: the other two placements (one relinearize per Cubic / per output pixel: fhe_relinearize_n's one-pass key switches for s^2 .. s^5, the
# scatter of the tail results) and the shared-offset resize with its bands
keys4 = fhe.KeyGenerator(ctx, seed=1).generate_evaluation_keys(30, 4).contiguous()
sxf, syf = ctx.random_ct(w, seed=34), ctx.random_ct(h, seed=35)


def placement_digests():
    oc = fhe.circuits.sample_bicubic(ev, pc, pix, taps[:half], xf[:half].contiguous(), yf[:half].contiguous(), relin=(keys4[:2].contiguous(), 30, "cubic"))
    os_ = fhe.circuits.sample_bicubic(ev, pc, pix, taps[half:], xf[half:].contiguous(), yf[half:].contiguous(), relin=(keys4, 30, "sample"))
    sh = fhe.circuits.resize_bicubic_shared(ev, pc, pix, W, H, w, h, sxf, syf, batch=16, band_rows=2)
    rl = ev.relinearize(ctx.random_ct(8, size=6, seed=36), keys4, 30)
    return [ctx.digest(t.reshape(-1)) for t in (oc, os_, sh, rl)]


ref_place = placement_digests()
# the servers' own encryptions as device batches (csrc/encrypt.hip): the staging ring of the encoder's values, the scratch reuse, the
# keyed sampler -- the same (key, index) range must give the same ciphertexts every time
der = fhe.DeviceEncryptor(ctx, fhe.KeyGenerator(ctx, seed=2).public_key(), key=bytes(range(32)), reproducible=True)
enc_vals = [i / 97.0 - 1.5 for i in range(192)]


def enc_digest():
    der.seek(1 << 33)
    return ctx.digest(der.encrypt_values(enc_vals).view(-1))


ref_enc = enc_digest()
soak_dec = fhe.Decryptor(ctx, fhe.KeyGenerator(ctx, seed=2).secret_key())


def dec_digest():                      # fhe_decrypt_batch of the same batch: the per-ciphertext atomicMax of the noise bits and the plaintexts
    der.seek(1 << 33)
    plains, budgets = soak_dec.decrypt_batch(der.encrypt_values(enc_vals), with_budget=True)
    return hash((plains.tobytes(), tuple(budgets)))


ref_dec = dec_digest()
bad = 0
t0 = time.time()
for i in range(iters):
    out.zero_()
    ev.dct8x8_quant(plan, blocks, out=out)
    if ctx.digest(out.view(-1)) != ref:
        bad += 1
        print("DCT digest mismatch at iteration", i, flush=True)
    if i % 4 == 0:
        r.copy_(r0); g.copy_(g0); b.copy_(b0)
        ev.rgb_to_ycc(r, g, b)
        if ctx.digest(torch.cat([r, g, b]).view(-1)) != ref_rgb:
            bad += 1
            print("rgb digest mismatch at iteration", i, flush=True)
        if ctx.digest(ev.ntt_inverse(ev.ntt_forward(a)).view(-1)) != ref_ntt:
            bad += 1
            print("ntt digest mismatch at iteration", i, flush=True)
        if ctx.digest(ev.multiply(ma, mb).view(-1)) != ref_mul:
            bad += 1
            print("multiply digest mismatch at iteration", i, flush=True)
    if i % 16 == 0:
        o1 = fhe.circuits.sample_bicubic(ev, pc, pix, taps[:half], xf[:half].contiguous(), yf[:half].contiguous())      # two calls queued back to back
        o2 = fhe.circuits.sample_bicubic(ev, pc, pix, taps[half:], xf[half:].contiguous(), yf[half:].contiguous())
        if [ctx.digest(o1.view(-1)), ctx.digest(o2.view(-1))] != ref_bic:
            bad += 1
            print("sample_bicubic digest mismatch at iteration", i, flush=True)
        if step_digest() != ref_step:
            bad += 1
            print("approximated_step digest mismatch at iteration", i, flush=True)
        if relin_digests() != ref_relin:
            bad += 1
            print("relinearised circuits digest mismatch at iteration", i, flush=True)
    if i % 16 == 8 and placement_digests() != ref_place:
        bad += 1
        print("relinearisation placements / shared resize digest mismatch at iteration", i, flush=True)
    if i % 8 == 0 and enc_digest() != ref_enc:
        bad += 1
        print("device encryption digest mismatch at iteration", i, flush=True)
    if i % 8 == 4 and dec_digest() != ref_dec:
        bad += 1
        print("device decryption mismatch at iteration", i, flush=True)
print("soak %s: %d iterations, %d mismatches, %.1f s" % (preset, iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
