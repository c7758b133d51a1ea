#!/usr/bin/env python3
"""Rate of the servers' own encryptions (homo/fhe_resize.h:230,234,262,266: two per output pixel; homo/fhe_decode.h:54,134): the host
sampler one ciphertext at a time (keys.Encryptor, rounds 2-4) against device batches (keys.DeviceEncryptor: fhe_frac_encode_batch +
fhe_encrypt_batch), and decryption the same way (keys.Decryptor.decrypt_host: CRT composition and big-integer rounding in Python; decrypt_batch:
fhe_decrypt_batch).  usage: python tools/bench_encrypt.py [preset=P8192] [batch=512]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

preset = sys.argv[1] if len(sys.argv) > 1 else "P8192"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = fhe.SEALContext.preset(preset)
kg = fhe.KeyGenerator(ctx)
vals = np.random.default_rng(1).uniform(0, 1, batch)
host = fhe.server.make_fraction_encryptor(ctx, kg.public_key(), device=False)
dev = fhe.server.make_fraction_encryptor(ctx, kg.public_key())
host(vals[:4]); dev(vals)
torch.cuda.synchronize()
t0 = time.perf_counter()
host(vals[:64])
torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) / 64
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    out = dev(vals)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / (reps * batch)
# device time alone (events): the five launches of a batch
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    dev(vals)
b.record()
torch.cuda.synchronize()
# decryption (the clients' half): the host big-integer rounding of rounds 1-4 against fhe_decrypt_batch
dec = fhe.Decryptor(ctx, kg.secret_key())
dec.decrypt_batch(out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4):
    dec.decrypt_host(out[i])
t_dhost = (time.perf_counter() - t0) / 4
t0 = time.perf_counter()
for _ in range(5):
    dec.decrypt_batch(out)
t_ddev = (time.perf_counter() - t0) / (5 * batch)
print(json.dumps({"workload": "server-side encryptions of encode(fraction), %s" % preset, "batch": batch,
                  "host_sampler_us_per_ciphertext": t_host * 1e6, "device_batch_us_per_ciphertext": t_dev * 1e6,
                  "device_batch_gpu_us_per_ciphertext": a.elapsed_time(b) * 1e3 / (reps * batch), "speedup": t_host / t_dev,
                  "encryptions_per_s_device": 1 / t_dev,
                  "decrypt_host_python_bigint_us_per_ciphertext": t_dhost * 1e6, "decrypt_device_batch_us_per_ciphertext_incl_download": t_ddev * 1e6}))
