import sys, subprocess, numpy as np, time
sys.path.insert(0, '.')
from oracle import oracle as om
orc = om.Oracle.preset("P4096")
cts = orc.random_ct(67, seed=om.SEED)
cts.tofile("/tmp/in.bin")
for i in range(2):
    t0 = time.time()
    r = subprocess.run(["oracle/_ref/ref_jpeg_circuit", "4096", "/tmp/in.bin", "/tmp/out.bin"], capture_output=True, text=True)
    print("wall", time.time() - t0, "stdout:", r.stdout.strip()[:200])
