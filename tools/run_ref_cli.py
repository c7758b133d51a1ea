"""Run the reference's unmodified client_jpeg / server_jpeg mains (oracle/_ref, built by oracle/Makefile
target `ref`) through the facade on this GPU: BASELINE.json configs[0] (48x48 RGB, n=4096) end to end.
Prints wall time per stage and the reference's own RMSError line."""
import os, subprocess, sys, tempfile, time
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = int(sys.argv[1]) if len(sys.argv) > 1 else 48
h = int(sys.argv[2]) if len(sys.argv) > 2 else 48
extra = sys.argv[3:]                 # e.g. --golden --pmod 3001
golden = "--golden" in extra
extra = [a for a in extra if a != "--golden"]
cl, sv = (os.path.join(ROOT, "oracle", "_ref", b) for b in ("ref_client_jpeg", "ref_server_jpeg"))
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    os.makedirs(d + "/keys"); os.makedirs(d + "/image")
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([40 + 4 * xx, 200 - 3 * yy, 90 + 2 * xx + 1 * yy], axis=-1).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(d + "/image/in.jpg", quality=95, subsampling=0)
    if golden:                       # the reference's own benchmark image (benchmark/benchmark.py:5), 48x48
        import shutil; shutil.copy(os.path.join(ROOT, "tests", "golden", "boazbarak.jpg"), d + "/image/in.jpg")
    for name, argv in (("client --send", [cl, "--send", "-f", "image/in.jpg", "-c", "image/ct_in.txt", "--cmod", "4096"] + extra),
                       ("server_jpeg", [sv, "-f", "image/ct_in.txt", "-o", "image/ct_out.txt", "--cmod", "4096"] + extra),
                       ("client --recieve", [cl, "--recieve", "-f", "image/in.jpg", "-i", "image/ct_out.txt", "-o", "image/out.jpg", "--cmod", "4096"] + extra)):
        # FHE_REF_SERVER_PREFIX="rocprofv3 --kernel-trace --stats -d <dir> -o p --": the server step under the profiler (its kernel
        # statistics are the launches the reference's unchanged loop turns into; the wall time then includes the profiler)
        if name == "server_jpeg" and os.environ.get("FHE_REF_SERVER_PREFIX"):
            argv = os.environ["FHE_REF_SERVER_PREFIX"].split() + argv
        t0 = time.time()
        r = subprocess.run(argv, cwd=d, capture_output=True, text=True, env=dict(os.environ, FHE_FACADE_STATS="1", TMPDIR="/tmp"))
        dt = time.time() - t0
        print(f"{name}: rc={r.returncode} {dt:.2f} s", flush=True)
        if r.returncode: print(r.stdout[-1500:], r.stderr[-1500:]); sys.exit(1)
        if name == "server_jpeg":
            dct = [float(x) for ln in r.stdout.splitlines() if ln.startswith("DCT,") for x in ln.split(",")[1:] if x]
            ycc = [float(x) for ln in r.stdout.splitlines() if ln.startswith("RGBYCC,") for x in ln.split(",")[1:] if x]
            print(f"  reference's own timers: encrypted_dct mean {np.mean(dct):.2f} ms x{len(dct)}, rgb_to_ycc mean {np.mean(ycc):.3f} ms x{len(ycc)}")
        for ln in r.stdout.splitlines():
            if ln.startswith("RMSError"): print(" ", ln)
        for ln in r.stderr.splitlines():
            if ln.startswith("[seal facade]") and name == "server_jpeg": print(" ", ln)
    print("ct file MB:", os.path.getsize(d + "/image/ct_in.txt") / 1e6)
