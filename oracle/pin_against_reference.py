"""Pin the CPU oracle against the reference's own published outputs (test infrastructure).

The reference's benchmark (benchmark/benchmark.py:33-43) runs client_jpeg --send, server_jpeg and
client_jpeg --recieve on image/boazbarak.jpg for poly degrees {2048,4096,8192,16384} x nine plain
moduli and records the resulting `RMSError` line in benchmark/results.txt (values below, identical
for every degree).  The values are deterministic functions of the plaintext-level semantics --
FractionalEncoder encode/decode, centred lifting, multiply_plain / add_plain / sub_plain wrap-around
mod t (five of the nine sets overflow t on purpose) -- and of a correct encrypt/decrypt.

This script runs the reference's UNMODIFIED mains, compiled by oracle/Makefile against seal/seal.h and
the oracle-backed C ABI (oracle/libfhe_cabi_oracle.so), on tests/golden/boazbarak.jpg and compares.
usage: python oracle/pin_against_reference.py [--n 2048] [--pmod 11 31 ...] [--jobs 3] [--gpu]
  --gpu uses oracle/_ref/ref_*_jpeg (libfhe_hip.so, needs an MI355X) instead of the *_cpu builds.
"""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# benchmark/results.txt: jpg_boaz_<n>_<t>.txt -> RMSError (same for n = 2048, 4096, 8192, 16384)
PUBLISHED = {11: "72.7491", 31: "77.6639", 101: "114.663", 307: "35.672", 1009: "1.71783",
             3001: "1.71767", 10007: "1.71767", 30011: "1.71767", 100003: "1.71767"}


def run_set(n, t, gpu=False, image=None):
    sfx = "" if gpu else "_cpu"
    cl = os.path.join(ROOT, "oracle", "_ref", "ref_client_jpeg" + sfx)
    sv = os.path.join(ROOT, "oracle", "_ref", "ref_server_jpeg" + sfx)
    image = image or os.path.join(ROOT, "tests", "golden", "boazbarak.jpg")
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        os.makedirs(d + "/keys"); os.makedirs(d + "/image")
        shutil.copy(image, d + "/image/in.jpg")
        par = ["--cmod", str(n), "--pmod", str(t)]
        for argv in ([cl, "--send", "-f", "image/in.jpg", "-c", "image/ct_in.txt"] + par,
                     [sv, "-f", "image/ct_in.txt", "-o", "image/ct_out.txt"] + par,
                     [cl, "--recieve", "-f", "image/in.jpg", "-i", "image/ct_out.txt", "-o", "image/out.jpg"] + par):
            r = subprocess.run(argv, cwd=d, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(" ".join(argv) + "\n" + r.stdout[-1000:] + r.stderr[-1000:])
        rms = [ln.split(",")[1] for ln in r.stdout.splitlines() if ln.startswith("RMSError,")]
    return rms[0] if rms else None, time.time() - t0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--pmod", type=int, nargs="*", default=sorted(PUBLISHED))
    ap.add_argument("--jobs", type=int, default=3)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    with ThreadPoolExecutor(a.jobs) as ex:
        res = list(ex.map(lambda t: (t,) + run_set(a.n, t, a.gpu), a.pmod))
    rows, ok = [], True
    for t, rms, dt in res:
        good = rms == PUBLISHED[t]
        ok &= good
        rows.append({"n": a.n, "plain_modulus": t, "rms": rms, "published": PUBLISHED[t], "match": good, "seconds": round(dt, 1)})
        print(rows[-1], flush=True)
    if a.out:
        json.dump({"backend": "libfhe_hip.so (MI355X)" if a.gpu else "CPU oracle (libfhe_cabi_oracle.so)", "rows": rows},
                  open(a.out, "w"), indent=1)
    sys.exit(0 if ok else 1)
