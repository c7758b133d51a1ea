"""Run the reference's unmodified client_resize / server_resize mains (oracle/_ref) through the facade on this GPU, as
benchmark/benchmark.py:18-29 does: image/boazbarak.jpg 48x48 -> 17x17.  The server runs in both facade modes (lazy, the
default, and FHE_FACADE_EAGER=1); prints its wall time, the facade's statistics and the reference's own RMSError line.
usage: python tools/run_ref_resize.py [bicubic|bilinear] [n=4096] [t=101]"""
import os, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
inter = sys.argv[1] if len(sys.argv) > 1 else "bicubic"
n = sys.argv[2] if len(sys.argv) > 2 else "4096"
t = sys.argv[3] if len(sys.argv) > 3 else "101"
cl, sv = (os.path.join(ROOT, "oracle", "_ref", b) for b in ("ref_client_resize", "ref_server_resize"))
par = ["--width", "17", "--height", "17", "--cmod", n, "--pmod", t]
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    os.makedirs(d + "/keys"); os.makedirs(d + "/image")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "boazbarak.jpg"), d + "/image/in.jpg")

    def run(name, argv, env=None):
        t0 = time.time()
        r = subprocess.run(argv, cwd=d, capture_output=True, text=True, env=dict(os.environ, FHE_FACADE_STATS="1", **(env or {})))
        print("%s: rc=%d %.2f s" % (name, r.returncode, time.time() - t0), flush=True)
        if r.returncode:
            print(r.stdout[-1500:], r.stderr[-1500:]); sys.exit(1)
        for ln in r.stderr.splitlines():
            if ln.startswith("[seal facade]") and "server" in name: print(" ", ln)
        return r.stdout

    run("client --send", [cl, "--send", "-f", "image/in.jpg", "-o", "image/ct_in.txt"] + par)
    sargs = [sv, "-f", "image/ct_in.txt"] + par + (["--bicubic"] if inter == "bicubic" else [])
    run("server_resize (%s, lazy)" % inter, sargs + ["-o", "image/ct_out.txt"])
    run("server_resize (%s, eager)" % inter, sargs + ["-o", "image/ct_out_eager.txt"], {"FHE_FACADE_EAGER": "1"})
    same = open(d + "/image/ct_out.txt", "rb").read() == open(d + "/image/ct_out_eager.txt", "rb").read()
    print("  output streams identical in both modes:", same, "(expected False here: the server encrypts the offsets with fresh randomness)")
    out = run("client --recieve", [cl, "--recieve", "-f", "image/in.jpg", "-c", "image/ct_out.txt", "-o", "image/out.png"] + par)
    for ln in out.splitlines():
        if ln.startswith("RMSError"): print(" ", ln)
