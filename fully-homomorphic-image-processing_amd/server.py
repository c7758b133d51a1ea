"""server_jpeg over a ciphertext stream: the driver loop of homo/server_jpeg.cpp:109-153 on the GPU.

The reference reads one 8x8 block of R, G and B ciphertexts at a time from ./image/nothingpersonnel.txt
(3 x 64 Ciphertext::load calls, :115-124), runs rgb_to_ycc_fhe on the 64 pixels and encrypted_dct on
the three channels (:127-135) and appends the block's 64 Y, then 64 Cb, then 64 Cr ciphertexts to
./image/zoop.txt (save_three_blocks_interleaved_ycc, :146-153: "interleaved by block", channel-major
inside a block -- the order homo/client_jpeg.cpp:266-271 reads back: for k<3, for j<64).  Here the same stream is processed in waves of many blocks: a reader thread fills pinned
host buffers while the GPU works on the previous wave (file -> pinned -> HBM overlap), the colour
conversion and the block transform are the fused kernels, and the writer drains the previous wave.

Stream format = the facade's Ciphertext::save records (seal/seal.h: "FHEHIP1" magic, u32 polys,
u32 k, u32 n, u32 reserved, raw little-endian u64).  SEAL 2.3's own wire format is not pinned by
anything in the reference (no sample files).
"""
import ctypes as C
import os
import struct

import numpy as np
import torch

from .evaluator import DctPlan, Evaluator

MAGIC = b"FHEHIP1\x00"
HEADER = struct.Struct("<8sIIII")


def write_ciphertext(f, ct):
    """ct: numpy uint64 [size, k, n] -> one stream record."""
    size, k, n = ct.shape
    f.write(HEADER.pack(MAGIC, size, k, n, 0))
    f.write(np.ascontiguousarray(ct, dtype="<u8").tobytes())


def write_evaluation_keys(f, evk_ntt, dbc):
    """evk_ntt: KeyGenerator.generate_evaluation_keys(dbc[, count]) ([k][digits][2][k][n] or [count][k][digits][2][k][n], device or host) -> the record
    seal::EvaluationKeys::load reads (seal/seal.h: magic "FHEHIPK", u32 dbc, digits, count, k, n, reserved, then the words, NTT form)"""
    import struct
    t = evk_ntt if evk_ntt.dim() == 6 else evk_ntt[None]
    count, k, digits, two, k2, n = t.shape
    assert two == 2 and k2 == k
    f.write(b"FHEHIPK\0" + struct.pack("<6I", int(dbc), digits, count, k, n, 0))
    f.write(np.ascontiguousarray(t.cpu().numpy().view(np.uint64), dtype="<u8").tobytes())


def read_evaluation_keys(ctx, f):
    """One record of write_evaluation_keys / seal::EvaluationKeys::save -> (evk_ntt [count][k][digits][2][k][n] on the context's device, dbc),
    what Evaluator.relinearize and Circuits(relin=...) take.  The stream is untrusted: the header must name THIS context's (k, n) and the digit count
    its moduli give at the stated dbc (so the payload size is fixed by the context and the number of powers, not by the stream), the payload must be
    complete and every residue reduced -- the checks of seal::EvaluationKeys::load plus the consumer's (seal/seal.h require_for)."""
    import struct
    hdr = f.read(32)
    if len(hdr) != 32 or hdr[:8] != b"FHEHIPK\0":
        raise ValueError("stream does not hold evaluation keys")
    dbc, digits, count, k, n, _ = struct.unpack("<6I", hdr[8:])
    if not (1 <= dbc <= 60 and 1 <= count <= 62):
        raise ValueError("evaluation key header out of range")
    if (k, n) != (ctx.k, ctx.n):
        raise ValueError("evaluation keys for (k, n) = %r, the context has %r" % ((k, n), (ctx.k, ctx.n)))
    want = max((int(q).bit_length() + dbc - 1) // dbc for q in ctx.q)
    if digits != want:
        raise ValueError("evaluation keys with %d digit(s) at dbc %d: this context's moduli take %d" % (digits, dbc, want))
    words = count * k * digits * 2 * k * n
    body = f.read(words * 8)
    if len(body) != words * 8:
        raise EOFError("truncated evaluation key stream")
    evk = np.frombuffer(body, dtype="<u8").reshape(count, k, digits, 2, k, n)
    for i, q in enumerate(ctx.q):
        if int(evk[:, :, :, :, i, :].max()) >= int(q):
            raise ValueError("evaluation key residue not reduced")
    return torch.from_numpy(evk.view(np.int64).copy()).to(ctx.device), dbc


def read_ciphertext_into(f, out):
    """Read one record into out (numpy uint64 [size, k, n], e.g. a view of a pinned buffer)."""
    hdr = f.read(HEADER.size)
    if len(hdr) != HEADER.size:
        raise EOFError("ciphertext stream ended")
    magic, size, k, n, _ = HEADER.unpack(hdr)
    if magic != MAGIC:
        raise ValueError("not a ciphertext record")
    if (size, k, n) != out.shape:
        raise ValueError("ciphertext shape %r does not match the context %r" % ((size, k, n), out.shape))
    got = f.readinto(memoryview(out).cast("B"))
    if got != out.nbytes:
        raise EOFError("truncated ciphertext record")


RECORD_HEADER = HEADER.size
_IO_THREADS = 8
_pinned_cache = {}


def _pinned(tag, shape):
    """pinned staging buffers are expensive to create (page locking): keep them between calls"""
    key = (tag, shape)
    if key not in _pinned_cache:
        _pinned_cache[key] = torch.empty(shape, dtype=torch.int64).pin_memory()
    return _pinned_cache[key]


def _pread_records(fd, views, first_record, rec_bytes, expect):
    """positional scatter reads (header -> scratch, payload -> straight into pinned memory); the
    records are of fixed size, so any number of threads can read disjoint ranges concurrently"""
    hdr = bytearray(RECORD_HEADER)
    for j, v in enumerate(views):
        mv = memoryview(v).cast("B")
        got = os.preadv(fd, [hdr, mv], (first_record + j) * rec_bytes)
        if got != rec_bytes:
            raise EOFError("ciphertext stream ended")
        magic, size, k, n, _ = HEADER.unpack(hdr)
        if magic != MAGIC:
            raise ValueError("not a ciphertext record")
        if (size, k, n) != expect:
            raise ValueError("ciphertext shape %r does not match the context %r" % ((size, k, n), expect))


def _pwrite_records(fd, views, first_record, rec_bytes, hdr):
    for j, v in enumerate(views):
        mv = memoryview(v).cast("B")
        if os.pwritev(fd, [hdr, mv], (first_record + j) * rec_bytes) != rec_bytes:
            raise IOError("short write on the ciphertext stream")


class StreamFile:
    """fhe_io_open / fhe_io_transfer (include/fhe_stream.h): one shared mapping of a ciphertext stream file; records move
    between it and a page-locked staging tensor by parallel memcpy in C threads (ctypes releases the GIL for the call).
    For writing, an existing file of exactly `size` bytes keeps its pages (a reused spool file is memcpy-bound; a fresh
    file is bound by the kernel's page allocation whatever the method)."""

    def __init__(self, path, write=False, size=0):
        from . import _lib
        self._lib = _lib
        h = C.c_void_p()
        _lib.call("fhe_io_open", os.fsencode(path), int(write), size, C.byref(h))
        self.h = h
        self.size = int(_lib.load().fhe_io_size(h))

    def transfer(self, first_record, count, polys, ctx, tensor, threads):
        self._lib.call("fhe_io_transfer", self.h, first_record, count, polys, ctx.k, ctx.n, C.c_void_p(tensor.data_ptr()), threads)

    def close(self):
        if self.h:
            self._lib.load().fhe_io_close(self.h)
            self.h = None


class _ResidueCheck:
    """fhe_count_unreduced on every uploaded wave: the streams' record headers are checked by the I/O layer, the payload here --
    a residue at or above its modulus would otherwise be computed on silently (seal::Ciphertext::load rejects it, and so does
    the facade's).  One u64 counter on the device, read once when the job is done.  The counter is zeroed on the stream that
    is current at construction; every count waits for that event on ITS stream first (the servers count on their copy
    streams, which are non-blocking: without the wait the zero fill and the first atomic add would be unordered)."""

    def __init__(self, ctx, enabled=True):
        from . import _lib
        self.ctx, self._lib, self.enabled = ctx, _lib, enabled
        self.count = torch.zeros(1, dtype=torch.int64, device=ctx.device) if enabled else None
        self._zeroed = None
        if enabled:
            self._zeroed = torch.cuda.Event()
            self._zeroed.record(torch.cuda.current_stream())

    def add(self, t):
        if self.enabled and t.numel():
            st = torch.cuda.current_stream()
            st.wait_event(self._zeroed)
            self._lib.call("fhe_count_unreduced", self.ctx.h, C.c_void_p(t.data_ptr()), t.numel() // (self.ctx.k * self.ctx.n), C.c_void_p(self.count.data_ptr()),
                           C.c_void_p(st.cuda_stream))

    def verdict(self, refuse_output=None):
        """raises ValueError when a residue was not reduced; `refuse_output()` runs first (the servers have written their
        whole output stream by then: it must not be left behind looking complete)"""
        if self.enabled:
            bad = int(self.count.item())
            if bad:
                if refuse_output is not None:
                    refuse_output()
                raise ValueError("the input stream holds %d residues that are not reduced modulo the coefficient moduli; the output was computed on them and has been "
                                 "discarded (an output mapping kept open by the caller is NOT truncated: discard its contents)" % bad)


def _copy_streams():
    """(upload stream, download stream), both of the default priority.  The runtime maps streams onto a few hardware queues and two
    streams can share one (measured in server_resize: both copy streams on HSA queue 4): what matters then is the ORDER of their work
    in that queue -- the servers enqueue the next wave's upload before this wave's download, see the loops.  Tried and rejected: a
    high-priority download stream (its own queue): its blit kernels then pre-empt the compute kernels (server_resize 1.25 -> 1.38 s,
    device compute 0.87 -> 1.25 s; server_jpeg 937 -> 662 colour blocks/s)."""
    return torch.cuda.Stream(), torch.cuda.Stream()


def _refuser(own_out, fout, out_path):
    """what a server does with its output stream when the input turns out to be invalid: a file it opened itself is closed and
    truncated to zero bytes (any reader then fails on the first record); a StreamFile the caller keeps open (a reused spool
    mapping) cannot be truncated under its mapping -- the ValueError tells the caller to discard it"""
    def refuse():
        if own_out:
            fout.close()
            try:
                os.truncate(out_path, 0)
            except OSError:
                pass
    return refuse


def _stop_pipeline(reader_thread, writer_thread, free_in, to_write):
    """Bring a streaming pipeline to rest before its mappings and staging buffers go away (normal end and error paths
    alike): both I/O threads get their end-of-work sentinel and are JOINED -- a thread may still be inside fhe_io_transfer,
    copying into or out of the file mapping with the GIL released, and unmapping under it would be a use-after-unmap --
    and the device is drained so that no asynchronous copy still targets the cached page-locked buffers."""
    free_in.put(None)
    to_write.put(None)
    for th in (reader_thread, writer_thread):
        if th.is_alive():
            th.join()
    torch.cuda.synchronize()


def server_jpeg(ctx, in_path, out_path, n_blocks, wave_blocks=8, quant=None, do_dct=True, io_threads=8, slots=3, stats=None, validate=True):
    """Process `n_blocks` colour blocks.  in_path / out_path: file names, or StreamFile objects a long-lived server keeps
    open (their mappings, and the page-table entries behind them, are then reused from call to call).  Input order per block: 64 R, 64 G, 64 B ciphertexts
    (homo/server_jpeg.cpp:115-124).  Output order per block: 64 Y, 64 Cb, 64 Cr
    (homo/server_jpeg.cpp:146-153; read back channel-major by homo/client_jpeg.cpp:266-271).  quant=None reproduces the reference server (no quantize_fhe call);
    a 64-entry table applies quantize_fhe to every channel as well.  Returns blocks processed.

    Five stages run concurrently on waves of `wave_blocks` blocks:
      file -> pinned   a reader thread: fhe_io_read_records (C threads, positional scatter reads straight into one of
                       `slots` page-locked buffers)
      pinned -> HBM    its own HIP stream, two device input buffers
      compute          fhe_rgb_to_ycc_blocks in place on the stream layout, then fhe_dct8x8_quant over the 3 * wave
                       channel-blocks -- the result already has the output stream's order; no gather / copy kernels
      HBM -> pinned    its own HIP stream, `slots` page-locked output buffers
      pinned -> file   a writer thread: fhe_io_write_records
    Events order the hand-overs; the host never waits for the device except where a buffer is about to be reused.
    stats (a dict), if given, receives wall seconds, device compute seconds and byte counts.
    validate: every uploaded wave is checked for residues that are not below their modulus (fhe_count_unreduced; ValueError at
    the end of the job -- what Ciphertext::load rejects per ciphertext, homo/server_jpeg.cpp:117-123)."""
    import queue
    import threading
    import time
    ev = Evaluator(ctx)
    plan = DctPlan(ctx, quant) if do_dct else None
    residues = _ResidueCheck(ctx, validate)
    wave_blocks = max(1, min(wave_blocks, n_blocks))
    slots = max(2, slots)
    shape = (wave_blocks, 3, 64, 2, ctx.k, ctx.n)            # wave: block, channel, pixel, poly, prime, coeff
    hin = [_pinned(("in", i), shape) for i in range(slots)]
    hout = [_pinned(("out", i), shape) for i in range(slots)]
    din = [torch.empty(shape, dtype=torch.int64, device=ctx.device) for _ in range(2)]
    dout = [torch.empty(shape, dtype=torch.int64, device=ctx.device) for _ in range(2)] if do_dct else din
    main = torch.cuda.current_stream()
    h2d, d2h = _copy_streams()
    rec = RECORD_HEADER + 2 * ctx.k * ctx.n * 8
    if (in_path.size if isinstance(in_path, StreamFile) else os.path.getsize(in_path)) < n_blocks * 192 * rec:
        raise EOFError("ciphertext stream ended")
    waves = [(s, min(s + wave_blocks, n_blocks)) for s in range(0, n_blocks, wave_blocks)]
    own_in, own_out = not isinstance(in_path, StreamFile), not isinstance(out_path, StreamFile)
    fin = StreamFile(in_path) if own_in else in_path
    fout = StreamFile(out_path, write=True, size=n_blocks * 192 * rec) if own_out else out_path
    if fout.size < n_blocks * 192 * rec:
        raise ValueError("output stream file is smaller than the result")
    trace = [] if stats is not None else None
    free_in, ready_in, free_out, to_write = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
    io_seconds = {"read": 0.0, "write": 0.0}
    for i in range(slots):
        free_in.put((i, None))
        free_out.put(i)
    errors = []

    def reader():
        try:
            for wi, (s, e) in enumerate(waves):
                item = free_in.get()
                if item is None:                                # the main loop is shutting the pipeline down
                    return
                slot, copied = item
                if copied is not None:
                    copied.synchronize()                        # the previous wave in this slot has left for the device
                t_io = time.perf_counter()
                fin.transfer(s * 192, (e - s) * 192, 2, ctx, hin[slot], io_threads)
                io_seconds["read"] += time.perf_counter() - t_io
                if trace is not None:
                    trace.append(("read", wi, t_io - t0, time.perf_counter() - t0))
                ready_in.put((wi, slot))
        except BaseException as exc:                           # surfaced by the main loop
            errors.append(exc)
            ready_in.put((None, None))

    def writer():
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                (s, e), slot, landed = item
                landed.synchronize()
                t_io = time.perf_counter()
                fout.transfer(s * 192, (e - s) * 192, 2, ctx, hout[slot], io_threads)
                io_seconds["write"] += time.perf_counter() - t_io
                if trace is not None:
                    trace.append(("write", s // wave_blocks, t_io - t0, time.perf_counter() - t0))
                free_out.put(slot)
        except BaseException as exc:
            errors.append(exc)
            free_out.put(None)

    t0 = time.perf_counter()
    rt, wt = threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)
    try:
        rt.start()
        wt.start()
        computed, drained = [None, None], [None, None]       # per device buffer: compute finished / left for the host
        t_start, t_stop = [], []
        copied_of = {}

        def upload(wi):
            """wave wi: page-locked slot -> device buffer wi & 1, on the upload stream (blocks until the reader thread has the wave)"""
            s, e = waves[wi]
            nb, d = e - s, wi & 1
            got, slot = ready_in.get()
            if got is None:
                raise errors[0]
            with torch.cuda.stream(h2d):
                for evt in (computed[d], drained[d]):           # the device buffer is free again
                    if evt is not None:
                        h2d.wait_event(evt)
                din[d][:nb].copy_(hin[slot][:nb], non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(h2d)
            free_in.put((slot, copied))
            copied_of[wi] = copied

        upload(0)
        for wi, (s, e) in enumerate(waves):
            nb, d = e - s, wi & 1
            main.wait_event(copied_of.pop(wi))
            if drained[d] is not None:
                main.wait_event(drained[d])                      # dout[d] has been copied out
            if stats is not None:
                t_start.append(torch.cuda.Event(enable_timing=True))
                t_start[-1].record(main)
            residues.add(din[d][:nb])
            ev.rgb_to_ycc_blocks(din[d][:nb])                    # in place: Y, Cb, Cr in the stream's block layout
            if do_dct:
                ev.dct8x8_quant(plan, din[d][:nb].view(nb * 3, 64, 2, ctx.k, ctx.n), out=dout[d][:nb].view(nb * 3, 64, 2, ctx.k, ctx.n))
            if stats is not None:
                t_stop.append(torch.cuda.Event(enable_timing=True))
                t_stop[-1].record(main)
            done = torch.cuda.Event()
            done.record(main)
            computed[d] = done
            if wi + 1 < len(waves):
                upload(wi + 1)                                   # before this wave's download: see _copy_streams
            t_w = time.perf_counter()
            oslot = free_out.get()
            if oslot is None:
                raise errors[0]
            if trace is not None:
                trace.append(("main", wi, t_w - t0, time.perf_counter() - t0))
            with torch.cuda.stream(d2h):
                d2h.wait_event(done)
                hout[oslot][:nb].copy_(dout[d][:nb], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(d2h)
            drained[d] = landed
            to_write.put(((s, e), oslot, landed))
        to_write.put(None)
        wt.join()
        rt.join()
        if errors:
            raise errors[0]
        torch.cuda.synchronize()
        residues.verdict(_refuser(own_out, fout, out_path))
        if stats is not None:
            stats.update(seconds=time.perf_counter() - t0, device_compute_seconds=sum(a.elapsed_time(b) for a, b in zip(t_start, t_stop)) / 1e3,
                         bytes_in=n_blocks * 192 * rec, bytes_out=n_blocks * 192 * rec, waves=len(waves),
                         file_read_seconds=io_seconds["read"], file_write_seconds=io_seconds["write"], trace=trace)
    finally:
        _stop_pipeline(rt, wt, free_in, to_write)
        if own_in:
            fin.close()
        if own_out:
            fout.close()
    return n_blocks


# ------------------------------------------------------------------------------------------------
# server_resize: ResizeImage (homo/fhe_resize.h:308-392) over a ciphertext stream
# ------------------------------------------------------------------------------------------------
class _IndexedEncryptions:
    """Test-only reproducible form of the server-side encryptions: the i-th encryption of the reference's sequence is
    drawn from a sampler seeded with (seed, i), so a process that starts in the middle of the sequence (a shard of the
    rows / positions) produces the same ciphertexts as the whole-job run.  `seek(i)` sets the position of the next draw."""

    def __init__(self, ctx, public_key, seed):
        from .keys import Encryptor
        self.ctx, self.seed, self.next = ctx, seed, 0
        self.er = Encryptor(ctx, public_key, seed=seed)

    def seek(self, i):
        self.next = int(i)

    def encrypt(self, plain):
        from .keys import _Sampler
        self.er._smp = _Sampler(self.ctx, seed=[int(self.seed), self.next])
        self.next += 1
        return self.er.encrypt(plain)


def _sampler_key(seed):
    """32-byte sampler key of a reproducible test encryptor (never for data that needs protecting: seed=None draws the key from the OS)"""
    import hashlib
    return hashlib.sha256(b"fhe-hip server-side encryptions\0" + repr(seed).encode()).digest()


def make_fraction_encryptor(ctx, public_key, encoder=None, seed=None, indexed=False, device=None):
    """The circuit's server-side encryptions (homo/fhe_resize.h:230,234,262,266): a callable
    values -> [len(values), 2, k, n] of fresh encryptions of encode(v) under `public_key`
    ([2, k, n] device tensor).  seed=None draws from the OS CSPRNG.  indexed=True (tests; needs a seed): the callable has
    `seek(i)` and the i-th encryption depends on (seed, i) only, so sharded and whole-image runs agree bit for bit.

    device=True (the default without a seed): the whole batch is encoded and encrypted on the GPU -- keys.DeviceEncryptor,
    fhe_frac_encode_batch + fhe_encrypt_batch, five launches per call -- from a ChaCha20 stream keyed by the OS generator
    (or by the seed: then `seek(i)` exists whatever `indexed` says); encryption i of the callable's life uses stream i.  device=False:
    the host sampler (numpy), one ciphertext at a time (five launches and three uploads each: what rounds 2-4 did)."""
    from .evaluator import FractionalEncoder
    from .keys import DeviceEncryptor, Encryptor
    enc = encoder or FractionalEncoder(ctx)
    if device is None:
        device = seed is None
    if device:
        der = DeviceEncryptor(ctx, public_key, key=None if seed is None else _sampler_key(seed), int_coeffs=enc.int_coeffs, frac_coeffs=enc.frac_coeffs,
                              reproducible=seed is not None)

        def encrypt_batch(values):
            return der.encrypt_values([float(v) for v in values])
        if seed is not None:                           # a key from the OS generator only ever counts upwards: no (key, index) pair repeats
            encrypt_batch.seek = der.seek
        return encrypt_batch
    er = _IndexedEncryptions(ctx, public_key, seed) if indexed else Encryptor(ctx, public_key, seed=seed)

    def encrypt(values):
        return torch.stack([er.encrypt(enc.encode(float(v))) for v in values])
    if indexed:
        encrypt.seek = er.seek
    return encrypt


def _row_windows(H, h, init_rows):
    """Per destination row: v (float32, homo/fhe_resize.h:351) and the first source row of the
    reference's sliding window (:352; the window never moves backwards)."""
    f32 = np.float32
    start, out = 0, []
    for y in range(h):
        v = f32(f32(y) / f32(h - 1) * f32(H)) - f32(0.5)
        new_start = min(int(v) - init_rows // 2 + 1, H - init_rows)
        if new_start > start:
            start = new_start
        out.append((v, start))
    return out


def server_resize(ctx, in_path, out_path, src_w, src_h, dst_w, dst_h, bicubic, encrypt_fractions, rows_per_step=4, io_threads=8, slots=3, stats=None, rows=None, validate=True,
                  relin=None, shared_offsets=False):
    """homo/server_resize.cpp:127-146 + ResizeImage (homo/fhe_resize.h:308-392) on the GPU.

    Input stream: src_w * src_h pixels, row by row, three ciphertext records (R, G, B) per pixel
    (homo/client_resize.cpp:141-150).  Output stream: dst_w * dst_h pixels, row by row, three
    records of size 4 (bilinear) or 6 (bicubic) per pixel (homo/server_resize.cpp:141-146).
    in_path / out_path: file names or open StreamFile objects.

    The reference keeps a sliding window of init_rows = 2 / 4 source rows resident (`for memory
    reasons`, :324-379), walks the destination rows serially and samples one pixel at a time
    (SampleLinear / SampleBicubic, :381-388).  Here the same window logic decides which source rows
    are resident in HBM -- a ring of row slots; rows are read from the file exactly once, in order, rows no
    destination row needs are skipped like :353-357 -- and a step takes up to `rows_per_step` destination rows,
    samples ALL their pixels per channel as one batch through the library's circuits (fhe_sample_bicubic /
    fhe_sample_linear; the taps index the interleaved R, G, B records of the ring directly, nothing is
    gathered), while a reader thread brings in the next step's rows (file -> page-locked -> HBM on its own
    stream) and a writer thread drains the previous step's results (HBM -> page-locked on its own stream -> file).

    encrypt_fractions(values) -> [len, 2, k, n] supplies the circuit's server-side encryptions in
    the reference's call order (per destination pixel: frac(x), then frac(y)).

    rows=(y0, y1) processes a SHARD of the destination rows -- the multi-GPU partition of the reference's outer loop
    (:350; parallel.row_range): this process reads only the source rows those destination rows need (its rows +- the
    sampler's halo, the same sliding window started at its first row), and writes its band at the band's own position
    of the output stream, so R processes with disjoint row ranges fill one output file (or R files) without any exchange.
    If encrypt_fractions has a `seek(i)` attribute it is called with the position of the next encryption in the
    reference's whole-image sequence (2 * pixel index) before every step -- reproducible test encryptors use it so that
    any partition produces the same bytes; a randomised encryptor needs none.  Returns the number of pixels produced.

    shared_offsets=True (bicubic only; NOT the reference's ciphertexts, the same decrypted image): frac(x) depends on the output column
    only and frac(y) on the output row only (:351,382), and they are public values the server encrypts itself -- so the server draws ONE
    ciphertext per output column (once per job) and ONE per output row instead of two per output pixel, and a step runs the
    shared-offset circuit (circuits.resize_bicubic_shared / fhe_resize_bicubic_shared_rows: repeated row Cubics, squares and prepared
    operands formed once; 156 ms per channel against 282 ms at 128x128 -> 64x64, n = 8192) on each channel's resident rows.  Encryption
    order for seekable encryptors: the dst_w column offsets first, then row y at position dst_w + y."""
    import queue
    import threading
    import time
    from . import circuits
    ev = Evaluator(ctx)
    pc = circuits.PlainCache(ctx)
    residues = _ResidueCheck(ctx, validate)
    init_rows = 4 if bicubic else 2
    # relin=(evk_ntt, dbc): the RELINEARISED mode (circuits.py; every product of Cubic / Linear relinearised) -- records of size 2
    out_size = 2 if relin is not None else (6 if bicubic else 4)
    if dst_w < 2 or dst_h < 2 or src_h < init_rows or src_w < 1:
        raise ValueError("image too small for the sampler")
    if shared_offsets and not bicubic:
        raise ValueError("shared offsets exist for the bicubic sampler only (fhe_resize_bicubic_shared)")
    rec_in = RECORD_HEADER + 2 * ctx.k * ctx.n * 8
    rec_out = RECORD_HEADER + out_size * ctx.k * ctx.n * 8
    own_in, own_out = not isinstance(in_path, StreamFile), not isinstance(out_path, StreamFile)
    if (in_path.size if not own_in else os.path.getsize(in_path)) < src_w * src_h * 3 * rec_in:
        raise EOFError("ciphertext stream ended")
    row0, row1 = (0, dst_h) if rows is None else (int(rows[0]), int(rows[1]))
    if not (0 <= row0 < row1 <= dst_h):
        raise ValueError("rows %r are not a range of the %d destination rows" % (rows, dst_h))
    windows = _row_windows(src_h, dst_h, init_rows)
    f32 = np.float32
    us = [f32(f32(x) / f32(dst_w - 1) * f32(src_w)) - f32(0.5) for x in range(dst_w)]
    # steps: consecutive destination rows whose source rows fit `init_rows + rows_per_step` resident rows
    steps, y = [], row0
    while y < row1:
        e = y + 1
        while e < row1 and e - y < rows_per_step and windows[e][1] + init_rows - windows[y][1] <= init_rows + rows_per_step:
            e += 1
        steps.append((y, e))
        y = e
    span = [(windows[a][1], windows[b - 1][1] + init_rows) for a, b in steps]            # source rows [lo, hi) a step needs
    # rows each step has to bring in: those not yet read (rows are consumed from the stream in order, once; a shard starts at its own first row)
    reads, next_row = [], span[0][0]
    for lo, hi in span:
        first = max(next_row, lo)
        reads.append((first, max(0, hi - first)))
        next_row = max(next_row, hi)
    max_rows = max(hi - lo for lo, hi in span)
    max_new = max(cnt for _, cnt in reads)
    max_px = max((b - a) * dst_w for a, b in steps)
    R = 2 * max_rows + max_new + 1                                                       # ring of resident source rows: row r lives in slot r % R
    ring = torch.empty((R, src_w, 3, 2, ctx.k, ctx.n), dtype=torch.int64, device=ctx.device)
    ring_flat = ring.view(-1, 2, ctx.k, ctx.n)                                           # record (slot, x, channel) = pixel index (slot * src_w + x) * 3 + channel
    slots = max(2, slots)
    hin = [_pinned(("rs_in", i), (max_new, src_w, 3, 2, ctx.k, ctx.n)) for i in range(slots)]
    hout = [_pinned(("rs_out", i, out_size), (max_px, 3, out_size, ctx.k, ctx.n)) for i in range(slots)]
    dout = [torch.empty((max_px, 3, out_size, ctx.k, ctx.n), dtype=torch.int64, device=ctx.device) for _ in range(2)]
    main = torch.cuda.current_stream()
    h2d, d2h = _copy_streams()
    fin = StreamFile(in_path) if own_in else in_path
    fout = StreamFile(out_path, write=True, size=dst_w * dst_h * 3 * rec_out) if own_out else out_path
    if fout.size < dst_w * dst_h * 3 * rec_out:
        raise ValueError("output stream file is smaller than the result")
    free_in, ready_in, free_out, to_write = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
    for i in range(slots):
        free_in.put((i, None))
        free_out.put(i)
    errors = []
    io_seconds = {"read": 0.0, "write": 0.0}

    def reader():
        try:
            for si, (first, cnt) in enumerate(reads):
                item = free_in.get()
                if item is None:
                    return
                slot, copied = item
                if copied is not None:
                    copied.synchronize()
                if cnt:
                    t_io = time.perf_counter()
                    fin.transfer(first * src_w * 3, cnt * src_w * 3, 2, ctx, hin[slot], io_threads)
                    io_seconds["read"] += time.perf_counter() - t_io
                ready_in.put((si, slot))
        except BaseException as exc:
            errors.append(exc)
            ready_in.put((None, None))

    def writer():
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                first_px, npx, slot, landed = item
                landed.synchronize()
                t_io = time.perf_counter()
                fout.transfer(first_px * 3, npx * 3, out_size, ctx, hout[slot], io_threads)
                io_seconds["write"] += time.perf_counter() - t_io
                free_out.put(slot)
        except BaseException as exc:
            errors.append(exc)
            free_out.put(None)

    t0 = time.perf_counter()
    rt, wt = threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)
    try:
        rt.start()
        wt.start()
        computed, drained = [], [None, None]
        t_start, t_stop = [], []
        sampler = circuits.sample_bicubic if bicubic else circuits.sample_linear
        offs = ([(dx, dy) for dy in (-1, 0, 1, 2) for dx in (-1, 0, 1, 2)] if bicubic else [(0, 0), (1, 0), (0, 1), (1, 1)])
        copied_of = {}

        def upload(si):
            """rows of step si: page-locked slot -> ring, on the upload stream (blocks until the reader thread has them)"""
            got, slot = ready_in.get()
            if got is None:
                raise errors[0]
            first, cnt = reads[si]
            with torch.cuda.stream(h2d):
                if si >= 2:
                    h2d.wait_event(computed[si - 2])            # the slots these rows overwrite were last read two steps ago ...
                if si >= 1 and cnt and {(first + i) % R for i in range(cnt)} & {r % R for r in range(*span[si - 1])}:
                    h2d.wait_event(computed[si - 1])            # ... unless the window jumped (strong down-scaling): then wait for the previous step
                done_rows = 0
                while done_rows < cnt:                          # consecutive rows sit in consecutive slots modulo R
                    s0 = (first + done_rows) % R
                    part = min(cnt - done_rows, R - s0)
                    ring[s0:s0 + part].copy_(hin[slot][done_rows:done_rows + part], non_blocking=True)
                    residues.add(ring[s0:s0 + part])                 # on the copy stream, behind the copy
                    done_rows += part
                copied = torch.cuda.Event()
                copied.record(h2d)
            free_in.put((slot, copied))
            copied_of[si] = copied

        xcols = None
        if shared_offsets:                                      # one offset ciphertext per output column, once per job
            if hasattr(encrypt_fractions, "seek"):
                encrypt_fractions.seek(0)
            xcols = encrypt_fractions([float(u - f32(np.floor(u))) for u in us]).contiguous()
        upload(0)
        for si, (y0, y1) in enumerate(steps):
            main.wait_event(copied_of.pop(si))
            npx, d = (y1 - y0) * dst_w, si & 1
            if shared_offsets:
                if hasattr(encrypt_fractions, "seek"):
                    encrypt_fractions.seek(dst_w + y0)
                yrows = encrypt_fractions([float(windows[yy][0] - f32(np.floor(windows[yy][0]))) for yy in range(y0, y1)]).contiguous()
                s0, sc = circuits.resize_source_rows(src_h, dst_h, y0, y1)
                assert span[si][0] <= s0 and s0 + sc <= span[si][1], (span[si], s0, sc)          # the rows the taps touch are resident
                slots_of = torch.tensor([r % R for r in range(s0, s0 + sc)], dtype=torch.int64, device=ctx.device)

                def channel(ch):                                # this channel's resident rows, contiguous -> the shared-offset circuit on rows [y0, y1)
                    chan = ring[slots_of, :, ch].reshape(sc * src_w, 2, ctx.k, ctx.n)
                    return circuits.resize_bicubic_shared(ev, pc, chan, src_w, src_h, dst_w, dst_h, xcols, yrows, rows=(y0, y1), src_rows=(s0, sc), relin=relin)
            else:
                # sample plan of these destination rows in terms of ring slots
                taps, fracs = [], []
                for yy in range(y0, y1):
                    v = windows[yy][0]
                    yi = int(v)
                    for xx in range(dst_w):
                        u = us[xx]
                        xi = int(u)
                        taps.append([((min(max(yi + dy, 0), src_h - 1) % R) * src_w + min(max(xi + dx, 0), src_w - 1)) * 3 for dx, dy in offs])
                        fracs += [float(u - f32(np.floor(u))), float(v - f32(np.floor(v)))]
                taps = np.asarray(taps, dtype=np.uint32)
                if hasattr(encrypt_fractions, "seek"):
                    encrypt_fractions.seek(2 * y0 * dst_w)
                fr = encrypt_fractions(fracs)                                        # xfract, yfract per pixel, in order
                xf, yf = fr[0::2].contiguous(), fr[1::2].contiguous()

                def channel(ch):                                # the taps index the interleaved R, G, B records of the ring directly
                    return sampler(ev, pc, ring_flat, taps + ch, xf, yf, relin=relin)
            if drained[d] is not None:
                main.wait_event(drained[d])                                          # dout[d] has left for the host
            if stats is not None:
                t_start.append(torch.cuda.Event(enable_timing=True))
                t_start[-1].record(main)
            for ch in range(3):
                dout[d][:npx, ch].copy_(channel(ch))                                 # [npx, out_size, k, n] into the interleaved record order
            if stats is not None:
                t_stop.append(torch.cuda.Event(enable_timing=True))
                t_stop[-1].record(main)
            done = torch.cuda.Event()
            done.record(main)
            computed.append(done)
            # the NEXT step's upload is enqueued before THIS step's download: the runtime maps streams onto a few hardware queues and both
            # copy streams can land on one (measured: they did) -- a download waiting for this step's kernels at the head of that queue
            # held the next upload back until the kernels were done, and upload, compute and download took turns (2.05 s for a 0.9 s job)
            if si + 1 < len(steps):
                upload(si + 1)
            oslot = free_out.get()
            if oslot is None:
                raise errors[0]
            with torch.cuda.stream(d2h):
                d2h.wait_event(done)
                hout[oslot][:npx].copy_(dout[d][:npx], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(d2h)
            drained[d] = landed
            to_write.put((y0 * dst_w, npx, oslot, landed))
        to_write.put(None)
        wt.join()
        rt.join()
        if errors:
            raise errors[0]
        torch.cuda.synchronize()
        residues.verdict(_refuser(own_out, fout, out_path))
        if stats is not None:
            stats.update(seconds=time.perf_counter() - t0, device_compute_seconds=sum(a.elapsed_time(b) for a, b in zip(t_start, t_stop)) / 1e3,
                         bytes_in=sum(c for _, c in reads) * src_w * 3 * rec_in, bytes_out=dst_w * (row1 - row0) * 3 * rec_out, steps=len(steps),
                         file_read_seconds=io_seconds["read"], file_write_seconds=io_seconds["write"])
    finally:
        _stop_pipeline(rt, wt, free_in, to_write)
        if own_in:
            fin.close()
        if own_out:
            fout.close()
    return (row1 - row0) * dst_w


# ------------------------------------------------------------------------------------------------
# server_decode: the run-length decoder's driver loop (homo/server_decode.cpp:113-148) over a ciphertext stream
# ------------------------------------------------------------------------------------------------
def make_zero_encryptor(ctx, public_key, encoder=None, seed=None, indexed=False, device=None):
    """The decode path's server-side encryptions (homo/server_decode.cpp:121,126; homo/fhe_decode.h:54,134): a
    callable count -> [count, 2, k, n] of fresh encryptions of encode(0.0) under `public_key` ([2, k, n] device
    tensor).  seed=None draws from the OS CSPRNG.  indexed=True / device: as make_fraction_encryptor."""
    from .evaluator import FractionalEncoder
    from .keys import DeviceEncryptor, Encryptor
    if device is None:
        device = seed is None
    if device:
        der = DeviceEncryptor(ctx, public_key, key=None if seed is None else _sampler_key(seed), reproducible=seed is not None)

        def encrypt_batch(count):
            return der.encrypt_zeros(count)
        if seed is not None:
            encrypt_batch.seek = der.seek
        return encrypt_batch
    enc = encoder or FractionalEncoder(ctx)
    er = _IndexedEncryptions(ctx, public_key, seed) if indexed else Encryptor(ctx, public_key, seed=seed)
    zero = enc.encode(0.0)

    def encrypt(count):
        return torch.stack([er.encrypt(zero) for _ in range(count)]) if count else ctx.empty(0)
    if indexed:
        encrypt.seek = er.seek
    return encrypt


def server_decode(ctx, in_path, out_path, width, height, pairs, encrypt_zeros, order=64, degree=12, delta=0.5, shard=None, group=None, relin=None):
    """homo/server_decode.cpp:113-148 on the GPU, with the HOMOMORPHIC overload of approximated_step
    (homo/fhe_decode.h:202-242; the reference's main passes its debugging Decryptor and thereby selects the
    decrypting overload, :244-282, which needs the secret key on the server -- out of scope).

    Input stream: for each of the three colour channels, pairs[ch] runs of two ciphertext records
    (elem = the run's amplitude, count = its length; :131-132).  Output stream: for every position i < width * height
    the three channels' accumulators, interleaved (:139-143); a record has 22 polynomials (2 for a channel without
    runs).  width, height, pairs are the five integers of keys/params.txt (:15-24).

    Per channel the library evaluates the whole driver loop (fhe_decode_channel): the running `index`
    ciphertext (:121,137), one approximated_step per run over all positions and harmonics as batches, and the
    accumulation of its results into the channel (:134-136).

    encrypt_zeros(count) -> [count, 2, k, n] supplies the server-side encryptions of encode(0.0) in the reference's
    call order: per channel `index` (:121), the width * height accumulators (:126), then for every run the
    Enc(0) of homomorphic_sin and homomorphic_cos per (position, harmonic) (homo/fhe_decode.h:231-232).

    shard=(rank, world) processes this rank's contiguous range of the 3 * width * height (channel, position) units
    (parallel.decode_shards; SURVEY.md section 8(e): "decode shards by (run, output position)") and writes their records at
    their own positions of the output stream (every rank may open the same file: records have fixed offsets and a
    positional write past the end extends the file, so no ordering between the ranks is needed).  Replicated per rank: the input runs (small), each touched channel's
    `index` chain (pairs additions) and, inside the library, the offset chain and the sine polynomials.  The only
    exchange: the three `index` ciphertexts are server-side encryptions every shard of a channel must share, so they
    are drawn on rank 0 and broadcast over `group` (3 ciphertexts; parallel.broadcast_from_root).  If encrypt_zeros has a
    `seek(i)` attribute it is called with the position of the next encryption in the reference's whole-job sequence
    before every draw (reproducible test encryptors; any partition then writes the same bytes).  Returns width * height
    for a whole run (as before round 4), the number of (channel, position) units this shard produced otherwise."""
    from . import circuits, parallel
    ev = Evaluator(ctx)
    pc = circuits.PlainCache(ctx)
    npos = width * height
    pairs = [int(p) for p in pairs]
    if len(pairs) != 3 or min(pairs) < 0 or npos < 1:
        raise ValueError("pairs must hold three non-negative run counts and the image must not be empty")
    total = sum(pairs)
    rec_in = RECORD_HEADER + 2 * ctx.k * ctx.n * 8
    if os.path.getsize(in_path) < 2 * total * rec_in:
        raise EOFError("ciphertext stream ended")
    expect = (2, ctx.k, ctx.n)
    rank, world = (0, 1) if shard is None else (int(shard[0]), int(shard[1]))
    pieces = parallel.decode_shards(rank, world, npos)
    seek = getattr(encrypt_zeros, "seek", None)
    per_channel = [1 + npos + p * npos * degree * 2 for p in pairs]          # the reference's encryptions per channel, in order
    base = [sum(per_channel[:ch]) for ch in range(3)]

    def draw(at, count):
        if seek is not None:
            seek(at)
        return encrypt_zeros(count)

    # the three `index` ciphertexts (:121): one draw per channel on the root, shared by every shard of the channel
    index0 = None
    if world > 1:
        import torch.distributed as dist
        connected = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) == world
        # without a process group (shards run one after the other, or by unrelated processes) every shard draws the three
        # itself: identical with an indexed encryptor, independent (and equally valid) encryptions of 0 otherwise
        index0 = torch.cat([draw(base[ch], 1) for ch in range(3)]) if (rank == 0 or not connected) else ctx.empty(3)
        if connected:
            torch.cuda.synchronize()
            parallel.broadcast_from_root(index0, 0, group)
    dev = None
    if total and pieces:
        host = _pinned(("dec_in",), (total, 2, 2, ctx.k, ctx.n))
        arr = host.numpy().view(np.uint64)
        fin = os.open(in_path, os.O_RDONLY)
        try:
            _pread_records(fin, [arr[r, j] for r in range(total) for j in range(2)], 0, rec_in, expect)
        finally:
            os.close(fin)
        dev = host.to(ctx.device, non_blocking=True)
        residues = _ResidueCheck(ctx)
        residues.add(dev)
        residues.verdict()
    first_run = [sum(pairs[:ch]) for ch in range(3)]
    res = []
    for ch, p0, p1 in pieces:
        p, np_ = pairs[ch], p1 - p0
        if world == 1:                                       # the reference's order: index, accumulators, then the runs' zeros -- one draw
            z = draw(base[ch], per_channel[ch])
            index = z[0:1].clone()
            acc0 = z[1:1 + npos].contiguous()
            zeros = z[1 + npos:].reshape(p, npos, degree, 2, 2, ctx.k, ctx.n).contiguous() if p and degree else None
        else:
            index = index0[ch:ch + 1].clone()
            acc0 = draw(base[ch] + 1 + p0, np_)
            zeros = None
            if p and degree:                                 # per run, this shard's positions are one contiguous stretch of the sequence
                per_pos = degree * 2
                zeros = torch.stack([draw(base[ch] + 1 + npos + (r * npos + p0) * per_pos, np_ * per_pos) for r in range(p)])
                zeros = zeros.reshape(p, np_, degree, 2, 2, ctx.k, ctx.n).contiguous()
        runs = dev[first_run[ch]:first_run[ch] + p].contiguous() if p else None
        res.append(circuits.decode_channel(ev, pc, runs, index, acc0, zeros, order, degree, delta, width, height, positions=(p0, p1), relin=relin))
    torch.cuda.synchronize()
    # output stream: position-major, the three channels interleaved; record sizes follow from `pairs`, so every record has a fixed offset
    # relin=(evk_ntt, dbc): the relinearised mode (circuits.py) -- every record then has two polynomials
    so = [(2 if relin is not None else int(_lib_out_size(degree))) if pairs[ch] else 2 for ch in range(3)]
    rec = [RECORD_HEADER + so[ch] * ctx.k * ctx.n * 8 for ch in range(3)]
    stride = sum(rec)
    fd = os.open(out_path, os.O_RDWR | os.O_CREAT, 0o644)
    try:
        if os.fstat(fd).st_size != npos * stride and (world == 1 or rank == 0):
            os.ftruncate(fd, npos * stride)
        for (ch, p0, p1), r in zip(pieces, res):
            host_r = to_host_array(r)
            hdr = HEADER.pack(MAGIC, so[ch], ctx.k, ctx.n, 0)
            for i in range(p0, p1):
                payload = np.ascontiguousarray(host_r[i - p0], dtype="<u8")
                if os.pwritev(fd, [hdr, memoryview(payload).cast("B")], i * stride + sum(rec[:ch])) != rec[ch]:
                    raise IOError("short write on the ciphertext stream")
    finally:
        os.close(fd)
    return npos if shard is None else sum(p1 - p0 for _, p0, p1 in pieces)


def _lib_out_size(degree):
    from . import _lib
    return _lib.load().fhe_approximated_step_out_size(degree)


def to_host_array(t):
    return t.detach().cpu().contiguous().numpy().view(np.uint64)
