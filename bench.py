#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X: seekable-zstd frame decode (GiB/s of decompressed data).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2]

A step = one pass of the decode hot path over one batch of synthetic frames that are already resident in
HBM (compressed payload + seek-table offsets in, decompressed bytes out, Content_Checksums verified).
Workloads (BASELINE.json configs, generator of SURVEY.md 8d):
  c3 (default)  configs[2]: 4 GiB per GPU = 2048 frames x 2 MiB, level 1, XXH64 checksums on  -- the
                configuration the north-star target is quoted on; weak scaling: every rank owns its own
                2048 frames (configs[4] = 8 x this, frames sharded across ranks, no data-path collective)
  c2            configs[1]: 256 MiB = 128 frames x 2 MiB, level 1, decode-only
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL for the barrier /
max-over-ranks only).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FRAME = 2 << 20
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


_SHM = None


def _make_part(job):
    """Worker (forked, CPU only): generate chunks [k0+i0, k0+i0+n) into the shared buffer; optionally compress
    them with the reference Encoder loop over the box's libzstd."""
    k0, i0, n, level, cks, want_archive = job
    from oracle import zko, libzstd_ref as Z
    data = zko.gen_chunks(n * FRAME, k0 + i0)
    _SHM[i0 * FRAME:(i0 + n) * FRAME] = np.frombuffer(data, np.uint8)
    hashes = [zko.xxh64(data[i * FRAME:(i + 1) * FRAME]) for i in range(n)]
    if not want_archive:
        return b"", [], hashes
    if Z.load("system") is None:
        raise RuntimeError("no libzstd on this box to prepare the archive")
    comp, frames = Z.encode_seekable_frames(data, FRAME, level, cks, "system")
    return comp, frames, hashes


def build_inputs(k0, nframes, level, cks, workers, want_archive, tag, cache=None):
    """Returns (data as a uint8 array in an anonymous shared mapping, libzstd payload, frames, per-frame XXH64).
    cache: a directory that keeps them between runs of the same configuration (profiling passes: the generator and the CPU
    compression of 4 GiB otherwise run once per counter pass)."""
    global _SHM
    import mmap
    key = None
    if cache:
        os.makedirs(cache, exist_ok=True)
        key = os.path.join(cache, f"in_{k0}_{nframes}_{level}_{int(cks)}_{int(want_archive)}")
        if os.path.exists(key + ".ok"):
            data = np.fromfile(key + ".data", np.uint8)
            comp = open(key + ".comp", "rb").read()
            meta = np.load(key + ".npz")
            return data, comp, [tuple(int(x) for x in f) for f in meta["frames"]], [int(h) for h in meta["hashes"]]
    _map = mmap.mmap(-1, max(1, nframes * FRAME))      # MAP_SHARED | MAP_ANONYMOUS: the forked workers fill it; no tmpfs quota involved
    _SHM = np.frombuffer(_map, dtype=np.uint8)
    per = max(1, min(16, nframes // max(1, workers)))
    jobs = [(k0, i, min(per, nframes - i), level, cks, want_archive) for i in range(0, nframes, per)]
    if workers <= 1:                                   # --no-fork: rocprofv3's signal handler can hang on the pool's worker exits
        parts = [_make_part(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_make_part, jobs)
    comp = b"".join(p[0] for p in parts)
    frames = [f for p in parts for f in p[1]]
    hashes = [h for p in parts for h in p[2]]
    if key:
        _SHM.tofile(key + ".data")
        open(key + ".comp", "wb").write(comp)
        np.savez(key + ".npz", frames=np.array(frames, np.uint64).reshape(-1, 2), hashes=np.array(hashes, np.uint64))
        open(key + ".ok", "w").close()
    return _SHM, comp, frames, hashes


_ZKB_FOUND = {}


def _find_libzstd():
    """Every libzstd the box offers (ldconfig, the usual library directories), by version: the newest optimised build is the baseline --
    the reference pins 1.5.7, whose decoder is faster than 1.4.x's.  Pillow's bundled 1.5.7 (a size-optimised build, ~5x slower) is
    never timed.  Returns (path, {path: version})."""
    import glob
    import subprocess
    from oracle import libzstd_ref as Z
    paths = [p for p in Z._CANDIDATES["system"]]
    try:
        for ln in subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if "libzstd.so" in ln and "=>" in ln:
                paths.append(ln.split("=>")[1].strip())
    except (OSError, subprocess.SubprocessError):
        pass
    for d in ("/usr/lib", "/usr/lib64", "/usr/local/lib", "/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", os.path.expanduser("~/.local/lib")):
        paths += glob.glob(os.path.join(d, "libzstd.so*"))
    from oracle import zko
    seen, found, rate = set(), {}, {}
    sample = zko.gen_chunks(4 << 20, 3)
    for p in paths:
        rp = os.path.realpath(p)
        if rp in seen or not os.path.exists(rp) or "pillow" in rp:
            continue
        seen.add(rp)
        try:
            l = C.CDLL(rp)
            l.ZSTD_versionString.restype = C.c_char_p
            ver = l.ZSTD_versionString().decode()
            l.ZSTD_compressBound.restype = C.c_size_t; l.ZSTD_compressBound.argtypes = [C.c_size_t]
            l.ZSTD_compress.restype = C.c_size_t; l.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
            l.ZSTD_decompress.restype = C.c_size_t; l.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            cb = C.create_string_buffer(l.ZSTD_compressBound(len(sample)))
            cn = l.ZSTD_compress(cb, len(cb), sample, len(sample), 1)
            ob = C.create_string_buffer(len(sample))
            best = 1e9
            for _ in range(4):                                   # a build's decoder speed, not its version, picks the baseline (the fastest one)
                t = time.perf_counter()
                dn = l.ZSTD_decompress(ob, len(ob), cb, cn)
                best = min(best, time.perf_counter() - t)
            if dn != len(sample):
                continue
            found[rp] = f"{ver} ({len(sample) / best / 2**30:.2f} GiB/s on a 4 MiB one-shot decode)"
            rate[rp] = len(sample) / best
        except (OSError, AttributeError):
            continue
    if not found:
        return None, {}
    return max(rate, key=rate.get), found


def _zkb():
    from oracle import zko
    lib = zko.lib()
    path, found = _find_libzstd()
    _ZKB_FOUND.update(found)
    if path is None or lib.zkb_open(path.encode()) != 0:
        return None
    lib.zkb_version.restype = C.c_char_p
    lib.zkb_time_decode.restype = C.c_double
    lib.zkb_time_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    lib.zkb_time_encode.restype = C.c_double
    lib.zkb_time_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_int64)]
    return lib


def _cpu_decode_rate(lib, comp, frames, nthreads, target_seconds):
    """GiB/s of the reference Decoder loop: `nthreads` independent Decoders, one per contiguous frame range (what a user of
    the reference can do with set_lower_frame / set_upper_frame), wall clock over all of them."""
    from concurrent.futures import ThreadPoolExecutor
    cb = np.frombuffer(comp, np.uint8)
    n = len(frames)
    c = np.zeros(n + 1, np.int64); d = np.zeros(n + 1, np.int64)
    c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
    nthreads = max(1, min(nthreads, n))
    cuts = [n * k // nthreads for k in range(nthreads + 1)]

    def one(k, reps):
        lo, hi = cuts[k], cuts[k + 1]
        sink = C.c_uint64(0)
        return lib.zkb_time_decode(cb.ctypes.data + int(c[lo]), int(c[hi] - c[lo]), int(d[hi] - d[lo]), reps, C.byref(sink))

    def run(reps):
        t = time.perf_counter()
        if nthreads == 1:
            r = [one(0, reps)]
        else:
            with ThreadPoolExecutor(nthreads) as ex:
                r = list(ex.map(lambda k: one(k, reps), range(nthreads)))
        wall = time.perf_counter() - t
        if min(r) <= 0:
            return None
        return wall / reps
    t1 = run(1)
    if t1 is None:
        return None, 0
    reps = max(1, min(20, int(target_seconds / t1)))
    t2 = run(reps)
    best = min(t1, t2) if t2 else t1
    return int(d[-1]) / best / 2**30, reps + 1


def cpu_baseline(comp, frames, data, sample_frames, level, cks, ref_comp=None, ref_frames=None, ref3=None):
    """Reference CPU path (C restatement of zeekstd's Decoder / Encoder loops over the box's libzstd): 1 thread (the
    reference's own semantics, the north-star comparison) and all host cores (independent Decoders per frame range)."""
    lib = _zkb()
    if lib is None:
        return None
    ver = lib.zkb_version().decode()
    cores = os.cpu_count() or 1
    n = min(sample_frames, len(frames))
    csz = sum(f[0] for f in frames[:n])
    dsz = sum(f[1] for f in frames[:n])
    v1, passes = _cpu_decode_rate(lib, comp[:csz], frames[:n], 1, 6.0)
    if v1 is None:
        return None
    pin_note = ("the reference pins libzstd 1.5.7 (Cargo.lock:1192-1193); " +
                ("this box's libzstd is the pinned version" if ver.startswith("1.5.7") else
                 f"this box's optimised libzstd is {ver}, whose decoder is slower than 1.5.7's by an unmeasured margin -- the speed-ups beside this value "
                 f"are optimistic by that margin (the image's only 1.5.7 is pillow's bundled build, ~5x slower than a distro build: not used)"))
    out = {"value": v1, "unit": "GiB/s", "cores": 1, "kind": f"port (zeekstd loops in C over dlopen'd libzstd {ver}); {pin_note}",
           "libzstd_version": ver, "reference_pin": "1.5.7", "libzstd_on_this_box": dict(_ZKB_FOUND),
           "sample": f"zeekstd::Decoder loop (decode.rs:201-270, bench protocol decompress.rs:18-39), first {n} frames ({dsz >> 20} MiB) "
                     f"of the same (GPU-made) archive, best of {passes} passes, 1 thread"}
    va, passes = _cpu_decode_rate(lib, comp, frames, cores, 4.0)
    if va:
        out["all_cores"] = {"value": round(va, 2), "unit": "GiB/s", "cores": cores,
                            "sample": f"{cores} threads (nproc), one independent Decoder per contiguous frame range over all {len(frames)} frames, "
                                      f"wall clock, best of {passes} passes"}
    if ref_comp is not None:
        rn = min(sample_frames, len(ref_frames))
        rc = sum(f[0] for f in ref_frames[:rn])
        vr, passes = _cpu_decode_rate(lib, ref_comp[:rc], ref_frames[:rn], 1, 4.0)
        if vr:
            out["on_reference_made_archive"] = {"value": round(vr, 3), "unit": "GiB/s", "cores": 1,
                                                "sample": f"same loop on the archive the reference Encoder loop wrote (libzstd {ver}, level {level}), first {rn} frames"}
            vra, passes = _cpu_decode_rate(lib, ref_comp, ref_frames, cores, 3.0)
            if vra:
                out["on_reference_made_archive"]["all_cores"] = {"value": round(vra, 2), "unit": "GiB/s", "cores": cores,
                                                                 "sample": f"{cores} threads, one Decoder per frame range, all {len(ref_frames)} frames, best of {passes} passes"}
        if ref3 is not None:
            r3c, r3f = ref3
            rn = min(sample_frames, len(r3f))
            rc = sum(f[0] for f in r3f[:rn])
            v3, passes = _cpu_decode_rate(lib, r3c[:rc], r3f[:rn], 1, 3.0)
            v3a, passes_a = _cpu_decode_rate(lib, r3c, r3f, cores, 3.0)
            if v3:
                out["on_reference_made_archive"]["level_3"] = {
                    "value": round(v3, 3), "unit": "GiB/s", "cores": 1, "sample": f"the same at level 3 (the reference CLI's default), first {rn} frames",
                    "all_cores": ({"value": round(v3a, 2), "unit": "GiB/s", "cores": cores, "sample": f"{cores} threads, all {len(r3f)} frames, best of {passes_a} passes"}
                                  if v3a else None)}
    # encode side, for the record: zeekstd::Encoder loop at the same level on 128 frames
    ne = min(128, len(frames))
    src = np.ascontiguousarray(data[:ne * FRAME])
    dst = np.empty(src.size + (src.size >> 6) + 65536, np.uint8)
    csize = C.c_int64()
    te = lib.zkb_time_encode(src.ctypes.data, src.size, FRAME, level, int(cks), 2, dst.ctypes.data, dst.size, C.byref(csize))
    if te > 0:
        out["encode"] = {"value": src.size / te / 2**30, "unit": "GiB/s", "ratio": round(src.size / csize.value, 3),
                         "sample": f"zeekstd::Encoder loop (encode.rs:311-354,438-472) level {level}, {ne} frames, best of 2, 1 thread"}
    return out


# the sources a kernel's code comes from: the counter passes under profiles/ are stamped with their hashes (tools/pmc_summary.py), and a
# `traffic` figure is only quoted while the kernel it belongs to is still the code it was measured on
_KERNEL_SOURCES = {"dec": ("zk_decode.hip", "zk_device.h"), "enc": ("zk_encode.hip", "zk_enc_match.h", "zk_enc_match2.h", "zk_enc_device.h", "zk_enc_plan.h")}


def kernel_source_hashes():
    import hashlib
    out = {}
    for fam, names in _KERNEL_SOURCES.items():
        h = hashlib.sha1()
        for n in names:
            with open(os.path.join(ROOT, "zeekstd_amd", "csrc", n), "rb") as f:
                h.update(f.read())
        out[fam] = h.hexdigest()[:16]
    return out


def _pmc_fresh(pmc, kernel, section=None):
    """is the committed counter pass still about today's code of `kernel`?  (passes collected before round 6 carry no hashes: stale)"""
    stamped = (pmc.get("section_source_hashes", {}).get(section) if section and section != "kernels" else None) or pmc.get("source_hashes")
    if not stamped:
        return False
    fam = "enc" if kernel.startswith("zk_k_enc") else "dec"
    try:
        return stamped.get(fam) == kernel_source_hashes()[fam]
    except OSError:
        return False


def _pmc(section):
    """HBM bytes per kernel from the committed counter passes (profiles/pmc_traffic.json), or {}: constants of an earlier run of this
    very command, labelled as such wherever they are quoted"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
    except OSError:
        return {}, None, {}
    sec = pmc.get(section) if section != "kernels" else pmc.get("kernels")
    if not isinstance(sec, dict):
        return {}, None, pmc
    return sec, f"profiles/pmc_traffic.json [{section}] ({pmc.get('collected', 'committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}), not this run", pmc


def roofline_of(kernel_ms, algo_bytes, step_ms=None, pmc_section=None, full_size=True):
    """The roofline object of one leg: its slowest kernel against the HBM peak on the leg's ALGORITHMIC bytes (c_i + d_i per
    frame, summed over the launch), measured with HIP events on the launch stream; `traffic` from the committed counter passes."""
    if not kernel_ms:
        return None
    dom = max(kernel_ms, key=kernel_ms.get)
    ach = algo_bytes / (kernel_ms[dom] * 1e-3) / 1e9
    traffic, src = None, None
    if pmc_section and full_size:
        sec, src_, pmc = _pmc(pmc_section)
        key = dom.split("(")[0]
        traffic = sec.get(key, {}).get("hbm_bytes")
        if traffic is None and key in ("zk_k_fse", "zk_k_xxh64"):      # the engine's timer covers a family of kernels (zk_k_fse_quad / _sets / _predef_fed; zk_k_xxh64_wide ...): their sum
            fam = [v.get("hbm_bytes", 0) for k, v in sec.items() if k.startswith(key)]
            traffic = sum(fam) if fam else None
        src = src_ if traffic is not None else None
        if traffic is not None and not _pmc_fresh(pmc, key, pmc_section):
            traffic, src = None, "the committed counter pass was taken on other kernel sources (profiles/pmc_traffic.json source_hashes): not quoted"
    out = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
           "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": int(algo_bytes)}
    if step_ms:
        out["step_frac"] = round(algo_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    return out


def _stats_ms(ts):
    a = sorted(ts)
    return {"n": len(a), "median": round(a[len(a) // 2] * 1e3, 3), "min": round(a[0] * 1e3, 3), "mean": round(sum(a) / len(a) * 1e3, 3)}


def end_to_end(eng, zk, data, nframes, cks, reps=3):
    """Host buffers in, host buffers out, through the zeekstd API of the C ABI: zk_encoder_compress + zk_encoder_finish into a
    buffer writer (Encoder<Vec<u8>> of lib/benches/compress.rs:47-51), zk_decoder_decompress into caller memory
    (lib/benches/decompress.rs:18-25).  PCIe both ways is inside the timed span.  Bit-exact or it raises."""
    from zeekstd_amd import api
    lib = zk.lib
    n = nframes * FRAME
    src = np.ascontiguousarray(data[:n])

    class BW(C.Structure):
        _fields_ = [("data", C.c_void_p), ("cap", C.c_uint64), ("len", C.c_uint64), ("engine", C.c_void_p)]
    cap = int(lib.zk_compress_bound(n, FRAME)) + nframes * 8 + 64
    out = np.empty(cap, np.uint8); out[:] = 0
    dec = np.empty(n, np.uint8); dec[:] = 0
    wfn = C.cast(lib.zk_buffer_writer_write, C.c_void_p)
    lib.zk_encoder_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.zk_encoder_compress.restype = C.c_int64
    lib.zk_encoder_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.zk_encoder_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.zk_encoder_free.argtypes = [C.c_void_p]
    lib.zk_decoder_open_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.zk_decoder_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.zk_decoder_decompress.restype = C.c_int64
    lib.zk_decoder_free.argtypes = [C.c_void_p]
    te, td, total = [], [], 0
    for _ in range(reps):
        bw = BW(out.ctypes.data, cap, 0, eng._h)
        o = api.zk_encode_opts(0, FRAME, 1, int(cks), 0)
        h = C.c_void_p()
        if lib.zk_encoder_new(eng._h, C.byref(o), wfn, C.byref(bw), C.byref(h)) != 0:
            raise RuntimeError("zk_encoder_new failed")
        t = time.perf_counter()
        r = lib.zk_encoder_compress(h, src.ctypes.data, n)
        tot = C.c_uint64()
        rc = lib.zk_encoder_finish(h, 1, C.byref(tot))
        te.append(time.perf_counter() - t)
        lib.zk_encoder_free(h)
        if r != n or rc != 0 or tot.value != bw.len:
            raise RuntimeError(f"end-to-end encode failed ({r}, {rc})")
        total = tot.value
    for _ in range(reps):
        o = api.zk_decode_opts()
        h = C.c_void_p()
        if lib.zk_decoder_open_bytes(eng._h, out.ctypes.data, total, C.byref(o), C.byref(h)) != 0:
            raise RuntimeError("zk_decoder_open_bytes failed")
        t = time.perf_counter()
        got = 0
        while got < n:
            r = lib.zk_decoder_decompress(h, dec.ctypes.data + got, n - got)
            if r <= 0:
                break
            got += r
        td.append(time.perf_counter() - t)
        lib.zk_decoder_free(h)
        if got != n:
            raise RuntimeError(f"end-to-end decode stopped at {got}")
    if not np.array_equal(dec, src):
        raise RuntimeError("end-to-end round trip differs from the input bytes")
    return {"decode": {"value": round(n / min(td) / 2**30, 2), "unit": "GiB/s", "ms": round(min(td) * 1e3, 1)},
            "encode": {"value": round(n / min(te) / 2**30, 2), "unit": "GiB/s", "ms": round(min(te) * 1e3, 1)},
            "archive_bytes": int(total), "reps": reps, "bit_exact": True,
            "note": "host buffers in, host buffers out (pageable numpy arrays) through zk_encoder_compress/zk_encoder_finish and "
                    "zk_decoder_decompress: the engine pins the caller's pages on the fly and overlaps PCIe with the kernels; "
                    "PCIe Gen5 x16 moves 4 GiB in ~75 ms, which bounds both directions"}


def small_input_leg(eng, zk):
    """configs[0] (BASELINE.json): the reference's own bench input -- assets/dickens.txt is absent, SURVEY 8d's stand-in is
    gen(10 192 446 bytes) = 5 frames at the default policy -- through the RawEncoder, Encoder<Vec<u8>> and Decoder handles with
    the reference's bench protocols (lib/benches/compress.rs:8-62: level 1, 131 591-byte output buffer, compress loop then
    end_frame loop / write_all + end_frame; lib/benches/decompress.rs:18-39: 131 072-byte buffer until 0, then reset), beside
    the same loops over the box's libzstd on one host thread.  A 10 MB input is latency-bound on a GPU: five frames, one engine
    call each for the raw encoder.  Bit-exact round trip or it raises."""
    from zeekstd_amd import EncodeOptions, DecodeOptions
    from oracle import zko
    n = 10192446
    data = zko.gen_chunks(n)
    reps = 5
    out = bytearray(131591)

    def raw_once():
        enc = EncodeOptions().engine(eng).compression_level(1).into_raw_encoder()
        t = time.perf_counter()
        pos = 0
        mv = memoryview(data)
        while pos < n:
            pos += enc.compress(mv[pos:pos + (4 << 20)], out).in_progress()      # (the binding copies its input: bounded slices)
        while enc.end_frame(out).data_left():
            pass
        return time.perf_counter() - t
    t_raw = min(raw_once() for _ in range(reps))

    class Sink:
        def __init__(self): self.parts = []
        def write(self, b): self.parts.append(bytes(b)); return len(b)
        def flush(self): pass
    t_enc, arch = [], None
    for _ in range(reps):
        sink = Sink()
        enc = EncodeOptions().engine(eng).compression_level(1).into_encoder(sink)
        t = time.perf_counter()
        enc.write_all(data)
        enc.end_frame()
        t_enc.append(time.perf_counter() - t)
        enc.finish()
        arch = b"".join(sink.parts)
    dec = DecodeOptions(arch).engine(eng).into_decoder()
    buf = bytearray(131072)
    t_dec, got = [], None
    for r in range(reps):
        parts = []
        t = time.perf_counter()
        while True:
            k = dec.decompress(buf)
            if k == 0:
                break
            if r == 0:
                parts.append(bytes(buf[:k]))
        t_dec.append(time.perf_counter() - t)
        dec.reset()
        if r == 0:
            got = b"".join(parts)
    if got != data:
        raise RuntimeError("configs[0] round trip differs from the input bytes")
    gib = n / 2**30
    res = {"input_bytes": n, "frames": dec.seek_table().num_frames(), "ratio": round(n / (len(arch) - 8 * 5 - 17), 3),
           "raw_encoder": {"value": round(gib / t_raw, 3), "unit": "GiB/s", "ms": round(t_raw * 1e3, 2)},
           "encoder": {"value": round(gib / min(t_enc), 3), "unit": "GiB/s", "ms": round(min(t_enc) * 1e3, 2)},
           "decoder": {"value": round(gib / min(t_dec), 3), "unit": "GiB/s", "ms": round(min(t_dec) * 1e3, 2)},
           "note": "host buffers through the Level-B handles (Python binding: its own copies are inside the spans); best of %d" % reps}
    lib = _zkb()
    if lib is not None:
        src = np.frombuffer(data, np.uint8)
        dst = np.empty(n + (n >> 6) + 65536, np.uint8)
        csize = C.c_int64()
        te = lib.zkb_time_encode(src.ctypes.data, n, FRAME, 1, 0, 3, dst.ctypes.data, dst.size, C.byref(csize))
        sink = C.c_uint64(0)
        td = lib.zkb_time_decode(dst.ctypes.data, csize.value, n, 3, C.byref(sink)) if te > 0 else -1
        if te > 0 and td > 0:
            res["cpu_1thread"] = {"encoder": round(gib / te, 3), "decoder": round(gib / td, 3), "unit": "GiB/s", "ratio": round(n / csize.value, 3),
                                  "kind": "zeekstd loops in C over the box's libzstd " + lib.zkb_version().decode()}
    return res


def _splitmix_seed(seed):
    """splitmix64(seed) once -> xorshift64* state."""
    M = (1 << 64) - 1
    x = (seed + 0x9E3779B97F4A7C15) & M
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return z or 0x9E3779B97F4A7C15


def seek_protocol(trials, total, seed=0x5EED0003):
    """BASELINE configs[3] / SURVEY 8d: xorshift64* seeded through splitmix64; off = next() % total, len = 1 + next() % 8192."""
    M = (1 << 64) - 1
    x = _splitmix_seed(seed)
    offs = np.zeros(trials, np.uint64)
    lens = np.zeros(trials, np.uint32)
    for i in range(trials):
        v = []
        for _ in range(2):
            x ^= x >> 12
            x ^= (x << 25) & M
            x ^= x >> 27
            v.append((x * 0x2545F4914F6CDD1D) & M)
        off = v[0] % total
        ln = 1 + v[1] % 8192
        offs[i] = off
        lens[i] = min(ln, total - off)
    return offs, lens


def _pct(ts, skip=0):
    t = np.asarray(ts[skip:], np.float64)
    return {"p50": round(float(np.percentile(t, 50)), 1), "p95": round(float(np.percentile(t, 95)), 1),
            "p99": round(float(np.percentile(t, 99)), 1), "mean": round(float(t.mean()), 1)}


def _z_part(job):
    i0, n, fsz, level, cks = job
    from oracle import libzstd_ref as Z
    return Z.encode_seekable_frames(bytes(_SHM[i0:i0 + n]), fsz, level, cks, "system")


def libzstd_archive_parallel(src, fsz, level, cks, workers):
    """The reference Encoder loop over the box's libzstd at frame size fsz, frames spread over forked workers."""
    global _SHM
    _SHM = src
    per = max(fsz, (src.size // max(1, workers * 4)) // fsz * fsz)
    jobs = [(i, min(per, src.size - i), fsz, level, cks) for i in range(0, src.size, per)]
    if workers <= 1:
        parts = [_z_part(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_z_part, jobs)
    return b"".join(p[0] for p in parts), [f for p in parts for f in p[1]]


def time_single_seeks(eng, zk, comp, frames, src, offs, lens):
    """One seek at a time through the zeekstd Decoder API of the C ABI (host buffers; zk_decoder_time_seeks), every read
    compared with the generator bytes; the reference's CPU loop on the same archive beside it."""
    from zeekstd_amd import api
    lib = zk.lib
    st = zk.SeekTable.new()
    for c_, d_ in frames:
        st.log_frame(c_, d_)
    seekable = comp + st.to_bytes()
    o = api.zk_decode_opts()
    h = C.c_void_p()
    lib.zk_decoder_open_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rc = lib.zk_decoder_open_bytes(eng._h, seekable, len(seekable), C.byref(o), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"zk_decoder_open_bytes: {rc}")
    n = len(offs)
    us = np.zeros(n, np.float64)
    buf = np.zeros(8192 + 64, np.uint8)
    lib.zk_decoder_time_seeks.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.zk_decoder_time_seeks.restype = C.c_int
    rc = lib.zk_decoder_time_seeks(h, offs.ctypes.data, lens.ctypes.data, n, buf.ctypes.data, buf.size, src.ctypes.data, us.ctypes.data)
    lib.zk_decoder_free.argtypes = [C.c_void_p]
    lib.zk_decoder_free(h)
    if rc != 0:
        raise RuntimeError(f"seek read failed or differs from the generator bytes (rc {rc})")
    out = {"gpu_decoder_us": _pct(us, min(50, n // 10))}
    from oracle import zko, libzstd_ref as Z
    zlib_ = zko.lib()
    path = next((p for p in Z._CANDIDATES["system"] if os.path.exists(p)), None)
    if path and zlib_.zkb_open(path.encode()) == 0:
        c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
        c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
        tus = np.zeros(n, np.float64)
        ob = np.zeros(8192 + 64, np.uint8)
        cb = np.frombuffer(comp, np.uint8)
        zlib_.zkb_time_seeks.restype = C.c_int
        zlib_.zkb_time_seeks.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 2 + [C.c_uint32] + [C.c_void_p] * 2
        if zlib_.zkb_time_seeks(cb.ctypes.data, c.ctypes.data, d.ctypes.data, len(frames), offs.ctypes.data, lens.ctypes.data,
                                n, tus.ctypes.data, ob.ctypes.data) == 0:
            out["cpu_reference_us"] = _pct(tus, min(50, n // 10))
    return out


def batched_seeks(eng, comp, frames, src, offs, B=1024, reps=7):
    """Batches of B seeks per submission against the archive resident in HBM (zk_decode_frame_list_dev)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
    c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
    d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    tb = []
    nb = max(1, min(reps, len(offs) // B))
    for r in range(nb):
        boffs = offs[r * B:(r + 1) * B]
        ids = (np.searchsorted(d, boffs, side="right") - 1).astype(np.uint32)
        sizes = (d[ids.astype(np.int64) + 1] - d[ids.astype(np.int64)]).astype(np.uint64)
        ooff = np.zeros(len(ids) + 1, np.uint64); ooff[1:] = np.cumsum(sizes)
        d_ids = torch.from_numpy(ids.view(np.int32)).to(dev); d_oo = torch.from_numpy(ooff.view(np.int64)).to(dev)
        d_o = torch.empty(int(ooff[-1]) + 64, dtype=torch.uint8, device=dev); d_s = torch.zeros(len(ids), dtype=torch.int32, device=dev)
        for k in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            rc = eng.decode_frame_list_dev(d_comp, len(comp), d_c, d_d, d_ids, d_oo, len(ids), d_o, int(ooff[-1]), True, d_s)
            torch.cuda.synchronize()
            if k:
                tb.append((time.perf_counter() - t) * 1e6)
        got = bytes(d_o[:int(ooff[3])].cpu().numpy())
        if rc != 0 or got != b"".join(src[int(d[i]):int(d[i + 1])].tobytes() for i in ids[:3]):
            raise RuntimeError("batched seek mismatch")
    return {"batch": B, "batches": nb, "batch_us_p50": round(float(np.median(tb)), 1), "us_per_seek": round(float(np.median(tb)) / B, 2)}


def seek_leg(eng, data, zk, z64, nbytes, trials, fsz=65536):
    """BASELINE configs[3] as written: nbytes of generator text in 64 KiB frames, `trials` random
    set_offset / set_offset_limit / read-to-exhaustion seeks (xorshift64* protocol), on BOTH the archive this engine's encoder
    writes and the archive the reference's CPU Encoder writes; p50 / p95 / p99 one at a time, and batches of 1024."""
    from oracle import libzstd_ref as Z
    src = np.ascontiguousarray(data[:nbytes])
    offs, lens = seek_protocol(trials, nbytes)
    out = {"config": f"configs[3]: {nbytes >> 20} MiB, {nbytes // fsz} x {fsz >> 10} KiB frames, {trials} seeks, xorshift64* seed 0x5EED0003",
           "frames": nbytes // fsz, "frame_size": fsz, "seeks": trials}
    comp, frames = eng.encode_frames(src, fsz, 1, True)
    g = time_single_seeks(eng, zk, comp, frames, src, offs, lens)
    g["batches_of_1024"] = batched_seeks(eng, comp, frames, src, offs)
    g["compressed_bytes"] = len(comp)
    out["gpu_made_archive"] = g
    if z64 is not None:
        comp, frames = z64
        r = time_single_seeks(eng, zk, comp, frames, src, offs, lens)
        r["batches_of_1024"] = batched_seeks(eng, comp, frames, src, offs)
        r["compressed_bytes"] = len(comp)
        out["reference_made_archive"] = r
    out["note"] = ("one seek = set_offset + set_offset_limit + decompress through the zeekstd Decoder API (host buffers, PCIe included): "
                   "one frame on the GPU is launch + chain latency bound; batches of seeks are what the GPU is for")
    return out


def dry_run(args, rank, world):
    """The N > 1 launch contract without a GPU (CI here: gloo): RANK / WORLD_SIZE / MASTER_* from the environment, the process
    group, barriers, max-over-ranks timing, the shard + gather leg (zeekstd_amd/parallel.py) and rank 0's single JSON line.
    The frames are stand-ins (one raw block each, written here): no codec runs and nothing is measured."""
    import torch
    import torch.distributed as dist
    from zeekstd_amd import parallel
    if world > 1:
        dist.init_process_group("gloo")
    nfr, fsz = 4, 1000

    class StoredFrames:                               # engine stand-in: Frame_Header + one Raw_Block per frame
        def encode_frames_dev(self, d_src, n, frame_size, level, checksum, d_comp, cap, d_cs, d_ds, stream=None):
            pos = 0
            nf = max(1, -(-n // frame_size))
            for f in range(nf):
                chunk = bytes(d_src[f * frame_size:min(n, (f + 1) * frame_size)].numpy())
                h = 1 | (len(chunk) << 3)
                fr = b"\x28\xb5\x2f\xfd\x00\x00" + bytes([h & 255, (h >> 8) & 255, (h >> 16) & 255]) + chunk
                d_comp[pos:pos + len(fr)] = torch.frombuffer(bytearray(fr), dtype=torch.uint8)
                d_cs[f] = len(fr); d_ds[f] = len(chunk)
                pos += len(fr)
            return nf, pos
    d_src = torch.full((nfr * fsz,), rank, dtype=torch.uint8)
    for _ in range(args.warmup):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gather = None
    if world > 1:
        dist.barrier(); tg = time.perf_counter()
        out, table = parallel.encode_sharded(StoredFrames(), d_src, fsz, 1, False, root=0)
        dist.barrier()
        gather = {"ms": round((time.perf_counter() - tg) * 1e3, 2), "frames_on_root": table.num_frames() if table is not None else None,
                  "stream_bytes_on_root": int(out.numel()) if out is not None else None}
    if rank == 0:
        print(json.dumps({"metric": "decode_decompressed_GiB_per_s", "value": 0.0, "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u8", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run (no GPU, nothing measured)", "frames_per_gpu": nfr, "frame_size": fsz},
                          "rccl_gather": gather}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2"])
    ap.add_argument("--frames", type=int, default=0, help="override frames per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seek", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (host buffers) leg")
    ap.add_argument("--no-c1", action="store_true", help="skip the configs[0] leg (the reference's own 10 MB bench input through the handles)")
    ap.add_argument("--seek-trials", type=int, default=10000)
    ap.add_argument("--no-fork", action="store_true", help="generate the inputs in this process (profiling runs)")
    ap.add_argument("--no-ref-archive", action="store_true", help="skip the secondary leg on a CPU-libzstd-made archive")
    ap.add_argument("--no-ref-level3", action="store_true", help="skip the level-3 part of that leg")
    ap.add_argument("--sync", action="store_true",
                    help="time one batch at a time (zk_decode_frames_dev) instead of two batches in flight; the "
                         "kernel-trace profile uses this so that kernel durations are not inflated by overlap")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run the launch contract (env, process group over gloo, shard + gather leg, the JSON line) on CPU "
                         "tensors with stored-block stand-in frames; measures nothing")
    ap.add_argument("--level", type=int, default=1, help="compression level of the archive (1 = BASELINE's configs; 3 = the reference CLI's default: profiling runs with --archive libzstd)")
    ap.add_argument("--cache", default=None, help="directory that keeps the generated input + the libzstd archive between runs (profiling passes)")
    ap.add_argument("--choice", action="append", default=[], metavar="KEY=VALUE",
                    help="pin a kernel variant for the whole run (zk_engine_set_kernel_choice: fse_own, fse_shared, exec_lanes, exec_ring, xxh64, "
                         "small_path, pipe_contexts, pipe_chunk_mib); A/B runs of tools/, never the driver's line")
    ap.add_argument("--one-gpu-transport", default=None, metavar="LIB",
                    help="TEST ONLY (tests/test_gpu_multirank_bench.py): every rank uses device 0, the process group runs over gloo, and the gather leg goes "
                         "through zk_gather_seekable with its collective entry points provided by LIB (tests/sim/libzk_shm_collectives.so) -- the N > 1 "
                         "control flow of this file (barriers, max over ranks, the gather's watchdog, the check of a remote frame, the line) on a "
                         "box with one GPU.  The numbers of such a run mean nothing")
    ap.add_argument("--gather-only", action="store_true", help="N > 1: one untimed-quality decode step, then the gather leg (what a scaling run is asked about first)")
    ap.add_argument("--archive", default="auto", choices=["auto", "gpu", "libzstd"],
                    help="who compresses the archive that is decoded: the GPU encoder (default for c3) or CPU libzstd (default for c2)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    nframes = args.frames or (2048 if args.workload == "c3" else 128)
    cks = args.workload == "c3"
    level = args.level
    one_gpu = bool(args.one_gpu_transport) and world > 1
    if one_gpu:
        local_rank = 0                                     # every rank on device 0
    if args.gather_only:
        args.steps, args.warmup = 1, 1
        args.no_cpu_baseline = args.no_seek = args.no_e2e = args.no_c1 = args.no_ref_archive = True
    if args.dry_run:
        return dry_run(args, rank, world)

    # ---- untimed setup on the host cores (before any HIP initialisation: workers are forked)
    cores = os.cpu_count() or 8
    workers = 1 if args.no_fork else max(1, min(64, cores // max(1, world) - 1))
    use_gpu_archive = args.archive == "gpu" or (args.archive == "auto" and args.workload == "c3")
    t0 = time.time()
    # a reference-made archive of the same input is prepared too (single-GPU runs): its decode rate is reported beside the
    # headline, because archives written by zeekstd's own CPU Encoder are what a drop-in user decodes first
    want_ref_archive = (not use_gpu_archive) or (world == 1 and not args.no_ref_archive)
    data, z_comp, z_frames, hashes = build_inputs(rank * nframes, nframes, level, cks, workers, want_ref_archive, rank, args.cache)
    do_seek = rank == 0 and world == 1 and not args.no_seek
    z64 = None
    if do_seek and want_ref_archive and z_comp:           # the reference-made 64 KiB-frame archive of configs[3] (forked workers: before HIP)
        z64 = libzstd_archive_parallel(np.asarray(data), 65536, level, True, workers)
    z3 = None
    if world == 1 and want_ref_archive and z_comp and use_gpu_archive and not args.no_ref_level3:
        z3 = libzstd_archive_parallel(np.asarray(data)[:nframes * FRAME], FRAME, 3, cks, workers)      # the reference CLI's default level (cli/src/args.rs:192)
    t_setup = time.time() - t0

    import torch
    import torch.distributed as dist
    import zeekstd_amd as zk
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    shm_comm = shm_lib = None
    if world > 1 and one_gpu:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        shm_lib = C.CDLL(args.one_gpu_transport)
        shm_lib.zkshm_comm_create.restype = C.c_void_p
        shm_lib.zkshm_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64]
        shm_lib.zkshm_comm_destroy.argtypes = [C.c_void_p]
        zk.lib.zk_set_collective_library.argtypes = [C.c_char_p]
        if zk.lib.zk_set_collective_library(args.one_gpu_transport.encode()) != 0:
            raise RuntimeError("the transport library could not be loaded")
        name = f"/zkbench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        shm_comm = shm_lib.zkshm_comm_create(name.encode(), rank, world, max(64 << 20, int(zk.lib.zk_compress_bound(nframes * FRAME, FRAME)) + (1 << 20)))
        if not shm_comm:
            raise RuntimeError("the shared-memory communicator could not be created")
    elif world > 1:
        import datetime
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))      # a hung collective ends the run in minutes, not in half an hour
    eng = zk.Engine(local_rank)
    for kv in args.choice:
        key, _, val = kv.partition("=")
        eng.set_kernel_choice(**{key: int(val)})
    dsize = nframes * FRAME
    d_src = torch.from_numpy(np.asarray(data)).to(dev)

    # ---- the archive: produced by the GPU encoder (configs[2]) or by the CPU reference path (configs[1])
    enc_info = None
    if use_gpu_archive:
        cap = int(zk.lib.zk_compress_bound(dsize, FRAME))
        d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
        d_cs = torch.zeros(nframes, dtype=torch.int32, device=dev)
        d_ds = torch.zeros(nframes, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()                                # torch's fills have landed before the engine's own queue writes the same buffers
        nf, csize = eng.encode_frames_dev(d_src, dsize, FRAME, level, cks, d_comp, cap, d_cs, d_ds)      # warm-up + the archive
        torch.cuda.synchronize()
        te = []
        for _ in range(max(3, args.steps // 2)):
            t = time.perf_counter()
            eng.encode_frames_dev(d_src, dsize, FRAME, level, cks, d_comp, cap, d_cs, d_ds)
            te.append(time.perf_counter() - t)
        eng.set_profiling(True)
        eng.encode_frames_dev(d_src, dsize, FRAME, level, cks, d_comp, cap, d_cs, d_ds)
        ek = eng.kernel_times()
        eng.set_profiling(False)
        cs = d_cs.cpu().numpy().astype(np.uint64)
        frames = [(int(c), FRAME) for c in cs]
        comp = None
        enc_info = {"value": round(dsize / min(te) / 2**30, 2), "unit": "GiB/s", "ratio": round(dsize / csize, 3),
                    "ms": round(min(te) * 1e3, 2), "kernel_ms": {k: round(v, 3) for k, v in ek.items()},
                    "calls_ms": _stats_ms(te),
                    "roofline": roofline_of(ek, dsize + csize, min(te) * 1e3, "kernels", nframes == 2048)}
        # (r6) the same input at the reference CLI's default level (cli/src/args.rs:192): dense far history (zk_k_enc_dense_cand); its first
        # eight frames decoded and compared with the input.  Not the headline: one line of the report.
        if rank == 0 and world == 1 and level == 1 and not args.no_e2e:
            try:
                d_comp3 = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
                d_cs3 = torch.zeros(nframes, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                _, csize3 = eng.encode_frames_dev(d_src, dsize, FRAME, 3, cks, d_comp3, cap, d_cs3, d_ds)
                t3 = []
                for _ in range(2):
                    t = time.perf_counter()
                    eng.encode_frames_dev(d_src, dsize, FRAME, 3, cks, d_comp3, cap, d_cs3, d_ds)
                    t3.append(time.perf_counter() - t)
                eng.set_profiling(True)
                eng.encode_frames_dev(d_src, dsize, FRAME, 3, cks, d_comp3, cap, d_cs3, d_ds)
                ek3 = eng.kernel_times()
                eng.set_profiling(False)
                nchk = min(8, nframes)
                c3 = np.zeros(nchk + 1, np.int64); c3[1:] = np.cumsum(d_cs3[:nchk].cpu().numpy().astype(np.int64))
                d3 = np.arange(nchk + 1, dtype=np.int64) * FRAME
                o3 = torch.empty(nchk * FRAME + 64, dtype=torch.uint8, device=dev)
                s3 = torch.zeros(nchk, dtype=torch.int32, device=dev)
                eng.decode_frames_dev(d_comp3, int(c3[-1]), torch.from_numpy(c3).to(dev), torch.from_numpy(d3).to(dev), 0, nchk, o3, nchk * FRAME, True, s3)
                ok3 = bool(torch.equal(o3[:nchk * FRAME], d_src[:nchk * FRAME])) and int(s3.abs().sum().item()) == 0
                enc_info["level_3"] = {"value": round(dsize / min(t3) / 2**30, 2), "unit": "GiB/s", "ratio": round(dsize / csize3, 3), "ms": round(min(t3) * 1e3, 2),
                                       "kernel_ms": {k: round(v, 3) for k, v in ek3.items() if "enc" in k}, "first_frames_round_trip": ok3,
                                       "roofline": roofline_of({k: v for k, v in ek3.items() if "enc" in k}, dsize + csize3, min(t3) * 1e3, None, False)}
                del d_comp3, d_cs3, o3
            except Exception as ex:                                   # noqa: BLE001  (a leg of the report, not the run)
                enc_info["level_3"] = {"error": f"{type(ex).__name__}: {ex}"}
    else:
        frames, comp = z_frames, z_comp
        d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
    c = np.zeros(nframes + 1, np.uint64)
    d = np.zeros(nframes + 1, np.uint64)
    c[1:] = np.cumsum([f[0] for f in frames])
    d[1:] = np.cumsum([f[1] for f in frames])
    csize = int(c[-1])
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if comp is None:
            comp = bytes(d_comp[:csize].cpu().numpy())             # CPU decodes the very same (GPU-made) archive
        base = cpu_baseline(comp, frames, data, 512 if args.workload == "c3" else 128, level, cks,
                            z_comp if (use_gpu_archive and z_comp) else None, z_frames, z3)
    d_c = torch.from_numpy(c.view(np.int64)).to(dev)
    d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_out = torch.empty(dsize + 64, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(nframes, dtype=torch.int32, device=dev)
    d_hash = torch.zeros(nframes, dtype=torch.int64, device=dev)

    def step():
        rc = eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nframes, d_out, dsize, True, d_st)
        if rc != 0:
            raise RuntimeError(f"decode failed: {zk.error_name(rc)}")

    def barrier():
        if world > 1:
            dist.barrier()

    torch.cuda.synchronize()                                    # (torch's fills and copies above have landed: the engine's queues are not ordered behind torch's stream)
    for _ in range(args.warmup):
        step()
    # parity gate before anything is timed: every frame's XXH64 equals the generator's
    eng.xxh64_frames_dev(d_out, d_d, nframes, d_hash)
    got = d_hash.cpu().numpy().view(np.uint64)
    if not np.array_equal(got, np.array(hashes, dtype=np.uint64)) or int(d_st.abs().sum().item()) != 0:
        raise RuntimeError("GPU decode is not bit-exact against the generator bytes")
    if not torch.equal(d_out[:dsize], d_src):
        raise RuntimeError("GPU decode differs from the input bytes")

    # Timed region: K steps, two batches in flight (zk_decode_submit_dev / zk_decode_wait: the checksum stage of one
    # batch -- a per-frame serial chain -- overlaps the entropy stages of the next).  Every step is a complete decode
    # + verification of the whole batch; both output buffers are compared with the input bytes afterwards.
    d_outs = [d_out, torch.empty(dsize + 64, dtype=torch.uint8, device=dev)]
    d_sts = [d_st, torch.zeros(nframes, dtype=torch.int32, device=dev)]

    def run_pipelined(k):
        pending = []
        for i in range(k):
            if len(pending) == 2:
                rc = eng.decode_wait(pending.pop(0))
                if rc != 0:
                    raise RuntimeError(f"decode failed: {zk.error_name(rc)}")
            pending.append(eng.decode_submit_dev(d_comp, csize, d_c, d_d, 0, nframes, d_outs[i & 1], dsize, True, d_sts[i & 1]))
        for sl in pending:
            rc = eng.decode_wait(sl)
            if rc != 0:
                raise RuntimeError(f"decode failed: {zk.error_name(rc)}")

    def run_sync(k):
        for _ in range(k):
            step()

    run_timed = run_sync if args.sync else run_pipelined
    if not args.sync:
        run_pipelined(2)                                       # sizes the second context's scratch (untimed)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_timed(args.steps)
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    if not args.sync and (not torch.equal(d_outs[1][:dsize], d_src) or int(d_sts[1].abs().sum().item()) != 0):
        raise RuntimeError("pipelined GPU decode differs from the input bytes")
    # the same K steps one batch at a time (zk_decode_frames_dev returns after each batch): reported beside the headline
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sync_steps = []
    for _ in range(args.steps):
        ts_ = time.perf_counter()
        step()                                                  # (returns when the batch is done: zk_decode_frames_dev synchronises its stream)
        sync_steps.append(time.perf_counter() - ts_)
    followed = eng.checksums_followed()                         # frames of the last lone batch whose checksums ran BESIDE the executor
    torch.cuda.synchronize()
    sync_elapsed = time.perf_counter() - t1
    # SURVEY 8d's protocol asks for median and min beside the mean: the headline's K pipelined steps are one span by contract, so two more
    # spans of K steps are timed the same way (rank-local, behind the contract's span) and every span's ms per step is listed
    spans = [elapsed]
    for _ in range(2):
        torch.cuda.synchronize(); tsp = time.perf_counter()
        run_timed(args.steps)
        torch.cuda.synchronize(); spans.append(time.perf_counter() - tsp)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_bytes = dsize * world * args.steps
    value = total_bytes / elapsed / 2**30

    # ---- secondary leg: the same steps on the archive the reference's CPU Encoder loop (libzstd) wrote for this input
    def ref_leg(rz_comp, rz_frames, note, pmc_section):
        r_c = np.zeros(nframes + 1, np.uint64); r_c[1:] = np.cumsum([f[0] for f in rz_frames])
        r_csize = int(r_c[-1])
        dr_comp = torch.from_numpy(np.frombuffer(rz_comp + b"\0" * 64, np.uint8).copy()).to(dev)
        dr_c = torch.from_numpy(r_c.view(np.int64)).to(dev)

        def ref_pipelined(k):
            pending = []
            for i in range(k):
                if len(pending) == 2:
                    if eng.decode_wait(pending.pop(0)) != 0:
                        raise RuntimeError("decode of the reference-made archive failed")
                pending.append(eng.decode_submit_dev(dr_comp, r_csize, dr_c, d_d, 0, nframes, d_outs[i & 1], dsize, True, d_sts[i & 1]))
            for sl in pending:
                if eng.decode_wait(sl) != 0:
                    raise RuntimeError("decode of the reference-made archive failed")

        ref_pipelined(2)
        if not torch.equal(d_outs[1][:dsize], d_src):
            raise RuntimeError("decode of the reference-made archive differs from the input bytes")
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ref_pipelined(args.steps)
        torch.cuda.synchronize()
        r_el = time.perf_counter() - t2
        eng.set_profiling(True)
        eng.decode_frames_dev(dr_comp, r_csize, dr_c, d_d, 0, nframes, d_out, dsize, True, d_st)
        r_k = eng.kernel_times()
        eng.set_profiling(False)
        return {"value": round(dsize * args.steps / r_el / 2**30, 3), "unit": "GiB/s", "ms_per_step": round(r_el / args.steps * 1e3, 3),
                "compressed_bytes": r_csize, "kernel_ms": {k: round(v, 3) for k, v in r_k.items()},
                "roofline": roofline_of(r_k, r_csize + dsize, r_el / args.steps * 1e3, pmc_section, nframes == 2048), "note": note}

    ref_info = None
    if use_gpu_archive and want_ref_archive and z_comp:
        ref_info = ref_leg(z_comp, z_frames, "archive written by the reference Encoder loop over the box's libzstd (level 1: 128 KiB blocks with "
                           "their own FSE tables -> zk_k_fse_quad instead of zk_k_fse_predef); bit-exact, checksums verified", "reference_made_level_1")
        if z3:
            ref_info["level_3"] = ref_leg(z3[0], z3[1], "the same at level 3, the reference CLI's default: ~60 % more sequences per frame, a 2 MiB window", "reference_made_level_3")

    # ---- roofline of the dominant kernel: HIP events on the launch stream, live
    eng.set_profiling(True)
    acc = {}
    nprof = 3
    for _ in range(nprof):
        step()
        for k, ms in eng.kernel_times().items():
            acc[k] = acc.get(k, 0.0) + ms / nprof
    eng.set_profiling(False)
    dom = max(acc, key=acc.get)
    algo_bytes = csize + dsize                    # decode = c_i + d_i per frame, summed over the launch
    achieved = algo_bytes / (acc[dom] * 1e-3) / 1e9
    # HBM bytes actually moved by that kernel: PMC counters cannot be read from inside this process; the
    # value comes from the committed separate rocprofv3 --pmc passes of this very command (profiles/README.md)
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("workload") == args.workload and nframes == 2048:
            traffic = pmc["kernels"].get(dom.split("(")[0], {}).get("hbm_bytes")
            # NOT measured by this run: a constant from the committed counter passes of this command
            traffic_src = f"profiles/pmc_traffic.json ({pmc.get('collected', 'committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}), not this run"
            if traffic is not None and not _pmc_fresh(pmc, dom.split("(")[0]):
                traffic, traffic_src = None, "the committed counter pass was taken on other kernel sources (profiles/pmc_traffic.json source_hashes): not quoted"
    except OSError:
        pass

    # ---- end to end: host buffers in, host buffers out through the zeekstd Encoder / Decoder API (PCIe inside the span)
    e2e_info = None
    if rank == 0 and world == 1 and not args.no_e2e:
        del d_outs, d_sts
        e2e_info = end_to_end(eng, zk, np.asarray(data), nframes, cks)

    # ---- configs[0]: the reference's own bench input through the RawEncoder / Encoder / Decoder handles
    c1_info = None
    if rank == 0 and world == 1 and not args.no_c1:
        c1_info = small_input_leg(eng, zk)

    # ---- seek-to-offset latency: BASELINE.json configs[3] as written (a failing leg fails the run)
    seek_info = None
    if do_seek:
        seek_info = seek_leg(eng, np.asarray(data), zk, z64, dsize, args.seek_trials)

    # ---- N > 1: the one exchange step of the path -- encode the local shard, gather stream + seek table on rank 0 (RCCL over
    # xGMI).  It runs BEFORE the line is printed and its result (or its error) is part of the line; a watchdog bounds it, so a
    # collective that hangs costs four minutes and an error field, not the line.
    gather_info, gather_failed, gather_hung = None, False, False
    if world > 1 and use_gpu_archive:
        import threading
        from zeekstd_amd import parallel
        box = {}

        def _gather_leg():
            try:
                torch.cuda.set_device(local_rank)
                tg, tx = [], []

                def sharded():
                    """encode this rank's shard, then the exchange: over torch.distributed (RCCL), or -- one GPU for all ranks, tests only --
                    through zk_gather_seekable over the named transport.  Returns (stream on the root, its SeekTable, seconds of the exchange alone)."""
                    if not one_gpu:
                        cap_ = int(zk.lib.zk_compress_bound(dsize, FRAME))
                        e_comp = torch.empty(cap_ + 64, dtype=torch.uint8, device=dev)
                        e_cs = torch.empty(nframes, dtype=torch.int32, device=dev); e_ds = torch.empty(nframes, dtype=torch.int32, device=dev)
                        torch.cuda.synchronize()                          # (nothing of torch's in flight on these buffers when the engine's own queue takes them)
                        nfo, written = eng.encode_frames_dev(d_src, dsize, FRAME, level, cks, e_comp, cap_, e_cs, e_ds)
                        torch.cuda.synchronize(); barrier(); tq = time.perf_counter()
                        o, tb = parallel.gather_seekable(e_comp[:written], e_cs[:nfo], e_ds[:nfo], 0)
                        torch.cuda.synchronize(); barrier()
                        return o, tb, time.perf_counter() - tq
                    from zeekstd_amd import SeekTable
                    cap_ = int(zk.lib.zk_compress_bound(dsize, FRAME))
                    e_comp = torch.empty(cap_ + 64, dtype=torch.uint8, device=dev)
                    e_cs = torch.empty(nframes, dtype=torch.int32, device=dev); e_ds = torch.empty(nframes, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    nfo, written = eng.encode_frames_dev(d_src, dsize, FRAME, level, cks, e_comp, cap_, e_cs, e_ds)
                    h_cs = e_cs[:nfo].cpu().numpy().astype(np.uint32); h_ds = e_ds[:nfo].cpu().numpy().astype(np.uint32)
                    ocap = world * cap_ + 8 * world * nframes + 64
                    o = torch.empty(ocap if rank == 0 else 1, dtype=torch.uint8, device=dev)
                    nbytes, tab = C.c_uint64(), C.c_void_p()
                    zk.lib.zk_gather_seekable.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                                          C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
                    if os.environ.get("ZK_BENCH_DEBUG"):
                        print(f"rank {rank}: written {written} sum of sizes {int(h_cs.sum())} frames {nfo}", file=sys.stderr, flush=True)
                    torch.cuda.synchronize(); barrier(); tq = time.perf_counter()
                    rc_ = zk.lib.zk_gather_seekable(eng._h, shm_comm, rank, world, 0, e_comp.data_ptr(), written, h_cs.ctypes.data, h_ds.ctypes.data, nfo, 1,
                                                    o.data_ptr() if rank == 0 else None, ocap if rank == 0 else 0, C.byref(nbytes), C.byref(tab), None)
                    torch.cuda.synchronize(); barrier()
                    tq = time.perf_counter() - tq
                    if rc_ != 0:
                        raise RuntimeError(f"zk_gather_seekable: {zk.error_name(rc_)}")
                    if rank != 0:
                        return None, None, tq
                    return o[:nbytes.value], SeekTable(tab.value), tq
                for _ in range(2):
                    barrier(); torch.cuda.synchronize(); t = time.perf_counter()
                    out, table, tq = sharded()
                    torch.cuda.synchronize(); barrier(); tg.append(time.perf_counter() - t); tx.append(tq)
                # (r5) the gathered archive is one archive: the root decodes the LAST rank's last four frames out of it -- bytes that crossed a
                # link, located through the gathered table -- and compares them with the generator's bytes for those chunks.  No collective
                # here (nothing that could hang a peer); a failure is a field of the line, not the end of the run.
                remote = None
                if rank == 0 and out is not None and table is not None:
                    try:
                        from oracle import zko as _zko                     # untimed: the generator, as for the inputs
                        nf_all = table.num_frames()
                        first_f = nf_all - 4
                        tc, td = table.offsets()
                        c0, c1 = int(tc[first_f]), int(tc[nf_all])
                        rel_c = torch.from_numpy((tc[first_f:] - tc[first_f]).astype(np.int64)).to(dev)
                        rel_d = torch.from_numpy((td[first_f:] - td[first_f]).astype(np.int64)).to(dev)
                        piece = torch.empty(c1 - c0 + 64, dtype=torch.uint8, device=dev)
                        piece[:c1 - c0] = out[c0:c1]
                        piece[c1 - c0:] = 0
                        o4 = torch.empty(4 * FRAME + 64, dtype=torch.uint8, device=dev)
                        s4 = torch.zeros(4, dtype=torch.int32, device=dev)
                        eng.decode_frames_dev(piece, c1 - c0, rel_c, rel_d, 0, 4, o4, 4 * FRAME, True, s4)
                        want4 = _zko.gen_chunks(4 * FRAME, first_f)       # rank r's chunk k is generator chunk r * nframes + k
                        got4 = bytes(o4[:4 * FRAME].cpu().numpy())
                        remote = {"frames": [first_f, nf_all], "bit_exact": got4 == want4 and int(s4.abs().sum().item()) == 0}
                        if not remote["bit_exact"]:
                            remote["status"] = [int(x) for x in s4.cpu().numpy()]
                            remote["first_difference"] = next((i for i in range(len(want4)) if got4[i] != want4[i]), None)
                            # where the gathered stream stops being frames: the first bytes at every rank's first frame
                            remote["table_bytes_per_rank"] = [int(tc[(r + 1) * nframes] - tc[r * nframes]) for r in range(world)]
                            remote["magic_at_each_ranks_first_frame"] = [bytes(out[int(tc[r * nframes]):int(tc[r * nframes]) + 4].cpu().numpy()).hex() for r in range(world)]
                    except Exception as ex:                               # noqa: BLE001
                        remote = {"error": f"{type(ex).__name__}: {ex}"}
                # what the exchange should take on xGMI (DESIGN "multi-GPU"): every peer has its own link into the root (~153 GB/s each way), so the
                # W - 1 receives run side by side and the slowest peer's bytes bound the step; a ring, or receives served one after the other,
                # would show as the SUM over the peers instead
                peer_bytes = csize
                box["info"] = {"encode_plus_gather_GiB_per_s": round(dsize * world / min(tg) / 2**30, 2), "ms": round(min(tg) * 1e3, 2),
                               "gather_alone_ms": round(min(tx) * 1e3, 3),
                               "gather_expectation": {"bytes_per_peer": peer_bytes, "peers": world - 1, "link_GB_s": 153,
                                                      "ms_if_the_receives_overlap": round(peer_bytes / 153e9 * 1e3, 2),
                                                      "ms_if_they_are_serialised": round(peer_bytes * (world - 1) / 153e9 * 1e3, 2),
                                                      "note": "plus two small all-gathers (~0.1 ms) and the root's table (< 1 ms); this rank's own bytes stand for every peer's"},
                               "transport": ("TEST: shared memory between processes on ONE GPU (--one-gpu-transport): the times mean nothing" if one_gpu
                                             else "RCCL (torch.distributed, backend nccl)"),
                               "last_ranks_frames_decoded_from_the_gathered_archive": remote,
                               "frames_on_root": table.num_frames() if table is not None else None,
                               "stream_bytes_on_root": int(out.numel()) if out is not None else None,
                               "note": "every rank encodes its 2048 frames, then all_gather of sizes, point-to-point payload gather into the "
                                       "root's buffer, all_gather of the seek entries, table serialised by the root (zeekstd_amd/parallel.py)"}
            except Exception as ex:                        # noqa: BLE001 -- reported in the line, the run then fails
                box["info"] = {"error": f"{type(ex).__name__}: {ex}"}
        th = threading.Thread(target=_gather_leg, daemon=True)
        th.start()
        th.join(240)
        if th.is_alive():
            gather_info, gather_failed, gather_hung = {"error": "timed out after 240 s: a collective did not complete"}, True, True
        else:
            gather_info = box.get("info", {"error": "no result"})
            gather_failed = "error" in gather_info

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        line = {
            "metric": "decode_decompressed_GiB_per_s", "value": round(value, 3), "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "reference_made_archive": ref_info,
            "spans_of_K_steps_ms_per_step": {"n": len(spans), "median": round(sorted(spans)[len(spans) // 2] / args.steps * 1e3, 3),
                                             "min": round(min(spans) / args.steps * 1e3, 3), "mean": round(sum(spans) / len(spans) / args.steps * 1e3, 3),
                                             "note": "the contract's span first (value / ms_per_step are its), then two more spans of K steps timed alike"},
            "one_batch_at_a_time": {"value": round(dsize * args.steps / sync_elapsed / 2**30, 3), "unit": "GiB/s",
                                    "ms_per_step": round(sync_elapsed / args.steps * 1e3, 3), "steps_ms": _stats_ms(sync_steps),
                                    "checksums_followed": followed,
                                    "checksums_followed_note": "frames verified by zk_k_xxh64_follow beside the executor (same-XCD hand-off: depends on how the "
                                                               "dispatcher deals workgroups to XCDs; the rest is verified behind the executor)",
                                    "note": "same steps through the synchronous zk_decode_frames_dev (rank-local, no overlap between batches)"},
            "config": {"workload": (f"configs[2]: 4 GiB/GPU, 2048 x 2 MiB frames, level {level}, XXH64 checksums verified"
                                    if args.workload == "c3" else f"configs[1]: 256 MiB, 128 x 2 MiB frames, level {level}, decode-only"),
                       "frames_per_gpu": nframes, "frame_size": FRAME, "compressed_bytes_per_gpu": csize,
                       "batches_in_flight": 1 if args.sync else 2,
                       "archive": ("GPU encoder of this engine (zk_encode_frames_dev)" if use_gpu_archive else "CPU libzstd (reference Encoder loop)")
                                  + ", inputs from the SURVEY 8d generator",
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective",
                       "bit_exact": True, **({"kernel_choice": args.choice} if args.choice else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "step_frac": round(algo_bytes / (sync_elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                         "step_frac_note": "the whole step (all kernels, one batch at a time) against the same peak",
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": {k: round(v, 3) for k, v in acc.items()}},
            "cpu_baseline": base,
            "encode": enc_info,
            "round_trip": ({"value": round(dsize / (enc_info["ms"] * 1e-3 + ms_step * 1e-3) / 2**30, 2), "unit": "GiB/s",
                            "note": "HBM-resident encode + decode of the same 4 GiB, encode ms + decode ms_per_step (BASELINE metric: encode+decode)"}
                           if enc_info else None),
            "end_to_end": e2e_info,
            "configs0_small_input": c1_info,
            "rccl_gather": gather_info,
            "seek": seek_info,
            "setup_s": round(t_setup, 1),
        }
        if base:
            # The pair that counts for a user of the reference: the archive zeekstd's own CPU Encoder wrote, GPU rate on it over the
            # CPU Decoder's rate on it.  The GPU-made archive (the headline: configs[2] is encode + decode on the GPU) is the slowest
            # one for the CPU, so its ratios are kept beside, named as what they are.
            r = base.get("on_reference_made_archive")
            if r and ref_info:
                line["speedup_vs_cpu_1thread"] = round(ref_info["value"] / r["value"], 1)
                if r.get("all_cores"):
                    line["speedup_vs_cpu_all_cores"] = round(ref_info["value"] / r["all_cores"]["value"], 2)
                l3 = r.get("level_3")
                if l3 and ref_info.get("level_3"):
                    line["speedup_vs_cpu_level_3"] = {"one_thread": round(ref_info["level_3"]["value"] / l3["value"], 1),
                                                      "all_cores": round(ref_info["level_3"]["value"] / l3["all_cores"]["value"], 2) if l3.get("all_cores") else None}
                line["speedup_note"] = "GPU and CPU both on the archive the reference Encoder loop (libzstd) wrote; *_gpu_made_archive: both on the archive this engine's encoder wrote"
            line["speedup_vs_cpu_1thread_gpu_made_archive"] = round(value / base["value"], 1)
            if base.get("all_cores"):
                line["speedup_vs_cpu_all_cores_gpu_made_archive"] = round(value / base["all_cores"]["value"], 2)
            if "speedup_vs_cpu_1thread" not in line:           # no reference-made leg in this run (N > 1, --no-ref-archive)
                line["speedup_vs_cpu_1thread"] = line["speedup_vs_cpu_1thread_gpu_made_archive"]
                line["speedup_note"] = "GPU and CPU both on the archive this engine's encoder wrote (no reference-made leg in this run)"
        # the numbers a reader of the line's first 500 characters should find (the driver's tail truncates long lines): short scalars in front
        def _g(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        front = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
        front.update({"reference_made_GiB_s": _g(ref_info, "value"), "reference_made_level3_GiB_s": _g(ref_info, "level_3", "value"),
                      "encode_GiB_s": _g(enc_info, "value"), "encode_level3_GiB_s": _g(enc_info, "level_3", "value"), "encode_level3_ratio": _g(enc_info, "level_3", "ratio"), "round_trip_GiB_s": _g(line.get("round_trip"), "value"),
                      "seek_p50_us": _g(seek_info, "gpu_made_archive", "gpu_decoder_us", "p50"), "seek_p50_us_reference_made": _g(seek_info, "reference_made_archive", "gpu_decoder_us", "p50"),
                      "seek_p50_us_cpu_reference": _g(seek_info, "reference_made_archive", "cpu_reference_us", "p50"),
                      "configs0_decoder_GiB_s": _g(c1_info, "decoder", "value"), "roofline_frac": round(achieved / HBM_PEAK_GBS, 5),
                      "speedup_vs_cpu_1thread": line.get("speedup_vs_cpu_1thread"), "speedup_vs_cpu_all_cores": line.get("speedup_vs_cpu_all_cores")})
        front.update({k: v for k, v in line.items() if k not in front})
        line = front
        print(json.dumps(line), flush=True)
    if shm_comm and not gather_hung:
        shm_lib.zkshm_comm_destroy(shm_comm)
    if gather_failed:                                        # the line above carries the error; the run still fails
        if gather_hung:
            os._exit(1)                                       # a collective that never completes cannot be torn down politely
        if world > 1:
            dist.destroy_process_group()
        sys.exit(1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
