"""The `zeekstd` command line (zeekstd_amd/cli.py) against the reference CLI's integration tests
(cli/tests/integration/main.rs:146-601), restated scenario by scenario: every command is a subprocess of
`python -m zeekstd_amd.cli`, like cargo_bin_cmd!("zeekstd").  The reference's input (assets/dickens.txt) is absent from the
checkout; SURVEY 8d's stand-in takes its place (a 1.2 MB cut keeps the 10-byte-frame cycles quick: 120 k frames each).
The list / parser scenarios need no GPU; compress / decompress run on the engine (-m gpu)."""
import os
import subprocess
import sys

import pytest

from oracle import zko
from oracle import seek_table as ST
from oracle import libzstd_ref as Z

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAME_SIZES = ["10", "123", "3K", "2M", "1G"]                      # main.rs:10
INPUT = zko.gen_chunks(1_200_000, 17)


def _msgs(stderr: bytes):
    """the command's own stderr lines (libdrm on the GPU box prints a line of its own about amdgpu.ids)"""
    return [l for l in stderr.decode().splitlines() if "amdgpu.ids" not in l]


def zeekstd(*args, stdin=b"", ok=True, cwd=None):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "zeekstd_amd.cli", *map(str, args)], input=stdin, capture_output=True, env=env, cwd=cwd, timeout=600)
    assert (r.returncode == 0) == ok, (args, r.returncode, r.stderr[-600:])
    return r


@pytest.fixture(scope="module")
def test_input(tmp_path_factory):
    p = tmp_path_factory.mktemp("in") / "dickens.txt"
    p.write_bytes(INPUT)
    return p


def compress_test_input(test_input, out_path, frame_size, *more):
    zeekstd("compress", test_input, "--output-file", out_path, "--frame-size", frame_size, *more, stdin=b"y")


def verify_compressed_file(path, tmp_path):
    out = tmp_path / "verify.out"
    zeekstd("decompress", path, "--output-file", out, stdin=b"y")
    assert out.read_bytes() == INPUT
    if Z.load("system") is not None and os.path.getsize(path) < 8 << 20:      # interchangeable with the box's libzstd (`zstd -d`)
        assert Z.decode_stream(open(path, "rb").read(), len(INPUT), "system") == INPUT


# ------------------------------------------------------------------------------------------------ no GPU needed
def test_value_parsers_and_default_subcommand():
    from zeekstd_amd import cli
    import argparse
    assert [cli.byte_value(s) for s in ("10", "10B", "3K", "3 kib", "2M", "2mib", "1G", "1 gib")] == [10, 10, 3072, 3072, 2 << 20, 2 << 20, 1 << 30, 1 << 30]
    for bad in ("1T", "K", "1 Kb", str(1 << 64)):                  # args.rs:345-435
        with pytest.raises((argparse.ArgumentTypeError, ValueError)):
            cli.byte_value(bad)
    assert cli.offset_limit("END") is None and cli.offset_limit("7K") == 7168 and cli.last_frame("End") == "end" and cli.last_frame("7") == 7
    with pytest.raises(argparse.ArgumentTypeError):
        cli.num_frames("0")
    assert cli.human_bytes(1023) == "1023 B" and cli.human_bytes(10192446) == "9.72 MiB" and cli.raw_bytes(5) == "5"
    a = cli.parse(["-q", "in.txt", "-o", "x"])                     # main.rs:14-31: no subcommand = compress
    assert (a.command, a.quiet, a.input_file, a.output_file, a.compression_level, a.frame_size) == ("compress", True, "in.txt", "x", 3, 2 << 20)
    assert cli.parse(["d", "a.zst", "--from", "5", "--to", "end"]).command == "decompress"
    assert cli.out_path_of(cli.parse(["c", "dir/file"])) == "dir/file.zst"
    assert cli.out_path_of(cli.parse(["d", "dir/file.tar.zst"])) == "dir/file.tar"
    with pytest.raises(cli.CliError):
        cli.out_path_of(cli.parse(["d", "dir/file.foo"]))
    assert cli.out_path_of(cli.parse(["c", "-c", "f"])) is None


def _archive(frame_size):
    frames, payload = [], bytearray()
    for o in range(0, len(INPUT), frame_size):
        f = zko.frame_encode(INPUT[o:o + frame_size], 1, True)
        payload += f
        frames.append((len(f), len(INPUT[o:o + frame_size])))
    return bytes(payload), frames


def test_list_seekable(tmp_path):                                   # main.rs:543-575
    fs = len(INPUT) // 14
    payload, frames = _archive(fs)
    p = tmp_path / "a.zst"
    p.write_bytes(payload + ST.serialize(frames, "foot"))
    out = zeekstd("list", p).stdout
    assert out.count(b"\n") == 2 and out.split(b"\n")[0].split() == b"Frames Compressed Uncompressed Max Frame Size Ratio Filename".split()
    assert out.split(b"\n")[1].split()[0] == b"15"
    out = zeekstd("list", "--detail", p).stdout
    assert out.count(b"\n") == 16
    out = zeekstd("-r", "list", p, "--from-frame", 3, "--num-frames", 2).stdout.decode().splitlines()
    c3 = sum(c for c, _ in frames[:3])
    assert len(out) == 3 and out[1].split() == ["3", str(frames[3][0]), str(fs), str(c3), str(3 * fs)]
    assert zeekstd("list", p, "--from-frame", 5, "--to-frame", 2, ok=False).stderr.startswith(b"Error: Start frame (5) cannot be greater than end frame (2)")
    zeekstd("list", tmp_path / "missing", ok=False)


def test_list_separate_seek_table(tmp_path):                        # main.rs:577-601
    _, frames = _archive(len(INPUT) // 6)
    t = tmp_path / "seek_table"
    t.write_bytes(ST.serialize(frames, "head"))
    out = zeekstd("list", t, "--seek-table-format", "head").stdout
    assert out.count(b"\n") == 2 and out.split(b"\n")[1].split()[0] == str(len(frames)).encode()
    zeekstd("list", t, ok=False)                                    # read as Foot: the magic is not at the end


# ------------------------------------------------------------------------------------------------ on the engine
def test_progress_counter(monkeypatch, capsys):
    """args.rs:122-135 + command.rs:190-204: "<done> of <total>", binary units unless --raw-bytes, nothing when quiet / --no-progress /
    STDERR is no terminal, cleared at the end."""
    import argparse
    import time
    from zeekstd_amd import cli

    class Tty:
        def __init__(self): self.s = ""
        def isatty(self): return True
        def write(self, x): self.s += x
        def flush(self): pass
    ns = argparse.Namespace(quiet=False, raw_bytes=False, no_progress=False)
    tty = Tty()
    monkeypatch.setattr(cli.sys, "stderr", tty)
    bar = cli.Progress(10 << 20, 0, ns)
    bar.inc(3 << 20)
    assert tty.s.endswith("3.00 MiB of 10.00 MiB")
    bar.inc(1 << 20)                                        # inside a fifth of a second: not redrawn
    assert tty.s.endswith("3.00 MiB of 10.00 MiB")
    bar.last -= 1.0
    bar.inc(1 << 20)
    assert tty.s.endswith("5.00 MiB of 10.00 MiB") and bar.pos == 5 << 20
    bar.finish_and_clear()
    assert tty.s.endswith("\r\x1b[2K")
    tty.s = ""
    raw = cli.Progress(None, 7, argparse.Namespace(quiet=False, raw_bytes=True, no_progress=False))
    raw.inc(5)
    assert tty.s.endswith("12")                              # STDIN: no total
    for ns2 in (argparse.Namespace(quiet=True, raw_bytes=False, no_progress=False), argparse.Namespace(quiet=False, raw_bytes=False, no_progress=True)):
        tty.s = ""
        b2 = cli.Progress(100, 0, ns2)
        b2.inc(50); b2.finish_and_clear()
        assert tty.s == "" and b2.pos == 50


@pytest.mark.gpu
@pytest.mark.parametrize("fs", FRAME_SIZES)
def test_cycle(test_input, tmp_path, fs):                           # main.rs:146-151
    c = tmp_path / "c.zst"
    compress_test_input(test_input, c, fs)
    verify_compressed_file(c, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("fs", FRAME_SIZES)
def test_cycle_stdin(tmp_path, fs):                                 # main.rs:153-158
    c = tmp_path / "test.zst"
    zeekstd("compress", "--output-file", c, "--frame-size", fs, stdin=INPUT)
    verify_compressed_file(c, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("fs", FRAME_SIZES)
def test_cycle_stdout(test_input, tmp_path, fs):                    # main.rs:160-165
    c = tmp_path / "c.zst"
    c.write_bytes(zeekstd("compress", test_input, "--stdout", "--frame-size", fs, stdin=b"y").stdout)
    verify_compressed_file(c, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("fs", FRAME_SIZES)
def test_cycle_stdin_to_stdout(tmp_path, fs):                       # main.rs:167-172
    c = tmp_path / "c.zst"
    c.write_bytes(zeekstd("compress", "--stdout", "--frame-size", fs, stdin=INPUT).stdout)
    verify_compressed_file(c, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("fs", FRAME_SIZES)
def test_cycle_with_separate_seek_table(test_input, tmp_path, fs):  # main.rs:174-179
    c, t, d = tmp_path / "seekable.zst", tmp_path / "seek_table", tmp_path / "out"
    zeekstd("compress", test_input, "--output-file", c, "--frame-size", fs, "--seek-table-file", t)
    assert t.read_bytes()[:4] == bytes.fromhex("5e2a4d18")         # a Head-format table: skippable magic first
    zeekstd("decompress", c, "--seek-table-file", t, "--output-file", d, stdin=b"y")
    assert d.read_bytes() == INPUT


@pytest.mark.gpu
def test_derive_out_names(tmp_path):                                # main.rs:181-228
    i = tmp_path / "tmpfile"
    i.write_bytes(b"foo")
    zeekstd("compress", i)
    assert (tmp_path / "tmpfile.zst").exists()
    c = tmp_path / "seekable.zst"
    zeekstd("compress", i, "--output-file", c)
    zeekstd("decompress", c)
    assert (tmp_path / "seekable").read_bytes() == b"foo"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["seekable", "seekable.foo"])
def test_fail_to_derive_out_name_when_decompressing(tmp_path, name):    # main.rs:230-282
    i = tmp_path / "in"
    i.write_bytes(b"foo")
    c = tmp_path / name
    zeekstd("compress", i, "--output-file", c)
    assert c.exists()
    assert b"unknown extension (.zst expected)" in zeekstd("decompress", c, ok=False).stderr
    d = tmp_path / "decompressed"                                   # main.rs:284-318: a given output path is used
    zeekstd("decompress", c, "--output-file", d)
    assert d.read_bytes() == b"foo"


@pytest.mark.gpu
def test_overwrite_rules(test_input, tmp_path):                     # main.rs:320-392
    out = tmp_path / "exists.zst"
    out.write_bytes(b"")
    assert b"already exists" in zeekstd("compress", test_input, "--output-file", out, ok=False).stderr          # no "y" on stdin
    assert b"already exists; not overwritten" in zeekstd("compress", "--output-file", out, stdin=INPUT, ok=False).stderr
    assert out.read_bytes() == b""
    t = tmp_path / "table"
    t.write_bytes(b"")
    o2 = tmp_path / "bar.zst"
    zeekstd("compress", test_input, "--output-file", o2, "--seek-table-file", t, ok=False)
    zeekstd("compress", "--output-file", o2, "--seek-table-file", t, stdin=INPUT, ok=False)
    zeekstd("compress", test_input, "--output-file", out, "--force")
    zeekstd("compress", "--output-file", out, "--force", stdin=INPUT)
    assert out.stat().st_size > 1000
    # quiet never asks (command.rs:63-66)
    assert b"not overwritten" in zeekstd("-q", "compress", test_input, "--output-file", out, stdin=b"y", ok=False).stderr


@pytest.mark.gpu
def test_do_not_create_out_file_if_input_file_does_not_exist(tmp_path):   # main.rs:394-408
    o = tmp_path / "bar.zst"
    zeekstd("compress", tmp_path / "foo", "--output-file", o, ok=False)
    assert not o.exists()


@pytest.mark.gpu
def test_decompress_frames(test_input, tmp_path):                   # main.rs:410-449, 451-493
    fs = len(INPUT) // 6
    s, t = tmp_path / "s.zst", tmp_path / "t"
    compress_test_input(test_input, s, fs)
    first = zeekstd("decompress", s, "-c", "--from-frame", 0, "--to-frame", 0).stdout
    assert len(first) == fs
    rest = zeekstd("decompress", s, "-c", "--from-frame", 1, "--to-frame", "end").stdout
    assert first + rest == INPUT
    zeekstd("compress", test_input, "--frame-size", fs, "--output-file", s, "--seek-table-file", t, "--force")
    assert zeekstd("decompress", s, "--seek-table-file", t, "-c", "--from-frame", 0, "--to-frame", 0).stdout == INPUT[:fs]


@pytest.mark.gpu
def test_decompress_frame_index_out_of_range(test_input, tmp_path):  # main.rs:495-518
    s = tmp_path / "one.zst"
    compress_test_input(test_input, s, len(INPUT))
    zeekstd("decompress", s, "-c", "--from-frame", 1, ok=False)
    zeekstd("decompress", s, "-c", "--from-frame", 0, "--to-frame", 1, ok=False)


@pytest.mark.gpu
def test_decompress_between_offset_and_offset_limit(test_input, tmp_path):   # main.rs:520-541
    fs = len(INPUT) // 9
    s = tmp_path / "s.zst"
    compress_test_input(test_input, s, fs)
    a, b = fs + fs // 2, 4 * fs + fs // 2
    assert zeekstd("decompress", s, "-c", "--from", a, "--to", b).stdout == INPUT[a:b]


@pytest.mark.gpu
def test_list_on_an_engine_made_archive_and_summary_lines(test_input, tmp_path):   # main.rs:543-575 on what `compress` wrote
    s = tmp_path / "s.zst"
    r = zeekstd("-r", "compress", test_input, "--output-file", s, "--frame-size", len(INPUT) // 14)
    size = s.stat().st_size
    assert _msgs(r.stderr) == [f"{test_input} : {100.0 / len(INPUT) * size:.2f}% ( {len(INPUT)} => {size}, {s})"]     # command.rs:349-357
    assert zeekstd("list", s).stdout.count(b"\n") == 2 and zeekstd("list", "--detail", s).stdout.count(b"\n") == 16
    assert _msgs(zeekstd("decompress", s, "-c").stderr) == [f"{s} : 1.14 MiB"]                                      # command.rs:372-378
    assert _msgs(zeekstd("-q", "decompress", s, "-c").stderr) == []


@pytest.mark.gpu
def test_patch_from_and_patch_apply(tmp_path):                      # cli/src/compress.rs:31-37, decompress.rs:54-65; lib.rs:202-263
    old = zko.gen_chunks(300_000, 40)
    new = old[:100_000] + b"-- a changed paragraph --" + old[100_000:250_000] + zko.gen_text(20_000, 41)
    o, n, p, r = tmp_path / "old", tmp_path / "new", tmp_path / "patch.zst", tmp_path / "restored"
    o.write_bytes(old); n.write_bytes(new)
    zeekstd("compress", n, "--patch-from", o, "--output-file", p)
    plain = tmp_path / "plain.zst"
    zeekstd("compress", n, "--output-file", plain)
    assert p.stat().st_size < plain.stat().st_size                  # the part of the old file the matcher reaches is reused
    zeekstd("decompress", p, "--patch-apply", o, "--output-file", r)
    assert r.read_bytes() == new
    zeekstd("decompress", p, "--patch-apply", o, "--mmap-prefix", "-c")
    zeekstd("decompress", p, "-c", ok=False)                        # without the reference the frames do not decode
    zeekstd("compress", n, "--patch-from", tmp_path / "missing", "-c", ok=False)


@pytest.mark.gpu
def test_patch_of_a_large_file_is_small(tmp_path):                  # cli/src/compress.rs:31-37: windowLog over the old file + long-distance matching
    """The old file is far larger than the matcher's ring: the patch still costs about what the changes cost (long-distance
    table over the prefix), and applies with this CLI."""
    old = zko.gen_text(6 << 20, 60)
    new = old[:1_000_000] + zko.gen_text(3000, 61) + old[1_000_400:4_000_000] + old[4_100_000:] + zko.gen_text(5000, 62)
    o, n, p, r = tmp_path / "old", tmp_path / "new", tmp_path / "patch.zst", tmp_path / "restored"
    o.write_bytes(old); n.write_bytes(new)
    zeekstd("compress", n, "--patch-from", o, "--output-file", p)
    assert p.stat().st_size < 16_000                                # 8 KB of new text, three seams, three frames
    zeekstd("decompress", p, "--patch-apply", o, "--output-file", r)
    assert r.read_bytes() == new


@pytest.mark.gpu
def test_levels_policy_and_checksum_flags(test_input, tmp_path):    # args.rs:185-207
    a, b, c = tmp_path / "l1.zst", tmp_path / "l19.zst", tmp_path / "nocks.zst"
    zeekstd("compress", test_input, "-l", 1, "-o", a)
    zeekstd("compress", test_input, "--compression-level", 19, "-o", b)
    assert b.stat().st_size < a.stat().st_size
    zeekstd("compress", test_input, "-l", 0, "-o", tmp_path / "x", ok=False)
    zeekstd("compress", test_input, "--no-checksum", "-o", c)
    assert c.read_bytes()[4] & 4 == 0 and a.read_bytes()[4] & 4 == 4      # Frame_Header_Descriptor bit 2 (encode.rs:834-870)
    k = tmp_path / "comp.zst"
    zeekstd("compress", test_input, "--frame-size", "64K", "--frame-size-policy", "compressed", "-o", k)
    sizes = [int(l.split()[1]) for l in zeekstd("-r", "list", "-d", k).stdout.decode().splitlines()[1:]]
    assert len(sizes) > 3 and all(65536 <= x < 65536 + 131591 for x in sizes[:-1])
    verify_compressed_file(k, tmp_path)
