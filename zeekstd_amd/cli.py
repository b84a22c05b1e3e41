"""`zeekstd` command line on the MI355X engine: compress / decompress / list with the reference CLI's arguments, file
naming, overwrite rules and output (cli/src/args.rs:112-342, command.rs:33-473, compress.rs:13-107, decompress.rs:10-118).

    python -m zeekstd_amd.cli [compress] [-l LEVEL] [-s 2M] [--frame-size-policy uncompressed] [--no-checksum]
                              [--patch-from FILE] [--seek-table-file FILE] [-o FILE | -c] [-f] [INPUT | -]
    python -m zeekstd_amd.cli decompress [--from N | --from-frame I] [--to N|end | --to-frame I|end] [--patch-apply FILE]
                              [--seek-table-file FILE] [-o FILE | -c] [-f] INPUT
    python -m zeekstd_amd.cli list [--from-frame I] [--to-frame I|end | --num-frames N] [-d] [--seek-table-format foot|head] INPUT

Everything byte-level runs through the C ABI (Encoder / Decoder / SeekTable handles, include/zeekstd_amd.h): a file is
memory-mapped and handed to the encoder where it lies (the engine pins its pages and overlaps PCIe with the kernels), STDIN
is read in 64 MiB pieces, the decoder fills a 64 MiB buffer per call.  Archives are interchangeable with the reference's and
with `zstd -d` (the seek table is a skippable frame)."""
import argparse
import ctypes as C
import mmap
import os
import stat
import sys

CHUNK = 64 << 20
MMAP_THRESHOLD = 0x0010_0000           # args.rs:8 (the constant; its comment says 128 MiB)


class CliError(Exception):
    pass


# ---------------------------------------------------------------- value parsers (args.rs:10-110)
def byte_value(s: str) -> int:
    digits = ""
    for ch in s:
        if ch.isascii() and ch.isdigit():
            digits += ch
        else:
            break
    unit = "".join(c for c in s[len(digits):] if not c.isspace())
    if not digits:
        raise argparse.ArgumentTypeError("invalid digit found in string")
    v = int(digits)
    mult = {"B": 1, "": 1, "K": 1 << 10, "kib": 1 << 10, "M": 1 << 20, "mib": 1 << 20, "G": 1 << 30, "gib": 1 << 30}
    if unit not in mult:
        raise argparse.ArgumentTypeError(f"Unknown unit: {unit!r}")
    v *= mult[unit]
    if v >= 1 << 64:
        raise argparse.ArgumentTypeError("Byte value too large")
    return v


def offset_limit(s: str):
    return None if s.lower() == "end" else byte_value(s)


def last_frame(s: str):
    if s.lower() == "end":
        return "end"
    return u32(s)


def u32(s: str) -> int:
    v = int(s)
    if not 0 <= v < 1 << 32:
        raise argparse.ArgumentTypeError("number too large to fit in target type")
    return v


def u64(s: str) -> int:
    v = int(s)
    if not 0 <= v < 1 << 64:
        raise argparse.ArgumentTypeError("number too large to fit in target type")
    return v


def num_frames(s: str) -> int:
    v = u32(s)
    if v == 0:
        raise argparse.ArgumentTypeError("frame number must be greater than 0")
    return v


def level(s: str) -> int:
    v = int(s)
    if not 1 <= v <= 19:
        raise argparse.ArgumentTypeError("compression level must be between 1 and 19")
    return v


# ---------------------------------------------------------------- byte formats (command.rs:20-28; indicatif HumanBytes)
def human_bytes(n: int) -> str:
    if n < 1024:
        return f"{n} B"
    v = float(n)
    for unit in ("KiB", "MiB", "GiB", "TiB", "PiB", "EiB"):
        v /= 1024.0
        if v < 1024.0 or unit == "EiB":
            return f"{v:.2f} {unit}"
    return f"{n} B"


def raw_bytes(n: int) -> str:
    return str(n)


class Progress:
    """The reference's progress counter (cli/src/args.rs:122-135, command.rs:190-204, compress.rs:69-70, decompress.rs:39-48, 94-95):
    "<done> of <total>" on STDERR, redrawn at most five times a second, cleared at the end; binary units unless --raw-bytes; not
    drawn when quiet, with --no-progress, or when STDERR is not a terminal (indicatif's draw target hides itself then)."""

    def __init__(self, total, pos, args):
        self.total, self.pos = total, pos
        self.fmt = raw_bytes if args.raw_bytes else human_bytes
        self.on = not args.quiet and not getattr(args, "no_progress", False) and sys.stderr.isatty()
        self.last = 0.0
        self.drawn = False

    def inc(self, n):
        self.pos += n
        if not self.on:
            return
        import time
        now = time.monotonic()
        if now - self.last >= 0.2:
            self.last = now
            total = f" of {self.fmt(self.total)}" if self.total is not None else ""
            sys.stderr.write(f"\r\x1b[2K{self.fmt(self.pos)}{total}")
            sys.stderr.flush()
            self.drawn = True

    def finish_and_clear(self):
        if self.drawn:
            sys.stderr.write("\r\x1b[2K")
            sys.stderr.flush()


# ---------------------------------------------------------------- arguments
def build_parser():
    top = argparse.ArgumentParser(prog="zeekstd", description="Compress and decompress data using the Zstandard Seekable Format.")
    flags = argparse.ArgumentParser(add_help=False)
    flags.add_argument("-q", "--quiet", action="store_true", help="Suppress output. Ignored in list mode.")
    flags.add_argument("-r", "--raw-bytes", action="store_true", help="Disable human-readable formatting for all byte numbers.")
    common = argparse.ArgumentParser(add_help=False)
    common.add_argument("-f", "--force", action="store_true", help="Disable input and output checks.")
    common.add_argument("-c", "--stdout", action="store_true", help="Write to STDOUT.")
    common.add_argument("--no-progress", action="store_true", help="Do not show the progress counter.")
    common.add_argument("--mmap-prefix", action="store_true", help="Force memory-mapping prefix (patch) files.")
    common.add_argument("--no-mmap-prefix", action="store_true", help="Force disable memory-mapping prefix (patch) files.")
    common.add_argument("--seek-table-file", help='Path to the seek table file. If specified, implies the "Head" seek table format.')
    sub = top.add_subparsers(dest="command")
    c = sub.add_parser("compress", aliases=["c"], parents=[flags, common],
                       help="Compress INPUT_FILE (default); reads from STDIN if INPUT_FILE is `-` or not provided")
    c.add_argument("-l", "--compression-level", type=level, default=3)
    c.add_argument("--no-checksum", action="store_true", help="Don't include frame checksums.")
    c.add_argument("-s", "--frame-size", type=byte_value, default=byte_value("2M"))
    c.add_argument("--frame-size-policy", choices=["compressed", "uncompressed"], default="uncompressed")
    c.add_argument("--patch-from", help="Provide a reference point for Zstandard's diff engine.")
    c.add_argument("input_file", nargs="?", default="-")
    c.add_argument("-o", "--output-file")
    d = sub.add_parser("decompress", aliases=["d"], parents=[flags, common], help="Decompress INPUT_FILE")
    start = d.add_mutually_exclusive_group()
    start.add_argument("--from", dest="from_", type=u64, default=0, help="The offset (of the uncompressed data) where decompression starts.")
    start.add_argument("--from-frame", type=u32, help="The frame number at which decompression starts.")
    end = d.add_mutually_exclusive_group()
    end.add_argument("--to", type=offset_limit, default=None, help="The offset (of the decompressed data) where decompression ends; accepts 'end'.")
    end.add_argument("--to-frame", type=last_frame, help="The frame number at which decompression ends (inclusive); accepts 'end'.")
    d.add_argument("--patch-apply", help="Provide a reference point for Zstandard's diff engine.")
    d.add_argument("input_file")
    d.add_argument("-o", "--output-file")
    ls = sub.add_parser("list", aliases=["l"], parents=[flags], help="Print information about seekable Zstandard-compressed files")
    ls.add_argument("--from-frame", type=u32)
    lend = ls.add_mutually_exclusive_group()
    lend.add_argument("--to-frame", type=last_frame)
    lend.add_argument("--num-frames", type=num_frames)
    ls.add_argument("-d", "--detail", action="store_true")
    ls.add_argument("--seek-table-format", choices=["head", "foot"], default="foot")
    ls.add_argument("input_file")
    return top


SUBCOMMANDS = {"compress": "compress", "c": "compress", "decompress": "decompress", "d": "decompress", "list": "list", "l": "list"}


def parse(argv):
    """`zeekstd FILE` means `zeekstd compress FILE` (main.rs:14-31: the compress arguments are flattened into the top level and
    conflict with subcommands): the subcommand, if any, is the first token behind the global flags."""
    argv = list(argv)
    if not argv:
        build_parser().print_help(sys.stderr)                # arg_required_else_help
        raise SystemExit(2)
    if argv[0] in ("-V", "--version"):
        print("zeekstd (zeekstd_amd, MI355X engine)")
        raise SystemExit(0)
    i = 0
    while i < len(argv) and argv[i] in ("-q", "--quiet", "-r", "--raw-bytes"):
        i += 1
    if argv[0] not in ("-h", "--help") and (i == len(argv) or argv[i] not in SUBCOMMANDS):
        argv = ["compress"] + argv
    elif i and i < len(argv):                                # global flags in front of the subcommand: argparse wants them behind it
        argv = [argv[i]] + argv[:i] + argv[i + 1:]
    args = build_parser().parse_args(argv)
    if args.command is None:
        build_parser().print_help(sys.stderr)
        raise SystemExit(2)
    args.command = SUBCOMMANDS[args.command]
    return args


# ---------------------------------------------------------------- files
def checked_out_file(path, in_is_file, quiet, force):
    """command.rs:47-82: an existing file is overwritten only after a "y" (never when quiet or when the input comes from STDIN)."""
    if not force and os.path.exists(path):
        try:
            is_char = stat.S_ISCHR(os.stat(path).st_mode)
        except OSError:
            is_char = False
        if not is_char:
            if quiet or not in_is_file:
                raise CliError(f"{path} already exists; not overwritten")
            sys.stderr.write(f"{path} already exists; overwrite (y/n) ? ")
            sys.stderr.flush()
            if sys.stdin.readline().rstrip("\n") != "y":
                raise CliError(f"{path} already exists")
    try:
        return open(path, "wb")
    except OSError as e:
        raise CliError(f"Failed to open output file: {e}")


def out_path_of(args):
    """command.rs:96-135"""
    in_path = None if args.input_file == "-" else args.input_file
    if args.command == "list" or args.stdout:
        return None
    if args.command == "compress":
        return args.output_file or (in_path + ".zst" if in_path else None)
    if args.output_file:
        return args.output_file
    if in_path is not None and not in_path.endswith(".zst"):
        raise CliError(f"{in_path}: unknown extension (.zst expected); cannot derive the output file name")
    return in_path[:-4] if in_path else None


def new_writer(args, out_path, in_is_file):
    if out_path is not None:
        return checked_out_file(out_path, in_is_file, args.quiet, args.force)
    if not args.force and sys.stdout.isatty():
        raise CliError("stdout is a terminal, aborting")
    return sys.stdout.buffer


class Mapped:
    """a file as read-only memory + its address (the C ABI takes pointers); empty files map to nothing"""
    def __init__(self, path, what):
        try:
            self.f = open(path, "rb")
        except OSError as e:
            raise CliError(f"Failed to open {what}: {e}")
        self.n = os.fstat(self.f.fileno()).st_size
        self.m = mmap.mmap(self.f.fileno(), 0, access=mmap.ACCESS_READ) if self.n else None
        self.view = memoryview(self.m) if self.m else memoryview(b"")
        self._empty = C.create_string_buffer(1)
        self.ptr = _addr(self.view) if self.m else C.addressof(self._empty)

    def close(self):
        self.view.release()
        if self.m:
            self.m.close()
        self.f.close()


def _addr(view):
    import numpy as np
    return np.frombuffer(view, np.uint8).ctypes.data


def load_prefix(path, use_mmap):
    """command.rs:396-418: the patch reference, memory-mapped or read"""
    if path is None:
        return None
    if use_mmap:
        return Mapped(path, "prefix (patch) file")
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError as e:
        raise CliError(f"Failed to load prefix (patch) file: {e}")

    class Held:
        pass
    h = Held()
    h.n = len(data)
    h.buf = C.create_string_buffer(data, max(len(data), 1))
    h.ptr = C.addressof(h.buf)
    h.close = lambda: None
    return h


def use_mmap(args, prefix_len):
    if args.mmap_prefix:
        return True
    if args.no_mmap_prefix:
        return False
    return prefix_len is not None and prefix_len >= MMAP_THRESHOLD


# ---------------------------------------------------------------- commands
def run_compress(args, api, lib, eng):
    in_path = None if args.input_file == "-" else args.input_file
    src = None
    if in_path is not None:
        src = Mapped(in_path, "input file")                  # validated before any output file is created (command.rs:142-156)
    elif not args.force and sys.stdin.isatty():
        raise CliError("stdin is a terminal, aborting")
    if args.frame_size >= 1 << 32:
        raise CliError("Frame size too big")
    out_path = out_path_of(args)
    prefix_len = os.path.getsize(args.patch_from) if args.patch_from and os.path.exists(args.patch_from) else None
    st_file = None
    if args.seek_table_file:
        try:
            st_file = checked_out_file(args.seek_table_file, in_path is not None, args.quiet, args.force)
        except CliError as e:
            raise CliError(f"Failed to create seek table file: {e}")
    writer = new_writer(args, out_path, in_path is not None)
    policy = api.FrameSizePolicy.Compressed(args.frame_size) if args.frame_size_policy == "compressed" else api.FrameSizePolicy.Uncompressed(args.frame_size)
    opts = api.EncodeOptions().engine(eng).frame_size_policy(policy).checksum_flag(not args.no_checksum).compression_level(args.compression_level)
    enc = api.Encoder(writer, opts)
    prefix = load_prefix(args.patch_from, use_mmap(args, prefix_len))
    if args.patch_from and prefix is None:
        raise CliError("Failed to load prefix (patch) file")
    read = 0
    bar = Progress(src.n if src is not None else None, 0, args)

    def feed(ptr, n):
        done = 0
        while done < n:
            if prefix is not None and prefix.n:
                k = lib.zk_encoder_compress_with_prefix(enc._h, ptr + done, n - done, prefix.ptr, prefix.n)
            else:
                k = lib.zk_encoder_compress(enc._h, ptr + done, n - done)
            if k < 0:
                raise CliError(f"Failed to compress data: {api.Error(int(k))}")
            done += k
    if src is not None:
        # the whole mapping in pieces of 1 GiB (256 MiB while a progress counter is drawn): each call is encoded where it lies
        piece_len = (256 << 20) if bar.on else (1 << 30)
        for at in range(0, src.n, piece_len):
            n = min(piece_len, src.n - at)
            feed(src.ptr + at, n)
            read += n
            bar.inc(n)
    else:
        while True:
            piece = sys.stdin.buffer.read(CHUNK)
            if not piece:
                break
            buf = C.create_string_buffer(piece, len(piece))
            feed(C.addressof(buf), len(piece))
            read += len(piece)
            bar.inc(len(piece))
    bar.finish_and_clear()
    if st_file is not None:                                  # compress.rs:86-96: frames to the output, the table (Head format) to its own file
        enc.end_frame()
        enc.flush()
        written = enc.written_compressed()
        ser = enc.seek_table().into_format_serializer(api.Format.Head)
        tbytes = ser.read()
        st_file.write(tbytes)
        st_file.close()
        written += len(tbytes)
    else:
        written = enc.finish()
    if writer is not sys.stdout.buffer:
        writer.close()
    else:
        writer.flush()
    if prefix is not None:
        prefix.close()
    if src is not None:
        src.close()
    if not args.quiet:
        fmt = raw_bytes if args.raw_bytes else human_bytes
        ratio = 100.0 / read * written if read else float("nan")
        sys.stderr.write(f"{in_path or 'STDIN'} : {ratio:.2f}% ( {fmt(read)} => {fmt(written)}, {out_path or 'STDOUT'})\n")


def run_decompress(args, api, lib, eng):
    out_path = out_path_of(args)
    if not os.path.exists(args.input_file):
        raise CliError("Failed to open input file: No such file or directory")
    prefix_len = os.path.getsize(args.patch_apply) if args.patch_apply and os.path.exists(args.patch_apply) else None
    writer = new_writer(args, out_path, True)
    try:
        if args.seek_table_file:
            try:
                with open(args.seek_table_file, "rb") as f:
                    table = api.SeekTable.from_reader(f.read())
            except OSError as e:
                raise CliError(f"Failed to open seek table file: {e}")
        else:
            m = Mapped(args.input_file, "input file")
            h = C.c_void_p()
            rc = lib.zk_seek_table_from_bytes(m.ptr, m.n, int(api.Format.Foot), C.byref(h))
            m.close()
            if rc < 0:
                raise api.Error(rc)
            table = api.SeekTable(h)
    except api.Error as e:
        raise CliError(f"Failed to parse seek table: {e}")
    try:                                                     # args.rs:264-293
        offset = table.frame_start_decomp(args.from_frame) if args.from_frame is not None else args.from_
    except api.Error as e:
        raise CliError(f"Failed to get decompression offset: {e}")
    try:
        if args.to_frame is not None:
            limit = table.size_decomp() if args.to_frame == "end" else table.frame_end_decomp(args.to_frame)
        else:
            limit = table.size_decomp() if args.to is None else args.to
    except api.Error as e:
        raise CliError(f"Failed to get decompression offset limit: {e}")
    try:
        dec = api.DecodeOptions(args.input_file).engine(eng).seek_table(table).offset(offset).offset_limit(limit).into_decoder()
    except api.Error as e:
        raise CliError(f"Failed to create decoder: {e}")
    prefix = load_prefix(args.patch_apply, use_mmap(args, prefix_len))
    buf = bytearray(min(CHUNK, max(131072, limit - offset if limit > offset else 131072)))
    arr = (C.c_uint8 * len(buf)).from_buffer(buf)
    written = 0
    bar = Progress(limit, offset, args)
    while True:
        got = C.c_size_t()
        if prefix is not None and prefix.n:
            rc = lib.zk_decoder_decompress_with_prefix(dec._h, arr, len(buf), prefix.ptr, prefix.n, C.byref(got))
            n = got.value if rc == 0 else rc
        else:
            n = lib.zk_decoder_decompress(dec._h, arr, len(buf))
        if n < 0:
            raise CliError(f"Failed to decompress data: {api.Error(int(n))}")
        if n == 0:
            break
        try:
            writer.write(memoryview(buf)[:n])
        except OSError as e:
            raise CliError(f"Failed to write decompressed data: {e}")
        written += n
        bar.inc(n)
    bar.finish_and_clear()
    del arr
    if writer is not sys.stdout.buffer:
        writer.close()
    else:
        writer.flush()
    if prefix is not None:
        prefix.close()
    if not args.quiet:
        fmt = raw_bytes if args.raw_bytes else human_bytes
        sys.stderr.write(f"{args.input_file} : {fmt(written)}\n")


def run_list(args, api, lib):
    m = Mapped(args.input_file, "input file")
    h = C.c_void_p()
    rc = lib.zk_seek_table_from_bytes(m.ptr, m.n, int(api.Format.Head if args.seek_table_format == "head" else api.Format.Foot), C.byref(h))
    m.close()
    if rc < 0:
        raise CliError(f"Failed to read seek table: {api.Error(rc)}")
    st = api.SeekTable(h)
    fmt = raw_bytes if args.raw_bytes else human_bytes
    if args.num_frames is not None:
        end = (args.from_frame or 0) + args.num_frames - 1
    elif args.to_frame is not None:
        end = st.num_frames() - 1 if args.to_frame == "end" else args.to_frame
    else:
        end = None
    out = sys.stdout
    if args.from_frame is None and end is None and not args.detail:          # command.rs:420-441
        n = st.num_frames()
        comp, unc = st.frame_end_comp(n - 1), st.frame_end_decomp(n - 1)
        out.write("{: <15} {: <15} {: <15} {: <15} {: <10} {: <15}\n".format("Frames", "Compressed", "Uncompressed", "Max Frame Size", "Ratio", "Filename"))
        out.write("{: <15} {: <15} {: <15} {: <15} {: <10.3f} {: <15}\n".format(n, fmt(comp), fmt(unc), fmt(st.max_frame_size_decomp()), unc / comp, args.input_file))
        return
    start = args.from_frame or 0                                               # command.rs:443-473
    if end is None:
        end = st.num_frames() - 1
    if start > end:
        raise CliError(f"Start frame ({start}) cannot be greater than end frame ({end})")
    out.write("{: <15} {: <15} {: <15} {: <20} {: <20}\n".format("Frame Index", "Compressed", "Uncompressed", "Compressed Offset", "Uncompressed Offset"))
    try:
        for i in range(start, end + 1):
            out.write("{: <15} {: <15} {: <15} {: <20} {: <20}\n".format(i, fmt(st.frame_size_comp(i)), fmt(st.frame_size_decomp(i)),
                                                                          fmt(st.frame_start_comp(i)), fmt(st.frame_start_decomp(i))))
    except api.Error as e:
        out.flush()
        raise CliError(str(e))


def main(argv=None):
    args = parse(sys.argv[1:] if argv is None else argv)
    try:
        os.environ.setdefault("ZEEKSTD_AMD_NO_TORCH", "1")    # the command line needs no torch: the library binds the system HIP runtime
        import zeekstd_amd as zk
        from zeekstd_amd import api
        lib = zk.lib
        lib.zk_encoder_compress.restype = C.c_int64
        lib.zk_encoder_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.zk_encoder_compress_with_prefix.restype = C.c_int64
        lib.zk_encoder_compress_with_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.zk_decoder_decompress.restype = C.c_int64
        lib.zk_decoder_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.zk_decoder_decompress_with_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.zk_seek_table_from_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        if args.command == "list":
            run_list(args, api, lib)
        else:
            eng = zk.Engine(0)
            try:
                (run_compress if args.command == "compress" else run_decompress)(args, api, lib, eng)
            finally:
                eng.close()
    except CliError as e:
        sys.stderr.write(f"Error: {e}\n")
        return 1
    except BrokenPipeError:
        return 1
    except Exception as e:                                   # engine / binding errors keep the reference's shape: "Error: ..."
        sys.stderr.write(f"Error: {type(e).__name__}: {e}\n")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
