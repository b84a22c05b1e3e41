//! zeekstd's public types on the MI355X engine: `EncodeOptions / Encoder<W> / DecodeOptions / Decoder<S> / SeekTable`
//! with the reference's method names, argument meaning and error behaviour (citations: /root/reference/lib/src), over the C
//! ABI of `libzeekstd_amd.so` (`ffi`, generated from `include/zeekstd_amd.h`).  The frame hot path -- what the reference
//! does through `ZSTD_compressStream2` / `ZSTD_decompressStream` (encode.rs:340-346, decode.rs:242-256) -- runs as HIP
//! kernels; seek table, offsets and sources keep their reference semantics.
//!
//! Upstream's public surface is complete here: the crate-root types, `seek_table::{Format, Serializer}`, `BytesWrapper`,
//! the four constants, `Error`'s predicates and `Display` -- tests/test_abi.py diffs this file's public items against that list.
//!
//! NOT COMPILED in the build image (no cargo / rustc there): kept in step with the header by tools/gen_rust_ffi.py and
//! tests/test_abi.py.  The C++ classes in zeekstd_amd/csrc/host/zeekstd.hpp are the compiled twin of this file.
pub mod ffi;

use core::ffi::{c_int, c_void, CStr};
use std::io::{self, Read, Seek, SeekFrom, Write};

/// error.rs:101-113 -- zeekstd's kinds; `Zstd` keeps the wrapped `0 - code` like error.rs:40-45.
#[derive(Debug)]
pub enum Error {
    NumberConversionFailed,
    OffsetOutOfRange,
    FrameIndexTooLarge,
    IO(io::Error),
    Zstd(usize),
}
pub type Result<T> = core::result::Result<T, Error>;
/// re-exported from zstd-safe upstream (lib.rs:49): the level is what `ZSTD_c_compressionLevel` takes
pub type CompressionLevel = i32;

/// lib.rs:52-58
pub const SEEKABLE_MAGIC_NUMBER: u32 = 0x8F92_EAB1;
pub const SEEKABLE_MAX_FRAMES: u32 = 0x0800_0000;
pub const SEEK_TABLE_INTEGRITY_SIZE: usize = 9;
pub const SEEKABLE_MAX_FRAME_SIZE: usize = 0x4000_0000;
pub(crate) const SKIPPABLE_HEADER_SIZE: usize = 8;      // lib.rs:62
const SKIPPABLE_MAGIC: u32 = 0x184D_2A5E;               // seekable_format.md:59-70

impl Error {
    pub fn is_offset_out_of_range(&self) -> bool { matches!(self, Error::OffsetOutOfRange) }       // error.rs:25
    pub fn is_frame_index_too_large(&self) -> bool { matches!(self, Error::FrameIndexTooLarge) }   // error.rs:36
    pub fn is_zstd(&self) -> bool { matches!(self, Error::Zstd(_)) }                               // error.rs:55
    pub fn is_number_conversion_failed(&self) -> bool { matches!(self, Error::NumberConversionFailed) }   // error.rs:14
    pub fn is_io(&self) -> bool { matches!(self, Error::IO(_)) }                                   // error.rs:50
    fn zstd(code: usize) -> Error { Error::Zstd(0usize.wrapping_sub(code)) }                       // error.rs:40-45: the wrapped value is 0 - code
    fn from_code(rc: i64) -> Error {
        match rc {
            -1001 => Error::OffsetOutOfRange,
            -1002 => Error::FrameIndexTooLarge,
            -1003 => Error::NumberConversionFailed,
            rc if rc > -1000 => Error::Zstd(0usize.wrapping_sub((-rc) as usize)),              // a ZSTD_ErrorCode
            rc => {
                let name = unsafe { CStr::from_ptr(ffi::zk_error_name(rc as c_int)) }.to_string_lossy().into_owned();
                Error::IO(io::Error::new(io::ErrorKind::Other, name))                           // HIP / device / IO failures
            }
        }
    }
}
impl From<io::Error> for Error { fn from(e: io::Error) -> Self { Error::IO(e) } }                // error.rs:85
impl From<core::num::TryFromIntError> for Error { fn from(_: core::num::TryFromIntError) -> Self { Error::NumberConversionFailed } }   // error.rs:75
/// error.rs:60-71: the kinds in words, a zstd code by its name
impl core::fmt::Display for Error {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        match self {
            Error::OffsetOutOfRange => f.write_str("offset out of range"),
            Error::FrameIndexTooLarge => f.write_str("frame index too large"),
            Error::NumberConversionFailed => f.write_str("number conversion failed"),
            Error::IO(e) => write!(f, "io error: {e}"),
            Error::Zstd(w) => {
                let code = 0usize.wrapping_sub(*w) as c_int;
                f.write_str(&unsafe { CStr::from_ptr(ffi::zk_error_name(-code)) }.to_string_lossy())
            }
        }
    }
}
impl std::error::Error for Error {}
fn check(rc: c_int) -> Result<()> { if rc == 0 { Ok(()) } else { Err(Error::from_code(rc as i64)) } }
fn check_len(rc: i64) -> Result<usize> { if rc >= 0 { Ok(rc as usize) } else { Err(Error::from_code(rc)) } }

/// One GPU's engine: what `CCtx::create()` / `DCtx::create()` are to the reference (encode.rs:130, decode.rs:31).
/// Inject it with `EncodeOptions::engine` / `DecodeOptions::engine` like `with_cctx` / `with_dctx` (encode.rs:142, decode.rs:43).
pub struct Engine(*mut ffi::ZkEngine);
unsafe impl Send for Engine {}
impl Engine {
    pub fn new(device: i32) -> Result<Engine> {
        let mut e = core::ptr::null_mut();
        check(unsafe { ffi::zk_engine_create(device, &mut e) })?;
        Ok(Engine(e))
    }
    pub fn as_ptr(&self) -> *mut ffi::ZkEngine { self.0 }
}
impl Drop for Engine { fn drop(&mut self) { unsafe { ffi::zk_engine_destroy(self.0) } } }

/// seek_table.rs:228-241
#[derive(Clone, Copy, PartialEq, Eq, Debug, Default)]
pub enum Format { Head, #[default] Foot }
impl Format { fn raw(self) -> c_int { match self { Format::Head => ffi::ZK_FORMAT_HEAD, Format::Foot => ffi::ZK_FORMAT_FOOT } } }

/// seek_table.rs:243-935
pub struct SeekTable(*mut ffi::ZkSeekTable);
unsafe impl Send for SeekTable {}
impl SeekTable {
    pub fn new() -> Self { SeekTable(unsafe { ffi::zk_seek_table_new() }) }                                           // :287
    pub fn from_bytes(src: &[u8], format: Format) -> Result<Self> {                                                 // from_seekable_format :379
        let mut t = core::ptr::null_mut();
        check(unsafe { ffi::zk_seek_table_from_bytes(src.as_ptr(), src.len(), format.raw(), &mut t) })?;
        Ok(SeekTable(t))
    }
    pub fn log_frame(&mut self, c_size: u32, d_size: u32) -> Result<()> { check(unsafe { ffi::zk_seek_table_log_frame(self.0, c_size, d_size) }) }   // :513
    /// engine-specific: the entries of a whole batch (what `zk_encode_frames*` returns) in one call
    pub fn log_frames(&mut self, c_sizes: &[u32], d_sizes: &[u32]) -> Result<()> {
        assert_eq!(c_sizes.len(), d_sizes.len());
        check(unsafe { ffi::zk_seek_table_log_frames(self.0, c_sizes.len() as u32, c_sizes.as_ptr(), d_sizes.as_ptr()) })
    }
    pub fn num_frames(&self) -> u32 { unsafe { ffi::zk_seek_table_num_frames(self.0) } }                              // :540
    pub fn frame_index_comp(&self, offset: u64) -> u32 { unsafe { ffi::zk_seek_table_frame_index_comp(self.0, offset) } }     // :560
    pub fn frame_index_decomp(&self, offset: u64) -> u32 { unsafe { ffi::zk_seek_table_frame_index_decomp(self.0, offset) } } // :579
    pub fn frame_start_comp(&self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_seek_table_frame_start_comp(self.0, i, &mut v) })?; Ok(v) }     // :604
    pub fn frame_start_decomp(&self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_seek_table_frame_start_decomp(self.0, i, &mut v) })?; Ok(v) } // :633
    pub fn frame_end_comp(&self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_seek_table_frame_end_comp(self.0, i, &mut v) })?; Ok(v) }         // :662
    pub fn frame_end_decomp(&self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_seek_table_frame_end_decomp(self.0, i, &mut v) })?; Ok(v) }     // :691
    pub fn frame_size_comp(&self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_seek_table_frame_size_comp(self.0, i, &mut v) })?; Ok(v) }       // :720
    pub fn frame_size_decomp(&self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_seek_table_frame_size_decomp(self.0, i, &mut v) })?; Ok(v) }   // :750
    pub fn max_frame_size_comp(&self) -> u64 { unsafe { ffi::zk_seek_table_max_frame_size_comp(self.0) } }            // :774
    pub fn max_frame_size_decomp(&self) -> u64 { unsafe { ffi::zk_seek_table_max_frame_size_decomp(self.0) } }        // :799
    pub fn size_comp(&self) -> u64 { unsafe { ffi::zk_seek_table_size_comp(self.0) } }                                // :827
    pub fn size_decomp(&self) -> u64 { unsafe { ffi::zk_seek_table_size_decomp(self.0) } }                            // :853
    /// seek_table.rs:338: the table at the END of a seekable source
    pub fn from_seekable(src: &mut impl Seekable) -> Result<Self> { Self::from_seekable_format(src, Format::Foot) }
    /// seek_table.rs:379-436.  The integrity field is the source's to find (`Seekable::seek_table_integrity`); its checks come in
    /// upstream's order (magic -> prefix_unknown, reserved descriptor bits -> corruption_detected, frame count), then the table's
    /// bytes are read where the count says they lie and parsed by the engine's host side (every remaining check: host/seek_table.cpp).
    pub fn from_seekable_format(src: &mut impl Seekable, format: Format) -> Result<Self> {
        let integ = src.seek_table_integrity(format)?;
        let (n, entry) = Self::integrity(&integ)?;
        let size = entry * n as usize + SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE;
        match format {
            Format::Head => src.set_offset(OffsetFrom::Start(0))?,
            Format::Foot => src.set_offset(OffsetFrom::End(-i64::try_from(size)?))?,
        };
        let mut bytes = vec![0u8; size];
        let mut at = 0;
        while at < size {
            let k = src.read(&mut bytes[at..])?;
            if k == 0 { return Err(Error::zstd(20)); }                  // the source ends inside the table: corruption_detected
            at += k;
        }
        Self::from_bytes(&bytes, format)
    }
    /// seek_table.rs:461-493: a table in Head format out of a plain reader (a side file, cli/src/compress.rs:86-96)
    pub fn from_reader(reader: &mut impl Read) -> Result<Self> {
        let mut bytes = vec![0u8; SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE];
        reader.read_exact(&mut bytes)?;
        let mut integ = [0u8; SEEK_TABLE_INTEGRITY_SIZE];
        integ.copy_from_slice(&bytes[SKIPPABLE_HEADER_SIZE..]);
        let (n, entry) = Self::integrity(&integ)?;
        let at = bytes.len();
        bytes.resize(at + entry * n as usize, 0);
        reader.read_exact(&mut bytes[at..])?;
        Self::from_bytes(&bytes, Format::Head)
    }
    /// the integrity field (seekable_format.md:96-132): Number_Of_Frames | Seek_Table_Descriptor | Seekable_Magic_Number -> (frames, bytes per entry)
    fn integrity(f: &[u8; SEEK_TABLE_INTEGRITY_SIZE]) -> Result<(u32, usize)> {
        if u32::from_le_bytes([f[5], f[6], f[7], f[8]]) != SEEKABLE_MAGIC_NUMBER { return Err(Error::zstd(10)); }   // prefix_unknown (seek_table.rs:146)
        if (f[4] >> 2) & 0x1f != 0 { return Err(Error::zstd(20)); }                                              // corruption_detected (:151)
        let n = u32::from_le_bytes([f[0], f[1], f[2], f[3]]);
        if n > SEEKABLE_MAX_FRAMES { return Err(Error::FrameIndexTooLarge); }
        Ok((n, if f[4] & 0x80 != 0 { 12 } else { 8 }))                                                            // legacy entries carry a checksum (:154-160)
    }
    /// seek_table.rs:883 / 907: the table turned into its serializer (Foot unless told otherwise)
    pub fn into_serializer(self) -> seek_table::Serializer { self.into_format_serializer(Format::Foot) }
    pub fn into_format_serializer(self, format: Format) -> seek_table::Serializer {
        seek_table::Serializer(unsafe { ffi::zk_seek_table_serializer(self.0, format.raw()) })
    }
    /// into_format_serializer(..) drained into a Vec (seek_table.rs:907, 967-1005)
    pub fn to_bytes(&self, format: Format) -> Vec<u8> {
        let s = unsafe { ffi::zk_seek_table_serializer(self.0, format.raw()) };
        let mut out = vec![0u8; unsafe { ffi::zk_serializer_encoded_len(s) }];
        let mut at = 0;
        loop {
            let n = unsafe { ffi::zk_serializer_write_into(s, out[at..].as_mut_ptr(), out.len() - at) };
            if n == 0 { break; }
            at += n;
        }
        unsafe { ffi::zk_serializer_free(s) };
        out
    }
}
impl Clone for SeekTable { fn clone(&self) -> Self { SeekTable(unsafe { ffi::zk_seek_table_clone(self.0) }) } }
impl Default for SeekTable { fn default() -> Self { Self::new() } }                                                  // :271
impl Eq for SeekTable {}                                                                                           // :266
impl core::fmt::Debug for SeekTable {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        f.debug_struct("SeekTable").field("num_frames", &self.num_frames()).field("size_comp", &self.size_comp()).field("size_decomp", &self.size_decomp()).finish()
    }
}
/// the public module of upstream (lib.rs:36: `pub mod seek_table`): `Format`, `SeekTable` and the `Serializer`
pub mod seek_table {
    pub use super::{Format, SeekTable};
    use super::ffi;
    use std::io;
    /// seek_table.rs:955-1059: the table's bytes, handed out piece by piece into buffers of any size
    pub struct Serializer(pub(super) *mut ffi::ZkSerializer);
    unsafe impl Send for Serializer {}
    impl Serializer {
        /// :967-1005 -- fills `buf` as far as it goes; 0 = everything has been written
        pub fn write_into(&mut self, buf: &mut [u8]) -> usize { unsafe { ffi::zk_serializer_write_into(self.0, buf.as_mut_ptr(), buf.len()) } }
        pub fn reset(&mut self) { unsafe { ffi::zk_serializer_reset(self.0) } }                        // :1034
        pub fn encoded_len(&self) -> usize { unsafe { ffi::zk_serializer_encoded_len(self.0) } }       // :1042
    }
    impl io::Read for Serializer { fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> { Ok(self.write_into(buf)) } }   // :1055-1059
    impl Drop for Serializer { fn drop(&mut self) { unsafe { ffi::zk_serializer_free(self.0) } } }
}
impl PartialEq for SeekTable { fn eq(&self, o: &Self) -> bool { unsafe { ffi::zk_seek_table_equal(self.0, o.0) != 0 } } }
impl Drop for SeekTable { fn drop(&mut self) { unsafe { ffi::zk_seek_table_free(self.0) } } }

// ------------------------------------------------------------------------------------------------ decode (decode.rs)
/// seekable.rs:8-13
#[derive(Debug, Clone, Copy)]
pub enum OffsetFrom { Start(u64), End(i64) }
impl From<OffsetFrom> for SeekFrom {                                                                    // seekable.rs:100-109
    fn from(o: OffsetFrom) -> Self { match o { OffsetFrom::Start(n) => SeekFrom::Start(n), OffsetFrom::End(n) => SeekFrom::End(n) } }
}
/// seekable.rs:42-97: a byte slice as a seekable source (what `Decoder::new(BytesWrapper::new(&archive))` reads from)
#[derive(Debug, Clone)]
pub struct BytesWrapper<'a> { bytes: &'a [u8], at: usize }
impl<'a> BytesWrapper<'a> {
    pub fn new(src: &'a [u8]) -> Self { BytesWrapper { bytes: src, at: 0 } }                                 // :50
}
impl Seekable for BytesWrapper<'_> {
    /// an offset outside [0, len] is `OffsetOutOfRange`, from either end (seekable.rs:56-74)
    fn set_offset(&mut self, offset: OffsetFrom) -> Result<u64> {
        let len = self.bytes.len() as i128;
        let target = match offset { OffsetFrom::Start(n) => n as i128, OffsetFrom::End(d) => len + d as i128 };
        if target < 0 || target > len { return Err(Error::OffsetOutOfRange); }
        self.at = target as usize;
        Ok(target as u64)
    }
    fn read(&mut self, buf: &mut [u8]) -> Result<usize> {                                                 // :76-82
        let k = buf.len().min(self.bytes.len() - self.at);
        buf[..k].copy_from_slice(&self.bytes[self.at..self.at + k]);
        self.at += k;
        Ok(k)
    }
    /// Head: the 9 bytes behind the skippable header; Foot: the last 9 bytes; a slice too short for them is out of range (:84-96)
    fn seek_table_integrity(&mut self, format: Format) -> Result<[u8; SEEK_TABLE_INTEGRITY_SIZE]> {
        let start = match format {
            Format::Head if self.bytes.len() >= SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE => SKIPPABLE_HEADER_SIZE,
            Format::Foot if self.bytes.len() >= SEEK_TABLE_INTEGRITY_SIZE => self.bytes.len() - SEEK_TABLE_INTEGRITY_SIZE,
            _ => return Err(Error::OffsetOutOfRange),
        };
        let mut out = [0u8; SEEK_TABLE_INTEGRITY_SIZE];
        out.copy_from_slice(&self.bytes[start..start + SEEK_TABLE_INTEGRITY_SIZE]);
        Ok(out)
    }
}
/// seekable.rs:16-39.  The blanket impl below gives it to every `Read + Seek` (seekable.rs:112-138).
pub trait Seekable {
    fn set_offset(&mut self, offset: OffsetFrom) -> Result<u64>;
    fn read(&mut self, buf: &mut [u8]) -> Result<usize>;
    /// required, like upstream (seekable.rs:33-38): a source may keep its integrity field anywhere
    fn seek_table_integrity(&mut self, format: Format) -> Result<[u8; 9]>;
}
impl<T: Read + Seek> Seekable for T {
    fn set_offset(&mut self, offset: OffsetFrom) -> Result<u64> {
        Ok(self.seek(match offset { OffsetFrom::Start(n) => SeekFrom::Start(n), OffsetFrom::End(n) => SeekFrom::End(n) })?)
    }
    fn seek_table_integrity(&mut self, format: Format) -> Result<[u8; 9]> {                                 // seekable.rs:126-137
        match format { Format::Head => self.seek(SeekFrom::Start(8))?, Format::Foot => self.seek(SeekFrom::End(-9))? };
        let mut buf = [0u8; 9];
        self.read_exact(&mut buf)?;
        Ok(buf)
    }
    fn read(&mut self, buf: &mut [u8]) -> Result<usize> { Ok(Read::read(self, buf)?) }
}
unsafe extern "C" fn seek_cb<S: Seekable>(user: *mut c_void, whence: c_int, value: i64) -> i64 {
    let s = &mut *(user as *mut S);
    match s.set_offset(if whence == 0 { OffsetFrom::Start(value as u64) } else { OffsetFrom::End(value) }) { Ok(p) => p as i64, Err(_) => -1 }
}
unsafe extern "C" fn read_cb<S: Seekable>(user: *mut c_void, buf: *mut u8, len: usize) -> i64 {
    let s = &mut *(user as *mut S);
    match s.read(core::slice::from_raw_parts_mut(buf, len)) { Ok(n) => n as i64, Err(_) => -1 }
}
/// Seekable::seek_table_integrity is a required method of the trait (seekable.rs:33-38): the source's own answer is used
unsafe extern "C" fn integrity_cb<S: Seekable>(user: *mut c_void, format: c_int, out: *mut u8) -> c_int {
    let s = &mut *(user as *mut S);
    match s.seek_table_integrity(if format == 0 { Format::Head } else { Format::Foot }) {
        Ok(a) => { core::ptr::copy_nonoverlapping(a.as_ptr(), out, a.len()); 0 }
        Err(_) => -1,
    }
}

/// decode.rs:13-114
pub struct DecodeOptions<'a, S: Seekable> {
    src: S, engine: Option<&'a Engine>, seek_table: Option<SeekTable>,
    lower_frame: Option<u32>, upper_frame: Option<u32>, offset: Option<u64>, offset_limit: Option<u64>,
}
impl<'a, S: Seekable> DecodeOptions<'a, S> {
    pub fn new(src: S) -> Self { DecodeOptions { src, engine: None, seek_table: None, lower_frame: None, upper_frame: None, offset: None, offset_limit: None } }   // :30
    pub fn try_new(src: S) -> Option<Self> { Some(Self::new(src)) }                                         // :37 (nothing to create yet: see EncodeOptions::try_new)
    pub fn engine(mut self, e: &'a Engine) -> Self { self.engine = Some(e); self }                          // with_dctx :43
    pub fn seek_table(mut self, t: SeekTable) -> Self { self.seek_table = Some(t); self }                   // :65
    pub fn lower_frame(mut self, i: u32) -> Self { self.lower_frame = Some(i); self }                       // :73
    pub fn upper_frame(mut self, i: u32) -> Self { self.upper_frame = Some(i); self }                       // :81
    pub fn offset(mut self, o: u64) -> Self { self.offset = Some(o); self }                                 // :90
    pub fn offset_limit(mut self, l: u64) -> Self { self.offset_limit = Some(l); self }                     // :99
    pub fn into_decoder(self) -> Result<Decoder<'a, S>> { Decoder::with_opts(self) }                        // :111
}

/// decode.rs:117-466.  The source is boxed so that its address stays put behind the C callbacks.
pub struct Decoder<'a, S: Seekable> { h: *mut ffi::ZkDecoder, _src: Box<S>, _table: Option<SeekTable>, _e: core::marker::PhantomData<&'a Engine> }
impl<'a, S: Seekable> Decoder<'a, S> {
    pub fn new(src: S) -> Result<Self> { DecodeOptions::new(src).into_decoder() }                           // :143
    pub fn with_opts(o: DecodeOptions<'a, S>) -> Result<Self> {                                             // :152-187
        let mut src = Box::new(o.src);
        let mut flags = 0u32;
        if o.offset.is_some() { flags |= ffi::ZK_DEC_HAS_OFFSET; }
        if o.offset_limit.is_some() { flags |= ffi::ZK_DEC_HAS_OFFSET_LIMIT; }
        if o.lower_frame.is_some() { flags |= ffi::ZK_DEC_HAS_LOWER_FRAME; }
        if o.upper_frame.is_some() { flags |= ffi::ZK_DEC_HAS_UPPER_FRAME; }
        let opts = ffi::ZkDecodeOpts {
            flags, lower_frame: o.lower_frame.unwrap_or(0), upper_frame: o.upper_frame.unwrap_or(0), offset: o.offset.unwrap_or(0),
            offset_limit: o.offset_limit.unwrap_or(0), seek_table: o.seek_table.as_ref().map_or(core::ptr::null(), |t| t.0 as *const _), batch_bytes: 0,
        };
        let mut h = core::ptr::null_mut();
        check(unsafe {
            ffi::zk_decoder_open_seekable(o.engine.map_or(core::ptr::null_mut(), |e| e.as_ptr()), Some(seek_cb::<S>), Some(read_cb::<S>),
                                          Some(integrity_cb::<S>), &mut *src as *mut S as *mut c_void, &opts, &mut h)
        })?;
        Ok(Decoder { h, _src: src, _table: o.seek_table, _e: core::marker::PhantomData })
    }
    pub fn decompress(&mut self, buf: &mut [u8]) -> Result<usize> { check_len(unsafe { ffi::zk_decoder_decompress(self.h, buf.as_mut_ptr(), buf.len()) }) }   // :314
    pub fn decompress_with_prefix<'b: 'a>(&mut self, buf: &mut [u8], prefix: Option<&'b [u8]>) -> Result<usize> {                                           // :201
        let (p, n) = prefix.map_or((core::ptr::null(), 0), |p| (p.as_ptr(), p.len()));
        let mut out = 0usize;
        check(unsafe { ffi::zk_decoder_decompress_with_prefix(self.h, buf.as_mut_ptr(), buf.len(), p, n, &mut out) })?;
        Ok(out)
    }
    pub fn reset(&mut self) { unsafe { ffi::zk_decoder_reset(self.h) } }                                                                   // :346
    pub fn set_lower_frame(&mut self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_decoder_set_lower_frame(self.h, i, &mut v) })?; Ok(v) }   // :367
    pub fn set_upper_frame(&mut self, i: u32) -> Result<u64> { let mut v = 0; check(unsafe { ffi::zk_decoder_set_upper_frame(self.h, i, &mut v) })?; Ok(v) }   // :383
    pub fn set_offset(&mut self, o: u64) -> Result<()> { check(unsafe { ffi::zk_decoder_set_offset(self.h, o) }) }                         // :402
    pub fn set_offset_limit(&mut self, l: u64) -> Result<()> { check(unsafe { ffi::zk_decoder_set_offset_limit(self.h, l) }) }             // :432
    pub fn read_compressed(&self) -> u64 { unsafe { ffi::zk_decoder_read_compressed(self.h) } }                                            // :448
    pub fn seek_table(&self) -> SeekTable { SeekTable(unsafe { ffi::zk_decoder_seek_table(self.h) }) }                                     // :453
    pub fn offset(&self) -> u64 { unsafe { ffi::zk_decoder_offset(self.h) } }                                                              // :458
    pub fn offset_limit(&self) -> u64 { unsafe { ffi::zk_decoder_offset_limit(self.h) } }                                                  // :463
}
impl<S: Seekable> Read for Decoder<'_, S> {                                                                                               // :510-514
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> { self.decompress(buf).map_err(|e| io::Error::new(io::ErrorKind::Other, format!("{e:?}"))) }
}
impl<S: Seekable> Seek for Decoder<'_, S> {                                                                                               // :545-579
    fn seek(&mut self, pos: SeekFrom) -> io::Result<u64> {
        let (w, n) = match pos { SeekFrom::Start(n) => (ffi::ZK_SEEK_START, n as i64), SeekFrom::End(n) => (ffi::ZK_SEEK_END, n), SeekFrom::Current(n) => (ffi::ZK_SEEK_CURRENT, n) };
        let mut v = 0;
        check(unsafe { ffi::zk_decoder_seek(self.h, w, n, &mut v) }).map_err(|e| io::Error::new(io::ErrorKind::InvalidInput, format!("{e:?}")))?;
        Ok(v)
    }
}
impl<S: Seekable> Drop for Decoder<'_, S> { fn drop(&mut self) { unsafe { ffi::zk_decoder_free(self.h) } } }

// ------------------------------------------------------------------------------------------------ encode (encode.rs)
/// encode.rs:21-39
#[derive(Clone, Copy)]
pub enum FrameSizePolicy { Compressed(u32), Uncompressed(u32) }
impl Default for FrameSizePolicy { fn default() -> Self { FrameSizePolicy::Uncompressed(0x200000) } }

/// encode.rs:110-207
#[derive(Default)]
pub struct EncodeOptions<'a> { engine: Option<&'a Engine>, policy: FrameSizePolicy, checksum: bool, level: i32 }
impl<'a> EncodeOptions<'a> {
    pub fn new() -> Self { Self::default() }                                                            // :129
    /// :136 -- upstream's fallible constructor creates the compression context; here the engine is created (or injected) when the
    /// encoder is made, so there is nothing that can fail yet
    pub fn try_new() -> Option<Self> { Some(Self::default()) }
    pub fn engine(mut self, e: &'a Engine) -> Self { self.engine = Some(e); self }                       // with_cctx :142
    pub fn frame_size_policy(mut self, p: FrameSizePolicy) -> Self { self.policy = p; self }             // :158
    pub fn checksum_flag(mut self, f: bool) -> Self { self.checksum = f; self }                          // :164
    pub fn compression_level(mut self, l: i32) -> Self { self.level = l; self }                          // :170
    fn raw(&self) -> ffi::ZkEncodeOpts {
        let (policy, frame_size) = match self.policy { FrameSizePolicy::Uncompressed(n) => (ffi::ZK_POLICY_UNCOMPRESSED, n), FrameSizePolicy::Compressed(n) => (ffi::ZK_POLICY_COMPRESSED, n) };
        ffi::ZkEncodeOpts { policy, frame_size, level: self.level, checksum: self.checksum as i32, batch_frames: 0 }
    }
    pub fn into_raw_encoder(self) -> Result<RawEncoder<'a>> {                                            // :180
        let mut h = core::ptr::null_mut();
        check(unsafe { ffi::zk_raw_encoder_new(self.engine.map_or(core::ptr::null_mut(), |e| e.as_ptr()), &self.raw(), &mut h) })?;
        Ok(RawEncoder { h, _e: core::marker::PhantomData })
    }
    pub fn into_encoder<W: Write>(self, writer: W) -> Result<Encoder<'a, W>> {                           // :204
        let mut w = Box::new(writer);
        let mut h = core::ptr::null_mut();
        check(unsafe { ffi::zk_encoder_new(self.engine.map_or(core::ptr::null_mut(), |e| e.as_ptr()), &self.raw(), Some(write_cb::<W>), &mut *w as *mut W as *mut c_void, &mut h) })?;
        Ok(Encoder { h, writer: w, _e: core::marker::PhantomData })
    }
}
unsafe extern "C" fn write_cb<W: Write>(user: *mut c_void, data: *const u8, len: usize) -> c_int {
    let w = &mut *(user as *mut W);
    match w.write_all(core::slice::from_raw_parts(data, len)) { Ok(()) => 0, Err(_) => 1 }
}

/// encode.rs:43-66 / 69-92
pub struct CompressionProgress { in_progress: usize, out_progress: usize }
impl CompressionProgress { pub fn in_progress(&self) -> usize { self.in_progress } pub fn out_progress(&self) -> usize { self.out_progress } }
pub struct EpilogueProgress { out_progress: usize, data_left: usize }
impl EpilogueProgress { pub fn out_progress(&self) -> usize { self.out_progress } pub fn data_left(&self) -> usize { self.data_left } }

/// encode.rs:209-545
pub struct RawEncoder<'a> { h: *mut ffi::ZkRawEncoder, _e: core::marker::PhantomData<&'a Engine> }
impl<'a> RawEncoder<'a> {
    pub fn new() -> Result<Self> { EncodeOptions::new().into_raw_encoder() }                              // :365
    pub fn with_opts(opts: EncodeOptions<'a>) -> Result<Self> { opts.into_raw_encoder() }                 // :280
    pub fn into_seek_table(self) -> SeekTable { self.seek_table() }                                       // :492
    pub fn compress(&mut self, input: &[u8], output: &mut [u8]) -> Result<CompressionProgress> { self.compress_with_prefix(input, output, None) }   // :398
    pub fn compress_with_prefix<'b: 'a>(&mut self, input: &[u8], output: &mut [u8], prefix: Option<&'b [u8]>) -> Result<CompressionProgress> {      // :311
        let (p, n) = prefix.map_or((core::ptr::null(), 0), |p| (p.as_ptr(), p.len()));
        let (mut i, mut o) = (0usize, 0usize);
        check(unsafe { ffi::zk_raw_encoder_compress_with_prefix(self.h, input.as_ptr(), input.len(), output.as_mut_ptr(), output.len(), p, n, &mut i, &mut o) })?;
        Ok(CompressionProgress { in_progress: i, out_progress: o })
    }
    pub fn end_frame(&mut self, output: &mut [u8]) -> Result<EpilogueProgress> {                          // :438
        let (mut o, mut left) = (0usize, 0usize);
        check(unsafe { ffi::zk_raw_encoder_end_frame(self.h, output.as_mut_ptr(), output.len(), &mut o, &mut left) })?;
        Ok(EpilogueProgress { out_progress: o, data_left: left })
    }
    pub fn seek_table(&self) -> SeekTable { SeekTable(unsafe { ffi::zk_raw_encoder_seek_table(self.h) }) } // :487
    pub fn reset_frame(&mut self) { unsafe { ffi::zk_raw_encoder_reset_frame(self.h) } }                  // :501
    pub fn reset_seek_table(&mut self) { unsafe { ffi::zk_raw_encoder_reset_seek_table(self.h) } }        // :524
}
impl Drop for RawEncoder<'_> { fn drop(&mut self) { unsafe { ffi::zk_raw_encoder_free(self.h) } } }

/// encode.rs:570-800.  The writer is boxed so that its address stays put behind the C callback.
pub struct Encoder<'a, W: Write> { h: *mut ffi::ZkEncoder, writer: Box<W>, _e: core::marker::PhantomData<&'a Engine> }
impl<'a, W: Write> Encoder<'a, W> {
    pub fn new(writer: W) -> Result<Self> { EncodeOptions::new().into_encoder(writer) }                   // :587
    pub fn with_opts(writer: W, opts: EncodeOptions<'a>) -> Result<Self> { opts.into_encoder(writer) }    // :596
    pub fn into_seek_table(self) -> SeekTable { self.seek_table() }                                       // :620
    pub fn compress(&mut self, buf: &[u8]) -> Result<usize> { check_len(unsafe { ffi::zk_encoder_compress(self.h, buf.as_ptr(), buf.len()) }) }     // :692
    pub fn compress_with_prefix<'b: 'a>(&mut self, buf: &[u8], prefix: Option<&'b [u8]>) -> Result<usize> {                                     // :641
        let (p, n) = prefix.map_or((core::ptr::null(), 0), |p| (p.as_ptr(), p.len()));
        check_len(unsafe { ffi::zk_encoder_compress_with_prefix(self.h, buf.as_ptr(), buf.len(), p, n) })
    }
    pub fn end_frame(&mut self) -> Result<usize> { check_len(unsafe { ffi::zk_encoder_end_frame(self.h) }) }                                   // :704
    pub fn finish(self) -> Result<u64> { self.finish_format(Format::Foot) }                                                                    // :743
    pub fn finish_format(self, format: Format) -> Result<u64> { let mut t = 0; check(unsafe { ffi::zk_encoder_finish(self.h, format.raw(), &mut t) })?; Ok(t) }   // :755
    pub fn written_compressed(&self) -> u64 { unsafe { ffi::zk_encoder_written_compressed(self.h) } }                                          // :615
    pub fn seek_table(&self) -> SeekTable { SeekTable(unsafe { ffi::zk_encoder_seek_table(self.h) }) }                                         // :610
    pub fn get_ref(&self) -> &W { &self.writer }
}
impl<W: Write> Write for Encoder<'_, W> {                                                                                                     // :791-799
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> { self.compress(buf).map_err(|e| io::Error::new(io::ErrorKind::Other, format!("{e:?}"))) }
    fn flush(&mut self) -> io::Result<()> { check(unsafe { ffi::zk_encoder_flush(self.h) }).map_err(|e| io::Error::new(io::ErrorKind::Other, format!("{e:?}"))) }
}
impl<W: Write> Drop for Encoder<'_, W> { fn drop(&mut self) { unsafe { ffi::zk_encoder_free(self.h) } } }

// ------------------------------------------------------------------------------------------------ Level A, device resident
/// N frames in, N frames out with everything in HBM (`hipMalloc` pointers, a `hipStream_t` or null): the batch engine a
/// serving pipeline drives directly.  See include/zeekstd_amd.h for every argument.
pub mod batch {
    pub use crate::ffi::{zk_compress_bound, zk_decode_frame_list_dev, zk_decode_frames, zk_decode_frames_dev, zk_decode_frames_prefix,
                         zk_decode_frames_prefix_dev, zk_decode_submit_dev, zk_decode_wait, zk_encode_frames, zk_encode_frames_dev,
                         zk_decode_shard, zk_encode_frames_prefix, zk_encode_frames_prefix_dev, zk_gather_seekable, zk_host_alloc, zk_host_free, zk_shard_range,
                         zk_xxh64_frames, zk_xxh64_frames_dev};
}
