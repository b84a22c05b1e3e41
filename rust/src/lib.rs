//! zeekstd's public types on the MI355X engine: `EncodeOptions / Encoder<W> / DecodeOptions / Decoder<S> / SeekTable`
//! with the reference's method names, argument meaning and error behaviour (citations: /root/reference/lib/src), over the C
//! ABI of `libzeekstd_amd.so` (`ffi`, generated from `include/zeekstd_amd.h`).  The frame hot path -- what the reference
//! does through `ZSTD_compressStream2` / `ZSTD_decompressStream` (encode.rs:340-346, decode.rs:242-256) -- runs as HIP
//! kernels; seek table, offsets and sources keep their reference semantics.
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

impl Error {
    pub fn is_offset_out_of_range(&self) -> bool { matches!(self, Error::OffsetOutOfRange) }       // error.rs:25
    pub fn is_frame_index_too_large(&self) -> bool { matches!(self, Error::FrameIndexTooLarge) }   // error.rs:36
    pub fn is_zstd(&self) -> bool { matches!(self, Error::Zstd(_)) }                               // error.rs:55
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
impl From<io::Error> for Error { fn from(e: io::Error) -> Self { Error::IO(e) } }
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
impl PartialEq for SeekTable { fn eq(&self, o: &Self) -> bool { unsafe { ffi::zk_seek_table_equal(self.0, o.0) != 0 } } }
impl Drop for SeekTable { fn drop(&mut self) { unsafe { ffi::zk_seek_table_free(self.0) } } }

// ------------------------------------------------------------------------------------------------ decode (decode.rs)
/// seekable.rs:8-13
pub enum OffsetFrom { Start(u64), End(i64) }
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
