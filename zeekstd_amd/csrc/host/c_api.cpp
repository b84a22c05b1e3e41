// c_api.cpp -- Level B of include/zeekstd_amd.h: C handles over the zeekstd:: host classes.
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <new>
#include "../../../include/zeekstd_amd.h"
#include "zeekstd.hpp"
#include "../zk_engine.h"

using namespace zeekstd;

struct zk_seek_table { SeekTable t; };
struct zk_serializer { Serializer s; };
struct zk_decoder { Decoder d; explicit zk_decoder(Decoder &&x) : d(std::move(x)) {} };
struct zk_raw_encoder { RawEncoder r; explicit zk_raw_encoder(RawEncoder &&x) : r(std::move(x)) {} };
struct CallbackWriter : Writer {
    zk_write_fn fn; void *user;
    CallbackWriter(zk_write_fn f, void *u) : fn(f), user(u) {}
    void write_all(const uint8_t *p, size_t n) override { if (n && fn(user, p, n) != 0) throw Error::io("writer callback failed"); }
};
struct zk_encoder { Encoder e; zk_encoder(std::shared_ptr<Writer> w, EncodeOptions &&o) : e(std::move(w), std::move(o)) {} };

static thread_local std::string g_last_error;

template <typename F>
static int guard(F &&f)
{
    try { f(); return 0; }
    catch (const Error &e) { g_last_error = e.what(); return e.abi_code(); }
    catch (const std::bad_alloc &) { g_last_error = "allocation failed"; return -64; }
    catch (const std::exception &e) { g_last_error = e.what(); return -1; }
}

// the gather of zk_engine_gather.hip hands its table over as a C handle
zk_seek_table *zk_seek_table_from_cpp(const zeekstd::SeekTable *t) { return new (std::nothrow) zk_seek_table{*t}; }

extern "C" {

const char *zk_last_error_message(void) { return g_last_error.c_str(); }

int zk_buffer_writer_write(void *user, const uint8_t *data, size_t len)
{
    zk_buffer_writer *w = (zk_buffer_writer *)user;
    if (!w || w->len + len > w->cap) return 1;
    if (w->engine) zk_host_copy(w->engine, w->data + w->len, data, len);
    else memcpy(w->data + w->len, data, len);
    w->len += len;
    return 0;
}

// ---------------------------------------------------------------- SeekTable
zk_seek_table *zk_seek_table_new(void) { return new (std::nothrow) zk_seek_table(); }
void zk_seek_table_free(zk_seek_table *t) { delete t; }
zk_seek_table *zk_seek_table_clone(const zk_seek_table *t) { return t ? new (std::nothrow) zk_seek_table(*t) : nullptr; }

int zk_seek_table_from_bytes(const uint8_t *src, size_t len, int format, zk_seek_table **out)
{
    if (!out) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] {
        BytesWrapper w(src, len);
        SeekTable t = SeekTable::from_seekable_format(w, format == ZK_FORMAT_HEAD ? Format::Head : Format::Foot);
        *out = new zk_seek_table{std::move(t)};
    });
}

int zk_seek_table_from_reader_bytes(const uint8_t *p, size_t len, size_t max_read, zk_seek_table **out)
{
    if (!out) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    struct R : SeekTable::Reader {
        const uint8_t *p; size_t len, pos = 0, cap;
        size_t read(uint8_t *buf, size_t n) override
        {
            n = std::min({n, len - pos, cap ? cap : n});
            memcpy(buf, p + pos, n); pos += n; return n;
        }
    } r;
    r.p = p; r.len = len; r.cap = max_read;
    return guard([&] { *out = new zk_seek_table{SeekTable::from_reader(r)}; });
}

int zk_seek_table_log_frame(zk_seek_table *t, uint32_t c_size, uint32_t d_size) { return guard([&] { t->t.log_frame(c_size, d_size); }); }
int zk_seek_table_log_frames(zk_seek_table *t, uint32_t n, const uint32_t *c_sizes, const uint32_t *d_sizes)
{
    if (!t || (n && (!c_sizes || !d_sizes))) return ZK_ERR_ARGUMENT;
    return guard([&] { for (uint32_t i = 0; i < n; i++) t->t.log_frame(c_sizes[i], d_sizes[i]); });
}
uint32_t zk_seek_table_num_frames(const zk_seek_table *t) { return t->t.num_frames(); }
uint32_t zk_seek_table_frame_index_comp(const zk_seek_table *t, uint64_t off) { return t->t.frame_index_comp(off); }
uint32_t zk_seek_table_frame_index_decomp(const zk_seek_table *t, uint64_t off) { return t->t.frame_index_decomp(off); }
int zk_seek_table_frame_start_comp(const zk_seek_table *t, uint32_t i, uint64_t *out) { return guard([&] { *out = t->t.frame_start_comp(i); }); }
int zk_seek_table_frame_start_decomp(const zk_seek_table *t, uint32_t i, uint64_t *out) { return guard([&] { *out = t->t.frame_start_decomp(i); }); }
int zk_seek_table_frame_end_comp(const zk_seek_table *t, uint32_t i, uint64_t *out) { return guard([&] { *out = t->t.frame_end_comp(i); }); }
int zk_seek_table_frame_end_decomp(const zk_seek_table *t, uint32_t i, uint64_t *out) { return guard([&] { *out = t->t.frame_end_decomp(i); }); }
int zk_seek_table_frame_size_comp(const zk_seek_table *t, uint32_t i, uint64_t *out) { return guard([&] { *out = t->t.frame_size_comp(i); }); }
int zk_seek_table_frame_size_decomp(const zk_seek_table *t, uint32_t i, uint64_t *out) { return guard([&] { *out = t->t.frame_size_decomp(i); }); }
uint64_t zk_seek_table_max_frame_size_comp(const zk_seek_table *t) { return t->t.max_frame_size_comp(); }
uint64_t zk_seek_table_max_frame_size_decomp(const zk_seek_table *t) { return t->t.max_frame_size_decomp(); }
uint64_t zk_seek_table_size_comp(const zk_seek_table *t) { return t->t.size_comp(); }
uint64_t zk_seek_table_size_decomp(const zk_seek_table *t) { return t->t.size_decomp(); }
int zk_seek_table_equal(const zk_seek_table *a, const zk_seek_table *b) { return a->t == b->t; }
size_t zk_seek_table_entries(const zk_seek_table *t, uint64_t *c_off, uint64_t *d_off, size_t cap)
{
    const auto &e = t->t.entries();
    for (size_t i = 0; i < e.size() && i < cap; i++) { if (c_off) c_off[i] = e[i].c_offset; if (d_off) d_off[i] = e[i].d_offset; }
    return e.size();
}

zk_serializer *zk_seek_table_serializer(const zk_seek_table *t, int format)
{
    return new (std::nothrow) zk_serializer{t->t.into_format_serializer(format == ZK_FORMAT_HEAD ? Format::Head : Format::Foot)};
}
size_t zk_serializer_write_into(zk_serializer *s, uint8_t *buf, size_t len) { return s->s.write_into(buf, len); }
void zk_serializer_reset(zk_serializer *s) { s->s.reset(); }
size_t zk_serializer_encoded_len(const zk_serializer *s) { return s->s.encoded_len(); }
void zk_serializer_free(zk_serializer *s) { delete s; }

// ---------------------------------------------------------------- Decoder
static DecodeOptions make_opts(std::shared_ptr<Seekable> src, zk_engine *e, const zk_decode_opts *o)
{
    DecodeOptions opts(std::move(src));
    if (e) opts.engine(e);
    if (o) {
        if (o->seek_table) opts.seek_table(o->seek_table->t);
        if (o->flags & ZK_DEC_HAS_LOWER_FRAME) opts.lower_frame(o->lower_frame);
        if (o->flags & ZK_DEC_HAS_UPPER_FRAME) opts.upper_frame(o->upper_frame);
        if (o->flags & ZK_DEC_HAS_OFFSET) opts.offset(o->offset);
        if (o->flags & ZK_DEC_HAS_OFFSET_LIMIT) opts.offset_limit(o->offset_limit);
        if (o->flags & ZK_DEC_NO_VERIFY) opts.verify_checksums(false);
        if (o->batch_bytes) opts.batch_bytes(o->batch_bytes);
    }
    return opts;
}

int zk_decoder_open_bytes(zk_engine *e, const uint8_t *src, size_t len, const zk_decode_opts *o, zk_decoder **out)
{
    if (!out) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] { *out = new zk_decoder(Decoder(make_opts(std::make_shared<BytesWrapper>(src, len), e, o))); });
}

int zk_decoder_open_file(zk_engine *e, const char *path, const zk_decode_opts *o, zk_decoder **out)
{
    if (!out || !path) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] {
        FILE *f = fopen(path, "rb");
        if (!f) throw Error::io(std::string("cannot open ") + path);
        *out = new zk_decoder(Decoder(make_opts(std::make_shared<FileSeekable>(f, true), e, o)));
    });
}

int zk_decoder_open_callbacks(zk_engine *e, zk_seek_fn set_offset, zk_read_fn read, void *user, const zk_decode_opts *o, zk_decoder **out)
{
    if (!out || !set_offset || !read) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] { *out = new zk_decoder(Decoder(make_opts(std::make_shared<CallbackSeekable>(set_offset, read, user), e, o))); });
}

int zk_decoder_open_seekable(zk_engine *e, zk_seek_fn set_offset, zk_read_fn read, zk_integrity_fn integrity, void *user, const zk_decode_opts *o,
                             zk_decoder **out)
{
    if (!out || !set_offset || !read) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] { *out = new zk_decoder(Decoder(make_opts(std::make_shared<CallbackSeekable>(set_offset, read, user, integrity), e, o))); });
}

void zk_decoder_free(zk_decoder *d) { delete d; }

// contiguous frame range of a rank (the same split as zeekstd_amd/parallel.py shard_range: the first n % world ranks take one more)
int zk_shard_range(uint32_t n_frames, int rank, int world, uint32_t *first, uint32_t *count)
{
    if (world < 1 || rank < 0 || rank >= world || !first || !count) return ZK_ERR_ARGUMENT;
    const uint32_t per = n_frames / (uint32_t)world, extra = n_frames % (uint32_t)world, r = (uint32_t)rank;
    *first = r * per + (r < extra ? r : extra);
    *count = per + (r < extra ? 1u : 0u);
    return 0;
}

int zk_decode_shard(zk_engine *e, const uint8_t *comp_shard, uint64_t shard_bytes, const zk_seek_table *table, int rank, int world,
                    uint8_t *dst, uint64_t dst_cap, int verify, uint32_t *first_out, uint32_t *count_out, uint64_t *written)
{
    if (written) *written = 0;
    if (!e || !table || (shard_bytes && !comp_shard)) return ZK_ERR_ARGUMENT;
    uint32_t first = 0, count = 0;
    int rc = zk_shard_range(table->t.num_frames(), rank, world, &first, &count);
    if (rc) return rc;
    if (first_out) *first_out = first;
    if (count_out) *count_out = count;
    if (!count) return 0;
    const auto &E = table->t.entries();
    std::vector<uint64_t> c(count + 1), d(count + 1);
    for (uint32_t i = 0; i <= count; i++) { c[i] = E[first + i].c_offset - E[first].c_offset; d[i] = E[first + i].d_offset - E[first].d_offset; }
    if (c[count] > shard_bytes) return -72;                  /* srcSize_wrong: the shard does not hold its frames */
    if (d[count] > dst_cap) return -70;                      /* dstSize_tooSmall */
    rc = zk_decode_frames(e, comp_shard, c[count], c.data(), d.data(), 0, count, dst, dst_cap, verify, nullptr);
    if (rc == 0 && written) *written = d[count];
    return rc;
}

int zk_decoder_time_seeks(zk_decoder *d, const uint64_t *offs, const uint32_t *lens, uint32_t n, uint8_t *buf, size_t buf_len,
                          const uint8_t *expect, double *us_out)
{
    if (!d || !offs || !lens || !buf || !us_out) return ZK_ERR_ARGUMENT;
    int bad = 0;
    int rc = guard([&] {
        const uint64_t total = d->d.seek_table().size_decomp();
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t lim = std::min<uint64_t>(offs[i] + lens[i], total);
            const auto t0 = std::chrono::steady_clock::now();
            d->d.set_offset_limit(total);
            d->d.set_offset(offs[i]);
            d->d.set_offset_limit(lim);
            size_t got = 0;
            for (;;) {
                const size_t k = d->d.decompress(buf + got, buf_len - got);
                if (k == 0) break;
                got += k;
            }
            us_out[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (got != lim - offs[i] || (expect && memcmp(buf, expect + offs[i], got) != 0)) {
                us_out[i] = -us_out[i];                      // marks the seek; buf keeps what was delivered
                bad = 1; return;
            }
        }
    });
    return rc ? rc : bad ? ZK_ERR_ARGUMENT : 0;
}
int64_t zk_decoder_decompress(zk_decoder *d, uint8_t *buf, size_t len)
{
    int64_t n = 0;
    int rc = guard([&] { n = (int64_t)d->d.decompress(buf, len); });
    return rc ? rc : n;
}
int zk_decoder_decompress_with_prefix(zk_decoder *d, uint8_t *buf, size_t len, const uint8_t *prefix, size_t plen, size_t *out)
{
    return guard([&] { *out = d->d.decompress_with_prefix(buf, len, prefix, plen); });
}
void zk_decoder_reset(zk_decoder *d) { d->d.reset(); }
int zk_decoder_set_lower_frame(zk_decoder *d, uint32_t i, uint64_t *out) { return guard([&] { uint64_t v = d->d.set_lower_frame(i); if (out) *out = v; }); }
int zk_decoder_set_upper_frame(zk_decoder *d, uint32_t i, uint64_t *out) { return guard([&] { uint64_t v = d->d.set_upper_frame(i); if (out) *out = v; }); }
int zk_decoder_set_offset(zk_decoder *d, uint64_t off) { return guard([&] { d->d.set_offset(off); }); }
int zk_decoder_set_offset_limit(zk_decoder *d, uint64_t lim) { return guard([&] { d->d.set_offset_limit(lim); }); }
uint64_t zk_decoder_read_compressed(const zk_decoder *d) { return d->d.read_compressed(); }
uint64_t zk_decoder_offset(const zk_decoder *d) { return d->d.offset(); }
uint64_t zk_decoder_offset_limit(const zk_decoder *d) { return d->d.offset_limit(); }
uint64_t zk_decoder_gpu_submissions(const zk_decoder *d) { return d->d.gpu_submissions(); }
zk_seek_table *zk_decoder_seek_table(const zk_decoder *d) { return new (std::nothrow) zk_seek_table{d->d.seek_table()}; }
int zk_decoder_seek(zk_decoder *d, int whence, int64_t n, uint64_t *out)
{
    return guard([&] {
        Decoder::SeekFrom w = whence == ZK_SEEK_START ? Decoder::SeekFrom::Start : whence == ZK_SEEK_END ? Decoder::SeekFrom::End : Decoder::SeekFrom::Current;
        uint64_t v = d->d.seek(w, n);
        if (out) *out = v;
    });
}

// ---------------------------------------------------------------- RawEncoder / Encoder
static EncodeOptions make_enc_opts(zk_engine *e, const zk_encode_opts *o)
{
    EncodeOptions opts;
    if (e) opts.engine(e);
    if (o) {
        if (o->frame_size) opts.frame_size_policy(o->policy == ZK_POLICY_COMPRESSED ? FrameSizePolicy::Compressed(o->frame_size)
                                                                                   : FrameSizePolicy::Uncompressed(o->frame_size));
        opts.checksum_flag(o->checksum != 0).compression_level(o->level);
        if (o->batch_frames) opts.batch_frames(o->batch_frames);
    }
    return opts;
}

int zk_raw_encoder_new(zk_engine *e, const zk_encode_opts *o, zk_raw_encoder **out)
{
    if (!out) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] { *out = new zk_raw_encoder(RawEncoder(make_enc_opts(e, o))); });
}
void zk_raw_encoder_free(zk_raw_encoder *r) { delete r; }
int zk_raw_encoder_compress(zk_raw_encoder *r, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, size_t *in_progress,
                            size_t *out_progress)
{
    return guard([&] { CompressionProgress p = r->r.compress(in, in_len, out, out_len); *in_progress = p.in_progress(); *out_progress = p.out_progress(); });
}
int zk_raw_encoder_compress_with_prefix(zk_raw_encoder *r, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                                        const uint8_t *prefix, size_t plen, size_t *in_progress, size_t *out_progress)
{
    return guard([&] { CompressionProgress p = r->r.compress_with_prefix(in, in_len, out, out_len, prefix, plen); *in_progress = p.in_progress(); *out_progress = p.out_progress(); });
}
int zk_raw_encoder_end_frame(zk_raw_encoder *r, uint8_t *out, size_t out_len, size_t *out_progress, size_t *data_left)
{
    return guard([&] { EpilogueProgress p = r->r.end_frame(out, out_len); *out_progress = p.out_progress(); *data_left = p.data_left(); });
}
zk_seek_table *zk_raw_encoder_seek_table(const zk_raw_encoder *r) { return new (std::nothrow) zk_seek_table{r->r.seek_table()}; }
void zk_raw_encoder_reset_frame(zk_raw_encoder *r) { r->r.reset_frame(); }
void zk_raw_encoder_reset_seek_table(zk_raw_encoder *r) { r->r.reset_seek_table(); }

int zk_encoder_new(zk_engine *e, const zk_encode_opts *o, zk_write_fn write, void *user, zk_encoder **out)
{
    if (!out || !write) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    return guard([&] { *out = new zk_encoder(std::make_shared<CallbackWriter>(write, user), make_enc_opts(e, o)); });
}
void zk_encoder_free(zk_encoder *e) { delete e; }
int64_t zk_encoder_compress(zk_encoder *e, const uint8_t *buf, size_t len)
{
    int64_t n = 0;
    int rc = guard([&] { n = (int64_t)e->e.compress(buf, len); });
    return rc ? rc : n;
}
int64_t zk_encoder_compress_with_prefix(zk_encoder *e, const uint8_t *buf, size_t len, const uint8_t *prefix, size_t plen)
{
    int64_t n = 0;
    int rc = guard([&] { n = (int64_t)e->e.compress_with_prefix(buf, len, prefix, plen); });
    return rc ? rc : n;
}
int64_t zk_encoder_end_frame(zk_encoder *e)
{
    int64_t n = 0;
    int rc = guard([&] { n = (int64_t)e->e.end_frame(); });
    return rc ? rc : n;
}
int zk_encoder_flush(zk_encoder *e) { return guard([&] { e->e.flush(); }); }
int zk_encoder_finish(zk_encoder *e, int format, uint64_t *total)
{
    return guard([&] { uint64_t t = e->e.finish_format(format == ZK_FORMAT_HEAD ? Format::Head : Format::Foot); if (total) *total = t; });
}
uint64_t zk_encoder_written_compressed(const zk_encoder *e) { return e->e.written_compressed(); }
zk_seek_table *zk_encoder_seek_table(const zk_encoder *e) { return new (std::nothrow) zk_seek_table{e->e.seek_table()}; }

}  // extern "C"
