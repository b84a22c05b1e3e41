// zk_device.h -- per-lane building blocks of the gfx950 frame decoder.
//
// Everything here is straight-line integer code that one lane executes on its own
// stream/block: bit readers, FSE / Huffman table construction, header parsing and
// the byte-source resolver of the sequence executor.  The kernels in zk_decode.hip
// own the orchestration (LDS placement, wave/workgroup cooperation, launches).
// The functions are ZK_HD (host+device) so that tests/sim/ can run the very same
// lane code on the CPU against the oracle before a GPU is involved.
//
// What this replaces in the reference: the arithmetic behind
// ZSTD_decompressStream at lib/src/decode.rs:242-256 (libzstd 1.5.7, not in the
// reference tree).  Format facts: RFC 8878 / SURVEY.md Appendix A.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD static inline
#endif
#if defined(__HIPCC__)
#define ZK_HDM __host__ __device__ __forceinline__      // member functions
#else
#define ZK_HDM inline
#endif

// Flags, mailboxes and rings that waves of one workgroup pass through LDS.  A `volatile` access through a generic pointer is a
// FLAT instruction with system scope on the device: it counts on vmcnt as well as lgkmcnt, and the wait for it also waits for
// every global store the wave has in flight (a store's acknowledgement is ~1 us away) -- the serial walkers issue such stores
// all the time.  Through an LDS-typed pointer the same access is a ds_read / ds_write that knows lgkmcnt only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_AS)       // ZK_NO_AS: experiment, the accesses as FLAT ones again
#define ZK_LDS_AS __attribute__((address_space(3)))
#else
#define ZK_LDS_AS
#endif
template <typename T> ZK_HD T zk_lds_ld(const volatile void *p) { return *(const volatile ZK_LDS_AS T *)(p); }
template <typename T> ZK_HD void zk_lds_st(volatile void *p, T v) { *(volatile ZK_LDS_AS T *)(p) = v; }
// the same by LDS byte address (a 32-bit integer: no generic pointer, no null check of a generic -> LDS cast in front of the access)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_AS)
ZK_HD uint32_t zk_lds_addr(const volatile void *p) { return (uint32_t)(uintptr_t)(const volatile ZK_LDS_AS uint8_t *)(p); }
template <typename T> ZK_HD T zk_lds_ld_at(uint32_t a) { return *(const volatile ZK_LDS_AS T *)(a); }
template <typename T> ZK_HD void zk_lds_st_at(uint32_t a, T v) { *(volatile ZK_LDS_AS T *)(a) = v; }
#else                                                           // (the host pass of hipcc only parses these)
ZK_HD uint32_t zk_lds_addr(const volatile void *p) { return (uint32_t)(uintptr_t)p; }
template <typename T> ZK_HD T zk_lds_ld_at(uint32_t a) { return *(const volatile T *)(uintptr_t)(a); }
template <typename T> ZK_HD void zk_lds_st_at(uint32_t a, T v) { *(volatile T *)(uintptr_t)(a) = v; }
#endif
// a store through a pointer that was itself read from memory (a struct in LDS): the compiler cannot tell where it points and would
// issue a FLAT store; this says "global"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_AS)
#define ZK_GLB_AS __attribute__((address_space(1)))
#else
#define ZK_GLB_AS
#endif
template <typename T> ZK_HD void zk_glb_st(void *p, T v) { *(ZK_GLB_AS T *)(p) = v; }

// ZSTD_ErrorCode values used on this path (per-frame status words; 0 = ok)
enum : uint32_t {
    ZK_OK = 0,
    ZK_E_GENERIC = 1,
    ZK_E_PREFIX_UNKNOWN = 10,
    ZK_E_FRAMEPARAM_UNSUPPORTED = 14,
    ZK_E_WINDOW_TOO_LARGE = 16,
    ZK_E_CORRUPTION = 20,
    ZK_E_CHECKSUM_WRONG = 22,
    ZK_E_DICT_WRONG = 32,
    ZK_E_DST_TOO_SMALL = 70,
    ZK_E_SRC_SIZE_WRONG = 72,
};

constexpr uint32_t ZK_BLOCK_MAX = 131072;          // Block_Maximum_Size upper bound
constexpr uint32_t ZK_MAX_FRAME = 0x40000000u;     // SEEKABLE_MAX_FRAME_SIZE (reference lib.rs:58)
constexpr uint64_t ZK_SIZE_UNKNOWN = ~0ull;        // zk_walk_frame: the frame's decompressed size is what is being asked for

// ---------------------------------------------------------------- data records in HBM
struct ZkFrameInfo {            // written by the frame walker, one per frame
    uint32_t n_blocks;
    uint32_t n_seq;             // record slots of the frame: every block's sequences rounded up to 8 (zk_walk_frame)
    uint32_t lit_bytes;         // total Huffman-coded literal bytes (need literal scratch)
    uint32_t status;            // ZSTD_ErrorCode, 0 ok
    uint32_t checksum_flag;
    uint32_t checksum;          // stored Content_Checksum (valid if flag)
    uint32_t window;            // clamped to 2^31
    uint32_t n_own_tables;      // blocks that DEFINE a sequence table (FSE_Compressed / RLE mode): how much zk_k_fse_quad may have to do
                                // (blocks that only repeat or use predefined tables share them with their neighbours: zk_k_fse_predef)
    uint64_t fcs;               // Frame_Content_Size where the header carries one, else ZK_SIZE_UNKNOWN (zk_k_frame_sizes holds the blocks' sum against it)
};

struct ZkFrameBase {            // exclusive prefix sums over frames
    uint64_t block_base;
    uint64_t seq_base;
    uint64_t lit_base;
};

struct ZkBlock {                // one per block, contiguous per frame
    uint64_t src;               // absolute offset (in the compressed buffer) of the block content
    uint64_t lit_base;          // literal scratch offset (Huffman literals)
    uint64_t seq_base;          // sequence scratch index
    uint32_t bsize;             // Block_Size
    uint32_t frame;             // frame index inside the batch
    uint8_t type;               // 0 raw, 1 rle, 2 compressed
    uint8_t lit_type;           // 0 raw, 1 rle, 2 huffman, 3 treeless
    uint8_t lit_streams;        // 1 or 4
    uint8_t seq_modes;          // Symbol_Compression_Modes byte
    uint32_t lit_regen;         // Regenerated_Size of the literals
    uint32_t lit_off;           // offset in the block content of the literal payload
    uint32_t lit_comp;          // payload bytes (huffman: tree description + streams)
    uint32_t seq_off;           // offset in the block content of the modes byte (nseq > 0)
    uint32_t nseq;
    uint32_t huf_def;           // global block index whose literal section carries the tree in force
    uint32_t tab_def[3];        // global block index whose sequence header defines LL / OF / ML table in force
    uint32_t out_size;          // regenerated size of the block
    uint32_t rep_out[3];        // repeat-offset history after the block (symbolic, see zk_rep_*)
    uint32_t status;
    uint32_t pad;
};
static_assert(sizeof(ZkBlock) == 96, "ZkBlock layout");

struct ZkSeq {                  // one per sequence, 16 B: the form the executor stages in LDS
    uint32_t out_end;           // block-relative output position after this sequence's match
    uint32_t ml;                // match length
    uint32_t off;               // offset, possibly symbolic (zk_rep_*)
    uint32_t lit_end;           // block-relative literal count after this sequence's literals
};
// The record as it travels through HBM between the sequence decoders and the executor: 8 bytes.
//   lit_end[16:0] | (out_end - 1)[33:17] | off[63:34]
// Both running sums stay absolute (block-relative), so the executor needs no prefix sum: a sequence's match length is
// out_end - previous out_end - (lit_end - previous lit_end), one neighbour away.  out_end is in [3, 2^17] and lit_end
// in [0, 2^17 - 3] for every valid block.  The 30-bit offset field holds concrete offsets up to ZK_OFF_MAX and, above
// them, the symbolic ones (slot 0..2, delta < 2^16: a block has at most 43690 sequences).
typedef uint64_t ZkSeqP;
constexpr uint32_t ZK_OFF_SYM = 0x3FFC0000u;          // offset fields >= this are symbolic: ZK_OFF_SYM | slot << 16 | delta
constexpr uint32_t ZK_OFF_MAX = ZK_OFF_SYM - 1;       // largest concrete offset (1 GiB - 256 KiB - 1)
ZK_HD ZkSeqP zk_seq_pack(uint32_t out_end, uint32_t lit_end, uint32_t off)
{
    const uint32_t f = (off & 0x80000000u) ? (ZK_OFF_SYM | (((off >> 28) & 3u) << 16) | (off & 0xFFFFu)) : off;
    return (uint64_t)(lit_end & 0x1FFFFu) | ((uint64_t)((out_end - 1u) & 0x1FFFFu) << 17) | ((uint64_t)f << 34);
}
ZK_HD uint32_t zk_seqp_lit_end(ZkSeqP p) { return (uint32_t)p & 0x1FFFFu; }
ZK_HD uint32_t zk_seqp_out_end(ZkSeqP p) { return ((uint32_t)(p >> 17) & 0x1FFFFu) + 1u; }
ZK_HD uint32_t zk_seqp_off(ZkSeqP p)             // back to the walk's form: concrete, or 0x80000000 | slot << 28 | delta
{
    const uint32_t f = (uint32_t)(p >> 34);
    return f >= ZK_OFF_SYM ? (0x80000000u | (((f >> 16) & 3u) << 28) | (f & 0xFFFFu)) : f;
}
// records [idx - 1, idx] -> the staged form (prev = 0 for the block's first sequence)
ZK_HD ZkSeq zk_seq_unpack(ZkSeqP prev, ZkSeqP cur, bool first)
{
    ZkSeq s;
    const uint32_t po = first ? 0u : zk_seqp_out_end(prev), pl = first ? 0u : zk_seqp_lit_end(prev);
    s.out_end = zk_seqp_out_end(cur); s.lit_end = zk_seqp_lit_end(cur); s.off = zk_seqp_off(cur);
    s.ml = s.out_end - po - (s.lit_end - pl);
    return s;
}

// ---------------------------------------------------------------- small helpers
ZK_HD uint32_t zk_highbit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
ZK_HD uint32_t zk_rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
ZK_HD uint32_t zk_rd24(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
ZK_HD uint32_t zk_rd32(const uint8_t *p) { return zk_rd24(p) | ((uint32_t)p[3] << 24); }

// 64-bit little-endian word at stream offset `off`; bytes outside [0,len) read as zero.
ZK_HD uint64_t zk_ldword(const uint8_t *base, int32_t off, uint32_t len)
{
    if (off >= 0 && (uint32_t)off + 8u <= len) {
        uint64_t v;
        memcpy(&v, base + off, 8);
        return v;
    }
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) {
        int32_t o = off + i;
        if (o >= 0 && (uint32_t)o < len) v |= (uint64_t)base[o] << (8 * i);
    }
    return v;
}

// ---------------------------------------------------------------- backward bit reader (A.7)
// cur: next unread bit is the MSB; cnt valid bits.  res: reserve word, left aligned.
// nxt: word prefetched one step ahead so the HBM/L2 latency overlaps the decode chain.
struct ZkBwd {
    const uint8_t *base;
    uint64_t cur, res, nxt;
    uint32_t len;
    int32_t cnt, rescnt, next_off;
    int32_t bits_left;          // real (non zero-fill) bits still unread; < 0 == over-read
};

ZK_HD bool zk_bwd_init(ZkBwd &b, const uint8_t *base, uint32_t len)
{
    if (len == 0) return false;
    uint32_t last = base[len - 1];
    if (last == 0) return false;
    uint32_t hb = zk_highbit(last);
    b.base = base; b.len = len;
    b.bits_left = (int32_t)((len - 1) * 8 + hb);
    uint64_t w = zk_ldword(base, (int32_t)len - 8, len);
    b.cur = w << (8 - hb);
    b.cnt = 56 + (int32_t)hb;
    b.res = zk_ldword(base, (int32_t)len - 16, len);
    b.rescnt = 64;
    b.nxt = zk_ldword(base, (int32_t)len - 24, len);
    b.next_off = (int32_t)len - 32;
    return true;
}

// top up cur to 64 bits (zero-fill once the stream is exhausted)
ZK_HD void zk_bwd_refill(ZkBwd &b)
{
    if (b.cnt >= 64) return;
    int32_t room = 64 - b.cnt;                       // 1..64
    b.cur |= b.cnt ? (b.res >> b.cnt) : b.res;
    if (b.rescnt > room) {                           // reserve still has bits left
        b.res <<= room; b.rescnt -= room; b.cnt = 64;
        return;
    }
    b.cnt += b.rescnt;                               // reserve fully consumed
    b.res = b.nxt; b.rescnt = 64;
    b.nxt = zk_ldword(b.base, b.next_off, b.len);
    b.next_off -= 8;
    room = 64 - b.cnt;
    if (room > 0) {                                  // cnt >= 1 here, so room <= 63
        b.cur |= b.res >> b.cnt;
        b.res <<= room; b.rescnt -= room; b.cnt = 64;
    }
}

// n <= 32 and n <= cnt (caller refills)
ZK_HD uint32_t zk_bwd_read(ZkBwd &b, uint32_t n)
{
    uint32_t v = n ? (uint32_t)(b.cur >> (64 - n)) : 0u;
    b.cur = n >= 64 ? 0 : b.cur << n;
    b.cnt -= (int32_t)n; b.bits_left -= (int32_t)n;
    return v;
}
ZK_HD uint32_t zk_bwd_peek(const ZkBwd &b, uint32_t n) { return (uint32_t)(b.cur >> (64 - n)); }   // 1 <= n <= 32
ZK_HD void zk_bwd_skip(ZkBwd &b, uint32_t n) { b.cur <<= n; b.cnt -= (int32_t)n; b.bits_left -= (int32_t)n; }

// ---------------------------------------------------------------- FSE
// packed decode cell: sym[7:0] | nb[11:8] | xbits[16:12] | base[31:17]
ZK_HD uint32_t zk_cell(uint32_t sym, uint32_t nb, uint32_t xbits, uint32_t base) { return sym | (nb << 8) | (xbits << 12) | (base << 17); }
ZK_HD uint32_t zk_cell_sym(uint32_t c) { return c & 0xff; }
ZK_HD uint32_t zk_cell_nb(uint32_t c) { return (c >> 8) & 0xf; }
ZK_HD uint32_t zk_cell_xbits(uint32_t c) { return (c >> 12) & 0x1f; }
ZK_HD uint32_t zk_cell_base(uint32_t c) { return c >> 17; }

// Two cell formats behind one interface (what the table build writes, what the walk reads):
//   ZkCells32  the packed 32-bit cell above (shared predefined tables, Huffman weight tables);
//   ZkCells16  16 bits, for tables that live per block in LDS (zk_k_fse), where the table footprint sets how many
//              blocks a CU decodes at once:  sym[15:10] | e[9:0].  e folds (nb, base): base is a multiple of 2^nb and
//              base + 2^nb <= 512, so for nb >= 1 the number n = base + 2^(nb-1) < 512 has its lowest set bit at nb-1
//              (nb = ctz(n) + 1, base = n with that bit cleared); nb == 0 cells take e = 512 + base.  The number of
//              extra value bits is not stored: it comes from the symbol's entry in the value table (LL / ML) or is
//              the symbol itself (OF).
struct ZkCells32 {
    typedef uint32_t cell_t;
    static constexpr bool kFat = false;          // (zk_seq_walk_quad: the cell alone tells bit counts and baseline)
    static ZK_HDM uint32_t touch(cell_t c) { return c; }
    static ZK_HDM cell_t make(uint32_t sym, uint32_t nb, uint32_t xbits, uint32_t base, uint32_t = 0) { return zk_cell(sym, nb, xbits, base); }
    static ZK_HDM uint32_t sym(cell_t c) { return zk_cell_sym(c); }
    static ZK_HDM uint32_t nb(cell_t c, uint32_t = 0) { return zk_cell_nb(c); }
    static ZK_HDM uint32_t base(cell_t c, uint32_t = 0, uint32_t = 0) { return zk_cell_base(c); }
    static ZK_HDM uint32_t xbits(cell_t c, const uint32_t *) { return zk_cell_xbits(c); }
    static ZK_HDM uint32_t baseline(cell_t c, const uint32_t *value_table) { return value_table[zk_cell_sym(c)] & 0xFFFFFFu; }
    static ZK_HDM cell_t tmp_sym(uint32_t s) { return s; }
    static ZK_HDM uint32_t tmp_get(cell_t c) { return c; }
    static ZK_HDM cell_t with_value(cell_t c, const uint32_t *, uint32_t) { return c; }
};
struct ZkCell64 { uint32_t c, v; };              // the 32-bit cell + the symbol's value baseline
struct ZkCells64 {                               // for tables shared by a workgroup (zk_k_fse_predef*): LDS is no constraint
    typedef ZkCell64 cell_t;                     // there, and the baseline in the cell saves two dependent LDS reads per sequence
    static constexpr bool kFat = true;
    static ZK_HDM uint32_t touch(cell_t c) { return c.c; }
    static ZK_HDM cell_t make(uint32_t sym, uint32_t nb, uint32_t xbits, uint32_t base, uint32_t = 0) { cell_t r; r.c = zk_cell(sym, nb, xbits, base); r.v = 0; return r; }
    static ZK_HDM uint32_t sym(cell_t c) { return zk_cell_sym(c.c); }
    static ZK_HDM uint32_t nb(cell_t c, uint32_t = 0) { return zk_cell_nb(c.c); }
    static ZK_HDM uint32_t base(cell_t c, uint32_t = 0, uint32_t = 0) { return zk_cell_base(c.c); }
    static ZK_HDM uint32_t xbits(cell_t c, const uint32_t *) { return zk_cell_xbits(c.c); }
    static ZK_HDM uint32_t baseline(cell_t c, const uint32_t *) { return c.v; }
    static ZK_HDM cell_t tmp_sym(uint32_t s) { cell_t r; r.c = s; r.v = 0; return r; }
    static ZK_HDM uint32_t tmp_get(cell_t c) { return c.c; }
    // (no value table = the offset table: the baseline of code s is 1 << s)
    static ZK_HDM cell_t with_value(cell_t c, const uint32_t *value_table, uint32_t s) { c.v = value_table ? value_table[s] & 0xFFFFFFu : 1u << (s & 31u); return c; }
};
struct ZkCells16 {
    typedef uint16_t cell_t;
    static constexpr bool kFat = false;
    static ZK_HDM uint32_t touch(cell_t c) { return c; }
    static ZK_HDM cell_t make(uint32_t sym, uint32_t nb, uint32_t, uint32_t base, uint32_t = 0)
    { return (cell_t)((sym << 10) | (nb ? base + (1u << (nb - 1)) : 512u + base)); }
    static ZK_HDM uint32_t sym(cell_t c) { return (uint32_t)c >> 10; }
    static ZK_HDM uint32_t nb(cell_t c, uint32_t = 0) { const uint32_t e = c & 1023u; return e >= 512u ? 0u : (uint32_t)__builtin_ctz(e | 512u) + 1u; }
    static ZK_HDM uint32_t base(cell_t c, uint32_t = 0, uint32_t = 0) { const uint32_t e = c & 1023u; return e >= 512u ? e - 512u : e & (e - 1u); }
    static ZK_HDM uint32_t xbits(cell_t c, const uint32_t *value_table) { const uint32_t s = (uint32_t)c >> 10; return value_table ? value_table[s] >> 24 : s; }
    static ZK_HDM uint32_t baseline(cell_t c, const uint32_t *value_table) { return value_table[(uint32_t)c >> 10] & 0xFFFFFFu; }
    static ZK_HDM cell_t tmp_sym(uint32_t s) { return (cell_t)s; }
    static ZK_HDM uint32_t tmp_get(cell_t c) { return c; }
    static ZK_HDM cell_t with_value(cell_t c, const uint32_t *, uint32_t) { return c; }
};

// ZkCellsX16: 16 bits again, for the chain walkers (zk_seq_walk_quad), where what counts is the number of DEPENDENT instructions
// between one cell read and the next: sym[15:10] | x[9:0] with x = the cell's position in its symbol's run, count <= x < 2 count
// (what the table build calls next[s]++).  With the table's accuracy log at hand -- a lane constant of the walker --
// nb = al - highbit(x) and base = (x << nb) - 2^al: a count-leading-zeros, a shift and two subtractions, where ZkCells16's folded
// field takes eleven instructions to take apart.
struct ZkCellsX16 {
    typedef uint16_t cell_t;
    static constexpr bool kFat = false;
    static ZK_HDM uint32_t touch(cell_t c) { return c; }
    static ZK_HDM cell_t make(uint32_t sym, uint32_t, uint32_t, uint32_t, uint32_t x) { return (cell_t)((sym << 10) | x); }
    static ZK_HDM uint32_t sym(cell_t c) { return (uint32_t)c >> 10; }
    static ZK_HDM uint32_t nb(cell_t c, uint32_t al) { return al + (uint32_t)__builtin_clz((uint32_t)c & 1023u) - 31u; }
    static ZK_HDM uint32_t base(cell_t c, uint32_t al, uint32_t nb) { return (((uint32_t)c & 1023u) << nb) - (1u << al); }
    static ZK_HDM uint32_t xbits(cell_t c, const uint32_t *value_table) { const uint32_t s = (uint32_t)c >> 10; return value_table ? value_table[s] >> 24 : s; }
    static ZK_HDM uint32_t baseline(cell_t c, const uint32_t *value_table) { return value_table[(uint32_t)c >> 10] & 0xFFFFFFu; }
    static ZK_HDM cell_t tmp_sym(uint32_t s) { return (cell_t)s; }
    static ZK_HDM uint32_t tmp_get(cell_t c) { return c; }
    static ZK_HDM cell_t with_value(cell_t c, const uint32_t *, uint32_t) { return c; }
};

// value tables: base | bits << 24
#define ZK_LLV(b, n) ((uint32_t)(b) | ((uint32_t)(n) << 24))
#define ZK_LL_TABLE { \
    ZK_LLV(0,0),ZK_LLV(1,0),ZK_LLV(2,0),ZK_LLV(3,0),ZK_LLV(4,0),ZK_LLV(5,0),ZK_LLV(6,0),ZK_LLV(7,0), \
    ZK_LLV(8,0),ZK_LLV(9,0),ZK_LLV(10,0),ZK_LLV(11,0),ZK_LLV(12,0),ZK_LLV(13,0),ZK_LLV(14,0),ZK_LLV(15,0), \
    ZK_LLV(16,1),ZK_LLV(18,1),ZK_LLV(20,1),ZK_LLV(22,1),ZK_LLV(24,2),ZK_LLV(28,2),ZK_LLV(32,3),ZK_LLV(40,3), \
    ZK_LLV(48,4),ZK_LLV(64,6),ZK_LLV(128,7),ZK_LLV(256,8),ZK_LLV(512,9),ZK_LLV(1024,10),ZK_LLV(2048,11),ZK_LLV(4096,12), \
    ZK_LLV(8192,13),ZK_LLV(16384,14),ZK_LLV(32768,15),ZK_LLV(65536,16) }
#define ZK_ML_TABLE { \
    ZK_LLV(3,0),ZK_LLV(4,0),ZK_LLV(5,0),ZK_LLV(6,0),ZK_LLV(7,0),ZK_LLV(8,0),ZK_LLV(9,0),ZK_LLV(10,0), \
    ZK_LLV(11,0),ZK_LLV(12,0),ZK_LLV(13,0),ZK_LLV(14,0),ZK_LLV(15,0),ZK_LLV(16,0),ZK_LLV(17,0),ZK_LLV(18,0), \
    ZK_LLV(19,0),ZK_LLV(20,0),ZK_LLV(21,0),ZK_LLV(22,0),ZK_LLV(23,0),ZK_LLV(24,0),ZK_LLV(25,0),ZK_LLV(26,0), \
    ZK_LLV(27,0),ZK_LLV(28,0),ZK_LLV(29,0),ZK_LLV(30,0),ZK_LLV(31,0),ZK_LLV(32,0),ZK_LLV(33,0),ZK_LLV(34,0), \
    ZK_LLV(35,1),ZK_LLV(37,1),ZK_LLV(39,1),ZK_LLV(41,1),ZK_LLV(43,2),ZK_LLV(47,2),ZK_LLV(51,3),ZK_LLV(59,3), \
    ZK_LLV(67,4),ZK_LLV(83,4),ZK_LLV(99,5),ZK_LLV(131,7),ZK_LLV(259,8),ZK_LLV(515,9),ZK_LLV(1027,10),ZK_LLV(2051,11), \
    ZK_LLV(4099,12),ZK_LLV(8195,13),ZK_LLV(16387,14),ZK_LLV(32771,15),ZK_LLV(65539,16) }

// predefined normalised distributions (A.6)
#define ZK_LL_DEFNORM {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1}
#define ZK_OF_DEFNORM {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1}
#define ZK_ML_DEFNORM {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1}

enum { ZK_TAB_LL = 0, ZK_TAB_OF = 1, ZK_TAB_ML = 2 };
ZK_HD uint32_t zk_tab_maxsym(int t) { return t == ZK_TAB_LL ? 35u : t == ZK_TAB_OF ? 31u : 52u; }
ZK_HD uint32_t zk_tab_maxal(int t) { return t == ZK_TAB_OF ? 8u : 9u; }

// A.5 normalised-count header.  Forward LSB-first bits.  norm[] gets max_sym+1 entries at most.
// Returns bytes consumed (>0) or 0 on corruption.
ZK_HD uint32_t zk_fse_read_ncount(const uint8_t *src, uint32_t len, uint32_t max_sym, uint32_t max_al,
                                  int16_t *norm, uint32_t *nsym_out, uint32_t *al_out)
{
    if (len == 0) return 0;
    uint32_t bitpos = 0;
    uint64_t w = zk_ldword(src, 0, len);
    uint32_t wbase = 0;                                   // byte offset of w
#define ZK_FPEEK(n) ((uint32_t)(w >> (bitpos - wbase * 8)) & ((1u << (n)) - 1u))
#define ZK_FADV(n) do { bitpos += (n); if (bitpos - wbase * 8 > 32) { wbase = bitpos >> 3; w = zk_ldword(src, (int32_t)wbase, len); } } while (0)
    uint32_t al = ZK_FPEEK(4) + 5; ZK_FADV(4);
    if (al > max_al) return 0;
    int32_t remaining = (1 << al) + 1, threshold = 1 << al;
    uint32_t nb = al + 1, sym = 0;
    while (remaining > 1 && sym <= max_sym) {
        int32_t max = 2 * threshold - 1 - remaining;
        uint32_t v = ZK_FPEEK(nb);
        int32_t cnt;
        if ((int32_t)(v & (uint32_t)(threshold - 1)) < max) { cnt = (int32_t)(v & (uint32_t)(threshold - 1)); ZK_FADV(nb - 1); }
        else { cnt = (int32_t)(v & (uint32_t)(2 * threshold - 1)); if (cnt >= threshold) cnt -= max; ZK_FADV(nb); }
        cnt -= 1;
        remaining -= cnt < 0 ? -cnt : cnt;
        norm[sym++] = (int16_t)cnt;
        if (cnt == 0) {
            for (;;) {
                uint32_t rep = ZK_FPEEK(2); ZK_FADV(2);
                for (uint32_t i = 0; i < rep; i++) { if (sym > max_sym) return 0; norm[sym++] = 0; }
                if (rep != 3) break;
            }
        }
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (bitpos > len * 8 + 16) return 0;
    }
#undef ZK_FPEEK
#undef ZK_FADV
    if (remaining != 1 || sym > max_sym + 1) return 0;
    uint32_t used = (bitpos + 7) >> 3;
    if (used > len) return 0;
    *nsym_out = sym; *al_out = al;
    return used;
}

// A.6 decode-table build.  cells[1<<al]; next[] scratch of nsym u16.  kind selects the xbits column.
template <typename CP = ZkCells32>
ZK_HD bool zk_fse_build(typename CP::cell_t *cells, const int16_t *norm, uint32_t nsym, uint32_t al, uint16_t *next,
                        const uint32_t *value_table /* LL/ML value table or nullptr (OF / weights) */)
{
    uint32_t size = 1u << al, mask = size - 1;
    int32_t high = (int32_t)size - 1;
    for (uint32_t s = 0; s < nsym; s++) {
        if (norm[s] == -1) { if (high < 0) return false; cells[high--] = CP::tmp_sym(s); next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    uint32_t step = (size >> 1) + (size >> 3) + 3, pos = 0;
    for (uint32_t s = 0; s < nsym; s++) {
        for (int32_t i = 0; i < norm[s]; i++) {
            cells[pos] = CP::tmp_sym(s);
            do { pos = (pos + step) & mask; } while ((int32_t)pos > high);
        }
    }
    if (pos != 0) return false;
    for (uint32_t i = 0; i < size; i++) {
        uint32_t s = CP::tmp_get(cells[i]);
        uint32_t x = next[s]++;
        uint32_t nb = al - zk_highbit(x);
        uint32_t xb = value_table ? (value_table[s] >> 24) : s;
        cells[i] = CP::with_value(CP::make(s, nb, xb & 31u, (x << nb) - size, x), value_table, s);
    }
    return true;
}

// ---------------------------------------------------------------- Huffman (A.4)
// table cell: sym | nbits << 8 (u16).  A block's table is built in two steps so that the kernel can size the
// LDS it needs from the tree's depth: zk_huf_read_weights (tree description -> weights + table layout), then
// zk_huf_fill_table.
struct ZkHufHdr {                    // survives until the table is filled
    uint8_t weights[256];
    uint16_t rank[16];               // first table index per weight
};
struct ZkHufTmp {                    // only while FSE-compressed weights are being read
    uint32_t fse[64];
    int16_t norm[16];
    uint16_t next[16];
};

// Parses the tree description at src (len = bytes available).  Returns the description size in bytes (0 on
// corruption); *n_out = number of symbols with a weight entry (implied last one included), *maxbits_out = table log.
ZK_HD uint32_t zk_huf_read_weights(const uint8_t *src, uint32_t len, ZkHufHdr *hd, ZkHufTmp *tmp, uint32_t *n_out,
                                   uint32_t *maxbits_out)
{
    if (len < 1) return 0;
    uint32_t h = src[0], n = 0, used;
    uint8_t *w = hd->weights;
    if (h >= 128) {
        n = h - 127;
        used = 1 + (n + 1) / 2;
        if (used > len) return 0;
        for (uint32_t i = 0; i < n; i++) { uint32_t b = src[1 + i / 2]; w[i] = (uint8_t)((i & 1) ? (b & 15) : (b >> 4)); }
    } else {
        used = 1 + h;
        if (used > len || h < 2) return 0;
        uint32_t nsym, al;
        uint32_t r = zk_fse_read_ncount(src + 1, h, 11, 6, tmp->norm, &nsym, &al);
        if (r == 0 || r >= h) return 0;
        if (!zk_fse_build(tmp->fse, tmp->norm, nsym, al, tmp->next, nullptr)) return 0;
        ZkBwd b;
        if (!zk_bwd_init(b, src + 1 + r, h - r)) return 0;
        uint32_t s1 = zk_bwd_read(b, al), s2 = zk_bwd_read(b, al);
        for (;;) {                                        // two interleaved states, stop on over-read
            if (n >= 254) return 0;
            zk_bwd_refill(b);
            uint32_t c1 = tmp->fse[s1];
            w[n++] = (uint8_t)zk_cell_sym(c1);
            if (b.bits_left < (int32_t)zk_cell_nb(c1)) { w[n++] = (uint8_t)zk_cell_sym(tmp->fse[s2]); break; }
            s1 = zk_cell_base(c1) + zk_bwd_read(b, zk_cell_nb(c1));
            if (n >= 254) return 0;
            uint32_t c2 = tmp->fse[s2];
            w[n++] = (uint8_t)zk_cell_sym(c2);
            if (b.bits_left < (int32_t)zk_cell_nb(c2)) { w[n++] = (uint8_t)zk_cell_sym(tmp->fse[s1]); break; }
            s2 = zk_cell_base(c2) + zk_bwd_read(b, zk_cell_nb(c2));
        }
    }
    uint32_t sum = 0;
    for (uint32_t i = 0; i < 13; i++) hd->rank[i] = 0;
    for (uint32_t i = 0; i < n; i++) { uint32_t x = w[i]; if (x > 11) return 0; if (x) sum += 1u << (x - 1); hd->rank[x]++; }
    if (sum == 0) return 0;
    uint32_t maxbits = zk_highbit(sum) + 1;
    if (maxbits > 11) return 0;
    uint32_t rest = (1u << maxbits) - sum;
    if (rest & (rest - 1)) return 0;
    uint32_t lastw = zk_highbit(rest) + 1;
    w[n++] = (uint8_t)lastw; hd->rank[lastw]++;
    // rank[wt] -> first table index of weight wt (weight 1 first)
    uint32_t pos = 0;
    for (uint32_t wt = 1; wt <= maxbits; wt++) { uint32_t c = hd->rank[wt]; hd->rank[wt] = (uint16_t)pos; pos += c << (wt - 1); }
    if (pos != (1u << maxbits)) return 0;
    *n_out = n; *maxbits_out = maxbits;
    return used;
}

// Fills table[1 << maxbits] from the weights (consumes hd->rank).
ZK_HD void zk_huf_fill_table(uint16_t *table, ZkHufHdr *hd, uint32_t n, uint32_t maxbits)
{
    for (uint32_t s = 0; s < n; s++) {
        uint32_t wt = hd->weights[s];
        if (!wt) continue;
        uint32_t cnt = 1u << (wt - 1), at = hd->rank[wt];
        uint16_t cell = (uint16_t)(s | ((maxbits + 1 - wt) << 8));
        for (uint32_t k = 0; k < cnt; k++) table[at + k] = cell;
        hd->rank[wt] = (uint16_t)(at + cnt);
    }
}

// Huffman stream reader.  The backward bitstream is consumed through 8-byte ALIGNED words of the buffer it lives in
// (an unaligned 8-byte access costs ~3.4 L1 tag lookups on gfx950 and the L1 lookup rate is what bounds this
// kernel): W(j) = the aligned word at AE - 8 (j + 1), AE = stream end rounded up to 8.  The (AE - end) bytes above
// the stream end are skipped like the final byte's padding.  Three words sit in registers (A current, B, C) and one
// load is in flight (P); a word is loaded exactly once, when the window advances.  Words below the stream start
// are clamped to the stream's lowest aligned word: they only matter for corrupt streams, which the final position
// check rejects.
struct ZkHufRd {
    const uint8_t *ptr;              // address of the next aligned word to load
    const uint8_t *lo;               // lowest aligned word touching the stream
    uint64_t A, B, C, P;
    uint32_t c, words;               // bits of A consumed; words fully consumed
};
ZK_HD uint64_t zk_hufrd_load(ZkHufRd &r)
{
    const uint8_t *p = r.ptr < r.lo ? r.lo : r.ptr;
    r.ptr -= 8;
    return *(const ZK_GLB_AS uint64_t *)(p);                        // (the compressed buffer: a global load, not a FLAT one)
}
ZK_HD uint64_t zk_hufrd_refill(ZkHufRd &r)
{
    if (r.c >= 64) {
        r.A = r.B; r.B = r.C; r.C = r.P;
        r.c -= 64; r.words++;
        r.P = zk_hufrd_load(r);
    }
    return (r.A << r.c) | ((r.B >> 1) >> (63 - r.c));
}

// Hand-over between a decoding lane and its companion lane (LDS, one per workgroup; nullptr on the host).  The
// decoder never issues a global store itself: it drops every 8-symbol pack into a 2-deep ring and publishes
// (packs written, stream position); the companion wave stores the packs to HBM and touches the stream's cache
// lines ahead.  gfx9 counts loads and stores in ONE in-order counter (vmcnt), so a store in the decode loop makes
// every wait for a stream word also wait for the store's write acknowledgement.
constexpr uint32_t ZK_HUF_RING = 8;              // packs per lane in the hand-over ring
constexpr uint32_t ZK_HUF_BURST = 4;             // the companion stores 4 packs = 32 contiguous bytes at a time: single 8-byte
                                                 // pieces of a literal line were mostly evicted from L2 half-filled (WRITE_SIZE 5x the
                                                 // payload).  Measured on 4 GiB (ring, burst): (2,1) 3.5-4.2 ms, (4,2) 3.27, (8,2) 3.57,
                                                 // (8,4) 3.10, (8,6) 3.3, (16,8) 3.3, (32,16) 3.8 -- deeper rings cost LDS, i.e. decoders per CU
struct ZkHufMail {
    uint64_t pack[ZK_HUF_RING][64];
    uint32_t state[64];              // packs written [13:0] | (next stream offset + 64) << 14
    uint32_t consumed[64];           // packs stored by the companion
};
ZK_HD uint32_t zk_huf_mail_state(uint32_t written, int32_t next_off)
{
    const int32_t o = next_off + 64;
    return (written & 0x3fffu) | ((uint32_t)(o < 0 ? 0 : o) << 14);
}

ZK_HD bool zk_huf_decode_stream(const uint16_t *table, uint32_t maxbits, const uint8_t *src, uint32_t len,
                                uint8_t *dst, uint32_t n, bool store = true, volatile ZkHufMail *mail = nullptr, uint32_t mlane = 0)
{
    if (len == 0) return false;
    const uint32_t last = src[len - 1];
    if (last == 0) return false;
    ZkHufRd r;
    const uintptr_t end = (uintptr_t)src + len, aend = (end + 7) & ~(uintptr_t)7;
    r.lo = reinterpret_cast<const uint8_t *>((uintptr_t)src & ~(uintptr_t)7);
    r.ptr = reinterpret_cast<const uint8_t *>(aend) - 8;
    r.A = zk_hufrd_load(r); r.B = zk_hufrd_load(r); r.C = zk_hufrd_load(r); r.P = zk_hufrd_load(r);
    const uint32_t skip = (uint32_t)(aend - end) * 8;    // bytes between the stream end and the aligned end
    r.c = skip + 8 - zk_highbit(last);                   // + zero padding + the sentinel bit
    r.words = 0;
    const uint32_t sh = 64 - maxbits;
    uint32_t i = 0;
    // head: single symbols until the output position is 8-byte aligned (aligned 8-byte stores: one L1 lookup each)
    {
        uint32_t head = (uint32_t)((0 - (uintptr_t)dst) & 7);
        if (head > n) head = n;
        while (i < head) {
            uint64_t cur = zk_hufrd_refill(r);
            const uint32_t lim = head - i < 4 ? head - i : 4;
            uint32_t used = 0;
            for (uint32_t k = 0; k < lim; k++) {
                const uint32_t c = table[cur >> sh];
                if (store) dst[i + k] = (uint8_t)c;
                cur <<= c >> 8; used += c >> 8;
            }
            r.c += used;
            i += lim;
        }
    }
    const uint32_t i0 = i;                               // the mailbox counts packs from here
    // (Measured and dropped, round 6: the four symbols of a window as two PAIRS out of its upper 32 bits -- the second symbol of a pair is
    //  (w << n) >> (32 - maxbits), one 64-bit shift per window instead of eight: bit-exact, zk_k_huf 2.80 -> 3.09 ms.  The 64-bit shifts are
    //  not what the chain shift -> cell -> shift waits for; the pair form has more instructions on it.  tools/gpu_calls/r6bc.sh)
    while (i + 8 <= n) {
        uint64_t pack = 0;
        uint64_t cur = zk_hufrd_refill(r);
        uint32_t used = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t c = table[cur >> sh];
            pack |= (uint64_t)(c & 0xff) << (8 * k);
            cur <<= c >> 8; used += c >> 8;
        }
        r.c += used;
        cur = zk_hufrd_refill(r);
        used = 0;
        for (uint32_t k = 4; k < 8; k++) {
            const uint32_t c = table[cur >> sh];
            pack |= (uint64_t)(c & 0xff) << (8 * k);
            cur <<= c >> 8; used += c >> 8;
        }
        r.c += used;
        if (mail) {
            if (store) {
                const uint32_t it = (i - i0) >> 3;
                // (LDS-typed accesses: through the generic pointer they were FLAT instructions, each followed by a wait for vmcnt)
                while (((it - zk_lds_ld<uint32_t>(&mail->consumed[mlane])) & 0x3fffu) >= ZK_HUF_RING) {}     // ring full: the companion is behind
                zk_lds_st<uint64_t>(&mail->pack[it % ZK_HUF_RING][mlane], pack);
                zk_lds_st<uint32_t>(&mail->state[mlane], zk_huf_mail_state(it + 1, (int32_t)(r.ptr - src)));
            }
        } else if (store) *reinterpret_cast<uint64_t *>(dst + i) = pack;
        i += 8;
    }
    while (i < n) {
        uint64_t cur = zk_hufrd_refill(r);
        const uint32_t lim = n - i < 4 ? n - i : 4;
        uint32_t used = 0;
        for (uint32_t k = 0; k < lim; k++) {
            const uint32_t c = table[cur >> sh];
            if (store) dst[i + k] = (uint8_t)c;
            cur <<= c >> 8; used += c >> 8;
        }
        r.c += used;
        i += lim;
    }
    return (uint64_t)r.words * 64 + r.c == (uint64_t)len * 8 + skip;
}

// ---------------------------------------------------------------- header parsing
struct ZkLitHdr { uint32_t type, streams, regen, comp, hdr; };

// literals section header (A.3); avail = bytes of block content
ZK_HD bool zk_parse_lit_hdr(const uint8_t *p, uint32_t avail, ZkLitHdr &h)
{
    if (avail < 1) return false;
    uint32_t b0 = p[0];
    h.type = b0 & 3;
    uint32_t sf = (b0 >> 2) & 3;
    h.streams = 4;
    if (h.type < 2) {
        if (sf == 0 || sf == 2) { h.hdr = 1; h.regen = b0 >> 3; }
        else if (sf == 1) { h.hdr = 2; if (avail < 2) return false; h.regen = (b0 >> 4) + ((uint32_t)p[1] << 4); }
        else { h.hdr = 3; if (avail < 3) return false; h.regen = (b0 >> 4) + ((uint32_t)p[1] << 4) + ((uint32_t)p[2] << 12); }
        h.comp = h.type == 0 ? h.regen : 1;
    } else {
        if (sf < 2) { h.hdr = 3; if (avail < 3) return false; uint32_t v = zk_rd24(p); h.regen = (v >> 4) & 0x3ff; h.comp = (v >> 14) & 0x3ff; if (sf == 0) h.streams = 1; }
        else if (sf == 2) { h.hdr = 4; if (avail < 4) return false; uint32_t v = zk_rd32(p); h.regen = (v >> 4) & 0x3fff; h.comp = v >> 18; }
        else { h.hdr = 5; if (avail < 5) return false; uint64_t v = (uint64_t)zk_rd32(p) | ((uint64_t)p[4] << 32); h.regen = (uint32_t)(v >> 4) & 0x3ffff; h.comp = (uint32_t)(v >> 22) & 0x3ffff; }
    }
    if (h.regen > ZK_BLOCK_MAX) return false;
    if ((uint64_t)h.hdr + h.comp > avail) return false;
    return true;
}

// Number_of_Sequences (A.8); returns header length (1..3) or 0
ZK_HD uint32_t zk_parse_nseq(const uint8_t *p, uint32_t avail, uint32_t &nseq)
{
    if (avail < 1) return 0;
    uint32_t b0 = p[0];
    if (b0 < 128) { nseq = b0; return 1; }
    if (b0 < 255) { if (avail < 2) return 0; nseq = ((b0 - 128) << 8) + p[1]; return 2; }
    if (avail < 3) return 0;
    nseq = (uint32_t)p[1] + ((uint32_t)p[2] << 8) + 0x7F00;
    return 3;
}

// ---------------------------------------------------------------- frame walker
// One lane walks one frame.  blocks == nullptr: count only.
// comp/c_begin/c_end: the frame occupies comp[c_begin, c_end).  d_size: expected decompressed size.
ZK_HD void zk_walk_frame(const uint8_t *comp, uint64_t c_begin, uint64_t c_end, uint64_t d_size,
                         uint32_t frame_idx, const ZkFrameBase *base, ZkBlock *blocks, ZkFrameInfo &fi)
{
    fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; fi.status = ZK_OK;
    fi.checksum_flag = 0; fi.checksum = 0; fi.window = 0; fi.n_own_tables = 0; fi.fcs = ZK_SIZE_UNKNOWN;
    uint64_t csz = c_end - c_begin;
    const uint8_t *f = comp + c_begin;
    if (csz < 6) { fi.status = ZK_E_SRC_SIZE_WRONG; return; }
    if (zk_rd32(f) != 0xFD2FB528u) { fi.status = ZK_E_PREFIX_UNKNOWN; return; }
    uint32_t fhd = f[4];
    uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, cks = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 0x08) { fi.status = ZK_E_FRAMEPARAM_UNSUPPORTED; return; }
    uint64_t p = 5, window = 0;
    if (!single) {
        uint32_t wd = f[p++];
        uint32_t e = wd >> 3, m = wd & 7;
        if (10 + e > 31) { fi.status = ZK_E_WINDOW_TOO_LARGE; return; }
        window = 1ull << (10 + e); window += (window >> 3) * m;
    }
    uint32_t dl = did == 3 ? 4 : did;
    uint32_t fl = fcs_flag == 0 ? single : (1u << fcs_flag);
    if (p + dl + fl > csz) { fi.status = ZK_E_SRC_SIZE_WRONG; return; }
    uint32_t dict = 0;
    for (uint32_t i = 0; i < dl; i++) dict |= (uint32_t)f[p + i] << (8 * i);
    if (dict) { fi.status = ZK_E_DICT_WRONG; return; }
    p += dl;
    uint64_t fcs = 0;
    for (uint32_t i = 0; i < fl; i++) fcs |= (uint64_t)f[p + i] << (8 * i);
    if (fl == 2) fcs += 256;
    p += fl;
    if (fl) fi.fcs = fcs;
    if (single) window = fcs;
    // d_size == ZK_SIZE_UNKNOWN: a frame nobody holds a seek entry for (zk_frame_content_sizes: the walk + the sequence walks then tell
    // its size); the bounds below are taken against the largest frame the format of the reference allows
    const bool unknown = d_size == ZK_SIZE_UNKNOWN;
    if (unknown) d_size = fl ? fcs : (uint64_t)ZK_MAX_FRAME;
    if (fl && fcs != d_size) { fi.status = ZK_E_CORRUPTION; return; }
    if (unknown && d_size > ZK_MAX_FRAME) { fi.status = ZK_E_FRAMEPARAM_UNSUPPORTED; return; }
    fi.checksum_flag = cks;
    fi.window = window > 0x80000000ull ? 0x80000000u : (uint32_t)window;
    uint32_t block_max = window < ZK_BLOCK_MAX ? (uint32_t)window : ZK_BLOCK_MAX;

    uint64_t blk = base ? base->block_base : 0, seqb = base ? base->seq_base : 0, litb = base ? base->lit_base : 0;
    uint32_t huf_def = 0xFFFFFFFFu, tab_def[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint64_t out_known = 0;                                 // regenerated bytes of raw/rle blocks (sanity bound)
    for (;;) {
        if (p + 3 > csz) { fi.status = ZK_E_SRC_SIZE_WRONG; return; }
        uint32_t bh = zk_rd24(f + p); p += 3;
        uint32_t last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
        if (type == 3 || bsize > block_max) { fi.status = ZK_E_CORRUPTION; return; }
        uint32_t content = type == 1 ? 1u : bsize;
        if (p + content > csz) { fi.status = ZK_E_SRC_SIZE_WRONG; return; }
        ZkBlock b;
        b.src = c_begin + p; b.lit_base = litb; b.seq_base = seqb; b.bsize = bsize; b.frame = frame_idx;
        b.type = (uint8_t)type; b.lit_type = 0; b.lit_streams = 0; b.seq_modes = 0;
        b.lit_regen = 0; b.lit_off = 0; b.lit_comp = 0; b.seq_off = 0; b.nseq = 0;
        b.out_size = type == 2 ? 0 : bsize;
        b.status = ZK_OK; b.pad = 0;
        if (type == 2) {
            const uint8_t *c = f + p;
            ZkLitHdr lh;
            if (bsize < 2 || !zk_parse_lit_hdr(c, bsize, lh)) { fi.status = ZK_E_CORRUPTION; return; }
            b.lit_type = (uint8_t)lh.type; b.lit_streams = (uint8_t)lh.streams;
            b.lit_regen = lh.regen; b.lit_off = lh.hdr; b.lit_comp = lh.comp;
            if (lh.type == 2) huf_def = (uint32_t)blk;
            if (lh.type >= 2) {
                if (huf_def == 0xFFFFFFFFu) { fi.status = ZK_E_CORRUPTION; return; }
                litb += lh.regen; fi.lit_bytes += lh.regen;
            }
            uint32_t so = lh.hdr + lh.comp, nseq = 0;
            uint32_t nh = zk_parse_nseq(c + so, bsize - so, nseq);
            if (nh == 0) { fi.status = ZK_E_CORRUPTION; return; }
            so += nh;
            b.nseq = nseq;
            if (nseq == 0) {
                if (so != bsize) { fi.status = ZK_E_CORRUPTION; return; }
                b.out_size = lh.regen;
            } else {
                if (so + 1 > bsize) { fi.status = ZK_E_CORRUPTION; return; }
                uint32_t modes = c[so];
                if (modes & 3) { fi.status = ZK_E_CORRUPTION; return; }
                b.seq_modes = (uint8_t)modes; b.seq_off = so;
                if ((modes ^ (modes >> 1)) & 0x54) fi.n_own_tables++;     // some table in mode 1 (RLE) or 2 (FSE_Compressed): the two bits of its field differ
                for (int t = 0; t < 3; t++) {
                    uint32_t m = (modes >> (6 - 2 * t)) & 3;
                    if (m != 3) tab_def[t] = (uint32_t)blk;
                    else if (tab_def[t] == 0xFFFFFFFFu) { fi.status = ZK_E_CORRUPTION; return; }
                }
                // a block's records start on a 64-byte line of the record scratch (8 records): the sequence kernels' 32- and 64-byte bursts
                // then never straddle a line (round 3: 4.26 GiB written for 2.57 GB of records)
                seqb += (nseq + 7u) & ~7u; fi.n_seq += (nseq + 7u) & ~7u;
            }
        } else {
            out_known += bsize;
            if (out_known > d_size) { fi.status = ZK_E_CORRUPTION; return; }
        }
        b.huf_def = huf_def;
        b.tab_def[0] = tab_def[0]; b.tab_def[1] = tab_def[1]; b.tab_def[2] = tab_def[2];
        b.rep_out[0] = 0x80000000u; b.rep_out[1] = 0x90000000u; b.rep_out[2] = 0xA0000000u;   // identity (symbolic)
        if (blocks) blocks[blk] = b;
        blk++; fi.n_blocks++;
        p += content;
        if (last) break;
    }
    if (cks) {
        if (p + 4 > csz) { fi.status = ZK_E_SRC_SIZE_WRONG; return; }
        fi.checksum = zk_rd32(f + p); p += 4;
    }
    if (p != csz) { fi.status = ZK_E_SRC_SIZE_WRONG; return; }   // seek-table c_size must match the frame exactly
    if (fi.lit_bytes > d_size) { fi.status = ZK_E_CORRUPTION; return; }
}

// ---------------------------------------------------------------- repeat offsets, symbolic across blocks
// A block's sequences are entropy-decoded before the previous block's final repeat
// history is known.  An offset value v is concrete if v < 2^31, else it names
// "history slot s at block entry, minus d": s = (v >> 28) & 3, d = v & 0x0FFFFFFF.
ZK_HD uint32_t zk_rep_sym(uint32_t slot) { return 0x80000000u | (slot << 28); }
ZK_HD bool zk_rep_is_sym(uint32_t v) { return (v & 0x80000000u) != 0; }
ZK_HD uint32_t zk_rep_resolve(uint32_t v, const uint32_t init[3])
{
    if (!(v & 0x80000000u)) return v;
    uint32_t s = (v >> 28) & 3, d = v & 0x0FFFFFFFu;
    uint32_t x = init[s];
    return d >= x ? 0u : x - d;             // 0 == invalid, caught by the caller
}

// ---------------------------------------------------------------- sequence section decode (one lane per block)
template <typename CP>
struct ZkSeqTablesT {                // LDS-resident, per lane
    typename CP::cell_t ll[512];
    typename CP::cell_t ml[512];
    typename CP::cell_t of[256];
    union {                          // table-build scratch, later the output ring (16 packed records)
        struct { int16_t norm[64]; uint16_t next[64]; };
        ZkSeqP ring[16];
    };
};
typedef ZkSeqTablesT<ZkCells32> ZkSeqTables;       // 5.25 KiB
typedef ZkSeqTablesT<ZkCells16> ZkSeqTables16;     // 2.75 KiB
typedef ZkSeqTablesT<ZkCellsX16> ZkSeqTablesX16;   // the same size

// Locate + build table t of block `def` (the block whose header defines the table in force).
// comp: compressed buffer.  Returns bytes the description occupies in `def` (for own-block parsing), or -1.
template <typename CP = ZkCells32>
ZK_HD int32_t zk_seq_table_setup(const uint8_t *comp, const ZkBlock &def, int t, ZkSeqTablesT<CP> *T, uint32_t *al_out,
                                 const uint32_t *ll_values, const uint32_t *ml_values)
{
    const int16_t ll_def[36] = ZK_LL_DEFNORM;
    const int16_t of_def[29] = ZK_OF_DEFNORM;
    const int16_t ml_def[53] = ZK_ML_DEFNORM;
    const uint8_t *c = comp + def.src;
    uint32_t modes = def.seq_modes;
    uint32_t p = def.seq_off + 1;
    uint32_t nsym, al;
    // skip the descriptions that precede table t
    for (int u = 0; u < t; u++) {
        uint32_t m = (modes >> (6 - 2 * u)) & 3;
        if (m == 1) p += 1;
        else if (m == 2) {
            if (p >= def.bsize) return -1;
            uint32_t r = zk_fse_read_ncount(c + p, def.bsize - p, zk_tab_maxsym(u), zk_tab_maxal(u), T->norm, &nsym, &al);
            if (!r) return -1;
            p += r;
        }
    }
    uint32_t m = (modes >> (6 - 2 * t)) & 3;
    typename CP::cell_t *cells = t == ZK_TAB_LL ? T->ll : t == ZK_TAB_OF ? T->of : T->ml;
    const uint32_t *vt = t == ZK_TAB_LL ? ll_values : t == ZK_TAB_ML ? ml_values : nullptr;
    if (m == 0) {
        const int16_t *d = t == ZK_TAB_LL ? ll_def : t == ZK_TAB_OF ? of_def : ml_def;
        nsym = t == ZK_TAB_LL ? 36 : t == ZK_TAB_OF ? 29 : 53;
        al = t == ZK_TAB_OF ? 5 : 6;
        for (uint32_t i = 0; i < nsym; i++) T->norm[i] = d[i];
        if (!zk_fse_build<CP>(cells, T->norm, nsym, al, T->next, vt)) return -1;
        *al_out = al;
        return 0;
    }
    if (m == 1) {
        if (p >= def.bsize) return -1;
        uint32_t s = c[p];
        if (s > zk_tab_maxsym(t)) return -1;
        cells[0] = CP::with_value(CP::make(s, 0, (vt ? (vt[s] >> 24) : s) & 31u, 0, 1), vt, s);       // accuracy log 0: x = 1
        *al_out = 0;
        return 1;
    }
    if (m == 2) {
        if (p >= def.bsize) return -1;
        uint32_t r = zk_fse_read_ncount(c + p, def.bsize - p, zk_tab_maxsym(t), zk_tab_maxal(t), T->norm, &nsym, &al);
        if (!r) return -1;
        if (!zk_fse_build<CP>(cells, T->norm, nsym, al, T->next, vt)) return -1;
        *al_out = al;
        return (int32_t)r;
    }
    return -1;   // a defining block never has Repeat_Mode for this table
}

// Reverse bit readers for the sequence bitstream: both present the next unread bits left-aligned in a 64-bit
// window and guarantee a minimum number of them after every step.
//   ZkRevU  one unaligned 8-byte load per sequence (window = bytes starting at max(0, (pos - 57) >> 3): >= 57
//           bits or everything that is left).  Fewest instructions per sequence: used where a lane runs alone on
//           its SIMD and every instruction is latency (zk_k_fse, per-block tables).
//   ZkRevA  8-byte ALIGNED words of the compressed buffer, each loaded exactly once: A (current, c bits
//           consumed), B, C in registers, one load in flight (P); 64 bits after every step.  An unaligned 8-byte
//           access costs ~3.4 L1 tag lookups on gfx950 and every lookup that misses L1 is an L2 request: used
//           where many waves share a CU and the memory pipeline is the limit (zk_k_fse_predef).
// Both may touch a few bytes outside the bitstream (ZkRevU up to 7 after a stream shorter than 8 bytes, ZkRevA up
// to 7 before and after): compressed buffers carry ZK_DEV_COMP_PADDING readable bytes at the end.
constexpr uint32_t ZK_DEV_COMP_PADDING = 8;     // == ZK_COMP_PADDING of include/zeekstd_amd.h
ZK_HD uint64_t zk_ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct ZkRevU {
    const uint8_t *base; uint64_t W; int32_t pos, wpos;      // W holds stream bits [wpos, wpos + 64); pos = unread bits
    static ZK_HDM int32_t byte_of(int32_t pos) { int32_t b = (pos - 57) >> 3; return b < 0 ? 0 : b; }
#ifdef ZK_KO_LOAD                                               // experiment (tools/variants): the walk's timing without its one load per step (wrong bytes)
    ZK_HDM void reload() { const int32_t bo = byte_of(pos); W = 0x0123456789ABCDEFull; wpos = bo * 8; }
#else
    ZK_HDM void reload() { const int32_t bo = byte_of(pos); W = zk_ld64(base + bo); wpos = bo * 8; }
#endif
    ZK_HDM bool init(const uint8_t *b, uint32_t len)
    {
        const uint32_t last = b[len - 1];
        if (last == 0) return false;
        base = b; pos = (int32_t)((len - 1) * 8 + zk_highbit(last));
        reload();
        return true;
    }
    ZK_HDM uint32_t avail() const { return (uint32_t)(pos - wpos); }
    ZK_HDM uint64_t window() const { return W << ((64 - (pos - wpos)) & 63); }
    ZK_HDM void consume(uint32_t n) { pos -= (int32_t)n; reload(); }          // n <= avail()
    ZK_HDM uint32_t read(uint32_t n)                                           // any n <= 31, reloads the window first; over-reads return 0
    {
        reload();
        pos -= (int32_t)n;
        if (pos < 0) return 0;
        return (uint32_t)(W >> ((pos - wpos) & 63)) & ((1u << n) - 1u);
    }
    ZK_HDM int32_t remaining() const { return pos; }
    ZK_HDM void clamp() { if (pos < 0) pos = 0; reload(); }                    // after a run of read()s
    ZK_HDM void attach(void *, uint32_t) {}
};

struct ZkRevA {
    const uint8_t *ptr, *lo;         // next aligned word to load; lowest aligned word touching the stream
    uint64_t A, B, C, P;
    uint32_t c;                      // bits of A consumed (< 64 between steps)
    int32_t rem;                     // unread stream bits; < 0 == over-read
    ZK_HDM uint64_t load()
    {
        const uint8_t *p = ptr < lo ? lo : ptr;     // below the stream: only a corrupt stream gets here, rem goes negative
        ptr -= 8;
        return *reinterpret_cast<const uint64_t *>(p);
    }
    ZK_HDM void advance() { if (c >= 64) { A = B; B = C; C = P; c -= 64; P = load(); } }
    ZK_HDM bool init(const uint8_t *b, uint32_t len)
    {
        const uint32_t last = b[len - 1];
        if (last == 0) return false;
        const uint32_t hb = zk_highbit(last);
        const uintptr_t end = (uintptr_t)b + len, aend = (end + 7) & ~(uintptr_t)7;
        lo = reinterpret_cast<const uint8_t *>((uintptr_t)b & ~(uintptr_t)7);
        ptr = reinterpret_cast<const uint8_t *>(aend) - 8;
        A = load(); B = load(); C = load(); P = load();
        c = (uint32_t)(aend - end) * 8 + 8 - hb;             // bytes above the stream end + zero padding + sentinel
        rem = (int32_t)((len - 1) * 8 + hb);
        advance();                                           // c == 64: the top word holds no stream bit
        return true;
    }
    ZK_HDM uint32_t avail() const { return 64; }
    ZK_HDM uint64_t window() const { return (A << c) | ((B >> 1) >> (63 - c)); }
    ZK_HDM void consume(uint32_t n) { c += n; rem -= (int32_t)n; advance(); }  // n <= 64
    ZK_HDM uint32_t read(uint32_t n)                                            // n <= 32
    {
        const uint64_t X = window();
        const uint32_t v = n ? (uint32_t)(X >> (64 - n)) : 0u;
        consume(n);
        return v;
    }
    ZK_HDM int32_t remaining() const { return rem; }
    ZK_HDM void clamp() {}
    ZK_HDM void attach(void *, uint32_t) {}
};

// ZkRevL: like ZkRevA, but the aligned stream words come out of an LDS ring that a companion ("feeder") lane of
// another wave fills (zk_k_fse_predef).  The walking wave then issues no global load at all.  Why: loads return in
// order per wave (vmcnt), a wave of 64 streams has some lane fetching its next word at almost every step, and so every
// step of the whole wave used to wait for one L2 round trip (~1 us under load) -- the floor of the sequence walk.
// An LDS read is ~100 cycles.  Hand-over per lane: feeder writes word f to ring[f % R][lane] and then publishes
// filled = f + 1; the walker spins until filled > k before it reads word k and publishes taken = k + 1 afterwards;
// the feeder only writes while filled - taken < R.  Words past the stream's first byte are not waited for (a corrupt
// stream reads zeros there and fails on `rem`).
constexpr uint32_t ZK_REVL_RING = 8;             // words per lane
struct ZkRevLShared {                            // LDS, one per walking wave
    uint64_t ring[ZK_REVL_RING][64];
    uint32_t filled[64], taken[64];
};
struct ZkRevL {
    ZkRevLShared *sh;
    uint64_t A, B;
    uint32_t lane, c, k, nwords;                 // k: next word index to take
    int32_t rem;
    ZK_HDM void attach(ZkRevLShared *s, uint32_t l) { sh = s; lane = l; }
    ZK_HDM uint64_t take()
    {
        if (k >= nwords) { k++; return 0; }
        while (zk_lds_ld<uint32_t>(&sh->filled[lane]) <= k) {}
        const uint64_t w = zk_lds_ld<uint64_t>(&sh->ring[k % ZK_REVL_RING][lane]);
        k++;
        zk_lds_st<uint32_t>(&sh->taken[lane], k);
        return w;
    }
    ZK_HDM void advance() { if (c >= 64) { A = B; B = take(); c -= 64; } }
    // words of the stream [b, b + len): W(j) = aligned 8 bytes at aend - 8 (j + 1), j < nwords
    static ZK_HDM uint32_t word_count(const uint8_t *b, uint32_t len)
    {
        const uintptr_t end = (uintptr_t)b + len, aend = (end + 7) & ~(uintptr_t)7, lo = (uintptr_t)b & ~(uintptr_t)7;
        return (uint32_t)((aend - lo) >> 3);
    }
    ZK_HDM bool init(const uint8_t *b, uint32_t len)
    {
        const uint32_t last = b[len - 1];
        if (last == 0) return false;
        const uint32_t hb = zk_highbit(last);
        const uintptr_t end = (uintptr_t)b + len, aend = (end + 7) & ~(uintptr_t)7;
        nwords = word_count(b, len);
        k = 0;
        A = take(); B = take();
        c = (uint32_t)(aend - end) * 8 + 8 - hb;
        rem = (int32_t)((len - 1) * 8 + hb);
        advance();
        return true;
    }
    ZK_HDM uint32_t avail() const { return 64; }
    ZK_HDM uint64_t window() const { return (A << c) | ((B >> 1) >> (63 - c)); }
    ZK_HDM void consume(uint32_t n) { c += n; rem -= (int32_t)n; advance(); }
    ZK_HDM uint32_t read(uint32_t n)
    {
        const uint64_t X = window();
        const uint32_t v = n ? (uint32_t)(X >> (64 - n)) : 0u;
        consume(n);
        return v;
    }
    ZK_HDM int32_t remaining() const { return rem; }
    ZK_HDM void clamp() {}
};

// The n (<= 31) most significant bits of a 32-bit word, 0 for n == 0: one bit-field extract on the device (a width
// of 0 yields 0), where the portable form needs a shift, a compare and a select.
ZK_HD uint32_t zk_top_bits(uint32_t hi, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(hi, 32u - n, n);
#else
    return n ? hi >> (32 - n) : 0u;
#endif
}

ZK_HD void *zk_rd_shared_type(const ZkRevU &) { return nullptr; }
ZK_HD void *zk_rd_shared_type(const ZkRevA &) { return nullptr; }
ZK_HD ZkRevLShared *zk_rd_shared_type(const ZkRevL &) { return nullptr; }

// Wave-cooperative flush of the 4-record rings of a 64-lane wave (device only, zk_k_fse_predef): a lane's four
// 16-B records are one 64-B line of its block's record array, so instead of every lane storing its own ring
// (4 instructions x 64 separate L2 write requests) four neighbouring lanes store one ring per instruction
// (4 x 16 requests of 64 B).  All 64 lanes must run the walk in lock step (nloop = the wave's longest block).
// WHERE record i of a lane's ring lies: every lane of the wave parks record i in the same step, the rings are RING x 8 bytes apart (64 bytes:
// sixteen lanes -- one pass of the LDS -- on two pairs of banks, the write ran eight times: 21 % of zk_k_fse_predef_fed's clocks were LDS bank
// conflicts, profiles/r06_pmc_entropy.txt), so lane l keeps record i in slot i ^ ((l >> 1) & 7): sixteen lanes, sixteen pairs of banks.
template <int RING> ZK_HD uint32_t zk_coop_ring_swz(uint32_t lane) { return RING == 8 ? (lane >> 1) & 7u : 0u; }
struct ZkCoopFlush {
    ZkSeqP *ring;                    // [64][RING]
    ZkSeqP *seqs;                    // the record array of the whole batch
    uint64_t base[64];               // record index of each lane's block
    uint32_t nseq[64];               // 0: lane stores nothing (shadow / inactive)
    uint32_t nloop;
};

// The 3-state walk over the sequence bitstream of block b (tables already built: LL / OF / ML cells, accuracy
// logs al[3], bitstream at offset bs_off of the block content).  ring: RING-record LDS staging of this lane.
// Fills seqs[], b.out_size / b.rep_out / b.status.
// store == false: a shadow lane (see zk_k_fse) -- it walks the same block as a real lane but never writes to HBM
// active == false (cooperative mode only): the lane has no block but takes part in the flushes.
template <int RING, typename RD, typename CP = ZkCells32>
ZK_HD void zk_seq_walk(const uint8_t *comp, ZkBlock &b, uint32_t bs_off, const typename CP::cell_t *LL, const typename CP::cell_t *OF,
                       const typename CP::cell_t *ML,
                       const uint32_t *al, ZkSeqP *ring, ZkSeqP *seqs, const uint32_t *ll_values, const uint32_t *ml_values,
                       bool store = true, ZkCoopFlush *coop = nullptr, bool active = true, uint32_t lane = 0, void *rd_shared = nullptr)
{
    RD r;
    r.attach(static_cast<decltype(zk_rd_shared_type(r))>(rd_shared), lane);
    uint32_t bad = 0;
    if (active && bs_off >= b.bsize) { bad = 1; active = false; }
    if (active && !r.init(comp + b.src + bs_off, b.bsize - bs_off)) { bad = 1; active = false; }
    const uint32_t nseq = active ? b.nseq : 0;
    uint32_t sl = 0, so = 0, sm = 0;
    if (active) { sl = r.read(al[0]); so = r.read(al[1]); sm = r.read(al[2]); bad |= r.remaining() < 0; r.clamp(); }
    uint32_t rep0 = zk_rep_sym(0), rep1 = zk_rep_sym(1), rep2 = zk_rep_sym(2);
    uint32_t out = 0, lit = 0;
    // Sequences are decoded in groups of RING (the LDS record ring).  The step is one basic block:
    // errors only accumulate into `bad` (a corrupt stream keeps walking harmlessly: states stay inside
    // their tables, window addresses are clamped) and the rare sequence that needs more bits than the
    // window guarantees leaves the block for a field-by-field slow step.
    typename CP::cell_t cl = LL[sl], co = OF[so], cm = ML[sm];
    const uint32_t nloop = coop ? coop->nloop : nseq;
    const uint32_t swz = coop ? zk_coop_ring_swz<RING>(lane) : 0u;
    for (uint32_t g0 = 0; g0 < nloop; g0 += RING) {
        for (uint32_t i = g0; i < g0 + RING; i++) {
            if (i >= nseq) break;
            const uint32_t nOf = CP::sym(co), nMl = CP::xbits(cm, ml_values), nLl = CP::xbits(cl, ll_values);
            const bool more = i + 1 < nseq;
            const uint32_t nbl = more ? CP::nb(cl) : 0, nbm = more ? CP::nb(cm) : 0, nbo = more ? CP::nb(co) : 0;
            const uint32_t nval = nOf + nMl + nLl;
            const uint32_t total = nval + nbl + nbm + nbo;
            uint32_t ofx, mlx, llx;
            const typename CP::cell_t csl = cl, csm = cm;                           // symbols of THIS sequence
            bad |= nOf > 30;
            if (total <= r.avail()) {                                               // every field lies inside the window
                const uint64_t X = r.window();
                r.consume(total);                                                   // the next window's load only needs the bit counts
                // state bits first: they gate the next cell reads (the dependent chain of the walk)
                uint64_t S = X << (nval & 63);
                const uint32_t h0 = (uint32_t)(S >> 32); S <<= nbl;
                const uint32_t h1 = (uint32_t)(S >> 32); S <<= nbm;
                const uint32_t h2 = (uint32_t)(S >> 32);
                sl = CP::base(cl) + zk_top_bits(h0, nbl);
                sm = CP::base(cm) + zk_top_bits(h1, nbm);
                so = CP::base(co) + zk_top_bits(h2, nbo);
                cl = LL[sl]; co = OF[so]; cm = ML[sm];                     // issued early; used next iteration
                // value bits (off the chain)
                uint64_t V = X;
                ofx = zk_top_bits((uint32_t)(V >> 32), nOf & 31); V <<= (nOf & 31);
                mlx = zk_top_bits((uint32_t)(V >> 32), nMl); V <<= nMl;
                llx = zk_top_bits((uint32_t)(V >> 32), nLl);
            } else {                                                                // more bits than the window guarantees (or over-read)
                ofx = r.read(nOf & 31); mlx = r.read(nMl); llx = r.read(nLl);
                sl = CP::base(cl) + r.read(nbl);
                sm = CP::base(cm) + r.read(nbm);
                so = CP::base(co) + r.read(nbo);
                bad |= r.remaining() < 0;
                r.clamp();
                cl = LL[sl]; co = OF[so]; cm = ML[sm];
            }
            const uint32_t ofv = (1u << (nOf & 31)) + ofx;
            const uint32_t ml = CP::baseline(csm, ml_values) + mlx;
            const uint32_t ll = CP::baseline(csl, ll_values) + llx;
            // offset + repeat history, select form (A.8)
            const bool is_rep = ofv <= 3;
            const uint32_t idx = ofv - 1 + (ll == 0);                               // 0..3 when is_rep
            const uint32_t r0m1 = zk_rep_is_sym(rep0) ? rep0 + 1 : rep0 - 1;        // "rep0 - 1" (symbolic: one more subtracted)
            const uint32_t cand = idx == 0 ? rep0 : idx == 1 ? rep1 : idx == 2 ? rep2 : r0m1;
            const uint32_t off = is_rep ? cand : ofv - 3;
            bad |= off == 0;                                                        // concrete rep0 - 1 == 0
            const bool sh1 = !is_rep || idx >= 1, sh2 = !is_rep || idx >= 2;
            rep2 = sh2 ? rep1 : rep2;
            rep1 = sh1 ? rep0 : rep1;
            rep0 = off;
            lit += ll; out += ll + ml;
            bad |= (lit > b.lit_regen) | (out > ZK_BLOCK_MAX) | (!zk_rep_is_sym(off) & (off > ZK_OFF_MAX));
            ring[(i & (RING - 1)) ^ swz] = zk_seq_pack(out, lit, off);
        }
        // records are parked in LDS and written out a group at a time (few, wide store bursts)
#if defined(__HIP_DEVICE_COMPILE__)
        if (coop) {
            // RING neighbouring lanes store one lane's ring (RING x 8 contiguous bytes), 64 / RING rings per instruction
            static_assert(RING == 4 || RING == 8 || RING == 16, "ring");
            for (uint32_t j = 0; j < (uint32_t)RING; j++) {
                const uint32_t m = (64 / RING) * j + lane / RING, piece = lane % RING, k = g0 + piece;
                if (k < coop->nseq[m]) zk_glb_st<ZkSeqP>(&coop->seqs[coop->base[m] + k], zk_lds_ld<ZkSeqP>(&coop->ring[m * RING + (piece ^ zk_coop_ring_swz<RING>(m))]));
            }
            continue;
        }
#endif
        const uint32_t gend = g0 + RING < nseq ? g0 + RING : nseq;
        if (store) for (uint32_t k = g0; k < gend; k++) seqs[k] = ring[k & (RING - 1)];
    }
    if (!active) { if (bad) b.status = ZK_E_CORRUPTION; return; }
    bad |= r.remaining() != 0;
    if (bad) { b.status = ZK_E_CORRUPTION; return; }
    out += b.lit_regen - lit;
    if (out > ZK_BLOCK_MAX) { b.status = ZK_E_CORRUPTION; return; }
    b.out_size = out;
    b.rep_out[0] = rep0; b.rep_out[1] = rep1; b.rep_out[2] = rep2;
}


// Decode all sequences of block b into seqs[]: builds the block's LL / OF / ML tables in T (per-lane LDS), then walks.
template <typename RD = ZkRevU, typename CP = ZkCells32>
ZK_HD void zk_decode_sequences(const uint8_t *comp, const ZkBlock *blocks, ZkBlock &b, ZkSeqTablesT<CP> *T, ZkSeqP *seqs,
                               const uint32_t *ll_values, const uint32_t *ml_values, bool store = true)
{
    uint32_t al[3];
    uint32_t own = 0;                                   // bytes of table descriptions in this block
    for (int t = 0; t < 3; t++) {
        uint32_t m = (b.seq_modes >> (6 - 2 * t)) & 3;
        const ZkBlock &def = m == 3 ? blocks[b.tab_def[t]] : b;
        int32_t r = zk_seq_table_setup<CP>(comp, def, t, T, &al[t], ll_values, ml_values);
        if (r < 0) { b.status = ZK_E_CORRUPTION; return; }
        if (m != 3) own += (uint32_t)r;
    }
    zk_seq_walk<16, RD, CP>(comp, b, b.seq_off + 1 + own, T->ll, T->of, T->ml, al, T->ring, seqs, ll_values, ml_values, store);
}


// ---------------------------------------------------------------- sequence section decode (a quad of lanes per block)
// A block's sequences are ONE serial chain -- state -> cell -> bit counts -> state bits -> next state -- and a lone wave issues
// in order: every instruction of the step is on the chain's clock whether it depends on it or not.  The work is therefore cut
// in two (round 4; one function did both before, ~110 instructions per step):
//   zk_seq_walk_quad    the chain.  One lane per FSE state machine of a block (XCH::bcast = a DPP quad_perm move on the device,
//                       lock-stepped fibers in tests/sim); lane t leaves the VALUE of its field of every sequence -- literal
//                       length, offset value, match length -- with OUT::put and nothing else: no repeat-offset history, no
//                       running sums, no record.
//   zk_seq_finish_quad  everything else, by another wave, four sequences of a block at a time (lane j of a quad = sequence
//                       i0 + j): repeat offsets, the running sums, the 8-byte records.  Off the chain and amortised.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_WAVE_BARRIER() __builtin_amdgcn_wave_barrier()
#else
#define ZK_WAVE_BARRIER() ((void)0)
#endif
constexpr uint32_t ZK_QUAD_ROUND = 4;            // sequences of a block that zk_seq_finish_quad takes at a time (= lanes of a quad)
// t = the lane's table (ZK_TAB_LL / ZK_TAB_OF / ZK_TAB_ML = its position in the quad); cells: that table's cells; vt: that table's
// BASELINES, nothing else in the words (literal lengths 0..65536, match lengths 3..65539, offsets 1 << code); al[3]: accuracy logs.
// out.gate(i0, stream position) is called by all three lanes in front of every round of ZK_QUAD_ROUND sequences (flow control of
// the hand-over, the toucher's cue), out.put(i0, k, v) once per sequence i0 + k and lane.  Returns 0, or 1 for a damaged
// stream (over-read, bits left over) -- all three lanes return the same.
// bits: where the lane reads the bitstream from -- nullptr = in place (comp + b.src + bs_off), or a staged copy of its
// b.bsize - bs_off bytes (LDS, readable 8 bytes past the end).
template <typename RD, typename CP, typename XCH, typename OUT>
ZK_HD uint32_t zk_seq_walk_quad(const uint8_t *comp, const ZkBlock &b, uint32_t bs_off, uint32_t t,
                                const typename CP::cell_t *cells, const uint32_t *vt, const uint32_t *al, OUT &out, const uint8_t *bits = nullptr)
{
    RD r;
    uint32_t bad = 0;
    const bool ok = bs_off < b.bsize && r.init(bits ? bits : comp + b.src + bs_off, b.bsize - bs_off);
    if (!ok) return 1;
    const uint32_t nseq = b.nseq;
    uint32_t state;
    {
        const uint32_t s0 = r.read(al[0]), s1 = r.read(al[1]), s2 = r.read(al[2]);
        bad |= r.remaining() < 0; r.clamp();
        state = t == ZK_TAB_LL ? s0 : t == ZK_TAB_OF ? s1 : s2;
    }
    // where the lane's fields start: value bits come OF, ML, LL; state bits LL, ML, OF
    const uint32_t kOfV = t == ZK_TAB_OF ? 0u : 0xFFu;              // OF value bits precede ML's and LL's
    const uint32_t kMlV = t == ZK_TAB_LL ? 0xFFu : 0u;              // ML value bits precede LL's
    const uint32_t kLlS = t == ZK_TAB_LL ? 0u : 0xFFu;              // LL state bits precede ML's and OF's
    const uint32_t kMlS = t == ZK_TAB_OF ? 0xFFu : 0u;              // ML state bits precede OF's
    // the number of a symbol's extra bits without a table in the way of the chain (the value table's entry is still read, for the
    // baseline, but nothing waits for it): 0 below code T1, a nibble of LUT up to T2, code - D from there; offsets: the code itself
    const uint32_t al_t = al[t];
    const uint32_t xT1 = t == ZK_TAB_LL ? 16u : t == ZK_TAB_ML ? 32u : 0u, xT2 = t == ZK_TAB_OF ? 0u : xT1 + 16u, xD = t == ZK_TAB_LL ? 19u : t == ZK_TAB_ML ? 36u : 0u;
    const uint64_t xLUT = t == ZK_TAB_LL ? 0xCBA9876433221111ull : 0xBA98754433221111ull;       // codes 16..31 (LL) / 32..47 (ML), low nibble first
    typename CP::cell_t c = cells[state];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(CP::touch(c)));                     // arrived before the loop: its waits then only count what the loop issues
#endif
    // one sequence; k: its place in the round
    auto step = [&](uint32_t i, uint32_t k) {
        uint32_t vv, xb, nbc;
        if constexpr (CP::kFat) {                               // 8-byte cells (small batches: LDS is no constraint): three field extractions
            vv = CP::baseline(c, nullptr);
            xb = CP::xbits(c, nullptr);
            nbc = CP::nb(c);
        } else {
            const uint32_t sy = CP::sym(c);
            vv = vt[sy];                                        // the baseline (used at the end of the step: nothing waits for it)
            const uint32_t xb_mid = (uint32_t)(xLUT >> ((4u * (sy - xT1)) & 63u)) & 15u;
            xb = sy < xT1 ? 0u : sy < xT2 ? xb_mid : sy - xD;
            nbc = CP::nb(c, al_t);
        }
        const uint32_t nb = i + 1 < nseq ? nbc : 0;
        const uint32_t pk = xb | (nb << 8);
        const uint32_t pL = XCH::bcast(pk, ZK_TAB_LL), pO = XCH::bcast(pk, ZK_TAB_OF), pM = XCH::bcast(pk, ZK_TAB_ML);
        const uint32_t sum = pL + pO + pM;                      // value bits in the low byte (<= 63), state bits above (<= 27)
        const uint32_t nval = sum & 0xFF, total = nval + (sum >> 8);
        const uint32_t voff = (pO & kOfV) + (pM & kMlV);
        const uint32_t soff = nval + ((pL >> 8) & kLlS) + ((pM >> 8) & kMlS);
        uint32_t vbits, sbits;                                  // (an offset code above 30 makes an offset value >= 2^31: zk_seq_finish_quad rejects it)
        if (total <= r.avail()) {
            const uint64_t X = r.window();
            r.consume(total);
            sbits = zk_top_bits((uint32_t)((X << (soff & 63)) >> 32), nb);
            vbits = zk_top_bits((uint32_t)((X << (voff & 63)) >> 32), xb & 31);
        } else {                                                // more bits than one window guarantees: field by field
            const uint32_t ofx = r.read(pO & 31), mlx = r.read(pM & 0xFF), llx = r.read(pL & 0xFF);
            const uint32_t sL = r.read(pL >> 8), sM = r.read(pM >> 8), sO = r.read(pO >> 8);
            bad |= r.remaining() < 0;
            r.clamp();
            vbits = t == ZK_TAB_LL ? llx : t == ZK_TAB_OF ? ofx : mlx;
            sbits = t == ZK_TAB_LL ? sL : t == ZK_TAB_OF ? sO : sM;
        }
        state = CP::base(c, al_t, nbc) + sbits;
        c = cells[state];                                       // issued early; used by the next step
        out.put(i - k, k, vv + vbits);
    };
    // (a round unrolled by hand, the ring slots constants: the same speed -- the guards of the unrolled steps cost what the slot
    // arithmetic saves; profiles/r04_fse_walk.txt)
    for (uint32_t i = 0; i < nseq; i++) {
        if ((i & (ZK_QUAD_ROUND - 1)) == 0) out.gate(i, b.src + bs_off + (uint32_t)(r.remaining() >> 3));
        step(i, i & (ZK_QUAD_ROUND - 1));
    }
    bad |= r.remaining() != 0;
    return bad ? 1u : 0u;
}

// What zk_seq_finish_quad carries from round to round (the same in all four lanes of a quad).
struct ZkSeqCarry { uint32_t rep0, rep1, rep2, lit, out, bad; };
ZK_HD void zk_seq_carry_init(ZkSeqCarry &c) { c.rep0 = zk_rep_sym(0); c.rep1 = zk_rep_sym(1); c.rep2 = zk_rep_sym(2); c.lit = 0; c.out = 0; c.bad = 0; }
// One round: lane j (0..3) of the quad brings the three values of sequence i0 + j (anything when j >= nvalid); nvalid (0..4) is
// the number of sequences of this round that exist, the same in all four lanes.  All four lanes run the block's recurrence over
// the round's sequences together -- the values come round by XCH::bcast -- and lane k keeps what belongs to ITS sequence: the
// record (returned; meaningless when j >= nvalid).  Offsets and the repeat history follow A.8 in select form, as zk_seq_walk.
template <typename XCH>
ZK_HD ZkSeqP zk_seq_finish_quad(uint32_t j, uint32_t nvalid, uint32_t ll, uint32_t ofv, uint32_t ml, ZkSeqCarry &c, uint32_t lit_regen)
{
    uint32_t my_out = 1, my_lit = 0, my_off = 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < ZK_QUAD_ROUND; k++) {
        const uint32_t llk = XCH::bcast(ll, (int)k), ofk = XCH::bcast(ofv, (int)k), mlk = XCH::bcast(ml, (int)k);
        const bool live = k < nvalid;
        const bool is_rep = ofk <= 3;
        const uint32_t idx = ofk - 1 + (llk == 0);
        const uint32_t r0m1 = zk_rep_is_sym(c.rep0) ? c.rep0 + 1 : c.rep0 - 1;
        uint32_t cand = idx == 0 ? c.rep0 : c.rep1;                 // plain selects: no control flow in the step
        cand = idx == 2 ? c.rep2 : cand;
        cand = idx == 3 ? r0m1 : cand;
        const uint32_t off = is_rep ? cand : ofk - 3;
        const bool sh1 = ((!is_rep) | (idx >= 1)) & live, sh2 = ((!is_rep) | (idx >= 2)) & live;
        c.rep2 = sh2 ? c.rep1 : c.rep2;
        c.rep1 = sh1 ? c.rep0 : c.rep1;
        c.rep0 = live ? off : c.rep0;
        c.lit += live ? llk : 0u;
        c.out += live ? llk + mlk : 0u;
        c.bad |= live & ((off == 0) | (c.lit > lit_regen) | (c.out > ZK_BLOCK_MAX) | (!zk_rep_is_sym(off) & (off > ZK_OFF_MAX)) | (ofk > 0x7FFFFFFFu));     // ofk >= 2^31: an offset code above 30
        const bool mine = j == k;
        my_out = mine ? c.out : my_out; my_lit = mine ? c.lit : my_lit; my_off = mine ? off : my_off;
    }
    return zk_seq_pack(my_out, my_lit, my_off);
}
// The block's verdict once every sequence went through zk_seq_finish_quad (walk_bad: what zk_seq_walk_quad returned).
ZK_HD void zk_seq_finish_block(ZkBlock &b, const ZkSeqCarry &c, uint32_t walk_bad)
{
    if (walk_bad | c.bad) { b.status = ZK_E_CORRUPTION; return; }
    const uint32_t out = c.out + (b.lit_regen - c.lit);
    if (out > ZK_BLOCK_MAX) { b.status = ZK_E_CORRUPTION; return; }
    b.out_size = out;
    b.rep_out[0] = c.rep0; b.rep_out[1] = c.rep1; b.rep_out[2] = c.rep2;
}

// ---------------------------------------------------------------- sequence execution: per-byte source map
// The executor produces a block's output in tiles of 16-byte slots.  Per tile:
//   * lane per SEQUENCE: for every slot whose first byte the sequence covers, slot_seq[slot] = staged index
//     (zk_exec_mark_slots);
//   * lane per SLOT: starting from slot_seq[slot] the lane walks the staged sequences over its 16 bytes and
//     produces one 32-bit source word per byte (zk_exec_slot_words), kept in registers and mirrored in LDS;
//   * every lane follows in-tile sources to their origin through the LDS mirror and gathers.
//   source word: bit 31 set -> literal index (block-relative);  else block-relative history position + 2^30
constexpr uint32_t ZK_SRC_LIT = 0x80000000u;
constexpr uint32_t ZK_SRC_BIAS = 0x40000000u;
// a raw-content prefix extends the history below the frame's first byte: positions down to -(2^30 - 2^27) stay
// representable next to block-relative positions of frames up to 128 MiB in; larger frames reach less of it
constexpr uint64_t ZK_MAX_PREFIX = (1ull << 30) - (1ull << 27);
constexpr uint32_t ZK_EXEC_SLOT = 16;        // bytes per slot (one 16-B store per lane)
constexpr uint32_t ZK_EXEC_LONG = 4;         // a sequence that starts more slots than this is marked by all lanes

// Slots (16-B aligned to the tile start ts) whose first byte lies in [lo, hi): first index and count.
ZK_HD void zk_exec_slot_span(uint32_t ts, uint32_t lo, uint32_t hi, uint32_t &s0, uint32_t &n)
{
    s0 = (lo - ts + (ZK_EXEC_SLOT - 1)) / ZK_EXEC_SLOT;
    const uint32_t s1 = (hi - ts + (ZK_EXEC_SLOT - 1)) / ZK_EXEC_SLOT;      // first slot starting at or after hi
    n = s1 > s0 ? s1 - s0 : 0;
}

// Per-sequence constants of the slot walk.
struct ZkSlotCur { uint32_t end, ms, litw, mbase, r, off; };
ZK_HD void zk_exec_slot_seq(const ZkSeq &e, uint32_t q, ZkSlotCur &c)
{
    c.end = e.out_end; c.ms = e.out_end - e.ml; c.off = e.off;
    c.litw = (ZK_SRC_LIT | e.lit_end) - c.ms;           // + q  (lit_end - (ms - q) never borrows into bit 31)
    c.mbase = ZK_SRC_BIAS + c.ms - e.off;               // + phase; phase wraps at off (match overlapping itself)
    uint32_t r = q > c.ms ? q - c.ms : 0;
    if (r >= e.off) r %= e.off;
    c.r = r;
}

ZK_HD void zk_exec_slot_words(const ZkSeq *S, uint32_t i, uint32_t q0, uint32_t nb, uint32_t *sw, uint32_t ring_mask);
// The common form of a slot's 16 source words.  Within a run -- the literals of a sequence, or its match unless the match
// overlaps its own output -- a byte's word is its position plus a constant: (LIT | lit_end) - ms for literals, BIAS - off for a
// match.  The lane that owns a SEQUENCE leaves, in the (emptied) map at the bytes where its two runs start, what each run ADDS to
// the constant of the run before it (zk_exec_mark_runs, part of the marking pass: the run before a sequence's literals is the
// match of the sequence before it, which the lane reads anyway); the lane that owns a SLOT starts from the constant of the
// sequence that covers its first byte and then takes ONE addition per byte, word[k] = word[k - 1] + 1 + map[k]
// (zk_exec_slot_words_marked) -- no walk over sequence ends, which used to be 16 steps with a divergent branch and an LDS round
// trip each (a third of the executor's time), and (r6) no compare-and-select per byte either (marks used to be the constants
// themselves: a compare, a select and an addition per byte, and a wait state between the first two).  A match that overlaps its
// own output (offset < length: its bytes repeat with the period of the offset) flags the slots it touches in a bitmap of the
// tile's slots: those take the general walk (zk_exec_slot_words).  A mark that falls on a slot's FIRST byte is never read: that
// byte's constant comes from the covering sequence.
// WHERE a position's word lies in the map.  A lane reads and writes the sixteen words of its slot as four 16-byte accesses; laid out
// in position order, consecutive lanes are 64 bytes apart and eight of them (one pass of the LDS: 8 x 16 bytes = its 32 banks) use
// two groups of four banks -- every such access runs four times (the map is emptied, read and written that way every tile: twelve
// accesses per lane).  So the 16-byte chunk c of the map lies at chunk c ^ ((c >> 3) & 7): inside every 128 bytes the chunks are
// permuted by the number of their 128-byte row, eight consecutive lanes' chunks fall into eight different groups of banks, a lane's own
// chunks stay 16-byte units at one xor from each other.  What pays: the accesses BY POSITION (marks, the chase), three instructions
// instead of one for the address.
ZK_HD uint32_t zk_exec_map_index(uint32_t p) { return p ^ (((p >> 5) & 7u) << 2); }
ZK_HD uint32_t zk_exec_lit_const(const ZkSeq &e) { return (ZK_SRC_LIT | e.lit_end) - (e.out_end - e.ml); }
ZK_HD uint32_t zk_exec_match_const(uint32_t off) { return ZK_SRC_BIAS - off; }
// e: the sequence, start: where its literals start (= the end of the sequence before it), prev_off: that sequence's offset (any
// value when start <= ts: no mark depends on it then); slow_or(word, bits): or into the tile's bitmap of slots for the general walk
template <typename OR>
ZK_HD void zk_exec_mark_runs(const ZkSeq &e, uint32_t prev_off, uint32_t start, uint32_t ts, uint32_t te, uint32_t *map, OR &&slow_or)
{
    const uint32_t ms = e.out_end - e.ml;
    const uint32_t cl = zk_exec_lit_const(e), cm = zk_exec_match_const(e.off), cp = zk_exec_match_const(prev_off);
    if (ms > start && start > ts && start < te) map[zk_exec_map_index(start - ts)] = cl - cp;
    if (e.ml && ms > ts && ms < te) map[zk_exec_map_index(ms - ts)] = cm - (ms > start ? cl : cp);
    if (e.ml && e.off < e.ml) {
        const uint32_t lo = ms > ts ? ms : ts, hi = e.out_end < te ? e.out_end : te;
        if (lo < hi) {
            const uint32_t s_lo = (lo - ts) / ZK_EXEC_SLOT, s_hi = (hi - 1 - ts) / ZK_EXEC_SLOT;       // slots [s_lo, s_hi] hold bytes of the match
            for (uint32_t w = s_lo >> 5; w <= (s_hi >> 5); w++) {
                const uint32_t b0 = s_lo > w * 32u ? s_lo - w * 32u : 0u, b1 = s_hi < w * 32u + 31u ? s_hi - w * 32u : 31u;
                slow_or(w, (0xFFFFFFFFu >> (31u - b1)) & (0xFFFFFFFFu << b0));
            }
        }
    }
}
// mk: the slot's 16 words of the map after the marking pass; i0: the sequence that covers q0; slow: the slot's bit of the bitmap
ZK_HD void zk_exec_slot_words_marked(const ZkSeq *S, uint32_t i0, bool slow, uint32_t q0, uint32_t nb, const uint32_t *mk, uint32_t *sw, uint32_t ring_mask = 0xFFFFFFFFu)
{
    if (slow) { zk_exec_slot_words(S, i0, q0, nb, sw, ring_mask); return; }
    const ZkSeq e0 = S[i0 & ring_mask];
    uint32_t w = q0 + (q0 >= e0.out_end - e0.ml ? zk_exec_match_const(e0.off) : zk_exec_lit_const(e0));
    sw[0] = w;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 1; k < ZK_EXEC_SLOT; k++) { w += mk[k] + 1u; sw[k] = w; }
    if (nb < ZK_EXEC_SLOT) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t k = 0; k < ZK_EXEC_SLOT; k++) sw[k] = k < nb ? sw[k] : ZK_SRC_LIT;
    }
}
ZK_HD void zk_exec_slot_words(const ZkSeq *S, uint32_t i, uint32_t q0, uint32_t nb, uint32_t *sw, uint32_t ring_mask)
{
    ZkSlotCur c;
    zk_exec_slot_seq(S[i & ring_mask], q0, c);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < ZK_EXEC_SLOT; k++) {
        const uint32_t q = q0 + k;
        if (k < nb) {
            if (q >= c.end) { i++; zk_exec_slot_seq(S[i & ring_mask], q, c); }
            const bool m = q >= c.ms;
            sw[k] = m ? c.mbase + c.r : c.litw + q;
            c.r += m ? 1u : 0u;
            if (c.r == c.off) c.r = 0;
        } else sw[k] = ZK_SRC_LIT;
    }
}

// Follow in-tile sources: returns a word that is a literal or a history position before the tile.
ZK_HD uint32_t zk_exec_origin(const uint32_t *srcmap, uint32_t s, uint32_t ts)
{
    while (!(s & ZK_SRC_LIT) && (int32_t)(s - ZK_SRC_BIAS) >= (int32_t)ts) s = srcmap[zk_exec_map_index((s - ZK_SRC_BIAS) - ts)];
    return s;
}

// ---------------------------------------------------------------- sequence execution in SEGMENTS (two passes)
// One workgroup per frame makes a frame a serial chain of tiles: a handful of frames leaves the device idle, and 1 280 resident frames
// of a 2 MiB window keep 2.5 GB of history live (every far match a line from HBM).  Blocks cannot simply be dealt to several
// workgroups -- a block's first bytes copy from the END of the block before it -- so the split is by DEPENDENCE:
//   pass 1 (zk_k_exec_seg): a frame is cut into SEGMENTS of whole blocks (<= ZK_SEG_BYTES of output each, zk_seg_step), every segment
//     a workgroup of its own, all at once.  It runs the tile machinery above with one change: a byte whose origin lies BEFORE its
//     segment -- or in an earlier tile of the segment at a byte that is itself such a byte (one taint bit per byte of the segment, in
//     LDS) -- is a HOLE: it is not written, a record (dst, offset, length <= 16) per run of holes is left in HBM.  Everything else of
//     the segment (literals and whatever copies from them, however indirectly) is final after this pass;
//   pass 2 (zk_k_exec_fill): one workgroup per frame walks the frame's hole records in order, L at a time, and copies the runs.  A
//     record's source lies before its own TILE (in-tile sources were followed to their origin by pass 1) and the list is in tile
//     order, so of a group of L records every one whose source ends at or below the group's lowest destination is independent of the
//     group; the group is cut at the first one that is not (the first lane never is: zk_fill_ready).
// Hole record: dst[29:0] (frame-relative) | (len - 1)[33:30] | off[63:34] (dst - src, 1 .. 2^30 - 1).
typedef uint64_t ZkHole;
constexpr uint32_t ZK_SEG_BYTES = 131072;          // a segment's output never exceeds this unless a single block does (and none does)
constexpr uint32_t ZK_E_SEG_OVERFLOW = 0x5E60u;    // internal: a segment left more hole records than its region holds -- the frame is executed again by zk_k_exec
struct ZkSeg {                   // one per segment, written by zk_k_seg_prep
    uint32_t b0, nb;             // blocks [b0, b0 + nb) of the frame
    uint32_t pos, out;           // frame-relative first byte, bytes
    uint32_t rep[3];             // repeat offsets at the segment's first block (concrete)
    uint32_t pad;
};
static_assert(sizeof(ZkSeg) == 32, "ZkSeg layout");
ZK_HD ZkHole zk_hole_pack(uint32_t dst, uint32_t len, uint32_t off) { return (uint64_t)dst | ((uint64_t)(len - 1u) << 30) | ((uint64_t)off << 34); }
ZK_HD uint32_t zk_hole_dst(ZkHole r) { return (uint32_t)r & 0x3FFFFFFFu; }
ZK_HD uint32_t zk_hole_len(ZkHole r) { return ((uint32_t)(r >> 30) & 15u) + 1u; }
ZK_HD uint32_t zk_hole_off(ZkHole r) { return (uint32_t)(r >> 34); }
// where segment j's records live: region index (in records) and capacity, from the frame's place in the batch's output (frame_off),
// the segment's place in the frame and its running number in the batch (f * max_segs + j)
// (a slot per 4 bytes of output: at the start of a segment nearly every match is a hole, a run per ~7 bytes on text; archives of
//  libzstd's level 3 leave a run per 5.9 bytes of a whole 128 KiB segment)
ZK_HD uint64_t zk_seg_region(uint64_t frame_off, uint32_t pos, uint64_t seg_no) { return ((frame_off + pos) >> 2) + seg_no * 16u; }
ZK_HD uint32_t zk_seg_region_cap(uint32_t out) { return (out >> 2) + 8u; }

// The serial walk over a frame's blocks that cuts it into segments (one lane's work; zk_k_seg_prep feeds it from a wave's registers).
struct ZkSegWalk { uint32_t pos, rep[3], b0, nb, seg_pos, seg_rep[3], nsegs, err; };
ZK_HD void zk_seg_walk_init(ZkSegWalk &w)
{
    w.pos = 0; w.rep[0] = 1; w.rep[1] = 4; w.rep[2] = 8; w.b0 = 0; w.nb = 0; w.seg_pos = 0;
    w.seg_rep[0] = 1; w.seg_rep[1] = 4; w.seg_rep[2] = 8; w.nsegs = 0; w.err = ZK_OK;
}
ZK_HD void zk_seg_emit(ZkSegWalk &w, ZkSeg *segs, uint32_t max_segs)
{
    if (w.nsegs < max_segs) {
        ZkSeg s;
        s.b0 = w.b0; s.nb = w.nb; s.pos = w.seg_pos; s.out = w.pos - w.seg_pos;
        s.rep[0] = w.seg_rep[0]; s.rep[1] = w.seg_rep[1]; s.rep[2] = w.seg_rep[2]; s.pad = 0;
        segs[w.nsegs] = s;
    } else w.err = ZK_E_GENERIC;                            // (cannot happen: max_segs is 2 * ceil(largest frame / seg_bytes) + 1)
    w.nsegs++;
}
// block bk of the frame: its status / regenerated size / symbolic repeat history as the entropy stage left them
ZK_HD void zk_seg_step(ZkSegWalk &w, uint32_t bk, uint32_t b_status, uint32_t b_out, const uint32_t b_rep[3], uint64_t d_size, uint32_t block_max,
                       uint32_t seg_bytes, ZkSeg *segs, uint32_t max_segs)
{
    if (w.err != ZK_OK) return;
    if (b_status != ZK_OK) { w.err = b_status; return; }
    if ((uint64_t)w.pos + b_out > d_size || b_out > block_max) { w.err = ZK_E_CORRUPTION; return; }     // as zk_k_exec
    if (w.nb && w.pos - w.seg_pos + b_out > seg_bytes) {    // the segment is full: this block opens the next one
        zk_seg_emit(w, segs, max_segs);
        w.b0 = bk; w.nb = 0; w.seg_pos = w.pos;
        w.seg_rep[0] = w.rep[0]; w.seg_rep[1] = w.rep[1]; w.seg_rep[2] = w.rep[2];
    }
    w.nb++;
    w.pos += b_out;
    const uint32_t r0 = zk_rep_resolve(b_rep[0], w.rep), r1 = zk_rep_resolve(b_rep[1], w.rep), r2 = zk_rep_resolve(b_rep[2], w.rep);
    w.rep[0] = r0; w.rep[1] = r1; w.rep[2] = r2;
}
ZK_HD void zk_seg_walk_end(ZkSegWalk &w, uint64_t d_size, ZkSeg *segs, uint32_t max_segs)
{
    if (w.err != ZK_OK) return;
    if (w.pos != d_size) { w.err = ZK_E_CORRUPTION; return; }
    if (w.nb) zk_seg_emit(w, segs, max_segs);
}

// A slot's 16 source words after the chase (each a literal or a history position before the tile) -> which bytes are holes.
// seg_lo: block-relative position of the segment's first byte (<= 0); tainted(p): the taint bit of segment-relative byte p.
template <typename TAINT>
ZK_HD uint32_t zk_seg_slot_holes(const uint32_t *sw, uint32_t nb, int32_t seg_lo, TAINT tainted)
{
    uint32_t hm = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < ZK_EXEC_SLOT; k++) {
        const uint32_t s = sw[k];
        const int32_t rel = (int32_t)(s - ZK_SRC_BIAS);
        bool h = false;
        if (k < nb && !(s & ZK_SRC_LIT)) h = rel < seg_lo ? true : tainted((uint32_t)(rel - seg_lo));
        hm |= h ? 1u << k : 0u;
    }
    return hm;
}
// Runs of holes with consecutive sources: bit k of the result = a run starts at byte k, len[k] its length (defined where the bit is set).
ZK_HD uint32_t zk_seg_slot_runs(const uint32_t *sw, uint32_t hm, uint32_t *len)
{
    uint32_t starts = 0, run = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = (int)ZK_EXEC_SLOT - 1; k >= 0; k--) {
        const bool h = (hm >> k) & 1u;
        const bool next_joins = k + 1 < (int)ZK_EXEC_SLOT && ((hm >> (k + 1)) & 1u) && sw[(k + 1) & 15] == sw[k] + 1u;
        const bool joins_prev = k > 0 && ((hm >> (k - 1)) & 1u) && sw[k] == sw[(k - 1) & 15] + 1u;
        run = h ? (next_joins ? run + 1u : 1u) : 0u;
        len[k] = run;
        starts |= (h && !joins_prev) ? 1u << k : 0u;
    }
    return starts;
}
// pass 2, one record of a group whose lowest destination is dmin: may it be copied before the group's other records have been?
ZK_HD bool zk_fill_ready(ZkHole r, uint32_t dmin) { return zk_hole_dst(r) - zk_hole_off(r) + zk_hole_len(r) <= dmin; }
