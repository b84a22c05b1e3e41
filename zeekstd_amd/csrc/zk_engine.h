// zk_engine.h -- engine object shared by the decode / encode halves of the C ABI
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "zk_enc_device.h"

enum { ZK_K_WALK_COUNT = 0, ZK_K_SCAN, ZK_K_WALK_FILL, ZK_K_HUF, ZK_K_FSE, ZK_K_EXEC, ZK_K_XXH64, ZK_K_STATUS,
       ZK_K_ENC_MATCH, ZK_K_ENC_ENTROPY, ZK_K_ENC_COMPACT, ZK_K_ENC_XXH64, ZK_NKERNELS };

struct zk_devbuf { void *p = nullptr; size_t cap = 0; };

struct zk_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;              // second queue: kernels with no mutual dependency overlap (huf || fse)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    char devname[320] = {0};
    std::string last_err;
    uint64_t *h_words = nullptr;            // pinned: small read-backs (totals, first error)
    // decode scratch
    zk_devbuf infos, bases, words, blocks, seqs, lit;
    // second decode context (zk_decode_submit_dev): own queues, scratch and read-back words, so that two batches can
    // be in flight -- the tail of one (checksum kernel: a per-frame serial chain) overlaps the head of the next
    hipStream_t stream2 = nullptr, aux2 = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    zk_devbuf infos2, bases2, words2, blocks2, seqs2, lit2;
    uint64_t *h_words2 = nullptr;
    bool slot_busy[2] = {false, false};
    int next_slot = 0;
    // staging for the host-pointer entry points
    zk_devbuf st_comp, st_off, st_dst, st_misc;
    zk_devbuf st_prefix;                    // staged prefix of zk_decode_frames_prefix / zk_encode_frames_prefix, kept between calls
    const void *st_prefix_src = nullptr; uint64_t st_prefix_len = 0, st_prefix_fp = 0;
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool profiling = false;
    int fse_kernel = 0;              // zk_engine_set_fse_kernel
    hipEvent_t ev_start[ZK_NKERNELS] = {}, ev_stop[ZK_NKERNELS] = {};
    bool ev_used[ZK_NKERNELS] = {};
    float kernel_ms[ZK_NKERNELS] = {};
    // encode scratch
    zk_devbuf enc_a, enc_b, enc_c, enc_d;
    zk_devbuf enc_hist;                     // prefix mode: [prefix tail | frame] records for the matcher
    ZkEncTables enc_tables;
    bool enc_tables_ready = false;
};

int zk_devbuf_reserve(zk_engine *e, zk_devbuf &b, size_t bytes);

// RAII-free helper: brackets one launch with events when profiling is on
struct zk_kernel_timer {
    zk_engine *e; int k; hipStream_t st;
    zk_kernel_timer(zk_engine *e_, int k_, hipStream_t st_) : e(e_), k(k_), st(st_) {
        if (e->profiling) { (void)hipEventRecord(e->ev_start[k], st); e->ev_used[k] = true; }
    }
    ~zk_kernel_timer() { if (e->profiling) (void)hipEventRecord(e->ev_stop[k], st); }
};
void zk_profile_begin(zk_engine *e);
void zk_profile_collect(zk_engine *e);
