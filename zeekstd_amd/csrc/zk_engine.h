// zk_engine.h -- engine object shared by the decode / encode halves of the C ABI
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

struct zk_devbuf { void *p = nullptr; size_t cap = 0; };

struct zk_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    char devname[320] = {0};
    std::string last_err;
    uint64_t *h_words = nullptr;            // pinned: small read-backs (totals, first error)
    // decode scratch
    zk_devbuf infos, bases, words, blocks, seqs, lit;
    // staging for the host-pointer entry points
    zk_devbuf st_comp, st_off, st_dst, st_misc;
    // encode scratch
    zk_devbuf enc_a, enc_b, enc_c, enc_d;
};

int zk_devbuf_reserve(zk_engine *e, zk_devbuf &b, size_t bytes);
