/*
 * ref_shim.c -- ORACLE helper.  The reference keeps its hash functions `static inline` in a
 * private header and instantiates khashl inside htab.c, so known-answer values for them can only be
 * taken by compiling a translation unit that #includes the reference headers from where they lie
 * (-I/root/reference).  This file contains no reference code, only calls; it is built into
 * oracle/_ref/libyakshim.so by `make ref` and used by tests/gen_golden.py to pin the KATs.
 */
#include <stdint.h>
#include "yak-priv.h"
#include "khashl.h"

uint64_t shim_hash64(uint64_t key, uint64_t mask) { return yak_hash64(key, mask); }
uint64_t shim_hash64_64(uint64_t key) { return yak_hash64_64(key); }
uint64_t shim_hash_long(uint64_t *x) { return yak_hash_long(x); }
uint64_t shim_hash64_inv(uint64_t key, uint64_t mask) { return yak_hash64_inv(key, mask); }
uint32_t shim_h2b(uint32_t hash, uint32_t bits) { return __kh_h2b(hash, bits); }
