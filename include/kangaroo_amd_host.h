/*
 * kangaroo_amd_host.h — host-side helpers that sit either side of the jump engine
 * (SURVEY §8f): curve set-up the reference does on the CPU before the first launch and the
 * DP bookkeeping it does after each one.  Plain C ABI, same conventions as kangaroo_amd.h
 * (256-bit values = four little-endian uint64_t limbs).
 *
 * These are NOT part of the GPUEngine boundary; the reference's own host code
 * (core/kangaroo.cpp, memory/hashtable.cpp) can keep doing this work unchanged.  They exist
 * so that the engine can be driven, benchmarked and used to solve keys on a machine where
 * the reference's host sources are absent.
 */
#ifndef KANGAROO_AMD_HOST_H
#define KANGAROO_AMD_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* k*G, affine.  Secp256K1::ComputePublicKey (math/secp256k1.cpp:36-58).  Returns 0 for k = 0. */
int kgh_ec_mul_g(uint64_t x[4], uint64_t y[4], const uint64_t k[4]);
/* n scalars at once.  Secp256K1::ComputePublicKeys (math/secp256k1.cpp:60-82). */
int kgh_ec_mul_g_batch(uint64_t *x, uint64_t *y, const uint64_t *k, size_t n);
/* Affine P1 + P2 (P1 != +-P2).  Secp256K1::AddDirect (math/secp256k1.cpp:196-216). */
int kgh_ec_add(uint64_t rx[4], uint64_t ry[4], const uint64_t x1[4], const uint64_t y1[4],
               const uint64_t x2[4], const uint64_t y2[4]);
/* 33/65-byte hex public key -> point.  Secp256K1::ParsePublicKeyHex (math/secp256k1.cpp:109-168).
 * Returns 0 on success, negative on malformed input / not on curve. */
int kgh_parse_pubkey(const char *hex, uint64_t x[4], uint64_t y[4]);

/* Kangaroo::CreateJumpTable (core/kangaroo.cpp:575-626): seed 0x600DCAFE, 64 jumps of
 * rangePower/2+1 bits.  Arrays are 64 x 4 limbs.  Returns the number of tries used. */
int kgh_create_jump_table(int range_power, uint64_t *jd, uint64_t *jpx, uint64_t *jpy);

/* Opaque std::mt19937 stream equal to the reference's rseed()/rndl() (math/random.h:10-43). */
typedef struct kgh_rng kgh_rng;
kgh_rng *kgh_rng_new(uint32_t seed);
void kgh_rng_free(kgh_rng *r);
uint32_t kgh_rng_next(kgh_rng *r);

/* Kangaroo::CreateHerd (core/kangaroo.cpp:521-573): alternating tame / wild kangaroos starting with
 * first_type, n entries.  key = search key already shifted by -rangeStart*G (InitSearchKey). */
int kgh_create_herd(kgh_rng *r, int range_power, const uint64_t range_width[4], const uint64_t key_x[4],
                    const uint64_t key_y[4], int first_type, size_t n, uint64_t *px, uint64_t *py, uint64_t *d);

/* The distance half of CreateHerd alone (same RNG draws, core/kangaroo.cpp:531-548): the points can then be
 * computed on the GPU with kg_herd_from_distances (kangaroo_amd.h). */
int kgh_herd_distances(kgh_rng *r, int range_power, const uint64_t range_width[4], int first_type, size_t n,
                       uint64_t *d);
/* Fixed-base comb table for the GPU herd kernel: tab[32][256][18] limbs, plus the start point. */
int kgh_comb_table(uint32_t *tab, uint64_t start_x[4], uint64_t start_y[4]);

/* Fast synthetic herd for throughput runs (SURVEY §8d "any valid curve points"): d_i = d0 + i*stride
 * (mod n), P_i = d_i*G, built with batched affine additions (about 1-2 M points/s per host core).
 * No counterpart in the reference. */
int kgh_synthetic_herd(const uint64_t d0[4], const uint64_t stride[4], size_t n, uint64_t *px, uint64_t *py,
                       uint64_t *d);

/* ---- DP table + collision resolution (memory/hashtable.cpp:65-226, core/kangaroo.cpp:155-256) ---- */
typedef struct kgh_solver kgh_solver;
/* range_start/range_end/pub as in the reference's input file (core/kangaroo.cpp:43-102). */
kgh_solver *kgh_solver_new(const uint64_t range_start[4], const uint64_t range_end[4], const uint64_t pub_x[4],
                           const uint64_t pub_y[4]);
void kgh_solver_free(kgh_solver *s);
int kgh_solver_range_power(const kgh_solver *s);
/* rangeWidth/2: the value the reference passes to SetWildOffset (core/kangaroo.cpp:436). */
void kgh_solver_wild_offset(const kgh_solver *s, uint64_t out[4]);
void kgh_solver_range_width(const kgh_solver *s, uint64_t out[4]);
/* search key shifted into [0, width): keyToSearch of InitSearchKey (core/kangaroo.cpp:662-674). */
void kgh_solver_search_key(const kgh_solver *s, uint64_t x[4], uint64_t y[4]);
/* Kangaroo::AddToTable for one DP (thread-safe: takes the table mutex like ghMutex,
 * core/kangaroo.cpp:475).  Returns 1 = stored, 0 = same-herd collision or duplicate (caller must
 * re-seed kangaroo k_idx, core/kangaroo.cpp:478-486), 2 = key found. */
int kgh_solver_add(kgh_solver *s, const uint64_t x[4], const uint64_t d[4], uint64_t k_idx);
/* Batch form: items laid out as kg_item (72 bytes).  dead[] receives the indices (into items) that
 * need a reset, *n_dead their count.  Returns 2 as soon as the key is found, else 0. */
int kgh_solver_add_items(kgh_solver *s, const void *items, uint32_t n, uint32_t *dead, uint32_t *n_dead);
/* Worker threads used for batches of 8192 items and more (0 = hardware concurrency, at most 16).  The table is
 * cut into 256 shards by x, so several callers (one per GPU) may also add batches concurrently. */
void kgh_solver_set_threads(kgh_solver *s, int threads);
int kgh_solver_found(const kgh_solver *s, uint64_t priv[4]);
uint64_t kgh_solver_table_size(const kgh_solver *s);
uint64_t kgh_solver_dead_count(const kgh_solver *s);
/* Items handed in that are neither in the table nor dead (the DP that solved the key, what arrived after it):
 * table_size + dead_count + unstored_count == items handed in. */
uint64_t kgh_solver_unstored_count(const kgh_solver *s);
/* Dump the table as (h, X[4], D[4]) records of 9 uint64 each, HashTable::Convert layout
 * (memory/hashtable.cpp:65-89); returns the number written (<= cap). */
uint64_t kgh_solver_dump(const kgh_solver *s, uint64_t *out, uint64_t cap);

/* ---- work file, the reference's HEADW layout (storage/backup.cpp:256-428, memory/hashtable.cpp:262-324) ---- */
/* Kangaroo::SaveWork: header, the DP table, and (if px/py/d are given) the kangaroos.  0 on success. */
int kgh_work_save(const kgh_solver *s, const char *path, uint32_t dp_size, uint64_t total_count, double total_time,
                  const uint64_t *px, const uint64_t *py, const uint64_t *d, uint64_t n_kangaroo);
/* Kangaroo::WorkInfo (storage/backup.cpp:430-...): header fields and item counts without loading. */
int kgh_work_info(const char *path, uint32_t *dp_size, uint64_t range_start[4], uint64_t range_end[4], uint64_t key_x[4],
                  uint64_t key_y[4], uint64_t *total_count, double *total_time, uint64_t *n_items, uint64_t *n_kangaroo);
/* Kangaroo::LoadWork + FetchWalks (storage/backup.cpp:54-253): the table into s, up to cap kangaroos into px/py/d.
 * -5 if the file belongs to another range or key. */
int kgh_work_load(kgh_solver *s, const char *path, uint64_t *px, uint64_t *py, uint64_t *d, uint64_t cap,
                  uint64_t *n_loaded);

#ifdef __cplusplus
}
#endif
#endif
