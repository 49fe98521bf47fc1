/* gl_oracle.h — CPU oracle for the plonky2 prover hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference's CPU algorithms for the path named in
 * BASELINE.json (Goldilocks NTT/LDE, Poseidon Merkle caps, PolynomialBatch commit). It exists
 * to CHECK the HIP product path and to provide the `cpu_baseline` leg of bench.py. Nothing in
 * the product (plonky2_gpu_amd/, include/) may include, link or call it.
 *
 * Parity pin: the reference (Rust, nightly, crates.io deps) cannot be built in this image, so the
 * oracle is pinned by the reference's own in-tree known answers (tests/test_oracle_*.py):
 *   - 4 Poseidon permutation vectors      plonky2/src/hash/poseidon_goldilocks.rs:286-309
 *   - fast == naive partial rounds         plonky2/src/hash/poseidon.rs:736-749
 *   - bit-reversal tables                  plonky2/src/util/mod.rs:70-102
 *   - field-op cross-checks vs big ints    field/src/prime_field_testing.rs:7-17, 79-180
 *   - fft == naive evaluation, r=0..3      field/src/fft.rs:242-309
 *   - coset fft/ifft vs naive              field/src/polynomial/mod.rs:482-522
 *   - every Merkle proof verifies          plonky2/src/hash/merkle_tree.rs:456-514
 * and by an independent Python big-int model (oracle/pyref.py).
 *
 * All values are Goldilocks field elements stored as uint64_t; like the reference
 * (field/src/goldilocks_field.rs:26) any u64 is a legal representative. Functions whose name
 * ends in _canon return canonical values.
 */
#ifndef GL_ORACLE_H
#define GL_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPSILON 0xFFFFFFFFULL

/* ---- field (field/src/goldilocks_field.rs, field/src/types.rs) ---- */
uint64_t glo_add(uint64_t a, uint64_t b);        /* goldilocks_field.rs:197-219 */
uint64_t glo_sub(uint64_t a, uint64_t b);        /* :234-256 */
uint64_t glo_neg(uint64_t a);                    /* :184-195 */
uint64_t glo_mul(uint64_t a, uint64_t b);        /* :265-272 + reduce128 :345-358 */
uint64_t glo_canon(uint64_t a);                  /* to_canonical_u64 :169-176 */
uint64_t glo_mac(uint64_t acc, uint64_t x, uint64_t y); /* multiply_accumulate :119-123 */
uint64_t glo_exp(uint64_t base, uint64_t power); /* types.rs:361-372 */
uint64_t glo_inverse(uint64_t a);                /* value of try_inverse (inversion.rs:66); Fermat */
uint64_t glo_inverse_2exp(unsigned exp);         /* types.rs:227-266 */
uint64_t glo_primitive_root_of_unity(unsigned n_log); /* types.rs:268-272 */

/* ---- bit reversal / transpose (util/src/lib.rs:188-237, plonky2/src/util/mod.rs:23-63) ---- */
size_t glo_reverse_bits(size_t n, unsigned num_bits);
void glo_reverse_index_bits_in_place(uint64_t *v, size_t n);
void glo_reverse_index_bits_rows_in_place(uint64_t *rows, size_t n_rows, size_t row_len);
void glo_transpose(const uint64_t *src, uint64_t *dst, size_t rows, size_t cols);

/* ---- FFT (field/src/fft.rs) ---- */
/* fft_root_table(n).concat(): lg n rows, row s = powers of w_{2^(s+1)}, max(2^s,2) entries
 * (fft.rs:15-34). Returns number of u64 written (n for n>=2... row 0 has 2). out may be NULL. */
size_t glo_fft_root_table_concat(size_t n, uint64_t *out);
/* fft_with_options(.., zero_factor=r, root_table=None): natural in -> natural out (fft.rs:58-66,188-229) */
void glo_fft(uint64_t *v, size_t n, unsigned r);
/* ifft_with_options (fft.rs:73-103) */
void glo_ifft(uint64_t *v, size_t n);
/* PolynomialCoeffs::lde(rate_bits).coset_fft_with_options(shift, Some(rate_bits)) (polynomial/mod.rs:205-207, 286-299).
 * coeffs: n -> out: n<<rate_bits */
void glo_coset_lde(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out);
/* PolynomialCoeffs::coset_fft (no padding) */
void glo_coset_fft(uint64_t *v, size_t n, uint64_t shift);
/* PolynomialValues::coset_ifft (polynomial/mod.rs:64-77) */
void glo_coset_ifft(uint64_t *v, size_t n, uint64_t shift);

/* the same transforms with a root table built once (fft_root_table, fft.rs:15-34; built once per circuit in the reference) */
void *glo_root_table_new(size_t n);
void glo_root_table_free(void *table);
void glo_fft_with_table(uint64_t *v, size_t n, unsigned r, const void *table);
void glo_ifft_with_table(uint64_t *v, size_t n, const void *table);
void glo_coset_lde_with_table(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out, const void *table_ext);

/* ---- Poseidon (plonky2/src/hash/poseidon.rs, hashing.rs, plonk/config.rs) ---- */
void glo_poseidon(uint64_t state[12]);        /* poseidon.rs:602-616 (fast partial rounds) */
void glo_poseidon_naive(uint64_t state[12]);  /* poseidon.rs:631-640 */
void glo_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]);  /* hashing.rs:81-108 */
void glo_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]); /* plonk/config.rs:56-67 */
void glo_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]); /* hashing.rs:65-72 */

/* ---- Merkle tree (plonky2/src/hash/merkle_tree.rs) ---- */
/* leaves: leaf-major [n_leaves][leaf_len]. digests: 4*2*(n_leaves - 2^cap_height) u64 in the
 * reference's recursive layout (merkle_tree.rs:46-54, 78-105); cap: 4*2^cap_height u64.
 * Returns 0, or -1 if cap_height > log2(n_leaves) (the reference panics, :285-290). */
int glo_merkle_tree(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                    uint64_t *digests, uint64_t *cap, int n_threads);
/* MerkleTree::prove (merkle_tree.rs:392-440): writes 4*num_layers u64, returns num_layers */
unsigned glo_merkle_prove(const uint64_t *digests, size_t n_leaves, unsigned cap_height,
                          size_t leaf_index, uint64_t *siblings);
/* verify_merkle_proof_to_cap (merkle_proofs.rs:53-80): 1 if valid */
int glo_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t leaf_index, const uint64_t *cap,
                      const uint64_t *siblings, unsigned num_layers);

/* ---- PolynomialBatch commit (plonky2/src/fri/oracle.rs:709-731, 911-1004) ---- */
/* values: column-major [n_polys][n] (evaluations on H). Outputs (any may be NULL):
 *   coeffs  [n_polys][n]           (PolynomialBatch.polynomials)
 *   leaves  [n<<rate][n_polys]     (merkle_tree.leaves: transposed, bit-reversed LDE; no salt)
 *   digests, cap as glo_merkle_tree.
 * blinding is not modelled (salt = OS randomness in the reference, oracle.rs:998-1002). */
int glo_commit_from_values(const uint64_t *values, size_t n_polys, size_t n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *coeffs, uint64_t *leaves,
                           uint64_t *digests, uint64_t *cap, int n_threads);
int glo_commit_from_coeffs(const uint64_t *coeffs, size_t n_polys, size_t n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *leaves, uint64_t *digests, uint64_t *cap,
                           int n_threads);

/* batch helpers used by the cpu_baseline timing (one task per column, like rayon par_iter) */
void glo_fft_batch(uint64_t *v, size_t n_polys, size_t n, int inverse, int n_threads);
void glo_coset_lde_batch(const uint64_t *coeffs, size_t n_polys, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out, int n_threads);
/* timed inside C, root table prebuilt, thread-local first-touched columns: see gl_oracle.c */
double glo_fft_bench(size_t n, int n_threads, int cols_per_thread, uint64_t seed, uint64_t *checksum);
int glo_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
