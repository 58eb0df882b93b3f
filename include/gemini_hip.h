/*
 * gemini_hip.h -- C ABI of libgemini_hip.so, the MI355X (gfx950) implementation of Gemini's
 * prover hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * arkworks-rs/gemini checkout; ark-ec/ark-ff are third-party crates pinned in Cargo.lock:44-64).
 * INTEGRATION.md shows the Rust `extern "C"` block and the feature-gated call sites.
 *
 * Conventions
 *   - Fr elements: 4 x u64 little-endian limbs.  "mont" = Montgomery form a*2^256 mod r, exactly
 *     the in-memory image of ark-ff `Fr` (zero-copy from `&[Fr]`).  "canonical" = the integer
 *     itself, i.e. `Fr::into_bigint()` / `BigInt<4>`.
 *   - G1 affine bases: x, y as 6 x u64 Montgomery limbs each (96 bytes).  `stride` >= 96; if
 *     stride >= 97, byte 96 is ark-ec's `infinity: bool` (Rust `G1Affine` has stride 104).
 *     With stride 96 the identity is encoded as x = y = 0.
 *   - G1 results: Jacobian X, Y, Z (18 x u64, Montgomery) = the image of ark-ec
 *     `Projective<g1::Config>`.  Results are normalised (Z = R, or (R, R, 0) for the identity)
 *     so equal group elements always produce equal bytes.
 *   - Return value: 0 = ok, negative = GM_E*.  Nothing throws or aborts across the boundary;
 *     gm_last_error() gives a message for the calling thread.
 *   - Ownership: the caller owns every host buffer for the duration of the call only; registered
 *     bases and vectors are copied to device memory owned by the library until *_free.
 *   - Threading: one process per GPU.  Every entry point may be called from any thread.  Calls on
 *     different prover handles run concurrently (sumcheck::prove_batch calls next_message on distinct
 *     provers from different rayon threads, src/subprotocols/sumcheck/proof.rs:85; `Prover: Send + Sync`,
 *     prover.rs:30); calls on the SAME prover handle are serialised by a per-handle lock.  MSM calls
 *     (gm_g1_msm*, herring G1 rounds) share the device workspaces and are serialised by one library
 *     lock held from the staging of host scalars to the result (the reference's MSM calls are
 *     sequential too, src/kzg/time.rs:103-106); the vector entry points (gm_fr_*) are serialised by a
 *     second lock because they stage per-call parameters in one scratch buffer.  Results never depend
 *     on the interleaving (tests/test_gpu_threads.py).  Freeing a handle another thread is still using
 *     is the caller's error, as with any Rust `Drop`.
 *   - Scalars passed as canonical integers (mont = 0) must be < r, as ark-ff's BigInt images of Fr
 *     always are; the library rejects a call holding a scalar >= 2^255 with GM_EINVAL and does not
 *     otherwise check (the reference panics on the out-of-range bucket index such a value produces).
 */
#ifndef GEMINI_HIP_H
#define GEMINI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK 0
#define GM_EINVAL (-1)    /* bad length / null pointer / bad stride */
#define GM_ENOTINIT (-2)  /* gm_init not called */
#define GM_EHANDLE (-3)   /* unknown or freed handle */
#define GM_EHIP (-4)      /* HIP runtime error, see gm_last_error() */
#define GM_ENOMEM (-5)
#define GM_ESTATE (-6)    /* call sequence error (e.g. round message after the last round) */

/* ---- lifecycle -------------------------------------------------------------------------- */
/* Bind this process to GPU `device` (hipSetDevice) and create the library's streams. */
int gm_init(int device);
void gm_shutdown(void);
const char* gm_last_error(void);
/* ABI version, bumped on any signature change. */
int gm_abi_version(void);

/* ---- G1 multi-scalar multiplication ------------------------------------------------------- */
/* Replaces VariableBaseMSM::msm_bigint(bases, bigints) of ark-ec 0.4.2 -- the function every
 * MSM call site of the reference bottoms out in (src/kzg/time.rs:82,129; src/kzg/space.rs:52;
 * ChunkedPippenger/HashMapPippenger flushes, in-tree copy src/kzg/msm/stream_pippenger.rs:180,250,264;
 * algorithm statement src/kzg/msm/variable_base.rs:99-176).
 * scalars: n x 4 u64 canonical (< r).  Like msm_unchecked, the caller passes n = min(lengths). */
int gm_g1_msm(const void* bases, size_t base_stride, const uint64_t* scalars, size_t n, uint64_t out_jac[18]);

/* SRS resident in HBM: replaces the `powers_of_g: Vec<G1Affine>` field of CommitterKey
 * (src/kzg/time.rs:24-27) as the MSM operand.  The bases are copied; `handle` is opaque.  Cost: one upload and one
 * repacking pass; NO fixed-base tables are built here (a registration may serve a single MSM) -- a key that stays
 * resident asks for them with gm_g1_bases_precompute(handle, -1). */
int gm_g1_bases_register(const void* bases, size_t base_stride, size_t n, uint64_t* handle);
int gm_g1_bases_free(uint64_t handle);
int gm_g1_bases_len(uint64_t handle, size_t* n);
/* Copy registered bases back (96-byte stride): used by tests and by CommitterKey::index_by. */
int gm_g1_bases_download(uint64_t handle, size_t offset, size_t n, void* out96);

/* Fixed-base window tables for a registered SRS: table[w][i] = 2^(c*w) * base[i] for w < ceil(256/c),
 * kept in HBM (W x n x 96 bytes -- 40 GB for the 2^25-point SRS of `snark -i 24`; this is what 288 GB
 * per GPU buys).  Afterwards MSMs of >= 2^17 pairs against this handle use ONE bucket set shared by
 * all windows: c = 20 instead of 16 (13 instead of 16 base additions per pair), no per-window bucket
 * reduction and c instead of 256 doublings at the end.  Results are identical (tests compare both
 * paths).  Setup cost is comparable to generating the SRS; like CommitterKey::new it is outside the
 * prover timer.  c = 0 picks the default (20 below 2^23 points; 22 from there on, which serves calls from 2^22 pairs on, PLUS a
 * c = 20 table over the first 2^22 points -- 5.2 GB -- for the calls of 2^17 .. 2^22 - 1 pairs whose index range stays inside that
 * prefix: the low levels of a folding tree -- and, for every key, a c = 16 table over the first 2^17 points, 201 MB, for the
 * latency-bound calls of 2^11 .. 2^17 - 1 pairs: one bucket set, 16 final doublings); c = -1 is AUTOMATIC: the rule of gm_set_auto_tables below (2^17 .. 2^26 - 1
 * points, byte budget, free memory; a no-op returning GM_OK when the tables do not fit or exist already).
 * No reference counterpart: ark-ec recomputes. */
int gm_g1_bases_precompute(uint64_t handle, int c);
/* Tables BY DEFAULT (on = 1 at gm_init) for the KEY CONSTRUCTORS: gm_g1_fixed_base_register, gm_g1_srs_register and
 * gm_g1_srs_register_segments (the CommitterKey::new analogues) build the tables above for 2^17 .. 2^26 - 1 points (a key of
 * 2^26 .. 2^28 - 1 points gets tables over its first 2^25 / 2^22 / 2^17 points only: 44 GB, spare memory -- see
 * gm_g1_release_spare_tables) when
 * W x n x 96 bytes fit `max_bytes` (0 = 30 % of the device memory) and the free memory; otherwise, silently, the plain path
 * serves the key.  Cost at registration: ~240 doublings per point and W x n x 96 bytes (1.3 GB at 2^20 points).
 * gm_g1_bases_register (points uploaded from the host) never builds them on its own: gm_g1_bases_precompute(handle, -1).
 * A committer key is registered once and serves ~3 N pairs of MSMs per proof; the build is setup, like
 * CommitterKey::new (src/kzg/time.rs:49-72, which builds a window table of its own to generate the key). */
int gm_set_auto_tables(int on, size_t max_bytes);
/* Window width of the main tables (0 = no tables) and size in bytes of all tables of a handle (prefix table included). */
int gm_g1_bases_table_info(uint64_t handle, int* c, size_t* bytes);
/* Give the freed blocks of the device-vector pool (up to 55 % of the device memory) back to the driver -- for a host
 * process that shares the GPU with another allocator (torch, a second rank). */
int gm_pool_trim(void);
/* Device-memory bookkeeping of the library (every allocation it makes is counted): out[0] = device total, [1] = device free
 * (hipMemGetInfo), [2] = bytes the library holds from the driver, [3] = high-water mark of [2], [4] = of [2]: freed blocks cached by
 * the vector pool, [5] = in use = [2] - [4], [6] = high-water mark of [5], [7] = fixed-base tables of all keys, [8] = the keys
 * themselves, [9] = how many times the prefix tables were released under memory pressure (they are not rebuilt: the calls they
 * served take the plain path), [10] = what the MSM workspaces hold (grow-only), [11] = 0.  gm_mem_reset_peak() restarts both
 * high-water marks from the current values.  The reference's memory
 * story is its constants (README.md:38-46: SPACE_TIME_THRESHOLD, MAX_MSM_BUFFER_LOG); on the device it is these figures and the
 * footprint contract below (gm_snark_footprint / gm_psnark_footprint). */
int gm_mem_stats(uint64_t out[12]);
/* out[0] = compute units of the device, [1] = of those: set aside for the tails of a batch (the XCD partition, GM_CU_SPLIT; 0 = off),
 * [2] = zero-copy result paths in use (bit 0 field kernels, bit 1 MSM planes), [3] = small-call lanes of a batch */
int gm_runtime_info(int out[4]);
int gm_mem_reset_peak(void);
/* THE FOOTPRINT CONTRACT.  What a proof will allocate beyond what is resident when it is called (the key and its tables, the
 * instance, the caller's vectors): out[0] = high-water mark of its device vectors and prover buffers, out[1] = what the MSM
 * workspaces may still have to grow by for its largest calls (an upper bound; it shrinks as they grow), out[2] = out[0] + out[1], out[3] =
 * what can be had right now (device free + the vector pool's cached blocks + the prefix tables, which are spare memory).  Upper
 * bounds, within ~25 % of the measured peaks (tests/test_gpu_footprint.py, profiles/r5_footprint.txt).  Every prover compiled into
 * the library checks the same figures before its first allocation: if the proof only fits without the prefix tables they are
 * released THEN (not by reflex after an allocation has failed half-way), and if it does not fit at all the prover returns
 * GM_ENOMEM with the numbers in gm_last_error() (GM_FOOTPRINT_CHECK=0 switches the admission off).
 *   gm_snark_footprint   snark::Proof::new_time / new_elastic on num_constraints = |z|; elastic != 0: gm_snark_new_elastic
 *   gm_psnark_footprint  psnark::Proof::new_time (elastic 0), new_elastic in the resident schedule (1, min_device_chunk > 1) or
 *                        the literal one (2: space provers over reversed streams, min_device_chunk = 1)
 * ck_bases: the key the proof will commit under (its tables decide the window widths, a cyclic share divides the MSM sizes). */
int gm_snark_footprint(uint64_t ck_bases, size_t num_constraints, int elastic, uint64_t out[4]);
int gm_psnark_footprint(uint64_t ck_bases, size_t num_variables, size_t nnz, int elastic, uint64_t out[4]);
int gm_footprint_admit(int psnark, uint64_t ck_bases, size_t n, size_t nnz, int elastic);
/* Frees the PREFIX tables of every key (the c = 22 / 20 / 16 tables over the first 2^25 / 2^22 / 2^17 points; up to 44 GB for a key of
 * 2^26+ points): the calls they served take the plain path from then on, with the same results.  The library does this by itself
 * when a device allocation fails twice (after the vector pool has given its freed blocks back); never while an MSM is running. */
int gm_g1_release_spare_tables(void);

/* MSM against registered bases.  Pair i uses base[offset + i] (reversed = 0) or base[offset - i]
 * (reversed = 1).  reversed/offset express CommitterKey::commit's prefix slice
 * (src/kzg/time.rs:81-83) and CommitterKeyStream's Reverse(..) + advance_by alignment
 * (src/kzg/space.rs:36-40,108-113,291-296) without moving the SRS. */
int gm_g1_msm_h(uint64_t handle, size_t offset, int reversed, const uint64_t* scalars, size_t n, uint64_t out_jac[18]);

/* Same, scalars taken from a device-resident Fr vector (see gm_fr_vec_*), elements
 * [voffset, voffset + n).  The vector is in Montgomery form; into_bigint happens on device. */
int gm_g1_msm_v(uint64_t bases_handle, size_t offset, int reversed, uint64_t vec_handle, size_t voffset, size_t n,
                uint64_t out_jac[18]);

/* k MSMs against the same bases (CommitterKey::batch_commit, src/kzg/time.rs:98-107): MSM j pairs
 * elements [0, ns[j]) of vector j with bases[offset ...].  Same results as k gm_g1_msm_v calls; the
 * host tail of call j overlaps the kernels of call j+1.  out_jac: k x 18 limbs. */
int gm_g1_msm_v_batch(uint64_t bases_handle, size_t offset, int reversed, const uint64_t* vec_handles, const size_t* ns, size_t k,
                      uint64_t* out_jac);
/* The same with UN-NORMALISED results (any Jacobian representative, no field inversion): the per-rank partials of a
 * sharded batch_commit -- all-gathered k x 144 bytes at a time, summed and normalised once per commitment (gm_g1_sum). */
int gm_g1_msm_v_batch_partial(uint64_t bases_handle, size_t offset, int reversed, const uint64_t* vec_handles, const size_t* ns, size_t k,
                              uint64_t* out_jac);

/* The same with a base offset PER CALL: call j pairs vector j with bases[offsets[j] ...] (reversed: walking down from it).
 * One key that holds several ranges of powers back to back -- the per-level slices of a block-sharded key
 * (gm_g1_srs_register_segments) -- serves every level through one pipelined batch.  partial != 0: un-normalised results. */
int gm_g1_msm_v_batch_at(uint64_t bases_handle, const size_t* offsets, int reversed, const uint64_t* vec_handles, const size_t* ns, size_t k,
                         int partial, uint64_t* out_jac);

/* Same, raw device pointer to n x 32-byte scalars already in HBM (mont != 0: Montgomery form).
 * This is the entry bench.py times: inputs resident, result = 144 bytes. */
int gm_g1_msm_d(uint64_t bases_handle, size_t offset, int reversed, const void* d_scalars, int mont, size_t n,
                uint64_t out_jac[18]);

/* Partial-result form for sharded MSMs: identical to gm_g1_msm_d but the
 * result is NOT normalised, so ranks can all-gather 144-byte partial points and combine them
 * with gm_g1_sum. */
int gm_g1_msm_d_partial(uint64_t bases_handle, size_t offset, int reversed, const void* d_scalars, int mont, size_t n,
                        uint64_t out_jac[18]);
/* out = normalise(sum of k Jacobian points): the local EC-add after the all-gather, and
 * ChunkedPippenger's `result += chunk` (src/kzg/msm/stream_pippenger.rs:248-256). */
int gm_g1_sum(const uint64_t* points_jac, size_t k, uint64_t out_jac[18]);

/* ---- streaming MSM over HOST-resident pairs: bounded device memory ---------------------------------
 * ChunkedPippenger (src/kzg/msm/stream_pippenger.rs:209-272: with_size / add / finalize) and msm_chunks
 * (src/kzg/space.rs:22-55) collect a buffer of pairs from the streams, run one MSM, add it to the running sum
 * and start over.  Here the buffer is two device slots of `chunk_pairs` pairs: the stream (a key and a
 * polynomial that may be larger than HBM) stays in host memory, blocks of any size are pushed with _add, the
 * copy of chunk i + 1 runs under the kernels of chunk i, and _finalize returns the normalised sum -- equal to
 * the one-call MSM of the whole stream wherever it is cut.  chunk_pairs <= 2^26.  Overlap happens inside one
 * _add call (nothing of a stream stays in flight on the shared MSM lanes when a call returns): push blocks of
 * several chunks.  Host buffers may be pageable or page-locked (gm_host_alloc: copied by DMA, several times
 * faster) and are free to be reused when _add returns.  A stream is used by one thread at a time; distinct
 * streams from distinct threads are safe (their MSMs queue behind each other).
 *   _new:   every pair carries its base: records of base_stride bytes as in gm_g1_bases_register
 *           (x, y Montgomery; optional infinity flag at byte 96), scalars 32-byte integers < r
 *           (scalars_mont != 0: ark-ff Montgomery form instead).
 *   _new_h: scalars only, against registered bases starting at `offset` and walking up (reversed != 0: down,
 *           the big-endian stream view of src/kzg/space.rs:287-296); bases_host of _add is ignored.
 *   _finalize also resets the stream (the reference's finalize consumes the object); pairs_or_null receives
 *           the number of pairs summed.  A failed _add (scalar >= 2^255, bases exhausted, ...) resets it as well. */
int gm_g1_msm_stream_new(size_t chunk_pairs, size_t base_stride, int scalars_mont, uint64_t* stream);
int gm_g1_msm_stream_new_h(uint64_t bases_handle, size_t offset, int reversed, size_t chunk_pairs, int scalars_mont, uint64_t* stream);
int gm_g1_msm_stream_add(uint64_t stream, const void* bases_host, const void* scalars_host, size_t n);
int gm_g1_msm_stream_finalize(uint64_t stream, uint64_t out_jac[18], size_t* pairs_or_null);
int gm_g1_msm_stream_free(uint64_t stream);
/* page-locked host memory for the streams above (hipHostMalloc / hipHostFree) */
int gm_host_alloc(size_t bytes, void** p);
int gm_host_free(void* p);

/* Fixed-base generation on device: out[i] = scalars[i] * base (affine, 96-byte stride), registered
 * directly as a bases handle.  Replaces FixedBase::msm + normalize_batch in CommitterKey::new
 * (src/kzg/time.rs:49-59; setup, outside the prover timer) and builds benchmark inputs.
 * scalars: n x 4 u64 canonical on the host. */
int gm_g1_fixed_base_register(const uint64_t base_affine[12], const uint64_t* scalars, size_t n, uint64_t* handle);
/* powers_of_g[i] = tau^i * g for i < n, entirely on device (tau canonical). */
int gm_g1_srs_register(const uint64_t base_affine[12], const uint64_t tau[4], size_t n, uint64_t* handle);
/* Several ranges of the same powers back to back in ONE handle: segment s = tau^(starts[s] + i) * g, i < counts[s]
 * (exponents below 2^39).  What a rank of a block-sharded prover holds of CommitterKey::powers_of_g (src/kzg/time.rs:24-27):
 * its block of every folding level, addressed by gm_g1_msm_v_batch_at. */
int gm_g1_srs_register_segments(const uint64_t base_affine[12], const uint64_t tau[4], const size_t* starts, const size_t* counts, size_t nseg,
                                uint64_t* handle);

/* Tuning knob (0 = automatic): window width c of the bucket method.  The result does not depend
 * on it (tests sweep it). */
int gm_set_msm_window(int c);
/* Smallest pair count for which an MSM uses the fixed-base tables of its handle (default 2^17:
 * below that the MSM is latency-bound and few buckets win).  Tuning/test knob. */
int gm_set_msm_table_min(size_t n);
/* Tuning knob (default 0), read when bases are REGISTERED: also store phi(P_i) = (beta x_i, y_i) = lambda P_i and run
 * MSMs on those bases with every scalar split as s = v1 + v2 lambda, |v1|, |v2| < 2^127 (GLV): half the windows, twice
 * the base memory.  Same results; no net gain on MI355X as measured.  A round-3 EXPERIMENT: the code
 * (gemini_amd/csrc/msm_glv.inc) is only in builds with -DGM_EXPERIMENTS; otherwise on != 0 returns GM_EINVAL. */
int gm_set_msm_glv(int on);
/* Tuning knob (default 0): run one-call MSMs of >= 2^17 pairs as two window groups pipelined over three streams.
 * Same results; slower on MI355X as measured.  Builds with -DGM_EXPERIMENTS only; otherwise on != 0 returns GM_EINVAL. */
int gm_set_msm_split(int on);
/* Affine tree levels in front of the XYZZ bucket accumulation (0 = none, -1 = automatic, <= 8): every
 * level adds the sorted entries of each bucket pairwise in affine coordinates with one shared field
 * inversion (6 instead of 10 field products per addition).  The result does not depend on it.
 * A round-2 EXPERIMENT, slower at every size: the kernels are only in builds with -DGM_EXPERIMENTS
 * (gemini_amd/csrc/msm_levels.inc); otherwise -1 and 0 both mean "none" and any levels > 0 returns GM_EINVAL. */
int gm_set_msm_affine_levels(int levels);

/* Per-stage device timing (HIP events on the library's stream).  Stages, in order:
 * 0 digits+histogram, 1 scan, 2 scatter, 3 bucket accumulate (k_acc0), 4 partial merge,
 * 5 bucket reduce, 6 sumcheck round.  gm_prof_enable(1) resets and starts accumulating; gm_prof_enable(2): stage 3 only (every
 * event record between two kernels of a call is a ~10 us bubble on the stream: five stages cost an MSM ~60 us, one ~20 us);
 * gm_prof_read returns total milliseconds and launch-group counts per stage.  No reference
 * counterpart (the reference only has start_timer!/end_timer! spans, src/snark/time_prover.rs:23). */
#define GM_PROF_NSTAGES 7
int gm_prof_enable(int on);
int gm_prof_read(double* ms_out, uint64_t* count_out, int n);
/* The shader clock (MHz) the bucket accumulation ran at while profiling was on: clock64() against the constant 100 MHz
 * wall_clock64() inside the kernel, summed over the calls since gm_prof_enable(1).  The issue bound of the accumulation is
 * proportional to it, and it is NOT the nominal clock (1.9 .. 2.3 GHz under this load). 0 if none ran. */
int gm_prof_read_clock(double* acc0_mhz);

/* ---- device-resident Fr vectors ------------------------------------------------------------ */
/* Stand in for the `Vec<F>` values the time prover keeps in RAM (src/snark/time_prover.rs:32-106). */
int gm_fr_vec_alloc(size_t n, uint64_t* handle);
int gm_fr_vec_free(uint64_t handle);
int gm_fr_vec_len(uint64_t handle, size_t* n);
int gm_fr_vec_upload(uint64_t handle, size_t offset, const uint64_t* mont, size_t n);
int gm_fr_vec_download(uint64_t handle, size_t offset, uint64_t* mont, size_t n);
int gm_fr_vec_fill(uint64_t handle, const uint64_t value_mont[4]);
/* raw device pointer (for torch.from_blob-style interop and RCCL); valid until free */
int gm_fr_vec_ptr(uint64_t handle, void** dptr);
/* shrink the logical length (DensePolynomial trims trailing zeros; fold halves lengths) */
int gm_fr_vec_set_len(uint64_t handle, size_t n);

/* ---- field vector helpers (src/misc.rs) ------------------------------------------------------ */
/* out[i] = in[n-1-i]: big-endian stream <-> little-endian vector        src/iterable/slice.rs:17-39 */
int gm_fr_reverse(uint64_t in, uint64_t out);
/* out[k] = in[start + k * stride], k < count: the scalars a rank multiplies against its share of a key sharded
 * element-cyclically over the GPUs (power i lives on rank i mod g; gemini_amd/dist.py), herring's even / odd halves. */
int gm_fr_stride(uint64_t in, size_t start, size_t stride, size_t count, uint64_t out);
/* out = [f[2i] + r * f[2i+1]], len ceil(n/2)                       src/misc.rs:52-56 */
int gm_fr_fold(uint64_t f, const uint64_t r_mont[4], uint64_t out);
/* foldings_polynomial (src/subprotocols/tensorcheck/mod.rs:124-133): outs[j] = fold(outs[j - 1], challenge j) with outs[-1] = f;
 * k launches behind one another, ONE wait.  outs[j] needs capacity ceil(len / 2^(j + 1)). */
int gm_fr_fold_chain(uint64_t f, const uint64_t* challenges_mont, size_t k, const uint64_t* outs);
/* out = [1, x, x^2, ...]                                          src/misc.rs:59-65 */
int gm_fr_powers(const uint64_t x_mont[4], size_t n, uint64_t out);
/* out[sum b_j 2^j] = prod rho_j^{b_j}, len 2^k                    src/misc.rs:133-149 */
int gm_fr_tensor(const uint64_t* rhos_mont, size_t k, uint64_t out);
/* out = a . b elementwise (equal lengths, else GM_EINVAL)          src/misc.rs:205-208 */
int gm_fr_hadamard(uint64_t a, uint64_t b, uint64_t out);
/* result = sum_i a_i b_i                                           src/misc.rs:215-218 */
int gm_fr_ip(uint64_t a, uint64_t b, uint64_t result_mont[4]);
/* results[j] = sum_i p_i x_j^i for up to 3 points in one pass      src/misc.rs:194-199
 * (tensorcheck evaluates every polynomial at beta^2, beta, -beta: tensorcheck/mod.rs:228-247) */
int gm_fr_eval_le(uint64_t poly, const uint64_t* xs_mont, size_t npoints, uint64_t* results_mont);
/* The same for k polynomials at the same points, one wait for all (results: k x npoints x 4 limbs) -- the foldings of a
 * tensor check at +-beta                                            tensorcheck/mod.rs:236-247 */
int gm_fr_eval_le_batch(const uint64_t* polys, size_t k, const uint64_t* xs_mont, size_t npoints, uint64_t* results_mont);
/* out = sum_j c_j p_j padded to the longest; logical length trimmed of high zeros
 *                                                                  src/misc.rs:37-48 */
int gm_fr_lincomb(const uint64_t* polys, const uint64_t* coeffs_mont, size_t k, uint64_t out);
/* out[out_offset + i] = c * in[i] for i < len(in), inside out's current length: scaled vectors laid out side by side in one
 * vector (the per-level quotients of the block-sharded opening, committed by ONE MSM against the back-to-back key slices) */
int gm_fr_scale_into(uint64_t in, const uint64_t c_mont[4], uint64_t out, size_t out_offset);
/* the same for k vectors into disjoint ranges of `out`: k launches, one wait */
int gm_fr_scale_into_many(const uint64_t* ins, const uint64_t* coeffs_mont, size_t k, uint64_t out, const size_t* out_offsets);
/* v[positions[j]] += values[j] for k <= 4096 DISTINCT positions inside the vector (seam corrections of the laid-out opening) */
int gm_fr_add_at(uint64_t v, const size_t* positions, const uint64_t* values_mont, size_t k);
/* quotient of f by the monic vanishing polynomial of `points` (degree k <= 3); rem gets k values.
 * Replaces DensePolynomial::div in open_multi_points           src/kzg/time.rs:134-145 */
int gm_fr_div_vanishing(uint64_t f, const uint64_t* points_mont, size_t k, uint64_t quotient, uint64_t* rem_mont);

/* ---- index vectors and the entry-product / plookup vector builders (preprocessing SNARK) ---------- */
/* `&[usize]` arguments (row_index, col_index, extended frequencies) copied to HBM once. */
int gm_idx_register(const uint32_t* index, size_t n, uint64_t* handle);
int gm_idx_free(uint64_t handle);
/* out[j] = src[index[j]]: `lookup` and `sorted`        src/subprotocols/plookup/time_prover.rs:5-8,67-74 */
int gm_fr_gather(uint64_t src, uint64_t index, uint64_t out);
/* out[i] = v[i] + F::from(index[i]) * zeta; index = 0 means the range 0..len   plookup/time_prover.rs:11-21 */
int gm_fr_alg_hash(uint64_t v, uint64_t index, const uint64_t zeta_mont[4], uint64_t out);
/* plookup_set: len + 1 entries (1+z)y + v[i-1] + z v[i]                         plookup/time_prover.rs:23-35 */
int gm_fr_plookup_set(uint64_t v, const uint64_t y_mont[4], const uint64_t z_mont[4], uint64_t out);
/* plookup_subset: out[i] = v[i] + y                                             plookup/time_prover.rs:62-64 */
int gm_fr_add_scalar(uint64_t v, const uint64_t y_mont[4], uint64_t out);
/* right_rotation(monic(v)) = [1, v...]                         src/subprotocols/entryproduct/time_prover.rs:14-23,47-51 */
int gm_fr_shift_monic(uint64_t v, uint64_t out);
/* accumulated_product(monic(v)): out[i] = prod_{j>=i} v[j], out[len] = 1 (reverse prefix-product scan)
 *                                                              src/subprotocols/entryproduct/time_prover.rs:25-45 */
int gm_fr_acc_product(uint64_t v, uint64_t out);
/* The same builders on ONE BLOCK of a block-sharded vector (gm_psnark_new_time_sharded; a block = the elements [lo, lo + M) that exist):
 *   gm_fr_tensor_range / gm_fr_powers_range   out[i] = tensor(rhos)[start + i] / x^(start + i), i < count
 *   gm_fr_plookup_set_block   plookup_set (plookup/time_prover.rs:23-35) for the outputs [lo, lo + out_count): v[v_offset .. + v_count) = the
 *                             elements lo .. of the hashed set that exist, prev = element lo - 1 (null at lo = 0)
 *   gm_fr_shift_block         right_rotation(monic(v)) (entryproduct/time_prover.rs:14-51): out[0] = first (1 on the lowest block, else the
 *                             last element of the block below), out[t] = v[t - 1]
 *   gm_fr_product             the product of the elements of v (the carry a higher block hands down)
 *   gm_fr_acc_product_block   accumulated_product(monic(v)) of a block: the suffix scan started from `carry` (the product of the blocks
 *                             above; null: one); write_monic: position |whole v| -- the entry 1 -- falls into this block */
int gm_fr_tensor_range(const uint64_t* rhos_mont, size_t k, size_t start, size_t count, uint64_t out);
/* `lookup(v, index)` (plookup/time_prover.rs:5-8) of a vector that is a FUNCTION of the index, without the vector: out[i] = tensor(rhos)[index[i]]
 * resp. x^index[i] (index[i] < 2^k / 2^log_len) -- one multiplication per element from two half tables that stay in L2.  gm_fr_alg_hash_from: the
 * algebraic hash of a RANGE of a vector, out[i] = v[i] + (first_index + i) zeta. */
int gm_fr_tensor_gather(const uint64_t* rhos_mont, size_t k, uint64_t index, uint64_t out);
int gm_fr_powers_gather(const uint64_t x_mont[4], size_t log_len, uint64_t index, uint64_t out);
int gm_fr_alg_hash_from(uint64_t v, size_t first_index, const uint64_t zeta_mont[4], uint64_t out);
int gm_fr_powers_range(const uint64_t x_mont[4], size_t start, size_t count, uint64_t out);
int gm_fr_plookup_set_block(uint64_t v, size_t v_offset, size_t v_count, const uint64_t* prev_or_null, size_t out_count, const uint64_t y_mont[4],
                            const uint64_t z_mont[4], uint64_t out);
int gm_fr_shift_block(uint64_t v, const uint64_t first_mont[4], size_t out_count, uint64_t out);
int gm_fr_product(uint64_t v, uint64_t total_mont[4]);
int gm_fr_acc_product_block(uint64_t v, const uint64_t* carry_or_null, int write_monic, uint64_t out);

/* ---- sparse R1CS matrices ------------------------------------------------------------------------ */
/* `Matrix<F> = Vec<Vec<(F, usize)>>` (src/circuit.rs:43) as CSR, copied to HBM once.  gm_spm_mul is
 * product_matrix_vector (src/misc.rs:100-110): y = M x.  Registering the TRANSPOSE turns the
 * abc_tensored scatter-add of src/snark/time_prover.rs:63-81 into three products M^T r. */
int gm_spm_register(const uint64_t* rowptr, const uint32_t* cols, const uint64_t* vals_mont, size_t nrows, size_t ncols,
                    size_t nnz, uint64_t* handle);
int gm_spm_free(uint64_t handle);
int gm_spm_mul(uint64_t matrix, uint64_t x, uint64_t y);
int gm_spm_shape(uint64_t matrix, size_t* nrows_or_null, size_t* ncols_or_null, size_t* nnz_or_null);
/* The CSR arrays back on the host (rowptr: nrows + 1, cols: nnz, vals: nnz x 4 Montgomery limbs; any pointer may be NULL). */
int gm_spm_download(uint64_t matrix, uint64_t* rowptr, uint32_t* cols, uint64_t* vals_mont);

/* Split-phase round for callers that drive MANY provers in lock-step (Sumcheck::prove_batch maps its provers over
 * rayon, src/subprotocols/sumcheck/proof.rs:85): _begin does what gm_sc_round does up to the launch of the kernel
 * and the asynchronous copy of its partial sums, _end waits for the stream and returns the message.  Begin the
 * round of every prover, then end them: one wait instead of one per prover.  has_msg = 0 means the prover is out
 * of rounds (no _end call then); a second _begin before the _end of the same prover is GM_ESTATE. */
int gm_sc_round_begin(uint64_t handle, const uint64_t* challenge_or_null, int* has_msg);
int gm_sc_round_end(uint64_t handle, uint64_t a_mont[4], uint64_t b_mont[4]);
/* gm_sc_round_begin for k provers with the SAME challenge (the provers of Sumcheck::prove_batch, proof.rs:85): those on the device share
 * ONE kernel launch per (fold, message) combination instead of one each; has_msg: k flags.  Collect each with gm_sc_round_end. */
int gm_sc_round_begin_many(const uint64_t* handles, size_t k, const uint64_t* challenge_or_null, int* has_msg);

/* ---- sumcheck time prover --------------------------------------------------------------------- */
/* Replaces TimeProver<F> behind `trait Prover<F>` (src/subprotocols/sumcheck/prover.rs:30-45,
 * src/subprotocols/sumcheck/time_prover.rs:42-137).  f, g are copied (Witness::new copies too,
 * time_prover.rs:26-32). */
int gm_sc_new(const uint64_t* f_mont, size_t nf, const uint64_t* g_mont, size_t ng, const uint64_t twist_mont[4],
              uint64_t* handle);
/* from device vectors (copied), for a prover that keeps its state in HBM */
int gm_sc_new_v(uint64_t f_vec, uint64_t g_vec, const uint64_t twist_mont[4], uint64_t* handle);
/* the same WITHOUT the copy: the prover reads f_vec and g_vec in place until its first fold has written their halves into its own
 * buffers -- the caller keeps both vectors alive and unmodified until then (the provers compiled into the library do: the vectors
 * are theirs).  2 x n x 32 bytes less to move and to hold per prover. */
int gm_sc_new_borrow(uint64_t f_vec, uint64_t g_vec, const uint64_t twist_mont[4], uint64_t* handle);
/* next_message(verifier_message): challenge_or_null == NULL is `None`.  *has_msg = 0 means the
 * reference returned None (round == tot_rounds).                     time_prover.rs:83-123 */
int gm_sc_round(uint64_t handle, const uint64_t* challenge_or_null, uint64_t a_mont[4], uint64_t b_mont[4], int* has_msg);
/* Prover::fold                                                        time_prover.rs:75-80 */
int gm_sc_fold(uint64_t handle, const uint64_t challenge_mont[4]);
/* rounds() / round()                                                  time_prover.rs:125-131 */
int gm_sc_rounds(uint64_t handle, size_t* tot_rounds, size_t* round);
/* final_foldings(): *has = 0 if round != tot_rounds                   time_prover.rs:133-137 */
int gm_sc_final(uint64_t handle, uint64_t f0_mont[4], uint64_t g0_mont[4], int* has);
int gm_sc_free(uint64_t handle);
/* partial-message form for sharded sumchecks: the shard holds pairs
 * [pair_offset, pair_offset + len/2) of the global vectors; twist powers start at tau^(2*pair_offset) */
int gm_sc_set_shard(uint64_t handle, uint64_t pair_offset);
/* the same for a block whose length says nothing about the rounds of the WHOLE vectors (a partial block of a block-sharded
 * prover): the prover keeps producing messages for tot_rounds rounds */
int gm_sc_set_shard_rounds(uint64_t handle, uint64_t pair_offset, size_t tot_rounds);
/* current state of the prover (TimeProver's pub fields f, g, twist: time_prover.rs:42-52): lengths and
 * twist, then the vectors themselves -- used when sharded provers hand their tails to one rank */
int gm_sc_lens(uint64_t handle, size_t* nf, size_t* ng, uint64_t twist_mont[4]);
int gm_sc_download(uint64_t handle, uint64_t* f_mont, uint64_t* g_mont);

/* ---- sumcheck space prover / elastic hand-off --------------------------------------------------- */
/* Replaces SpaceProver<F, S1, S2> behind `trait Prover<F>` (src/subprotocols/sumcheck/space_prover.rs).
 * f_stream / g_stream are the BIG-ENDIAN coefficient streams the reference consumes (Reverse(slice),
 * src/iterable/slice.rs:17-39).  Only the streams and the challenges are kept; every message is
 * recomputed from the streams (:117-240).  rounds = ceil(log2(min(len))) as in :74-77.
 * gm_sp_to_time is `TimeProver::from(&SpaceProver)` (:269-307), the switch ElasticProver::fold makes
 * when rounds - round < SPACE_TIME_THRESHOLD = 22 (elastic_prover.rs:44-57, src/lib.rs:76). */
int gm_sp_new(const uint64_t* f_stream_mont, size_t nf, const uint64_t* g_stream_mont, size_t ng, const uint64_t twist_mont[4],
              uint64_t* handle);
/* same, the streams taken from device-resident vectors (copied, like gm_sc_new_v) */
int gm_sp_new_v(uint64_t f_stream_vec, uint64_t g_stream_vec, const uint64_t twist_mont[4], uint64_t* handle);
/* without the copy: a space prover never writes its streams, so it reads the caller's vectors for its whole life -- they
 * stay alive and unmodified until gm_sp_free (gm_sp_to_time materialises the time prover in vectors of its own). */
int gm_sp_new_borrow(uint64_t f_stream_vec, uint64_t g_stream_vec, const uint64_t twist_mont[4], uint64_t* handle);
int gm_sp_round(uint64_t handle, const uint64_t* challenge_or_null, uint64_t a_mont[4], uint64_t b_mont[4], int* has_msg);
int gm_sp_fold(uint64_t handle, const uint64_t challenge_mont[4]);
int gm_sp_rounds(uint64_t handle, size_t* tot_rounds, size_t* round);
int gm_sp_final(uint64_t handle, uint64_t f0_mont[4], uint64_t g0_mont[4], int* has);
int gm_sp_to_time(uint64_t handle, uint64_t* time_handle);
int gm_sp_free(uint64_t handle);

/* ---- herring: sumcheck over a bilinear module (src/herring) --------------------------------------- */
/* FModule (F x F -> F, module.rs:127-146): the gm_sc_* prover with herring semantics -- the twist is
 * used when folding only, the message is a = <f_e, g_e>, b = <f_e, g_o> + <f_o, g_e>
 * (src/herring/time_prover.rs:91-123) and rounds = ceil(log2(min(len))) (:36-39).  Call right after
 * gm_sc_new. */
int gm_sc_set_herring(uint64_t handle, int on);
/* G1Module (G1 x F -> G1, module.rs:81-102): f is a vector of G1 points (affine records as in
 * gm_g1_bases_register), g a vector of Fr.  Each message is three device MSMs over the even/odd
 * halves (`M::ip` = msm_unchecked); fold is split_fold (time_prover.rs:72-76): f'[i] = f[2i] +
 * (r*twist) f[2i+1] on the device, g folds in Fr.  Messages / final f are normalised Jacobian. */
int gm_hg1_new(const void* f_bases, size_t base_stride, size_t nf, const uint64_t* g_mont, size_t ng, const uint64_t twist_mont[4],
               uint64_t* handle);
int gm_hg1_round(uint64_t handle, const uint64_t* challenge_or_null, uint64_t a_jac[18], uint64_t b_jac[18], int* has_msg);
int gm_hg1_fold(uint64_t handle, const uint64_t challenge_mont[4]);
int gm_hg1_rounds(uint64_t handle, size_t* tot_rounds, size_t* round);
int gm_hg1_final(uint64_t handle, uint64_t f0_jac[18], uint64_t g0_mont[4], int* has);
int gm_hg1_free(uint64_t handle);

/* ---- Fiat-Shamir transcript (host; no GPU needed) ------------------------------------------------ */
/* merlin::Transcript::new(label) (merlin 3.0.0, Cargo.lock:606-608); the prover uses
 * Transcript::new(PROTOCOL_NAME) with PROTOCOL_NAME = b"GEMINI-v0" (src/lib.rs:74). */
int gm_transcript_new(const uint8_t* label, size_t len, uint64_t* handle);
int gm_transcript_free(uint64_t handle);
/* Transcript::append_message / challenge_bytes */
int gm_transcript_append_message(uint64_t handle, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen);
int gm_transcript_challenge_bytes(uint64_t handle, const uint8_t* label, size_t llen, uint8_t* out, size_t n);
/* GeminiTranscript::append_serializable (src/transcript.rs:16-24) for `count` consecutive Fr
 * (an Fr, a RoundMsg(a, b), an [F; 2]) and for G1 elements (Commitment; with_len != 0 prefixes the
 * u64 length like Vec<Commitment>). */
int gm_transcript_append_fr(uint64_t handle, const uint8_t* label, size_t llen, const uint64_t* mont, size_t count);
int gm_transcript_append_g1(uint64_t handle, const uint8_t* label, size_t llen, const uint64_t* jac, size_t count, int with_len);
/* The ark-serialize framing gm_transcript_append_g1 uses, fixed by the curve crate of the build: 0 (default) =
 * ark-ec's short-Weierstrass default as in ark-test-curves' bls12_381 -- what the reference's examples and tests
 * link (x || y little-endian, flags in the top bits of the last byte); 1 = the zcash framing that ark-bls12-381
 * substitutes (big-endian, flags in the top bits of the first byte) -- what the reference's benches link.  A shim
 * may instead pass pre-serialised bytes through gm_transcript_append_message. */
int gm_transcript_set_g1_encoding(uint64_t handle, int encoding);
/* GeminiTranscript::get_challenge::<Fr> (src/transcript.rs:26-34) */
int gm_transcript_challenge_fr(uint64_t handle, const uint8_t* label, size_t llen, uint64_t out_mont[4]);
/* Sumcheck::prove round loop (src/subprotocols/sumcheck/proof.rs:36-66) over a gm_sc_* prover.
 * messages: cap_rounds x 8 u64 (a || b), challenges: cap_rounds x 4, final_foldings: f0 || g0. */
int gm_sumcheck_prove(uint64_t transcript, uint64_t prover, uint64_t* messages, uint64_t* challenges, size_t cap_rounds,
                      uint64_t final_foldings[8], size_t* rounds_out);

/* Sumcheck::prove_batch (src/subprotocols/sumcheck/proof.rs:69-122) over k gm_sc_* provers.
 * messages: cap_rounds x 8, challenges: cap_rounds x 4, final_foldings: k x 8 (lhs || rhs). */
int gm_sumcheck_prove_batch(uint64_t transcript, const uint64_t* provers, size_t k, uint64_t* messages, uint64_t* challenges,
                            size_t cap_rounds, uint64_t* final_foldings, size_t* rounds_out);

/* ---- the whole prover in one call ---------------------------------------------------------------------
 * snark::Proof::new_time (src/snark/time_prover.rs:19-117) with its tensor check (tensorcheck/mod.rs:190-275) and KZG
 * openings (src/kzg/time.rs:81-159): the orchestration the reference compiles into the prover, compiled into the
 * library over the entry points above (gemini_amd/csrc/snark.cpp) -- one FFI call per proof for a shim that replaces
 * `Proof::new_time(&r1cs, &ck)`.  matrices = {A, B, C, A^T, B^T, C^T} (gm_spm_register handles; the transposes serve
 * the column sums of :63-81), z and w vector handles, ck_bases the registered powers_of_g, g1_encoding as in
 * gm_transcript_set_g1_encoding.  The caller provides messages[k] (cap_rounds x 8), fold_commitments (cap_rounds x 18)
 * and fold_evaluations (cap_rounds x 8); cap_rounds >= ceil(log2 |z|) + 1.  Everything is in the memory images of the
 * rest of this header (Montgomery Fr limbs, 18-limb Jacobian points).  spans: seconds of the reference's
 * start_timer! spans -- matrix products, commitment to w, first sumcheck, tensor/powers/hadamard/abc_tensored, second
 * sumcheck, tensor check, the whole prover.  Same bytes as the step-by-step drivers (tests/test_gpu_snark.py). */
typedef struct gm_snark_proof {
  uint64_t witness_commitment[18];
  uint64_t zc_alpha[4];
  size_t rounds[2];            /* messages of the first / second sumcheck */
  uint64_t* messages[2];       /* RoundMsg(a, b) = 8 limbs per round */
  uint64_t final_foldings[2][8];
  size_t nfold;                /* folded polynomials of the tensor check = rounds[1] - 1 */
  uint64_t* fold_commitments;
  uint64_t* fold_evaluations;  /* at beta and -beta */
  uint64_t evaluation_proof[18];
  uint64_t base_evaluations[12]; /* w at beta^2, beta, -beta */
  double spans[7];
} gm_snark_proof;
int gm_snark_new_time(const uint64_t matrices[6], uint64_t z, uint64_t w, uint64_t ck_bases, int g1_encoding, size_t cap_rounds,
                      gm_snark_proof* proof);

/* snark::Proof::new_elastic(r1cs_stream, ck_stream, max_msm_buffer) (src/snark/elastic_prover.rs:174-266) as one call.
 * matrices_t: gm_spm handles of A^T, B^T, C^T (the MatrixTensor streams, :233-238); z_stream, w_stream, za/zb/zc_stream:
 * the BIG-ENDIAN streams of R1csStream (`Reverse(..)`, src/snark/tests.rs:38-52) as device vectors; ck_bases: the key in time
 * order (its stream view Reverse(powers_of_g) is the reversed addressing of the MSMs), at least as long as every stream.
 * min_device_chunk = 1: the LITERAL elastic prover -- sumchecks on the space prover until fewer than SPACE_TIME_THRESHOLD = 22
 * rounds remain, flushes of max_msm_buffer pairs.  min_device_chunk > 1 (the mirror's default 2^26): everything is resident and
 * max_msm_buffer advisory -- every flush of a streaming MSM shorter than `min_device_chunk` pairs is merged, and the sumchecks
 * take the RESIDENT schedule: time provers on the little-endian vectors from the first round (the same field elements,
 * sumcheck/tests.rs:42-87; less memory than reversed stream copies; the elastic prover then costs what the time prover costs).
 * Same proof bytes as gm_snark_new_time on the same instance and key (src/snark/tests.rs:56).  spans[0] is 0: the matrix
 * products belong to the construction of the streams. */
int gm_snark_new_elastic(const uint64_t matrices_t[3], uint64_t z_stream, uint64_t w_stream, uint64_t za_stream, uint64_t zb_stream,
                         uint64_t zc_stream, uint64_t ck_bases, size_t max_msm_buffer, size_t min_device_chunk, int g1_encoding,
                         size_t cap_rounds, gm_snark_proof* proof);

/* psnark::Proof::new_time(&ck, &r1cs, &index) (src/psnark/time_prover.rs:69-384) as one call: the preprocessing prover's
 * orchestration -- three sumchecks (the third a 13-prover batch), three lookups with their nine entry products, ~25
 * commitments, the 22-polynomial tensor check -- compiled into the library (gemini_amd/csrc/psnark.cpp).
 * The instance record holds what depends on the R1CS only and is resident before the prover starts: the matrices, z and w,
 * the joint support of A, B, C walked column-major (src/misc.rs:269-366) as index vectors (gm_idx_register) and as field
 * vectors, the three value vectors over it, the extended-frequency index vectors of the row / column lookups
 * (plookup/time_prover.rs:66-79), the five index commitments of Proof::index (:49-64) and the bytes the reference absorbs as
 * b"ck" (the serialised G2 powers of the key).  All caller-owned arrays: messages (cap_rounds x 8 limbs each),
 * fold_commitments (cap_folds x 18), fold_evaluations (cap_folds x 8).  products: the full products of the nine lookup
 * vectors in the order r (set, subset, sorted), alpha (..), z (..).  Same bytes as gemini_amd/psnark.py (tests). */
typedef struct gm_psnark_instance {
  uint64_t a, b, c;                 /* gm_spm handles */
  uint64_t z, w;                    /* vectors */
  uint64_t row_index, col_index;    /* gm_idx handles over the joint support */
  size_t nnz;
  uint64_t row, col, val_a, val_b, val_c; /* vectors of length nnz */
  uint64_t ext_fre_row, ext_fre_col;      /* gm_idx handles */
  size_t ext_fre_row_len, ext_fre_col_len;
  const uint64_t* index_commitments;      /* 5 x 18 limbs */
  const uint8_t* ck_g2_bytes;
  size_t ck_g2_len;
} gm_psnark_instance;
typedef struct gm_psnark_proof {
  uint64_t witness_commitment[18];
  uint64_t zc_alpha[4];
  size_t rounds[3];
  uint64_t* messages[3];
  uint64_t final_foldings[2][8];        /* first and second sumcheck */
  uint64_t third_final_foldings[13][8]; /* one (lhs, rhs) per prover of the batch */
  uint64_t r_star_commitments[3][18];
  uint64_t z_star_commitment[18];
  uint64_t sorted_commitments[3][18];   /* r, alpha, z */
  uint64_t products[9][4];
  uint64_t acc_v_commitments[9][18];
  uint64_t claimed_sumchecks[9][4];
  uint64_t ralpha_star_acc_mu_evals[10][4];
  uint64_t ralpha_star_acc_mu_proof[18];
  uint64_t rstars_vals[2][4];
  size_t nfold;
  size_t cap_folds;
  uint64_t* fold_commitments;
  uint64_t* fold_evaluations;
  uint64_t evaluation_proof[18];
  uint64_t base_evaluations[22][12];
  double spans[12];
} gm_psnark_proof;
int gm_psnark_new_time(const gm_psnark_instance* instance, uint64_t ck_bases, int g1_encoding, size_t cap_rounds, gm_psnark_proof* proof);
/* psnark::Proof::new_elastic (src/psnark/elastic_prover.rs:60-634) over device-resident big-endian streams (reversed vectors: z,
 * the witness, A z, B z, C z -- R1csStream, src/circuit.rs), everything else of `instance` as for gm_psnark_new_time (its z / w
 * fields are not read).  Commitments and openings are the chunked stream MSMs of CommitterKeyStream (src/kzg/space.rs:95-285)
 * with the flush rule of gm_snark_new_elastic (max_msm_buffer [/ depth], never below min_device_chunk); with min_device_chunk = 1
 * the sumchecks run on space provers that become time provers below SPACE_TIME_THRESHOLD rounds (sumcheck/elastic_prover.rs:44-57),
 * otherwise on time provers from the first round (the resident schedule, as for gm_snark_new_elastic); the third one as
 * prove_batch over 13 of them.  Same bytes as gm_psnark_new_time (src/psnark/tests.rs:56-124); spans as there. */
int gm_psnark_new_elastic(const gm_psnark_instance* instance, uint64_t z_stream, uint64_t w_stream, uint64_t za_stream, uint64_t zb_stream,
                          uint64_t zc_stream, uint64_t ck_bases, size_t max_msm_buffer, size_t min_device_chunk, int g1_encoding, size_t cap_rounds,
                          gm_psnark_proof* proof);

/* Everything of `instance` that depends on the MATRICES only, built inside the library from the three registered matrices
 * (`sum_matrices` + `joint_matrices`, src/misc.rs:269-366: the union of the supports of A, B, C walked column-major, the three value
 * vectors with zeros where a matrix has no entry, the last of repeated (row, column) entries kept as BTreeMap::collect does; `row` /
 * `col` as field vectors; the extended frequencies of the two lookups, plookup/time_prover.rs:66-79) and left resident in HBM:
 * fills a, b, c, row_index, col_index, nnz, row, col, val_a, val_b, val_c, ext_fre_row, ext_fre_col and their lengths.  z, w,
 * index_commitments and the G2 bytes stay the caller's.  Setup like `Proof::index` (src/psnark/time_prover.rs:49-64), once per
 * circuit; gm_psnark_preprocess_free releases what it created (not the matrices). */
int gm_psnark_preprocess(uint64_t a, uint64_t b, uint64_t c, size_t num_variables, gm_psnark_instance* instance);
int gm_psnark_preprocess_free(gm_psnark_instance* instance);
/* psnark::Proof::index (src/psnark/time_prover.rs:49-64): commitments to row, col, val_a, val_b, val_c; out = 5 x 18 limbs */
int gm_psnark_index(const gm_psnark_instance* instance, uint64_t ck_bases, uint64_t* out_jac);

/* ---- Multi-GPU: the collective layer (gemini_amd/csrc/dist.cpp) ------------------------------------------------------
 * One process per GPU (gm_init(LOCAL_RANK)).  The reference has no multi-device code: what shards is its own loop structure --
 * the MSM behind every commitment (src/kzg/time.rs:81-107: pairs split across ranks, 144-byte partial points all-gathered and
 * added on every rank), the rounds of Sumcheck::prove (src/subprotocols/sumcheck/proof.rs:36-66: 64 bytes per round), the
 * O(n) vector passes of snark::Proof::new_time (src/snark/time_prover.rs:19-117).  The ONE primitive is an all-gather; three
 * transports provide it:
 *   gm_dist_init_rccl  ncclAllGather over xGMI on a communicator of the library's own (librccl dlopen'ed at this call).
 *                      Rank 0 draws the id with gm_dist_rccl_unique_id and hands the 128 bytes to its peers out of band
 *                      (the launcher's store, a file, MPI ...).
 *   gm_dist_init_hook  the embedder's all-gather of host buffers: fn(ctx, send, bytes, recv) fills recv with the `world`
 *                      payloads in rank order and returns 0.
 *   gm_dist_init_shm   ranks of one node over a POSIX shared-memory segment `name` ("/something", the same on every rank;
 *                      slot_bytes = 0: 1 MiB per rank and call, longer payloads are cut): the payloads of this path are
 *                      host results of <= 1 KiB, which cross processes in ~1 us this way.  A name may be reused: rank 0 poisons
 *                      and removes a stale segment of that name before creating its own, a peer trusts a segment only after the
 *                      rank 0 OF THIS RUN has echoed the peer's fresh nonce, and every wait accepts a peer's call counter only at
 *                      the one or two values it can legitimately have (anything else is GM_ESTATE, never stale data).
 * Without any of them (or world = 1) an all-gather is a copy.  Collective calls must be made by every rank in the same order. */
typedef int (*gm_allgather_fn)(void* ctx, const void* send, size_t bytes, void* recv);
int gm_dist_init_hook(int rank, int world, gm_allgather_fn fn, void* ctx);
int gm_dist_rccl_unique_id(uint8_t out[128]);
int gm_dist_init_rccl(int rank, int world, const uint8_t unique_id[128]);
int gm_dist_init_shm(int rank, int world, const char* name, size_t slot_bytes);
/* RCCL on one node without any out-of-band channel: the ranks meet in the shared-memory segment `name`, which carries rank 0's
 * unique id to the peers, then build the communicator (gm_dist_rccl_unique_id + gm_dist_init_rccl in one call per rank).  The
 * segment stays open as the side channel of small host payloads (gm_dist_allgather_host_class).  A rank whose RCCL call FAILS
 * aborts the communicator (ncclCommAbort), so its peers return with an error instead of waiting inside the collective. */
int gm_dist_init_rccl_node(int rank, int world, const char* name);
int gm_dist_finalize(void);
/* A prover that fails on this rank OUTSIDE a collective would leave its peers waiting in their next all-gather until a timeout.  gm_dist_abort raises a
 * flag in the node's segment (every wait on it, and every bounded wait on an RCCL collective with the segment as side channel, returns GM_ESTATE at
 * once), aborts the communicator and poisons this rank's transport (transport 4) until it is initialised again.  The sharded provers call it on every
 * failure.  No effect with one rank; a hook transport has no channel for it. */
int gm_dist_abort(void);
/* transport: 0 none, 1 hook, 2 RCCL, 3 shm */
int gm_dist_info(int* rank, int* world, int* transport);
/* recv = world x bytes, rank order.  Host buffers (an MSM partial is finished by the host Horner; sumcheck messages and
 * evaluations are host values too). */
int gm_dist_allgather_host(const void* send, size_t bytes, void* recv);
/* The same with the payload's class.  It matters under gm_dist_init_rccl_node only, whose rendezvous segment stays open as a side
 * channel: FIELD values (sumcheck messages, evaluations, gathered tails; what gm_dist_allgather_host sends) cross it as a store and a
 * load, a few microseconds, where host staging around ncclAllGather costs H2D + collective + D2H + a stream wait; partial G1 POINTS
 * -- the all-gather behind every sharded commitment, north_star's "final RCCL reduce of partial G1 points over xGMI" -- go through
 * ncclAllGather (GM_DIST_G1_ROUTE=shm / GM_DIST_FIELD_ROUTE=rccl override either).  Measured: profiles/r5_collective_latency.txt. */
#define GM_DIST_CLASS_FIELD 0
#define GM_DIST_CLASS_G1 1
int gm_dist_allgather_host_class(const void* send, size_t bytes, void* recv, int payload_class);
/* out = the local vectors of ranks 0 .. world - 1 back to back (equal lengths; out needs capacity world x len and is
 * resized): device to device over RCCL, staged through the host on the other transports. */
int gm_dist_allgather_vec(uint64_t local_vec, uint64_t out_vec);
/* RE-BLOCKING: k device vectors, each block-distributed with equal blocks (rank p holds elements [p b, (p + 1) b), b = its local
 * length); out j (capacity >= new_block) = the elements [rank new_block, (rank + 1) new_block) of global vector j that exist, its
 * length set accordingly (0 past the end).  One group of ncclSend / ncclRecv over xGMI (every element crosses one link once); staged
 * through the host on the shm / hook transports; a copy with one rank.  What the n / g opening of gm_snark_new_time_sharded needs. */
int gm_dist_reblock_vecs(const uint64_t* local_vecs, size_t k, size_t new_block, const uint64_t* out_vecs);
/* collectives issued by this rank so far, the bytes they received and the wall time spent inside them */
int gm_dist_stats(uint64_t* calls, uint64_t* bytes, double* seconds, int reset_counters);
/* the same split by route: [0] copy (world 1), [1] hook, [2] RCCL with host staging, [3] RCCL device vectors, [4] shared memory */
int gm_dist_stats_routes(uint64_t calls[5], uint64_t bytes[5], double seconds[5]);
/* `iters` back-to-back all-gathers of `bytes` per rank -> microseconds per call.  route -1: wherever the class goes; 2 / 4: force RCCL
 * with host staging / the side segment (RCCL transport only).  Collective: every rank calls it with the same arguments. */
int gm_dist_bench(size_t bytes, int iters, int payload_class, int route, double* usec_per_call);
/* Patterns of 8 B .. 64 KiB through the active transport, checked on every rank.  With NO transport initialised it opens
 * a one-rank RCCL communicator for the test, so the binding runs on a single-GPU box too. */
int gm_dist_selftest(void);

/* A committer key SHARDED ELEMENT-CYCLICALLY over the ranks: power i of the key lives on rank i mod world, local index
 * i / world.  Every polynomial the provers commit to is a prefix of the key, foldings of length n/2, n/4, ... included, so
 * every rank gets len / world pairs (+-1) of EVERY commitment.  gm_g1_bases_set_cyclic marks a registered handle as such a
 * share (its length must be the rank's count); gm_g1_srs_register_cyclic generates the share on the device (base tau^rank g,
 * ratio tau^world: CommitterKey::new, src/kzg/time.rs:49-72, every rank its own powers).
 * gm_ck_*: CommitterKey::{commit, batch_commit} (:81-107) and the stream view of src/kzg/space.rs:95-125 against a key
 * handle that is either a whole key (then these are gm_g1_msm_v / gm_g1_msm_v_batch) or a cyclic share: strided gather of the
 * rank's scalars (the polynomials are replicated), local MSMs as one pipelined batch, ONE all-gather of k x 144 bytes, the EC
 * adds -- identical bytes on every rank.  offset / reversed address the GLOBAL key as in gm_g1_msm_v.
 * gm_snark_new_time, gm_snark_new_elastic, gm_psnark_new_time and gm_psnark_index commit through these: handed a cyclic
 * share they run on N GPUs with the MSMs sharded and the field arithmetic replicated. */
int gm_g1_bases_set_cyclic(uint64_t handle, size_t n_global, int rank, int world);
int gm_g1_srs_register_cyclic(const uint64_t base_affine[12], const uint64_t tau[4], size_t n_global, int rank, int world, uint64_t* handle);
int gm_ck_len(uint64_t ck, size_t* n_global);
int gm_ck_msm(uint64_t ck, size_t offset, int reversed, uint64_t vec, size_t voffset, size_t n, uint64_t out_jac[18]);
int gm_ck_msm_batch(uint64_t ck, const uint64_t* vecs, const size_t* ns, size_t k, uint64_t* out_jac);

/* Sumcheck::prove (src/subprotocols/sumcheck/proof.rs:36-66) with f and g BLOCK-sharded: this rank holds elements
 * [lo, lo + len) (lo even) of vectors of n_global elements.  Per round the rank's partial message (64 bytes) is all-gathered
 * and summed mod r; when the blocks get short (<= 2^10 elements) they are gathered once and every rank finishes the protocol
 * on the whole vectors.  Same outputs as gm_sumcheck_prove, on every rank. */
int gm_sumcheck_prove_sharded(uint64_t transcript, uint64_t f_block, uint64_t g_block, const uint64_t twist_mont[4], size_t lo, size_t n_global,
                              uint64_t* messages, uint64_t* challenges, size_t cap_rounds, uint64_t final_foldings[8], size_t* rounds);

/* snark::Proof::new_time (src/snark/time_prover.rs:19-117) with the FIELD ARITHMETIC sharded as well: rank r of g (powers of
 * two) holds elements [r m, (r + 1) m), m = n / g, of every vector of the prover.
 *   matrices     row blocks [r m, (r + 1) m) of A, B, C and of A^T, B^T, C^T (gm_spm handles, m rows each), either with GLOBAL
 *                column indices (m x n: any Matrix<F>, src/misc.rs:100-110, src/circuit.rs:43 -- then z is the WHOLE z on every
 *                rank, and tensor(rho) / powers(alpha) are computed whole on every rank: an O(n) pass at HBM speed is cheaper
 *                than n elements over xGMI) or with LOCAL ones (m x m: a block-diagonal instance such as dummy_r1cs,
 *                src/circuit.rs:349-365 -- then z is this rank's block and nothing of size n is ever touched)
 *   w_block      elements [r m, (r + 1) m) of w (the top rank's block is shorter: |w| = n - |x|)
 *   key          gm_snark_shard_key_new: this rank's slice of the key for every block-sharded level of the folding tree plus the
 *                replicated prefix for the gathered levels, ONE handle (offsets / counts: key_segments entries)
 * Sumchecks through gm_sumcheck_prove_sharded; foldings keep the top index bits, so level j stays block-sharded while its
 * blocks hold >= 2^tail_log elements and is gathered after that; commitments are one pipelined batch against the slices and
 * one all-gather; evaluations are block evaluations scaled on the host; the opening is sum_i eta_i commit(p_i div Z) with the
 * carry between blocks interpolated from the evaluations already gathered.  The proof is byte-identical to
 * gm_snark_new_time's on every rank (tests/test_gpu_dist_native.py: 1 / 2 / 4 / 8 ranks). */
typedef struct gm_snark_shard {
  uint64_t matrices[6];
  uint64_t z;
  uint64_t w_block;
  uint64_t key;
  const size_t* key_offsets;
  const size_t* key_counts;
  size_t key_segments;
  size_t n;
  size_t tail_log;
} gm_snark_shard;
int gm_snark_shard_key_new(const uint64_t base_affine[12], const uint64_t tau[4], size_t n, size_t tail_log, uint64_t* key, size_t offsets[64],
                           size_t counts[64], size_t* segments);
int gm_snark_new_time_sharded(const gm_snark_shard* shard, int g1_encoding, size_t cap_rounds, gm_snark_proof* proof);
/* snark::Proof::new_elastic (src/snark/elastic_prover.rs:174-266; BASELINE configs[3]: `snark -i 28`, elastic, 8 GPUs) over the same blocks: the
 * RESIDENT schedule of gm_snark_new_elastic (min_device_chunk > 1: time provers on the little-endian vectors, every stream MSM one call) over
 * blocks is the block-sharded time prover.  min_device_chunk = 1 (the literal schedule) is refused: it is single-GPU.  The DummyStreamer key of
 * examples/snark.rs:59-63 in slices: gm_snark_shard_key_new with tau = 1.  Same bytes as gm_snark_new_elastic / gm_snark_new_time. */
int gm_snark_new_elastic_sharded(const gm_snark_shard* shard, size_t max_msm_buffer, size_t min_device_chunk, int g1_encoding, size_t cap_rounds,
                                 gm_snark_proof* proof);

/* psnark::Proof::new_time (src/psnark/time_prover.rs:69-384; the resident schedule of new_elastic, elastic_prover.rs:60-634, is the same
 * entry) with the FIELD ARITHMETIC block-sharded as well -- BASELINE configs[4] on N GPUs.  One block size B for the proof
 * (gm_psnark_shard_block(longest vector, world); any world, no power-of-two requirement) and a level per family of vectors
 * (gm_psnark_shard_level, below): rank r of g holds the elements [r (B >> s), (r + 1) (B >> s)) THAT EXIST of a vector of level s (a block may
 * be partial or empty: handle 0 or a vector of length 0).
 *   a, b, c        the row blocks of A, B, C with GLOBAL column indices (gm_spm handles; 0 when no row falls into the block)
 *   z              the WHOLE z on every rank (the lookups of z* and the matrix products read arbitrary entries)
 *   w_block        this rank's block of w, w_len its whole length
 *   row_index .. ext_fre_col   this rank's blocks of the joint-matrix index vectors, field vectors and extended frequencies
 *                  (gm_psnark_instance, whole lengths nnz / ext_fre_*_len)
 *   key            gm_psnark_shard_key_new: this rank's slices of the key, key_len = powers of the whole key
 *   index_commitments   gm_psnark_index_sharded (Proof::index over the same blocks)
 * What crosses ranks: partial G1 points (144 B per commitment), 64 B per prover and sumcheck round (the 13 provers of the third
 * sumcheck in ONE all-gather per round), the products of the blocks of the nine lookup vectors (the carries of the suffix scans of
 * accumulated_product, entryproduct/time_prover.rs:34-45) and 32-byte halos (right_rotation, plookup_set read element i - 1),
 * evaluations, short gathered tails, the re-blocked level sums of the opening.  `lookup(v, index)` (plookup/time_prover.rs:5-8)
 * needs no exchange and no whole vector: tensor(rho) and powers(alpha) are functions of the index (gm_fr_tensor_gather / gm_fr_powers_gather).
 * The proof is byte-identical to gm_psnark_new_time's on every rank (tests/test_gpu_dist_native.py: 1 / 2 / 3 / 4 / 8 ranks). */
typedef struct gm_psnark_shard {
  uint64_t a, b, c;
  uint64_t z;
  uint64_t w_block;
  size_t w_len;
  uint64_t row_index, col_index;
  uint64_t row, col, val_a, val_b, val_c;
  uint64_t ext_fre_row, ext_fre_col;
  size_t ext_fre_row_len, ext_fre_col_len;
  size_t num_constraints, num_variables, nnz;
  size_t block;
  size_t tail_log;
  uint64_t key;
  const size_t* key_offsets;
  const size_t* key_counts;
  size_t key_segments;
  size_t key_len;
  const uint64_t* index_commitments; /* 5 x 18 limbs */
  const uint8_t* ck_g2_bytes;
  size_t ck_g2_len;
} gm_psnark_shard;
size_t gm_psnark_shard_block(size_t longest, int world);
/* LEVELS.  The prover's vectors come in several lengths (dummy_r1cs: ~n and ~2 n; a general instance: n, nnz ~ 6 n, n + nnz): with one block size
 * the short ones would sit on the lower ranks only.  A FAMILY of vectors whose longest member has `len` elements lives in blocks of
 * block >> level, level = the largest one (up to the number of folding levels that stay sharded) whose `world` blocks still hold it: every rank
 * holds ~1 / world of every vector.  The caller cuts its inputs accordingly:
 *   a, b, c          level(num_constraints)        w_block                     level(w_len)
 *   row_index, col_index, row, col, val_a, val_b, val_c                        level(nnz + 1)
 *   ext_fre_row      level(ext_fre_row_len + 2)    ext_fre_col                 level(ext_fre_col_len + 2)
 * (a vector of level s is cut at multiples of block >> s).  Combinations across levels are re-blocked inside the prover. */
size_t gm_psnark_shard_level(size_t len, size_t block, size_t tail_log, int world);
int gm_psnark_shard_key_new(const uint64_t base_affine[12], const uint64_t tau[4], size_t n_key, size_t block, size_t tail_log, uint64_t* key,
                            size_t offsets[64], size_t counts[64], size_t* segments);
int gm_psnark_index_sharded(const gm_psnark_shard* shard, uint64_t* out_jac);
/* The footprint contract (gm_psnark_footprint) of ONE RANK of the block-sharded prover: out = {vectors and prover buffers, MSM workspaces still
 * to grow, needed, what can be had}, bytes.  admit != 0: also make room (prefix tables are given back) or fail with GM_ENOMEM and the numbers --
 * what gm_psnark_new_time_sharded does before its first allocation.  z (whole on every rank) and the instance's blocks are the caller's. */
int gm_psnark_shard_footprint(uint64_t key, size_t num_constraints, size_t num_variables, size_t nnz, size_t block, int world, int admit, uint64_t out[4]);
int gm_psnark_new_time_sharded(const gm_psnark_shard* shard, int g1_encoding, size_t cap_rounds, gm_psnark_proof* proof);

#ifdef __cplusplus
}
#endif
#endif /* GEMINI_HIP_H */
