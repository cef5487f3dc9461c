/*
 * lol_gpu_diag.h — what tests, profiles and a curious maintainer ask liblol_gpu.so beside the frames: which tuning
 * switches took effect, the exhaustive proofs behind the kernel's shortcuts run one by one, the culling bounds of an
 * object, and the scene SDF and the renderer's powf on their own.  NOTHING a host needs to render: the plug-in
 * (integration/hip_renderer.c) and the other hosts include lol_gpu.h alone.  Same library, same ABI version.
 */
#ifndef LOL_GPU_DIAG_H
#define LOL_GPU_DIAG_H

#include "lol_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The tuning switches in effect in this process, "NAME=value NAME=value ..." ("" when none).  The library has ten
 * LOL_GPU_* environment switches for A/B runs and debugging (INTEGRATION.md lists them) — the scene kernel's instruction scheduling
 * among them.  They are honoured ONLY in a process that also sets LOL_GPU_TUNING=1; one that is set without it is ignored and
 * reported once on stderr; every one that took effect is listed here, in lol_gpu_specialize_log() and in bench.py's
 * record, so that a number can never silently come from a shell's leftovers.  (Not fenced: LOL_GPU_CACHE_DIR, LOL_GPU_ROCTX —
 * where code objects are kept, whether frames are marked: neither changes what is computed.)  The string belongs to the library
 * and is valid until the next call of this function. */
const char* lol_gpu_tuning_switches(void);
/* Frame ranges pushed to roctx so far by this process (LOL_GPU_ROCTX=1 marks every frame launch for
 * `rocprofv3 --marker-trace`, the counterpart of the reference's -j/--jitdump aid); 0 when not asked for, -1 when asked
 * for but no roctx library could be loaded (also reported once on stderr). */
long lol_gpu_roctx_ranges(void);

/*
 * The specialised kernel may replace sqrt and the smooth-min division x/k by cheaper sequences
 * (lol_kernel.h "fast exact paths").  Each is used only after the device has run ALL 2^32 float
 * inputs through it and through the plain expression and found no difference; this call runs
 * those checks directly and returns the mismatch counts (0 = proven; ~0 = could not run).
 */
/* sqrt_mismatches[0..2] = sqrt_pm, sqrt_gs, sqrt_r2 (lol_kernel.h); div_mismatches for the divisor k */
int         lol_gpu_verify_fast_paths(lol_gpu* ctx, float k, unsigned long long sqrt_mismatches[3],
                                      unsigned long long* div_mismatches);
/* The same exhaustive run for the blend factor WITHOUT its v_div_fixup_f32 (lol_kernel.h, smin_h_fast<false>): inputs on
 * which it differs from the exact factor, or — for dlt = +-inf — fails to make the smooth minimum NaN.  0 = proven. */
int         lol_gpu_verify_smin_no_fixup(lol_gpu* ctx, float k, unsigned long long* mismatches);
/* Gamma and quantisation of a colour channel — Uint8 v = powf(c, 1 / 2.2f) * 255 (naive_renderer.c:231-232, renderer.h:17-22) —
 * through a table of 256 thresholds instead of the powf (lol_kernel.h, gamma_u8_table): used by frames only after this sweep of
 * every float in [0, 1] found no difference on the context's device (it runs at the first upload; lol_gpu_set_specialize(ctx, 3) keeps the
 * powf).  *mismatches = floats on which the two routes differ (0 = proven, ~0 = could not run); table (may be NULL): the 257
 * thresholds, T[k] = the smallest c whose channel value is >= k, T[0] = 0, T[256] = +inf. */
int         lol_gpu_verify_gamma_table(lol_gpu* ctx, unsigned long long* mismatches, float* table);

/* The bound behind the culling test (lol_gpu.h, lol_gpu_set_cull) for top-level object `root` (0-based, file order); no device needed.  Returns 1 and
 * the bounding sphere (centre, inflated radius R') when the object has one, 0 when it has none (planes, unions
 * with a plane or with smoothness <= 0, non-finite fields) and is therefore never culled. */
int         lol_gpu_cull_bounds(const lol_program* prog, uint32_t root, float c_out[3], float* r_out);
/* The tighter bound of a union of at least three primitives whose leaves split into two clusters much smaller than the
 * one enclosing sphere: value(p) >= min_j (|p - c_j| - r'_j).  Returns how many spheres out[j] = {cx, cy, cz, r'} were
 * written (0: the object is tested with its single sphere only; else 2); the specialised kernel skips such an object where
 * the tests of BOTH spheres pass. */
int         lol_gpu_cull_bounds_clusters(const lol_program* prog, uint32_t root, float out[3][4]);
/*
 * Diagnostic: out[i] = the renderer's powf(x[i], y[i]) (device pointers, asynchronous on `stream`, NULL = the
 * context's stream).  The kernel's powf restates the algorithm of the CPU libm's powf so that colours round
 * identically on both sides (lol_kernel.h, powf_glibc); this entry point lets a test compare the two bit for bit.
 */
int         lol_gpu_powf_batch(lol_gpu* ctx, const float* x_dev, const float* y_dev, float* out_dev, size_t n,
                               void* stream);
/*
 * Diagnostic: the scene SDF alone — sdf() of naive_renderer.c:31-44, i.e. get_obj_dist over every top-level object
 * with the first strict minimum — at n arbitrary points: pts = n x {x, y, z}, dist[i] / id[i] out (device pointers,
 * asynchronous on `stream`).  Runs the SAME SDF code the frames run (the specialised module's or the interpreter's,
 * fast paths and their fallback included), so a test can hold the device's distances against known answers
 * (tests/golden/ref_sdf_points.json) without a march in between.
 */
int         lol_gpu_sdf_batch(lol_gpu* ctx, const float* pts_dev, float* dist_dev, uint32_t* id_dev, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOL_GPU_DIAG_H */
