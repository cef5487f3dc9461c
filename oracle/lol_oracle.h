/*
 * lol_oracle.h — CPU oracle for loltracer's per-pixel ray-march path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar-C restatement of
 * /root/reference/naive_renderer.c (+ sdf.h, float.h, vec.h), used as the
 * checker for the HIP renderer.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load it; the product (loltracer_amd/,
 * include/lol_gpu.h) never links or calls anything in oracle/.
 *
 * Pinning (details and honest limits in LABNOTES.md §5): the reference ships no tests or golden
 * images, and naive_renderer.c itself cannot be compiled in this image (it
 * includes <SDL.h> through renderer.h; SDL2 is absent).  The restatement is pinned by
 *  (1) oracle/_ref — the reference's own SDL-free headers (float.h, vec.h, sdf.h) and scene.c
 *      compiled where they lie; their outputs on committed input vectors
 *      (tests/golden/ref_primitives.json, ref_scenes.json) must match bit for bit;
 *  (2) compositions of those compiled pieces: the scene SDF walked over the reference's own struct object
 *      tree (tests/golden/ref_sdf_points.json) and WHOLE PIXELS — naive_renderer.c:48-236 followed statement by
 *      statement by tests/golden/make_golden.py with every vector / min / max / clamp / smooth-min / distance
 *      operation executed by the reference's compiled code (ref_frames.npz: four 64x36 frames; ref_pixels.json:
 *      per-stage values) — all reproduced bit for bit;
 *  (3) the known pixels and per-pixel work counters the survey recorded from the unmodified
 *      naive_renderer.c — reproduced by this oracle AND by the composition of (2).
 * What no fixture can come from is naive_renderer.c's own object code: its loop and branch structure is
 * restated (here and in make_golden.py), its arithmetic is the reference's.
 */
#ifndef LOL_ORACLE_H
#define LOL_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "lol_scene.h"   /* data types only (include/), no product code is linked */

#ifdef __cplusplus
extern "C" {
#endif

/* Work counters summed over the rendered pixels (SURVEY.md §8d "algorithmic flops"). */
typedef struct lol_oracle_counters {
	uint64_t pixels;
	uint64_t sdf_evals;        /* calls of sdf()                      naive_renderer.c:31 */
	uint64_t node_evals;       /* calls of get_obj_dist()             naive_renderer.c:11 */
	uint64_t march_steps;      /* iterations of get_intersection loop naive_renderer.c:56 */
	uint64_t shadow_steps;     /* iterations of softshadow loop       naive_renderer.c:80 */
	uint64_t miss_pixels;      /* id == 0 after the march */
	uint64_t settle_violations;/* checker-side: shadow rays whose running factor was <= 0 at some step and still came out
	                            * != 0 — the property the renderer's early exit (FLAG_SHADOW_SETTLED) rests on; must be 0 */
} lol_oracle_counters;

/* The surface's SDL_PixelFormat fields that SDL_MapRGB reads (renderer.h:17-22); NULL / default = XRGB8888.
 * Process-wide state of the checker (tests only). */
typedef struct lol_oracle_pixel_format {
	uint8_t  r_shift, g_shift, b_shift, r_loss, g_loss, b_loss;
	uint32_t a_mask;
} lol_oracle_pixel_format;
void lol_oracle_set_pixel_format(const lol_oracle_pixel_format* f);

/* Per-pixel probe of the intermediate values (for debugging parity failures).  The per-light fields hold the first
 * LOL_ORACLE_PROBE_LIGHTS lights of the scene (a scene may have any number: lol_scene.h). */
#define LOL_ORACLE_PROBE_LIGHTS 64
typedef struct lol_oracle_probe {
	float    rd[3];
	float    hit_dist;
	uint32_t hit_id;
	uint32_t march_steps;
	float    normal[3];
	float    shadow[LOL_ORACLE_PROBE_LIGHTS];
	uint32_t shadow_steps[LOL_ORACLE_PROBE_LIGHTS];
	float    rgb_linear[3];    /* get_light() result, before gamma */
	float    rgb[3];           /* after gamma */
	uint32_t xrgb;
} lol_oracle_probe;

/*
 * Render rows [y0, y1) of a w×h frame of `scene` seen from `cam`
 * (naive_renderer.c:207-236).  max_steps is MAX_STEPS of get_intersection
 * (256 in the reference, naive_renderer.c:49).
 *   xrgb   : h rows of `pitch_bytes`; pixel (x,y) at xrgb + y*pitch + 4*x,
 *            value r<<16|g<<8|b (renderer.h:17-22 with an XRGB8888 surface)
 *   rgb    : optional w*h*3 floats, post-gamma pre-quantisation, row-major
 *   steps  : optional w*h x 12 uint16 per pixel: {march steps, shadow steps (all lights), hit id,
 *            bit mask of lights whose diffuse incidence is exactly 0, shadow steps of lights 0..3,
 *            "settled" shadow steps of lights 0..3 = steps up to and including the first that left res <= 0}
 *   ctr    : optional counters, accumulated (caller zeroes)
 */
void lol_oracle_render_rows(const lol_scene* scene, const lol_camera* cam,
                            int w, int h, int max_steps, int y0, int y1,
                            void* xrgb, size_t pitch_bytes, float* rgb,
                            uint16_t* steps, lol_oracle_counters* ctr);

/*
 * Whole frame on `threads` workers that claim rows from one atomic counter,
 * the reference's scheduling (naive_renderer.c:216, main.c:189-194).
 */
void lol_oracle_render_frame(const lol_scene* scene, const lol_camera* cam,
                             int w, int h, int max_steps, int threads,
                             void* xrgb, size_t pitch_bytes, float* rgb,
                             lol_oracle_counters* ctr);

/* Rows y = y0 + k*stride < y_end only (bounded sample for the CPU baseline). */
void lol_oracle_render_sample(const lol_scene* scene, const lol_camera* cam,
                              int w, int h, int max_steps, int threads,
                              int y0, int y_end, int stride,
                              void* xrgb, size_t pitch_bytes,
                              lol_oracle_counters* ctr);

void lol_oracle_probe_pixel(const lol_scene* scene, const lol_camera* cam,
                            int w, int h, int max_steps, int x, int y,
                            lol_oracle_probe* out);

/* sdf(p) alone: returns distance, stores the 1-based object id (0 = none). */
float lol_oracle_sdf(const lol_scene* scene, float px, float py, float pz, uint32_t* id);

/* Primitives exported for the oracle/_ref cross-check (float.h, vec.h, sdf.h). */
float lol_oracle_minf(float a, float b);
float lol_oracle_maxf(float a, float b);
float lol_oracle_clamp(float v, float lo, float hi);
float lol_oracle_sminf(float a, float b, float k);
float lol_oracle_v3dot(const float a[3], const float b[3]);
float lol_oracle_v3len(const float a[3]);
void  lol_oracle_v3normalize(const float a[3], float out[3]);
void  lol_oracle_v3cross(const float a[3], const float b[3], float out[3]);
void  lol_oracle_v3clamp(const float a[3], float lo, float hi, float out[3]);
float lol_oracle_sd_sphere(const float p[3], float r);
float lol_oracle_sd_round_box(const float p[3], const float b[3], float r);

/* out[i] = powf(x[i], y[i]) with the host libm (what naive_renderer.c calls at :158 and :231 via vec.h:66-67) */
void lol_oracle_powf_batch(const float* x, const float* y, float* out, size_t n);

/* FNV-1a-style hash over 32-bit pixels, row-major, skipping row padding
 * (the survey's known-answer hash). */
uint64_t lol_oracle_hash_xrgb(const void* xrgb, int w, int h, size_t pitch_bytes);

#ifdef __cplusplus
}
#endif
#endif
