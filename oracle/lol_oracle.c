/*
 * lol_oracle.c — scalar-C oracle of loltracer's per-pixel path.  TEST INFRASTRUCTURE ONLY
 * (see lol_oracle.h for who may use it and how it is pinned).
 *
 * Every function cites the reference lines it restates.  Arithmetic is IEEE
 * binary32 in the reference's operation order; the SSE intrinsics of float.h /
 * vec.h are spelled out as the scalar expressions they compute:
 *   _mm_min_ss(a,b) = a < b ? a : b      (second operand on NaN / equal)
 *   _mm_max_ss(a,b) = a > b ? a : b
 *   _mm_dp_ps(a,b,0x71) = (ax*bx + ay*by) + (az*bz + 0)
 * Build with -ffp-contract=off and never with -ffast-math / -mfma: the
 * reference is compiled without FMA (Makefile:3), so no product may be fused.
 */
#include "lol_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

struct tally {               /* per-pixel work counters, folded into lol_oracle_counters */
	uint64_t sdf_evals, node_evals, march_steps, shadow_steps;
	uint64_t settled_steps;        /* shadow steps up to and including the first one that left res <= 0 (all of them if none did) */
	uint64_t settle_violations;    /* shadow rays whose res was <= 0 at some step and whose factor still came out != 0 */
};

/* ------------------------------------------------------------------ float.h */

static inline float maxf(float a, float b) { return a > b ? a : b; }            /* float.h:6-9   */
static inline float minf(float a, float b) { return a < b ? a : b; }            /* float.h:11-14 */
static inline float clampf(float v, float lo, float hi) {                        /* float.h:16-22 */
	return minf(maxf(v, lo), hi);
}
static inline float lerpf(float from, float to, float ratio) {                   /* float.h:24-27 */
	return from + (to - from) * ratio;
}
static inline float sminf(float a, float b, float k) {                           /* float.h:29-33 */
	float h = clampf(.5f + .5f * (b - a) / k, 0.f, 1.f);
	return lerpf(b, a, h) - k * h * (1.f - h);
}

/* -------------------------------------------------------------------- vec.h */

static inline v3 v3add(v3 a, v3 b) { return (v3){ a.x + b.x, a.y + b.y, a.z + b.z }; }   /* vec.h:42 */
static inline v3 v3sub(v3 a, v3 b) { return (v3){ a.x - b.x, a.y - b.y, a.z - b.z }; }   /* vec.h:44 */
static inline v3 v3mul(v3 a, v3 b) { return (v3){ a.x * b.x, a.y * b.y, a.z * b.z }; }   /* vec.h:46 */
static inline float v3dot(v3 a, v3 b) {                                                  /* vec.h:50-51 */
	float lo = a.x * b.x + a.y * b.y;
	float hi = a.z * b.z + 0.0f;        /* the masked w lane contributes +0 */
	return lo + hi;
}
static inline float v3len(v3 a) { return sqrtf(v3dot(a, a)); }                           /* vec.h:52-53 */
static inline v3 v3scale(v3 v, float f) { return (v3){ v.x * f, v.y * f, v.z * f }; }    /* vec.h:56-57 */
static inline v3 v3normalize(v3 v) { return v3scale(v, 1 / v3len(v)); }                  /* vec.h:58-59 */
static inline v3 v3abs(v3 v) { return (v3){ fabsf(v.x), fabsf(v.y), fabsf(v.z) }; }      /* vec.h:60-62 */
static inline v3 v3clamp(v3 v, float lo, float hi) {                                     /* vec.h:63-65 */
	/* _mm_max_ps(_mm_min_ps(v, hi), lo): NaN → hi */
	return (v3){ maxf(minf(v.x, hi), lo), maxf(minf(v.y, hi), lo), maxf(minf(v.z, hi), lo) };
}
static inline v3 v3pow(v3 v, float e) { return (v3){ powf(v.x, e), powf(v.y, e), powf(v.z, e) }; }   /* vec.h:66-67 */
static inline v3 v3cross(v3 a, v3 b) {                                                   /* vec.h:68-71 */
	return (v3){ a.y * b.z - a.z * b.y,
	             a.z * b.x - a.x * b.z,
	             a.x * b.y - a.y * b.x };
}
static inline v3 from_lol(lol_v3 v) { return (v3){ v.x, v.y, v.z }; }

/* -------------------------------------------------------------------- sdf.h */

static inline float sdSphere(v3 p, float s) { return v3len(p) - s; }             /* sdf.h:8-10 */
static inline float sdRoundBox(v3 p, v3 b, float r) {                            /* sdf.h:18-22 */
	v3 q = v3sub(v3abs(p), b);
	v3 cq = { maxf(q.x, 0.f), maxf(q.y, 0.f), maxf(q.z, 0.f) };
	return v3len(cq) + minf(maxf(q.x, maxf(q.y, q.z)), 0.f) - r;
}

/* --------------------------------------------------------- naive_renderer.c */

struct world_dist { float dist; uint32_t id; };                                  /* :5-8 */

/* get_obj_dist, naive_renderer.c:11-28 (children receive the untranslated p) */
static float get_obj_dist(const lol_scene* sc, int32_t idx, v3 p, struct tally* t) {
	const lol_node* obj = &sc->nodes[idx];
	v3 point = v3sub(p, from_lol(obj->point));
	t->node_evals++;
	switch (obj->type) {
	case LOL_NODE_SPHERE:
		return sdSphere(point, obj->radius);
	case LOL_NODE_BOX:
		return sdRoundBox(point, from_lol(obj->half_extent), obj->radius);
	case LOL_NODE_PLANE:
		return point.y;
	case LOL_NODE_SMOOTH_UNION: {
		float a_dist = get_obj_dist(sc, obj->a, p, t);
		float b_dist = get_obj_dist(sc, obj->b, p, t);
		return sminf(a_dist, b_dist, obj->smoothness);
	}
	default:
		return 0.f;
	}
}

/* sdf, naive_renderer.c:31-44: strict '<', first minimum wins, ids 1-based */
static struct world_dist sdf(const lol_scene* sc, v3 p, struct tally* t) {
	struct world_dist rval = { INFINITY, 0 };
	t->sdf_evals++;
	for (size_t i = 0; i < sc->n_roots; i++) {
		float d = get_obj_dist(sc, sc->roots[i], p, t);
		if (d < rval.dist)
			rval = (struct world_dist){ d, (uint32_t)(i + 1) };
	}
	return rval;
}

/* get_intersection, naive_renderer.c:48-69 (MAX_STEPS made a parameter) */
static struct world_dist get_intersection(const lol_scene* sc, v3 ro, v3 rd, int max_steps,
                                          struct tally* t) {
	const float EPSILON = 0.001f, MAX_DIST = 100.f;
	uint32_t id = 0;
	float dist = 0.f;
	for (int i = 0; i < max_steps; i++) {
		v3 p = v3add(ro, v3scale(rd, dist));
		struct world_dist sd = sdf(sc, p, t);
		t->march_steps++;
		dist += sd.dist;
		id = sd.id;
		if (sd.dist < EPSILON || dist > MAX_DIST)
			break;
	}
	if (dist >= MAX_DIST)
		id = 0;
	return (struct world_dist){ dist, id };
}

/* softshadow, naive_renderer.c:73-90 */
static float softshadow(const lol_scene* sc, v3 ro, v3 rd, int max_steps, float max_dist, float w,
                        struct tally* t) {
	float res = 1.f, dist = 0.f;
	int settled = 0;      /* checker-side bookkeeping only (not in the reference): see lol_oracle.h, settled_steps */
	for (int i = 0; i < max_steps; i++) {
		v3 p = v3add(ro, v3scale(rd, dist));
		float scene_dist = sdf(sc, p, t).dist;
		t->shadow_steps++;
		if (!settled) t->settled_steps++;
		res = minf(res, w * scene_dist / dist);
		dist += scene_dist;
		if (res <= 0.f) settled = 1;
		if (res < -1 || dist > max_dist)
			break;
	}
	if (settled && maxf(res, 0.f) != 0.f) t->settle_violations++;
	return maxf(res, 0.f);
}

/* in_shadow, naive_renderer.c:93-100 */
static float in_shadow(const lol_scene* sc, const lol_light* light, v3 p, struct tally* t) {
	v3 lp = from_lol(light->point);
	float light_dist = v3len(v3sub(lp, p));
	v3 dir = v3normalize(v3sub(lp, p));
	p = v3add(p, dir);
	return softshadow(sc, p, dir, 128, light_dist, 50.f, t);
}

/* get_material, naive_renderer.c:103-112 */
static lol_material get_material(const lol_scene* sc, uint32_t obj_id) {
	size_t mid = obj_id ? sc->nodes[sc->roots[obj_id - 1]].material : 0;
	return sc->materials[mid];
}

/* get_normal, naive_renderer.c:114-125 */
static v3 get_normal(const lol_scene* sc, v3 p, float dist, struct tally* t) {
	const v3 k0 = {  1.f, -1.f, -1.f }, k1 = { -1.f, -1.f,  1.f };
	const v3 k2 = { -1.f,  1.f, -1.f }, k3 = {  1.f,  1.f,  1.f };
	const float h = dist / 100.f;
	const v3 p0 = v3scale(k0, sdf(sc, v3add(p, v3scale(k0, h)), t).dist);
	const v3 p1 = v3scale(k1, sdf(sc, v3add(p, v3scale(k1, h)), t).dist);
	const v3 p2 = v3scale(k2, sdf(sc, v3add(p, v3scale(k2, h)), t).dist);
	const v3 p3 = v3scale(k3, sdf(sc, v3add(p, v3scale(k3, h)), t).dist);
	return v3normalize(v3add(p0, v3add(p1, v3add(p2, p3))));
}

/* get_light, naive_renderer.c:129-175 */
static v3 get_light(const lol_scene* sc, v3 cam_pos, v3 p, v3 n, uint32_t obj_id, struct tally* t,
                    lol_oracle_probe* probe, uint32_t* dark_mask, uint16_t* per_light) {
	lol_material mat = get_material(sc, obj_id);
	v3 total = { 0.f, 0.f, 0.f };
	for (size_t li = 0; li < sc->n_lights; li++) {
		const lol_light* light = &sc->lights[li];
		uint64_t before = t->shadow_steps, settled_before = t->settled_steps;
		float shadow = in_shadow(sc, light, p, t);
		if (probe && li < LOL_ORACLE_PROBE_LIGHTS) {
			probe->shadow[li] = shadow;
			probe->shadow_steps[li] = (uint32_t)(t->shadow_steps - before);
		}
		if (per_light && li < 4) {
			per_light[li] = (uint16_t)(t->shadow_steps - before);
			per_light[4 + li] = (uint16_t)(t->settled_steps - settled_before);
		}
		v3 Id = from_lol(light->diffuse_intensity);
		v3 Is = from_lol(light->specular_intensity);
		v3 light_dir = v3normalize(v3sub(from_lol(light->point), p));
		v3 reflected = v3sub(v3scale(n, 2.f * v3dot(light_dir, n)), light_dir);
		v3 camera_dir = v3normalize(v3sub(cam_pos, p));

		float diffuse_incidence = clampf(v3dot(n, light_dir), 0.f, 1.f);
		if (dark_mask && li < 16 && diffuse_incidence == 0.f) *dark_mask |= 1u << li;
		Id = v3scale(Id, shadow * diffuse_incidence);
		Id = v3mul(Id, from_lol(mat.diffuse));
		total = v3add(total, Id);

		float specular_incidence = diffuse_incidence *
			powf(clampf(v3dot(reflected, camera_dir), 0.f, 1.f), mat.shininess);
		Is = v3scale(Is, shadow * specular_incidence);
		Is = v3mul(Is, from_lol(mat.specular));
		total = v3add(total, Is);
	}
	total = v3add(total, v3mul(from_lol(sc->ambient_color), from_lol(mat.ambient)));
	return v3clamp(total, 0.f, 1.f);
}

/* get_camera_ray, naive_renderer.c:179-193 (recomputed per pixel, as there) */
static v3 get_camera_ray(const lol_camera* cam, float vx, float vy, float aspect) {
	v3 up_guide = { 0.f, 1.f, 0.f };
	v3 cdir = from_lol(cam->direction);
	float half_fov = cam->fov / 2.f;
	float height = atanf(half_fov);
	float width = aspect * height;
	v3 right = v3normalize(v3cross(cdir, up_guide));
	v3 up = v3cross(right, cdir);
	v3 r = v3add(v3scale(right, vx * width), v3scale(up, vy * height));
	return v3normalize(v3add(r, cdir));
}

/* colorf_to_pixfmt, renderer.h:17-22: Uint8 channels, then SDL_MapRGB(fmt, r, g, b).  SDL2 is a third-party
 * dependency of the reference (Makefile:3-4, `sdl2-config`, no version pinned; absent from this image); for a
 * non-palettised format its SDL_MapRGB is (SDL 2.0.x src/video/SDL_pixels.c)
 *     (r >> Rloss) << Rshift | (g >> Gloss) << Gshift | (b >> Bloss) << Bshift | Amask
 * — restated here over the SDL_PixelFormat fields of the surface.  Default XRGB8888 (shifts 16/8/0, no loss, Amask 0),
 * i.e. r << 16 | g << 8 | b.  Process-wide; set by the tests that check other formats. */
static lol_oracle_pixel_format g_fmt = { 16, 8, 0, 0, 0, 0, 0 };
void lol_oracle_set_pixel_format(const lol_oracle_pixel_format* f) {
	static const lol_oracle_pixel_format xrgb = { 16, 8, 0, 0, 0, 0, 0 };
	g_fmt = f ? *f : xrgb;
}
static uint32_t pack_xrgb(v3 c) {
	uint8_t r = (uint8_t)(c.x * 255);
	uint8_t g = (uint8_t)(c.y * 255);
	uint8_t b = (uint8_t)(c.z * 255);
	return (uint32_t)(r >> g_fmt.r_loss) << g_fmt.r_shift | (uint32_t)(g >> g_fmt.g_loss) << g_fmt.g_shift |
	       (uint32_t)(b >> g_fmt.b_loss) << g_fmt.b_shift | g_fmt.a_mask;
}

/* the pixel body of render_thread, naive_renderer.c:217-235 */
static uint32_t shade_pixel(const lol_scene* sc, const lol_camera* cam, int x, int y,
                            float fwidth, float fheight, int max_steps, struct tally* t,
                            float* rgb_out, lol_oracle_probe* probe, int* missed, uint32_t* hit_id,
                            uint32_t* dark_out, uint16_t* per_light) {
	v3 ro = from_lol(cam->point);
	float aspect = fwidth / fheight;
	float vx = (x + .5f) / fwidth * 2.f - 1.f;
	float vy = 1.f - (y + .5f) / fheight * 2.f;

	v3 rd = get_camera_ray(cam, vx, vy, aspect);
	struct world_dist hit = get_intersection(sc, ro, rd, max_steps, t);
	uint64_t march = t->march_steps;
	v3 p = v3add(ro, v3scale(rd, hit.dist));
	v3 n = get_normal(sc, p, hit.dist, t);
	uint32_t dark = 0;
	v3 lin = get_light(sc, ro, p, n, hit.id, t, probe, &dark, per_light);
	if (dark_out) *dark_out = dark;
	v3 c = v3pow(lin, 1.f / 2.2f);
	uint32_t px = pack_xrgb(c);

	if (missed) *missed = hit.id == 0;
	if (hit_id) *hit_id = hit.id;
	if (rgb_out) { rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z; }
	if (probe) {
		probe->rd[0] = rd.x; probe->rd[1] = rd.y; probe->rd[2] = rd.z;
		probe->hit_dist = hit.dist;
		probe->hit_id = hit.id;
		probe->march_steps = (uint32_t)march;
		probe->normal[0] = n.x; probe->normal[1] = n.y; probe->normal[2] = n.z;
		probe->rgb_linear[0] = lin.x; probe->rgb_linear[1] = lin.y; probe->rgb_linear[2] = lin.z;
		probe->rgb[0] = c.x; probe->rgb[1] = c.y; probe->rgb[2] = c.z;
		probe->xrgb = px;
	}
	return px;
}

static void render_row(const lol_scene* sc, const lol_camera* cam, int w, int h, int max_steps, int y,
                       void* xrgb, size_t pitch, float* rgb, uint16_t* steps,
                       lol_oracle_counters* ctr) {
	float fw = (float)w, fh = (float)h;
	for (int x = 0; x < w; x++) {
		struct tally t = { 0, 0, 0, 0, 0, 0 };
		int missed = 0;
		uint32_t hit_id = 0, dark = 0;
		uint16_t per_light[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		uint32_t px = shade_pixel(sc, cam, x, y, fw, fh, max_steps, &t,
		                          rgb ? rgb + ((size_t)y * w + x) * 3 : NULL, NULL, &missed, &hit_id, &dark,
		                          per_light);
		if (xrgb) memcpy((char*)xrgb + (size_t)y * pitch + (size_t)x * 4, &px, 4);
		if (steps) {
			uint16_t* o = steps + ((size_t)y * w + x) * 12;
			o[0] = (uint16_t)t.march_steps;
			o[1] = (uint16_t)(t.shadow_steps > 65535 ? 65535 : t.shadow_steps);
			o[2] = (uint16_t)(hit_id > 65535 ? 65535 : hit_id);
			o[3] = (uint16_t)dark;             /* bit i: light i has diffuse incidence exactly 0 here */
			for (int k = 0; k < 8; k++) o[4 + k] = per_light[k];
		}
		if (ctr) {
			ctr->pixels++;
			ctr->sdf_evals += t.sdf_evals;
			ctr->node_evals += t.node_evals;
			ctr->march_steps += t.march_steps;
			ctr->shadow_steps += t.shadow_steps;
			ctr->settle_violations += t.settle_violations;
			ctr->miss_pixels += (uint64_t)missed;
		}
	}
}

void lol_oracle_render_rows(const lol_scene* sc, const lol_camera* cam, int w, int h, int max_steps,
                            int y0, int y1, void* xrgb, size_t pitch, float* rgb, uint16_t* steps,
                            lol_oracle_counters* ctr) {
	for (int y = y0; y < y1; y++)
		render_row(sc, cam, w, h, max_steps, y, xrgb, pitch, rgb, steps, ctr);
}

/* ---- threaded frame: rows claimed from one atomic counter (naive_renderer.c:216) ---- */

struct job {
	const lol_scene* sc;
	const lol_camera* cam;
	int w, h, max_steps, y0, y_end, stride;
	void* xrgb;
	size_t pitch;
	float* rgb;
	atomic_int next;              /* current_line */
	pthread_mutex_t lock;
	lol_oracle_counters* ctr;
};

static void* worker(void* arg) {
	struct job* j = arg;
	lol_oracle_counters local;
	memset(&local, 0, sizeof local);
	for (;;) {
		int k = atomic_fetch_add(&j->next, 1);
		long y = (long)j->y0 + (long)k * j->stride;
		if (y >= j->y_end) break;
		render_row(j->sc, j->cam, j->w, j->h, j->max_steps, (int)y, j->xrgb, j->pitch, j->rgb, NULL, &local);
	}
	if (j->ctr) {
		pthread_mutex_lock(&j->lock);
		j->ctr->pixels += local.pixels;
		j->ctr->sdf_evals += local.sdf_evals;
		j->ctr->node_evals += local.node_evals;
		j->ctr->march_steps += local.march_steps;
		j->ctr->shadow_steps += local.shadow_steps;
		j->ctr->settle_violations += local.settle_violations;
		j->ctr->miss_pixels += local.miss_pixels;
		pthread_mutex_unlock(&j->lock);
	}
	return NULL;
}

static void run_job(struct job* j, int threads) {
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	atomic_init(&j->next, 0);
	pthread_mutex_init(&j->lock, NULL);
	pthread_t tid[256];
	int started = 0;
	for (int i = 1; i < threads; i++)
		if (pthread_create(&tid[started], NULL, worker, j) == 0) started++;
	worker(j);
	for (int i = 0; i < started; i++) pthread_join(tid[i], NULL);
	pthread_mutex_destroy(&j->lock);
}

void lol_oracle_render_frame(const lol_scene* sc, const lol_camera* cam, int w, int h, int max_steps,
                             int threads, void* xrgb, size_t pitch, float* rgb,
                             lol_oracle_counters* ctr) {
	struct job j = { .sc = sc, .cam = cam, .w = w, .h = h, .max_steps = max_steps,
	                 .y0 = 0, .y_end = h, .stride = 1, .xrgb = xrgb, .pitch = pitch,
	                 .rgb = rgb, .ctr = ctr };
	run_job(&j, threads);
}

void lol_oracle_render_sample(const lol_scene* sc, const lol_camera* cam, int w, int h, int max_steps,
                              int threads, int y0, int y_end, int stride, void* xrgb, size_t pitch,
                              lol_oracle_counters* ctr) {
	if (stride < 1) stride = 1;
	if (y_end > h) y_end = h;
	struct job j = { .sc = sc, .cam = cam, .w = w, .h = h, .max_steps = max_steps,
	                 .y0 = y0, .y_end = y_end, .stride = stride, .xrgb = xrgb, .pitch = pitch,
	                 .rgb = NULL, .ctr = ctr };
	run_job(&j, threads);
}

void lol_oracle_probe_pixel(const lol_scene* sc, const lol_camera* cam, int w, int h, int max_steps,
                            int x, int y, lol_oracle_probe* out) {
	struct tally t = { 0, 0, 0, 0, 0, 0 };
	memset(out, 0, sizeof *out);
	shade_pixel(sc, cam, x, y, (float)w, (float)h, max_steps, &t, NULL, out, NULL, NULL, NULL, NULL);
}

float lol_oracle_sdf(const lol_scene* sc, float px, float py, float pz, uint32_t* id) {
	struct tally t = { 0, 0, 0, 0, 0, 0 };
	struct world_dist d = sdf(sc, (v3){ px, py, pz }, &t);
	if (id) *id = d.id;
	return d.dist;
}

/* ------------------------------------------------ exported primitives / hash */

float lol_oracle_minf(float a, float b) { return minf(a, b); }
float lol_oracle_maxf(float a, float b) { return maxf(a, b); }
float lol_oracle_clamp(float v, float lo, float hi) { return clampf(v, lo, hi); }
float lol_oracle_sminf(float a, float b, float k) { return sminf(a, b, k); }
float lol_oracle_v3dot(const float a[3], const float b[3]) {
	return v3dot((v3){ a[0], a[1], a[2] }, (v3){ b[0], b[1], b[2] });
}
float lol_oracle_v3len(const float a[3]) { return v3len((v3){ a[0], a[1], a[2] }); }
void lol_oracle_v3normalize(const float a[3], float out[3]) {
	v3 r = v3normalize((v3){ a[0], a[1], a[2] });
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void lol_oracle_v3cross(const float a[3], const float b[3], float out[3]) {
	v3 r = v3cross((v3){ a[0], a[1], a[2] }, (v3){ b[0], b[1], b[2] });
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void lol_oracle_v3clamp(const float a[3], float lo, float hi, float out[3]) {
	v3 r = v3clamp((v3){ a[0], a[1], a[2] }, lo, hi);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float lol_oracle_sd_sphere(const float p[3], float r) { return sdSphere((v3){ p[0], p[1], p[2] }, r); }
float lol_oracle_sd_round_box(const float p[3], const float b[3], float r) {
	return sdRoundBox((v3){ p[0], p[1], p[2] }, (v3){ b[0], b[1], b[2] }, r);
}

/* libm's powf on arrays: the reference for the device's powf restatement (tests/test_gpu_powf.py) */
void lol_oracle_powf_batch(const float* x, const float* y, float* out, size_t n) {
	for (size_t i = 0; i < n; i++) out[i] = powf(x[i], y[i]);
}

uint64_t lol_oracle_hash_xrgb(const void* xrgb, int w, int h, size_t pitch) {
	uint64_t hsh = 0xcbf29ce484222325ull;
	for (int y = 0; y < h; y++) {
		const unsigned char* row = (const unsigned char*)xrgb + (size_t)y * pitch;
		for (int x = 0; x < w; x++) {
			uint32_t px;
			memcpy(&px, row + (size_t)x * 4, 4);
			hsh ^= px;
			hsh *= 0x100000001b3ull;
		}
	}
	return hsh;
}
