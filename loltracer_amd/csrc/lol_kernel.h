/*
 * lol_kernel.h — device code of the gfx950 sphere-tracer (one lane per primary ray).
 *
 * One kernel does the whole per-pixel pipeline of naive_renderer.c:217-235:
 * camera ray → sphere trace → 4-tap normal → per-light soft shadow + Phong →
 * gamma → XRGB8888.  Arithmetic is IEEE binary32 in the reference's operation
 * order (compile with -ffp-contract=off: the reference has no FMA), with
 * correctly rounded '/' and sqrt (hipcc's default), so every loop exit
 * (naive_renderer.c:61,85) is taken on the same iteration as on the CPU; only
 * powf (colour, never control flow) may differ from glibc by an ulp.
 *
 * Shape of the code on a 64-wide wavefront:
 *  - control flow is wave-uniform: the march / shadow loops run while
 *    __ballot(alive) != 0 and lanes that have hit or escaped keep their state
 *    by predication, so the SDF program is interpreted with scalar branches;
 *  - the SDF program, lights and materials are staged once per block into LDS
 *    and read back with wave-uniform (broadcast) addresses;
 *  - the operand stack of the SDF program lives in registers (fixed depth,
 *    push = shift), never in scratch;
 *  - pixels are written through an LDS tile so each wave stores whole 128-byte
 *    row segments.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lol_scene.h"

namespace lol {

constexpr int TILE_W = 32;            /* pixels per block row  */
constexpr int TILE_H = 8;             /* pixel rows per block  */
constexpr int BLOCK  = TILE_W * TILE_H;   /* 256 threads = 4 waves; wave k owns columns [8k, 8k+8) */

constexpr int OP_DWORDS       = sizeof(lol_op) / 4;        /* 10 */
constexpr int LIGHT_DWORDS    = sizeof(lol_light) / 4;     /*  9 */
constexpr int MATERIAL_DWORDS = sizeof(lol_material) / 4;  /* 10 */

/* Kernel arguments: by value, so they arrive in SGPRs. */
struct Launch {
	lol_frame_camera cam;
	float    fw, fh;              /* (float)w, (float)h */
	int32_t  w, h;
	int32_t  max_steps;
	int32_t  n_rows;              /* local rows this launch renders */
	int32_t  band_rows, n_parts, part;
	uint32_t n_ops, n_lights, n_materials, n_roots;
	const lol_program* prog;      /* device copy of the flattened scene */
	uint32_t* dst;                /* XRGB8888, pitch_px dwords per local row */
	uint32_t pitch_px;
	float*    dbg_rgb;
	float*    dbg_hit_dist;
	uint32_t* dbg_hit_id;
	uint32_t* dbg_steps;
};

struct V3 { float x, y, z; };

/* ---- float.h / vec.h semantics (see oracle/lol_oracle.c for the citations) ---- */
__device__ __forceinline__ float minf_(float a, float b) { return a < b ? a : b; }   /* MINSS: b on NaN/equal */
__device__ __forceinline__ float maxf_(float a, float b) { return a > b ? a : b; }   /* MAXSS */
__device__ __forceinline__ float clampf_(float v, float lo, float hi) { return minf_(maxf_(v, lo), hi); }

__device__ __forceinline__ V3 add(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
__device__ __forceinline__ V3 scale(V3 v, float f) { return { v.x * f, v.y * f, v.z * f }; }
__device__ __forceinline__ float dot(V3 a, V3 b) {
	float lo = a.x * b.x + a.y * b.y;
	float hi = a.z * b.z + 0.0f;              /* DPPS 0x71: the masked lane adds +0 */
	return lo + hi;
}
/* |v|^2 of a real vector: products are >= +0 or NaN, so the "+0" is a no-op */
__device__ __forceinline__ float len2(V3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
__device__ __forceinline__ float len(V3 a) { return __builtin_sqrtf(len2(a)); }
__device__ __forceinline__ V3 normalize(V3 v) { return scale(v, 1.0f / len(v)); }
__device__ __forceinline__ V3 from(lol_v3 v) { return { v.x, v.y, v.z }; }

/* sminf, float.h:29-33 */
__device__ __forceinline__ float sminf_(float a, float b, float k) {
	float h = clampf_(.5f + .5f * (b - a) / k, 0.f, 1.f);
	return (b + (a - b) * h) - k * h * (1.f - h);
}

/* ------------------------------------------------------------ SDF interpreter
 * Runs the post-order program (lol_scene.h) for one point per lane.  `ops`
 * points into LDS; every lane reads the same address, and the opcode is moved
 * to an SGPR so the switch is a scalar branch.  STACK is the register stack
 * depth (>= program max_stack). */
template <int STACK>
struct Interp {
	const uint32_t* ops;      /* LDS */
	uint32_t        n_ops;

	__device__ __forceinline__ void eval(V3 p, float& best, uint32_t& best_id) const {
		float s[STACK];
#pragma unroll
		for (int i = 0; i < STACK; i++) s[i] = 0.f;
		best = __builtin_inff();
		best_id = 0;
		for (uint32_t i = 0; i < n_ops; i++) {
			const uint32_t* o = ops + i * OP_DWORDS;
			const float* f = reinterpret_cast<const float*>(o + 2);
			uint32_t op = __builtin_amdgcn_readfirstlane(o[0]);
			if (op <= LOL_OP_PLANE) {
				float d;
				if (op == LOL_OP_SPHERE) {                       /* sdSphere(p - c, r), sdf.h:8-10 */
					V3 q = { p.x - f[0], p.y - f[1], p.z - f[2] };
					d = len(q) - f[3];
				} else if (op == LOL_OP_RBOX) {                  /* sdRoundBox, sdf.h:18-22 */
					V3 q = { __builtin_fabsf(p.x - f[0]) - f[3],
					         __builtin_fabsf(p.y - f[1]) - f[4],
					         __builtin_fabsf(p.z - f[2]) - f[5] };
					V3 cq = { maxf_(q.x, 0.f), maxf_(q.y, 0.f), maxf_(q.z, 0.f) };
					d = len(cq) + minf_(maxf_(q.x, maxf_(q.y, q.z)), 0.f) - f[6];
				} else {                                         /* plane: (p - (0,y,0)).y */
					d = p.y - f[0];
				}
#pragma unroll
				for (int j = STACK - 1; j > 0; j--) s[j] = s[j - 1];
				s[0] = d;
			} else if (op == LOL_OP_TOP) {                       /* sdf(): strict '<', naive_renderer.c:39 */
				uint32_t id = __builtin_amdgcn_readfirstlane(o[1]);
				if (s[0] < best) { best = s[0]; best_id = id; }
#pragma unroll
				for (int j = 0; j < STACK - 1; j++) s[j] = s[j + 1];
			} else {                                             /* SMIN: top is b; SMIN_R: top is a */
				float top = s[0], under = s[1];
				float a = op == LOL_OP_SMIN ? under : top;
				float b = op == LOL_OP_SMIN ? top : under;
				s[0] = sminf_(a, b, f[0]);
#pragma unroll
				for (int j = 1; j < STACK - 1; j++) s[j] = s[j + 1];
			}
		}
	}
};

/* --------------------------------------------------------------- the pipeline */

struct Hit { float dist; uint32_t id; uint32_t steps; };

/* get_intersection, naive_renderer.c:48-69 */
template <class Sdf>
__device__ __forceinline__ Hit march(const Sdf& sdf, V3 ro, V3 rd, int max_steps) {
	const float EPSILON = 0.001f, MAX_DIST = 100.f;
	float dist = 0.f;
	uint32_t id = 0, steps = 0;
	bool alive = true;
	for (int i = 0; i < max_steps; i++) {
		if (__ballot(alive) == 0) break;              /* every lane has hit or escaped */
		V3 p = add(ro, scale(rd, dist));
		float d; uint32_t did;
		sdf.eval(p, d, did);
		if (alive) {
			dist += d;
			id = did;
			steps++;
			if (d < EPSILON || dist > MAX_DIST) alive = false;
		}
	}
	if (dist >= MAX_DIST) id = 0;
	return { dist, id, steps };
}

/* in_shadow + softshadow, naive_renderer.c:73-100.  dir/light_dist come from the caller,
 * which needs the same normalize(light - p) for the Phong term. */
template <class Sdf>
__device__ __forceinline__ float soft_shadow(const Sdf& sdf, V3 p, V3 dir, float max_dist, uint32_t& steps) {
	V3 ro = add(p, dir);
	float res = 1.f, t = 0.f;
	bool alive = true;
	for (int i = 0; i < 128; i++) {
		if (__ballot(alive) == 0) break;
		V3 q = add(ro, scale(dir, t));
		float s; uint32_t sid;
		sdf.eval(q, s, sid);
		if (alive) {
			res = minf_(res, 50.f * s / t);
			t += s;
			steps++;
			if (res < -1.f || t > max_dist) alive = false;
		}
	}
	return maxf_(res, 0.f);
}

/* get_normal, naive_renderer.c:114-125: k0=(1,-1,-1) k1=(-1,-1,1) k2=(-1,1,-1) k3=(1,1,1) */
template <class Sdf>
__device__ __forceinline__ V3 normal_at(const Sdf& sdf, V3 p, float dist) {
	const float h = dist / 100.f;
	const float nh = -1.f * h;       /* v3scale(k, h) multiplies; -1*h == -h bit for bit */
	float s0, s1, s2, s3; uint32_t unused;
	sdf.eval({ p.x + h,  p.y + nh, p.z + nh }, s0, unused);
	sdf.eval({ p.x + nh, p.y + nh, p.z + h  }, s1, unused);
	sdf.eval({ p.x + nh, p.y + h,  p.z + nh }, s2, unused);
	sdf.eval({ p.x + h,  p.y + h,  p.z + h  }, s3, unused);
	V3 p0 = {  s0, -s0, -s0 }, p1 = { -s1, -s1,  s1 }, p2 = { -s2,  s2, -s2 }, p3 = { s3, s3, s3 };
	return normalize(add(p0, add(p1, add(p2, p3))));
}

__device__ __forceinline__ V3 lds_v3(const uint32_t* base) {
	const float* f = reinterpret_cast<const float*>(base);
	return { f[0], f[1], f[2] };
}

/*
 * Block layout in LDS (dwords): ops | lights | materials | root_material | ambient(3) | out tile
 */
template <int STACK>
__global__ __launch_bounds__(BLOCK)
void render_kernel(const Launch L) {
	extern __shared__ uint32_t lds[];
	uint32_t* l_ops   = lds;
	uint32_t* l_light = l_ops + L.n_ops * OP_DWORDS;
	uint32_t* l_mat   = l_light + L.n_lights * LIGHT_DWORDS;
	uint32_t* l_rootm = l_mat + L.n_materials * MATERIAL_DWORDS;
	uint32_t* l_amb   = l_rootm + L.n_roots;
	uint32_t* l_tile  = l_amb + 3;

	/* stage the scene once per block */
	{
		const uint32_t* g_ops   = reinterpret_cast<const uint32_t*>(L.prog->ops);
		const uint32_t* g_light = reinterpret_cast<const uint32_t*>(L.prog->lights);
		const uint32_t* g_mat   = reinterpret_cast<const uint32_t*>(L.prog->materials);
		for (uint32_t i = threadIdx.x; i < L.n_ops * OP_DWORDS; i += BLOCK) l_ops[i] = g_ops[i];
		for (uint32_t i = threadIdx.x; i < L.n_lights * LIGHT_DWORDS; i += BLOCK) l_light[i] = g_light[i];
		for (uint32_t i = threadIdx.x; i < L.n_materials * MATERIAL_DWORDS; i += BLOCK) l_mat[i] = g_mat[i];
		for (uint32_t i = threadIdx.x; i < L.n_roots; i += BLOCK) l_rootm[i] = L.prog->root_material[i];
		if (threadIdx.x < 3) l_amb[threadIdx.x] = reinterpret_cast<const uint32_t*>(&L.prog->ambient_color)[threadIdx.x];
	}
	__syncthreads();

	/* lane → pixel: wave k covers an 8x8 patch at columns 8k.. of the 32x8 tile */
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tx = wave * 8 + (lane & 7), ty = lane >> 3;
	int x = blockIdx.x * TILE_W + tx;
	int r = blockIdx.y * TILE_H + ty;                     /* local row */
	/* out-of-frame lanes shade a clamped pixel and skip the store: keeps the wave uniform */
	x = x < L.w ? x : L.w - 1;
	r = r < L.n_rows ? r : L.n_rows - 1;
	const int band = r / L.band_rows;
	const int y = (band * L.n_parts + L.part) * L.band_rows + (r - band * L.band_rows);

	/* naive_renderer.c:218-221 */
	const float vx = ((float)x + .5f) / L.fw * 2.f - 1.f;
	const float vy = 1.f - ((float)y + .5f) / L.fh * 2.f;

	/* get_camera_ray with the per-frame basis hoisted, naive_renderer.c:188-190 */
	const V3 ro = from(L.cam.origin), cdir = from(L.cam.dir);
	V3 rd = add(scale(from(L.cam.right), vx * L.cam.width), scale(from(L.cam.up), vy * L.cam.height));
	rd = normalize(add(rd, cdir));

	Interp<STACK> sdf{ l_ops, L.n_ops };

	Hit hit = march(sdf, ro, rd, L.max_steps);
	V3 p = add(ro, scale(rd, hit.dist));
	V3 n = normal_at(sdf, p, hit.dist);

	/* get_material, naive_renderer.c:103-112 (per-lane table lookups) */
	uint32_t mid = hit.id ? l_rootm[hit.id - 1] : 0u;
	const float* m = reinterpret_cast<const float*>(l_mat + mid * MATERIAL_DWORDS);
	const float shininess = m[0];
	const V3 m_diff = { m[1], m[2], m[3] }, m_spec = { m[4], m[5], m[6] }, m_amb = { m[7], m[8], m[9] };

	/* get_light, naive_renderer.c:129-175 */
	V3 total = { 0.f, 0.f, 0.f };
	uint32_t shadow_steps = 0;
	const V3 camera_dir = normalize(sub(ro, p));
	for (uint32_t li = 0; li < L.n_lights; li++) {
		const uint32_t* lp = l_light + li * LIGHT_DWORDS;
		V3 to_light = sub(lds_v3(lp), p);
		float light_dist = len(to_light);
		V3 light_dir = scale(to_light, 1.0f / light_dist);      /* == v3normalize(light - p) */
		float shadow = soft_shadow(sdf, p, light_dir, light_dist, shadow_steps);

		V3 refl = sub(scale(n, 2.f * dot(light_dir, n)), light_dir);
		float di = clampf_(dot(n, light_dir), 0.f, 1.f);
		V3 Id = mul(scale(lds_v3(lp + 3), shadow * di), m_diff);
		total = add(total, Id);
		float si = di * powf(clampf_(dot(refl, camera_dir), 0.f, 1.f), shininess);
		V3 Is = mul(scale(lds_v3(lp + 6), shadow * si), m_spec);
		total = add(total, Is);
	}
	total = add(total, mul(lds_v3(l_amb), m_amb));
	/* v3clamp: max(min(v, 1), 0) — NaN → 1 (vec.h:63-65) */
	V3 c = { maxf_(minf_(total.x, 1.f), 0.f), maxf_(minf_(total.y, 1.f), 0.f), maxf_(minf_(total.z, 1.f), 0.f) };

	/* gamma + colorf_to_pixfmt, naive_renderer.c:231-232, renderer.h:17-22 */
	const float g = 1.f / 2.2f;
	c = { powf(c.x, g), powf(c.y, g), powf(c.z, g) };
	uint32_t px = ((uint32_t)(c.x * 255.f) & 0xFFu) << 16 | ((uint32_t)(c.y * 255.f) & 0xFFu) << 8 |
	              ((uint32_t)(c.z * 255.f) & 0xFFu);

	const int gx = blockIdx.x * TILE_W + tx, gr = blockIdx.y * TILE_H + ty;
	const bool inside = gx < L.w && gr < L.n_rows;
	if (inside) {
		size_t o = (size_t)gr * L.w + gx;
		if (L.dbg_rgb) { L.dbg_rgb[o * 3 + 0] = c.x; L.dbg_rgb[o * 3 + 1] = c.y; L.dbg_rgb[o * 3 + 2] = c.z; }
		if (L.dbg_hit_dist) L.dbg_hit_dist[o] = hit.dist;
		if (L.dbg_hit_id) L.dbg_hit_id[o] = hit.id;
		if (L.dbg_steps) L.dbg_steps[o] = (hit.steps & 0xFFFFu) | (shadow_steps << 16);
	}

	/* through LDS so a wave stores two full 128-byte row segments */
	l_tile[ty * TILE_W + tx] = px;
	__syncthreads();
	{
		const int sx = threadIdx.x & (TILE_W - 1), sy = threadIdx.x >> 5;
		const int ox = blockIdx.x * TILE_W + sx, orow = blockIdx.y * TILE_H + sy;
		if (ox < L.w && orow < L.n_rows)
			L.dst[(size_t)orow * L.pitch_px + ox] = l_tile[sy * TILE_W + sx];
	}
}

__host__ inline size_t lds_bytes(const Launch& L) {
	return (size_t)(L.n_ops * OP_DWORDS + L.n_lights * LIGHT_DWORDS + L.n_materials * MATERIAL_DWORDS +
	                L.n_roots + 3 + TILE_W * TILE_H) * 4;
}

}  // namespace lol
