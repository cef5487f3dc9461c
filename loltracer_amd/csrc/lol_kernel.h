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
 * The file is compiled twice:
 *  - ahead of time by hipcc (lol_gpu.hip) with Interp<STACK, KIND>: the flattened SDF
 *    program is staged once per block into LDS and interpreted with
 *    wave-uniform scalar branches and a register operand stack;
 *  - at render_prepare time by hipRTC (lol_gpu.hip: specialise()) together with
 *    a generated `SpecSdf` whose eval() is the scene's SDF as straight-line
 *    code with every constant an immediate — the GPU counterpart of the
 *    reference's tracing JIT (tracing_jit_renderer.dasc:76-216).
 * Both use the same pipeline below, so they produce the same bits.
 *
 * Shape of the code on a 64-wide wavefront:
 *  - control flow is wave-uniform: the march / shadow loops run while
 *    __ballot(alive) != 0 and lanes that have hit or escaped keep their state
 *    by predication;
 *  - lights and materials are staged once per block into LDS and read back
 *    with wave-uniform (lights) or per-lane (material of the hit) addresses;
 *  - pixels are written through an LDS tile so each wave stores whole row
 *    segments (4 rows x 64 bytes with the default 16x4 patch).
 */
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace lol {

typedef unsigned int u32;
typedef int i32;

/* Pixel footprint: each wave owns a WAVE_W x WAVE_H patch (64 pixels); a block is WAVES_X patches side by side.
 * Default 16x4 per wave, one wave per block: single-wave blocks free their slot as soon as their own rays are
 * done (pixels differ 100x in cost, so a 4-wave block often waits for one straggler) — measured +4 % over
 * 8x8 patches in 4-wave blocks; 16x4 keeps 64-byte row segments for the framebuffer stores.  The specialised
 * kernel can be compiled with other shapes for experiments (LOL_GPU_WAVE_SHAPE=WxHxN, lol_gpu.hip). */
#ifndef LOL_WAVE_W
#define LOL_WAVE_W 16
#endif
#ifndef LOL_WAVE_H
#define LOL_WAVE_H 4
#endif
#ifndef LOL_WAVES_X
#define LOL_WAVES_X 1
#endif
static_assert(LOL_WAVE_W * LOL_WAVE_H == 64, "a wave shades 64 pixels");
constexpr int WAVE_W = LOL_WAVE_W, WAVE_H = LOL_WAVE_H;
constexpr int TILE_W = WAVE_W * LOL_WAVES_X;   /* pixels per block row  */
constexpr int TILE_H = WAVE_H;                 /* pixel rows per block  */
constexpr int BLOCK  = TILE_W * TILE_H;        /* 64 * WAVES_X threads; wave k owns columns [WAVE_W*k, WAVE_W*(k+1)) */

/* dword layouts of lol_op / lol_light / lol_material (lol_scene.h); checked by static_asserts in lol_gpu.hip */
constexpr int OP_DWORDS = 10, LIGHT_DWORDS = 9, MATERIAL_DWORDS = 10;
enum { OP_SPHERE = 0, OP_RBOX = 1, OP_PLANE = 2, OP_SMIN = 3, OP_SMIN_R = 4, OP_TOP = 5,
       OP_SMINF = 6, OP_SMINF_R = 7 };   /* device-only: smooth min with a proven fast blend factor */

/* Launch.flags */
constexpr u32 FLAG_MISS_SKIP = 1u;   /* a wave whose rays all escaped may skip normal + lights (see shade_pixel) */
constexpr u32 FLAG_DARK_SKIP = 2u;   /* lanes whose diffuse incidence for a light is exactly 0 need no shadow march for it */

/* = lol_frame_camera */
struct Cam { float origin[3], dir[3], right[3], up[3]; float width, height; };

/* Kernel arguments: by value, so they arrive in SGPRs. */
struct Launch {
	Cam    cam;
	float  fw, fh;               /* (float)w, (float)h */
	i32    w, h;
	i32    max_steps;
	i32    n_rows;               /* local rows this launch renders */
	i32    band_rows, n_parts, part;
	u32    n_ops, n_lights, n_materials, n_roots;
	const u32* ops;              /* device copies of the flattened scene's tables */
	const u32* lights;
	const u32* materials;
	const u32* root_material;
	float  ambient[3];
	u32    flags;                /* FLAG_* */
	u32*   dst;                  /* XRGB8888, pitch_px dwords per local row */
	u32    pitch_px;
	float* dbg_rgb;
	float* dbg_hit_dist;
	u32*   dbg_hit_id;
	u32*   dbg_steps;
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
__device__ __forceinline__ V3 v3(const float* f) { return { f[0], f[1], f[2] }; }

/* sminf, float.h:29-33 */
__device__ __forceinline__ float sminf_(float a, float b, float k) {
	float h = clampf_(.5f + .5f * (b - a) / k, 0.f, 1.f);
	return (b + (a - b) * h) - k * h * (1.f - h);
}
/* sdSphere(p - c, r), sdf.h:8-10 */
__device__ __forceinline__ float sd_sphere(V3 p, float cx, float cy, float cz, float r) {
	V3 q = { p.x - cx, p.y - cy, p.z - cz };
	return len(q) - r;
}
/* sdRoundBox(p - c, b, r), sdf.h:18-22 */
__device__ __forceinline__ float sd_round_box(V3 p, float cx, float cy, float cz, float bx, float by, float bz, float r) {
	V3 q = { __builtin_fabsf(p.x - cx) - bx, __builtin_fabsf(p.y - cy) - by, __builtin_fabsf(p.z - cz) - bz };
	V3 cq = { maxf_(q.x, 0.f), maxf_(q.y, 0.f), maxf_(q.z, 0.f) };
	return len(cq) + minf_(maxf_(q.x, maxf_(q.y, q.z)), 0.f) - r;
}

/* ---------------------------------------------------------------- fast exact paths
 * Used only by the hipRTC-specialised kernel, and only after lol_gpu.hip has PROVED them
 * equal to the plain expressions by running every one of the 2^32 float inputs through
 * both on the device (verify_sqrt_kernel / verify_div_kernel): same bits or both NaN.
 *
 * Measured issue costs on MI355X (tools/valu_rate.hip, 8 waves/SIMD): v_mul/v_add/v_fma ~2.3
 * cycles per wave-instruction, v_sqrt / v_rsq / v_rcp ~8.2 and not overlapped with anything —
 * so time ~ 2 x instructions + 6 x transcendentals, and hipcc's 16-instruction sqrt and
 * 11-instruction division are the first things to shrink.
 *
 * sqrt_gs: Goldschmidt iteration from v_rsq_f32 (what hipcc itself emits for a correctly rounded
 * sqrt when it need not handle denormals): 1 transcendental + 7 plain ops, no compares.  Proven
 * equal to sqrtf for 2^-96 <= x < inf and NaN.  x == 0 and x == inf give NaN here (0*inf), and
 * v_rsq_f32 flushes denormal inputs, so callers track the range of its arguments (Range) and a
 * wave that saw one outside [SQRT_FAST_MIN, inf) shades its pixels again through the plain path.
 * Its only argument is a sum of squares, never negative. */
constexpr float SQRT_FAST_MIN = 0x1p-96f;
constexpr u32 SQRT_FAST_MIN_BITS = 0x0f800000u;      /* 2^-96 */
constexpr u32 F32_INF_BITS = 0x7f800000u;
__device__ __forceinline__ float sqrt_gs(float x) {
	float y = __builtin_amdgcn_rsqf(x);
	float s = x * y;
	float h = y * 0.5f;
	float e = __builtin_fmaf(-h, s, 0.5f);
	h = __builtin_fmaf(h, e, h);
	s = __builtin_fmaf(s, e, s);
	float d = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(d, h, s);
}
/* sqrt_r2: the same without the first refinement of (s, h) — v_rsq_f32, 2 mul, 2 fma.  An exhaustive run
 * (tools/sqrt_search.hip) found it correctly rounded on the whole domain as well; preferred when it verifies. */
__device__ __forceinline__ float sqrt_r2(float x) {
	float y = __builtin_amdgcn_rsqf(x);
	float s = x * y;
	float h = y * 0.5f;
	float d = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(d, h, s);
}
/* sqrt_pm: v_sqrt_f32 (1 ulp) + the +-1 ulp residual test of hipcc's denormal-safe expansion, minus
 * its input scaling and class fix-up.  Proven for x == 0, 2^-96 <= x <= inf and NaN.  Fallback when
 * sqrt_gs does not verify on a device. */
__device__ __forceinline__ float sqrt_pm(float x) {
	float r = __builtin_amdgcn_sqrtf(x);
	int ri = __builtin_bit_cast(int, r);
	float rm = __builtin_bit_cast(float, ri - 1), rp = __builtin_bit_cast(float, ri + 1);
	float em = __builtin_fmaf(-rm, r, x);
	float ep = __builtin_fmaf(-rp, r, x);
	r = em <= 0.f ? rm : r;
	r = ep > 0.f ? rp : r;
	return r;
}
/* Unsigned min / max of the bit patterns of every squared length an evaluation took the root of:
 * for non-negative floats the bit order is the value order (NaN sorts above inf). */
struct Range {
	u32 lo = F32_INF_BITS, hi = 0u;
	__device__ __forceinline__ void see(float l2) {
		u32 b = __builtin_bit_cast(u32, l2);
		lo = b < lo ? b : lo;
		hi = b > hi ? b : hi;
	}
	__device__ __forceinline__ bool outside() const { return lo < SQRT_FAST_MIN_BITS || hi >= F32_INF_BITS; }
};
template <int KIND> __device__ __forceinline__ float sqrt_fast(float x) {
	return KIND == 3 ? sqrt_r2(x) : KIND == 2 ? sqrt_gs(x) : sqrt_pm(x);
}

/* The smooth-min blend factor h = clamp(.5f + (.5f*(b-a))/k, 0, 1) (float.h:30) as a function of
 * dlt = b - a, for a scene constant k.  .5f*dlt is an exact scaling, so with k2 = 2k and hrk = .5f*(1/k):
 *   q = dlt*hrk;  r = fma(-q, k2, dlt);  q = fma(r, hrk, q);  q = div_fixup(q, k2, dlt)
 * is the product / one-residual-correction / hardware special-case fix-up form of (.5f*dlt)/k: 4
 * instructions instead of the 11 of a correctly rounded division (the .5f multiply folds away too).
 * smin_h_fast is what gets proven equal to smin_h_exact over all 2^32 values of dlt, per k. */
__device__ __forceinline__ float smin_h_exact(float dlt, float k) { return clampf_(.5f + .5f * dlt / k, 0.f, 1.f); }
__device__ __forceinline__ float smin_h_fast(float dlt, float k2, float hrk) {
	float q = dlt * hrk;
	float r = __builtin_fmaf(-q, k2, dlt);
	q = __builtin_fmaf(r, hrk, q);
	q = __builtin_amdgcn_div_fixupf(q, k2, dlt);
	return clampf_(.5f + q, 0.f, 1.f);
}
/* sminf (float.h:29-33) with the proven blend factor.  lerp(b, a, h) = b + (a - b)*h is written b - (b - a)*h:
 * a - b and b - a are exact negatives of each other unless a == b with equal signs, where h = .5 and the two
 * forms can only differ in the sign of a zero sum (a == b == -0) that the subtraction of k*h*(1-h) = k/4 != 0
 * then erases — lol_gpu.hip uses this form only for |k| >= 2^-100 so that k/4 is a non-zero normal number. */
__device__ __forceinline__ float sminf_fastdiv(float a, float b, float k, float k2, float hrk) {
	float dlt = b - a;
	float h = smin_h_fast(dlt, k2, hrk);
	return (b - dlt * h) - k * h * (1.f - h);
}

/* sd_sphere / sd_round_box on a proven fast sqrt (KIND 1 = sqrt_pm, 2 = sqrt_gs, 3 = sqrt_r2) */
template <int KIND>
__device__ __forceinline__ float sd_sphere_fast(V3 p, float cx, float cy, float cz, float r, Range& rg) {
	V3 q = { p.x - cx, p.y - cy, p.z - cz };
	float l2 = len2(q);
	rg.see(l2);
	return sqrt_fast<KIND>(l2) - r;
}
template <int KIND>
__device__ __forceinline__ float sd_round_box_fast(V3 p, float cx, float cy, float cz, float bx, float by, float bz, float r, Range& rg) {
	V3 q = { __builtin_fabsf(p.x - cx) - bx, __builtin_fabsf(p.y - cy) - by, __builtin_fabsf(p.z - cz) - bz };
	V3 cq = { maxf_(q.x, 0.f), maxf_(q.y, 0.f), maxf_(q.z, 0.f) };
	float l2 = len2(cq);
	/* inside the box l2 is exactly 0 and the root is 0: do not take it (nor flag it) there */
	float l2s = l2 == 0.f ? 1.f : l2;
	rg.see(l2s);
	float root = sqrt_fast<KIND>(l2s);
	root = l2 == 0.f ? 0.f : root;
	return root + minf_(maxf_(q.x, maxf_(q.y, q.z)), 0.f) - r;
}

/* ------------------------------------------------------------ SDF interpreter
 * Runs the post-order program (lol_scene.h) for one point per lane.  `ops`
 * points into LDS; every lane reads the same address, and the opcode is moved
 * to an SGPR so the switch is a scalar branch.  STACK is the register stack
 * depth (>= program max_stack); push = shift, so nothing goes to scratch. */
template <int STACK, int KIND = 0>
struct Interp {
	const u32* ops;      /* LDS */
	u32        n_ops;
	Range      rg;       /* KIND != 0: range of the squared lengths given to the proven fast sqrt (see above) */

	__device__ __forceinline__ void eval(V3 p, float& best, u32& best_id) {
		float s[STACK];
#pragma unroll
		for (int i = 0; i < STACK; i++) s[i] = 0.f;
		best = __builtin_inff();
		best_id = 0;
		for (u32 i = 0; i < n_ops; i++) {
			const u32* o = ops + i * OP_DWORDS;
			const float* f = reinterpret_cast<const float*>(o + 2);
			/* (prefetching op i+1 while op i executes was tried: more VGPRs, lower occupancy, 20 % slower) */
			const u32 w0 = o[0], w1 = o[1];
			u32 op = __builtin_amdgcn_readfirstlane(w0);
			if (op <= OP_PLANE) {
				float d;
				if (op == OP_SPHERE)
					d = KIND ? sd_sphere_fast<KIND ? KIND : 1>(p, f[0], f[1], f[2], f[3], rg)
					         : sd_sphere(p, f[0], f[1], f[2], f[3]);
				else if (op == OP_RBOX)
					d = KIND ? sd_round_box_fast<KIND ? KIND : 1>(p, f[0], f[1], f[2], f[3], f[4], f[5], f[6], rg)
					         : sd_round_box(p, f[0], f[1], f[2], f[3], f[4], f[5], f[6]);
				else
					d = p.y - f[0];                              /* plane: (p - (0,y,0)).y */
#pragma unroll
				for (int j = STACK - 1; j > 0; j--) s[j] = s[j - 1];
				s[0] = d;
			} else if (op == OP_TOP) {                           /* sdf(): strict '<', naive_renderer.c:39 */
				u32 id = __builtin_amdgcn_readfirstlane(w1);
				if (s[0] < best) { best = s[0]; best_id = id; }
#pragma unroll
				for (int j = 0; j < STACK - 1; j++) s[j] = s[j + 1];
			} else {                                             /* SMIN*: top is b; SMIN*_R: top is a */
				float top = s[0], under = s[1];
				const bool swapped = op == OP_SMIN_R || op == OP_SMINF_R;
				float a = swapped ? top : under;
				float b = swapped ? under : top;
				/* OP_SMINF*: lol_gpu.hip rewrote the op after proving the fast blend factor for this k: f = {k, 2k, .5/k} */
				s[0] = op >= OP_SMINF ? sminf_fastdiv(a, b, f[0], f[1], f[2]) : sminf_(a, b, f[0]);
#pragma unroll
				for (int j = 1; j < STACK - 1; j++) s[j] = s[j + 1];
			}
		}
	}
};

/* --------------------------------------------------------------- the pipeline */

struct Hit { float dist; u32 id; u32 steps; };

/* get_intersection, naive_renderer.c:48-69 */
template <class Sdf>
__device__ __forceinline__ Hit march(Sdf& sdf, V3 ro, V3 rd, int max_steps) {
	const float EPSILON = 0.001f, MAX_DIST = 100.f;
	float dist = 0.f;
	u32 id = 0, steps = 0;
	bool alive = true;
	for (int i = 0; i < max_steps; i++) {
		if (__ballot(alive) == 0) break;              /* every lane has hit or escaped */
		V3 p = add(ro, scale(rd, dist));
		float d; u32 did;
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
__device__ __forceinline__ float soft_shadow(Sdf& sdf, V3 p, V3 dir, float max_dist, u32& steps, bool needed) {
	V3 ro = add(p, dir);
	float res = 1.f, t = 0.f;
	bool alive = needed;          /* a lane that does not need the factor never marches (returns 1) */
	for (int i = 0; i < 128; i++) {
		if (__ballot(alive) == 0) break;
		V3 q = add(ro, scale(dir, t));
		float s; u32 sid;
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

/* get_normal, naive_renderer.c:114-125: k0=(1,-1,-1) k1=(-1,-1,1) k2=(-1,1,-1) k3=(1,1,1);
 * n = normalize(k0*s0 + (k1*s1 + (k2*s2 + k3*s3))), s_i = sdf(p + k_i*h).  The taps run as a rolled loop from
 * k3 down to k0 (one copy of the SDF code instead of four; + is commutative, so adding each new term on the
 * left of the running sum reproduces the reference's association). */
template <class Sdf>
__device__ __forceinline__ V3 normal_at(Sdf& sdf, V3 p, float dist) {
	const float h = dist / 100.f;
	const float nh = -1.f * h;       /* v3scale(k, h) multiplies; -1*h == -h bit for bit */
	V3 acc = { 0.f, 0.f, 0.f };
#pragma unroll 1
	for (int k = 3; k >= 0; k--) {
		/* sign pattern of tap k, wave-uniform: x is + for k0,k3; y is + for k2,k3; z is + for k1,k3 */
		const bool px = k == 0 || k == 3, py = k >= 2, pz = k == 1 || k == 3;
		float s; u32 unused;
		sdf.eval({ p.x + (px ? h : nh), p.y + (py ? h : nh), p.z + (pz ? h : nh) }, s, unused);
		const float ns = -s;          /* -1.f * s */
		V3 term = { px ? s : ns, py ? s : ns, pz ? s : ns };
		acc = k == 3 ? term : add(term, acc);
	}
	return normalize(acc);
}

__device__ __forceinline__ V3 lds_v3(const u32* base) {
	const float* f = reinterpret_cast<const float*>(base);
	return { f[0], f[1], f[2] };
}

/* dwords of LDS the pipeline needs besides what the Sdf policy stages itself */
__host__ __device__ inline u32 common_lds_dwords(u32 n_lights, u32 n_materials, u32 n_roots) {
	return n_lights * LIGHT_DWORDS + n_materials * MATERIAL_DWORDS + n_roots + TILE_W * TILE_H;
}

struct Pixel { u32 px; V3 rgb; Hit hit; u32 shadow_steps; };

/*
 * The per-pixel body, naive_renderer.c:217-235, for the pixel this lane owns.
 * `lds` = lights | materials | root_material | out tile (already staged and synchronised).
 */
template <class Sdf>
__device__ __forceinline__ Pixel shade_pixel(const Launch& L, Sdf& sdf, const u32* lds) {
	const u32* l_light = lds;
	const u32* l_mat   = l_light + L.n_lights * LIGHT_DWORDS;
	const u32* l_rootm = l_mat + L.n_materials * MATERIAL_DWORDS;

	/* lane → pixel: wave k covers a WAVE_W x WAVE_H patch of the block's tile */
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tx = wave * WAVE_W + (lane % WAVE_W), ty = lane / WAVE_W;
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
	const V3 ro = v3(L.cam.origin), cdir = v3(L.cam.dir);
	V3 rd = add(scale(v3(L.cam.right), vx * L.cam.width), scale(v3(L.cam.up), vy * L.cam.height));
	rd = normalize(add(rd, cdir));

	Hit hit = march(sdf, ro, rd, L.max_steps);

	/* get_material, naive_renderer.c:103-112 (per-lane table lookups) */
	u32 mid = hit.id ? l_rootm[hit.id - 1] : 0u;
	const float* m = reinterpret_cast<const float*>(l_mat + mid * MATERIAL_DWORDS);
	const float shininess = m[0];
	const V3 m_diff = { m[1], m[2], m[3] }, m_spec = { m[4], m[5], m[6] }, m_amb = { m[7], m[8], m[9] };

	/*
	 * A ray that escaped is shaded with material #0 (naive_renderer.c:103-112).  When the host has checked
	 * that material #0 has diffuse == specular == 0, shininess >= 0 and every light intensity is finite
	 * (lol_gpu.hip: miss_skip_ok), each light's two terms are (finite) * (+-0) = +-0 whatever the normal and
	 * the shadow factor turn out to be, the running sum stays +0, and get_light() returns exactly
	 * clamp(ambient_color * material.ambient).  So a wave in which EVERY lane escaped skips the 4 normal taps
	 * and the shadow marches (about 13 % of all SDF evaluations on scene4) and falls through to the same
	 * `0 + ambient*mat.ambient` expression; a wave with at least one hit runs everything for all its lanes.
	 */
	const bool lit = !((L.flags & FLAG_MISS_SKIP) && __ballot(hit.id != 0u) == 0);

	/* get_light, naive_renderer.c:129-175 */
	V3 total = { 0.f, 0.f, 0.f };
	u32 shadow_steps = 0;
	if (lit) {
		V3 p = add(ro, scale(rd, hit.dist));
		V3 n = normal_at(sdf, p, hit.dist);
		const V3 camera_dir = normalize(sub(ro, p));
		for (u32 li = 0; li < L.n_lights; li++) {
			const u32* lp = l_light + li * LIGHT_DWORDS;
			V3 to_light = sub(lds_v3(lp), p);
			float light_dist = len(to_light);
			V3 light_dir = scale(to_light, 1.0f / light_dist);      /* == v3normalize(light - p) */
			float di = clampf_(dot(n, light_dir), 0.f, 1.f);
			/*
			 * With di == 0 (surface faces away from the light; clamp also maps NaN to 0) both terms of this
			 * light are I * (shadow * 0) * colour = +-0 for ANY shadow factor in [0, 1], and adding +-0 never
			 * changes the running sum (it starts at +0).  The host sets FLAG_DARK_SKIP only when every light
			 * intensity and material colour is finite and every shininess >= 0 (so powf stays finite and
			 * 0 * powf is 0).  Such lanes — and escaped lanes under FLAG_MISS_SKIP — then sit the shadow march
			 * out; a wave with no lane left skips it.
			 */
			bool needed = true;
			if (L.flags & FLAG_DARK_SKIP) needed = di > 0.f;
			if ((L.flags & FLAG_MISS_SKIP) && hit.id == 0u) needed = false;
			float shadow = soft_shadow(sdf, p, light_dir, light_dist, shadow_steps, needed);

			V3 refl = sub(scale(n, 2.f * dot(light_dir, n)), light_dir);
			V3 Id = mul(scale(lds_v3(lp + 3), shadow * di), m_diff);
			total = add(total, Id);
			float si = di * powf(clampf_(dot(refl, camera_dir), 0.f, 1.f), shininess);
			V3 Is = mul(scale(lds_v3(lp + 6), shadow * si), m_spec);
			total = add(total, Is);
		}
	}
	total = add(total, mul(v3(L.ambient), m_amb));
	/* v3clamp: max(min(v, 1), 0) — NaN → 1 (vec.h:63-65) */
	V3 c = { maxf_(minf_(total.x, 1.f), 0.f), maxf_(minf_(total.y, 1.f), 0.f), maxf_(minf_(total.z, 1.f), 0.f) };

	/* gamma + colorf_to_pixfmt, naive_renderer.c:231-232, renderer.h:17-22 */
	const float g = 1.f / 2.2f;
	c = { powf(c.x, g), powf(c.y, g), powf(c.z, g) };
	u32 px = ((u32)(c.x * 255.f) & 0xFFu) << 16 | ((u32)(c.y * 255.f) & 0xFFu) << 8 | ((u32)(c.z * 255.f) & 0xFFu);
	return { px, c, hit, shadow_steps };
}

/* Write the lane's pixel (and the optional diagnostics).  Every thread of the block must call this. */
__device__ __forceinline__ void store_pixel(const Launch& L, const Pixel& P, u32* lds) {
	u32* l_tile = lds + L.n_lights * LIGHT_DWORDS + L.n_materials * MATERIAL_DWORDS + L.n_roots;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tx = wave * WAVE_W + (lane % WAVE_W), ty = lane / WAVE_W;
	const int gx = blockIdx.x * TILE_W + tx, gr = blockIdx.y * TILE_H + ty;
	if (gx < L.w && gr < L.n_rows) {
		unsigned long long o = (unsigned long long)gr * L.w + gx;
		if (L.dbg_rgb) { L.dbg_rgb[o * 3 + 0] = P.rgb.x; L.dbg_rgb[o * 3 + 1] = P.rgb.y; L.dbg_rgb[o * 3 + 2] = P.rgb.z; }
		if (L.dbg_hit_dist) L.dbg_hit_dist[o] = P.hit.dist;
		if (L.dbg_hit_id) L.dbg_hit_id[o] = P.hit.id;
		if (L.dbg_steps) L.dbg_steps[o] = (P.hit.steps & 0xFFFFu) | (P.shadow_steps << 16);
	}
	/* through LDS so the block stores whole row segments (64 bytes each with the default 16x4 patch) */
	l_tile[ty * TILE_W + tx] = P.px;
	__syncthreads();
	const int sx = threadIdx.x % TILE_W, sy = threadIdx.x / TILE_W;
	const int ox = blockIdx.x * TILE_W + sx, orow = blockIdx.y * TILE_H + sy;
	if (ox < L.w && orow < L.n_rows)
		L.dst[(unsigned long long)orow * L.pitch_px + ox] = l_tile[sy * TILE_W + sx];
}

/* stage lights | materials | root_material into `lds` (no barrier) */
__device__ __forceinline__ void stage_common(const Launch& L, u32* lds) {
	u32* l_light = lds;
	u32* l_mat   = l_light + L.n_lights * LIGHT_DWORDS;
	u32* l_rootm = l_mat + L.n_materials * MATERIAL_DWORDS;
	for (u32 i = threadIdx.x; i < L.n_lights * LIGHT_DWORDS; i += BLOCK) l_light[i] = L.lights[i];
	for (u32 i = threadIdx.x; i < L.n_materials * MATERIAL_DWORDS; i += BLOCK) l_mat[i] = L.materials[i];
	for (u32 i = threadIdx.x; i < L.n_roots; i += BLOCK) l_rootm[i] = L.root_material[i];
}

/* Generic kernel: LDS = ops | common.  KIND != 0 selects the proven fast sqrt (the host launches that
 * instantiation only after the exhaustive check passed on the device); a wave that fed it a squared length
 * outside its proven domain shades its pixels again with the plain interpreter, as in the specialised kernel. */
template <int STACK, int KIND>
__global__ __launch_bounds__(BLOCK)
void render_interp(const Launch L) {
	extern __shared__ u32 lds[];
	u32* l_ops = lds;
	u32* l_common = l_ops + L.n_ops * OP_DWORDS;
	for (u32 i = threadIdx.x; i < L.n_ops * OP_DWORDS; i += BLOCK) l_ops[i] = L.ops[i];
	stage_common(L, l_common);
	__syncthreads();
	Interp<STACK, KIND> sdf{ l_ops, L.n_ops, {} };
	Pixel P = shade_pixel(L, sdf, l_common);
	if (KIND != 0 && __ballot(sdf.rg.outside()) != 0) {
		Interp<STACK, 0> exact{ l_ops, L.n_ops, {} };
		P = shade_pixel(L, exact, l_common);
	}
	store_pixel(L, P, l_common);
}

}  // namespace lol
