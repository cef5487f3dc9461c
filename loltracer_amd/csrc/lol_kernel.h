/*
 * lol_kernel.h — device code of the gfx950 sphere-tracer (one lane per primary ray).
 *
 * One kernel does the whole per-pixel pipeline of naive_renderer.c:217-235:
 * camera ray → sphere trace → 4-tap normal → per-light soft shadow + Phong →
 * gamma → XRGB8888.  Arithmetic is IEEE binary32 in the reference's operation
 * order (compile with -ffp-contract=off: the reference has no FMA), with
 * correctly rounded '/' and sqrt (hipcc's default), so every loop exit
 * (naive_renderer.c:61,85) is taken on the same iteration as on the CPU, and
 * powf (colour only) restates the CPU libm's algorithm (powf_glibc below), so the
 * frames come out bit-identical to the CPU oracle's, packed pixels included.
 *
 * The file is compiled twice:
 *  - ahead of time by hipcc (lol_gpu.hip) with Interp<SSIZE, KIND>: the flattened SDF
 *    program, as a list of macro-ops, is fetched with wave-uniform scalar loads and
 *    interpreted with scalar branches, an accumulator and a register operand stack;
 *  - at render_prepare time by hipRTC (lol_gpu.hip: specialise()) together with
 *    a generated `SpecSdf` whose eval() is the scene's SDF as straight-line
 *    code with every constant an immediate — the GPU counterpart of the
 *    reference's tracing JIT (tracing_jit_renderer.dasc:76-216).
 * Both use the same pipeline below, so they produce the same bits.
 *
 * Shape of the code on a 64-wide wavefront:
 *  - the march / shadow loops are per-lane loops: a lane that has hit, escaped or settled leaves
 *    EXEC (s_andn2 exec + s_cbranch_execnz — the wave goes on while any lane is left) and its
 *    state stays in its registers untouched; everything INSIDE an SDF evaluation is wave-uniform
 *    (skips are taken when the ballot of the lanes still in EXEC allows them);
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

/* dword layouts of lol_light / lol_material (lol_scene.h); checked by static_asserts in lol_gpu.hip */
constexpr int LIGHT_DWORDS = 9, MATERIAL_DWORDS = 10;

/* Launch.flags */
constexpr u32 FLAG_MISS_SKIP = 1u;   /* a wave whose rays all escaped may skip normal + lights (see shade_pixel) */
constexpr u32 FLAG_DARK_SKIP = 2u;   /* lanes whose diffuse incidence for a light is exactly 0 need no shadow march for it */
constexpr u32 FLAG_TILE_COLS = 8u;   /* the launch grid is transposed: tiles are handed out column by column (lol_gpu_set_tile_order) */
constexpr u32 FLAG_SHADOW_SETTLED = 4u;   /* a shadow march ends as soon as its factor can only be 0 (soft_shadow) */
constexpr u32 FLAG_TILE_TABLE = 32u;      /* a one-dimensional grid of one-wave blocks: block b shades the 64 pixels of wave slot Launch::tile_order[b] of the
                                           * pixel table Launch::lane_pixels (pixels dealt to waves by cost, waves handed out longest first: lol_gpu.hip) */
constexpr u32 FLAG_FIRST_STEP = 128u;     /* Launch::first_dist / first_id hold sdf(camera origin): the primary march's first step, the same for every
                                           * pixel (naive_renderer.c:56-57 with dist = 0), is not taken again per pixel (march) */
constexpr u32 FLAG_GAMMA_TABLE = 64u;     /* gamma + quantisation of a channel through Launch::gamma_table (gamma_u8_table), proven equal to the
                                           * powf route on this device for every float in [0, 1] (lol_gpu.hip, verify_gamma_kernel) */

/* = lol_frame_camera */
struct Cam { float origin[3], dir[3], right[3], up[3]; float width, height; };

/* Kernel arguments: by value, so they arrive in SGPRs. */
struct Launch {
	Cam    cam;
	float  fw, fh;               /* (float)w, (float)h */
	i32    w, h;
	i32    max_steps;
	i32    n_rows;               /* local rows this launch renders */
	i32    band_rows, cycle_rows, offset_rows;   /* lol_gpu_rows: local row r is frame row (r / band_rows) * cycle_rows + offset_rows + r % band_rows */
	u32    n_ops, n_lights, n_materials, n_roots;   /* n_ops: macro-ops in `ops` (interpreter kernel only) */
	const u32* ops;              /* device copies of the flattened scene's tables (ops: the macro-op list) */
	const u32* lights;
	const u32* materials;
	const u32* root_material;
	float  ambient[3];
	u32    flags;                /* FLAG_* */
	u32*   dst;                  /* 32-bit pixels (fmt_* below), pitch_px dwords per local row */
	u32    pitch_px;
	/* SDL_MapRGB's description of the surface's pixel format (renderer.h:17-22; lol_gpu_pixel_format), one byte per channel:
	 * pixel = (r >> Rloss) << Rshift | (g >> Gloss) << Gshift | (b >> Bloss) << Bshift | Amask */
	u32    fmt_shift;            /* Rshift | Gshift << 8 | Bshift << 16 */
	u32    fmt_loss;             /* Rloss  | Gloss  << 8 | Bloss  << 16 */
	u32    fmt_amask;
	float* dbg_rgb;
	float* dbg_hit_dist;
	u32*   dbg_hit_id;
	u32*   dbg_steps;
	/* FLAG_TILE_TABLE: tile_order[b] = the wave slot the b-th block of the launch shades; tile_cost[b] (may be NULL) receives
	 * what that block's wave cost: how long it ran (store_pixel).  lane_pixels[64 * slot + lane] = that lane's pixel: column |
	 * local row << 16 (| LANE_PADDING for a lane that only fills up its wave: it shades the pixel it names and stores nothing).
	 * pixel_cost (may be NULL) receives every pixel's step count, w per local row: what the pixels are dealt by. */
	const u32* tile_order;
	u32*   tile_cost;
	u32    tile_stride;          /* both tables are indexed by tile_slot(block): ceil(blocks / 8) */
	const u32* lane_pixels;
	unsigned short* pixel_cost;
	const float* gamma_table;    /* FLAG_GAMMA_TABLE: GAMMA_LEVELS + 1 thresholds (gamma_u8_table) */
	float  first_dist;           /* FLAG_FIRST_STEP: sdf(cam.origin) ... */
	u32    first_id;             /* ... and the object it belongs to (lol_gpu.hip, first_step) */
};
constexpr u32 LANE_PADDING = 1u << 31;

/* Fields of the launch arguments that only the last few instructions of a kernel need (destination, pitch, pixel
 * format, diagnostics) are read THERE, from the kernel-argument segment, through a pointer the compiler cannot see
 * through: read from the by-value parameter they are loaded at kernel entry and stay in SGPRs for the whole kernel —
 * a dozen registers of a file that is full (the interpreter kernel spilled SGPRs into a VGPR and lost its eighth
 * wave per SIMD over three new fields: 2860 -> 2720 Mpixels/s on C3). */
struct LaunchTail {
	u32*   dst; u32 pitch_px; u32 fmt_shift, fmt_loss, fmt_amask;
	float* dbg_rgb; float* dbg_hit_dist; u32* dbg_hit_id; u32* dbg_steps;
	const float* gamma_table;
};
__device__ __forceinline__ LaunchTail launch_tail(const Launch& L0) {
#if defined(__HIP_DEVICE_COMPILE__)
	typedef const __attribute__((address_space(4))) Launch* kernarg_ptr;
	kernarg_ptr L = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(L));                    /* opaque: these loads are not merged with the ones at kernel entry */
	return { L->dst, L->pitch_px, L->fmt_shift, L->fmt_loss, L->fmt_amask, L->dbg_rgb, L->dbg_hit_dist, L->dbg_hit_id, L->dbg_steps, L->gamma_table };
#else
	return { L0.dst, L0.pitch_px, L0.fmt_shift, L0.fmt_loss, L0.fmt_amask, L0.dbg_rgb, L0.dbg_hit_dist, L0.dbg_hit_id, L0.dbg_steps, L0.gamma_table };
#endif
}

/* Where block b's entry lies in the launch's tile tables.  Consecutive blocks of a launch go to the 8 XCDs in turn, each with an
 * L2 of its own: stored in block order, every XCD would fetch every line of the order table (8 x the table per frame) and the
 * 4-byte cost of a block would be a 32-byte write transaction of its own.  Stored XCD by XCD — block b at (b mod 8) * stride +
 * b / 8 — an XCD reads one contiguous eighth, and the costs its waves write fall into lines its L2 gathers before it writes
 * them back.  (A layout, not an assumption anything but the traffic rests on.) */
__device__ __forceinline__ u32 tile_slot(u32 b, u32 stride) { return (b & 7u) * stride + (b >> 3); }

/* Which tile of the frame this block renders in the fixed orders: its position in the grid, row by row, or column by column
 * with FLAG_TILE_COLS. */
__device__ __forceinline__ void tile_of_block(const Launch& L0, int& bx, int& by) {
	const bool cols = (L0.flags & FLAG_TILE_COLS) != 0u;
	bx = cols ? blockIdx.y : blockIdx.x;
	by = cols ? blockIdx.x : blockIdx.y;
}
/* FLAG_TILE_TABLE: this lane's entry of the pixel table — the block's wave slot from the order table (one scalar load), the
 * lane's pixel from that slot's 64 entries (one coalesced load).  Read where it is needed (at the start for the pixel's
 * coordinates, at the end for the store) like the other late fields: no register is held for it in between. */
__device__ __forceinline__ u32 lane_pixel(const Launch& L0) {
#if defined(__HIP_DEVICE_COMPILE__)
	typedef const __attribute__((address_space(4))) Launch* kernarg_ptr;
	kernarg_ptr L = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(L));
	typedef const __attribute__((address_space(4))) u32* table_ptr;
	const u32 slot = ((table_ptr)(unsigned long long)L->tile_order)[tile_slot(blockIdx.x, L->tile_stride)];
	return L->lane_pixels[(unsigned long long)slot * 64u + (threadIdx.x & 63u)];
#else
	return L0.lane_pixels[(unsigned long long)L0.tile_order[tile_slot(blockIdx.x, L0.tile_stride)] * 64u + (threadIdx.x & 63u)];
#endif
}
struct V3 { float x, y, z; };

/* ---- float.h / vec.h semantics (see oracle/lol_oracle.c for the citations) ---- */
/* Wave vote.  The compiler turns a vote on a COMPARISON into that comparison writing its lane mask to an SGPR pair,
 * but a vote on anything else — `a && b`, a bool carried around a loop — is first materialised as 0 / 1 in a VGPR
 * (v_cndmask) and compared again (v_cmp_ne): two half-rate instructions and a hazard nop per vote, and this pipeline
 * votes several times per SDF evaluation.  So votes are only ever taken on single comparisons and combined as masks on
 * the scalar side.  Inside the march / shadow loops a ballot covers the lanes still in EXEC — the ones still marching. */
typedef unsigned long long u64;
__device__ __forceinline__ u64 vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ float minf_(float a, float b) { return a < b ? a : b; }   /* MINSS: b on NaN/equal */
__device__ __forceinline__ float maxf_(float a, float b) { return a > b ? a : b; }   /* MAXSS */
/* v_min_f32 itself (fminf would first canonicalise an operand the compiler cannot prove quiet: one more instruction) */
__device__ __forceinline__ float vmin_(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
	float r;
	asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
#else
	return __builtin_fminf(a, b);
#endif
}
/* max(x, -0.f) as one v_max_f32 */
__device__ __forceinline__ float vmax_neg0_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	float r;
	asm("v_max_f32 %0, 0x80000000, %1" : "=v"(r) : "v"(x));
	return r;
#else
	return __builtin_fmaxf(x, -0.f);
#endif
}
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

/* ------------------------------------------------------------------------ powf
 * The colour path calls powf 3 + n_lights times per pixel (naive_renderer.c:158,231).  The device
 * library's powf and the CPU's (glibc) each err by < 1 ulp but not identically, which moves a few dozen
 * 8-bit channels per 4K frame by one step.  To round colours exactly like the reference does on the CPU,
 * this restates the algorithm of the powf the CPU oracle runs — glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c
 * (Szabolcs Nagy's design from ARM optimized-routines): log2(x) from a 16-entry table + degree-5 polynomial,
 * y*log2(x) in binary64, 2^z from a 32-entry table + cubic, every a*b+c fused as in glibc's FMA build
 * (the variant x86-64 glibc selects on any CPU with FMA/AVX2).  Table values are the published ones.
 * Provenance: algorithm and tables of glibc 2.35 e_powf.c / e_powf_log2_data.c / e_exp2f_data.c (GNU C Library,
 * LGPL-2.1-or-later; originally ARM optimized-routines, MIT) — third-party numerics restated for bit-parity with
 * the host libm, not part of the loltracer reference.
 * tests/test_gpu_powf.py compares it with the CPU's powf bit for bit over every float in [0, 1] for the
 * exponents in use and over millions of random (x, y) pairs including NaN / inf / negative / subnormal. */
__device__ const double POWF_LOG2_TAB[16][2] = {
	{ 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
	{ 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
	{ 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3 },
	{ 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
	{ 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1.0000000000000p+0, 0x0.0p+0 },
	{ 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3 },
	{ 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
	{ 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 }, { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 },
};
__device__ const unsigned long long POWF_EXP2_TAB[32] = {
	0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
	0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
	0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
	0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
	0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
	0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
	0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
	0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
__device__ __forceinline__ bool powf_zeroinfnan(u32 i) { return 2u * i - 1u >= 2u * 0x7f800000u - 1u; }
/* 0: y is not an integer, 1: odd integer, 2: even integer */
__device__ __forceinline__ int powf_checkint(u32 iy) {
	int e = (int)(iy >> 23 & 0xffu);
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
__device__ __noinline__ float powf_glibc(float x, float y) {
	const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
	             A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
	const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
	const double SHIFT = 0x1.8p+47;                     /* 0x1.8p52 / 32 */
	u32 sign_bias = 0;
	u32 ix = __builtin_bit_cast(u32, x), iy = __builtin_bit_cast(u32, y);
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
		/* either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan) */
		if (powf_zeroinfnan(iy)) {
			if (2u * iy == 0u) return 1.0f;
			if (ix == 0x3f800000u) return 1.0f;
			if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
			if (2u * ix == 2u * 0x3f800000u) return 1.0f;
			if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;   /* |x|<1 && y==inf or |x|>1 && y==-inf */
			return y * y;
		}
		if (powf_zeroinfnan(ix)) {
			float x2 = x * x;
			if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
			return (iy & 0x80000000u) ? 1 / x2 : x2;
		}
		/* x and y are non-zero finite */
		if (ix & 0x80000000u) {
			int yint = powf_checkint(iy);
			if (yint == 0) return (x - x) / (x - x);
			if (yint == 1) sign_bias = 1u << (5 + 11);
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u) {                          /* normalise subnormal x */
			ix = __builtin_bit_cast(u32, x * 0x1p23f);
			ix &= 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	/* log2_inline */
	u32 tmp = ix - 0x3f330000u;
	int i = (int)((tmp >> (23 - 4)) % 16u);
	u32 top = tmp & 0xff800000u;
	u32 iz = ix - top;
	int k = (int)top >> 23;
	double invc = POWF_LOG2_TAB[i][0], logc = POWF_LOG2_TAB[i][1];
	double z = (double)__builtin_bit_cast(float, iz);
	double r = __builtin_fma(z, invc, -1.0);
	double y0 = logc + (double)k;
	double r2 = r * r;
	double yy = __builtin_fma(A0, r, A1);
	double pp = __builtin_fma(A2, r, A3);
	double r4 = r2 * r2;
	double q = __builtin_fma(A4, r, y0);
	q = __builtin_fma(pp, r2, q);
	double logx = __builtin_fma(yy, r4, q);
	double ylogx = (double)y * logx;                      /* cannot overflow, y is single precision */
	if ((__builtin_bit_cast(unsigned long long, ylogx) >> 47 & 0xffffu) >=
	    (__builtin_bit_cast(unsigned long long, 126.0) >> 47)) {
		/* |y*log(x)| >= 126 */
		if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
		if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
	}
	/* exp2_inline */
	double kd = ylogx + SHIFT;
	unsigned long long ki = __builtin_bit_cast(unsigned long long, kd);
	kd -= SHIFT;
	double rr = ylogx - kd;
	unsigned long long t = POWF_EXP2_TAB[ki % 32u];
	unsigned long long ski = ki + sign_bias;
	t += ski << (52 - 5);
	double sc = __builtin_bit_cast(double, t);
	double zz = __builtin_fma(C0, rr, C1);
	double rr2 = rr * rr;
	double res = __builtin_fma(C2, rr, 1.0);
	res = __builtin_fma(zz, rr2, res);
	res = res * sc;
	return (float)res;
}

/* ---------------------------------------------------------------- gamma + quantisation
 * What a surface receives of a colour channel is 8 bits: Uint8 r = powf(c, 1 / 2.2f) * 255 (naive_renderer.c:231-232,
 * renderer.h:17-22), c in [+0, 1] after v3clamp.  As a function of c that is a staircase of 256 steps, so the 8 bits can be had
 * without the powf (5 % of a C3 frame went into the five powf of a pixel): GAMMA_LEVELS + 1 thresholds — T[k] = the smallest
 * c whose exact value is >= k, T[0] = 0, T[256] = +inf, found on the device with powf_glibc itself — an estimate of k from the
 * hardware's log2 / exp2, and two comparisons that correct an estimate one step off.  Used only after the device has run every
 * float in [0, 1] (2^30 of them) through both routes and found no difference (lol_gpu.hip: verify_gamma_kernel — which also
 * proves the staircase monotone and the estimate never more than one step off).  The float colour itself (diagnostics:
 * lol_gpu_debug::rgb) still comes from powf_glibc. */
constexpr int GAMMA_LEVELS = 256;
__device__ __forceinline__ u32 gamma_u8_exact(float c) { return (u32)(powf_glibc(c, 1.f / 2.2f) * 255.f) & 0xFFu; }
__device__ __forceinline__ u32 gamma_u8_table(float c, const float* T) {
#if defined(__HIP_DEVICE_COMPILE__)
	const float a = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.f / 2.2f)) * 255.f;      /* c = +0: exp2(-inf) = 0 */
#else
	const float a = 0.f;
#endif
	u32 k = (u32)a;
	k = k < (u32)GAMMA_LEVELS - 1u ? k : (u32)GAMMA_LEVELS - 1u;
	const float lo = T[k], hi = T[k + 1u];
	return k + (c >= hi ? 1u : 0u) - (c < lo ? 1u : 0u);
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
/* FIXUP = false leaves v_div_fixup_f32 out (a half-rate instruction at the end of every blend factor's dependent chain).
 * Used where the device has shown, again for every float dlt and per k, that the clamped factor is the same without it
 * for every FINITE dlt and that the quotient is NaN for dlt = +-inf.  An infinite dlt makes the smooth minimum NaN in
 * either form unless one OPERAND is -inf — and then this form's NaN reaches the object's value, where the generated code
 * votes on it (like for a sphere without range tracker): that wave shades again through the plain path. */
template <bool FIXUP = true>
__device__ __forceinline__ float smin_h_fast(float dlt, float k2, float hrk) {
	float q = dlt * hrk;
	float r = __builtin_fmaf(-q, k2, dlt);
	q = __builtin_fmaf(r, hrk, q);
	if (FIXUP) q = __builtin_amdgcn_div_fixupf(q, k2, dlt);
	return clampf_(.5f + q, 0.f, 1.f);
}
/* sminf (float.h:29-33) with the proven blend factor.  lerp(b, a, h) = b + (a - b)*h is written b - (b - a)*h:
 * a - b and b - a are exact negatives of each other unless a == b with equal signs, where h = .5 and the two
 * forms can only differ in the sign of a zero sum (a == b == -0) that the subtraction of k*h*(1-h) = k/4 != 0
 * then erases — lol_gpu.hip uses this form only for |k| >= 2^-100 so that k/4 is a non-zero normal number. */
template <bool FIXUP = true>
__device__ __forceinline__ float sminf_fastdiv(float a, float b, float k, float k2, float hrk) {
	float dlt = b - a;
	float h = smin_h_fast<FIXUP>(dlt, k2, hrk);
	return (b - dlt * h) - k * h * (1.f - h);
}

/* ... and with the saturated cases taken out of the arithmetic.  For k > 0 and ks >= k(1 + 2^-21):
 *   dlt >=  ks  =>  (.5*dlt)/k >= .5(1 + 2^-21), so .5f + q >= 1 and h == 1:  result = (b - dlt*1) - k*1*(+0) = b - dlt;
 *   dlt <= -ks  =>  .5f + q < 0 and h == +0:                                   result = (b - dlt*0) - k*0*1  = b - dlt*0.f
 * (dlt*0.f is kept: it is -0, or NaN for dlt = -inf, exactly as in the full expression).  verify_div_kernel checks
 * both implications for every float dlt on the device next to the proof of smin_h_fast.  The shortcut is taken per
 * WAVE — when every lane (in EXEC) is saturated — so it costs one compare and a scalar branch where it does
 * not apply: two spheres more than k apart in distance is the common case away from the seams of a blob
 * (scene4 C3: +8.7 %).  NaN dlt is never saturated.  "Every lane" = every lane in EXEC: inside the march and shadow loops the
 * lanes still marching. */
template <bool FIXUP = true>
__device__ __forceinline__ float sminf_fastdiv_sat(float a, float b, float k, float k2, float hrk, float ks) {
	const float dlt = b - a;
	if (vote(!(__builtin_fabsf(dlt) >= ks)) == 0) {
		asm volatile("");                         /* keep the branch: if-converted, every evaluation would pay for both sides */
		const float r1 = b - dlt;
		const float r0 = b - dlt * 0.f;
		return dlt > 0.f ? r1 : r0;
	}
	const float h = smin_h_fast<FIXUP>(dlt, k2, hrk);
	return (b - dlt * h) - k * h * (1.f - h);
}
/* ... and where both operands are known to be finite or NaN — spheres on the proven root without range tracker (a squared
 * length that overflows gives NaN there, never inf: sd_sphere_fast_nr) and smooth minima of such — so that a saturated dlt
 * (|dlt| >= ks > 0; NaN is never saturated) is finite and not 0:
 *   dlt > 0:  b - dlt;      dlt < 0:  b - dlt*0.f = b - (-0.f)      i.e.  b - max(dlt, -0.f)
 * two full-rate instructions for the multiply, compare, select and subtract above (round 6: +0.9 % on C3, +1.7 % for a new
 * view; profiles/r6_ab2.txt).  lol_gpu.hip (emit_sdf: Node::fon) decides per smooth union which form is generated. */
template <bool FIXUP = true>
__device__ __forceinline__ float sminf_fastdiv_sat2(float a, float b, float k, float k2, float hrk, float ks) {
	const float dlt = b - a;
	if (vote(!(__builtin_fabsf(dlt) >= ks)) == 0) {
		asm volatile("");
		return b - vmax_neg0_(dlt);
	}
	const float h = smin_h_fast<FIXUP>(dlt, k2, hrk);
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
/* The same without the range tracker, for spheres of radius r >= 2^-20 once the device has also shown that the
 * fast root of every x in [0, 2^-96) is either NaN or smaller than 2^-47 in magnitude (verify_sqrt_kernel, second
 * counter).  Then the sphere's value is right or NaN for EVERY l2:
 *   l2 in [2^-96, inf): the proven domain;   l2 = +inf or NaN: the fast root gives NaN (the plain one inf / NaN);
 *   l2 in [0, 2^-96): the true root is < 2^-48 < ulp(r)/2, so the true value is exactly -r, and so is (root') - r for
 *   any root' below 2^-47 — or it is NaN.
 * A NaN operand makes every smooth minimum above it NaN (dlt = NaN, h clamps to 0, b - NaN*0 = NaN), so it reaches
 * the object's value, where the generated code votes on `t != t` — one comparison per OBJECT instead of a min and a
 * max per squared length (six half-rate instructions per evaluation of scene4's blob).  A wave with such a NaN
 * shades its pixels again through the plain path, like one that left the range. */
template <int KIND>
__device__ __forceinline__ float sd_sphere_fast_nr(V3 p, float cx, float cy, float cz, float r) {
	V3 q = { p.x - cx, p.y - cy, p.z - cz };
	return sqrt_fast<KIND>(len2(q)) - r;
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
 * Runs the scene's SDF for one point per lane from a list of MACRO-OPS (lol_gpu.hip: build_mops translates the
 * post-order program of lol_scene.h).  The machine has an accumulator `acc` (the top of the operand stack) and a
 * small register stack under it; one macro-op is "produce x, combine it into acc, maybe finish a root":
 *   x     = sphere | round box | plane distance of this lane's point, or the popped stack entry
 *   comb  = SET (acc = x) | PUSH (push acc; acc = x) | SMIN (acc = sminf(acc, x)) | SMIN_X (acc = sminf(x, acc))
 *   top   : if (acc < best) { best = acc; best_id = id }            — sdf(), strict '<' (naive_renderer.c:39)
 * so a chain of smooth unions whose one child is a primitive never touches the stack, and scene4's ten
 * post-order ops are seven macro-ops (six with MOPB_POST, below).
 *
 * Record = 12 dwords: header bits | object id (TOP) | 7 primitive parameters | k, 2k, .5/k of the smooth min.
 * (Round 3 tried 8-dword records — id in the header's upper bits, stack slots in unused words, a round box's b and r sent ahead
 * in a record of their own: +0.6 % on scene4, -8 % on scene.lol, whose box then costs two records; a second granule loaded inside
 * the box's branch made the compiler structurize the whole group of rare kinds: -8 % on scene4.  16 dwords in ONE
 * s_load_dwordx16: -3.5 %.)
 * The records are wave-uniform data and are fetched with SCALAR loads straight into SGPRs
 * (constant address space, uniform index → s_load_dwordx4 through the scalar cache): measured on MI355X
 * (tools/salu_rate.hip) a scalar-cache dword costs ~2 cycles per SIMD against 8.6 for a ds_read_b32 in which all
 * 64 lanes read one LDS address, and it needs no LDS-address VGPR and no v_readfirstlane; the header word is
 * then already scalar, so the dispatch is s_cmp / s_cbranch.  This replaced the LDS-staged op list of round 1
 * (VALU 2.4x, SALU 11x the specialised kernel's; profiles/README.md). */
constexpr int MOP_DWORDS = 12;
enum { MOP_SPHERE = 0, MOP_RBOX = 1, MOP_PLANE = 2, MOP_POP = 3 };          /* what x is          */
enum { MOP_SET = 0, MOP_PUSH = 1, MOP_SMIN = 2, MOP_SMIN_X = 3 };           /* how it is combined */
/* header word: one bit per decision */
constexpr u32 MOPB_SPHERE = 1u, MOPB_RBOX = 2u, MOPB_PLANE = 4u, MOPB_POP = 8u;
constexpr u32 MOP_FASTDIV = 16u;     /* the smooth min's blend factor was proven for this k (word 9): words 10, 11 = 2k, .5/k */
constexpr u32 MOP_TOP = 32u;         /* acc is a finished top-level object: id in word 1                           */
constexpr u32 MOPB_PUSH = 64u, MOPB_SMIN = 128u, MOPB_X_IS_A = 256u;   /* SET = none of PUSH / SMIN */
/* summary bits, so that the common record (a sphere, no push, not the end of an object) pays ONE s_bitcmp for each
 * group of rare cases instead of an s_and + s_cmp: */
constexpr u32 MOPB_NOT_SPHERE = 512u;   /* = RBOX | PLANE | POP */
constexpr u32 MOPB_TAIL = 1024u;        /* = PUSH | TOP | POST (set by build_mops when it sets MOP_TOP / MOPB_POST) */
/* which smooth min, one bit each (set by build_mops next to MOPB_SMIN / MOPB_X_IS_A / MOP_FASTDIV): */
constexpr u32 MOPB_SMIN_AF = 2048u;     /* sminf_fastdiv<false>(acc, x): k proven without v_div_fixup_f32 as well (MOP_NOFIXUP: only in the list
                                         * that lol_gpu.hip launches when every operand difference is finite) */
constexpr u32 MOPB_SMIN_XF = 4096u;     /* sminf_fastdiv<false>(x, acc) */
constexpr u32 MOPB_SMIN_EXACT = 8192u;  /* unproven k: sminf_ with the correctly rounded division, order by MOPB_X_IS_A */
constexpr u32 MOPB_SMIN_REST = 131072u; /* = SMIN_AFX | SMIN_XFX | SMIN_EXACT: the two common smooth minima pay ONE test for these */
constexpr u32 MOPB_SMIN_AFX = 262144u;  /* sminf_fastdiv<true>(acc, x): k proven only with the fixup */
constexpr u32 MOPB_SMIN_XFX = 524288u;  /* sminf_fastdiv<true>(x, acc) */
/* A smooth min of two finished sub-unions (a POP record of its own in round 2) rides on the record that finished the second
 * one when both use the same, proven k: after that record's own combine, acc' = sminf(s[slot], acc') or sminf(acc', s[slot]).
 * Lives in the TAIL group (ordinary records pay nothing); slot in bits 25-28. */
constexpr u32 MOPB_POST = 1u << 29, MOPB_POST_YA = 1u << 30;      /* YA: the popped value is the smooth min's first operand */
constexpr u32 MOP_POST_SLOT_SHIFT = 25u;
constexpr u32 MOPB_STACK = 1u << 31;                              /* = POST | PUSH */
constexpr u32 MOP_NOFIXUP = 1u << 24;   /* with MOP_FASTDIV: the blend factor is also proven without v_div_fixup_f32 (smin_h_fast<false>) */
/* Exact object culling (lol_gpu.hip, "exact culling") in the interpreter: a record that finishes a top-level object
 * (MOP_TOP) may carry MOPB_CULL_NEXT / MOPB_CULL_CHAIN — the NEXT record is then not a macro-op but a test's constants
 * {word 0 = CULLC_* flags, word 1 = how many records after it belong to the objects the test guards, f[2..4] = C,
 * f[5] = R', f[6] = K}; if every lane in EXEC may skip them, they are jumped over.  Runs nest (lol_gpu.hip, plan_culling):
 * CULLC_NEXT says the record after this one is the test of a run nested in this one, CULLC_AFTER that the record
 * after the guarded ones is the test of the run that follows.  Lives in the rare TAIL branch, so ordinary records
 * pay nothing. */
constexpr u32 MOPB_CULL_NEXT = 16384u;    /* the test of the run of ALL bounded objects: the one with a cool-down */
constexpr u32 MOPB_CULL_CHAIN = 65536u;   /* one or more tests of inner runs follow (after the NEXT one, if both) */
constexpr u32 CULLC_NEXT = 1u, CULLC_AFTER = 2u;
constexpr u32 MOP_TIE = 32768u;         /* with MOP_TOP: evaluated after an object that follows it in the file → ties go to the lower id */
constexpr u32 CULL_COOLDOWN = 3u;       /* after a test that did not allow the skip, this many evaluations do not test */
/* The operand stack under the accumulator is a set of numbered slots, and the slot a PUSH fills / a POP empties is known
 * when the list is built (the post-order depth): it travels in the record (bits 20-23), so the slots never move.  (As a
 * shifting stack — s[j] = s[j - 1] — every record of the general path paid four v_mov for values it did not touch: the
 * loop-carried copies of the shifted registers; read off the ISA.) */
constexpr u32 MOP_SLOT_SHIFT = 20u, MOP_SLOT_MASK = 15u;
__host__ __device__ constexpr u32 mop_smin_bits(u32 hdr) {
	return !(hdr & MOPB_SMIN) ? 0u : !(hdr & MOP_FASTDIV) ? (MOPB_SMIN_EXACT | MOPB_SMIN_REST)
	     : (hdr & MOP_NOFIXUP) ? ((hdr & MOPB_X_IS_A) ? MOPB_SMIN_XF : MOPB_SMIN_AF)
	     : ((hdr & MOPB_X_IS_A) ? MOPB_SMIN_XFX : MOPB_SMIN_AFX) | MOPB_SMIN_REST;
}
__host__ __device__ constexpr u32 mop_header(u32 kind, u32 comb) {
	return (1u << kind) | (kind != MOP_SPHERE ? MOPB_NOT_SPHERE : 0u) |
	       (comb == MOP_PUSH ? (MOPB_PUSH | MOPB_TAIL | MOPB_STACK) : comb == MOP_SMIN ? MOPB_SMIN : comb == MOP_SMIN_X ? (MOPB_SMIN | MOPB_X_IS_A) : 0u);
}

/* scalar (constant address space) view of the list; loaded dword by dword — the compiler merges the loads into
 * s_load_dwordx8 + x4 (vector-typed loads of such records miscompile under hipcc 7.2: lane .y read as .x) */
typedef const __attribute__((address_space(4))) u32* mop_ptr;

/* Operand stacks deeper than the 4-bit slot fields (MOP_DEEP_FROM entries under the accumulator: a tree of more than 4096
 * primitives in ONE object) use the instantiation Interp<MOP_DEEP_SLOTS>: the slot number travels in a word of its own —
 * word 9 of a PUSH record (it has no smooth min), word 2 of a POP record (it has no primitive) — the stack is indexed
 * with it directly (registers or scratch, the compiler's choice) and lol_gpu.hip's build_mops does not fuse pops into
 * MOPB_POST for such lists.  The reference recurses without a limit (naive_renderer.c:11-28); this is its counterpart. */
constexpr int MOP_DEEP_FROM = 12, MOP_DEEP_SLOTS = 63;

template <int SSIZE, int KIND = 0>
struct Interp {
	static constexpr bool DEEP = SSIZE >= MOP_DEEP_FROM;
	const u32* mops;     /* global memory, MOP_DWORDS per macro-op, 16-byte aligned */
	u32        n_mops;
	Range      rg;       /* KIND != 0: range of the squared lengths given to the proven fast sqrt (see above) */
	u32        cool;     /* evaluations left before a CULL_NEXT record tests again (wave-uniform) */
	float      nanacc = 0.f;  /* (the specialised SDF's "an object's value went NaN" accumulator; never set here) */
	/* the fast pipeline of the SPECIALISED kernel takes FLAG_SHADOW_SETTLED for granted (soft_shadow); this one reads the flag */
	static constexpr bool ASSUME_SETTLED = false;
	static constexpr bool ASK_ID_ONCE = false;           /* (march(): measured a loss for this kernel on scene4, profiles/r6_ab_id_asked_once.txt) */
	/* `cool` is wave-uniform state that the march / shadow loops change while lanes leave them one by one: what comes out of
	 * such a loop counts as per-lane for the compiler (it would travel in a VGPR and every later test of it would be a
	 * per-lane branch).  The loops end by calling this: never testing — and testing at once — are both always allowed. */
	__device__ __forceinline__ void loop_done() { cool = 0u; }

	/* Inlined into the march / normal / shadow loops: as a real (noinline) function it was 8 % slower — arguments
	 * travel in VGPRs and need v_readfirstlane, plus call / return and the callee's register shuffling. */
	/* eval_dist: the distance alone, for the shadow marches, which never look at the id: a finished object joins the running minimum
	 * with ONE v_min_f32 and no test of MOP_TIE.  Same value as the strict-'<' / lower-id-wins rule except for the sign of a zero,
	 * which no result of a shadow march depends on (soft_shadow: t + s, 50 s / t against 0, maxf(+-0, 0) = +0).  Round 6: the
	 * interpreter +4.3 ... 5.6 %, the specialised kernel +0.8 % (C3) ... +8.6 % (C2: four flat objects); profiles/r6_ab_eval_dist.txt.
	 * (The primary march on the distance alone too, its id asked for once: a loss for this kernel on scene4 — march(), ASK_ID_ONCE.) */
	__device__ __forceinline__ void eval_dist(V3 p, float& best_out) { u32 unused; eval<false>(p, best_out, unused); }
	template <bool WITH_ID = true>
	__device__ __forceinline__ void eval(V3 p, float& best_out, u32& id_out) {
		/* everything the loop carries is a local: the trip count and the list pointer must stay provably
		 * wave-uniform (scalar loop, scalar loads, scalar branches), which they do not when they live behind
		 * `this` next to per-lane results written through references */
		const u32 n = n_mops;
		mop_ptr rec = (mop_ptr)(unsigned long long)mops;
		Range r = rg;
		u32 cl = cool;
		float s[SSIZE];
#pragma unroll
		for (int i = 0; i < SSIZE; i++) s[i] = 0.f;
		float acc = 0.f;
		float best = __builtin_inff();
		u32 best_id = 0;
		const mop_ptr end = rec + n * MOP_DWORDS;                         /* (no trip counter: one scalar instruction less per record) */
		for (; rec != end; rec += MOP_DWORDS) {                           /* (the TAIL branch may jump further) */
			/* the whole record at once, up here: read where they are used, the fields arrived in three or four
			 * separate scalar loads per record, each with its own wait (the branches below are barriers for the
			 * compiler's load merging) — one s_load_dwordx8 + one x4 and ONE wait instead */
			const u32 w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3], w4 = rec[4], w5 = rec[5], w6 = rec[6], w7 = rec[7],
			          w8 = rec[8], w9 = rec[9], w10 = rec[10], w11 = rec[11];
			asm volatile("" :: "s"(w0), "s"(w1), "s"(w2), "s"(w3), "s"(w4), "s"(w5), "s"(w6), "s"(w7), "s"(w8), "s"(w9), "s"(w10), "s"(w11));
			const u32 hdr = w0;
			const u32 wv[MOP_DWORDS] = { w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11 };
			auto F = [&](int j) { return __builtin_bit_cast(float, wv[j]); };
			auto C = [&](int j) { return __builtin_bit_cast(float, rec[j]); };      /* a test's constants record, after rec has moved on to it */
			/* one-hot header bits, tested one by one (s_bitcmp1 + s_cbranch each) with no else-chains: every `if`
			 * is a plain skip-ahead, which the compiler lowers without the flag registers it needs for if / else-if */
			/* LOL_KEEP_BRANCH: an empty volatile asm keeps the compiler from turning a rarely taken uniform branch into
			 * v_cndmask selects that every macro-op would then pay for (half-rate VALU, 4 cycles each) */
#define LOL_KEEP_BRANCH() asm volatile("" ::: "memory")
			/* LOL_RARE / LOL_OFTEN: where the body of a header test goes — rare bodies out of line, so that the common
			 * record (a sphere smooth-minned into the accumulator) falls through its tests instead of jumping over them:
			 * a taken branch costs the wave an instruction refetch (+0.7 % on C3, measured) */
#define LOL_RARE(c) __builtin_expect(!!(c), 0)
#define LOL_OFTEN(c) __builtin_expect(!!(c), 1)
			float x = 0.f;
			if (LOL_OFTEN(hdr & MOPB_SPHERE))
				x = KIND ? sd_sphere_fast<KIND ? KIND : 1>(p, F(2), F(3), F(4), F(5), r) : sd_sphere(p, F(2), F(3), F(4), F(5));
			if (LOL_RARE(hdr & MOPB_NOT_SPHERE)) {                   /* grouped: a sphere macro-op pays one test for these */
				LOL_KEEP_BRANCH();
				if (hdr & MOPB_RBOX)
					x = KIND ? sd_round_box_fast<KIND ? KIND : 1>(p, F(2), F(3), F(4), F(5), F(6), F(7), F(8), r)
					         : sd_round_box(p, F(2), F(3), F(4), F(5), F(6), F(7), F(8));
				if (hdr & MOPB_PLANE)
					x = p.y - F(2);                              /* plane: (p - (0,y,0)).y */
				if (hdr & MOPB_POP) {
					LOL_KEEP_BRANCH();
					if constexpr (DEEP) {
						x = s[w2 < (u32)SSIZE ? w2 : 0u];
					} else {
						const u32 slot = hdr >> MOP_SLOT_SHIFT & MOP_SLOT_MASK;
						x = s[0];
#pragma unroll
						for (int j = 1; j < SSIZE; j++) x = slot == (u32)j ? s[j] : x;
					}
				}
			}
			if (LOL_OFTEN(hdr & MOPB_SMIN)) {
				/* four independent skip-aheads (operand order x fast / exact blend factor), the common fast ones first:
				 * a nested if / ?: here made the compiler hoist the FASTDIV test through a VGPR and add flag registers */
				LOL_KEEP_BRANCH();
				if (LOL_OFTEN(hdr & MOPB_SMIN_AF)) { LOL_KEEP_BRANCH(); x = sminf_fastdiv<false>(acc, x, F(9), F(10), F(11)); }
				if (LOL_RARE(hdr & MOPB_SMIN_XF)) { LOL_KEEP_BRANCH(); x = sminf_fastdiv<false>(x, acc, F(9), F(10), F(11)); }
				if (LOL_RARE(hdr & MOPB_SMIN_REST)) {                /* grouped, like the kinds above */
					LOL_KEEP_BRANCH();
					if (hdr & MOPB_SMIN_AFX) { LOL_KEEP_BRANCH(); x = sminf_fastdiv<true>(acc, x, F(9), F(10), F(11)); }
					if (hdr & MOPB_SMIN_XFX) { LOL_KEEP_BRANCH(); x = sminf_fastdiv<true>(x, acc, F(9), F(10), F(11)); }
					if (hdr & MOPB_SMIN_EXACT) {
						LOL_KEEP_BRANCH();
						if (!(hdr & MOPB_X_IS_A)) { LOL_KEEP_BRANCH(); x = sminf_(acc, x, F(9)); }
						if (hdr & MOPB_X_IS_A) { LOL_KEEP_BRANCH(); x = sminf_(x, acc, F(9)); }
					}
				}
			}
			if (LOL_RARE(hdr & MOPB_TAIL)) {
				LOL_KEEP_BRANCH();
				if (hdr & MOPB_STACK) {                                 /* POST | PUSH: the end of an object pays one test for both */
					LOL_KEEP_BRANCH();
					if constexpr (!DEEP) if (hdr & MOPB_POST) {                /* (deep lists carry no POST: build_mops) */
						LOL_KEEP_BRANCH();
						const u32 ps = hdr >> MOP_POST_SLOT_SHIFT & MOP_SLOT_MASK;
						float y = s[0];
#pragma unroll
						for (int j = 1; j < SSIZE; j++) y = ps == (u32)j ? s[j] : y;
						const bool ya = (hdr & MOPB_POST_YA) != 0u;
						const float a = ya ? y : x, b = ya ? x : y;
						x = sminf_fastdiv<true>(a, b, F(9), F(10), F(11));      /* (one body for both orders and both lists: the form with the fixup) */
					}
					if (hdr & MOPB_PUSH) {
						LOL_KEEP_BRANCH();
						if constexpr (DEEP) {
							s[w9 < (u32)SSIZE ? w9 : 0u] = acc;
						} else {
							const u32 slot = hdr >> MOP_SLOT_SHIFT & MOP_SLOT_MASK;
#pragma unroll
							for (int j = 0; j < SSIZE; j++) s[j] = slot == (u32)j ? acc : s[j];
						}
					}
				}
				if (hdr & MOP_TOP) {                                    /* (tests only follow finished objects) */
					LOL_KEEP_BRANCH();
					if constexpr (WITH_ID) {
						const u32 id = w1;
						bool take = x < best;
						/* (no short circuit: `take || (eq && gt)` became exec juggling around the second compare; +1.5 % as three plain compares) */
						if (hdr & MOP_TIE) { LOL_KEEP_BRANCH(); take = take | ((x == best) & (best_id > id)); }
						if (take) { best = x; best_id = id; }
					} else {
						best = vmin_(x, best);
					}
					if (hdr & MOPB_CULL_NEXT) {                             /* the run of all bounded objects: the test that cools down */
						LOL_KEEP_BRANCH();
						rec += MOP_DWORDS;                                  /* the constants record is consumed either way */
						if (cl == 0u) {
							LOL_KEEP_BRANCH();
							const float cx = p.x - C(2), cy = p.y - C(3), cz = p.z - C(4);
							const float l2 = (cx * cx + cy * cy) + cz * cz;
							const float u = (best + C(5)) * C(6);
							/* any lane (in EXEC: still marching) that may not skip (skip = l2 > u*u && u > 0).  One ballot per comparison, combined
							 * on the scalar side: the ballot of the compound condition went through a VGPR (v_cndmask + v_cmp_ne) */
							const u64 needed = vote(!(l2 > u * u)) | vote(!(u > 0.f));
							cl = CULL_COOLDOWN + 1u;                        /* (one branch: `if (needed != 0)` after it cost a flag register) */
							if (needed == 0) {
								LOL_KEEP_BRANCH();
								const u32 k = rec[1];                       /* to the last record: nothing follows that run */
								rec += k * MOP_DWORDS;
								cl = 0u;
							}
						}
						cl = cl ? cl - 1u : 0u;
					}
					if (hdr & MOPB_CULL_CHAIN) {
						LOL_KEEP_BRANCH();
						u32 more = rec + MOP_DWORDS != end;                 /* 0: the test above has just jumped to the last record */
						while (more) {                                      /* wave-uniform: one turn per test record met */
							LOL_KEEP_BRANCH();
							rec += MOP_DWORDS;
							const u32 ch = rec[0];
							more = ch & CULLC_NEXT;
							const float cx = p.x - C(2), cy = p.y - C(3), cz = p.z - C(4);
							const float l2 = (cx * cx + cy * cy) + cz * cz;
							const float u = (best + C(5)) * C(6);
							if ((vote(!(l2 > u * u)) | vote(!(u > 0.f))) == 0) {
								LOL_KEEP_BRANCH();
								const u32 k = rec[1];
								rec += k * MOP_DWORDS;
								more = ch & CULLC_AFTER;
							}
						}
					}
				}
			}
			acc = x;
#undef LOL_RARE
#undef LOL_OFTEN
#undef LOL_KEEP_BRANCH
		}
		rg = r;
		cool = cl;
		best_out = best;
		id_out = best_id;
	}
};

/* did this wave's fast SDF leave what was proven for it (a squared length outside the fast root's domain, or a NaN / inf
 * value of an object whose spheres carry no range tracker: nanacc = fma(value, 0, nanacc) stays 0 for finite values)?
 * Then its results are not used. */
template <class Sdf>
__device__ __forceinline__ bool unproven(const Sdf& sdf) { return (vote(sdf.rg.outside()) | vote(sdf.nanacc != sdf.nanacc)) != 0; }

/* --------------------------------------------------------------- the pipeline */

struct Hit { float dist; u32 id; u32 steps; };
/* what march() hands on: the distance marched, the distance BEFORE its last step (the point that step evaluated), the step count,
 * and the object id as far as it is known without asking — `ask`: this lane's id is that of the object nearest to the point of
 * its last step, which normal_and_id() asks for (Sdf::ASK_ID_ONCE) */
struct Marched { float dist, prev; u32 steps; u32 id; bool ask; };

/* get_intersection, naive_renderer.c:48-69.
 *
 * A per-lane loop: a lane that has hit or escaped takes the `break`, i.e. leaves EXEC — its dist / id / steps stay where they
 * are, no select keeps them — and the wave goes round while any lane is left (s_andn2 exec; s_cbranch_execnz).  Rounds 1 - 5
 * kept the control flow wave-uniform instead (`while (vote(alive))`, every update a v_cndmask under `alive`): three selects and
 * two extra compares per step at half rate; round 6 measured the plain loop +4.7 % on C3, +7 % on C2, +3 ... 4.7 % on the
 * interpreter (profiles/r6_ab1.txt, r6_ab_interp.txt).  Inside sdf.eval every ballot now covers the marching lanes only, which
 * is exactly what the "care" masks of those rounds were handed down for.
 *
 * FLAG_FIRST_STEP: step 0 evaluates the SDF at ro + rd * 0 = ro — the camera, the same point for every pixel of the frame — so the
 * host passes that one value (lol_gpu.hip, first_step: computed once per camera position) and the loop starts at step 1.  What
 * makes ro + rd * 0 equal ro bit for bit: rd finite (rd * 0 = +-0; checked here, per wave: a wave with a lane whose direction
 * is not finite marches from step 0) and no component of ro a negative zero (-0 + +0 = +0; the host checks, with first_dist in
 * [EPSILON, MAX_DIST] — a march that would end on its first step is left to the loop — and max_steps >= 1).
 * Round 6: +1.9 % on C3, +1.2 % for a new view, +4 % with two frames in flight (profiles/r6_ab4.txt).
 *
 * COUNT = false compiles the step counters out (Hit::steps stays 0): only the diagnostics and the one frame of a view that
 * records what its pixels cost read them (+1.3 %, profiles/r6_ab1.txt).
 *
 * Sdf::ASK_ID_ONCE (a property of the scene's own kernel: lol_codegen.hip sets it for scenes of three or more top-level objects):
 * the march evaluates the scene's DISTANCE alone (eval_dist), like the shadow marches.  The reference carries the id of the
 * nearest object through every step (id = d.id, naive_renderer.c:60) and uses the LAST one — unless the ray escaped (dist >=
 * MAX_DIST: id = 0, :65-66).  So the id is asked for once, at the point the last step evaluated (ro + rd * prev, the same
 * expression: the same bits), by one more turn of normal_and_id()'s evaluation loop, and per step an object joins the minimum
 * with one v_min_f32 instead of a compare and two selects.  The distance-only minimum differs from the strict-'<' one in the
 * sign of a zero at most, and nothing here looks at that: dist + (+-0), (+-0) < EPSILON.  Which lanes must ask: those whose march
 * took at least one turn of the loop (wave-uniform: every lane takes the first turn of a loop that runs at all) and did not escape.
 * Round 6 (profiles/r6_ab_id_asked_once.txt): scene.lol's four flat objects +3.6 % (C2), scene4's two -0.8 % (C3) — the extra
 * evaluation per pixel against 8 cycles per object and step — hence per scene; the interpreter -0.4 ... -0.9 % / +2.5 %: it keeps
 * the id in its march. */
template <bool COUNT, class Sdf>
__device__ __forceinline__ Marched march(Sdf& sdf, V3 ro, V3 rd, int max_steps, bool first_given, float first_dist, u32 first_id) {
	const float EPSILON = 0.001f, MAX_DIST = 100.f;
	/* (a NaN or infinite component makes the squared length NaN or inf) */
	const bool skip_first = first_given && vote(!(len2(rd) < 4.f)) == 0;
	float dist = skip_first ? first_dist : 0.f, prev = 0.f;
	u32 id = skip_first ? first_id : 0u, steps = skip_first ? 1u : 0u;
	const int first_turn = skip_first ? 1 : 0;
	for (int i = first_turn; i < max_steps; i++) {
		V3 p = add(ro, scale(rd, dist));
		float d;
		if constexpr (Sdf::ASK_ID_ONCE) {
			sdf.eval_dist(p, d);
			prev = dist;
		} else {
			u32 did;
			sdf.eval(p, d, did);
			id = did;
		}
		dist += d;
		if (COUNT) steps++;
		if (d < EPSILON || dist > MAX_DIST) break;
	}
	sdf.loop_done();
	const bool far = dist >= MAX_DIST;
	if (far) id = 0;
	if constexpr (Sdf::ASK_ID_ONCE) {
		const bool ran = first_turn < max_steps;
		if (ran) id = 0;                                 /* (asked for below, unless the ray escaped) */
		return { dist, prev, steps, id, ran && !far };
	}
	return { dist, prev, steps, id, false };
}

/* in_shadow + softshadow, naive_renderer.c:73-100.  dir/light_dist come from the caller,
 * which needs the same normalize(light - p) for the Phong term. */
/*
 * A per-lane loop like march(): lanes that do not need the factor never enter it, a lane whose march is over leaves EXEC.
 *
 * `settled` (FLAG_SHADOW_SETTLED, wave-uniform): the march of a lane also ends once res <= 0.  The factor returned is
 * maxf(res, 0), and from res <= 0 on every further step can only keep it there: res' = minf(res, v) = (res < v ? res : v)
 * is <= 0 again for every v that is not NaN, and v = 50 s / t is NaN only for a non-finite s or t, or for 0 / 0 — which
 * the host rules out before it sets the flag (lol_gpu.hip, shadow_settle_ok: every scene constant, light and the camera
 * finite and below 10^15, so nothing the 128 steps can reach overflows; t returns to exactly 0 only at the ray's own
 * origin, where s is what it was on the first step, and had that been 0 res would have been NaN from the first step on,
 * never <= 0).  The reference goes on until res < -1 or t > L; a ray that grazes along just inside a surface does so for
 * all 128 steps while its 63 neighbours wait.  Same pixels; the shadow step counts of lol_gpu_debug shrink.
 *
 * Sdf::ASSUME_SETTLED (the specialised kernel's fast pipeline, which the kernel only enters under FLAG_SHADOW_SETTLED): the
 * running minimum is ONE v_min_f32 instead of the compare + select of minf_ (float.h:6: the SECOND operand on NaN or
 * equal).  The two differ in the sign of a zero result — which ends the march and comes out as maxf_(+-0, 0) = +0 — and for
 * a NaN quotient, which under the flag's conditions is 0 / 0 of a first step with s == +-0 alone: t stays 0, the point never
 * moves, the reference's factor is NaN for all 128 steps and comes out 0; here res stays 1 for the same 128 steps and
 * `t == 0` names the case afterwards (a t that RETURNED to 0 did so on a step with 50 s / t = -50: the factor is 0 as well).
 * Round 6: +0.5 % (profiles/r6_ab1.txt).
 */
template <bool COUNT, class Sdf>
__device__ __forceinline__ float soft_shadow(Sdf& sdf, V3 p, V3 dir, float max_dist, u32& steps, bool needed, bool settled) {
	constexpr bool VMIN = Sdf::ASSUME_SETTLED;
	V3 ro = add(p, dir);
	float res = 1.f, t = 0.f;
	/* res < -1 (naive_renderer.c:85) is res <= the float below -1 for every res that is not NaN (and NaN fails both) */
	const float stop = (VMIN || settled) ? 0.f : -0x1.000002p+0f;
	if (needed) {
		for (int i = 0; i < 128; i++) {
			V3 q = add(ro, scale(dir, t));
			float s;
			sdf.eval_dist(q, s);                         /* (the distance alone: see Interp::eval_dist / lol_codegen.hip, emit_sdf) */
			/* The quotient 50 s / t (11 instructions, one of them v_rcp) is only needed where it can lower the minimum.  In the fast
			 * pipeline (VMIN: every lane still marching has 0 < res <= 1, never NaN, and 0 <= t < 10^17), with a = 50 s and
			 * b = fl(res t):
			 *   a > b  =>  a >= the float above b  >  b + ulp(b) / 2  >=  res t      (b is res t rounded to nearest: off by at most half
			 *                                                                          a spacing — denormal b and b = 0 included)
			 *          =>  a / t > res  =>  fl(a / t) >= res                           (rounding is monotone, res is a float; t = 0: +inf)
			 * and v_min leaves res as it is.  Taken per wave, when EVERY lane still marching has a > b (a NaN fails it): one multiply,
			 * a compare and a scalar branch where it does not apply.  Round 6, six repetitions per library on one box
			 * (profiles/r6_ab_division_skip.txt): C3 +4.2 % (new view +3.2 %), C4 +4.0 %, C2 -3.3 % (rays skimming a plane keep
			 * v == res: they divide every step and pay for the test); a cool-down after a step that did divide made all three worse.
			 * (Round 3 measured a form of this at -1.2 % in the wave-uniform loops of the time, where lanes that had finished took part
			 * in the vote.)  tests/test_shadow_division_bound.py checks the implication in exact rational arithmetic. */
			const float a = 50.f * s;
			if (!VMIN || vote(!(a > res * t)) != 0) {
				asm volatile("");                        /* keep the branch */
				const float v = a / t;
				res = VMIN ? vmin_(res, v) : minf_(res, v);
			}
			t += s;
			if (COUNT) steps++;
			if (res <= stop || t > max_dist) break;
		}
		if (VMIN && t == 0.f) res = 0.f;
	}
	sdf.loop_done();
	return maxf_(res, 0.f);
}

/* get_normal, naive_renderer.c:114-125: k0=(1,-1,-1) k1=(-1,-1,1) k2=(-1,1,-1) k3=(1,1,1);
 * n = normalize(k0*s0 + (k1*s1 + (k2*s2 + k3*s3))), s_i = sdf(p + k_i*h).  The taps run as a rolled loop from
 * k3 down to k0 (one copy of the SDF code instead of four; + is commutative, so adding each new term on the
 * left of the running sum reproduces the reference's association).
 *
 * ... and the hit's object id where the march left it open (Sdf::ASK_ID_ONCE, march()): where a lane of the wave has to ask,
 * the loop takes one turn more, FIRST, at the point its march's last step evaluated — the same copy of the SDF code.  Then the
 * wave knows whether anything was hit at all: `lit` = false is FLAG_MISS_SKIP's "every ray of this wave escaped: no normal, no
 * shadows" (shade_pixel), and the taps are not taken. */
template <class Sdf>
__device__ __forceinline__ V3 normal_and_id(Sdf& sdf, V3 ro, V3 rd, const Marched& hit, V3 p, bool miss_skip, u32& id, bool& lit) {
	const float h = hit.dist / 100.f;
	const float nh = -1.f * h;       /* v3scale(k, h) multiplies; -1*h == -h bit for bit */
	V3 acc = { 0.f, 0.f, 0.f };
	const bool any_ask = Sdf::ASK_ID_ONCE && vote(hit.ask) != 0;
	id = hit.id;
	lit = true;
	if (!any_ask && miss_skip && vote(id != 0u) == 0) { lit = false; return acc; }
#pragma unroll 1      /* (unrolled — four independent evaluations in flight — it is 0.8 % slower within the 64-VGPR budget) */
	for (int k = any_ask ? 4 : 3; k >= 0; k--) {
		/* sign pattern of tap k, wave-uniform: x is + for k0,k3; y is + for k2,k3; z is + for k1,k3 */
		const bool ask = Sdf::ASK_ID_ONCE && k == 4, px = k == 0 || k == 3, py = k >= 2, pz = k == 1 || k == 3;
		V3 q = { p.x + (px ? h : nh), p.y + (py ? h : nh), p.z + (pz ? h : nh) };
		if (ask) q = add(ro, scale(rd, hit.prev));
		float s; u32 did;
		sdf.eval(q, s, did);
		if (ask) {
			if (hit.ask) id = did;
			if (miss_skip && vote(id != 0u) == 0) { lit = false; break; }
			continue;
		}
		const float ns = -s;          /* -1.f * s */
		V3 term = { px ? s : ns, py ? s : ns, pz ? s : ns };
		acc = k == 3 ? term : add(term, acc);
	}
	return lit ? normalize(acc) : acc;
}

__device__ __forceinline__ V3 lds_v3(const u32* base) {
	const float* f = reinterpret_cast<const float*>(base);
	return { f[0], f[1], f[2] };
}

/* Where the pipeline reads lights, materials and the objects' material indices from.  Small scenes: staged once per block
 * into LDS (stage_common; scene4: 276 bytes).  That staging is per ONE-WAVE block, so it cannot scale with the scene — at
 * the old capacity (64 lights, 256 materials, 1024 objects) it was 16.9 KB per wave, 9 waves per CU instead of 32 — and the
 * tables have no capacity any more (lol_scene.h).  Beyond TABLES_LDS_MAX_DWORDS (4 KB: 32 waves x 4 KB + their tiles still
 * fit the CU's 160 KB) the kernels are instantiated with TABLES_GLOBAL and read the tables where they lie: lights with
 * wave-uniform addresses, the hit's material per lane, a few dozen loads per pixel against thousands of instructions. */
constexpr u32 TABLES_LDS_MAX_DWORDS = 1024;
__host__ __device__ inline u32 table_dwords(u32 n_lights, u32 n_materials, u32 n_roots) {
	return n_lights * LIGHT_DWORDS + n_materials * MATERIAL_DWORDS + n_roots;
}
__host__ __device__ inline bool tables_in_lds(u32 n_lights, u32 n_materials, u32 n_roots) {
	return table_dwords(n_lights, n_materials, n_roots) <= TABLES_LDS_MAX_DWORDS;
}
/* dwords of LDS a block needs: the tables (when staged) + its output tile */
__host__ __device__ inline u32 common_lds_dwords(u32 n_lights, u32 n_materials, u32 n_roots) {
	return (tables_in_lds(n_lights, n_materials, n_roots) ? table_dwords(n_lights, n_materials, n_roots) : 0u) + TILE_W * TILE_H;
}

struct Pixel { V3 rgb; Hit hit; u32 shadow_steps; };      /* rgb: the clamped colour BEFORE gamma; store_pixel applies gamma and packs it into the surface's format */

/* frame row of local row r of this launch's part (the inverse: lol_gpu_part_frame_row) */
__device__ __forceinline__ int frame_row(const Launch& L, int r) {
	const int band = r / L.band_rows;
	return band * L.cycle_rows + L.offset_rows + (r - band * L.band_rows);
}

/*
 * The per-pixel body, naive_renderer.c:217-235, for the pixel this lane owns.
 * `lds` = lights | materials | root_material | out tile (already staged and synchronised); with TABLES_GLOBAL the
 * three tables are read from global memory instead (L.lights, L.materials, L.root_material) and `lds` is the tile alone.
 */
template <class Sdf, bool TABLES_GLOBAL = false, bool COUNT = true>
__device__ __forceinline__ Pixel shade_pixel(const Launch& L, Sdf& sdf, const u32* lds) {
	const u32* l_light = TABLES_GLOBAL ? L.lights : lds;
	const u32* l_mat   = TABLES_GLOBAL ? L.materials : l_light + L.n_lights * LIGHT_DWORDS;
	const u32* l_rootm = TABLES_GLOBAL ? L.root_material : l_mat + L.n_materials * MATERIAL_DWORDS;

	/* lane → pixel: wave k covers a WAVE_W x WAVE_H patch of the block's tile — or, FLAG_TILE_TABLE, the pixel the table deals
	 * this lane (in the frame by construction) */
	int x, r;                                                                       /* r: local row */
	if (L.flags & FLAG_TILE_TABLE) {
		const u32 e = lane_pixel(L);
		x = (int)(e & 0xFFFFu);
		r = (int)(e >> 16 & 0x7FFFu);
	} else {
		const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
		const int tx = wave * WAVE_W + (lane % WAVE_W), ty = lane / WAVE_W;
		int tbx, tby;
		tile_of_block(L, tbx, tby);
		x = tbx * TILE_W + tx;
		r = tby * TILE_H + ty;
		/* out-of-frame lanes shade a clamped pixel and skip the store: keeps the wave uniform */
		x = x < L.w ? x : L.w - 1;
		r = r < L.n_rows ? r : L.n_rows - 1;
	}
	const int y = frame_row(L, r);

	/* naive_renderer.c:218-221 */
	const float vx = ((float)x + .5f) / L.fw * 2.f - 1.f;
	const float vy = 1.f - ((float)y + .5f) / L.fh * 2.f;

	/* get_camera_ray with the per-frame basis hoisted, naive_renderer.c:188-190 */
	const V3 ro = v3(L.cam.origin), cdir = v3(L.cam.dir);
	V3 rd = add(scale(v3(L.cam.right), vx * L.cam.width), scale(v3(L.cam.up), vy * L.cam.height));
	rd = normalize(add(rd, cdir));

	const Marched marched = march<COUNT>(sdf, ro, rd, L.max_steps, (L.flags & FLAG_FIRST_STEP) != 0u, L.first_dist, L.first_id);

	/*
	 * A ray that escaped is shaded with material #0 (naive_renderer.c:103-112).  When the host has checked
	 * that material #0 has diffuse == specular == 0, shininess >= 0 and every light intensity is finite
	 * (lol_gpu.hip: miss_skip_ok), each light's two terms are (finite) * (+-0) = +-0 whatever the normal and
	 * the shadow factor turn out to be, the running sum stays +0, and get_light() returns exactly
	 * clamp(ambient_color * material.ambient).  So a wave in which EVERY lane escaped skips the 4 normal taps
	 * and the shadow marches (about 13 % of all SDF evaluations on scene4) and falls through to the same
	 * `0 + ambient*mat.ambient` expression; a wave with at least one hit runs everything for all its lanes.
	 */
	const V3 p = add(ro, scale(rd, marched.dist));
	bool lit;
	Hit hit = { marched.dist, 0u, marched.steps };
	const V3 n = normal_and_id(sdf, ro, rd, marched, p, (L.flags & FLAG_MISS_SKIP) != 0u, hit.id, lit);

	/* get_material, naive_renderer.c:103-112 (per-lane table lookups) */
	u32 mid = hit.id ? l_rootm[hit.id - 1] : 0u;
	const float* m = reinterpret_cast<const float*>(l_mat + mid * MATERIAL_DWORDS);
	const float shininess = m[0];
	const V3 m_diff = { m[1], m[2], m[3] }, m_spec = { m[4], m[5], m[6] }, m_amb = { m[7], m[8], m[9] };

	/* get_light, naive_renderer.c:129-175 */
	V3 total = { 0.f, 0.f, 0.f };
	u32 shadow_steps = 0;
	if (lit) {
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
			float shadow = soft_shadow<COUNT>(sdf, p, light_dir, light_dist, shadow_steps, needed, (L.flags & FLAG_SHADOW_SETTLED) != 0u);

			V3 refl = sub(scale(n, 2.f * dot(light_dir, n)), light_dir);
			V3 Id = mul(scale(lds_v3(lp + 3), shadow * di), m_diff);
			total = add(total, Id);
			/* The specular term is I * (shadow * si) * colour.  Under FLAG_DARK_SKIP's conditions (finite intensities and colours,
			 * shininess >= 0: powf of a base in [0, 1] is finite) a lane with di == 0 or shadow == 0 contributes +-0 whatever the
			 * powf returns, and +-0 leaves the running sum as it is: a wave in which no lane has both skips the call (the lanes that
			 * do not need it take si = 0, which gives the same +-0). */
			float si = 0.f;
			if (!(L.flags & FLAG_DARK_SKIP) || vote(needed && shadow != 0.f) != 0)      /* (needed: di > 0, and not an escaped lane under FLAG_MISS_SKIP — its material's specular colour is 0) */
				si = di * powf_glibc(clampf_(dot(refl, camera_dir), 0.f, 1.f), shininess);
			V3 Is = mul(scale(lds_v3(lp + 6), shadow * si), m_spec);
			total = add(total, Is);
		}
	}
	total = add(total, mul(v3(L.ambient), m_amb));
	/* v3clamp: max(min(v, 1), 0) — NaN → 1 (vec.h:63-65) */
	V3 c = { maxf_(minf_(total.x, 1.f), 0.f), maxf_(minf_(total.y, 1.f), 0.f), maxf_(minf_(total.z, 1.f), 0.f) };

	return { c, hit, shadow_steps };      /* gamma + colorf_to_pixfmt: store_pixel */
}

/* Pack the lane's colour for the surface, write it (and the optional diagnostics).  Every thread of the block must call this. */
template <bool TABLES_GLOBAL = false>
__device__ __forceinline__ void store_pixel(const Launch& L, const Pixel& P, u32* lds) {
	const LaunchTail T = launch_tail(L);
	/* colorf_to_pixfmt, renderer.h:17-22: Uint8 r = colorf.x * 255 …; SDL_MapRGB(fmt, r, g, b) for a non-palettised format
	 * (SDL2 src/video/SDL_pixels.c): (r >> Rloss) << Rshift | (g >> Gloss) << Gshift | (b >> Bloss) << Bshift | Amask.
	 * XRGB8888 = shifts 16 / 8 / 0, no loss, no alpha. */
	/* gamma, naive_renderer.c:231-232: the float colour through powf (always for the diagnostics), the 8 bits through the
	 * proven table when the launch has one */
	const bool by_table = (L.flags & FLAG_GAMMA_TABLE) != 0u;
	V3 post = P.rgb;
	if (!by_table || T.dbg_rgb) {
		const float g = 1.f / 2.2f;
		post = { powf_glibc(P.rgb.x, g), powf_glibc(P.rgb.y, g), powf_glibc(P.rgb.z, g) };
	}
	u32 r8, g8, b8;
	if (by_table) { r8 = gamma_u8_table(P.rgb.x, T.gamma_table); g8 = gamma_u8_table(P.rgb.y, T.gamma_table); b8 = gamma_u8_table(P.rgb.z, T.gamma_table); }
	else { r8 = (u32)(post.x * 255.f) & 0xFFu; g8 = (u32)(post.y * 255.f) & 0xFFu; b8 = (u32)(post.z * 255.f) & 0xFFu; }
	const u32 px = (r8 >> (T.fmt_loss & 0xFFu)) << (T.fmt_shift & 0xFFu) |
	               (g8 >> (T.fmt_loss >> 8 & 0xFFu)) << (T.fmt_shift >> 8 & 0xFFu) |
	               (b8 >> (T.fmt_loss >> 16 & 0xFFu)) << (T.fmt_shift >> 16 & 0xFFu) | T.fmt_amask;
	u32* l_tile = TABLES_GLOBAL ? lds : lds + L.n_lights * LIGHT_DWORDS + L.n_materials * MATERIAL_DWORDS + L.n_roots;
	if (L.flags & FLAG_TILE_TABLE) {
		/* the pixels of this wave lie where the table dealt them: every lane stores its own (4 bytes; the tables trade the
		 * idle memory system for lanes that finish together — lol_gpu.hip, "pixels dealt by cost") */
		const u32 e = lane_pixel(L);
		const int gx = (int)(e & 0xFFFFu), gr = (int)(e >> 16 & 0x7FFFu);
		u32* cost;
		unsigned short* pixel_cost;
		u32 stride;
#if defined(__HIP_DEVICE_COMPILE__)
		{
			typedef const __attribute__((address_space(4))) Launch* kernarg_ptr;
			kernarg_ptr K = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
			asm volatile("" : "+s"(K));
			cost = K->tile_cost;
			stride = K->tile_stride;
			pixel_cost = K->pixel_cost;
		}
#else
		cost = L.tile_cost;
		stride = L.tile_stride;
		pixel_cost = L.pixel_cost;
#endif
		if (!(e & LANE_PADDING)) {
			const unsigned long long o = (unsigned long long)gr * L.w + gx;
			if (T.dbg_rgb) { T.dbg_rgb[o * 3 + 0] = post.x; T.dbg_rgb[o * 3 + 1] = post.y; T.dbg_rgb[o * 3 + 2] = post.z; }
			if (T.dbg_hit_dist) T.dbg_hit_dist[o] = P.hit.dist;
			if (T.dbg_hit_id) T.dbg_hit_id[o] = P.hit.id;
			if (T.dbg_steps) T.dbg_steps[o] = (P.hit.steps & 0xFFFFu) | (P.shadow_steps << 16);
			T.dst[(unsigned long long)gr * T.pitch_px + gx] = px;
			if (pixel_cost) {                                   /* what the pixels are dealt to waves by: the evaluations this one needed */
				const u32 c = P.hit.steps + P.shadow_steps;
				pixel_cost[o] = (unsigned short)(c < 0xFFFFu ? c : 0xFFFFu);
			}
		}
		/* what this wave cost, for the order of the NEXT frames' waves (lol_gpu.hip, "longest tiles first"): how long it ran —
		 * shader-clock ticks since start_tile_clock() (parked in the first word of the unused output tile), as 32 x log2 (5
		 * fraction bits: steps of 2 %; waves shorter than 1024 ticks all count 0) — at most 21 * 32 + 31 = 703, one bucket of
		 * the host's sort each */
		if (cost && threadIdx.x == 0) {
			const u32 dt = (u32)__builtin_readcyclecounter() - l_tile[0];
			const u32 le = 31u - (u32)__builtin_clz(dt | 1u);
			cost[tile_slot(blockIdx.x, stride)] = le < 10u ? 0u : ((le - 10u) << 5 | ((dt >> (le - 5u)) & 31u));
		}
		return;
	}
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tx = wave * WAVE_W + (lane % WAVE_W), ty = lane / WAVE_W;
	int bx, by;
	tile_of_block(L, bx, by);
	const int gx = bx * TILE_W + tx, gr = by * TILE_H + ty;
	if (gx < L.w && gr < L.n_rows) {
		unsigned long long o = (unsigned long long)gr * L.w + gx;
		if (T.dbg_rgb) { T.dbg_rgb[o * 3 + 0] = post.x; T.dbg_rgb[o * 3 + 1] = post.y; T.dbg_rgb[o * 3 + 2] = post.z; }
		if (T.dbg_hit_dist) T.dbg_hit_dist[o] = P.hit.dist;
		if (T.dbg_hit_id) T.dbg_hit_id[o] = P.hit.id;
		if (T.dbg_steps) T.dbg_steps[o] = (P.hit.steps & 0xFFFFu) | (P.shadow_steps << 16);
	}
	/* through LDS so the block stores whole row segments (64 bytes each with the default 16x4 patch) */
	l_tile[ty * TILE_W + tx] = px;
	__syncthreads();
	const int sx = threadIdx.x % TILE_W, sy = threadIdx.x / TILE_W;
	const int ox = bx * TILE_W + sx, orow = by * TILE_H + sy;
	if (ox < L.w && orow < L.n_rows)
		T.dst[(unsigned long long)orow * T.pitch_px + ox] = l_tile[sy * TILE_W + sx];
}

/* FLAG_TILE_TABLE: when this block's wave started, parked in the first word of its (still unused) output tile in LDS rather
 * than in a register for the whole kernel; store_pixel turns it into the tile's cost.  Call once, before the first shade_pixel
 * (a wave that shades again through the plain path is timed over both passes: that is what it costs). */
template <bool TABLES_GLOBAL = false>
__device__ __forceinline__ bool start_tile_clock(const Launch& L, u32* lds) {
	if (L.flags & FLAG_TILE_TABLE) {
		/* false: every lane of this wave only fills up a region at the frame's edge — nothing to shade */
		if (vote((lane_pixel(L) & LANE_PADDING) == 0u) == 0) return false;
		u32* l_tile = TABLES_GLOBAL ? lds : lds + L.n_lights * LIGHT_DWORDS + L.n_materials * MATERIAL_DWORDS + L.n_roots;
		if (threadIdx.x == 0) l_tile[0] = (u32)__builtin_readcyclecounter();
	}
	return true;
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

/* Generic kernel (ahead of time): the SDF is interpreted from the macro-op list in global memory (scalar loads);
 * LDS holds lights | materials | root_material | out tile like in the specialised kernel.  KIND != 0 selects the
 * proven fast sqrt (the host launches that instantiation only after the exhaustive check passed on the device); a
 * wave that fed it a squared length outside its proven domain shades its pixels again with the plain interpreter,
 * as in the specialised kernel. */
template <int SSIZE, int KIND, bool TABLES_GLOBAL = false>
__global__ __launch_bounds__(BLOCK)
void render_interp(const Launch L) {
	extern __shared__ u32 lds[];
	if constexpr (!TABLES_GLOBAL) {
		stage_common(L, lds);
		__syncthreads();
	}
	if (!start_tile_clock<TABLES_GLOBAL>(L, lds)) return;
	Interp<SSIZE, KIND> sdf{ L.ops, L.n_ops, {}, 0u };
	Pixel P = shade_pixel<Interp<SSIZE, KIND>, TABLES_GLOBAL>(L, sdf, lds);
	if (KIND != 0 && unproven(sdf)) {
		Interp<SSIZE, 0> exact{ L.ops, L.n_ops, {}, 0u };
		P = shade_pixel<Interp<SSIZE, 0>, TABLES_GLOBAL>(L, exact, lds);
	}
	store_pixel<TABLES_GLOBAL>(L, P, lds);
}

/* Diagnostic: the scene SDF alone — sdf() of naive_renderer.c:31-44 — at arbitrary points, one per lane, through the
 * same Sdf policies the frames use (lol_gpu_sdf_batch).  `fast` / `exact` as in the render kernels: a wave that fed
 * the proven fast sqrt a squared length outside its domain evaluates again with the plain SDF. */
template <class SdfFast, class SdfExact>
__device__ __forceinline__ void sdf_points(SdfFast& fast, SdfExact& exact, bool have_fast, const float* pts, float* dist, u32* id, u32 n) {
	const u32 i = blockIdx.x * 64u + threadIdx.x, j = i < n ? i : n - 1;
	const V3 p = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
	float d; u32 k;
	if (have_fast) {
		fast.eval(p, d, k);
		if (unproven(fast)) exact.eval(p, d, k);
	} else {
		exact.eval(p, d, k);
	}
	if (i < n) { dist[i] = d; id[i] = k; }
}

template <int SSIZE, int KIND>
__global__ __launch_bounds__(64)
void sdf_points_interp(const u32* mops, u32 n_mops, const float* pts, float* dist, u32* id, u32 n) {
	Interp<SSIZE, KIND> fast{ mops, n_mops, {}, 0u };
	Interp<SSIZE, 0> exact{ mops, n_mops, {}, 0u };
	sdf_points(fast, exact, KIND != 0, pts, dist, id, n);
}

}  // namespace lol
