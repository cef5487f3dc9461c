/*
 * lol_gpu_internal.h — what the translation units of liblol_gpu.so share: the context (struct lol_gpu), the scene compiler's job,
 * the proven shortcuts (FastPaths), and the few functions that cross from one unit to another.  Nothing here is part of the
 * C ABI (include/lol_gpu.h); everything declared between the visibility pragmas stays inside the library.
 *
 *   lol_gpu.hip      the context and the C ABI around it: upload, the tiers of the scene compiler, frames (device and host
 *                    surfaces, frames in flight), the first march step, diagnostics
 *   lol_proofs.hip   the exhaustive on-device proofs of the fast paths, the gamma table, lol_gpu_verify_*
 *   lol_codegen.hip  exact culling (bounds, plan), the interpreter's macro-op lists, the scene -> HIP source generator,
 *                    hipRTC + the code-object cache + the long-branch trip-wire, lol_gpu_compile_offline
 *   lol_sched.hip    the order in which a frame's tiles are handed out: fixed orders and their trials, longest tiles first,
 *                    pixels dealt by cost (tables and the kernels that make them)
 *   lol_multi.hip    several devices behind the boundary (RCCL exchange)
 */
#pragma once
#include "lol_gpu.h"
#include "lol_gpu_diag.h"
#include "lol_gpu_testing.h"
#include "lol_kernel.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <thread>
#include <functional>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>


static_assert(sizeof(lol_light) == lol::LIGHT_DWORDS * 4, "lol_light layout");
static_assert(sizeof(lol_material) == lol::MATERIAL_DWORDS * 4, "lol_material layout");
static_assert(sizeof(lol_frame_camera) == sizeof(lol::Cam), "lol_frame_camera layout");

#pragma GCC visibility push(hidden)

/* The scene compiler runs on a thread of its own with a LARGE stack: the compiler inside hipRTC recurses over the long
 * dependent chains of a big scene's straight-line SDF — a field of 3000 objects overflowed the usual 8 MB in the hipRTC
 * that PyTorch bundles (ROCm 7.0's; the system's 7.2 survived) and took the process down.  1 GB of address space; only
 * the pages really used are ever committed. */
struct BigStackThread {
	pthread_t t{};
	bool started = false;
	std::function<void()> fn;
	static void* entry(void* self) { static_cast<BigStackThread*>(self)->fn(); return nullptr; }
	bool start(std::function<void()> f) {
		fn = std::move(f);
		pthread_attr_t attr;
		if (pthread_attr_init(&attr) != 0) return false;
		(void)pthread_attr_setstacksize(&attr, (size_t)1 << 30);
		started = pthread_create(&t, &attr, entry, this) == 0;
		if (!started) {                                 /* (no gigabyte of address space to be had: the default stack) */
			pthread_attr_t plain;
			if (pthread_attr_init(&plain) == 0) { started = pthread_create(&t, &plain, entry, this) == 0; pthread_attr_destroy(&plain); }
		}
		pthread_attr_destroy(&attr);
		return started;
	}
	bool joinable() const { return started; }
	void join() { if (started) { pthread_join(t, nullptr); started = false; } }
};

/* What the device has proven about its own arithmetic, as far as the scene at hand needs it (lol_proofs.hip: prove_fast_paths);
 * the generator (lol_codegen.hip) and the interpreter's lists use a shortcut only where this says so. */
struct FastPaths {
	int sqrt_kind = 0;                    /* 0 plain sqrtf; 1 sqrt_pm, 2 sqrt_gs, 3 sqrt_r2 — proven on this device */
	bool sqrt_tiny_ok = false;            /* ... and NaN-or-tiny below its domain: spheres may drop the range tracker (sd_sphere_fast_nr) */
	std::vector<float> div_ok;            /* smoothness constants k whose smin_h_fast verified */
	std::vector<float> div_nf_ok;         /* ... and verified without v_div_fixup as well (smin_h_fast<false>) */
	bool gamma_ok = false;                /* the gamma table route == the powf route for every float in [0, 1] (verify_gamma_kernel) */
	bool has(float k) const {
		for (float v : div_ok) if (memcmp(&v, &k, 4) == 0) return true;
		return false;
	}
	bool has_nf(float k) const {
		for (float v : div_nf_ok) if (memcmp(&v, &k, 4) == 0) return true;
		return false;
	}
};

struct OwnedProgram;
/* One run of the scene compiler on a host thread (tiered start-up: start_specialise / finish_specialise below). */
struct SpecJob {
	std::mutex mu;
	std::condition_variable cv;
	bool done = false, ok = false;
	std::vector<char> code;
	std::string log, note;
	std::chrono::steady_clock::time_point started;
	double compile_ms = 0;
	/* what the run compiles — its own copies: the context may take another scene meanwhile */
	std::shared_ptr<OwnedProgram> prog;
	std::shared_ptr<FastPaths> fast;
	std::string arch;
	bool cull = true;
	int form = 0;                        /* SpecForm: by size, or the form a tier asks for */
	BigStackThread th;
};

/* A program and the memory behind its four tables (lol_program itself only points: include/lol_scene.h). */
struct OwnedProgram {
	lol_program p{};
	std::vector<lol_op> ops;
	std::vector<lol_light> lights;
	std::vector<lol_material> materials;
	std::vector<uint32_t> root_material;
	/* all or nothing: the copies are made on the side (any of them may throw std::bad_alloc) and swapped in together, so a
	 * failed assign leaves the old program — tables AND counts — as it was */
	void assign(const lol_program& src) {
		std::vector<lol_op> o(src.ops, src.ops + src.n_ops);
		std::vector<lol_light> l(src.lights, src.lights + src.n_lights);
		std::vector<lol_material> m(src.materials, src.materials + src.n_materials);
		std::vector<uint32_t> r(src.root_material, src.root_material + src.n_roots);
		ops.swap(o); lights.swap(l); materials.swap(m); root_material.swap(r);      /* (noexcept) */
		p = src;
		p.ops = ops.data(); p.lights = lights.data(); p.materials = materials.data(); p.root_material = root_material.data();
	}
	OwnedProgram() = default;
	OwnedProgram(const OwnedProgram&) = delete;
	OwnedProgram& operator=(const OwnedProgram&) = delete;
};

struct lol_gpu {
	int          device = -1;
	hipStream_t  stream = nullptr;
	/* device tables, two sets: an upload fills the set no frame reads and flips `cur` only when every fallible step
	 * has succeeded (lol_gpu_upload_program is all-or-nothing).  Sized by the program (grown when an upload needs more). */
	uint32_t*    d_tables[2] = { nullptr, nullptr };   /* lights | materials | root_material, as dwords */
	size_t       tables_cap[2] = { 0, 0 };             /* ... dwords allocated */
	uint32_t*    d_mops[2] = { nullptr, nullptr };     /* the interpreter's two macro-op lists (lol_kernel.h, Interp) */
	size_t       mops_cap[2] = { 0, 0 };               /* ... dwords allocated */
	int          cur = 0;
	uint32_t     n_mops = 0;
	OwnedProgram h_own;                  /* host copy of the uploaded program ... */
	lol_program& h_prog = h_own.p;       /* ... and its lol_program view (counts, tables, max_stack) */
	bool         have_prog = false;
	/* the surface's pixel format (lol_gpu_set_pixel_format), packed as lol::Launch wants it; default XRGB8888 */
	uint32_t     fmt_shift = 16u | 8u << 8 | 0u << 16, fmt_loss = 0, fmt_amask = 0;
	/* host-surface path */
	uint32_t*    d_frame = nullptr;      /* framebuffer for lol_gpu_render_host */
	size_t       frame_bytes = 0;
	/* lol_gpu_render_host_begin / _end: frames in flight, one device framebuffer each (sized per slot, so frames of
	 * different sizes can be in flight while the host's window is being resized); slot = frame number % PIPE_SLOTS.  The kernel
	 * of a frame with a NEW view goes to the next of the context's frame streams, one under the view of the frame before it
	 * follows that frame (lol_gpu_render_host_begin): consecutive frames of a moving camera overlap */
	static constexpr int PIPE_SLOTS = 4;
	uint32_t*    d_pipe[PIPE_SLOTS] = { nullptr, nullptr, nullptr, nullptr };
	size_t       pipe_bytes[PIPE_SLOTS] = { 0, 0, 0, 0 };
	hipStream_t  copy_stream = nullptr;
	hipEvent_t   pipe_rendered[PIPE_SLOTS] = { nullptr, nullptr, nullptr, nullptr }, pipe_copied[PIPE_SLOTS] = { nullptr, nullptr, nullptr, nullptr };
	hipStream_t  pipe_stream[PIPE_SLOTS] = { nullptr, nullptr, nullptr, nullptr };   /* the stream the slot's kernel was queued on */
	int          pipe_w[PIPE_SLOTS] = { 0, 0, 0, 0 }, pipe_h[PIPE_SLOTS] = { 0, 0, 0, 0 };
	unsigned     pipe_begun = 0, pipe_ended = 0;
	unsigned     pipe_rr = 0;                    /* rotation of the kernels' streams: advanced by every frame whose view is new */
	hipStream_t  pipe_last_stream = nullptr;     /* ... a frame under the view of the frame before it follows that frame on its stream */
	lol_frame_camera pipe_last_cam{};
	int          pipe_last_geom[3] = { 0, 0, 0 };
	int          want_spec = 1;
	bool         want_second_tier = true; /* lol_gpu_set_specialize(ctx, 5) says no */
	uint32_t     spec_max_ops = 0;       /* lol_gpu_set_specialize_max_ops: 0 = LOL_SPEC_MAX_OPS */
	hipModule_t  spec_module = nullptr;
	hipFunction_t spec_fn = nullptr;
	hipFunction_t spec_steps_fn = nullptr;   /* lol_render_spec_steps, the same pipeline with the per-lane step counters (generate_source);
	                                          * nullptr where the module holds one kernel only: spec_fn counts then */
	hipFunction_t spec_sdf_fn = nullptr; /* lol_sdf_spec of the same module (lol_gpu_sdf_batch) */
	std::string  spec_log;
	hipModule_t  spec_module_old = nullptr;   /* the first tier's module once the second has taken over: frames in flight may still run it, so it
	                                           * stays loaded until the next upload (which drains the device) or the end of the context */
	bool         second_tier_pending = false; /* when the running job's kernel is in use, the INLINED form is compiled next (start_specialise) */
	bool         second_tier_running = false; /* `job` is that second run */
	SpecJob*     job = nullptr;       /* the scene compiler's run for the CURRENT program, until its module is swapped in */
	std::vector<SpecJob*> old_jobs;   /* runs for programs since replaced: joined when they have finished */
	int          spec_state = 0;         /* 0 no specialised kernel wanted / possible, 1 compiling, 2 in use, -1 failed */
	double       spec_compile_ms = 0;    /* how long the last finished run took (wall clock of its thread) */
	std::string  spec_key;               /* FNV-1a of the code object the frames run (lol_gpu_kernel_key) */
	std::string  interp_key;             /* ... and of {this build, the uploaded macro-op lists} for the interpreter */
	int          fail_uploads = 0;       /* lol_gpu_testing_fail_uploads: that many uploads still fail at the copy */
	int          fail_first_tier = 0;    /* lol_gpu_testing_fail_first_tier: that many out-of-line first runs of the scene compiler "fail" */
	int          want_fast = 1;          /* allow the proven-exact shortcuts in the specialised kernel */
	unsigned     want_skips = 7;         /* exact skips allowed when the program qualifies: bit 0 escaped waves, 1 zero incidence, 2 settled shadows */
	int          want_cull = 1;          /* allow the exact culling of top-level objects (plan_culling) */
	bool         miss_skip = false;      /* the uploaded program qualifies (miss_skip_ok) */
	bool         dark_skip = false;      /* the uploaded program qualifies (dark_skip_ok) */
	bool         shadow_settle = false;  /* the uploaded program qualifies (shadow_settle_ok) */
	bool         finite_scene = false;   /* shadow_settle_ok(program), whatever the switches say: the interpreter's no-fixup list may run */
	int          interp_sqrt_kind = 0;   /* fast sqrt of the interpreter: 3 (sqrt_r2) when proven and allowed, else 0 */
	int          sqrt_verified = -1;     /* -1 not run, 0 none proven, else the lol::sqrt_fast KIND proven on this device */
	bool         sqrt_tiny_ok = false;   /* the second counter of that run was 0 too (sd_sphere_fast_nr) */
	struct DivProof { uint32_t k_bits; bool ok, no_fixup_ok; };
	std::vector<DivProof> div_verified;  /* per smoothness constant: smin_h_fast proven / proven without v_div_fixup too */
	unsigned long long* d_bad = nullptr; /* mismatch counter of the verification kernels */
	float*       d_gamma = nullptr;      /* gamma thresholds (lol_kernel.h, gamma_u8_table): GAMMA_LEVELS + 1 floats */
	int          gamma_verified = -1;    /* -1 not run, 1 the table route == the powf route for every float in [0, 1] on this device, 0 not */
	bool         gamma_table = false;    /* frames of the current scene use it (want_fast at the last upload) */
	/* lol_gpu_set_tile_order.  AUTO: the first frames of a (scene, size, partition) alternate between the two orders, each
	 * between two events on its launch stream; later frames collect the finished ones without waiting (tile_auto_*) */
	struct TileAuto {
		int   mode = LOL_GPU_TILES_LPT;
		int   chosen = LOL_GPU_TILES_ROWS;       /* order outside trials */
		bool  deciding = false;
		int   key[6] = { 0, 0, 0, 0, 0, 0 };    /* w, h, max_steps, band_rows, cycle_rows, program generation */
		int   issued = 0, harvested = 0, decisions = 0;
		static constexpr int SKIP = 6, TOTAL = SKIP + 2 * LOL_GPU_TILE_TRIALS;
		/* trial i: untimed row-order frames first, then pairs (rows, columns), (columns, rows), (rows, columns) ... */
		static constexpr int order_of_trial(int i) { return i < SKIP ? LOL_GPU_TILES_ROWS : ((((i - SKIP) >> 1) ^ (i - SKIP)) & 1); }
		hipEvent_t ev[2 * TOTAL] = {};          /* start / end of trial frame i at [2i], [2i + 1]; created on first use */
		bool  have_events = false;
		float ms[TOTAL] = {};
		float typical[2] = { 0.f, 0.f };
		/* after the decision: a timed pair (order in use, other order) every MONITOR_PERIOD frames (tile_order_for_frame) */
		static constexpr unsigned MONITOR_PERIOD = 8, MONITOR_WINDOW = 5;
		unsigned mon_frames = 0, mon_n = 0, swaps = 0;
		bool  mon_pending = false;
		float mon_ratio[MONITOR_WINDOW] = {};
	} tiles;
	int          generation = 0;         /* uploads so far */
	/* the primary march's first step (first_step): sdf(camera origin) of program `first_gen`, kept while the camera stays where it is */
	float        first_origin[3] = { 0, 0, 0 };
	int          first_gen = -1;
	float        first_dist = 0;
	uint32_t     first_id = 0;
	std::vector<float> first_stack;
	int          kernel_epoch = 0;       /* changes whenever the frames' kernel does: an upload (the interpreter takes over), each swap of
	                                      * finish_specialise — what a kernel's tiles cost says nothing about another kernel's */
	/* LOL_GPU_TILES_LPT: longest tiles first ("longest tiles first" below).  One SET of tables per stream that launches frames
	 * of a repeated view (lpt_table_for_frame): everything about a set happens on its home stream, so frames, the costs they
	 * write and the sorts that read them are ordered by that stream itself — and frames in flight on several streams
	 * (lol_gpu_set_frames_in_flight, lol_gpu_render_host_begin) each keep their schedule. */
	struct TileLpt {
		int      key[7] = { 0, 0, 0, 0, 0, 0, 0 };   /* w, h, max_steps, band_rows, cycle_rows, offset_rows, kernel_epoch */
		uint32_t n_tiles = 0;
		uint32_t* d_order[2] = { nullptr, nullptr };   /* tile_order tables: frames read [cur], a sort writes [cur ^ 1] */
		uint32_t* d_cost = nullptr;          /* what the blocks of the last frame cost, by launch position */
		uint32_t* d_keys = nullptr;          /* the sort's snapshot of the costs (bucket numbers) */
		uint32_t* d_hist = nullptr;          /* 2 x LPT_BUCKETS: bucket sizes, then the scatter's cursors */
		uint32_t* d_lanes = nullptr;         /* the pixel table: 64 entries per wave slot ("pixels dealt by cost") */
		unsigned short* d_pixel_cost = nullptr;   /* what every pixel of the view's first frame cost */
		size_t   lanes_cap = 0, pixels_cap = 0;
		size_t   cap = 0;                    /* tiles the buffers hold */
		int      cur = 0;
		unsigned frames = 0, sorts = 0;      /* frames launched with this key; sorts done */
		hipStream_t home = nullptr;          /* the stream these tables live on; nullptr = the set is free */
		bool     launched = false;           /* a frame (or a table kernel) has been queued on `home` through these tables since the
		                                      * stream last ran dry: they may be in use whatever `key` says (cleared where that stream is
		                                      * waited for: before the tables are freed, or change hands) */
		unsigned long long stamp = 0;        /* when the set was last used (the least recently used one makes room for a fifth stream) */
		lol_frame_camera cam_epoch{};        /* the view the tables' costs belong to */
		hipEvent_t done = nullptr;           /* recorded behind every frame queued through these tables (lpt_frame_queued) */
		bool     done_recorded = false;
	};
	static constexpr int LPT_SETS = 4;
	TileLpt      lpt[LPT_SETS];
	unsigned long long lpt_clock = 0;
	unsigned     lpt_homeless = 0;       /* consecutive still frames on a stream that has no set while all sets are taken */
	int          lpt_last_set = -1;      /* the set the last frame went through, or -1: it was launched in a fixed order (lol_gpu_tile_order) */
	unsigned     lpt_sorts = 0;          /* sorts of all sets so far (lol_gpu_tile_order) */
	/* the frame launched before this one, on whatever stream: its key and view (lpt_table_for_frame: `still`) */
	int          lpt_last_key[7] = { 0, 0, 0, 0, 0, 0, 0 };
	lol_frame_camera lpt_last_cam{};
	bool         lpt_have_last = false;
	/* Frames in flight (lol_gpu_set_frames_in_flight): frames launched with stream == NULL go round-robin over the first
	 * n_frame_streams of these; [0] is `stream`.  lol_gpu_render_host_begin's slots use them too. */
	static constexpr int MAX_FRAME_STREAMS = 4;
	hipStream_t  frame_streams[MAX_FRAME_STREAMS] = { nullptr, nullptr, nullptr, nullptr };
	int          n_frame_streams = 1;
	unsigned     frame_rr = 0;
	char         err[512] = { 0 };
	char         kernel_name[64] = "render_interp";
};

/* an A/B switch from the environment, honoured only beside LOL_GPU_TUNING=1 and recorded when it is (lol_gpu.hip) */
const char* lol_gpu_internal_tuning_env(const char* name);
static inline const char* tuning_env(const char* name) { return lol_gpu_internal_tuning_env(name); }

int fail(lol_gpu* ctx, int status, const char* what, hipError_t e = hipSuccess);
#define LOL_HIP(ctx, call)                                                        \
	do {                                                                          \
		hipError_t e_ = (call);                                                   \
		if (e_ != hipSuccess) return fail((ctx), LOL_GPU_ERR_HIP, #call, e_);     \
	} while (0)

/* operand-stack entries under the accumulator a program needs → the instantiation that has them */
constexpr int interp_stack_class(uint32_t max_stack) {
	const uint32_t need = max_stack > 1 ? max_stack - 1 : 1;      /* the accumulator holds the top entry */
	return need <= 1 ? 1 : need <= 3 ? 3 : need <= 7 ? 7 : need < (uint32_t)lol::MOP_DEEP_FROM ? lol::MOP_DEEP_FROM - 1 : lol::MOP_DEEP_SLOTS;
}

/*
 * The kernel shades every pixel with the fast SDF; a wave in which any squared length fell outside
 * [SQRT_FAST_MIN, inf) (a sample within 2^-48 of a sphere centre, or an overflow) shades its pixels
 * again with the plain SDF, so the shortcut never decides a result.
 */
/* Scenes above this many ops get their SDF as an out-of-line function (emit_sdf).  Rounds 2 - 4: 256 — the inlined form took
 * 1.3 s (256 ops) to 15 s (1024) to compile against 0.35 - 2.2 s out of line, and render_prepare WAITED for the compiler.  It
 * no longer does (tiered start-up: the compiler runs on its own thread, frames render on the interpreter meanwhile), so what
 * decides now is the kernel that comes out.  Measured on MI355X in round 5 (tools/large_scene_ab.py, chains of smooth unions at
 * 1080p, profiles/r5_large_scene_ab.jsonl; inlined / out of line / interpreter, Mpixels/s): 284 ops 311 / 165 / 185 (the
 * out-of-line kernel was SLOWER than the interpreter it replaced), 504 ops 174 / 126 / 105, 1024 ops 80.6 / 60.1 / 43.2 — the
 * inlined form +88 % / +38 % / +34 % for 1.4 / 2.6 / 6.6 s of background compile (0.4 / 0.6 / 1.4 out of line) — and at 2048 ops
 * the other way round: 13.7 / 15.6 (960x540; 19 s against 3.8 s: six copies of a 300 KB function no longer pay).
 * LOL_GPU_SPEC_INLINE_MAX (a tuning switch) overrides. */
constexpr uint32_t LOL_SPEC_INLINE_MAX_OPS = 1024;
/* ... and above THIS many ops the inlined form is the scene's SECOND kernel: the out-of-line form, which hipRTC delivers 3 - 6
 * times sooner, renders until it is there (start_specialise) */
constexpr uint32_t LOL_SPEC_FIRST_TIER_INLINE_MAX_OPS = 256;
/* ... and up to THIS many ops the module holds the pipeline twice: with and without the per-lane step counters (generate_source) */
constexpr uint32_t LOL_SPEC_TWO_KERNELS_MAX_OPS = 256;
/* specialise(): larger scenes stay on the interpreter.  The scene compiler cannot be interrupted, lol_gpu_destroy has to wait
 * for it, and a second upload's run queues behind it — so what it takes on is bounded by what was MEASURED as tolerable
 * (profiles/r4_big_scene_probe.jsonl, fields of N objects on the GPU box: 5.3 s at 1320 ops, 14.6 s at 2640, 40.7 s at 5060,
 * about n^1.5: a minute at 6500, four at 16,384 — round 4's cap).  LOL_GPU_SPEC_MAX_OPS (a tuning switch) moves it. */
constexpr uint32_t LOL_SPEC_MAX_OPS = 6144;

/* which form of the scene's SDF a run of the compiler produces: by the program's size, or the one the tiers ask for */
enum SpecForm { SPEC_BY_SIZE = 0, SPEC_OUT_OF_LINE = 1, SPEC_INLINE = 2 };

/* ---- lol_proofs.hip */
FastPaths prove_fast_paths(lol_gpu* ctx, const lol_program& prog);
float smooth_sat_threshold(float k);

/* ---- lol_codegen.hip */
inline bool culling_enabled(int want) { return want != 0; }      /* (lol_gpu_set_cull) */
bool spec_out_of_line(const lol_program& P, int form = SPEC_BY_SIZE);
bool compile_spec(const lol_program& P, const FastPaths* fast, const std::string& arch, std::vector<char>& code,
                  std::string& log, std::string* src_out = nullptr, bool cull = true, int form = SPEC_BY_SIZE);
/* the interpreter's two macro-op lists for `P`, one after the other (with / without v_div_fixup in the proven blend factors);
 * false: the two differ in length (cannot happen: same records by construction) */
bool build_interp_lists(const lol_program& P, const FastPaths& fast, bool cull, std::vector<uint32_t>& lists, uint32_t& n_mops);
std::string fnv_hex(const void* data, size_t n);
extern std::mutex g_rtc_mutex;               /* one run of the scene compiler at a time (lol_gpu.hip) */

/* ---- lol_sched.hip */
struct FrameTables { const uint32_t* order; uint32_t* cost; const uint32_t* lanes; unsigned short* pixel_cost; uint32_t n_waves; };
void lpt_release(lol_gpu* ctx);
bool lpt_table_for_frame(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps, const lol_gpu_rows* R, int n_rows,
                         int block, hipStream_t s, FrameTables* out);
int tile_order_for_frame(lol_gpu* ctx, int w, int h, int max_steps, const lol_gpu_rows* R, bool diagnostics, int* trial);
void lpt_frame_queued(lol_gpu* ctx, hipStream_t s);

#pragma GCC visibility pop
