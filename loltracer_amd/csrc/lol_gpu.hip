/*
 * lol_gpu.hip — C ABI (include/lol_gpu.h) over the gfx950 render kernels (lol_kernel.h).
 *
 * Host side of the drop-in: context = {device, stream, device copy of the
 * flattened scene, the scene-specialised kernel, a device framebuffer for the
 * host-surface path}.  No CPU rendering path exists here; without a HIP device
 * every call fails.
 *
 * Two kernels render the same bits:
 *  - render_interp<STACK>   compiled ahead of time; interprets the scene's macro-op list, fetched with
 *    wave-uniform scalar loads (lol_kernel.h, Interp);
 *  - lol_render_spec        compiled by hipRTC in lol_gpu_upload_program() from
 *    lol_kernel.h + a generated SpecSdf::eval() — the scene's SDF as straight-line
 *    code with immediates (the GPU analogue of tracing_jit_renderer.dasc:76-216,
 *    whose render_prepare JIT-compiles the scene the same way).  Used when the
 *    compile succeeds (LOL_GPU_SPECIALIZE=0 or lol_gpu_set_specialize(ctx,0)
 *    keep the interpreter).
 */
#include "lol_gpu.h"
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

/* lol_kernel.h's text, embedded at build time (csrc/Makefile: lol_kernel_src.inc) for hipRTC */
#include "lol_kernel_src.inc"
/* LOL_BUILD_ID: a digest of this library's sources and compiler flags (csrc/Makefile: lol_build_id.inc) — the identity
 * of the ahead-of-time kernels (lol_gpu_kernel_key) */
#include "lol_build_id.inc"

static_assert(sizeof(lol_light) == lol::LIGHT_DWORDS * 4, "lol_light layout");
static_assert(sizeof(lol_material) == lol::MATERIAL_DWORDS * 4, "lol_material layout");
static_assert(sizeof(lol_frame_camera) == sizeof(lol::Cam), "lol_frame_camera layout");


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

namespace { struct FastPaths; }      /* (defined below, among the proofs) */
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

/*
 * Tuning switches.  Two dozen LOL_GPU_* environment variables select code paths and compiler options for A/B runs (the list:
 * INTEGRATION.md) — LOL_GPU_RTC_FLAGS appends arbitrary options to the hot kernel's compile, and "-ffp-contract=fast"
 * inherited from some shell would silently end parity with the reference.  So they are read ONLY in a process that also has
 * LOL_GPU_TUNING=1 set, every one that was read and found is recorded (lol_gpu_tuning_switches(), the scene compiler's log,
 * bench.py's `config.env`), and one that is set without LOL_GPU_TUNING=1 is reported once on stderr and ignored.  Not fenced,
 * because they change where things are kept or what is traced, never what is computed: LOL_GPU_CACHE_DIR, LOL_GPU_ROCTX.
 */
static std::mutex g_tuning_mutex;
static std::vector<std::pair<std::string, std::string>> g_tuning_seen;      /* switches that took effect: name, value */
static std::vector<std::string> g_tuning_ignored;                           /* set, but LOL_GPU_TUNING=1 was not */
static std::string g_tuning_text;

__attribute__((visibility("hidden"))) const char* lol_gpu_internal_tuning_env(const char* name) {
	const char* v = getenv(name);
	if (!v) return nullptr;
	const char* on = getenv("LOL_GPU_TUNING");
	/* (called from entry points of the C ABI, some of which hold no try block of their own: a switch that cannot be RECORDED —
	 * no memory for its name — is not honoured, and nothing is thrown) */
	try {
		std::lock_guard<std::mutex> lock(g_tuning_mutex);
		if (!(on && on[0] == '1' && !on[1])) {
			if (std::find(g_tuning_ignored.begin(), g_tuning_ignored.end(), name) == g_tuning_ignored.end()) {
				g_tuning_ignored.push_back(name);
				fprintf(stderr, "lol_gpu: %s is set but LOL_GPU_TUNING=1 is not: ignored (tuning switches are for A/B runs)\n", name);
			}
			return nullptr;
		}
		for (auto& e : g_tuning_seen) if (e.first == name) { e.second = v; return v; }
		g_tuning_seen.emplace_back(name, v);
		return v;
	} catch (...) { return nullptr; }
}
static inline const char* tuning_env(const char* name) { return lol_gpu_internal_tuning_env(name); }

extern "C" const char* lol_gpu_tuning_switches(void) {
	try {
		std::lock_guard<std::mutex> lock(g_tuning_mutex);
		g_tuning_text.clear();
		for (const auto& e : g_tuning_seen) { if (!g_tuning_text.empty()) g_tuning_text += ' '; g_tuning_text += e.first + "=" + e.second; }
		return g_tuning_text.c_str();
	} catch (...) { return "(out of memory)"; }
}

namespace {

int fail(lol_gpu* ctx, int status, const char* what, hipError_t e = hipSuccess) {
	if (ctx) {
		if (e != hipSuccess) snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, hipGetErrorString(e));
		else snprintf(ctx->err, sizeof ctx->err, "%s", what);
	}
	return status;
}

#define LOL_HIP(ctx, call)                                                        \
	do {                                                                          \
		hipError_t e_ = (call);                                                   \
		if (e_ != hipSuccess) return fail((ctx), LOL_GPU_ERR_HIP, #call, e_);     \
	} while (0)

template <int SSIZE>
hipError_t launch_sdf_interp(const uint32_t* mops, uint32_t n_mops, const float* pts, float* dist, uint32_t* id, uint32_t n,
                             hipStream_t s, int sqrt_kind) {
	dim3 grid((n + 63) / 64);
	if (sqrt_kind == 3) hipLaunchKernelGGL((lol::sdf_points_interp<SSIZE, 3>), grid, dim3(64), 0, s, mops, n_mops, pts, dist, id, n);
	else                hipLaunchKernelGGL((lol::sdf_points_interp<SSIZE, 0>), grid, dim3(64), 0, s, mops, n_mops, pts, dist, id, n);
	return hipGetLastError();
}

template <int SSIZE, bool TABLES_GLOBAL = false>
hipError_t launch_interp(const lol::Launch& L, dim3 grid, size_t lds, hipStream_t s, int sqrt_kind) {
	if (sqrt_kind == 3) hipLaunchKernelGGL((lol::render_interp<SSIZE, 3, TABLES_GLOBAL>), grid, dim3(lol::BLOCK), lds, s, L);
	else                hipLaunchKernelGGL((lol::render_interp<SSIZE, 0, TABLES_GLOBAL>), grid, dim3(lol::BLOCK), lds, s, L);
	return hipGetLastError();
}

/* operand-stack entries under the accumulator a program needs → the instantiation that has them */
constexpr int interp_stack_class(uint32_t max_stack) {
	const uint32_t need = max_stack > 1 ? max_stack - 1 : 1;      /* the accumulator holds the top entry */
	return need <= 1 ? 1 : need <= 3 ? 3 : need <= 7 ? 7 : need < (uint32_t)lol::MOP_DEEP_FROM ? lol::MOP_DEEP_FROM - 1 : lol::MOP_DEEP_SLOTS;
}


/* Conditions under which an escaped ray's colour is exactly clamp(ambient * materials[0].ambient), so
 * that waves of escaped rays may skip normal + lights (lol_kernel.h, FLAG_MISS_SKIP): material #0 has
 * diffuse == specular == 0 (either sign), shininess >= 0 and not NaN (powf(c in [0,1], s >= 0) is finite),
 * and every light intensity is finite (finite * 0 = 0, never NaN). */
bool miss_skip_ok(const lol_program& P) {
	if (P.n_materials == 0) return false;
	const lol_material& m = P.materials[0];
	const float z[6] = { m.diffuse.x, m.diffuse.y, m.diffuse.z, m.specular.x, m.specular.y, m.specular.z };
	for (float v : z) if (!(v == 0.0f)) return false;
	if (!(m.shininess >= 0.0f)) return false;
	for (uint32_t i = 0; i < P.n_lights; i++) {
		const lol_light& l = P.lights[i];
		const float f[6] = { l.diffuse_intensity.x, l.diffuse_intensity.y, l.diffuse_intensity.z,
		                     l.specular_intensity.x, l.specular_intensity.y, l.specular_intensity.z };
		for (float v : f) if (!(v - v == 0.0f)) return false;      /* inf or NaN */
	}
	return true;
}

/* Conditions for FLAG_DARK_SKIP (lol_kernel.h): with diffuse incidence exactly 0 a light contributes
 * I * (shadow * 0) * colour and I * (shadow * (0 * powf(c, shininess))) * colour, which is +-0 for any shadow in
 * [0, 1] provided I and the colours are finite and powf is finite (c in [0, 1], shininess >= 0). */
bool dark_skip_ok(const lol_program& P) {
	auto finite = [](float v) { return v - v == 0.0f; };
	for (uint32_t i = 0; i < P.n_lights; i++) {
		const lol_light& l = P.lights[i];
		const float f[6] = { l.diffuse_intensity.x, l.diffuse_intensity.y, l.diffuse_intensity.z,
		                     l.specular_intensity.x, l.specular_intensity.y, l.specular_intensity.z };
		for (float v : f) if (!finite(v)) return false;
	}
	for (uint32_t i = 0; i < P.n_materials; i++) {
		const lol_material& m = P.materials[i];
		const float f[6] = { m.diffuse.x, m.diffuse.y, m.diffuse.z, m.specular.x, m.specular.y, m.specular.z };
		for (float v : f) if (!finite(v)) return false;
		if (!(m.shininess >= 0.0f)) return false;
	}
	return true;
}

/* Conditions for FLAG_SHADOW_SETTLED (lol_kernel.h, soft_shadow): nothing a shadow march can compute overflows or turns
 * NaN, so that a factor that has reached 0 stays there.  Every number of the scene finite and below 10^15 in magnitude
 * (positions, radii, box sizes, smoothness, light positions).  What bounds the march is its own `t > L` exit, not the
 * step count: t starts at 0 and stays in [0, L] up to the step that ends the march — a step with s < 0 either ends it
 * (res = 50 s / t < -1) or has |s| <= t / 50 and leaves t positive — with L = |light - p| and |p| <= |camera| + 100 + one
 * step of the primary march, all below 10^16; the last step adds one SDF value at such a point.  So every coordinate
 * stays below 10^17 and every squared length below 10^35 < FLT_MAX.  The camera is checked per frame (launch). */
bool shadow_settle_ok(const lol_program& P) {
	auto sane = [](float v) { return v - v == 0.0f && fabsf(v) < 1e15f; };
	for (uint32_t i = 0; i < P.n_ops; i++) {
		const lol_op& o = P.ops[i];
		const int nf = o.op == LOL_OP_SPHERE ? 4 : o.op == LOL_OP_RBOX ? 7 : (o.op == LOL_OP_PLANE || o.op == LOL_OP_SMIN || o.op == LOL_OP_SMIN_R) ? 1 : 0;
		for (int j = 0; j < nf; j++) if (!sane(o.f[j])) return false;
	}
	for (uint32_t i = 0; i < P.n_lights; i++)
		if (!sane(P.lights[i].point.x) || !sane(P.lights[i].point.y) || !sane(P.lights[i].point.z)) return false;
	return true;
}
bool camera_sane(const lol_frame_camera& c) {
	const float* f = reinterpret_cast<const float*>(&c);
	for (size_t i = 0; i < sizeof c / 4; i++) if (!(f[i] - f[i] == 0.0f && fabsf(f[i]) < 1e15f)) return false;
	return true;
}

/* ---------------------------------------------------------------- the primary march's first step, once per camera position
 * Step 0 of get_intersection (naive_renderer.c:56-57) evaluates sdf(ro + rd * 0): the camera's position, the same point for
 * every pixel of the frame.  Like the camera basis of get_camera_ray (naive_renderer.c:183-186, computed once per frame in
 * lol_frame_camera_init) it is a per-frame constant, hoisted: the value is computed HERE, once per camera position, in the
 * reference's own arithmetic — sdf() / get_obj_dist() / sdSphere / sdRoundBox / sminf (naive_renderer.c:11-44, sdf.h:8-22,
 * float.h:6-33) over the post-order program of lol_scene.h, objects in file order, first strict minimum — and handed to the
 * kernels as two launch arguments (lol_kernel.h, FLAG_FIRST_STEP / march).  Every operation is an IEEE binary32 +, -, *, /,
 * sqrt or comparison, correctly rounded here as on the device (this file is compiled with -ffp-contract=off; no libm call), so
 * the value IS what every lane's first step would have computed.  Not used (the kernels take the step themselves) unless:
 * the camera is sane (camera_sane: then every ray direction that is finite makes ro + rd * 0 = ro, see march), no component of
 * the origin is a negative zero, max_steps >= 1, and the value lies in [0.001, 100] — a march that ends on its first step, or
 * goes on with a NaN, is left to the loop.  tests/test_gpu_parity.py compares every pixel's distance, id and step count
 * (this step included) with the oracle's either way. */
inline float h_minf(float a, float b) { return a < b ? a : b; }      /* MINSS: b on NaN / equal (float.h:6) */
inline float h_maxf(float a, float b) { return a > b ? a : b; }
inline float h_len3(float x, float y, float z) { return __builtin_sqrtf((x * x + y * y) + z * z); }      /* DPPS 0x71 (vec.h:52-56): (x² + y²) + (z² + 0) */
bool host_sdf(const lol_program& P, const float p[3], std::vector<float>& st, float* dist_out, uint32_t* id_out) {
	if (st.size() < (size_t)P.max_stack + 1) st.resize((size_t)P.max_stack + 1);
	size_t sp = 0;
	float best = __builtin_inff();
	uint32_t best_id = 0;
	for (uint32_t i = 0; i < P.n_ops; i++) {
		const lol_op& o = P.ops[i];
		switch (o.op) {
		case LOL_OP_SPHERE:
			if (sp >= st.size()) return false;
			st[sp++] = h_len3(p[0] - o.f[0], p[1] - o.f[1], p[2] - o.f[2]) - o.f[3];
			break;
		case LOL_OP_RBOX: {
			if (sp >= st.size()) return false;
			const float qx = __builtin_fabsf(p[0] - o.f[0]) - o.f[3], qy = __builtin_fabsf(p[1] - o.f[1]) - o.f[4], qz = __builtin_fabsf(p[2] - o.f[2]) - o.f[5];
			st[sp++] = h_len3(h_maxf(qx, 0.f), h_maxf(qy, 0.f), h_maxf(qz, 0.f)) + h_minf(h_maxf(qx, h_maxf(qy, qz)), 0.f) - o.f[6];
			break;
		}
		case LOL_OP_PLANE:
			if (sp >= st.size()) return false;
			st[sp++] = p[1] - o.f[0];
			break;
		case LOL_OP_SMIN: case LOL_OP_SMIN_R: {
			if (sp < 2) return false;
			const float top = st[--sp], under = st[--sp];
			const float a = o.op == LOL_OP_SMIN ? under : top, b = o.op == LOL_OP_SMIN ? top : under, k = o.f[0];
			const float h = h_minf(h_maxf(.5f + .5f * (b - a) / k, 0.f), 1.f);
			st[sp++] = (b + (a - b) * h) - k * h * (1.f - h);
			break;
		}
		case LOL_OP_TOP: {
			if (sp < 1) return false;
			const float d = st[--sp];
			if (d < best) { best = d; best_id = o.id; }
			break;
		}
		default: return false;
		}
	}
	*dist_out = best;
	*id_out = best_id;
	return true;
}
/* FLAG_FIRST_STEP for a frame of `cam`?  Fills ctx->first_dist / first_id (kept while the camera stays where it is). */
bool first_step(lol_gpu* ctx, const lol_frame_camera& cam, int max_steps) {
	if (max_steps < 1 || !camera_sane(cam)) return false;
	float origin[3] = { cam.origin.x, cam.origin.y, cam.origin.z };
	uint32_t bits[3];
	memcpy(bits, origin, sizeof bits);
	for (uint32_t b : bits) if (b == 0x80000000u) return false;
	if (ctx->first_gen != ctx->generation || memcmp(ctx->first_origin, origin, sizeof origin) != 0) {
		float d = 0.f; uint32_t id = 0;
		bool ok = false;
		try { ok = host_sdf(ctx->h_prog, origin, ctx->first_stack, &d, &id); } catch (...) { ok = false; }
		ctx->first_dist = ok ? d : __builtin_nanf("");
		ctx->first_id = id;
		ctx->first_gen = ctx->generation;
		memcpy(ctx->first_origin, origin, sizeof origin);
	}
	return ctx->first_dist >= 0.001f && ctx->first_dist <= 100.f;
}

/* --------------------------------------------- exhaustive proofs of the fast paths
 * Each kernel feeds all 2^32 float bit patterns through the shortcut and through the plain
 * expression it replaces and counts the inputs on which they differ (same bits, or both NaN,
 * count as equal).  A shortcut is generated into the specialised kernel only when the count is 0
 * on the device that will run it. */
__device__ __forceinline__ bool same_float(float a, float b) {
	return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b) || (a != a && b != b);
}
constexpr unsigned VERIFY_BLOCKS = 65536, VERIFY_THREADS = 256, VERIFY_ITERS = 256;   /* product = 2^32 */

template <int KIND>
__global__ __launch_bounds__(VERIFY_THREADS) void verify_sqrt_kernel(unsigned long long* bad) {
	uint32_t base = blockIdx.x * VERIFY_THREADS + threadIdx.x;
	unsigned n = 0, m = 0;
	for (uint32_t it = 0; it < VERIFY_ITERS; it++) {
		float x = __builtin_bit_cast(float, base + it * (VERIFY_BLOCKS * VERIFY_THREADS));
		/* The fast roots are only ever given a sum of squares (len2): never negative.  A wave that saw
		 * an argument outside [2^-96, inf) re-shades through the plain path (lol::Range), so the proof
		 * obligation is exactly that interval plus NaN. */
		bool in_domain = (x >= lol::SQRT_FAST_MIN && x < __builtin_inff()) || x != x;
		const float r = lol::sqrt_fast<KIND>(x);
		if (in_domain && !same_float(r, __builtin_sqrtf(x))) n++;
		/* second counter — what sd_sphere_fast_nr relies on: below the proven domain (x in [+0, 2^-96)) the fast root
		 * is NaN or tiny, and +inf gives NaN (never a wrong finite value, never inf) */
		const uint32_t xb = __builtin_bit_cast(uint32_t, x);
		if (xb < lol::SQRT_FAST_MIN_BITS && !(r != r || __builtin_fabsf(r) < 0x1p-47f)) m++;
		if (xb == lol::F32_INF_BITS && !(r != r)) m++;
	}
	if (n) atomicAdd(bad, (unsigned long long)n);
	if (m) atomicAdd(bad + 1, (unsigned long long)m);
}

/* smooth_sat_threshold: the |dlt| from which sminf_fastdiv_sat treats the blend factor as saturated: k(1 + 2^-20),
 * rounded up (so >= k(1 + 2^-21) whatever the rounding); 0 for k <= 0 or non-finite (no shortcut). */
float smooth_sat_threshold(float k) {
	if (!(k > 0.f) || !std::isfinite(k)) return 0.f;
	const double want = (double)k * (1.0 + 0x1p-20);
	float ks = (float)want;
	if ((double)ks < want) ks = nextafterf(ks, INFINITY);
	return std::isfinite(ks) ? ks : 0.f;
}

__global__ __launch_bounds__(VERIFY_THREADS) void verify_div_kernel(float k, float k2, float hrk, float ks, unsigned long long* bad) {
	uint32_t base = blockIdx.x * VERIFY_THREADS + threadIdx.x;
	unsigned n = 0, m = 0;
	for (uint32_t it = 0; it < VERIFY_ITERS; it++) {
		float x = __builtin_bit_cast(float, base + it * (VERIFY_BLOCKS * VERIFY_THREADS));
		const float h = lol::smin_h_exact(x, k);
		if (!same_float(lol::smin_h_fast(x, k2, hrk), h)) n++;
		/* third counter — the same without v_div_fixup (smin_h_fast<false>): equal for every finite dlt and for NaN; for
		 * dlt = +-inf the smooth minimum built on it must come out NaN (it does whenever the quotient is NaN: h clamps to
		 * +0 and b - inf * 0 is NaN) */
		const uint32_t xb = __builtin_bit_cast(uint32_t, x) & 0x7fffffffu;
		if (xb != lol::F32_INF_BITS) { if (!same_float(lol::smin_h_fast<false>(x, k2, hrk), h)) m++; }
		else { const float v = lol::sminf_fastdiv<false>(0.f - x, 0.f, k, k2, hrk); if (!(v != v)) m++; }
		/* what sminf_fastdiv_sat relies on (ks > 0 only): saturated inputs have h == 1 / h == +0 exactly */
		if (ks > 0.f && x >= ks && __builtin_bit_cast(uint32_t, h) != 0x3f800000u) n++;
		if (ks > 0.f && x <= -ks && __builtin_bit_cast(uint32_t, h) != 0u) n++;
	}
	if (n) atomicAdd(bad, (unsigned long long)n);
	if (m) atomicAdd(bad + 1, (unsigned long long)m);
}

/* The gamma staircase (lol_kernel.h, "gamma + quantisation").  Thread k finds T[k], the smallest float in [0, 1] whose channel
 * value (Uint8)(powf(c, 1 / 2.2f) * 255) is >= k, by bisection over the bit patterns (for floats >= +0 the order of the bits is
 * the order of the values) — which presumes the staircase monotone; verify_gamma_kernel then proves table route == powf route
 * for EVERY c, and with it the presumption. */
__global__ __launch_bounds__(lol::GAMMA_LEVELS) void gamma_thresholds_kernel(float* T) {
	const uint32_t k = threadIdx.x;
	if (k == 0) { T[0] = 0.f; T[lol::GAMMA_LEVELS] = __builtin_inff(); return; }
	uint32_t lo = 0u, hi = 0x3f800000u;                 /* value(lo) = 0 < k <= 255 = value(hi) */
	while (hi - lo > 1u) {
		const uint32_t mid = lo + (hi - lo) / 2u;
		if (lol::gamma_u8_exact(__builtin_bit_cast(float, mid)) >= k) hi = mid; else lo = mid;
	}
	T[k] = __builtin_bit_cast(float, hi);
}
/* every float in [+0, 1] — bit patterns 0 ... 0x3f800000 — through both routes */
constexpr unsigned GAMMA_VERIFY_BLOCKS = 16384;         /* x VERIFY_THREADS x VERIFY_ITERS = 2^30 > 0x3f800000 */
__global__ __launch_bounds__(VERIFY_THREADS) void verify_gamma_kernel(const float* T, unsigned long long* bad) {
	__shared__ float t[lol::GAMMA_LEVELS + 1];
	for (uint32_t i = threadIdx.x; i <= (uint32_t)lol::GAMMA_LEVELS; i += VERIFY_THREADS) t[i] = T[i];
	__syncthreads();
	const uint32_t base = blockIdx.x * VERIFY_THREADS + threadIdx.x;
	unsigned n = 0;
	for (uint32_t it = 0; it < VERIFY_ITERS; it++) {
		const uint32_t bits = base + it * (GAMMA_VERIFY_BLOCKS * VERIFY_THREADS);
		if (bits > 0x3f800000u) continue;
		const float c = __builtin_bit_cast(float, bits);
		if (lol::gamma_u8_table(c, t) != lol::gamma_u8_exact(c)) n++;
	}
	if (n) atomicAdd(bad, (unsigned long long)n);
}

/* diagnostic: out[i] = powf_glibc(x[i], y[i]) — lets the tests compare the device's powf with the CPU's */
__global__ __launch_bounds__(256) void powf_batch_kernel(const float* x, const float* y, float* out, size_t n) {
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = lol::powf_glibc(x[i], y[i]);
}

/* returns mismatch count, or ~0ull when the check itself could not run */
unsigned long long run_verify(lol_gpu* ctx, int sqrt_kind, float k, unsigned long long* second = nullptr) {
	if (!ctx->d_bad && hipMalloc(reinterpret_cast<void**>(&ctx->d_bad), 2 * sizeof(unsigned long long)) != hipSuccess)
		return ~0ull;
	unsigned long long bad[2] = { 0, 0 };
	if (hipMemcpy(ctx->d_bad, bad, sizeof bad, hipMemcpyHostToDevice) != hipSuccess) return ~0ull;
	if (sqrt_kind == 3) hipLaunchKernelGGL(verify_sqrt_kernel<3>, dim3(VERIFY_BLOCKS), dim3(VERIFY_THREADS), 0, ctx->stream, ctx->d_bad);
	else if (sqrt_kind == 2) hipLaunchKernelGGL(verify_sqrt_kernel<2>, dim3(VERIFY_BLOCKS), dim3(VERIFY_THREADS), 0, ctx->stream, ctx->d_bad);
	else if (sqrt_kind == 1) hipLaunchKernelGGL(verify_sqrt_kernel<1>, dim3(VERIFY_BLOCKS), dim3(VERIFY_THREADS), 0, ctx->stream, ctx->d_bad);
	else hipLaunchKernelGGL(verify_div_kernel, dim3(VERIFY_BLOCKS), dim3(VERIFY_THREADS), 0, ctx->stream, k, 2.0f * k, 0.5f * (1.0f / k),
	                        smooth_sat_threshold(k), ctx->d_bad);
	if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return ~0ull;
	if (hipMemcpy(bad, ctx->d_bad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) return ~0ull;
	if (second) *second = bad[1];
	return bad[0];
}

/* the thresholds alone (a context whose device another context of this process has proven) */
bool build_gamma_table(lol_gpu* ctx) {
	if (ctx->d_gamma) return true;
	if (hipMalloc(reinterpret_cast<void**>(&ctx->d_gamma), (lol::GAMMA_LEVELS + 1) * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); ctx->d_gamma = nullptr; return false; }
	hipLaunchKernelGGL(gamma_thresholds_kernel, dim3(1), dim3(lol::GAMMA_LEVELS), 0, ctx->stream, ctx->d_gamma);
	return hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
}

/* builds the gamma table of this context (once) and proves it; mismatch count, ~0ull when the check could not run */
unsigned long long run_verify_gamma(lol_gpu* ctx) {
	if (!ctx->d_bad && hipMalloc(reinterpret_cast<void**>(&ctx->d_bad), 2 * sizeof(unsigned long long)) != hipSuccess) return ~0ull;
	const bool fresh = ctx->d_gamma == nullptr;         /* (a table frames may be reading is proven again, not rebuilt) */
	if (fresh && hipMalloc(reinterpret_cast<void**>(&ctx->d_gamma), (lol::GAMMA_LEVELS + 1) * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); ctx->d_gamma = nullptr; return ~0ull; }
	unsigned long long bad[2] = { 0, 0 };
	if (hipMemcpy(ctx->d_bad, bad, sizeof bad, hipMemcpyHostToDevice) != hipSuccess) return ~0ull;
	if (fresh) hipLaunchKernelGGL(gamma_thresholds_kernel, dim3(1), dim3(lol::GAMMA_LEVELS), 0, ctx->stream, ctx->d_gamma);
	hipLaunchKernelGGL(verify_gamma_kernel, dim3(GAMMA_VERIFY_BLOCKS), dim3(VERIFY_THREADS), 0, ctx->stream, ctx->d_gamma, ctx->d_bad);
	if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return ~0ull;
	if (hipMemcpy(bad, ctx->d_bad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) return ~0ull;
	return bad[0];
}

/* ------------------------------------------------- scene → HIP source (the "JIT") */

std::string fbits(float v) {
	uint32_t u;
	memcpy(&u, &v, 4);
	char b[48];
	snprintf(b, sizeof b, "__builtin_bit_cast(float, 0x%08xu)", u);
	return b;
}

/* Which proven-exact shortcuts the generated code may use (see lol_kernel.h "fast exact paths"). */
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

/* ------------------------------------------------ exact culling of top-level objects
 * sdf() (naive_renderer.c:31-44) is a strict-'<' minimum over the top-level objects.  An object whose distance is
 * PROVABLY greater than the running minimum cannot change it, so its evaluation may be skipped — exactly, not
 * approximately.  The proof is a bounding sphere (C, R) per object, computed here in double precision:
 *   sphere(c, r):            value = |p-c| - r                                        → (c, max(r, 0))
 *   round box(c, b, r):      value >= |p-c| - |b| - r   (b >= 0, r >= 0)               → (c, |b| + r)
 *   smooth_union(a, b, k>0): value >= min(a, b) - k/4   (h(1-h) <= 1/4 on the clamped h) → sphere enclosing both + k/4
 *   plane, k <= 0, non-finite or absurdly large fields:                                 no bound — never skipped
 * so value(p) >= |p-C| - R in exact arithmetic.  The kernel's binary32 evaluation differs from that by a few ulps
 * of the magnitudes involved (<= 2^-19 relative to |p-C| + R), which the test below swallows:
 *   R' = R (1 + 2^-10) + (|C|_max + 1) 2^-20, rounded up;   u = (best + R') (1 + 2^-12);
 *   skip  iff  u > 0  and  |p-C|^2 > u^2      (all in binary32; any NaN makes the comparisons false = no skip)
 * which implies |p-C| > (best + R')(1 + 2^-14), hence value(p) > best.  The decision is taken per WAVE: the
 * object is evaluated unless every lane that still cares about the result may skip it (lanes that may skip but
 * run anyway compute a value > best and change nothing).
 *
 * To have a running minimum to compare with, objects WITHOUT a bound (planes: one subtraction) are evaluated first
 * and the bounded ones after them, each group in file order.  The reference's tie rule — the FIRST object of
 * equal distance wins — is kept by comparing ids on ties wherever an object is evaluated after one that follows
 * it in the file:  t < best || (t == best && best_id > id)   (best_id = 0 only while best = +inf, where the
 * reference's inf < inf is false too). */
struct Sphere { bool ok; double c[3], r; uint32_t levels = 1; };   /* levels: nesting depth of the operations under it (a primitive is 1) */

struct RootBound {
	uint32_t first = 0, top = 0;        /* ops [first, top) compute the object, ops[top] is its LOL_OP_TOP */
	uint32_t id = 0, prims = 0;
	bool     bounded = false;
	double   c[3] = { 0, 0, 0 }, r = 0;
	uint32_t levels = 1;                /* nesting depth of its expression (sets the rounding slack of its test) */
	Sphere   sphere() const { return { bounded, { c[0], c[1], c[2] }, r, levels }; }
	std::vector<Sphere> clusters;       /* optional: two spheres that together bound the object more tightly (cluster_bounds) */
};

Sphere enclose(const Sphere& a, const Sphere& b) {
	const uint32_t levels = a.levels > b.levels ? a.levels : b.levels;
	if (!a.ok || !b.ok) return { false, { 0, 0, 0 }, 0, levels };
	const double dx = b.c[0] - a.c[0], dy = b.c[1] - a.c[1], dz = b.c[2] - a.c[2];
	const double d = sqrt(dx * dx + dy * dy + dz * dz);
	if (d + b.r <= a.r) { Sphere r = a; r.levels = levels; return r; }
	if (d + a.r <= b.r) { Sphere r = b; r.levels = levels; return r; }
	const double R = 0.5 * (d + a.r + b.r), t = d > 0 ? (R - a.r) / d : 0.0;
	return { true, { a.c[0] + dx * t, a.c[1] + dy * t, a.c[2] + dz * t }, R * (1.0 + 1e-12), levels };
}

void cluster_bounds(const lol_program& P, RootBound& R);

std::vector<RootBound> analyse_roots(const lol_program& P) {
	std::vector<RootBound> roots;
	std::vector<Sphere> st;
	auto sane = [](double v) { return v - v == 0.0 && fabs(v) < 1e15; };
	RootBound cur;
	for (uint32_t i = 0; i < P.n_ops; i++) {
		const lol_op& o = P.ops[i];
		switch (o.op) {
		case LOL_OP_SPHERE: {
			const bool ok = sane(o.f[0]) && sane(o.f[1]) && sane(o.f[2]) && sane(o.f[3]);
			st.push_back({ ok, { o.f[0], o.f[1], o.f[2] }, o.f[3] > 0 ? (double)o.f[3] : 0.0 });
			cur.prims++;
			break;
		}
		case LOL_OP_RBOX: {
			bool ok = true;
			for (int j = 0; j < 7; j++) ok = ok && sane(o.f[j]);
			ok = ok && o.f[3] >= 0 && o.f[4] >= 0 && o.f[5] >= 0 && o.f[6] >= 0;
			const double hb = sqrt((double)o.f[3] * o.f[3] + (double)o.f[4] * o.f[4] + (double)o.f[5] * o.f[5]);
			st.push_back({ ok, { o.f[0], o.f[1], o.f[2] }, hb * (1.0 + 1e-12) + o.f[6] });
			cur.prims++;
			break;
		}
		case LOL_OP_PLANE:
			st.push_back({ false, { 0, 0, 0 }, 0 });
			cur.prims++;
			break;
		case LOL_OP_SMIN: case LOL_OP_SMIN_R: {
			Sphere b = st.back(); st.pop_back();
			Sphere a = st.back(); st.pop_back();
			Sphere u = enclose(a, b);
			if (!(sane(o.f[0]) && o.f[0] > 0)) u.ok = false;
			u.r += 0.25 * (double)o.f[0];
			u.levels++;
			st.push_back(u);
			break;
		}
		case LOL_OP_TOP: {
			Sphere v = st.back(); st.pop_back();
			cur.top = i; cur.id = o.id;
			cur.bounded = v.ok && sane(v.r);
			cur.c[0] = v.c[0]; cur.c[1] = v.c[1]; cur.c[2] = v.c[2]; cur.r = v.r; cur.levels = v.levels;
			cluster_bounds(P, cur);
			roots.push_back(cur);
			cur = RootBound();
			cur.first = i + 1;
			break;
		}
		}
	}
	return roots;
}

/* Two spheres instead of one (round 3).  One sphere around a long or L-shaped union is mostly empty.  For a union tree with
 * every k > 0:  smooth_union(a, b, k) >= min(a, b) - k/4, so by induction  value(p) >= min over the LEAVES i of
 * (prim_i(p) - slack_i),  slack_i = the sum of k/4 over the unions above leaf i;  and prim_i(p) >= |p - c_i| - r_i for a sphere
 * (round box: r_i = |b| + r).  Split the leaves into two clusters and let sphere S_j enclose the spheres (c_i, r_i + slack_i) of
 * its cluster: then  value(p) >= min_j (|p - C_j| - R_j)  in exact arithmetic, and the object may be skipped where BOTH of the
 * usual tests pass (make_test: each with the rounding slack of the object's depth).  The split: along the widest axis of the
 * leaf centres, at the position that minimises R_A^3 + R_B^3; used when the larger of the two is at most 0.75 of the single
 * sphere's radius (scene4's blob: 8.6 and 7.8 against 11.1; a numpy model of C3 — tools/cull_model.py — puts the wave-evaluations
 * that may skip the blob at 35 % against 30 %). */
void cluster_bounds(const lol_program& P, RootBound& R) {
	R.clusters.clear();
	if (!R.bounded || R.prims < 3) return;
	std::vector<std::vector<Sphere>> st;
	for (uint32_t i = R.first; i < R.top; i++) {
		const lol_op& o = P.ops[i];
		if (o.op == LOL_OP_SPHERE) st.push_back({ { true, { o.f[0], o.f[1], o.f[2] }, o.f[3] > 0 ? (double)o.f[3] : 0.0, R.levels } });
		else if (o.op == LOL_OP_RBOX) {
			const double hb = sqrt((double)o.f[3] * o.f[3] + (double)o.f[4] * o.f[4] + (double)o.f[5] * o.f[5]);
			st.push_back({ { true, { o.f[0], o.f[1], o.f[2] }, hb * (1.0 + 1e-12) + o.f[6], R.levels } });
		} else if (o.op == LOL_OP_SMIN || o.op == LOL_OP_SMIN_R) {
			std::vector<Sphere> b = std::move(st.back()); st.pop_back();
			std::vector<Sphere>& a = st.back();
			a.insert(a.end(), b.begin(), b.end());
			for (Sphere& l : a) l.r += 0.25 * (double)o.f[0];          /* the slack of this union, for every leaf under it */
		} else return;                                                   /* (a plane: the object has no bound at all) */
	}
	if (st.size() != 1 || st[0].size() < 3 || st[0].size() > 4096) return;      /* (the cut search below is quadratic in the leaves) */
	std::vector<Sphere>& leaves = st[0];
	double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
	for (const Sphere& l : leaves) for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], l.c[a]); mx[a] = fmax(mx[a], l.c[a]); }
	int axis = 0;
	for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
	std::stable_sort(leaves.begin(), leaves.end(), [&](const Sphere& x, const Sphere& y) { return x.c[axis] < y.c[axis]; });
	auto hull = [&](size_t lo, size_t hi) { Sphere g = leaves[lo]; for (size_t k = lo + 1; k < hi; k++) g = enclose(g, leaves[k]); g.levels = R.levels; return g; };
	double best = 1e300; size_t cut = 0;
	for (size_t c = 1; c < leaves.size(); c++) {
		const Sphere a = hull(0, c), b = hull(c, leaves.size());
		const double cost = a.r * a.r * a.r + b.r * b.r * b.r;
		if (cost < best) { best = cost; cut = c; }
	}
	const Sphere a = hull(0, cut), b = hull(cut, leaves.size());
	auto sane = [](double v) { return v - v == 0.0 && fabs(v) < 1e15; };
	if (!a.ok || !b.ok || !sane(a.r) || !sane(b.r)) return;
	if (fmax(a.r, b.r) > 0.75 * R.r) return;
	R.clusters = { a, b };
}

struct CullTest { float c[3]; float rm; float k; };     /* skip iff u = (best + rm)*k > 0 and |p - c|^2 > u^2 */

/* The in-kernel test's constants from a double-precision bound: centre rounded to binary32 (its rounding error is
 * covered by the |C| 2^-20 term), radius and comparison inflated by the rounding the guarded expression can
 * accumulate.  With D = |p-C|: the exact-arithmetic value is >= D - R; one smooth minimum evaluated in binary32 adds
 * at most 2^-24 (4 M + 8.5 k) to the error of its operands (it is 1-Lipschitz in them), a primitive at most
 * 2^-24 * 4 M, with M <= D + R and k <= 4 R — so the binary32 value is >= D(1 - e) - R(1 + e), e = 40 * 2^-24 * levels.
 * The test gives D > (best + R')K(1 - 2^-21); with K >= 1 + 2e + 2^-19 and R' >= R(1 + 2(K-1) + 2e) that is
 * > best in both signs of best (for best < 0 use |best| < R').  Shallow objects (levels <= 50) keep the constants
 * of the first version, K = 1 + 2^-12 and R' = R(1 + 2^-10): a chain of 500 unions gets K = 1.0024. */
CullTest make_test(const Sphere& b) {
	CullTest t;
	double cmax = 0;
	for (int j = 0; j < 3; j++) { t.c[j] = (float)b.c[j]; cmax = fmax(cmax, fabs(b.c[j])); }
	const double e = 40.0 * 0x1p-24 * (double)b.levels;
	const double K = 1.0 + fmax(0x1p-12, 2.0 * e + 0x1p-19);
	t.k = (float)K;
	if ((double)t.k < K) t.k = nextafterf(t.k, INFINITY);
	const double rho = fmax(0x1p-10, 2.0 * ((double)t.k - 1.0) + 2.0 * e);
	const double rm = b.r * (1.0 + rho) + (cmax + 1.0) * 0x1p-20;
	t.rm = nextafterf((float)rm, INFINITY);
	return t;
}

std::vector<CullTest> cluster_tests(const RootBound& r) {
	std::vector<CullTest> t;
	for (const Sphere& c : r.clusters) t.push_back(make_test(c));
	return t;
}

/* A test guards a run of consecutive objects of the evaluation order: [begin, end) positions in `order`.  Runs nest
 * (the run of all bounded objects, inside it spatial clusters, inside those single heavy objects). */
/* `both`: when not empty the run (always a single object) is skipped where ALL of these pass — the object's two cluster
 * spheres — instead of the one test of its enclosing sphere; the interpreter keeps the one sphere (`test`). */
struct CullInterval { size_t begin, end; CullTest test; std::vector<CullTest> both; };

struct CullPlan {
	std::vector<uint32_t> order;          /* evaluation order: indices into the root list */
	size_t   n_unbounded = 0;             /* the first n_unbounded entries of `order` have no bound */
	std::vector<CullInterval> intervals;  /* outer before inner, by position */
	bool     group = false;               /* intervals[0] is the run of ALL bounded objects (what the interpreter carries) */
	CullTest group_test{};
};

bool culling_enabled(int want) { return want != 0; }      /* (lol_gpu_set_cull) */

/* Objects that lie together are evaluated together, behind a test of their common bounding sphere: a k-d split of
 * the bounded objects (median cut along the widest axis of their centres, down to runs of at most three) gives the
 * evaluation order, and every node of that tree gets a test — single objects too: 11 instructions against a sphere's
 * ~20 with its square root, measured faster on every scene tried (tools/flat_scene_ab.py: leaf sizes 2…8, tests from
 * 1…4 primitives up; profiles/r2_flat_scene_ab.jsonl).  A ray that is far from a whole cluster pays one test for it
 * instead of one evaluation per object — what makes a scene of hundreds of separate objects affordable. */
static void kd_build(const std::vector<RootBound>& roots, std::vector<uint32_t>& ids, size_t lo, size_t hi,
                     size_t base, bool has_predecessor, size_t leaf_max, uint32_t min_prims, CullPlan& plan) {
	const size_t count = hi - lo;
	Sphere g = roots[ids[lo]].sphere();
	uint32_t prims = roots[ids[lo]].prims;
	for (size_t k = lo + 1; k < hi; k++) {
		const RootBound& r = roots[ids[k]];
		g = enclose(g, r.sphere());
		prims += r.prims;
	}
	/* a test needs a running minimum to compare with (something evaluated before the run); a node that covers
	 * exactly what its parent covers adds nothing */
	const bool same_as_parent = !plan.intervals.empty() && plan.intervals.back().begin == base + lo && plan.intervals.back().end == base + hi;
	if ((has_predecessor || lo > 0) && prims >= min_prims && !same_as_parent)
		plan.intervals.push_back({ base + lo, base + hi, make_test(g), count == 1 ? cluster_tests(roots[ids[lo]]) : std::vector<CullTest>() });
	if (count <= leaf_max) {
		if (count > 1)                       /* inside a small run: the objects' own tests */
			for (size_t k = lo; k < hi; k++) {
				const RootBound& r = roots[ids[k]];
				if (r.prims >= min_prims && (has_predecessor || k > 0))
					plan.intervals.push_back({ base + k, base + k + 1, make_test(r.sphere()), cluster_tests(r) });
			}
		return;
	}
	double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
	for (size_t k = lo; k < hi; k++)
		for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], roots[ids[k]].c[a]); mx[a] = fmax(mx[a], roots[ids[k]].c[a]); }
	int axis = 0;
	for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
	const size_t mid = lo + count / 2;
	std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi,
	                 [&](uint32_t x, uint32_t y) { return roots[x].c[axis] < roots[y].c[axis] || (roots[x].c[axis] == roots[y].c[axis] && x < y); });
	kd_build(roots, ids, lo, mid, base, has_predecessor, leaf_max, min_prims, plan);
	kd_build(roots, ids, mid, hi, base, has_predecessor, leaf_max, min_prims, plan);
}

CullPlan plan_culling(const std::vector<RootBound>& roots, bool enabled) {
	CullPlan plan;
	std::vector<uint32_t> bounded;
	if (enabled)
		for (uint32_t i = 0; i < roots.size(); i++) (roots[i].bounded ? bounded : plan.order).push_back(i);
	else
		for (uint32_t i = 0; i < roots.size(); i++) plan.order.push_back(i);
	plan.n_unbounded = plan.order.size();
	if (!enabled || bounded.empty()) return plan;
	/* LOL_GPU_CULL_CLUSTERS=0: one run of all bounded objects in scene order, no spatial clusters (for A/B runs) */
	const char* e = tuning_env("LOL_GPU_CULL_CLUSTERS");
	const size_t leaf_max = (e && atoi(e) == 0) ? (size_t)-1 : (e && atoi(e) > 1 ? (size_t)atoi(e) : 3);
	const uint32_t min_prims = 1;      /* every node of the tree gets its test (profiles/r2_flat_scene_ab.jsonl: tests from 1 ... 4 primitives up) */
	kd_build(roots, bounded, 0, bounded.size(), plan.n_unbounded, plan.n_unbounded > 0, leaf_max, min_prims, plan);
	plan.order.insert(plan.order.end(), bounded.begin(), bounded.end());
	/* outer runs before inner ones at the same position (kd_build emits parents first; keep that order stable) */
	std::stable_sort(plan.intervals.begin(), plan.intervals.end(), [](const CullInterval& a, const CullInterval& b) {
		return a.begin < b.begin || (a.begin == b.begin && a.end > b.end);
	});
	if (!plan.intervals.empty() && plan.intervals[0].begin == plan.n_unbounded && plan.intervals[0].end == plan.order.size() &&
	    plan.n_unbounded > 0) {
		plan.group = true;
		plan.group_test = plan.intervals[0].test;
	}
	return plan;
}

/*
 * Post-order program → macro-ops of the interpreter (lol_kernel.h, Interp).  The accumulator is the top of the
 * post-order operand stack:
 *   primitive followed by SMIN / SMIN_R  → one macro-op: x = primitive, combined with acc at once.  In post-order
 *                                           SMIN pops b (top) then a; the primitive is the top, so x = b and the
 *                                           combine is sminf(acc, x); SMIN_R (top is a) gives sminf(x, acc);
 *   primitive otherwise                   → SET when nothing is on the stack yet, else PUSH (acc goes under);
 *   SMIN / SMIN_R after a non-primitive   → x = the popped entry under acc: SMIN has a = x, b = acc → sminf(x, acc);
 *                                           SMIN_R has a = acc, b = x → sminf(acc, x) — a POP record of its own, unless the
 *                                           record before it has a smooth min of the same, proven k: then it rides on that
 *                                           record (MOPB_POST: after the record's own combine);
 *   TOP                                   → a flag on the macro-op that produced the value (+ MOP_TIE where the
 *                                           object is evaluated after one that follows it in the file).
 * The objects come in the order of `plan` (unbounded ones first).  Every test of the plan becomes a constants
 * record in front of the first object of its run (outer runs first); the macro-op that finishes the object before
 * it gets MOPB_CULL_NEXT (the run of all bounded objects) and / or MOPB_CULL_CHAIN (inner runs), and the
 * CULLC_NEXT / CULLC_AFTER flags chain test records that follow one another directly.
 * `fast` lists the smoothness constants whose fast blend factor was proven on the device; allow_nofixup: this list may use the
 * form without v_div_fixup where that was proven too (the caller builds both lists: same records, other smooth-min bits).
 */
std::vector<uint32_t> build_mops(const lol_program& P, const FastPaths* fast, const std::vector<RootBound>& roots,
                                 const CullPlan& plan, bool allow_nofixup) {
	std::vector<uint32_t> out;
	auto fbits32 = [](float v) { uint32_t u; memcpy(&u, &v, 4); return u; };
	auto smin_fields = [&](uint32_t* m, const lol_op& sm) {
		m[9] = fbits32(sm.f[0]);
		if (fast && fast->has(sm.f[0])) {
			m[0] |= lol::MOP_FASTDIV | (allow_nofixup && fast->has_nf(sm.f[0]) ? lol::MOP_NOFIXUP : 0u);
			m[10] = fbits32(2.0f * sm.f[0]);
			m[11] = fbits32(0.5f * (1.0f / sm.f[0]));
		}
		m[0] |= lol::mop_smin_bits(m[0]);
	};
	/* (every test of the plan is kept: leaving out those of runs with few primitives was measured slower here too — a test is one
	 * turn of a scalar loop inside the rare TAIL branch) */
	/* deep: operand stacks beyond the 4-bit slot fields — slots travel in words of their own and pops are not fused (lol_kernel.h, MOP_DEEP_FROM) */
	const bool deep = interp_stack_class(P.max_stack) == lol::MOP_DEEP_SLOTS;
	const bool fuse_pops = !deep && !(tuning_env("LOL_GPU_INTERP_FUSE_POPS") && tuning_env("LOL_GPU_INTERP_FUSE_POPS")[0] == '0');     /* A/B switch */
	const std::vector<CullInterval>& ivs = plan.intervals;
	const bool group_first = plan.group && !ivs.empty() && ivs[0].begin == plan.n_unbounded && ivs[0].end == plan.order.size();
	std::vector<size_t> at(ivs.size());              /* where each test's constants record went */
	std::vector<uint32_t> begins(plan.order.size() + 1, 0);
	for (const CullInterval& iv : ivs) begins[iv.begin]++;
	std::vector<std::vector<size_t>> ends_at(plan.order.size() + 1);      /* the runs that end after object oi - 1 (linear, not a scan per object) */
	for (size_t k = 0; k < ivs.size(); k++) ends_at[ivs[k].end].push_back(k);
	uint32_t max_id_seen = 0;
	size_t next_iv = 0;
	for (size_t oi = 0; oi < plan.order.size(); oi++) {
		const RootBound& R = roots[plan.order[oi]];
		for (uint32_t n = 0; n < begins[oi]; n++, next_iv++) {             /* (never at oi == 0: plan_culling) */
			const CullInterval& iv = ivs[next_iv];
			uint32_t c[lol::MOP_DWORDS] = { 0 };
			c[0] = (n + 1 < begins[oi] ? lol::CULLC_NEXT : 0u) | (begins[iv.end] ? lol::CULLC_AFTER : 0u);
			for (int j = 0; j < 3; j++) c[2 + j] = fbits32(iv.test.c[j]);
			c[5] = fbits32(iv.test.rm);
			c[6] = fbits32(iv.test.k);
			at[next_iv] = out.size();
			out.insert(out.end(), c, c + lol::MOP_DWORDS);
		}
		int depth = 0;                                   /* post-order stack depth before the current op */
		bool emitted = false;                            /* this object has a record yet (`last` is one of its own) */
		size_t last = 0;                                 /* start of the macro-op that produced the current acc */
		for (uint32_t i = R.first; i < R.top; i++) {
			const lol_op& o = P.ops[i];
			uint32_t m[lol::MOP_DWORDS] = { 0 };
			if (o.op <= LOL_OP_PLANE) {
				const uint32_t kind = o.op == LOL_OP_SPHERE ? lol::MOP_SPHERE : o.op == LOL_OP_RBOX ? lol::MOP_RBOX : lol::MOP_PLANE;
				for (int j = 0; j < 7; j++) m[2 + j] = fbits32(o.f[j]);
				const lol_op* nx = i + 1 < R.top ? &P.ops[i + 1] : nullptr;
				if (nx && (nx->op == LOL_OP_SMIN || nx->op == LOL_OP_SMIN_R) && depth >= 1) {
					m[0] = lol::mop_header(kind, nx->op == LOL_OP_SMIN ? lol::MOP_SMIN : lol::MOP_SMIN_X);
					smin_fields(m, *nx);
					i++;                                /* the smooth min is part of this macro-op; depth unchanged */
				} else {
					m[0] = lol::mop_header(kind, depth == 0 ? lol::MOP_SET : lol::MOP_PUSH);
					if (depth > 0) {                                                       /* the accumulator goes to this slot */
						if (deep) m[9] = (uint32_t)(depth - 1);
						else m[0] |= (uint32_t)(depth - 1) << lol::MOP_SLOT_SHIFT;
					}
					depth++;
				}
			} else {                                     /* SMIN / SMIN_R on two computed operands */
				/* ... rides on the record that has just finished the second operand when that record's own smooth min has the
				 * same, proven k (the record's k words serve both; lol_kernel.h, MOPB_POST): one record less to fetch and dispatch */
				if (fuse_pops && emitted && (out[last] & lol::MOPB_SMIN) && (out[last] & lol::MOP_FASTDIV) && !(out[last] & lol::MOPB_POST) &&
				    out[last + 9] == fbits32(o.f[0])) {
					out[last] |= lol::MOPB_POST | lol::MOPB_STACK | lol::MOPB_TAIL | (o.op == LOL_OP_SMIN ? lol::MOPB_POST_YA : 0u) |
					             (uint32_t)(depth - 2) << lol::MOP_POST_SLOT_SHIFT;
					depth--;
					continue;
				}
				m[0] = lol::mop_header(lol::MOP_POP, o.op == LOL_OP_SMIN ? lol::MOP_SMIN_X : lol::MOP_SMIN);
				if (deep) m[2] = (uint32_t)(depth - 2);                                         /* the operand under the accumulator */
				else m[0] |= (uint32_t)(depth - 2) << lol::MOP_SLOT_SHIFT;
				smin_fields(m, o);
				depth--;
			}
			last = out.size();
			emitted = true;
			out.insert(out.end(), m, m + lol::MOP_DWORDS);
		}
		out[last] |= lol::MOP_TOP | lol::MOPB_TAIL | (R.id < max_id_seen ? lol::MOP_TIE : 0u);
		out[last + 1] = R.id;
		if (R.id > max_id_seen) max_id_seen = R.id;
		if (begins[oi + 1]) {
			const bool group_here = group_first && oi + 1 == plan.n_unbounded;
			if (group_here) out[last] |= lol::MOPB_CULL_NEXT;
			if (begins[oi + 1] > (group_here ? 1u : 0u)) out[last] |= lol::MOPB_CULL_CHAIN;
		}
		for (size_t k : ends_at[oi + 1])                                   /* every run that ends here: how far its test jumps */
			out[at[k] + 1] = (uint32_t)((out.size() - at[k]) / lol::MOP_DWORDS - 1);
	}
	return out;
}

/* Emits one `struct <name>` with eval(): one SSA temporary per op, same operation order WITHIN every top-level object
 * as the post-order program; the objects themselves in the order of `plan` (file order when culling is off).
 * out_of_line: the body becomes ONE real function (`<name>_fn`, __noinline__) that the march, normal and shadow
 * loops call, instead of being inlined into each of them — for large scenes, whose straight-line SDF would
 * otherwise be replicated six times (three loops x fast / exact) and outgrow the instruction cache. */
void emit_sdf(std::string& s, const lol_program& P, const char* name, const FastPaths* fast, bool out_of_line,
              const std::vector<RootBound>& roots, const CullPlan& plan, const std::string& occupancy) {
	char line[768];
	const int fsqrt = fast ? fast->sqrt_kind : 0;
	char fs[32] = "";
	if (fsqrt) snprintf(fs, sizeof fs, "_fast<%d>", fsqrt);
	/* only the outermost test keeps a cool-down counter (wave-uniform state in the Sdf struct) */
	char cool_decl[64] = "";
	if (!plan.intervals.empty()) snprintf(cool_decl, sizeof cool_decl, "\tu32 cool[1] = {};\n");
	if (out_of_line) {
		/* out of line the cool-down state is per call (always 0: every evaluation tests) */
		/* (amdgpu_waves_per_eu applies to kernels only: the function is scheduled with the default register budget) */
		(void)occupancy;
		snprintf(line, sizeof line, "__device__ __noinline__ SdfOut %s_fn(float px, float py, float pz, u32 rg_lo, u32 rg_hi) {\n"
		         "\t\tconst V3 p = { px, py, pz };\n\t\tRange rg; rg.lo = rg_lo; rg.hi = rg_hi;\n\t\tfloat nanacc = 0.f;\n\t\tfloat best; u32 best_id;\n\t%s",
		         name, cool_decl);
		s += line;
	} else {
		/* ASSUME_SETTLED: the fast pipeline only runs under FLAG_SHADOW_SETTLED (generate_source; lol_kernel.h, soft_shadow).
		 * loop_done(): what the wave-uniform cool-down counter is after a loop that lanes leave one by one (lol_kernel.h, Interp) */
		snprintf(line, sizeof line, "struct %s {\n\tstatic constexpr bool ASSUME_SETTLED = %s;\n\tRange rg;\n\tfloat nanacc = 0.f;\n%s"
		         "\t__device__ __forceinline__ void loop_done() { %s }\n", name, fast ? "true" : "false", cool_decl,
		         plan.intervals.empty() ? "" : "cool[0] = 0u;");
		s += line;
		s += "\t__device__ __forceinline__ void eval(V3 p, float& best, u32& best_id) {\n";
	}
	s += "\t\tbest = __builtin_inff(); best_id = 0u;\n";
	int t = 0, n_tests = 0;
	/* After a test that did not allow the skip, the next `cooldown` evaluations of this SDF object do not test
	 * again (a ray that is near the object now is near it on its next steps too): the test costs 11 VALU
	 * instructions, and where it keeps failing that is pure overhead.  Never testing is always allowed — the
	 * test only ever permits a skip — so this changes no result.  `cool` lives in the Sdf struct, wave-uniform. */
	const int cooldown = 3;
	/* spheres of radius >= 2^-20 carry no range tracker where the device has shown the fast root harmless below its proven domain
	 * (lol_kernel.h, sd_sphere_fast_nr): a NaN reaches the object's value instead, which is looked at once */
	const bool nan_flag = fast && fast->sqrt_tiny_ok;
	bool object_has_nr = false;
	/* LOL_GPU_SAT_CULL_MIN_PRIMS: `a` operands of a smooth union with at least this many primitives get a saturation-
	 * culling test (see emit_node below); 0 = none.  Measured (tools/tree_scene_ab.py, balanced trees of spheres at
	 * 1080p, profiles/r2_tree_scene_ab.jsonl): the 17-instruction test pays from a few dozen primitives — 128 spheres
	 * 232 -> 405 Mpixels/s, 256 spheres 127 -> 166 at 32 (375 / 157 at 16); with a test on every operand scene4
	 * loses 14 %, a 32-sphere tree 30 %. */
	int sat_cull_min_prims = 32;
	if (const char* e = tuning_env("LOL_GPU_SAT_CULL_MIN_PRIMS")) sat_cull_min_prims = atoi(e);
	/* one test = one bounding sphere; a run guarded by several (an object's two cluster spheres) is skipped where ALL pass */
	auto open_test = [&](const CullInterval& iv, bool with_cooldown) {
		const std::vector<CullTest> one = { iv.test };
		const std::vector<CullTest>& tests = iv.both.empty() ? one : iv.both;
		std::string decl, votes;
		const int k0 = n_tests;
		for (const CullTest& ct : tests) {
			const int k = n_tests++;
			snprintf(line, sizeof line,
			         "\t\t  const float cx%d = p.x - %s, cy%d = p.y - %s, cz%d = p.z - %s;\n"
			         "\t\t  const float cl%d = (cx%d * cx%d + cy%d * cy%d) + cz%d * cz%d;\n"
			         "\t\t  const float cu%d = (best + %s) * %s;\n",
			         k, fbits(ct.c[0]).c_str(), k, fbits(ct.c[1]).c_str(), k, fbits(ct.c[2]).c_str(),
			         k, k, k, k, k, k, k, k, fbits(ct.rm).c_str(), fbits(ct.k).c_str());
			decl += line;
			snprintf(line, sizeof line, "%svote(!(cl%d > cu%d * cu%d)) | vote(!(cu%d > 0.f))", votes.empty() ? "" : " | ", k, k, k, k);
			votes += line;
		}
		if (with_cooldown) {
			snprintf(line, sizeof line, "\t\t{ bool need%d = true;\n\t\t  if (cool[0] == 0u) {\n", k0);
			s += line;
			s += decl;
			snprintf(line, sizeof line, "\t\t  need%d = ((", k0);
			s += line;
			s += votes;
			snprintf(line, sizeof line, ")) != 0;\n\t\t  if (need%d) cool[0] = %du;\n\t\t  } else cool[0]--;\n\t\t  if (need%d) {\n", k0, cooldown, k0);
			s += line;
		} else {
			s += "\t\t{\n";
			s += decl;
			s += "\t\t  if (((";
			s += votes;
			s += ")) != 0) {\n";
		}
	};
	uint32_t max_id_seen = 0;
	size_t next_iv = 0;
	std::vector<uint32_t> runs_ending(plan.order.size() + 1, 0);          /* how many runs end after object oi - 1 */
	for (const CullInterval& iv : plan.intervals) runs_ending[iv.end]++;
	for (size_t oi = 0; oi < plan.order.size(); oi++) {
		const RootBound& R = roots[plan.order[oi]];
		while (next_iv < plan.intervals.size() && plan.intervals[next_iv].begin == oi) {      /* outer runs first */
			open_test(plan.intervals[next_iv], next_iv == 0);
			next_iv++;
		}
		/* the object's expression tree from its post-order ops (child `a` / `b` = the operands of sminf(a, b, k)) */
		struct Node { uint32_t op; int a, b; Sphere bound; uint32_t prims; bool fon = false; };      /* fon: the fast SDF's value of this node is finite or NaN */
		std::vector<Node> nodes;
		{
			std::vector<int> st;
			auto sane = [](double v) { return v - v == 0.0 && fabs(v) < 1e15; };
			for (uint32_t i = R.first; i < R.top; i++) {
				const lol_op& o = P.ops[i];
				Node n{ i, -1, -1, { false, { 0, 0, 0 }, 0, 1 }, 1 };
				if (o.op == LOL_OP_SPHERE) {
					n.fon = fsqrt && nan_flag && o.f[3] >= 0x1p-20f && std::isfinite(o.f[3]);      /* sd_sphere_fast_nr (emit_node) */
					n.bound = { sane(o.f[0]) && sane(o.f[1]) && sane(o.f[2]) && sane(o.f[3]), { o.f[0], o.f[1], o.f[2] }, o.f[3] > 0 ? (double)o.f[3] : 0.0, 1 };
				} else if (o.op == LOL_OP_RBOX) {
					bool ok = true;
					for (int j = 0; j < 7; j++) ok = ok && sane(o.f[j]);
					ok = ok && o.f[3] >= 0 && o.f[4] >= 0 && o.f[5] >= 0 && o.f[6] >= 0;
					const double hb = sqrt((double)o.f[3] * o.f[3] + (double)o.f[4] * o.f[4] + (double)o.f[5] * o.f[5]);
					n.bound = { ok, { o.f[0], o.f[1], o.f[2] }, hb * (1.0 + 1e-12) + o.f[6], 1 };
				} else if (o.op == LOL_OP_SMIN || o.op == LOL_OP_SMIN_R) {
					const int top = st.back(); st.pop_back();
					const int under = st.back(); st.pop_back();
					n.a = o.op == LOL_OP_SMIN ? under : top;
					n.b = o.op == LOL_OP_SMIN ? top : under;
					n.bound = enclose(nodes[n.a].bound, nodes[n.b].bound);
					if (!(sane(o.f[0]) && o.f[0] > 0)) n.bound.ok = false;
					n.bound.r += 0.25 * (double)o.f[0];
					n.bound.levels++;
					if (!sane(n.bound.r)) n.bound.ok = false;
					n.prims = nodes[n.a].prims + nodes[n.b].prims;
					n.fon = nodes[n.a].fon && nodes[n.b].fon && fast && fast->has(o.f[0]);
				}
				st.push_back((int)nodes.size());
				nodes.push_back(n);
			}
		}
		/* Saturation culling inside a smooth union (fast struct only, proven k > 0): sminf(a, b, k) is EXACTLY
		 * b - dlt*0.f = b + 0.f when dlt = b - a <= -ks (sminf_fastdiv_sat), so operand `a` need not be evaluated
		 * where it is provably that much greater than b.  b is evaluated first; with sb = fl(b + ks) the bounding-
		 * sphere test of the top-level culling (best := sb) gives a > sb for the binary32 value of `a`, hence
		 * dlt = fl(b - a) <= fl(b - sb) =: sw by the monotonicity of rounding, and sw <= -ks is checked directly;
		 * |p - C|^2 < 2^120 keeps every primitive of `a` (all within R < 10^15 of C) finite, so dlt is finite and
		 * dlt*0.f = -0.  A NaN or infinite b fails the comparisons.  Per wave, like every other skip. */
		std::function<int(int)> emit_node = [&](int ni) -> int {
			const Node& n = nodes[ni];
			const lol_op& o = P.ops[n.op];
			switch (o.op) {
			case LOL_OP_SPHERE:
				if (fsqrt && nan_flag && o.f[3] >= 0x1p-20f && std::isfinite(o.f[3])) {      /* sd_sphere_fast_nr: no range tracker */
					snprintf(line, sizeof line, "\t\tconst float t%d = sd_sphere_fast_nr<%d>(p, %s, %s, %s, %s);\n", t, fsqrt,
					         fbits(o.f[0]).c_str(), fbits(o.f[1]).c_str(), fbits(o.f[2]).c_str(), fbits(o.f[3]).c_str());
					object_has_nr = true;
				} else
				snprintf(line, sizeof line, "\t\tconst float t%d = sd_sphere%s(p, %s, %s, %s, %s%s);\n", t, fs,
				         fbits(o.f[0]).c_str(), fbits(o.f[1]).c_str(), fbits(o.f[2]).c_str(), fbits(o.f[3]).c_str(),
				         fsqrt ? ", rg" : "");
				s += line; return t++;
			case LOL_OP_RBOX:
				snprintf(line, sizeof line, "\t\tconst float t%d = sd_round_box%s(p, %s, %s, %s, %s, %s, %s, %s%s);\n", t,
				         fs,
				         fbits(o.f[0]).c_str(), fbits(o.f[1]).c_str(), fbits(o.f[2]).c_str(), fbits(o.f[3]).c_str(),
				         fbits(o.f[4]).c_str(), fbits(o.f[5]).c_str(), fbits(o.f[6]).c_str(), fsqrt ? ", rg" : "");
				s += line; return t++;
			case LOL_OP_PLANE:
				snprintf(line, sizeof line, "\t\tconst float t%d = p.y - %s;\n", t, fbits(o.f[0]).c_str());
				s += line; return t++;
			default: break;
			}
			/* LOL_OP_SMIN / LOL_OP_SMIN_R */
			const float ks = smooth_sat_threshold(o.f[0]);
			const bool proven = fast && fast->has(o.f[0]);
			/* without v_div_fixup where that is proven too; such an object's value is then voted on for NaN (an infinite
			 * operand difference — lol_kernel.h, smin_h_fast) like one with spheres that carry no range tracker */
			const char* fx = proven && fast->has_nf(o.f[0]) ? "<false>" : "";
			if (fx[0]) object_has_nr = true;
			/* the arithmetic shortcut only where the SDF is inlined: in the out-of-line function its four-way branching
			 * costs more than it saves (504-op chain: 100 -> 42 Mpixels/s) */
			const bool sat_arith = !out_of_line && ks > 0.f;
			const std::string kk = fbits(o.f[0]), k2 = fbits(2.0f * o.f[0]), hrk = fbits(0.5f * (1.0f / o.f[0])), kss = fbits(ks);
			if (proven && ks > 0.f && sat_cull_min_prims > 0 && nodes[n.a].bound.ok && nodes[n.a].prims >= (uint32_t)sat_cull_min_prims) {
				const int b = emit_node(n.b);
				const CullTest ct = make_test(nodes[n.a].bound);
				const int r = t++, q = n_tests++;
				snprintf(line, sizeof line,
				         "\t\tfloat t%d;\n"
				         "\t\t{ const float sb%d = t%d + %s, sw%d = t%d - sb%d;\n"
				         "\t\t  const float sx%d = p.x - %s, sy%d = p.y - %s, sz%d = p.z - %s;\n"
				         "\t\t  const float sl%d = (sx%d * sx%d + sy%d * sy%d) + sz%d * sz%d;\n"
				         "\t\t  const float su%d = (sb%d + %s) * %s;\n"
				         "\t\t  if ((vote(!(sl%d > su%d * su%d)) | vote(!(su%d > 0.f)) | vote(!(sw%d <= -%s)) | vote(!(sl%d < 0x1p120f))) != 0) {\n",
				         r, q, b, kss.c_str(), q, b, q,
				         q, fbits(ct.c[0]).c_str(), q, fbits(ct.c[1]).c_str(), q, fbits(ct.c[2]).c_str(),
				         q, q, q, q, q, q, q,
				         q, q, fbits(ct.rm).c_str(), fbits(ct.k).c_str(),
				         q, q, q, q, q, kss.c_str(), q);
				s += line;
				const int a = emit_node(n.a);
				if (sat_arith)
					snprintf(line, sizeof line, "\t\t  t%d = sminf_fastdiv_sat%s(t%d, t%d, %s, %s, %s, %s);\n\t\t  } else t%d = t%d + 0.f;\n\t\t}\n",
					         r, fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str(), kss.c_str(), r, b);
				else
					snprintf(line, sizeof line, "\t\t  t%d = sminf_fastdiv%s(t%d, t%d, %s, %s, %s);\n\t\t  } else t%d = t%d + 0.f;\n\t\t}\n",
					         r, fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str(), r, b);
				s += line;
				return r;
			}
			/* operands in program order (a flattened chain keeps its deep operand first and its operand stack shallow) */
			const bool a_first = o.op == LOL_OP_SMIN;
			const int first = emit_node(a_first ? n.a : n.b), second = emit_node(a_first ? n.b : n.a);
			const int a = a_first ? first : second, b = a_first ? second : first;
			if (proven && sat_arith)
				snprintf(line, sizeof line, "\t\tconst float t%d = sminf_fastdiv_sat%s%s(t%d, t%d, %s, %s, %s, %s);\n", t,
				         nodes[n.a].fon && nodes[n.b].fon ? "2" : "", fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str(), kss.c_str());
			else if (proven)
				snprintf(line, sizeof line, "\t\tconst float t%d = sminf_fastdiv%s(t%d, t%d, %s, %s, %s);\n", t, fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str());
			else
				snprintf(line, sizeof line, "\t\tconst float t%d = sminf_(t%d, t%d, %s);\n", t, a, b, kk.c_str());
			s += line; return t++;
		};
		object_has_nr = false;
		const int d = emit_node((int)nodes.size() - 1);
		if (object_has_nr) {                 /* a NaN from a sphere without range tracker reaches the object's value: 0 * NaN (or inf) = NaN */
			snprintf(line, sizeof line, "\t\tnanacc = __builtin_fmaf(t%d, 0.f, nanacc);\n", d);
			s += line;
		}
		if (R.id < max_id_seen)      /* evaluated after an object that follows it in the file: ties go to the lower id */
			snprintf(line, sizeof line, "\t\tif (t%d < best || (t%d == best && best_id > %uu)) { best = t%d; best_id = %uu; }\n", d, d, R.id, d, R.id);
		else
			snprintf(line, sizeof line, "\t\tif (t%d < best) { best = t%d; best_id = %uu; }\n", d, d, R.id);
		s += line;
		if (R.id > max_id_seen) max_id_seen = R.id;
		for (uint32_t k = 0; k < runs_ending[oi + 1]; k++) s += "\t\t} }\n";               /* every run that ends here */
	}
	if (out_of_line) {
		s += "\t\treturn { best, best_id, rg.lo, rg.hi, nanacc };\n}\n";
		snprintf(line, sizeof line, "struct %s {\n\tstatic constexpr bool ASSUME_SETTLED = %s;\n\tRange rg;\n\tfloat nanacc = 0.f;\n"
		         "\t__device__ __forceinline__ void loop_done() {}\n"
		         "\t__device__ __forceinline__ void eval(V3 p, float& best, u32& best_id) {\n"
		         "\t\tconst SdfOut o = %s_fn(p.x, p.y, p.z, rg.lo, rg.hi);\n"
		         "\t\tbest = o.best; best_id = o.id; rg.lo = o.lo; rg.hi = o.hi; nanacc += o.nanacc;\n\t}\n};\n", name, fast ? "true" : "false", name);
		s += line;
	} else {
		s += "\t}\n};\n";
	}
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

bool spec_out_of_line(const lol_program& P, int form = SPEC_BY_SIZE) {
	if (form == SPEC_OUT_OF_LINE) return true;
	if (form == SPEC_INLINE) return false;
	uint32_t limit = LOL_SPEC_INLINE_MAX_OPS;
	if (const char* e = tuning_env("LOL_GPU_SPEC_INLINE_MAX")) limit = (uint32_t)strtoul(e, nullptr, 10);
	return P.n_ops > limit;
}

std::string generate_source(const lol_program& P, const FastPaths* fast, bool cull, int form = SPEC_BY_SIZE) {
	std::string s;
	const bool ool = spec_out_of_line(P, form);
	const std::vector<RootBound> roots = analyse_roots(P);
	const CullPlan plan = plan_culling(roots, cull);
	s += "#include \"lol_kernel.h\"\n";
	s += "namespace lol {\n";
	/* Register budget.  The SDF of one object is a long dependent chain (every smooth min waits for the one below
	 * it) fed by independent primitives, and the kernel is compiled with the max-ILP scheduling strategy
	 * (compile_spec): the more registers a wave may use, the more primitives it keeps in flight.  Measured on
	 * MI355X (profiles/r2_large_scene_ab.jsonl): with max-ILP, scene4 (12 ops) is fastest when 8 waves per SIMD are
	 * kept (64 VGPRs: 4640 vs 4530 Mpixels/s unconstrained), a 44-op chain at >= 6, chains of 142+ ops at >= 4
	 * (128 VGPRs: 427 vs 400 Mpixels/s at 8). */
	const int waves_lo = P.n_ops <= 32 ? 8 : P.n_ops <= 96 ? 6 : 4, waves_hi = 8;
	const std::string occupancy = " __attribute__((amdgpu_waves_per_eu(" + std::to_string(waves_lo) + ", " + std::to_string(waves_hi) + ")))";
	if (ool) s += "struct SdfOut { float best; u32 id; u32 lo, hi; float nanacc; };\n";
	emit_sdf(s, P, "SpecSdfExact", nullptr, ool, roots, plan, occupancy);
	const bool any_fast = fast && (fast->sqrt_kind || !fast->div_ok.empty());
	if (any_fast) emit_sdf(s, P, "SpecSdfFast", fast, ool, roots, plan, occupancy);
	s += "}  // namespace lol\n";
	/* where lights / materials are read from is a property of the scene too (lol_kernel.h, TABLES_LDS_MAX_DWORDS) */
	const bool tables_global = !lol::tables_in_lds(P.n_lights, P.n_materials, P.n_roots);
	const std::string tg = tables_global ? "true" : "false";
	/* The pipeline once as a template on COUNT (lol_kernel.h, march: the per-lane step counters), and as one or two kernels:
	 *   lol_render_spec_steps  counts steps: frames with diagnostics (lol_gpu_debug::steps) and the one frame of a view that records
	 *                          what its pixels cost (lol_gpu.hip, "pixels dealt by cost");
	 *   lol_render_spec        does not (+1.3 % on C3): every other frame.
	 * A scene above LOL_SPEC_TWO_KERNELS_MAX_OPS gets the counting kernel alone, under the name lol_render_spec: a second copy of
	 * its pipeline would nearly double what the compiler takes for it. */
	const bool two = P.n_ops <= LOL_SPEC_TWO_KERNELS_MAX_OPS;
	s += "template <bool COUNT> __device__ __forceinline__ void lol_spec_body(const lol::Launch& L, lol::u32* lds) {\n";
	if (!tables_global) {
		s += "\tlol::stage_common(L, lds);\n";
		s += "\t__syncthreads();\n";
	}
	s += "\tif (!lol::start_tile_clock<" + tg + ">(L, lds)) return;\n";
	if (any_fast) {
		/* the fast pipeline takes FLAG_SHADOW_SETTLED for granted (lol_kernel.h, soft_shadow): a launch without it is the plain pipeline's */
		s += "\tlol::Pixel P;\n";
		s += "\tbool plain = !(L.flags & lol::FLAG_SHADOW_SETTLED);\n";
		s += "\tif (!plain) {\n";
		s += "\t\tlol::SpecSdfFast fast;\n";
		s += "\t\tP = lol::shade_pixel<lol::SpecSdfFast, " + tg + ", COUNT>(L, fast, lds);\n";
		s += "\t\tplain = lol::unproven(fast);\n";
		s += "\t}\n";
		s += "\tif (plain) {\n";
		s += "\t\tlol::SpecSdfExact exact;\n";
		s += "\t\tP = lol::shade_pixel<lol::SpecSdfExact, " + tg + ", COUNT>(L, exact, lds);\n";
		s += "\t}\n";
	} else {
		s += "\tlol::SpecSdfExact exact;\n";
		s += "\tlol::Pixel P = lol::shade_pixel<lol::SpecSdfExact, " + tg + ", COUNT>(L, exact, lds);\n";
	}
	s += "\tlol::store_pixel<" + tg + ">(L, P, lds);\n";
	s += "}\n";
	const std::string head = "extern \"C\" __global__ __launch_bounds__(lol::BLOCK)" + occupancy + " void ";
	const std::string tail = "(const lol::Launch L) {\n\textern __shared__ lol::u32 lds[];\n\tlol_spec_body<";
	if (two) s += head + "lol_render_spec_steps" + tail + "true>(L, lds);\n}\n";
	s += head + "lol_render_spec" + tail + (two ? "false" : "true") + ">(L, lds);\n}\n";
	/* the SDF alone at arbitrary points (lol_gpu_sdf_batch) */
	s += "extern \"C\" __global__ __launch_bounds__(64) void lol_sdf_spec(const float* pts, float* dist, lol::u32* id, lol::u32 n) {\n";
	s += "\tlol::SpecSdfExact exact;\n";
	if (any_fast) s += "\tlol::SpecSdfFast fast;\n\tlol::sdf_points(fast, exact, true, pts, dist, id, n);\n";
	else          s += "\tlol::sdf_points(exact, exact, false, pts, dist, id, n);\n";
	s += "}\n";
	return s;
}

/* hipRTC: generated source + lol_kernel.h → code object for `arch`.  Needs no device. */
/* ---- optional roctx ranges (LOL_GPU_ROCTX=1): one range per frame launch, visible to `rocprofv3 --marker-trace`.
 * The counterpart of the reference's perf/jitdump aid (jitdump.c) on this side; resolved with dlopen so the library
 * is only needed when asked for. */
struct Roctx {
	int  (*push)(const char*) = nullptr;
	int  (*pop)() = nullptr;
	std::atomic<long> ranges{0};        /* ranges pushed so far (lol_gpu_roctx_ranges) */
	bool asked = false;                 /* LOL_GPU_ROCTX=1 was set */
	std::once_flag once;
	void init() { std::call_once(once, [this] { resolve(); }); }      /* frames may be launched from several host threads */
	void resolve() {
		const char* e = getenv("LOL_GPU_ROCTX");
		if (!e || e[0] != '1') return;
		asked = true;
		for (const char* name : { "librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so" }) {
			if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
				push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
				pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
				if (push && pop) return;
				push = nullptr; pop = nullptr;
			}
		}
		/* asked for and not there: say so once instead of silently tracing nothing */
		fprintf(stderr, "lol_gpu: LOL_GPU_ROCTX=1 but no roctx library could be loaded (%s): frames are not marked\n", dlerror());
	}
} g_roctx;

unsigned long long fnv64(const void* data, size_t n);
std::string fnv_hex(const void* data, size_t n) {
	char b[20];
	snprintf(b, sizeof b, "%016llx", fnv64(data, n));
	return b;
}

/* Process-wide cache of compiled kernels: hosts (and the tests) upload the same scene many times. */
std::mutex g_cache_mutex;
std::unordered_map<std::string, std::vector<char>> g_code_cache;

/* ... and a cache on disk, so that the N ranks of a multi-GPU run (and the next run of the same host) do not each
 * pay the 0.3 - 1 s hipRTC compile of the same scene.  One file per key under LOL_GPU_CACHE_DIR (default
 * $XDG_CACHE_HOME/lol_gpu or $HOME/.cache/lol_gpu; set it to the empty string to switch the disk cache off): the
 * file holds the full key in front of the code object and is only used when that key matches byte for byte, so a
 * hash collision or a stale file can never hand out the wrong kernel; writes go through a temporary name + rename.
 * Any I/O failure simply means "not cached". */
std::string disk_cache_path(const std::string& key) {
	const char* e = getenv("LOL_GPU_CACHE_DIR");
	std::string dir;
	if (e) { if (!e[0]) return ""; dir = e; }
	else if (const char* x = getenv("XDG_CACHE_HOME")) { if (!x[0]) return ""; dir = std::string(x) + "/lol_gpu"; }
	else if (const char* h = getenv("HOME")) { if (!h[0]) return ""; dir = std::string(h) + "/.cache/lol_gpu"; }
	else return "";
	for (size_t i = 1; i <= dir.size(); i++)              /* mkdir -p */
		if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0700);
	unsigned long long h = 0xcbf29ce484222325ull;         /* FNV-1a of the key names the file */
	for (unsigned char c : key) { h ^= c; h *= 0x100000001b3ull; }
	char name[40];
	snprintf(name, sizeof name, "/%016llx.co", h);
	return dir + name;
}

/* GPU code is only ever loaded from a file this user wrote: the file and its directory must belong to the effective
 * user and must not be writable by group or others (a shared LOL_GPU_CACHE_DIR / XDG_CACHE_HOME would otherwise let
 * another user plant kernels), and the code object must match the checksum stored next to it. */
bool private_to_user(const struct stat& st) { return st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH)); }

unsigned long long fnv64(const void* data, size_t n) {
	unsigned long long h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < n; i++) { h ^= static_cast<const unsigned char*>(data)[i]; h *= 0x100000001b3ull; }
	return h;
}

bool disk_cache_load(const std::string& key, std::vector<char>& code) {
	const std::string path = disk_cache_path(key);
	if (path.empty()) return false;
	struct stat dir_st, file_st;
	const std::string dir = path.substr(0, path.rfind('/'));
	if (stat(dir.c_str(), &dir_st) != 0 || !S_ISDIR(dir_st.st_mode) || !private_to_user(dir_st)) return false;
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	bool ok = false;
	unsigned long long klen = 0, clen = 0, sum = 0;
	if (fstat(fileno(f), &file_st) == 0 && S_ISREG(file_st.st_mode) && private_to_user(file_st) &&
	    fread(&klen, 8, 1, f) == 1 && fread(&clen, 8, 1, f) == 1 && fread(&sum, 8, 1, f) == 1 &&
	    klen == key.size() && clen > 0 && clen < (1ull << 28)) {
		std::string k(klen, 0);
		code.resize(clen);
		ok = fread(&k[0], 1, klen, f) == klen && k == key && fread(code.data(), 1, clen, f) == clen && fnv64(code.data(), clen) == sum;
	}
	fclose(f);
	return ok;
}

void disk_cache_store(const std::string& key, const std::vector<char>& code) {
	const std::string path = disk_cache_path(key);
	if (path.empty()) return;
	char tmp[64];
	snprintf(tmp, sizeof tmp, ".%ld.tmp", (long)getpid());
	const std::string t = path + tmp;
	FILE* f = fopen(t.c_str(), "wb");
	if (!f) return;
	(void)fchmod(fileno(f), 0600);
	const unsigned long long klen = key.size(), clen = code.size(), sum = fnv64(code.data(), code.size());
	const bool ok = fwrite(&klen, 8, 1, f) == 1 && fwrite(&clen, 8, 1, f) == 1 && fwrite(&sum, 8, 1, f) == 1 &&
	                fwrite(key.data(), 1, klen, f) == klen && fwrite(code.data(), 1, clen, f) == clen;
	if (fclose(f) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0) (void)remove(t.c_str());
}

/* The long-branch bug described in compile_spec, as it looks in the code: a relaxed branch that goes through s[30:31], the
 * register pair a function RETURNS through —
 *     s_getpc_b64 s[30:31];  s_add_u32 s30, s30, <lit>;  s_addc_u32 s31, s31, <lit>;  s_setpc_b64 s[30:31]
 * (a call is s_getpc into some OTHER pair + s_swappc_b64 s[30:31], <pair>; a return is a bare s_setpc_b64 s[30:31] with no
 * s_getpc of that pair before it).  Recognised by instruction fields, not by four literal words (round-4 review): SOP1
 * s_getpc_b64 with SDST = s30, followed within a few instructions by SOP1 s_setpc_b64 with SSRC0 = s30 and no s_swappc_b64
 * in between — whatever arithmetic (add / sub, literal in either source position, a scavenged temporary, s_nop padding)
 * sits between the two.  The words are read with memcpy at EVERY byte offset: no assumption about where the buffer or the
 * text section inside the ELF begins, and nothing that can make the check pass by default.  A false alarm — constants that
 * happen to spell the two instructions eight dwords apart — only costs the scene its own kernel (the interpreter renders). */
bool has_return_clobbering_branch(const void* data, size_t n_bytes) {
	constexpr uint32_t SOP1 = 0xBE800000u, SOP1_MASK = 0xFF800000u;         /* [31:23] = 0b1_0111_1101 */
	constexpr uint32_t OP_GETPC = 28, OP_SETPC = 29, OP_SWAPPC = 30;        /* SOP1 opcodes (GFX9 / gfx950 encoding), bits [15:8] */
	constexpr uint32_t RETURN_PAIR = 30;                                    /* s[30:31] */
	constexpr size_t WINDOW = 12;                                           /* dwords after the s_getpc in which the s_setpc counts */
	const unsigned char* b = static_cast<const unsigned char*>(data);
	auto word = [&](size_t at) { uint32_t w; memcpy(&w, b + at, 4); return w; };
	for (size_t at = 0; at + 8 <= n_bytes; at++) {
		const uint32_t w = word(at);
		if ((w & SOP1_MASK) != SOP1 || ((w >> 8) & 0xFF) != OP_GETPC || ((w >> 16) & 0x7F) != RETURN_PAIR) continue;
		for (size_t k = 1; k <= WINDOW && at + 4 * k + 4 <= n_bytes; k++) {
			const uint32_t v = word(at + 4 * k);
			if ((v & SOP1_MASK) != SOP1) continue;
			const uint32_t op = (v >> 8) & 0xFF;
			if (op == OP_SWAPPC) break;                                     /* a call: the pair is being written as a link register */
			if (op == OP_SETPC && (v & 0xFF) == RETURN_PAIR) return true;
		}
	}
	return false;
}
bool has_return_clobbering_branch(const std::vector<char>& code) { return has_return_clobbering_branch(code.data(), code.size()); }

bool compile_spec(const lol_program& P, const FastPaths* fast, const std::string& arch, std::vector<char>& code,
                  std::string& log, std::string* src_out = nullptr, bool cull = true, int form = SPEC_BY_SIZE) {
	std::string src = generate_source(P, fast, cull, form);
	if (src_out) *src_out = src;
	if (const char* dump = tuning_env("LOL_GPU_DUMP_SPEC_SOURCE"))       /* debugging aid: the source as really generated on this device */
		if (FILE* f = fopen(dump, "w")) { fputs(src.c_str(), f); fclose(f); }
	int rtc_major = 0, rtc_minor = 0;
	(void)hiprtcVersion(&rtc_major, &rtc_minor);
	/* the option list first: it is part of the cache key (a changed flag — -ffp-contract above all — must never be served
	 * a code object compiled under the old one) */
	std::string arch_opt = "--offload-arch=" + arch;
	/* -ffp-contract=off: no FMA contraction (the reference has none); the rest are hipcc's defaults made explicit */
	/* -fno-slp-vectorize: the SLP pass pairs scalar f32 ops into v_pk_*_f32, which issue at half the
	 * rate of two scalar ops on gfx950 (tools/valu_rate.hip); measured +10 % Mpixels/s without it. */
	/* -amdgpu-sched-strategy=max-ilp: schedule for instruction-level parallelism within a wave rather than for
	 * occupancy.  The default strategy serialises the independent primitives of a long smooth-union chain to save
	 * registers; with max-ILP the same instructions run 1.4x faster on 142 - 1024-op scenes and 2 - 5 % faster on the
	 * example scenes (generate_source sets the matching register budget).  Scheduling only: same instructions, same bits.
	 * An LLVM that does not know an -mllvm option ends the PROCESS from its option parser, so the option is only
	 * passed to hipRTC versions it was verified on (hiprtcVersion >= 9.0 = ROCm 7.x); LOL_GPU_SCHED=default leaves it out. */
	std::vector<const char*> opts = { arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
	                                  "-fno-slp-vectorize" };
	const char* sched = tuning_env("LOL_GPU_SCHED");
	if (rtc_major >= 9 && !(sched && !strcmp(sched, "default"))) {
		opts.push_back("-mllvm"); opts.push_back("-amdgpu-sched-strategy=max-ilp");
		/* ... and no post-RA scheduling pass: it re-orders the ILP-friendly schedule after register allocation and
		 * costs 10 % on C3 (4650 -> 5150 Mpixels/s without it, same box and call; nothing on the large scenes) */
		opts.push_back("-mllvm"); opts.push_back("-enable-post-misched=0");
		/* ... and SimplifyCFG may turn small two-sided branches into selects more readily (default threshold 2): the per-lane
		 * `if (alive) { ... }` updates around the SDF become straight-line code for the ILP scheduler.  Sweep of 3 ... 64 on one
		 * box (tools/rtc_flag_sweep.sh phi): from 4 upwards C2 +1.5 ... 2.5 %, C3 +0.2 %, the large scenes +-3 %; same bits. */
		opts.push_back("-mllvm"); opts.push_back("-phi-node-folding-threshold=8");
		/* ... and NO register reserved ahead of time for long branches.  An out-of-line SDF function of more than 128 KB (about
		 * 800 ops: a field of 600 objects) has forward branches beyond s_cbranch's 16-bit reach; LLVM's AMDGPU backend then
		 * reserves "an unused" SGPR pair for the s_getpc / s_add / s_setpc sequence before register allocation — and in a leaf
		 * function picks s[30:31], the RETURN ADDRESS: the function jumps, and at its end "returns" to the branch target for
		 * ever (found in round 4 when scenes lost their 1024-op capacity: the kernel never finished; ROCm 7.0 and 7.2 alike).
		 * With the factor 0 no register is reserved and the branch relaxation scavenges a dead one at the branch, correctly.
		 * has_return_clobbering_branch() below refuses any code object that still shows the pattern. */
		opts.push_back("-mllvm"); opts.push_back("-amdgpu-long-branch-factor=0");
	}
	/* ... and so is the compiler: which libhiprtc this process has loaded.  A Python host gets the one torch ships, a C host the
	 * system's, a process under rocprofv3 yet another mix — the same source came out as three different code objects — and one
	 * compiler's output must not be handed to a process that would have compiled something else.
	 * LOL_GPU_CACHE_ANY_COMPILER=1 leaves the compiler out of the key: a profiling aid (tools/final_profile.sh lets a plain run
	 * compile the kernels, and the runs under the profiler load exactly those). */
	std::string compiler = "?";
	{
		Dl_info info;
		if (dladdr(reinterpret_cast<void*>(&hiprtcCompileProgram), &info) && info.dli_fname) compiler = info.dli_fname;
		compiler = std::to_string(rtc_major) + "." + std::to_string(rtc_minor) + " " + compiler;
		const char* any = tuning_env("LOL_GPU_CACHE_ANY_COMPILER");
		if (any && any[0] == '1') compiler = "*";         /* its version too: torch's hipRTC and the system's differ in it */
	}
	std::string key = "lol_gpu/4|hiprtc " + compiler + "|";
	for (const char* o : opts) { key += o; key += ' '; }
	key += "|" + src;
	{
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		auto it = g_code_cache.find(key);
		if (it != g_code_cache.end()) { code = it->second; log.clear(); return true; }
	}
	/* on disk the pipeline source (lol_kernel.h, embedded in this library) is part of the key: another build of the
	 * library must not pick up this one's kernels */
	const std::string disk_key = key + "|" + LOL_KERNEL_H_TEXT;
	if (disk_cache_load(disk_key, code)) {
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		g_code_cache[key] = code;
		log = "(code object from the disk cache)";
		return true;
	}
	const char* hdr_src[] = { LOL_KERNEL_H_TEXT };
	const char* hdr_name[] = { "lol_kernel.h" };
	hiprtcProgram prog = nullptr;
	if (hiprtcCreateProgram(&prog, src.c_str(), "lol_render_spec.hip", 1, hdr_src, hdr_name) != HIPRTC_SUCCESS) {
		log = "hiprtcCreateProgram failed";
		return false;
	}
	bool options_dropped = false;
	hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
	if (r != HIPRTC_SUCCESS) {
		/* a hipRTC that does not know the scheduling option must not cost the specialisation: once more without it */
		std::vector<const char*> plain;
		for (size_t i = 0; i < opts.size(); i++) {
			if (!strcmp(opts[i], "-mllvm") && i + 1 < opts.size() &&
			    (!strcmp(opts[i + 1], "-amdgpu-sched-strategy=max-ilp") || !strcmp(opts[i + 1], "-enable-post-misched=0") ||
			     !strcmp(opts[i + 1], "-phi-node-folding-threshold=8") || !strcmp(opts[i + 1], "-amdgpu-long-branch-factor=0"))) { i++; continue; }
			plain.push_back(opts[i]);
		}
		if (plain.size() != opts.size()) {
			/* what comes out now was NOT compiled under the options the key lists: it serves this process (the retry would
			 * give the same again) but never goes to disk under that key */
			options_dropped = true;
			hiprtcDestroyProgram(&prog);
			prog = nullptr;
			if (hiprtcCreateProgram(&prog, src.c_str(), "lol_render_spec.hip", 1, hdr_src, hdr_name) != HIPRTC_SUCCESS) {
				log = "hiprtcCreateProgram failed";
				return false;
			}
			r = hiprtcCompileProgram(prog, (int)plain.size(), plain.data());
		}
	}
	size_t log_size = 0;
	hiprtcGetProgramLogSize(prog, &log_size);
	log.clear();
	if (log_size > 1) { log.resize(log_size); hiprtcGetProgramLog(prog, &log[0]); }
	if (r != HIPRTC_SUCCESS) {
		log = std::string("hipRTC: ") + hiprtcGetErrorString(r) + "\n" + log;
		hiprtcDestroyProgram(&prog);
		return false;
	}
	size_t code_size = 0;
	hiprtcGetCodeSize(prog, &code_size);
	code.resize(code_size);
	hiprtcGetCode(prog, code.data());
	hiprtcDestroyProgram(&prog);
	if (has_return_clobbering_branch(code)) {
		/* a kernel that would never finish is worse than no kernel: the interpreter renders this scene */
		log = "the compiler relaxed a long branch through s[30:31], the return address of a function (LLVM AMDGPU long-branch "
		      "register bug; see compile_spec): code object refused";
		code.clear();
		return false;
	}
	if (options_dropped && spec_out_of_line(P, form)) {
		/* The retry above also dropped -amdgpu-long-branch-factor=0, the option that KEEPS the compiler from that bug, and an
		 * out-of-line SDF is where it bites (a function beyond s_cbranch's reach).  The pattern check would be all that is left
		 * between this code object and a launch that never ends: not enough — the interpreter renders this scene. */
		log = "this hipRTC refused the -mllvm options (among them the workaround for LLVM's long-branch register bug) and the scene's "
		      "SDF is out of line: code object refused";
		code.clear();
		return false;
	}
	{
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		g_code_cache[key] = code;
	}
	if (!options_dropped) disk_cache_store(disk_key, code);
	return true;
}

/* Prove, on this device, the shortcuts `prog` could use (results are cached per context). */
/* What one context has proven about a device holds for every context of this process on that device: same silicon, same
 * code.  (A second context of a host — a second window, the ranks of a test — then spends its render_prepare on the scene.) */
struct DeviceProofs {
	int  sqrt_verified = -1; bool sqrt_tiny_ok = false;
	int  gamma_verified = -1;
	std::vector<lol_gpu::DivProof> div;
};
std::mutex g_proofs_mutex;
std::unordered_map<int, DeviceProofs> g_proofs;

void proofs_from_process(lol_gpu* ctx) {
	std::lock_guard<std::mutex> lock(g_proofs_mutex);
	auto it = g_proofs.find(ctx->device);
	if (it == g_proofs.end()) return;
	const DeviceProofs& P = it->second;
	if (ctx->sqrt_verified < 0 && P.sqrt_verified >= 0) { ctx->sqrt_verified = P.sqrt_verified; ctx->sqrt_tiny_ok = P.sqrt_tiny_ok; }
	if (ctx->gamma_verified < 0) ctx->gamma_verified = P.gamma_verified;
	for (const auto& e : P.div) {
		bool known = false;
		for (const auto& c : ctx->div_verified) known = known || c.k_bits == e.k_bits;
		if (!known) ctx->div_verified.push_back(e);
	}
}
void proofs_to_process(const lol_gpu* ctx) {
	std::lock_guard<std::mutex> lock(g_proofs_mutex);
	DeviceProofs& P = g_proofs[ctx->device];
	if (ctx->sqrt_verified >= 0) { P.sqrt_verified = ctx->sqrt_verified; P.sqrt_tiny_ok = ctx->sqrt_tiny_ok; }
	if (ctx->gamma_verified >= 0) P.gamma_verified = ctx->gamma_verified;
	for (const auto& e : ctx->div_verified) {
		bool known = false;
		for (const auto& c : P.div) known = known || c.k_bits == e.k_bits;
		if (!known) P.div.push_back(e);
	}
}

FastPaths prove_fast_paths(lol_gpu* ctx, const lol_program& prog) {
	FastPaths fast;
	if (!ctx->want_fast) return fast;                  /* (lol_gpu_set_specialize 0 / 3) */
	proofs_from_process(ctx);
	if (ctx->sqrt_verified < 0) {
		ctx->sqrt_verified = 0;                     /* cheapest proven sequence wins */
		for (int kind = 3; kind >= 1 && !ctx->sqrt_verified; kind--) {
			unsigned long long tiny_bad = 1;
			if (run_verify(ctx, kind, 0.f, &tiny_bad) == 0) { ctx->sqrt_verified = kind; ctx->sqrt_tiny_ok = tiny_bad == 0; }
		}
	}
	fast.sqrt_kind = ctx->sqrt_verified;
	fast.sqrt_tiny_ok = ctx->sqrt_tiny_ok;
	/* gamma + quantisation through the table (lol_kernel.h) */
	if (ctx->gamma_verified < 0) ctx->gamma_verified = run_verify_gamma(ctx) == 0 ? 1 : 0;
	else if (ctx->gamma_verified == 1 && !ctx->d_gamma && !build_gamma_table(ctx)) ctx->gamma_verified = 0;      /* proven by another context of this device: only the table */
	fast.gamma_ok = ctx->gamma_verified == 1;
	for (uint32_t i = 0; i < prog.n_ops; i++) {
		const lol_op& o = prog.ops[i];
		if (o.op != LOL_OP_SMIN && o.op != LOL_OP_SMIN_R) continue;
		uint32_t kb;
		memcpy(&kb, &o.f[0], 4);
		bool known = false, ok = false, nf = false;
		for (auto& e : ctx->div_verified) if (e.k_bits == kb) { known = true; ok = e.ok; nf = e.no_fixup_ok; }
		if (!known) {
			/* |k| >= 2^-100: see sminf_fastdiv (lol_kernel.h); then the exhaustive proof of the blend factor */
			unsigned long long nf_bad = 1;
			ok = (o.f[0] >= 0x1p-100f || o.f[0] <= -0x1p-100f) && run_verify(ctx, 0, o.f[0], &nf_bad) == 0;
			nf = ok && nf_bad == 0;
			ctx->div_verified.push_back({ kb, ok, nf });
		}
		if (ok && !fast.has(o.f[0])) fast.div_ok.push_back(o.f[0]);
		if (nf && !fast.has_nf(o.f[0])) fast.div_nf_ok.push_back(o.f[0]);
	}
	proofs_to_process(ctx);
	return fast;
}

/*
 * Tiered start-up.  The reference's render_prepare returns at once (naive_renderer.c:242-244 is empty; the tracing JIT's
 * takes milliseconds, tracing_jit_renderer.dasc:416-434); hipRTC takes 0.6 s for scene4 and half a minute for 5000 ops.
 * So lol_gpu_upload_program commits the tables and the interpreter's lists, starts the scene compiler on a host thread
 * and returns: frames render on render_interp at once — same bits — and the first frame launched after the compiler has
 * finished loads the module and runs lol_render_spec (a swap at a frame boundary, in the calling thread: no second
 * process, nothing re-executed).  lol_gpu_specialize_wait() blocks until then (tests, benchmarks).
 */
/* hipRTC is entered by one thread at a time, and the second context that wants the same scene finds it in the cache */
std::mutex g_rtc_mutex;

/* the compiler runs of programs this context has since replaced: the finished ones are joined; `all` (lol_gpu_destroy): every
 * one is waited for.  A run cannot be left behind: hipRTC cannot be interrupted, and a thread still inside it when the process
 * exits crashes in the compiler's own teardown (comgr is loaded on first use, so its statics go BEFORE this library's — tried in
 * round 5 with a process-lifetime reaper: the C host segfaulted at exit).  What bounds the wait instead is LOL_SPEC_MAX_OPS. */
void reap(lol_gpu* ctx, bool all) {
	for (size_t i = 0; i < ctx->old_jobs.size();) {
		SpecJob* j = ctx->old_jobs[i];
		bool done;
		{ std::lock_guard<std::mutex> lock(j->mu); done = j->done; }
		if (done || all) {
			if (j->th.joinable()) j->th.join();
			delete j;
			ctx->old_jobs.erase(ctx->old_jobs.begin() + (long)i);
		} else i++;
	}
}

void launch_job(SpecJob* job);

/* Start compiling the specialised kernel of ctx's (just committed) program.  The previous scene's module is gone already
 * (the caller has drained the device); until finish_specialise() swaps the new one in, the interpreter renders. */
void start_specialise(lol_gpu* ctx, const FastPaths& fast) {
	if (ctx->spec_module) { (void)hipModuleUnload(ctx->spec_module); ctx->spec_module = nullptr; }
	if (ctx->spec_module_old) { (void)hipModuleUnload(ctx->spec_module_old); ctx->spec_module_old = nullptr; }
	ctx->second_tier_pending = ctx->second_tier_running = false;
	ctx->kernel_epoch++;                                /* the interpreter renders the new scene until its kernel is there */
	ctx->spec_fn = ctx->spec_steps_fn = nullptr;
	ctx->spec_sdf_fn = nullptr;
	snprintf(ctx->kernel_name, sizeof ctx->kernel_name, "render_interp");
	ctx->spec_log.clear();
	ctx->spec_state = 0;
	if (ctx->job) { ctx->old_jobs.push_back(ctx->job); ctx->job = nullptr; }      /* a compile of the scene before: its result is not wanted any more */
	reap(ctx, false);
	const char* env = tuning_env("LOL_GPU_SPECIALIZE");
	if (!ctx->want_spec || (env && env[0] == '0')) return;
	/* every program up to LOL_SPEC_MAX_OPS is specialised, large ones with their SDF out of line (emit_sdf).  Beyond that the
	 * straight-line source (two SDF bodies of ~150 bytes per op) takes hipRTC minutes, and programs no longer have a
	 * capacity (lol_scene.h): such a scene renders on the interpreter, which reads it as data.  Not a failure: no complaint. */
	{
		uint32_t limit = ctx->spec_max_ops ? ctx->spec_max_ops : LOL_SPEC_MAX_OPS;
		if (const char* e = tuning_env("LOL_GPU_SPEC_MAX_OPS")) limit = (uint32_t)strtoul(e, nullptr, 10);
		if (ctx->h_prog.n_ops > limit) {
			char b[160];
			snprintf(b, sizeof b, "%u ops: above the %u the scene compiler takes on (lol_gpu_set_specialize_max_ops / LOL_GPU_SPEC_MAX_OPS); rendered by the interpreter", ctx->h_prog.n_ops, limit);
			ctx->spec_log = b;
			return;
		}
	}
	hipDeviceProp_t prop;
	std::string arch = "gfx950";
	if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.gcnArchName[0]) {
		std::string name = prop.gcnArchName;             /* e.g. "gfx950:sramecc+:xnack-" */
		arch = name.substr(0, name.find(':'));
	}
	SpecJob* job = nullptr;
	try {
		job = new SpecJob;
		job->prog = std::make_shared<OwnedProgram>();    /* the thread's own copy: the context may take another scene meanwhile */
		job->prog->assign(ctx->h_prog);
		job->fast = std::make_shared<FastPaths>(fast);
		job->arch = arch;
	} catch (...) { delete job; ctx->spec_log = "out of host memory"; return; }
	{
		char b[160];
		snprintf(b, sizeof b, "fast paths proven on device: sqrt=%d, smin divisors=%zu (without div_fixup: %zu)\n", fast.sqrt_kind,
		         fast.div_ok.size(), fast.div_nf_ok.size());
		job->note = b;
		const std::string sw = lol_gpu_tuning_switches();
		if (!sw.empty()) job->note += "tuning switches in effect (LOL_GPU_TUNING=1): " + sw + "\n";
	}
	job->cull = culling_enabled(ctx->want_cull);
	/* Two tiers for mid-size scenes (round 5).  With its SDF inlined into the three loops a scene of 257 ... 1024 ops renders
	 * 14 - 88 % faster than with the one out-of-line function (profiles/r5_large_scene_ab.jsonl, r5_field_inline_ab.jsonl) —
	 * and takes hipRTC 3 - 18 s instead of 0.4 - 3 s.  So such a scene gets the out-of-line kernel first and the inlined one
	 * when that is ready: interpreter -> out-of-line kernel -> inlined kernel, each swap at a frame boundary, same pixels on
	 * all three.  (LOL_GPU_SPEC_INLINE_MAX, a tuning switch, pins ONE form by size as before.) */
	ctx->second_tier_pending = !tuning_env("LOL_GPU_SPEC_INLINE_MAX") && ctx->h_prog.n_ops > LOL_SPEC_FIRST_TIER_INLINE_MAX_OPS &&
	                           ctx->h_prog.n_ops <= LOL_SPEC_INLINE_MAX_OPS;
	ctx->second_tier_running = false;
	job->form = ctx->second_tier_pending ? SPEC_OUT_OF_LINE : SPEC_BY_SIZE;
	ctx->job = job;
	ctx->spec_state = 1;
	launch_job(job);
}

/* start the run on its own (large-stack) thread; without a thread to be had, or with LOL_GPU_ASYNC_COMPILE=0, it runs / is waited
 * for here */
void launch_job(SpecJob* job) {
	job->started = std::chrono::steady_clock::now();
	auto work = [job]() {
		bool ok = false;
		std::vector<char> code;
		std::string log;
		try {
			std::lock_guard<std::mutex> rtc(g_rtc_mutex);
			ok = compile_spec(job->prog->p, job->fast.get(), job->arch, code, log, nullptr, job->cull, job->form);
		} catch (...) { ok = false; log = "the scene compiler ran out of memory"; }
		std::lock_guard<std::mutex> lock(job->mu);
		job->compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job->started).count();
		job->code = std::move(code);
		job->log = std::move(log);
		job->ok = ok;
		job->done = true;
		job->cv.notify_all();
	};
	const char* async = tuning_env("LOL_GPU_ASYNC_COMPILE");
	bool threaded = !(async && async[0] == '0');
	/* LOL_GPU_ASYNC_COMPILE=0: the upload itself waits for the compiler (still on the large-stack thread) */
	bool started = false;
	try { started = job->th.start(work); } catch (...) { started = false; }
	if (!started) work();                                  /* no thread to be had: compile here */
	else if (!threaded) job->th.join();
}

/* The frame boundary: when the compiler has finished (or `wait`), load the module and switch the context over to it; a
 * scene of the middle sizes then has its second run started (the inlined form), which takes over the same way when IT is
 * done — `wait` waits for both.  Returns true when the state changed.  The device of the context is current. */
bool finish_specialise(lol_gpu* ctx, bool wait) {
	bool changed = false;
	for (;;) {
		SpecJob* job = ctx->job;
		if (!job) return changed;
		{
			std::unique_lock<std::mutex> lock(job->mu);
			if (!job->done) {
				if (!wait) return changed;
				job->cv.wait(lock, [job] { return job->done; });
			}
		}
		if (job->th.joinable()) job->th.join();
		ctx->job = nullptr;
		ctx->spec_compile_ms = job->compile_ms;
		const bool second = ctx->second_tier_running;
		ctx->second_tier_running = false;
		changed = true;
		/* the scene's second run (the form with the SDF inlined) behind the first: same program and proofs */
		auto start_second_tier = [&]() {
			ctx->second_tier_pending = false;
			SpecJob* next = nullptr;
			try {
				next = new SpecJob;
				next->prog = job->prog; next->fast = job->fast; next->arch = job->arch; next->cull = job->cull;
				next->form = SPEC_INLINE;
			} catch (...) { delete next; next = nullptr; }
			if (!next) return false;
			ctx->job = next;
			ctx->second_tier_running = true;
			try { launch_job(next); } catch (...) { ctx->job = nullptr; ctx->second_tier_running = false; delete next; return false; }
			return true;
		};
		/* an unexpected failure is reported once on stderr: frames still render, through the (slower) interpreter — or, when it
		 * is the second run that failed, through the first run's kernel, which stays.  A FIRST run that fails where a second was
		 * to follow (257 ... 1024 ops: the out-of-line form, the one the long-branch trip-wire and the dropped-options refusal of
		 * compile_spec are about) does not cost the scene its kernel: the inlined form, which has no out-of-line function to
		 * trip them, is compiled all the same while the interpreter renders (round-5 advisor). */
		auto complain = [&](const std::string& why) {
			if (second) {
				if (ctx->spec_fn) {
					ctx->spec_log += "(the inlined form of the kernel was not to be had: " + why + "; the out-of-line form stays)\n";
					ctx->spec_state = 2;
				} else {
					ctx->spec_log += "(nor was the inlined form: " + why + ")\n";
					ctx->spec_state = -1;
					fprintf(stderr, "lol_gpu: scene specialisation failed, using the interpreter kernel: %s\n", ctx->spec_log.c_str());
				}
			} else if (ctx->second_tier_pending && start_second_tier()) {
				ctx->spec_log = "(the out-of-line form of the kernel was not to be had: " + why + "; compiling the inlined form)\n";
				ctx->spec_state = 1;
			} else {
				ctx->spec_log = why;
				ctx->spec_state = -1;
				ctx->second_tier_pending = false;
				fprintf(stderr, "lol_gpu: scene specialisation failed, using the interpreter kernel: %s\n", ctx->spec_log.c_str());
			}
			delete job;
		};
		if (!second && ctx->second_tier_pending && ctx->fail_first_tier > 0) {      /* lol_gpu_testing_fail_first_tier */
			ctx->fail_first_tier--;
			job->ok = false;
			job->log = "injected failure of the first run (lol_gpu_testing_fail_first_tier)";
		}
		if (!job->ok) { complain(job->log); if (!wait || !ctx->job) return true; continue; }
		hipModule_t mod = nullptr;
		hipFunction_t fn = nullptr, steps_fn = nullptr, sdf_fn = nullptr;
		if (hipModuleLoadData(&mod, job->code.data()) != hipSuccess) { complain("hipModuleLoadData failed"); if (!wait || !ctx->job) return true; continue; }
		if (hipModuleGetFunction(&fn, mod, "lol_render_spec") != hipSuccess) {
			(void)hipModuleUnload(mod);
			complain("lol_render_spec not found in the compiled module");
			if (!wait || !ctx->job) return true;
			continue;
		}
		/* (a module with one kernel: that one counts.  Asked for only where it was generated — a failed look-up leaves
		 * hipErrorNotFound behind as the thread's last error, which the host's next HIP call would trip over — and cleared anyway) */
		if (job->prog->p.n_ops > LOL_SPEC_TWO_KERNELS_MAX_OPS || hipModuleGetFunction(&steps_fn, mod, "lol_render_spec_steps") != hipSuccess) steps_fn = nullptr;
		if (hipModuleGetFunction(&sdf_fn, mod, "lol_sdf_spec") != hipSuccess) sdf_fn = nullptr;
		(void)hipGetLastError();
		if (ctx->spec_module) {
			/* the first tier's module: frames launched through it may still be in flight, so it is only unloaded by the next
			 * upload (which drains the device first) or with the context */
			if (ctx->spec_module_old) (void)hipModuleUnload(ctx->spec_module_old);      /* (cannot happen: one second tier per upload) */
			ctx->spec_module_old = ctx->spec_module;
		}
		ctx->spec_module = mod;
		ctx->spec_fn = fn;
		ctx->spec_steps_fn = steps_fn;
		ctx->spec_sdf_fn = sdf_fn;
		ctx->spec_log = (second ? ctx->spec_log + "second tier (SDF inlined): " : job->note) + job->log + (job->log.empty() || job->log.back() == '\n' ? "" : "\n");
		snprintf(ctx->kernel_name, sizeof ctx->kernel_name, "lol_render_spec");
		ctx->spec_key = fnv_hex(job->code.data(), job->code.size());
		ctx->spec_state = 2;
		ctx->kernel_epoch++;
		/* the out-of-line kernel renders from now on; the inlined form is compiled behind it */
		if (ctx->second_tier_pending && start_second_tier()) ctx->spec_state = 5;
		delete job;
		if (!wait) return true;
	}
}

}  // namespace

extern "C" {

int lol_gpu_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int lol_gpu_create(int device, lol_gpu** out) {
	if (!out) return LOL_GPU_ERR_ARG;
	*out = nullptr;
	int n = lol_gpu_device_count();
	if (n <= 0 || device < 0 || device >= n) return LOL_GPU_ERR_NO_DEVICE;
	lol_gpu* ctx = new (std::nothrow) lol_gpu;
	if (!ctx) return LOL_GPU_ERR_HIP;
	ctx->device = device;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	ctx->frame_streams[0] = ctx->stream;
	/* (the device tables are sized by the first upload) */
	if (e != hipSuccess) {
		fprintf(stderr, "lol_gpu_create: %s\n", hipGetErrorString(e));
		lol_gpu_destroy(ctx);
		return LOL_GPU_ERR_HIP;
	}
	*out = ctx;
	return LOL_GPU_OK;
}

static void lpt_release(lol_gpu* ctx);

void lol_gpu_destroy(lol_gpu* ctx) {
	if (!ctx) return;
	if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
	for (int i = lol_gpu::MAX_FRAME_STREAMS - 1; i >= 0; i--) {      /* ([0] is ctx->stream) */
		hipStream_t fs = i ? ctx->frame_streams[i] : ctx->stream;
		if (fs) { (void)hipStreamSynchronize(fs); (void)hipStreamDestroy(fs); }
	}
	if (ctx->job) { ctx->old_jobs.push_back(ctx->job); ctx->job = nullptr; }
	reap(ctx, true);                         /* a compiler thread still running is waited for: it must not outlive the library */
	if (ctx->spec_module) (void)hipModuleUnload(ctx->spec_module);
	if (ctx->spec_module_old) (void)hipModuleUnload(ctx->spec_module_old);
	for (int i = 0; i < 2; i++) {
		if (ctx->d_tables[i]) (void)hipFree(ctx->d_tables[i]);
		if (ctx->d_mops[i]) (void)hipFree(ctx->d_mops[i]);
	}
	if (ctx->d_frame) (void)hipFree(ctx->d_frame);
	if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
	for (int i = 0; i < lol_gpu::PIPE_SLOTS; i++) {
		if (ctx->d_pipe[i]) (void)hipFree(ctx->d_pipe[i]);
		if (ctx->pipe_rendered[i]) (void)hipEventDestroy(ctx->pipe_rendered[i]);
		if (ctx->pipe_copied[i]) (void)hipEventDestroy(ctx->pipe_copied[i]);
	}

	if (ctx->d_bad) (void)hipFree(ctx->d_bad);
	if (ctx->d_gamma) (void)hipFree(ctx->d_gamma);
	if (ctx->tiles.have_events) for (hipEvent_t e : ctx->tiles.ev) (void)hipEventDestroy(e);
	lpt_release(ctx);
	delete ctx;
}

const char* lol_gpu_error(const lol_gpu* ctx) { return ctx ? ctx->err : "null context"; }

int lol_gpu_device(const lol_gpu* ctx) { return ctx ? ctx->device : -1; }

int lol_gpu_set_specialize(lol_gpu* ctx, int enable) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	/* 0 interpreter, plain | 1 specialised + proven fast paths (default) | 3 specialised, plain | 4 interpreter + fast paths */
	ctx->want_spec = (enable == 1 || enable == 3) ? 1 : 0;
	ctx->want_fast = (enable == 1 || enable == 4) ? 1 : 0;
	return LOL_GPU_OK;
}

int lol_gpu_set_specialize_max_ops(lol_gpu* ctx, unsigned max_ops) {
	if (!ctx || max_ops > LOL_MAX_OPS) return LOL_GPU_ERR_ARG;
	ctx->spec_max_ops = max_ops;             /* takes effect at the next lol_gpu_upload_program */
	return LOL_GPU_OK;
}

/* which of the wanted skips the uploaded program (and the environment) allows */
static void resolve_skips(lol_gpu* ctx) {
	const unsigned want = ctx->want_skips;
	ctx->miss_skip = (want & 1u) && miss_skip_ok(ctx->h_prog);
	ctx->dark_skip = (want & 2u) && dark_skip_ok(ctx->h_prog);
	ctx->shadow_settle = (want & 4u) && shadow_settle_ok(ctx->h_prog);
}

int lol_gpu_set_exact_skips(lol_gpu* ctx, unsigned mask) {
	if (!ctx || mask > 7u) return LOL_GPU_ERR_ARG;
	ctx->want_skips = mask;
	if (ctx->have_prog) resolve_skips(ctx);
	return LOL_GPU_OK;
}

int lol_gpu_set_miss_skip(lol_gpu* ctx, int enable) { return lol_gpu_set_exact_skips(ctx, enable ? 7u : 0u); }

/* Host-only view of the bound behind the culling test of top-level object `root` (0-based, file order):
 * 1 = bounded (centre and inflated radius R' out), 0 = no bound (never culled), < 0 = bad argument. */
int lol_gpu_cull_bounds(const lol_program* prog, uint32_t root, float c_out[3], float* r_out) {
	if (!prog || !c_out || !r_out || root >= prog->n_roots || prog->n_ops > LOL_MAX_OPS) return LOL_GPU_ERR_ARG;
	const std::vector<RootBound> roots = analyse_roots(*prog);
	if (root >= roots.size()) return LOL_GPU_ERR_ARG;
	const RootBound& R = roots[root];
	if (!R.bounded) return 0;
	const CullTest t = make_test(R.sphere());
	c_out[0] = t.c[0]; c_out[1] = t.c[1]; c_out[2] = t.c[2];
	*r_out = t.rm;
	return 1;
}

/* ... and the tighter two-sphere bound, where the object has one (cluster_bounds): value(p) >= min_j (|p - c_j| - r_j) with the
 * inflated radii the kernel tests with.  Returns the number of spheres written to out[j] = {cx, cy, cz, r'} (0 or 2). */
int lol_gpu_cull_bounds_clusters(const lol_program* prog, uint32_t root, float out[3][4]) {
	if (!prog || !out || root >= prog->n_roots || prog->n_ops > LOL_MAX_OPS) return LOL_GPU_ERR_ARG;
	const std::vector<RootBound> roots = analyse_roots(*prog);
	if (root >= roots.size()) return LOL_GPU_ERR_ARG;
	int n = 0;
	for (const Sphere& c : roots[root].clusters) {
		if (n == 3) break;
		const CullTest t = make_test(c);
		out[n][0] = t.c[0]; out[n][1] = t.c[1]; out[n][2] = t.c[2]; out[n][3] = t.rm;
		n++;
	}
	return n;
}

int lol_gpu_set_tile_order(lol_gpu* ctx, int order) {
	if (!ctx || order < LOL_GPU_TILES_ROWS || order > LOL_GPU_TILES_LPT) return LOL_GPU_ERR_ARG;
	lol_gpu::TileAuto& T = ctx->tiles;
	T.mode = order;
	T.deciding = false;                      /* a running series of trials is abandoned (its events are simply reused) */
	T.key[0] = 0;                            /* ... and AUTO starts afresh at the next frame */
	T.chosen = order == LOL_GPU_TILES_COLS ? LOL_GPU_TILES_COLS : LOL_GPU_TILES_ROWS;
	/* longest-first starts afresh too: its tables stay allocated, and stay on their streams — frames launched through them may
	 * still be in flight, and whatever rewrites a set does so on the set's own stream, behind them (TileLpt::launched) */
	for (lol_gpu::TileLpt& P : ctx->lpt) P.key[0] = 0;
	ctx->lpt_have_last = false;
	ctx->lpt_last_set = -1;
	return LOL_GPU_OK;
}

/*
 * Longest tiles first.  A frame is ONE launch of one-wave blocks (129,600 for C3) that the hardware hands out in block order;
 * blocks differ 100x in cost (sky against penumbra), the launch ends when the LAST wave ends, and while the slowest waves
 * of the tail run the SIMDs stand half empty — frames issued on three streams so that the next frame's waves fill that
 * tail render 9 % (C3) to 67 % (scene.lol at 1080p) faster (tools/stream_overlap_ab.py, profiles/r4_stream_overlap_ab.jsonl),
 * but the reference's frame loop is sequential (main.c:189-194).  The same packing INSIDE one frame: hand the tiles out in the
 * order of decreasing cost (list scheduling, longest processing time first), the cost being what the tile cost in the frame
 * before — the camera moves a little per frame (main.c:70-112), a still camera not at all.  Every block writes how long its
 * wave ran (shader clock, 32 x log2: store_pixel, tile_cost); a counting sort on the device (three small kernels on the
 * frame's stream, once after the first frame of a scene / size and then every LPT_RESORT frames) turns the costs into the
 * next order table; the kernel reads its tile from the table (tile_of_block).  Same pixels: only the ORDER in which tiles
 * are rendered changes.  Measured, one stream, one box (tools/tile_order_ab.py, profiles/r4_tile_order_ab.jsonl,
 * r4_lpt_sweep*.txt; best of rows / columns -> longest first): C3 7850 -> 8470 Mpixels/s, scene.lol at 1080p 14,400 ->
 * 18,900, rank 0's bands of an 8-way C4 split 7370 -> 8580, a whole C4 frame 8600 -> 8770, the orbit (the costs lag the
 * camera by up to LPT_RESORT frames) 10,050 -> 10,300.  (The run time predicts better than the evaluation count, the first
 * cost tried: C3 8200.)
 */
constexpr unsigned LPT_BUCKETS = 1024, LPT_THREADS = 256, LPT_RESORT = 16;

/* (the tables are stored XCD by XCD: lol_kernel.h, tile_slot)
 * launch position i shades wave slot i */
__global__ __launch_bounds__(LPT_THREADS) void lpt_identity_kernel(uint32_t* order, uint32_t n, uint32_t stride) {
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	if (i < n) order[lol::tile_slot(i, stride)] = i;
}

/*
 * Pixels dealt by cost.  A wave runs every loop to its slowest lane: the 16x4 pixels of a rectangle execute 66.9 SDF
 * evaluations per pixel on C3 where their pixels need 57.0 — lane efficiency 0.85, 0.79 in the shadow marches, whose long
 * tails are single pixels (a ray grazing a surface) among quick neighbours.  For a camera that stands still the step
 * counts of the frame before are EXACT, so the pixels can be dealt to waves ahead of time: the frame is cut into regions
 * of REGION_W x REGION_H pixels (16 waves' worth), the pixels of a region are sorted by the evaluations they needed, and
 * wave k of the region gets the k-th 64 of them — waves of like pixels, no compaction at run time, and still neighbours
 * within 64 x 16 pixels (the culling votes of a wave keep working: pixels of like cost are pixels of like fate).  The
 * oracle's step counts put the evaluations a frame executes at -11.4 % for this region size (-9.3 % for 32 x 8, -11.2 %
 * for 128 x 32; keeping runs of 2 / 4 / 8 adjacent pixels together leaves -7.7 / -4.6 / -2.2 %: the stragglers really are
 * single pixels — tests/tools/sorted_region_model.py, profiles/r4_sorted_region_model.json).  The price is memory traffic,
 * of which this path has 250x to spare: a 4-byte table entry read per pixel, and every lane storing its own pixel.
 * lane_pixels[64 * slot + lane] = column | local row << 16 | LANE_PADDING; wave slot = 16 * region + k.
 */
struct RegionShape { uint32_t w, h; };       /* multiples of 16 x 4; w * h a power of two <= 4096 (the sort's LDS) */
static RegionShape region_shape() { return { 64, 16 }; }      /* (swept in round 4: profiles/r4_region_sweep.txt) */

/* the first frame of a view: wave k of a region = its k-th 16x4 rectangle (what a launch without tables shades) */
__global__ __launch_bounds__(LPT_THREADS) void deal_rectangles_kernel(uint32_t* lane_pixels, uint32_t n_lanes, uint32_t w, uint32_t n_rows, uint32_t regions_x,
                                                                      uint32_t REGION_W, uint32_t REGION_H) {
	const uint32_t REGION_PIXELS = REGION_W * REGION_H;
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	if (i >= n_lanes) return;
	const uint32_t region = i / REGION_PIXELS, j = i % REGION_PIXELS, k = j / 64, lane = j % 64;
	uint32_t x = (region % regions_x) * REGION_W + (k % (REGION_W / 16)) * 16 + lane % 16;
	uint32_t r = (region / regions_x) * REGION_H + (k / (REGION_W / 16)) * 4 + lane / 16;
	uint32_t pad = 0;
	if (x >= w) { x = w - 1; pad = lol::LANE_PADDING; }
	if (r >= n_rows) { r = n_rows - 1; pad = lol::LANE_PADDING; }
	lane_pixels[i] = x | r << 16 | pad;
}
/* one block per region: sort its pixels by what they cost (bitonic, in LDS; the pixels beyond the frame's edge first) and
 * deal them to the region's waves in that order */
__global__ __launch_bounds__(LPT_THREADS) void deal_by_cost_kernel(const unsigned short* pixel_cost, uint32_t* lane_pixels, uint32_t w, uint32_t n_rows,
                                                                   uint32_t regions_x, uint32_t REGION_W, uint32_t REGION_H) {
	__shared__ uint32_t key[4096];                         /* cost + 1 (0 = beyond the edge) << 12 | position in the region */
	const uint32_t REGION_PIXELS = REGION_W * REGION_H;
	const uint32_t region = blockIdx.x, x0 = (region % regions_x) * REGION_W, r0 = (region / regions_x) * REGION_H;
	for (uint32_t j = threadIdx.x; j < REGION_PIXELS; j += LPT_THREADS) {
		const uint32_t x = x0 + j % REGION_W, r = r0 + j / REGION_W;
		const uint32_t c = (x < w && r < n_rows) ? (uint32_t)pixel_cost[(size_t)r * w + x] + 1u : 0u;
		key[j] = c << 12 | j;
	}
	__syncthreads();
	for (uint32_t k = 2; k <= REGION_PIXELS; k <<= 1)
		for (uint32_t d = k >> 1; d > 0; d >>= 1) {
			for (uint32_t t = threadIdx.x; t < REGION_PIXELS / 2; t += LPT_THREADS) {
				const uint32_t lo = 2 * t - (t & (d - 1)), hi = lo + d;      /* the pair (lo, lo + d) of this compare-exchange network step */
				const bool up = (lo & k) == 0;
				const uint32_t a = key[lo], b = key[hi];
				if ((a > b) == up) { key[lo] = b; key[hi] = a; }
			}
			__syncthreads();
		}
	for (uint32_t j = threadIdx.x; j < REGION_PIXELS; j += LPT_THREADS) {
		const uint32_t q = key[j] & (REGION_PIXELS - 1);
		uint32_t x = x0 + q % REGION_W, r = r0 + q / REGION_W, pad = 0;
		if (x >= w) { x = w - 1; pad = lol::LANE_PADDING; }
		if (r >= n_rows) { r = n_rows - 1; pad = lol::LANE_PADDING; }
		lane_pixels[(size_t)region * REGION_PIXELS + j] = x | r << 16 | pad;
	}
}
/* pass 1: snapshot every block's cost as a bucket number (bucket 0 = the most expensive), count the buckets */
__global__ __launch_bounds__(LPT_THREADS) void lpt_hist_kernel(const uint32_t* cost, uint32_t* keys, uint32_t* hist, uint32_t n, uint32_t stride) {
	__shared__ uint32_t h[LPT_BUCKETS];
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) h[b] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	if (i < n) {
		uint32_t k = cost[lol::tile_slot(i, stride)];          /* <= 703 (lol_kernel.h, store_pixel); clamped all the same */
		k = LPT_BUCKETS - 1 - (k < LPT_BUCKETS ? k : LPT_BUCKETS - 1);
		keys[i] = k;
		atomicAdd(&h[k], 1u);
	}
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) if (h[b]) atomicAdd(&hist[b], h[b]);
}
/* pass 2 (one block): hist[LPT_BUCKETS + b] = where bucket b starts (exclusive prefix sum) */
__global__ __launch_bounds__(LPT_BUCKETS) void lpt_scan_kernel(uint32_t* hist) {
	__shared__ uint32_t s[LPT_BUCKETS];
	const uint32_t b = threadIdx.x;
	s[b] = hist[b];
	__syncthreads();
	for (uint32_t d = 1; d < LPT_BUCKETS; d <<= 1) {
		const uint32_t v = b >= d ? s[b - d] : 0u;
		__syncthreads();
		s[b] += v;
		__syncthreads();
	}
	hist[LPT_BUCKETS + b] = s[b] - hist[b];
}
/* pass 3: every block reserves room for its members of each bucket with ONE atomic per bucket and places them in their
 * old order; what it places is the TILE the old table named for that launch position */
__global__ __launch_bounds__(LPT_THREADS) void lpt_scatter_kernel(const uint32_t* keys, const uint32_t* order_in, uint32_t* order_out,
                                                                  uint32_t* hist, uint32_t n, uint32_t stride) {
	__shared__ uint32_t h[LPT_BUCKETS], base[LPT_BUCKETS];
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) h[b] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	uint32_t k = 0, rank = 0;
	if (i < n) { k = keys[i]; rank = atomicAdd(&h[k], 1u); }
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) if (h[b]) base[b] = atomicAdd(&hist[LPT_BUCKETS + b], h[b]);
	__syncthreads();
	if (i < n) order_out[lol::tile_slot(base[k] + rank, stride)] = order_in[lol::tile_slot(i, stride)];
}

static unsigned lpt_resort_period() { return LPT_RESORT; }

static void lpt_release_set(lol_gpu::TileLpt& T) {
	for (uint32_t** p : { &T.d_order[0], &T.d_order[1], &T.d_cost, &T.d_keys, &T.d_hist, &T.d_lanes })
		if (*p) { (void)hipFree(*p); *p = nullptr; }
	if (T.d_pixel_cost) { (void)hipFree(T.d_pixel_cost); T.d_pixel_cost = nullptr; }
	T.cap = 0; T.lanes_cap = 0; T.pixels_cap = 0; T.n_tiles = 0; T.key[0] = 0;
}
static void lpt_release(lol_gpu* ctx) {
	for (lol_gpu::TileLpt& T : ctx->lpt) { lpt_release_set(T); T.home = nullptr; T.launched = false; }
}

/* The table for the frame about to be launched on `s` (device current), or nullptr: a launch in one of the fixed orders.
 *
 * Longest-first is for a camera that stands still: then a tile costs this frame exactly what it cost the frame before.
 * Handed out by STALE costs the dear tiles come late, which is worse than any fixed order — measured on the 256-frame
 * orbit taken at 1 / 2 / 4 / 8 frames per step (1.4 / 2.8 / 5.6 / 11 degrees; the reference's arrow keys turn the camera by
 * atan(0.1) = 5.7 degrees a frame, main.c:70-112) with a sort before every frame: +3 % / -4 % / -7 % / -10 % against the
 * column order, the +3 % being what is left of +7 % after paying for the sort (profiles/r4_orbit_stride_ab.jsonl,
 * r4_lpt_verdict_ab.jsonl; a device-side verdict on how far the costs had moved was built and dropped: it needs the sort it
 * wants to avoid).  So: a frame whose camera differs from the frame before it is launched in the fixed order (the caller
 * falls back to AUTO's choice), without table, cost or sort — no overhead while the camera moves; the first frame under
 * the camera of its predecessor goes through the row-order table and reports its tiles' costs; the next one is sorted.
 *
 * Sets up (or re-creates) the tables when the scene, size, partition or kernel changed.  Everything about one set of
 * tables happens on ONE stream — the one its first frame was launched on: frames, the costs they write and the sorts that
 * read them are then ordered by the stream itself, and a table is never rewritten under a frame that still reads it.  A
 * frame of the same key on another stream is launched without a table, unless the host has moved over for good. */
struct FrameTables { const uint32_t* order; uint32_t* cost; const uint32_t* lanes; unsigned short* pixel_cost; uint32_t n_waves; };

static bool lpt_table_for_frame(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps, const lol_gpu_rows* R, int n_rows,
                                int block, hipStream_t s, FrameTables* out) {
	const uint32_t REGION_W = region_shape().w, REGION_H = region_shape().h, REGION_WAVES = REGION_W * REGION_H / 64;
	const uint32_t regions_x = ((uint32_t)w + REGION_W - 1) / REGION_W, regions_y = ((uint32_t)n_rows + REGION_H - 1) / REGION_H;
	const uint32_t n = regions_x * regions_y * REGION_WAVES;            /* wave slots = blocks of the launch */
	const size_t n_lanes = (size_t)n * 64, n_pixels = (size_t)w * (size_t)n_rows;
	ctx->lpt_last_set = -1;
	if (block != 64 || w > 0xFFFF || n_rows > 0x7FFF || n_lanes > 0xFFFFFFFFull) return false;      /* (an entry is column | row << 16 | flag; one-wave blocks) */
	const int key[7] = { w, h, max_steps, R->band_rows, R->cycle_rows, R->offset_rows, ctx->kernel_epoch };
	/* what the frame before this one was (on whatever stream): the same view of the same frame? */
	const bool still = ctx->lpt_have_last && memcmp(key, ctx->lpt_last_key, sizeof key) == 0 && memcmp(cam, &ctx->lpt_last_cam, sizeof *cam) == 0;
	memcpy(ctx->lpt_last_key, key, sizeof key);
	ctx->lpt_last_cam = *cam;
	ctx->lpt_have_last = true;
	if (!still) return false;
	auto ok = [](hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess; };
	/* the set that lives on this stream; else a free one; else — once two such frames in a row have found no set, i.e. the host
	 * has moved to streams without one and is not merely rotating over more streams than there are sets (then the first
	 * LPT_SETS streams keep theirs and the others run in the fixed order: taking turns at a set would cost a stream
	 * synchronisation per frame) — the least recently used one, after ITS stream has run dry */
	lol_gpu::TileLpt* Tp = nullptr;
	for (lol_gpu::TileLpt& P : ctx->lpt) if (P.home == s) Tp = &P;
	if (!Tp) for (lol_gpu::TileLpt& P : ctx->lpt) if (!P.home && !Tp) Tp = &P;
	if (!Tp) {
		if (++ctx->lpt_homeless < 2) return false;
		for (lol_gpu::TileLpt& P : ctx->lpt) if (!Tp || P.stamp < Tp->stamp) Tp = &P;
		/* (a caller's stream may have been destroyed since: then whatever it still had queued is waited for with the device) */
		if (!ok(hipStreamSynchronize(Tp->home)) && !ok(hipDeviceSynchronize())) return false;
		Tp->launched = false;
		Tp->key[0] = 0;                                  /* whatever it knew was another stream's schedule */
	}
	ctx->lpt_homeless = 0;
	lol_gpu::TileLpt& T = *Tp;
	T.home = s;
	T.stamp = ++ctx->lpt_clock;
	const bool new_key = memcmp(key, T.key, sizeof key) != 0;
	if (new_key) {
		/* frames of the old key may still read these tables — on this very stream, so the kernels that rewrite them queue up
		 * behind those frames; only FREEING the tables needs the stream to have run dry first */
		const bool grow = n > T.cap || n_lanes > T.lanes_cap || n_pixels > T.pixels_cap;
		if (grow) {
			if (T.launched && !ok(hipStreamSynchronize(s))) return false;
			T.launched = false;
			lpt_release_set(T);
			const size_t cap = (size_t)n + n / 4 + 1024;      /* (tile_slot reaches 8 * ceil(n / 8) - 1 < n + 8) */
			const bool good = ok(hipMalloc(reinterpret_cast<void**>(&T.d_order[0]), cap * 4)) && ok(hipMalloc(reinterpret_cast<void**>(&T.d_order[1]), cap * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_cost), cap * 4)) && ok(hipMalloc(reinterpret_cast<void**>(&T.d_keys), cap * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_hist), 2 * LPT_BUCKETS * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_lanes), n_lanes * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_pixel_cost), n_pixels * 2));
			if (!good) { lpt_release_set(T); T.home = nullptr; return false; }
			T.cap = cap; T.lanes_cap = n_lanes; T.pixels_cap = n_pixels;
		}
		memcpy(T.key, key, sizeof key);
		T.n_tiles = n;
	}
	const dim3 grid((n + LPT_THREADS - 1) / LPT_THREADS);
	const uint32_t stride = (n + 7u) >> 3;
	bool record_pixels = false;
	T.launched = true;                                   /* (from here on something of this set is queued on s) */
	if (new_key || memcmp(cam, &T.cam_epoch, sizeof *cam) != 0) {
		/* a view these tables know nothing about: rectangles, handed out in region order; this frame reports what every
		 * pixel and every wave cost */
		T.cam_epoch = *cam;
		T.cur = 0; T.frames = 0;
		hipLaunchKernelGGL(lpt_identity_kernel, grid, dim3(LPT_THREADS), 0, s, T.d_order[0], n, stride);
		hipLaunchKernelGGL(deal_rectangles_kernel, dim3((unsigned)((n_lanes + LPT_THREADS - 1) / LPT_THREADS)), dim3(LPT_THREADS), 0, s,
		                   T.d_lanes, (uint32_t)n_lanes, (uint32_t)w, (uint32_t)n_rows, regions_x, REGION_W, REGION_H);
		if (!ok(hipGetLastError()) || !ok(hipMemsetAsync(T.d_cost, 0, ((size_t)n + 8) * 4, s))) { T.key[0] = 0; return false; }
		record_pixels = true;
	} else if (T.frames == 1) {
		/* the second frame of the view: its pixels dealt to the waves of their region by what they cost (exact: nothing moved);
		 * the waves are new ones, so they go out in region order once more and report how long THEY run */
		hipLaunchKernelGGL(deal_by_cost_kernel, dim3(regions_x * regions_y), dim3(LPT_THREADS), 0, s, T.d_pixel_cost, T.d_lanes, (uint32_t)w, (uint32_t)n_rows, regions_x, REGION_W, REGION_H);
		if (!ok(hipGetLastError())) { T.key[0] = 0; return false; }
	} else if (T.frames == 2 || T.frames % lpt_resort_period() == 0) {
		/* the run times of the frame before are in (same stream): three small kernels, then the other table is the current
		 * one.  (Again every LPT_RESORT frames: the run times drift a little with what runs beside a wave.) */
		if (ok(hipMemsetAsync(T.d_hist, 0, 2 * LPT_BUCKETS * 4, s))) {
			hipLaunchKernelGGL(lpt_hist_kernel, grid, dim3(LPT_THREADS), 0, s, T.d_cost, T.d_keys, T.d_hist, n, stride);
			hipLaunchKernelGGL(lpt_scan_kernel, dim3(1), dim3(LPT_BUCKETS), 0, s, T.d_hist);
			hipLaunchKernelGGL(lpt_scatter_kernel, grid, dim3(LPT_THREADS), 0, s, T.d_keys, T.d_order[T.cur], T.d_order[T.cur ^ 1], T.d_hist, n, stride);
			if (ok(hipGetLastError())) { T.cur ^= 1; T.sorts++; ctx->lpt_sorts++; }
		}
	}
	T.frames++;
	ctx->lpt_last_set = (int)(Tp - ctx->lpt);
	*out = { T.d_order[T.cur], T.d_cost, T.d_lanes, record_pixels ? T.d_pixel_cost : nullptr, n };
	return true;
}

/* AUTO: collect the trial frames that have finished (never waits) and decide once all of them have */
static void tile_auto_harvest(lol_gpu* ctx) {
	lol_gpu::TileAuto& T = ctx->tiles;
	if (!T.deciding) return;
	while (T.harvested < T.issued) {
		const int i = T.harvested;
		if (hipEventQuery(T.ev[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); return; }      /* (hipErrorNotReady is not an error) */
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, T.ev[2 * i], T.ev[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
		T.ms[i] = ms;
		T.harvested++;
	}
	if (T.harvested < lol_gpu::TileAuto::TOTAL) return;
	/* the typical frame of each order (reported): the mean of the faster half of its trials (a frame that shared the device
	 * with something else, or ran before the clocks had settled, does not count) */
	for (int o = 0; o < 2; o++) {
		float v[LOL_GPU_TILE_TRIALS];
		int n = 0;
		for (int i = lol_gpu::TileAuto::SKIP; i < lol_gpu::TileAuto::TOTAL; i++) if (lol_gpu::TileAuto::order_of_trial(i) == o && T.ms[i] > 0.f) v[n++] = T.ms[i];
		std::sort(v, v + n);
		const int half = n > 1 ? n / 2 : n;
		float sum = 0.f;
		for (int i = 0; i < half; i++) sum += v[i];
		T.typical[o] = half ? sum / (float)half : 0.f;
	}
	/* The decision: the trials come in PAIRS of consecutive frames, one of each order — a host whose camera moves (the orbit: a
	 * frame costs 0.7 to 1.0 ms depending on where the camera is) renders nearly the same view twice in a pair, so the ratio
	 * columns / rows of a pair is about the orders and not about the view; which order goes first alternates from pair to
	 * pair, so a cost that drifts one way cancels; the median ratio decides (round 4: the means of two interleaved series
	 * picked rows for the orbit, 4 % behind). */
	float ratio[LOL_GPU_TILE_TRIALS];
	int n_ratios = 0;
	for (int i = lol_gpu::TileAuto::SKIP; i + 1 < lol_gpu::TileAuto::TOTAL; i += 2) {
		const float a = T.ms[i], b = T.ms[i + 1];
		if (!(a > 0.f && b > 0.f)) continue;
		ratio[n_ratios++] = lol_gpu::TileAuto::order_of_trial(i) == LOL_GPU_TILES_ROWS ? b / a : a / b;      /* columns / rows */
	}
	std::sort(ratio, ratio + n_ratios);
	const float median = n_ratios ? (n_ratios & 1 ? ratio[n_ratios / 2] : 0.5f * (ratio[n_ratios / 2 - 1] + ratio[n_ratios / 2])) : 1.f;
	T.chosen = median < 0.99f ? LOL_GPU_TILES_COLS : LOL_GPU_TILES_ROWS;
	T.deciding = false;
	T.decisions++;
}

/* the order of the frame about to be launched; *trial = the trial slot whose events bracket it, or -1.  Device is current. */
static int tile_order_for_frame(lol_gpu* ctx, int w, int h, int max_steps, const lol_gpu_rows* R, bool diagnostics, int* trial) {
	lol_gpu::TileAuto& T = ctx->tiles;
	*trial = -1;
	if (T.mode != LOL_GPU_TILES_AUTO && T.mode != LOL_GPU_TILES_LPT) return T.chosen;
	const int key[6] = { w, h, max_steps, R->band_rows, R->cycle_rows, ctx->kernel_epoch };      /* (the kernel too: interpreter, or which form of the scene's own) */
	if (memcmp(key, T.key, sizeof key) != 0) {          /* another scene, size or partition: measure again */
		memcpy(T.key, key, sizeof key);
		if (!T.have_events) {
			bool ok = true;
			for (hipEvent_t& e : T.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
			if (!ok) { (void)hipGetLastError(); for (hipEvent_t& e : T.ev) { if (e) (void)hipEventDestroy(e); e = nullptr; } T.mode = LOL_GPU_TILES_ROWS; return T.chosen; }
			T.have_events = true;
		}
		T.deciding = true;
		T.issued = T.harvested = 0;
		T.chosen = LOL_GPU_TILES_ROWS;
		T.mon_frames = T.mon_n = 0;
		T.mon_pending = false;
	}
	tile_auto_harvest(ctx);
	if (T.deciding) {
		if (T.issued >= lol_gpu::TileAuto::TOTAL || diagnostics) return T.chosen;
		*trial = T.issued++;
		return lol_gpu::TileAuto::order_of_trial(*trial);
	}
	/* Decided — and watched from then on: which fixed order is better depends on the VIEW as well (the orbit's first forty
	 * frames favour rows by 5 %, the orbit as a whole columns by 3.5 %), and the host moves the camera (main.c:180).  Every
	 * MONITOR_PERIOD frames one frame in the order in use and the next one in the other order are timed like trial frames
	 * (two event pairs, collected without waiting); when the other order has been faster by more than 1 % in the median of
	 * the last MONITOR_WINDOW such pairs, the orders change places.  A probe frame costs what the orders differ by. */
	if (diagnostics) return T.chosen;
	if (T.mon_pending) {
		if (hipEventQuery(T.ev[3]) == hipSuccess) {
			float a = 0.f, b = 0.f;
			if (hipEventElapsedTime(&a, T.ev[0], T.ev[1]) == hipSuccess && hipEventElapsedTime(&b, T.ev[2], T.ev[3]) == hipSuccess && a > 0.f && b > 0.f) {
				T.mon_ratio[T.mon_n % lol_gpu::TileAuto::MONITOR_WINDOW] = b / a;      /* other / in use */
				T.mon_n++;
				if (T.mon_n >= lol_gpu::TileAuto::MONITOR_WINDOW) {
					float r[lol_gpu::TileAuto::MONITOR_WINDOW];
					memcpy(r, T.mon_ratio, sizeof r);
					std::sort(r, r + lol_gpu::TileAuto::MONITOR_WINDOW);
					if (r[lol_gpu::TileAuto::MONITOR_WINDOW / 2] < 0.99f) {
						T.chosen = T.chosen == LOL_GPU_TILES_COLS ? LOL_GPU_TILES_ROWS : LOL_GPU_TILES_COLS;
						T.mon_n = 0;
						T.swaps++;
					}
				}
			} else (void)hipGetLastError();
			T.mon_pending = false;
		} else { (void)hipGetLastError(); return T.chosen; }      /* (the pair is still in flight: no new one) */
	}
	const unsigned phase = T.mon_frames++ % lol_gpu::TileAuto::MONITOR_PERIOD;
	if (phase == lol_gpu::TileAuto::MONITOR_PERIOD - 2) { *trial = 0; return T.chosen; }
	if (phase == lol_gpu::TileAuto::MONITOR_PERIOD - 1) {
		*trial = 1;
		T.mon_pending = true;
		return T.chosen == LOL_GPU_TILES_COLS ? LOL_GPU_TILES_ROWS : LOL_GPU_TILES_COLS;
	}
	return T.chosen;
}

int lol_gpu_tile_order(lol_gpu* ctx, lol_gpu_tile_order_info* out) {
	if (!ctx || !out) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	tile_auto_harvest(ctx);
	const lol_gpu::TileAuto& T = ctx->tiles;
	if (T.mode == LOL_GPU_TILES_LPT) {
		/* longest first: the last frame went through a table (order LPT; "deciding" until its costs have been sorted once), or
		 * the camera moves and AUTO's fixed order is in use (its state and trial times) */
		if (ctx->lpt_last_set >= 0) *out = { T.mode, LOL_GPU_TILES_LPT, ctx->lpt[ctx->lpt_last_set].frames < 3 ? 1 : 0, (int32_t)ctx->lpt_sorts, T.typical[0], T.typical[1] };
		else *out = { T.mode, T.chosen, (T.deciding || ctx->lpt_sorts == 0) ? 1 : 0, (int32_t)ctx->lpt_sorts, T.typical[0], T.typical[1] };
	} else
		*out = { T.mode, T.chosen, T.deciding ? 1 : 0, T.decisions, T.typical[0], T.typical[1] };
	return LOL_GPU_OK;
}

int lol_gpu_set_cull(lol_gpu* ctx, int enable) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	ctx->want_cull = enable ? 1 : 0;          /* takes effect at the next lol_gpu_upload_program */
	return LOL_GPU_OK;
}

/* bit 0: escaped-wave skip active; bit 1: zero-incidence shadow skip active */
int lol_gpu_miss_skip_active(const lol_gpu* ctx) {
	return ctx ? (ctx->miss_skip ? 1 : 0) | (ctx->dark_skip ? 2 : 0) | (ctx->shadow_settle ? 4 : 0) : 0;
}

/* Run the exhaustive (all 2^32 inputs) equivalence checks directly: mismatch counts out. */
int lol_gpu_verify_fast_paths(lol_gpu* ctx, float k, unsigned long long sqrt_mismatches[3],
                              unsigned long long* div_mismatches) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	if (sqrt_mismatches)
		for (int kind = 1; kind <= 3; kind++) sqrt_mismatches[kind - 1] = run_verify(ctx, kind, 0.f);
	if (div_mismatches) *div_mismatches = run_verify(ctx, 0, k);
	return LOL_GPU_OK;
}

/* ... and for the blend factor without v_div_fixup (smin_h_fast<false>): inputs on which it differs from the exact
 * factor (finite and NaN dlt) or fails to turn the smooth minimum NaN (dlt = +-inf); 0 = proven, ~0 = could not run */
int lol_gpu_verify_smin_no_fixup(lol_gpu* ctx, float k, unsigned long long* mismatches) {
	if (!ctx || !mismatches) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	unsigned long long second = ~0ull;
	*mismatches = run_verify(ctx, 0, k, &second) == ~0ull ? ~0ull : second;
	return LOL_GPU_OK;
}

/* ... and for the gamma table (lol_kernel.h, gamma_u8_table): floats in [0, 1] on which the table route and the powf route
 * give different channel values; 0 = proven, ~0 = could not run.  table (may be NULL) receives the 257 thresholds. */
int lol_gpu_verify_gamma_table(lol_gpu* ctx, unsigned long long* mismatches, float* table) {
	if (!ctx || !mismatches) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	*mismatches = run_verify_gamma(ctx);
	if (table && ctx->d_gamma) LOL_HIP(ctx, hipMemcpy(table, ctx->d_gamma, (lol::GAMMA_LEVELS + 1) * sizeof(float), hipMemcpyDeviceToHost));
	return LOL_GPU_OK;
}

const char* lol_gpu_specialize_log(const lol_gpu* ctx) { return ctx ? ctx->spec_log.c_str() : ""; }

int lol_gpu_specialize_wait(lol_gpu* ctx) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	finish_specialise(ctx, true);
	return LOL_GPU_OK;
}

int lol_gpu_specialize_state(lol_gpu* ctx, double* compile_ms) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	if (ctx->job) {                                    /* has the compiler finished?  (the swap itself happens at a frame or a wait) */
		std::lock_guard<std::mutex> lock(ctx->job->mu);
		/* (while the second run is at work: what the first one took) */
		if (compile_ms) *compile_ms = ctx->job->done ? ctx->job->compile_ms : ctx->second_tier_running ? ctx->spec_compile_ms : 0.0;
		if (ctx->second_tier_running) return ctx->job->done ? 6 : 5;
		return ctx->job->done ? 3 : 1;
	}
	if (compile_ms) *compile_ms = ctx->spec_compile_ms;
	return ctx->spec_state;
}

static int upload_program(lol_gpu* ctx, const lol_program* prog);

/* No exception crosses the C boundary: programs may have 2^20 ops, and the analysis of a scene (culling plan, the interpreter's
 * lists, the tables) allocates as it goes — a std::bad_alloc anywhere in it is an upload that failed, with the scene the
 * context had still rendering (every step before the commit works on the side). */
int lol_gpu_upload_program(lol_gpu* ctx, const lol_program* prog) {
	try { return upload_program(ctx, prog); }
	catch (const std::bad_alloc&) { return fail(ctx, LOL_GPU_ERR_HIP, "out of host memory while preparing the scene"); }
	catch (...) { return fail(ctx, LOL_GPU_ERR_HIP, "unexpected failure while preparing the scene"); }
}

static int upload_program(lol_gpu* ctx, const lol_program* prog) {
	if (!ctx || !prog) return LOL_GPU_ERR_ARG;
	/* sanity caps (lol_scene.h): counts beyond them are corruption, not scenes; every table a count speaks of must be there */
	if (prog->n_ops > LOL_MAX_OPS || prog->n_lights > LOL_MAX_LIGHTS || prog->n_materials > LOL_MAX_MATERIALS ||
	    prog->n_roots > LOL_MAX_OPS || prog->max_stack > LOL_MAX_STACK || prog->n_materials == 0)
		return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "program exceeds the sanity caps of lol_scene.h (or has no material)");
	if ((prog->n_ops && !prog->ops) || (prog->n_lights && !prog->lights) || !prog->materials || (prog->n_roots && !prog->root_material))
		return fail(ctx, LOL_GPU_ERR_ARG, "malformed program: a table is missing");
	/* validate what the kernel indexes with: stack discipline and material indices */
	int depth = 0;
	for (uint32_t i = 0; i < prog->n_ops; i++) {
		switch (prog->ops[i].op) {
		case LOL_OP_SPHERE: case LOL_OP_RBOX: case LOL_OP_PLANE: depth++; break;
		case LOL_OP_SMIN: case LOL_OP_SMIN_R:
			if (depth < 2) return fail(ctx, LOL_GPU_ERR_ARG, "malformed program: smin underflow");
			depth--; break;
		case LOL_OP_TOP:
			if (depth != 1 || prog->ops[i].id == 0 || prog->ops[i].id > prog->n_roots)
				return fail(ctx, LOL_GPU_ERR_ARG, "malformed program: bad top");
			depth = 0; break;
		default: return fail(ctx, LOL_GPU_ERR_ARG, "malformed program: unknown opcode");
		}
		if (depth > (int)prog->max_stack) return fail(ctx, LOL_GPU_ERR_ARG, "malformed program: max_stack too small");
	}
	if (depth != 0) return fail(ctx, LOL_GPU_ERR_ARG, "malformed program: dangling operands");
	for (uint32_t i = 0; i < prog->n_roots; i++)
		if (prog->root_material[i] >= prog->n_materials)
			return fail(ctx, LOL_GPU_ERR_ARG, "material index out of range");

	LOL_HIP(ctx, hipSetDevice(ctx->device));
	/* frames already queued — on the context's stream or on a caller's — still read the old tables and code
	 * object: drain the whole device before replacing them */
	LOL_HIP(ctx, hipDeviceSynchronize());
	/* Everything that can fail happens on the side: the tables for both kernels and the interpreter's macro-op list
	 * (smooth unions whose blend factor is proven on this device carry {k, 2k, .5/k}; the proven sqrt is selected by
	 * instantiation at launch) go to the table set no frame reads.  Only then is the context switched over, so a
	 * rejected program leaves the previous scene rendering (the reference asserts instead: scene.c:284-292). */
	FastPaths fast = prove_fast_paths(ctx, *prog);
	const std::vector<RootBound> roots = analyse_roots(*prog);
	/* Two lists of the same records: the second one takes the blend factors without v_div_fixup where the device proved
	 * them.  That proof covers every FINITE difference of operands; the launch picks the second list only when nothing an
	 * evaluation computes can be infinite (finite_scene, and a sane camera for that frame), the SDF of arbitrary points
	 * (lol_gpu_sdf_batch) never does.  (The specialised kernel votes on NaN per object instead; here a vote per object
	 * costs more than the fixup saves.) */
	const CullPlan cull_plan = plan_culling(roots, culling_enabled(ctx->want_cull));
	std::vector<uint32_t> mops = build_mops(*prog, &fast, roots, cull_plan, false);
	const uint32_t n_mops = (uint32_t)(mops.size() / lol::MOP_DWORDS);
	{
		const std::vector<uint32_t> nofix = build_mops(*prog, &fast, roots, cull_plan, true);
		if (nofix.size() != mops.size()) return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "interpreter lists differ in length");      /* (same records by construction) */
		mops.insert(mops.end(), nofix.begin(), nofix.end());
	}
	const int next = ctx->cur ^ 1;
	const bool injected = ctx->fail_uploads > 0;      /* lol_gpu_testing_fail_uploads (tests/test_gpu_boundary.py) */
	if (injected) ctx->fail_uploads--;
	/* lights | materials | root_material as one array of dwords, in the set no frame reads; grown when this scene needs more */
	std::vector<uint32_t> tables(lol::table_dwords(prog->n_lights, prog->n_materials, prog->n_roots));
	{
		uint32_t* t = tables.data();
		if (prog->n_lights) memcpy(t, prog->lights, (size_t)prog->n_lights * sizeof(lol_light));
		t += (size_t)prog->n_lights * lol::LIGHT_DWORDS;
		memcpy(t, prog->materials, (size_t)prog->n_materials * sizeof(lol_material));
		t += (size_t)prog->n_materials * lol::MATERIAL_DWORDS;
		if (prog->n_roots) memcpy(t, prog->root_material, (size_t)prog->n_roots * 4);
	}
	auto fit = [&](uint32_t*& buf, size_t& cap, size_t need) -> hipError_t {
		if (need <= cap) return hipSuccess;
		uint32_t* nb = nullptr;
		const size_t want = need + need / 2 + 256;
		const hipError_t me = hipMalloc(reinterpret_cast<void**>(&nb), want * 4);
		if (me != hipSuccess) return me;
		if (buf) (void)hipFree(buf);                  /* (the device is idle and no frame reads this set) */
		buf = nb; cap = want;
		return hipSuccess;
	};
	hipError_t e = injected ? hipErrorOutOfMemory : fit(ctx->d_tables[next], ctx->tables_cap[next], tables.size());
	if (e == hipSuccess) e = fit(ctx->d_mops[next], ctx->mops_cap[next], mops.size());
	if (e == hipSuccess) e = hipMemcpy(ctx->d_tables[next], tables.data(), tables.size() * 4, hipMemcpyHostToDevice);
	if (e == hipSuccess && !mops.empty())
		e = hipMemcpy(ctx->d_mops[next], mops.data(), mops.size() * 4, hipMemcpyHostToDevice);
	if (e != hipSuccess) return fail(ctx, LOL_GPU_ERR_HIP, "upload of the scene tables", e);
	const int interp_sqrt_kind = fast.sqrt_kind == 3 ? 3 : 0;
	std::string interp_key;
	{
		/* what render_interp executes = this build's code (lol_kernel.h AND this file: record layout, flags) + the lists */
		std::string id = std::string(LOL_BUILD_ID) + "|" + fnv_hex(mops.data(), mops.size() * 4) + "|" + std::to_string(interp_sqrt_kind) +
		                 (fast.gamma_ok ? "|gamma" : "");
		interp_key = fnv_hex(id.data(), id.size());
	}
	ctx->h_own.assign(*prog);                         /* the last fallible step (host memory; all or nothing itself): the old scene is intact until here */
	/* commit (nothing below allocates on the way to the new scene being in place) */
	ctx->generation++;
	ctx->cur = next;
	ctx->have_prog = true;
	ctx->n_mops = n_mops;
	ctx->finite_scene = shadow_settle_ok(*prog);
	ctx->interp_sqrt_kind = interp_sqrt_kind;
	ctx->gamma_table = fast.gamma_ok;
	ctx->interp_key.swap(interp_key);
	resolve_skips(ctx);
	/* the scene compiler starts on its own thread; the new scene renders on the interpreter until its kernel is there
	 * (a failed specialisation is not an error either: the interpreter goes on rendering) */
	try { start_specialise(ctx, fast); }
	catch (...) { ctx->spec_state = 0; }              /* (no memory for a compiler run: the interpreter renders the scene) */
	return LOL_GPU_OK;
}

int lol_gpu_part_rows(int h, const lol_gpu_rows* rows) {
	if (h <= 0) return 0;
	if (!rows) return h;
	if (rows->band_rows <= 0 || rows->cycle_rows < rows->band_rows || rows->offset_rows < 0 ||
	    rows->offset_rows > rows->cycle_rows - rows->band_rows)
		return -1;
	long n = 0;
	for (long y0 = rows->offset_rows; y0 < h; y0 += rows->cycle_rows) {
		long y1 = y0 + rows->band_rows;
		if (y1 > h) y1 = h;
		n += y1 - y0;
	}
	return (int)n;
}

int lol_gpu_render_device(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps,
                          const lol_gpu_rows* rows, void* dst, size_t pitch_bytes,
                          const lol_gpu_debug* dbg, void* stream) {
	if (!ctx || !cam || !dst) return LOL_GPU_ERR_ARG;
	if (!ctx->have_prog) return fail(ctx, LOL_GPU_ERR_NO_PROGRAM, "no scene program uploaded");
	if (w <= 0 || h <= 0 || max_steps < 0 || pitch_bytes % 4 || pitch_bytes < (size_t)w * 4)
		return fail(ctx, LOL_GPU_ERR_ARG, "bad frame geometry");
	lol_gpu_rows whole = { h, h, 0 };
	const lol_gpu_rows* R = rows ? rows : &whole;
	int n_rows = lol_gpu_part_rows(h, R);
	if (n_rows < 0) return fail(ctx, LOL_GPU_ERR_ARG, "bad row partition");
	if (n_rows == 0) return LOL_GPU_OK;
	/* h need not be a multiple of cycle_rows: a part's band in the last, partial cycle is cut or absent, and it is the
	 * part's last, so every part's local rows stay dense (lol_gpu_part_frame_row is the mapping) */

	lol::Launch L;
	memset(&L, 0, sizeof L);
	memcpy(&L.cam, cam, sizeof L.cam);
	L.fw = (float)w; L.fh = (float)h;
	L.w = w; L.h = h; L.max_steps = max_steps;
	L.n_rows = n_rows;
	L.band_rows = R->band_rows; L.cycle_rows = R->cycle_rows; L.offset_rows = R->offset_rows;
	const lol_program& P = ctx->h_prog;
	L.n_ops = ctx->n_mops; L.n_lights = P.n_lights; L.n_materials = P.n_materials; L.n_roots = P.n_roots;
	L.ops           = ctx->d_mops[ctx->cur] + (ctx->finite_scene && camera_sane(*cam) ? (size_t)ctx->n_mops * lol::MOP_DWORDS : 0u);
	L.lights        = ctx->d_tables[ctx->cur];
	L.materials     = L.lights + (size_t)P.n_lights * lol::LIGHT_DWORDS;
	L.root_material = L.materials + (size_t)P.n_materials * lol::MATERIAL_DWORDS;
	L.ambient[0] = P.ambient_color.x; L.ambient[1] = P.ambient_color.y; L.ambient[2] = P.ambient_color.z;
	L.flags = (ctx->miss_skip ? lol::FLAG_MISS_SKIP : 0u) | (ctx->dark_skip ? lol::FLAG_DARK_SKIP : 0u) |
	          (ctx->shadow_settle && camera_sane(*cam) ? lol::FLAG_SHADOW_SETTLED : 0u);
	if (ctx->gamma_table) { L.flags |= lol::FLAG_GAMMA_TABLE; L.gamma_table = ctx->d_gamma; }
	if (first_step(ctx, *cam, max_steps)) { L.flags |= lol::FLAG_FIRST_STEP; L.first_dist = ctx->first_dist; L.first_id = ctx->first_id; }
	L.dst = static_cast<uint32_t*>(dst);
	L.pitch_px = (uint32_t)(pitch_bytes / 4);
	L.fmt_shift = ctx->fmt_shift; L.fmt_loss = ctx->fmt_loss; L.fmt_amask = ctx->fmt_amask;
	if (dbg) {
		L.dbg_rgb = dbg->rgb; L.dbg_hit_dist = dbg->hit_dist;
		L.dbg_hit_id = dbg->hit_id; L.dbg_steps = dbg->steps;
	}

	/* LOL_GPU_STREAM_DEFAULT == hipStreamLegacy; NULL = the context's own stream(s), in turn (lol_gpu_set_frames_in_flight) */
	hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->frame_streams[ctx->frame_rr++ % (unsigned)ctx->n_frame_streams];
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	finish_specialise(ctx, false);           /* the frame boundary at which a finished scene kernel takes over */
	const int tile_w = lol::TILE_W, tile_h = lol::TILE_H;      /* both kernels: one 16 x 4 wave per block (lol_kernel.h) */
	const int block = tile_w * tile_h;
	dim3 grid((w + tile_w - 1) / tile_w, (n_rows + tile_h - 1) / tile_h);
	const size_t common = (size_t)(lol::common_lds_dwords(P.n_lights, P.n_materials, P.n_roots) - lol::TILE_W * lol::TILE_H + block) * 4;
	int trial = -1;
	bool table = false;
	if (ctx->tiles.mode == LOL_GPU_TILES_LPT) {
		FrameTables F;
		if ((table = lpt_table_for_frame(ctx, cam, w, h, max_steps, R, n_rows, block, s, &F))) {
			L.flags |= lol::FLAG_TILE_TABLE;
			L.tile_order = F.order;
			L.tile_cost = F.cost;
			L.tile_stride = (F.n_waves + 7u) >> 3;
			L.lane_pixels = F.lanes;
			L.pixel_cost = F.pixel_cost;
			grid = dim3(F.n_waves, 1);
		}
	}
	if (!table) ctx->lpt_last_set = -1;
	/* (longest-first without a table — the camera moves, or another stream —: the better of the two fixed orders, like AUTO) */
	if (!table && tile_order_for_frame(ctx, w, h, max_steps, R, dbg != nullptr, &trial) == LOL_GPU_TILES_COLS) {
		L.flags |= lol::FLAG_TILE_COLS;
		const unsigned t = grid.x; grid.x = grid.y; grid.y = t;      /* (both stay far below the 65535 blocks a grid may have in y) */
	}
	hipError_t e;
	if (trial >= 0) LOL_HIP(ctx, hipEventRecord(ctx->tiles.ev[2 * trial], s));
	g_roctx.init();
	if (g_roctx.push) {
		char label[96];
		snprintf(label, sizeof label, "lol frame %dx%d rows=%d band=%d@%d/%d %s", w, h, n_rows, R->band_rows, R->offset_rows, R->cycle_rows, ctx->kernel_name);
		g_roctx.push(label);
		g_roctx.ranges++;
	}
	if (ctx->spec_fn) {
		void* args[] = { &L };
		/* the step counters are compiled into lol_render_spec_steps alone (generate_source): who reads them gets that kernel */
		const bool counts = (dbg && dbg->steps) || L.pixel_cost;
		e = hipModuleLaunchKernel(counts && ctx->spec_steps_fn ? ctx->spec_steps_fn : ctx->spec_fn, grid.x, grid.y, 1, block, 1, 1, (unsigned)common, s, args, nullptr);
	} else {
		const int kind = ctx->interp_sqrt_kind;
		const int cls = interp_stack_class(P.max_stack);
		if (lol::tables_in_lds(P.n_lights, P.n_materials, P.n_roots)) {
			if (cls == 1)      e = launch_interp<1>(L, grid, common, s, kind);
			else if (cls == 3) e = launch_interp<3>(L, grid, common, s, kind);
			else if (cls == 7) e = launch_interp<7>(L, grid, common, s, kind);
			else if (cls == lol::MOP_DEEP_FROM - 1) e = launch_interp<lol::MOP_DEEP_FROM - 1>(L, grid, common, s, kind);
			else               e = launch_interp<lol::MOP_DEEP_SLOTS>(L, grid, common, s, kind);
		} else {               /* large tables, read from global memory (lol_kernel.h, TABLES_LDS_MAX_DWORDS): three stack classes */
			if (cls <= 3)      e = launch_interp<3, true>(L, grid, common, s, kind);
			else if (cls <= lol::MOP_DEEP_FROM - 1) e = launch_interp<lol::MOP_DEEP_FROM - 1, true>(L, grid, common, s, kind);
			else               e = launch_interp<lol::MOP_DEEP_SLOTS, true>(L, grid, common, s, kind);
		}
	}
	if (g_roctx.pop) g_roctx.pop();
	if (e != hipSuccess) return fail(ctx, LOL_GPU_ERR_HIP, "kernel launch", e);
	if (trial >= 0) LOL_HIP(ctx, hipEventRecord(ctx->tiles.ev[2 * trial + 1], s));
	return LOL_GPU_OK;
}

int lol_gpu_set_pixel_format(lol_gpu* ctx, const lol_gpu_pixel_format* fmt) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	static const lol_gpu_pixel_format xrgb8888 = { 16, 8, 0, 0, 0, 0, 4, 0, 0 };
	const lol_gpu_pixel_format& f = fmt ? *fmt : xrgb8888;
	/* the reference stores a Uint32 per pixel whatever the format says (naive_renderer.c:233-235): 4-byte formats only,
	 * and SDL_MapRGB's shift formula only describes non-palettised ones */
	if (f.palettised) return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "palettised surfaces are not supported");
	if (f.bytes_per_pixel != 4) return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "only 32-bit surfaces are supported");
	if (f.r_shift > 31 || f.g_shift > 31 || f.b_shift > 31 || f.r_loss > 8 || f.g_loss > 8 || f.b_loss > 8)
		return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "pixel format shifts / losses out of range");
	ctx->fmt_shift = (uint32_t)f.r_shift | (uint32_t)f.g_shift << 8 | (uint32_t)f.b_shift << 16;
	ctx->fmt_loss = (uint32_t)f.r_loss | (uint32_t)f.g_loss << 8 | (uint32_t)f.b_loss << 16;
	ctx->fmt_amask = f.a_mask;
	return LOL_GPU_OK;
}

/* the context's second, third ... frame stream, created when first wanted */
static int ensure_frame_streams(lol_gpu* ctx, int n) {
	for (int i = 1; i < n && i < lol_gpu::MAX_FRAME_STREAMS; i++)
		if (!ctx->frame_streams[i]) LOL_HIP(ctx, hipStreamCreateWithFlags(&ctx->frame_streams[i], hipStreamNonBlocking));
	return LOL_GPU_OK;
}

static int ensure_copy_stream(lol_gpu* ctx) {
	if (ctx->copy_stream) return LOL_GPU_OK;
	LOL_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
	for (int i = 0; i < lol_gpu::PIPE_SLOTS; i++) {
		LOL_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe_rendered[i], hipEventDisableTiming));
		LOL_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe_copied[i], hipEventDisableTiming));
	}
	return LOL_GPU_OK;
}

/*
 * The surface is memory the HOST owns (SDL's window surface, main.c:182): nothing about it is remembered between
 * calls, and it is never registered with the device by this library — the HIP runtime pins the pages of a copy's
 * destination for the duration of that copy by itself and reaches PCIe line rate that way (56 GB/s into plain malloc'd
 * memory on MI355X, the same as into hipHostRegister'd memory: tools/d2h_bench.hip, profiles/r3_d2h_routes.jsonl).
 * So one frame costs kernel + copy here (C3: 1.09 + 0.59 + 0.1 ms); a host that can give the next camera early hides the
 * copy completely with lol_gpu_render_host_begin / _end below.  Two ways to hide it inside ONE call were built in round 3,
 * measured and removed (profiles/r3_host_surface_routes.md):
 *  - the surface registered once by address (hipHostRegister) and the kernel storing straight into it: +27 % per frame,
 *    but a host that unmaps and re-maps its surface at the same address behind the library's back — what SDL may do to a
 *    window surface on a resize — left the device writing into pages that were gone: the runtime aborted the process;
 *  - the frame as row chunks, chunk i copied while chunk i+1 renders: no gain — a copy into unregistered memory costs
 *    ~0.1 ms of pinning per call and did not run under the following kernels (4 chunks 3840 vs 3880 Mpixels/s, 8: 3020).
 */
int lol_gpu_render_host(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps,
                        void* host_pixels, size_t pitch_bytes) {
	if (!ctx || !host_pixels) return LOL_GPU_ERR_ARG;
	if (w <= 0 || h <= 0 || pitch_bytes < (size_t)w * 4) return fail(ctx, LOL_GPU_ERR_ARG, "bad frame geometry");
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	size_t need = (size_t)w * h * 4;
	if (need > ctx->frame_bytes) {          /* the surface may be resized between frames (main.c:182-187) */
		if (ctx->d_frame) { LOL_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_frame); }
		ctx->d_frame = nullptr; ctx->frame_bytes = 0;
		LOL_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_frame), need));
		ctx->frame_bytes = need;
	}
	int st = lol_gpu_render_device(ctx, cam, w, h, max_steps, nullptr, ctx->d_frame, (size_t)w * 4, nullptr, ctx->stream);
	if (st != LOL_GPU_OK) return st;
	LOL_HIP(ctx, hipMemcpy2DAsync(host_pixels, pitch_bytes, ctx->d_frame, (size_t)w * 4, (size_t)w * 4, h,
	                              hipMemcpyDeviceToHost, ctx->stream));
	LOL_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LOL_GPU_OK;
}

/*
 * The host-surface path with frames in flight: begin() queues frame i+1's kernel while end() copies frame i
 * into the host's surface, so the 33 MB device-to-host copy of a 4K frame (0.6 ms at PCIe Gen5 rates) runs under
 * the next frame's kernel instead of after its own.  Copies go to a stream of their own, one device framebuffer per
 * frame in flight.
 *
 * Round 5: and the KERNELS of consecutive frames go to different streams.  A frame is one launch that ends with its slowest
 * waves on half-empty SIMDs; the reference's loop cannot start frame i+1 before frame i has been shown
 * (main.c:189-194), but a host that has handed over the next camera already can: frame i+1's first waves fill frame i's
 * tail (tools/stream_overlap_ab.py: +9 % C3, +13 % the orbit, +67 % scene.lol at 1080p).  That is what helps a camera that
 * MOVES, whose frames cannot be scheduled by their predecessors' costs; a camera that stands still keeps its schedule as
 * well — one set of tables per stream (lpt_table_for_frame).  Frames in flight: two by default, up to PIPE_SLOTS after
 * lol_gpu_set_frames_in_flight(ctx, n).  What keeps a framebuffer safe is events, not stream order: a slot's kernel
 * waits for the copy that last read the slot, a slot's copy for the kernel that wrote it.
 */
int lol_gpu_render_host_begin(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps) {
	if (!ctx || !cam) return LOL_GPU_ERR_ARG;
	if (w <= 0 || h <= 0) return fail(ctx, LOL_GPU_ERR_ARG, "bad frame geometry");
	const unsigned depth = (unsigned)std::max(2, ctx->n_frame_streams);
	if (ctx->pipe_begun - ctx->pipe_ended >= depth)
		return fail(ctx, LOL_GPU_ERR_ARG, depth == 2 ? "two frames already in flight: call lol_gpu_render_host_end first"
		                                             : "every frame slot is in flight (lol_gpu_set_frames_in_flight): call lol_gpu_render_host_end first");
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	{
		int st = ensure_copy_stream(ctx);
		if (st == LOL_GPU_OK) st = ensure_frame_streams(ctx, (int)depth);
		if (st != LOL_GPU_OK) return st;
	}
	const int slot = (int)(ctx->pipe_begun % lol_gpu::PIPE_SLOTS);
	/* Which stream.  A frame whose view is NEW goes to the next stream of the rotation: it runs in a fixed tile order, its launch
	 * has a long tail, and the frame after it fills that tail (the orbit through the C host: 9170 -> 11,000 Mpixels/s).  A frame
	 * under the SAME view as the frame before it follows that frame on its stream: such frames are scheduled by their
	 * predecessor's costs and have no tail to fill, and two of them side by side finish together — after which nothing is
	 * queued while the host waits in _end() for the first one's copy (measured: 1.03 ms per frame on two streams against 0.83
	 * on one, profiles/r5_frames_in_flight.md). */
	const int geom[3] = { w, h, max_steps };
	const bool same_view = ctx->pipe_last_stream && memcmp(geom, ctx->pipe_last_geom, sizeof geom) == 0 && memcmp(cam, &ctx->pipe_last_cam, sizeof *cam) == 0;
	hipStream_t ks = same_view ? ctx->pipe_last_stream : ctx->frame_streams[ctx->pipe_rr++ % depth];
	const size_t need = (size_t)w * h * 4;
	if (need > ctx->pipe_bytes[slot]) {
		/* the surface grew (main.c:182-187).  This slot's last frame was ended PIPE_SLOTS calls ago; its copy may still run, and
		 * so may the kernel of a frame that was discarded: wait for both, then replace this slot's framebuffer only — the
		 * frames queued in the OTHER slots stay valid */
		LOL_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
		if (ctx->pipe_stream[slot]) LOL_HIP(ctx, hipStreamSynchronize(ctx->pipe_stream[slot]));
		if (ctx->d_pipe[slot]) (void)hipFree(ctx->d_pipe[slot]);
		ctx->d_pipe[slot] = nullptr;
		ctx->pipe_bytes[slot] = 0;
		LOL_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_pipe[slot]), need));
		ctx->pipe_bytes[slot] = need;
	}
	/* the copy that last read this framebuffer must be done before the kernel overwrites it — and so must the kernel that
	 * last wrote it, where that one ran on another stream (a frame that was discarded; a changed number of streams) */
	LOL_HIP(ctx, hipStreamWaitEvent(ks, ctx->pipe_copied[slot], 0));
	if (ctx->pipe_stream[slot] && ctx->pipe_stream[slot] != ks) LOL_HIP(ctx, hipStreamWaitEvent(ks, ctx->pipe_rendered[slot], 0));
	int st = lol_gpu_render_device(ctx, cam, w, h, max_steps, nullptr, ctx->d_pipe[slot], (size_t)w * 4, nullptr, ks);
	if (st != LOL_GPU_OK) return st;
	LOL_HIP(ctx, hipEventRecord(ctx->pipe_rendered[slot], ks));
	ctx->pipe_stream[slot] = ks;
	ctx->pipe_last_stream = ks;
	ctx->pipe_last_cam = *cam;
	memcpy(ctx->pipe_last_geom, geom, sizeof geom);
	ctx->pipe_w[slot] = w; ctx->pipe_h[slot] = h;
	ctx->pipe_begun++;
	return LOL_GPU_OK;
}

int lol_gpu_render_host_end(lol_gpu* ctx, void* host_pixels, size_t pitch_bytes, int w, int h) {
	if (!ctx || !host_pixels) return LOL_GPU_ERR_ARG;
	if (ctx->pipe_begun == ctx->pipe_ended) return fail(ctx, LOL_GPU_ERR_ARG, "no frame in flight");
	const int slot = (int)(ctx->pipe_ended % lol_gpu::PIPE_SLOTS);
	/* the surface's size, as the caller sees it NOW, decides what may be written: a frame queued before a resize
	 * is never copied into a surface of another size (it stays queued: discard it, or end it into a fitting one) */
	if (w != ctx->pipe_w[slot] || h != ctx->pipe_h[slot])
		return fail(ctx, LOL_GPU_ERR_ARG, "the queued frame's size differs from the surface's: lol_gpu_render_host_discard");
	if (pitch_bytes < (size_t)w * 4) return fail(ctx, LOL_GPU_ERR_ARG, "bad frame geometry");
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	LOL_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->pipe_rendered[slot], 0));
	LOL_HIP(ctx, hipMemcpy2DAsync(host_pixels, pitch_bytes, ctx->d_pipe[slot], (size_t)w * 4, (size_t)w * 4, h,
	                              hipMemcpyDeviceToHost, ctx->copy_stream));
	LOL_HIP(ctx, hipEventRecord(ctx->pipe_copied[slot], ctx->copy_stream));
	ctx->pipe_ended++;
	LOL_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
	return LOL_GPU_OK;
}

int lol_gpu_render_host_pending(const lol_gpu* ctx) { return ctx ? (int)(ctx->pipe_begun - ctx->pipe_ended) : 0; }

int lol_gpu_render_host_pending_size(const lol_gpu* ctx, int* w, int* h) {
	if (!ctx || !w || !h) return LOL_GPU_ERR_ARG;
	const bool any = ctx->pipe_begun != ctx->pipe_ended;
	const int slot = (int)(ctx->pipe_ended % lol_gpu::PIPE_SLOTS);
	*w = any ? ctx->pipe_w[slot] : 0;
	*h = any ? ctx->pipe_h[slot] : 0;
	return LOL_GPU_OK;
}

int lol_gpu_render_host_discard(lol_gpu* ctx) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	/* the kernels still run to completion into their own framebuffers; nothing of them reaches a surface */
	ctx->pipe_ended = ctx->pipe_begun;
	return LOL_GPU_OK;
}

/* every stream of the context's own: frames in flight may be on any of them */
static int sync_own_streams(lol_gpu* ctx) {
	for (hipStream_t fs : ctx->frame_streams) if (fs) LOL_HIP(ctx, hipStreamSynchronize(fs));
	return LOL_GPU_OK;
}

int lol_gpu_sync(lol_gpu* ctx) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	return sync_own_streams(ctx);
}

int lol_gpu_set_frames_in_flight(lol_gpu* ctx, int n) {
	if (!ctx || n < 1 || n > lol_gpu::MAX_FRAME_STREAMS) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	{
		const int st = ensure_frame_streams(ctx, n);
		if (st != LOL_GPU_OK) return st;
	}
	/* a frame queued on a stream that is about to fall out of the rotation stays ordered before whatever comes next */
	{
		const int st = sync_own_streams(ctx);
		if (st != LOL_GPU_OK) return st;
	}
	ctx->n_frame_streams = n;
	ctx->frame_rr = 0;
	return LOL_GPU_OK;
}

int lol_gpu_frames_in_flight(const lol_gpu* ctx) { return ctx ? ctx->n_frame_streams : 0; }

void* lol_gpu_next_stream(lol_gpu* ctx) {
	return ctx ? static_cast<void*>(ctx->frame_streams[ctx->frame_rr % (unsigned)ctx->n_frame_streams]) : nullptr;
}

int lol_gpu_malloc(lol_gpu* ctx, size_t bytes, void** out) {
	if (!ctx || !out) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	LOL_HIP(ctx, hipMalloc(out, bytes));
	return LOL_GPU_OK;
}

int lol_gpu_free(lol_gpu* ctx, void* ptr) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	LOL_HIP(ctx, hipFree(ptr));
	return LOL_GPU_OK;
}

int lol_gpu_memcpy_d2h(lol_gpu* ctx, void* host, const void* dev, size_t bytes) {
	if (!ctx || !host || !dev) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	{
		const int st = sync_own_streams(ctx);
		if (st != LOL_GPU_OK) return st;
	}
	LOL_HIP(ctx, hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
	return LOL_GPU_OK;
}

const char* lol_gpu_kernel_name(const lol_gpu* ctx) { return ctx ? ctx->kernel_name : ""; }

long lol_gpu_roctx_ranges(void) {
	g_roctx.init();
	return g_roctx.asked && !g_roctx.push ? -1 : g_roctx.ranges.load();
}

const char* lol_gpu_kernel_key(const lol_gpu* ctx) {
	if (!ctx) return "";
	return ctx->spec_fn ? ctx->spec_key.c_str() : ctx->interp_key.c_str();
}

int lol_gpu_abi_version(void) { return LOL_GPU_ABI_VERSION; }

int lol_gpu_testing_has_return_clobbering_branch(const void* code, size_t n_bytes) {
	if (!code) return LOL_GPU_ERR_ARG;
	return has_return_clobbering_branch(code, n_bytes) ? 1 : 0;
}

int lol_gpu_testing_fail_first_tier(lol_gpu* ctx, int n) {
	if (!ctx || n < 0) return LOL_GPU_ERR_ARG;
	ctx->fail_first_tier = n;
	return LOL_GPU_OK;
}

int lol_gpu_testing_fail_uploads(lol_gpu* ctx, int n) {
	if (!ctx || n < 0) return LOL_GPU_ERR_ARG;
	ctx->fail_uploads = n;
	return LOL_GPU_OK;
}

int lol_gpu_powf_batch(lol_gpu* ctx, const float* x_dev, const float* y_dev, float* out_dev, size_t n, void* stream) {
	if (!ctx || !x_dev || !y_dev || !out_dev) return LOL_GPU_ERR_ARG;
	if (n == 0) return LOL_GPU_OK;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
	hipLaunchKernelGGL(powf_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x_dev, y_dev, out_dev, n);
	LOL_HIP(ctx, hipGetLastError());
	return LOL_GPU_OK;
}

int lol_gpu_sdf_batch(lol_gpu* ctx, const float* pts_dev, float* dist_dev, uint32_t* id_dev, size_t n, void* stream) {
	if (!ctx || !pts_dev || !dist_dev || !id_dev || n > 0xFFFFFFFFu) return LOL_GPU_ERR_ARG;
	if (!ctx->have_prog) return fail(ctx, LOL_GPU_ERR_NO_PROGRAM, "no scene program uploaded");
	if (n == 0) return LOL_GPU_OK;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
	uint32_t n32 = (uint32_t)n;
	hipError_t e;
	finish_specialise(ctx, false);
	if (ctx->spec_fn && ctx->spec_sdf_fn) {
		void* args[] = { &pts_dev, &dist_dev, &id_dev, &n32 };
		e = hipModuleLaunchKernel(ctx->spec_sdf_fn, (n32 + 63) / 64, 1, 1, 64, 1, 1, 0, s, args, nullptr);
	} else {
		const int kind = ctx->interp_sqrt_kind;
		const int cls = interp_stack_class(ctx->h_prog.max_stack);
		if (cls == 1)      e = launch_sdf_interp<1>(ctx->d_mops[ctx->cur], ctx->n_mops, pts_dev, dist_dev, id_dev, n32, s, kind);
		else if (cls == 3) e = launch_sdf_interp<3>(ctx->d_mops[ctx->cur], ctx->n_mops, pts_dev, dist_dev, id_dev, n32, s, kind);
		else if (cls == 7) e = launch_sdf_interp<7>(ctx->d_mops[ctx->cur], ctx->n_mops, pts_dev, dist_dev, id_dev, n32, s, kind);
		else if (cls == lol::MOP_DEEP_FROM - 1) e = launch_sdf_interp<lol::MOP_DEEP_FROM - 1>(ctx->d_mops[ctx->cur], ctx->n_mops, pts_dev, dist_dev, id_dev, n32, s, kind);
		else               e = launch_sdf_interp<lol::MOP_DEEP_SLOTS>(ctx->d_mops[ctx->cur], ctx->n_mops, pts_dev, dist_dev, id_dev, n32, s, kind);
	}
	if (e != hipSuccess) return fail(ctx, LOL_GPU_ERR_HIP, "sdf kernel launch", e);
	return LOL_GPU_OK;
}

/* Offline use (tests, ISA inspection; needs no device): compile the scene-specialised kernel for
 * `arch` and write `<out_base>.hip` (generated source) and `<out_base>.co` (code object). */
int lol_gpu_compile_offline(const lol_program* prog, const char* arch, const char* out_base, int assume_fast,
                            char* log, size_t logcap) {
	if (!prog || !arch) return LOL_GPU_ERR_ARG;
	std::vector<char> code;
	std::string lg, src;
	FastPaths fast;
	if (assume_fast) {                 /* ISA inspection only: pretend every shortcut was proven */
		fast.sqrt_kind = assume_fast >= 1 && assume_fast <= 3 ? 4 - assume_fast : 3;   /* 1 → sqrt_r2, 2 → sqrt_gs, 3 → sqrt_pm */
		fast.sqrt_tiny_ok = true;
		for (uint32_t i = 0; i < prog->n_ops; i++)
			if ((prog->ops[i].op == LOL_OP_SMIN || prog->ops[i].op == LOL_OP_SMIN_R) && !fast.has(prog->ops[i].f[0]))
				{ fast.div_ok.push_back(prog->ops[i].f[0]); fast.div_nf_ok.push_back(prog->ops[i].f[0]); }
	}
	bool ok = false;
	{
		/* on the large-stack thread, like every run of the scene compiler (BigStackThread) */
		BigStackThread th;
		auto work = [&]() {
			try { std::lock_guard<std::mutex> rtc(g_rtc_mutex); ok = compile_spec(*prog, &fast, arch, code, lg, &src, culling_enabled(1)); }
			catch (...) { ok = false; lg = "the scene compiler ran out of memory"; }
		};
		bool started = false;
		try { started = th.start(work); } catch (...) { started = false; }
		if (started) th.join(); else work();
	}
	if (log && logcap) snprintf(log, logcap, "%s", lg.c_str());
	if (out_base && out_base[0]) {
		std::string base = out_base;
		if (FILE* f = fopen((base + ".hip").c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
		if (ok) if (FILE* f = fopen((base + ".co").c_str(), "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
	}
	return ok ? LOL_GPU_OK : LOL_GPU_ERR_UNSUPPORTED;
}

}  // extern "C"
