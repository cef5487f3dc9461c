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
#include "lol_gpu_internal.h"

/* LOL_BUILD_ID: a digest of this library's sources and compiler flags (csrc/Makefile: lol_build_id.inc) — the identity
 * of the ahead-of-time kernels (lol_gpu_kernel_key) */
#include "lol_build_id.inc"

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

extern "C" const char* lol_gpu_tuning_switches(void) {
	try {
		std::lock_guard<std::mutex> lock(g_tuning_mutex);
		g_tuning_text.clear();
		for (const auto& e : g_tuning_seen) { if (!g_tuning_text.empty()) g_tuning_text += ' '; g_tuning_text += e.first + "=" + e.second; }
		return g_tuning_text.c_str();
	} catch (...) { return "(out of memory)"; }
}

#pragma GCC visibility push(hidden)
int fail(lol_gpu* ctx, int status, const char* what, hipError_t e) {
	if (ctx) {
		if (e != hipSuccess) snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, hipGetErrorString(e));
		else snprintf(ctx->err, sizeof ctx->err, "%s", what);
	}
	return status;
}
std::mutex g_rtc_mutex;
#pragma GCC visibility pop

namespace {

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
	ctx->second_tier_pending = ctx->want_second_tier && !tuning_env("LOL_GPU_SPEC_INLINE_MAX") && ctx->h_prog.n_ops > LOL_SPEC_FIRST_TIER_INLINE_MAX_OPS &&
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
	/* 0 interpreter, plain | 1 specialised + proven fast paths (default) | 3 specialised, plain | 4 interpreter + fast paths |
	 * 5 = 1 without the second tier of a mid-size scene (the out-of-line kernel stays: nothing compiles behind it) */
	if (enable < 0 || enable == 2 || enable > 5) return LOL_GPU_ERR_ARG;
	ctx->want_spec = (enable == 1 || enable == 3 || enable == 5) ? 1 : 0;
	ctx->want_fast = (enable == 1 || enable == 4 || enable == 5) ? 1 : 0;
	ctx->want_second_tier = enable != 5;
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
int lol_gpu_set_cull(lol_gpu* ctx, int enable) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	ctx->want_cull = enable ? 1 : 0;          /* takes effect at the next lol_gpu_upload_program */
	return LOL_GPU_OK;
}

/* bit 0: escaped-wave skip active; bit 1: zero-incidence shadow skip active */
int lol_gpu_miss_skip_active(const lol_gpu* ctx) {
	return ctx ? (ctx->miss_skip ? 1 : 0) | (ctx->dark_skip ? 2 : 0) | (ctx->shadow_settle ? 4 : 0) : 0;
}
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
	/* Two lists of the same records: the second one takes the blend factors without v_div_fixup where the device proved
	 * them.  That proof covers every FINITE difference of operands; the launch picks the second list only when nothing an
	 * evaluation computes can be infinite (finite_scene, and a sane camera for that frame), the SDF of arbitrary points
	 * (lol_gpu_sdf_batch) never does.  (The specialised kernel looks at NaN per object instead; here that costs more than
	 * the fixup saves.) */
	std::vector<uint32_t> mops;
	uint32_t n_mops = 0;
	if (!build_interp_lists(*prog, fast, culling_enabled(ctx->want_cull), mops, n_mops))
		return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "interpreter lists differ in length");      /* (same records by construction) */
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
	if (table) lpt_frame_queued(ctx, s);
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
 * waves on half-empty SIMDs (LABNOTES.md §3.9); the reference's loop cannot start frame i+1 before frame i has been shown
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
}  // extern "C"
