/*
 * lol_proofs.hip — the exhaustive proofs behind the "fast exact paths" of lol_kernel.h, run on the device that will use them.
 * (Part of liblol_gpu.so; see lol_gpu_internal.h for how the library is cut.)
 */
#include "lol_gpu_internal.h"

#pragma GCC visibility push(hidden)
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

#pragma GCC visibility pop

extern "C" {

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

int lol_gpu_powf_batch(lol_gpu* ctx, const float* x_dev, const float* y_dev, float* out_dev, size_t n, void* stream) {
	if (!ctx || !x_dev || !y_dev || !out_dev) return LOL_GPU_ERR_ARG;
	if (n == 0) return LOL_GPU_OK;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
	hipLaunchKernelGGL(powf_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x_dev, y_dev, out_dev, n);
	LOL_HIP(ctx, hipGetLastError());
	return LOL_GPU_OK;
}

}  // extern "C"
