/*
 * lol_gpu.hip — C ABI (include/lol_gpu.h) over the gfx950 render kernel (lol_kernel.h).
 *
 * Host side of the drop-in: context = {device, stream, device copy of the
 * flattened scene, a device framebuffer for the host-surface path}.  No CPU
 * rendering path exists here; without a HIP device every call fails.
 */
#include "lol_gpu.h"
#include "lol_kernel.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

struct lol_gpu {
	int          device = -1;
	hipStream_t  stream = nullptr;
	lol_program* d_prog = nullptr;       /* device */
	lol_program  h_prog;                 /* host mirror (counts, max_stack) */
	bool         have_prog = false;
	uint32_t*    d_frame = nullptr;      /* framebuffer for lol_gpu_render_host */
	size_t       frame_bytes = 0;
	char         err[256] = { 0 };
};

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

template <int STACK>
hipError_t launch(const lol::Launch& L, dim3 grid, size_t lds, hipStream_t s) {
	hipLaunchKernelGGL(lol::render_kernel<STACK>, grid, dim3(lol::BLOCK), lds, s, L);
	return hipGetLastError();
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
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_prog), sizeof(lol_program));
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
	if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
	if (ctx->d_prog) (void)hipFree(ctx->d_prog);
	if (ctx->d_frame) (void)hipFree(ctx->d_frame);
	delete ctx;
}

const char* lol_gpu_error(const lol_gpu* ctx) { return ctx ? ctx->err : "null context"; }

int lol_gpu_upload_program(lol_gpu* ctx, const lol_program* prog) {
	if (!ctx || !prog) return LOL_GPU_ERR_ARG;
	if (prog->n_ops > LOL_MAX_OPS || prog->n_lights > LOL_MAX_LIGHTS || prog->n_materials > LOL_MAX_MATERIALS ||
	    prog->n_roots > LOL_MAX_OPS || prog->max_stack > LOL_MAX_STACK || prog->n_materials == 0)
		return fail(ctx, LOL_GPU_ERR_UNSUPPORTED, "program exceeds interpreter limits");
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
	/* ordered after frames already queued on the context stream */
	LOL_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LOL_HIP(ctx, hipMemcpy(ctx->d_prog, prog, sizeof *prog, hipMemcpyHostToDevice));
	ctx->h_prog = *prog;
	ctx->have_prog = true;
	return LOL_GPU_OK;
}

int lol_gpu_part_rows(int h, const lol_gpu_rows* rows) {
	if (h <= 0) return 0;
	if (!rows) return h;
	if (rows->band_rows <= 0 || rows->n_parts <= 0 || rows->part < 0 || rows->part >= rows->n_parts) return -1;
	long bands = ((long)h + rows->band_rows - 1) / rows->band_rows;
	long n = 0;
	for (long b = rows->part; b < bands; b += rows->n_parts) {
		long y0 = b * rows->band_rows, y1 = y0 + rows->band_rows;
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
	lol_gpu_rows whole = { h, 1, 0 };
	const lol_gpu_rows* R = rows ? rows : &whole;
	int n_rows = lol_gpu_part_rows(h, R);
	if (n_rows < 0) return fail(ctx, LOL_GPU_ERR_ARG, "bad row partition");
	if (n_rows == 0) return LOL_GPU_OK;
	/* a partial last band is only laid out compactly when it is the part's last one */
	if (R->n_parts > 1 && h % R->band_rows != 0)
		return fail(ctx, LOL_GPU_ERR_ARG, "h must be a multiple of band_rows when n_parts > 1");

	lol::Launch L;
	memset(&L, 0, sizeof L);
	L.cam = *cam;
	L.fw = (float)w; L.fh = (float)h;
	L.w = w; L.h = h; L.max_steps = max_steps;
	L.n_rows = n_rows;
	L.band_rows = R->band_rows; L.n_parts = R->n_parts; L.part = R->part;
	L.n_ops = ctx->h_prog.n_ops; L.n_lights = ctx->h_prog.n_lights;
	L.n_materials = ctx->h_prog.n_materials; L.n_roots = ctx->h_prog.n_roots;
	L.prog = ctx->d_prog;
	L.dst = static_cast<uint32_t*>(dst);
	L.pitch_px = (uint32_t)(pitch_bytes / 4);
	if (dbg) {
		L.dbg_rgb = dbg->rgb; L.dbg_hit_dist = dbg->hit_dist;
		L.dbg_hit_id = dbg->hit_id; L.dbg_steps = dbg->steps;
	}

	dim3 grid((w + lol::TILE_W - 1) / lol::TILE_W, (n_rows + lol::TILE_H - 1) / lol::TILE_H);
	size_t lds = lol::lds_bytes(L);
	hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	hipError_t e;
	uint32_t need = ctx->h_prog.max_stack;
	if (need <= 2)      e = launch<2>(L, grid, lds, s);
	else if (need <= 4) e = launch<4>(L, grid, lds, s);
	else                e = launch<LOL_MAX_STACK>(L, grid, lds, s);
	if (e != hipSuccess) return fail(ctx, LOL_GPU_ERR_HIP, "kernel launch", e);
	return LOL_GPU_OK;
}

int lol_gpu_render_host(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps,
                        void* host_pixels, size_t pitch_bytes) {
	if (!ctx || !host_pixels) return LOL_GPU_ERR_ARG;
	if (w <= 0 || h <= 0 || pitch_bytes < (size_t)w * 4) return fail(ctx, LOL_GPU_ERR_ARG, "bad frame geometry");
	size_t need = (size_t)w * h * 4;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	if (need > ctx->frame_bytes) {          /* the surface may be resized between frames (main.c:182-187) */
		if (ctx->d_frame) { LOL_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_frame); }
		ctx->d_frame = nullptr; ctx->frame_bytes = 0;
		LOL_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_frame), need));
		ctx->frame_bytes = need;
	}
	int st = lol_gpu_render_device(ctx, cam, w, h, max_steps, nullptr, ctx->d_frame, (size_t)w * 4, nullptr, nullptr);
	if (st != LOL_GPU_OK) return st;
	LOL_HIP(ctx, hipMemcpy2DAsync(host_pixels, pitch_bytes, ctx->d_frame, (size_t)w * 4, (size_t)w * 4, h,
	                              hipMemcpyDeviceToHost, ctx->stream));
	LOL_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LOL_GPU_OK;
}

int lol_gpu_sync(lol_gpu* ctx) {
	if (!ctx) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	LOL_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LOL_GPU_OK;
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
	LOL_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LOL_HIP(ctx, hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
	return LOL_GPU_OK;
}

const char* lol_gpu_kernel_name(const lol_gpu* ctx) {
	(void)ctx;
	return "render_kernel";
}

}  // extern "C"
