/*
 * hip_renderer.c — loltracer renderer plug-in backed by the MI355X kernel.
 *
 * Implements the three entry points of the reference's renderer.h:24-26 with
 * the same protocol as naive_renderer.c:195-247, so it links in its place
 * (`make hip`):
 *
 *   render_prepare  once, main thread, workers already parked on the entry
 *                   semaphore (main.c:147-161): create the GPU context,
 *                   flatten the scene, upload it.  Renderer flags start at
 *                   argv[3] (main.c:223-242): --device N (one GPU),
 *                   --pipeline (one GPU: the copy of frame i into the surface runs
 *                   under frame i+1's kernel, and frame i+1's first waves fill the tail
 *                   of frame i's launch: consecutive kernels go to different streams;
 *                   the surface then shows frame i-1; the first frame, and the first one
 *                   after the window changed size, are delivered at once),
 *                   --pipeline-depth N (2 = --pipeline; 3, 4: the surface shows frame
 *                   i-2, i-3 and that many kernels overlap),
 *                   --devices A,B,... (the frame's rows are dealt in bands over
 *                   these GPUs, each writing its bands into the surface,
 *                   include/lol_gpu.h lol_gpu_multi_*), --parts-per-device N,
 *                   --root-band-rows N (the first device's smaller share),
 *                   --max-steps N, --tile-columns / --tile-rows (pin the order in
 *                   which a launch hands out its tiles, lol_gpu_set_tile_order;
 *                   without either the library's default applies: a frame under the
 *                   camera of the frame before it is scheduled by what that frame
 *                   cost, any other frame runs in the better of the two fixed orders),
 *                   --report (one line on stdout at render_destroy: kernel, its key,
 *                   tile order mode / order in use / sorts so far),
 *                   --wait-kernel (render_prepare returns only when the scene's own
 *                   kernel is in place, as the tracing JIT's does — the LAST one, for a
 *                   scene of 257 ... 1024 ops, which gets two; without it the first
 *                   frames render on the interpreter kernel: benchmarks), and
 *                   --dump-kernel BASE, the counterpart of the JIT renderer's
 *                   -j/--jitdump (tracing_jit_renderer.dasc:424-433): writes the
 *                   scene-specialised kernel as BASE.hip (generated source) and
 *                   BASE.co (gfx950 code object, for llvm-objdump / rocprofv3).
 *   render_thread   every worker: wait entry → return 0 if exiting → work →
 *                   post exit exactly once (naive_renderer.c:203-205,238).
 *                   The reference's workers claim one row at a time with
 *                   SDL_AtomicAdd(&current_line, 1); here the worker whose
 *                   SDL_AtomicAdd(&current_line, height) returns 0 has claimed
 *                   every row: it launches the frame on the GPU and copies it
 *                   into surf->pixels honouring pitch; the others see
 *                   current_line >= height and fall through.
 *   render_destroy  once, after the workers were joined (main.c:166-171,213).
 *
 * Errors follow the reference's style: message on stderr, keep going
 * (naive_renderer.c:26); there is no CPU rendering fallback — a failed frame
 * leaves the surface untouched.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hip_renderer_host.h"
#include "lol_gpu.h"
#include "lol_scene.h"

struct hip_renderer {
	lol_gpu*       gpu;        /* --device N */
	lol_gpu_multi* multi;      /* --devices A,B,...: used instead of `gpu` */
	lol_program    program;
	int         max_steps;      /* MAX_STEPS, naive_renderer.c:49 */
	int         pipeline;       /* --pipeline: frame i's copy into the surface overlaps frame i+1's kernel (one frame of latency);
	                             * --pipeline-depth N: N frames queued before the oldest is delivered (N - 1 frames of latency) */
	int         shown_w, shown_h;/* --pipeline: size of the surface that last received a frame */
	int         ready;
	int         have_format;    /* the surface's pixel format has been handed to the library ... */
	lol_gpu_pixel_format format;/* ... and was this */
	int         format_refused; /* ... or was refused (reported once) */
	int         report;         /* --report: one line about what the library did, at render_destroy */
};

/* surf->format → the library (colorf_to_pixfmt maps through the SURFACE's format, renderer.h:17-22; the host re-fetches
 * the surface every frame, main.c:182).  Returns 0 when this format cannot be rendered. */
static int sync_format(struct hip_renderer* r, host_surface* surf) {
	lol_gpu_pixel_format f;
	memset(&f, 0, sizeof f);
	HOST_SURF_FORMAT(surf, &f);
	if (r->have_format && !memcmp(&f, &r->format, sizeof f)) return !r->format_refused;
	int st = r->multi ? lol_gpu_multi_set_pixel_format(r->multi, &f) : lol_gpu_set_pixel_format(r->gpu, &f);
	r->have_format = 1;
	r->format = f;
	r->format_refused = st != LOL_GPU_OK;
	if (st != LOL_GPU_OK)
		fprintf(stderr, "hip_renderer: %s\n", r->multi ? lol_gpu_multi_error(r->multi) : lol_gpu_error(r->gpu));
	return st == LOL_GPU_OK;
}

void render_prepare(struct render_data* data, int argc, const char* argv[]) {
	struct hip_renderer* r = calloc(1, sizeof *r);
	int device = 0;
	int devices[LOL_GPU_MULTI_MAX_DEVICES], n_devices = 0;
	int parts_per_device = 0, root_band = -1, tile_order = -1 /* the library's default: LOL_GPU_TILES_LPT */, wait_kernel = 0;
	const char* dump = NULL;
	HOST_PRIVATE(data) = r;
	if (!r) { fprintf(stderr, "hip_renderer: out of memory\n"); return; }
	/* a liblol_gpu.so of another interface version would take this file's structures for something else */
	if (lol_gpu_abi_version() != LOL_GPU_ABI_VERSION) {
		fprintf(stderr, "hip_renderer: built for lol_gpu ABI %d, the loaded library speaks %d: not rendering\n",
		        LOL_GPU_ABI_VERSION, lol_gpu_abi_version());
		return;
	}
	r->max_steps = 256;
	for (int i = 3; i < argc; i++) {
		const int is_device = !strcmp(argv[i], "--device"), is_devices = !strcmp(argv[i], "--devices");
		const int is_steps = !strcmp(argv[i], "--max-steps"), is_dump = !strcmp(argv[i], "--dump-kernel");
		const int is_ppd = !strcmp(argv[i], "--parts-per-device"), is_root = !strcmp(argv[i], "--root-band-rows");
		if (!strcmp(argv[i], "--pipeline")) { if (r->pipeline < 2) r->pipeline = 2; continue; }
		if (!strcmp(argv[i], "--pipeline-depth") && i + 1 < argc) {
			const int d = atoi(argv[++i]);
			if (d >= 2 && d <= 4) r->pipeline = d; else fprintf(stderr, "hip_renderer: --pipeline-depth wants 2, 3 or 4\n");
			continue;
		}
		if (!strcmp(argv[i], "--tile-columns")) { tile_order = LOL_GPU_TILES_COLS; continue; }
		if (!strcmp(argv[i], "--tile-rows")) { tile_order = LOL_GPU_TILES_ROWS; continue; }
		if (!strcmp(argv[i], "--report")) { r->report = 1; continue; }
		if (!strcmp(argv[i], "--wait-kernel")) { wait_kernel = 1; continue; }      /* render_prepare returns with the scene's own kernel in place (benchmarks) */
		if (!(is_device || is_devices || is_steps || is_dump || is_ppd || is_root)) continue;      /* the host's own flags */
		if (i + 1 >= argc) { fprintf(stderr, "hip_renderer: %s needs a value, ignored\n", argv[i]); break; }
		const char* v = argv[++i];
		if (is_device) device = atoi(v);
		else if (is_steps) r->max_steps = atoi(v);
		else if (is_dump) dump = v;
		else if (is_ppd) parts_per_device = atoi(v);
		else if (is_root) root_band = atoi(v);
		else {
			n_devices = 0;
			for (const char* p = v; *p && n_devices < LOL_GPU_MULTI_MAX_DEVICES;) {
				char* end;
				long d = strtol(p, &end, 10);
				if (end == p) break;
				devices[n_devices++] = (int)d;
				p = *end == ',' ? end + 1 : end;
				if (*end != ',' ) break;
			}
			if (n_devices == 0) fprintf(stderr, "hip_renderer: --devices wants a list like 0,1,2,3\n");
		}
	}

	const lol_scene* scene = HOST_SCENE_TO_LOL(data->scene);
	int st = scene ? lol_scene_flatten(scene, &r->program) : LOL_ERR_NOMEM;
#ifdef LOL_HOST_SDL
	lol_scene_free((lol_scene*)scene);      /* the converted copy; the program holds everything */
#endif
	if (st != LOL_OK) { fprintf(stderr, "hip_renderer: cannot flatten scene: %s\n", lol_status_str(st)); return; }

	if (dump) {
		char log[1024];
		if (lol_gpu_compile_offline(&r->program, "gfx950", dump, 0, log, sizeof log) != LOL_GPU_OK)
			fprintf(stderr, "hip_renderer: --dump-kernel failed: %s\n", log);
	}

	if (n_devices > 0) {
		st = lol_gpu_multi_create(devices, n_devices, &r->multi);
		if (st != LOL_GPU_OK) { fprintf(stderr, "hip_renderer: cannot set up %d device(s) (status %d)\n", n_devices, st); return; }
		if (parts_per_device > 0 && lol_gpu_multi_set_parts_per_device(r->multi, parts_per_device) != LOL_GPU_OK)
			fprintf(stderr, "hip_renderer: --parts-per-device %d refused\n", parts_per_device);
		if (root_band >= 0 && lol_gpu_multi_set_root_band_rows(r->multi, root_band) != LOL_GPU_OK)
			fprintf(stderr, "hip_renderer: --root-band-rows %d refused\n", root_band);
		if (tile_order >= 0) (void)lol_gpu_multi_set_tile_order(r->multi, tile_order);
		st = lol_gpu_multi_upload_program(r->multi, &r->program);
		if (st != LOL_GPU_OK) { fprintf(stderr, "hip_renderer: %s\n", lol_gpu_multi_error(r->multi)); return; }
		if (wait_kernel) (void)lol_gpu_multi_specialize_wait(r->multi);
	} else {
		st = lol_gpu_create(device, &r->gpu);
		if (st != LOL_GPU_OK) { fprintf(stderr, "hip_renderer: no usable HIP device %d (status %d)\n", device, st); return; }
		st = lol_gpu_upload_program(r->gpu, &r->program);
		if (st != LOL_GPU_OK) { fprintf(stderr, "hip_renderer: %s\n", lol_gpu_error(r->gpu)); return; }
		if (tile_order >= 0) (void)lol_gpu_set_tile_order(r->gpu, tile_order);
		if (r->pipeline > 2 && lol_gpu_set_frames_in_flight(r->gpu, r->pipeline) != LOL_GPU_OK) {
			fprintf(stderr, "hip_renderer: %s; two frames in flight\n", lol_gpu_error(r->gpu));
			r->pipeline = 2;
		}
		if (wait_kernel) (void)lol_gpu_specialize_wait(r->gpu);
	}
	r->ready = 1;
}

int render_thread(void* ptr) {
	struct render_data* data = ptr;

	for (;;) {
		HOST_SEM_WAIT(frame_entry_barrier);
		if (HOST_ATOMIC_GET(&exiting))
			return 0;

		host_surface* surf = data->surf;
		const int width = surf->w, height = surf->h;

		/* claim all rows at once; exactly one worker gets 0 back */
		if (height > 0 && HOST_ATOMIC_ADD(&current_line, height) == 0) {
			struct hip_renderer* r = HOST_PRIVATE(data);
			if (!r || !r->ready) {
				fprintf(stderr, "hip_renderer: not initialised, frame skipped\n");
			} else if (!sync_format(r, surf)) {
				/* palettised or not 32 bits per pixel: refused (reported once by sync_format), never approximated */
			} else {
				lol_camera cam;
				lol_frame_camera fc;
				HOST_SCENE_CAMERA(data->scene, &cam);   /* the host moves the camera between frames (main.c:180) */
				lol_frame_camera_init(&fc, &cam, width, height);
				int st;
				if (r->multi) {
					st = lol_gpu_multi_render_host(r->multi, &fc, width, height, r->max_steps, surf->pixels, (size_t)surf->pitch);
				} else if (r->pipeline) {
					/* queue this frame, then deliver the one queued on the previous call: the surface shows frame i-1
					 * while frame i renders.  The window is resizable (main.c:157) and the surface is re-fetched every
					 * frame (main.c:182): a frame queued for another size can never go into this surface — it is
					 * dropped, and this call's frame is delivered at once instead (so is the very first frame). */
					int pw = 0, ph = 0;
					lol_gpu_render_host_pending_size(r->gpu, &pw, &ph);
					if ((pw || ph) && (pw != width || ph != height))       /* queued for a surface that no longer exists */
						lol_gpu_render_host_discard(r->gpu);
					const int first = r->shown_w != width || r->shown_h != height;   /* this surface has not been given a frame yet */
					st = lol_gpu_render_host_begin(r->gpu, &fc, width, height, r->max_steps);
					if (st == LOL_GPU_OK && (first || lol_gpu_render_host_pending(r->gpu) >= r->pipeline))
						st = lol_gpu_render_host_end(r->gpu, surf->pixels, (size_t)surf->pitch, width, height);
					if (st == LOL_GPU_OK) { r->shown_w = width; r->shown_h = height; }
				} else {
					st = lol_gpu_render_host(r->gpu, &fc, width, height, r->max_steps, surf->pixels, (size_t)surf->pitch);
				}
				if (st != LOL_GPU_OK)
					fprintf(stderr, "hip_renderer: %s\n", r->multi ? lol_gpu_multi_error(r->multi) : lol_gpu_error(r->gpu));
			}
		}

		HOST_SEM_POST(frame_exit_barrier);
	}
}

void render_destroy(struct render_data* data) {
	struct hip_renderer* r = HOST_PRIVATE(data);
	if (!r) return;
	if (r->report && (r->gpu || r->multi)) {
		/* the counterpart of the frame log (main.c:196-204) for what happens below the boundary */
		lol_gpu* g = r->gpu ? r->gpu : lol_gpu_multi_context(r->multi, 0);
		lol_gpu_tile_order_info t;
		static const char* const names[] = { "rows", "columns", "auto", "lpt" };
		if (g && lol_gpu_tile_order(g, &t) == LOL_GPU_OK)
			printf("hip_renderer: kernel %s (%s), tile order mode %s, in use %s, %d sort(s) / decision(s)\n", lol_gpu_kernel_name(g),
			       lol_gpu_kernel_key(g), names[t.mode & 3], names[t.order & 3], t.decisions);
	}
	lol_gpu_multi_destroy(r->multi);
	lol_gpu_destroy(r->gpu);
	lol_program_free(&r->program);
	free(r);
	HOST_PRIVATE(data) = NULL;
}
