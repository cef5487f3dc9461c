/*
 * lol_headless.c — a window-less host that drives a renderer.h plug-in the way the
 * reference's main.c does, minus SDL (SDL2 exists neither in the build image nor on
 * the GPU box).
 *
 *   lol_headless <threads> <scene.lol> [--size WxH] [--frames N] [--out frame.ppm]
 *                [--orbit] [--keys SCRIPT] [--dump-camera FILE] [--pipeline | --pipeline-depth N]
 *                [--format NAME] [--resize-script WxH,WxH,..] [--dump-frames PREFIX]
 *                [renderer flags: --device N | --devices A,B,.. --max-steps N ...]
 *   --format NAME            pixel format of the surface, as SDL names it: xrgb8888 (default), argb8888, bgrx8888,
 *                            rgba8888, abgr8888; rgb565 and index8 exist to see the plug-in refuse them
 *   --resize-script LIST     the window is resizable (main.c:157) and its surface re-fetched every frame (main.c:182):
 *                            frame i is rendered at the i-th size of the list (the last one repeats); a change of size
 *                            frees the surface and allocates a new one, as SDL does
 *   --dump-frames PREFIX     after every frame: PREFIX%04d.raw = "LOLF", w, h (int32), then h rows of w pixels
 *   --keys "W,W,WA,<,<^,."   held keys per frame (W A S D, _ = Space, c = LCtrl, ^ v < > = arrows): the camera is moved
 *                            by main.c's update_camera before every frame, as the windowed host does with key events
 *   --dump-camera FILE       one line per frame: the camera's point and direction as binary32 hex
 *
 * Protocol reproduced from main.c (the caller side of SURVEY.md §8b):
 *   - argv[1] = worker threads, argv[2] = scene file, argv[3..] go to render_prepare  (main.c:223-242)
 *   - workers are created BEFORE render_prepare and park on the entry semaphore      (main.c:145-161)
 *   - per frame: camera update; current_line = 0; post entry ×N; wait exit ×N          (main.c:180-194)
 *   - per-frame ms with running min / max / avg                                         (main.c:196-204)
 *   - shutdown: exiting = 1; post entry ×N; join; render_destroy                        (main.c:166-171,213)
 * The window surface is a malloc'd XRGB8888 buffer whose pitch is deliberately wider
 * than w*4 so the plug-in's pitch handling is exercised.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hip_renderer_host.h"
#include "lol_host_input.h"

atomic_int exiting;
atomic_int current_line;
sem_t*     frame_entry_barrier;
sem_t*     frame_exit_barrier;

#define LOG(fmt, ...) printf("[%s:%d] " fmt "\n", __FILE__, __LINE__, ##__VA_ARGS__)

static void* worker_main(void* arg) { render_thread(arg); return NULL; }

/* SDL_PixelFormat fields of the formats --format knows (SDL_pixels.h: SDL_PIXELFORMAT_*8888 are packed 32-bit) */
static int format_by_name(const char* name, lol_gpu_pixel_format* f, int* bytes_per_pixel) {
	static const struct { const char* name; lol_gpu_pixel_format f; } known[] = {
		{ "xrgb8888", { 16, 8, 0, 0, 0, 0, 4, 0, 0x00000000u } },
		{ "argb8888", { 16, 8, 0, 0, 0, 0, 4, 0, 0xFF000000u } },
		{ "bgrx8888", { 8, 16, 24, 0, 0, 0, 4, 0, 0x00000000u } },
		{ "rgba8888", { 24, 16, 8, 0, 0, 0, 4, 0, 0x000000FFu } },
		{ "abgr8888", { 0, 8, 16, 0, 0, 0, 4, 0, 0xFF000000u } },
		{ "rgb565",   { 11, 5, 0, 3, 2, 3, 2, 0, 0x00000000u } },
		{ "index8",   { 0, 0, 0, 8, 8, 8, 1, 1, 0x00000000u } },
	};
	for (size_t i = 0; i < sizeof known / sizeof known[0]; i++)
		if (!strcmp(name, known[i].name)) { *f = known[i].f; *bytes_per_pixel = known[i].f.bytes_per_pixel; return 1; }
	return 0;
}

static double now_ms(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* camera orbit of bench.py / SURVEY.md §8d, frame i of n */
static void orbit_camera(lol_camera* cam, int i, int n) {
	const double cx = 0, cy = 1, cz = -6, R = sqrt(85.0);
	double th = atan2(-2.0, 9.0) + 2.0 * M_PI * i / n;
	double px = cx + R * sin(th), py = cy + 5.0, pz = cz + R * cos(th);
	double dx = cx - px, dy = cy - py, dz = cz - pz, inv = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
	cam->point = (lol_v3){ (float)px, (float)py, (float)pz };
	cam->direction = (lol_v3){ (float)(dx * inv), (float)(dy * inv), (float)(dz * inv) };
}

int main(int argc, const char* argv[]) {
	int threads = argc > 1 ? atoi(argv[1]) : 1;
	const char* path = argc > 2 ? argv[2] : NULL;
	int w = 320, h = 240, frames = 1, orbit = 0, pipeline = 0;         /* main.c:152-159 opens 320x240 */
	const char* out = NULL;
	const char* keys = NULL;
	const char* dump_camera = NULL;
	const char* resize_script = NULL;
	const char* dump_frames = NULL;
	lol_gpu_pixel_format format = { 16, 8, 0, 0, 0, 0, 4, 0, 0 };
	int bytes_per_pixel = 4;
	for (int i = 3; i < argc; i++) {
		if (!strcmp(argv[i], "--size") && i + 1 < argc) sscanf(argv[++i], "%dx%d", &w, &h);
		else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
		else if (!strcmp(argv[i], "--orbit")) orbit = 1;
		else if (!strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
		else if (!strcmp(argv[i], "--dump-camera") && i + 1 < argc) dump_camera = argv[++i];
		else if (!strcmp(argv[i], "--pipeline")) { if (pipeline < 1) pipeline = 1; }      /* also read by the plug-in: the surface lags one frame */
		else if (!strcmp(argv[i], "--pipeline-depth") && i + 1 < argc) {                  /* ... or depth - 1 frames */
			const int d = atoi(argv[++i]);
			if (d >= 2 && d <= 4) pipeline = d - 1;
		}
		else if (!strcmp(argv[i], "--resize-script") && i + 1 < argc) resize_script = argv[++i];
		else if (!strcmp(argv[i], "--dump-frames") && i + 1 < argc) dump_frames = argv[++i];
		else if (!strcmp(argv[i], "--format") && i + 1 < argc) {
			if (!format_by_name(argv[++i], &format, &bytes_per_pixel)) { fprintf(stderr, "unknown --format %s\n", argv[i]); return 2; }
		}
	}
	/* --resize-script: sizes[f] for frame f */
	int sizes[64][2], n_sizes = 0;
	if (resize_script) {
		for (const char* q = resize_script; *q && n_sizes < 64;) {
			int a = 0, b = 0, used = 0;
			if (sscanf(q, "%dx%d%n", &a, &b, &used) != 2 || a < 1 || b < 1) { fprintf(stderr, "bad --resize-script\n"); return 2; }
			sizes[n_sizes][0] = a; sizes[n_sizes][1] = b; n_sizes++;
			q += used;
			if (*q == ',') q++;
		}
		if (n_sizes == 0) { fprintf(stderr, "bad --resize-script\n"); return 2; }
		w = sizes[0][0]; h = sizes[0][1];
		if (frames < n_sizes) frames = n_sizes;
	}
	if (threads < 1) threads = 1;
	if (!path || w < 1 || h < 1) {
		fprintf(stderr, "usage: %s <threads> <scene.lol> [--size WxH] [--frames N] [--out f.ppm] [--orbit]\n", argv[0]);
		return 2;
	}

	char err[256];
	lol_scene* scene = NULL;
	int st = lol_scene_parse_file(path, &scene, err, sizeof err);
	if (st != LOL_OK) { fprintf(stderr, "%s\n", err); return 1; }
	if (!lol_scene_validate_materials(scene)) {            /* main.c:235 asserts this */
		fprintf(stderr, "scene_validate_materials failed\n");
		return 1;
	}

	sem_t entry, exit_;
	sem_init(&entry, 0, 0);
	sem_init(&exit_, 0, 0);
	frame_entry_barrier = &entry;
	frame_exit_barrier = &exit_;
	atomic_store(&exiting, 0);

	host_surface surf = { .w = w, .h = h, .pitch = (w + 13) * 4, .bytes_per_pixel = bytes_per_pixel, .format = format };
	surf.pixels = calloc((size_t)surf.pitch, (size_t)h);
	struct render_data data = { .surf = &surf, .scene = scene, .private_ = NULL };

	LOG("Inicializando threads");
	pthread_t* tid = calloc((size_t)threads, sizeof *tid);
	for (int i = 0; i < threads; i++) pthread_create(&tid[i], NULL, worker_main, &data);

	render_prepare(&data, argc, argv);

	FILE* cam_fp = dump_camera ? fopen(dump_camera, "w") : NULL;
	const char* key_at = keys;
	double tmin = 1e30, tmax = 0, tsum = 0;
	double* times = calloc((size_t)(frames + pipeline), sizeof *times);
	/* with --pipeline the plug-in delivers frame i-1 on round i (i - depth + 1 with --pipeline-depth): extra rounds (same camera)
	 * bring the last frames in */
	for (int f = 0; f < frames + pipeline; f++) {
		if (orbit) orbit_camera(&scene->camera, f < frames ? f : frames - 1, frames);
		if (keys && f < frames) {                                /* update_camera(), main.c:180 */
			const char* endp = key_at ? strchr(key_at, ',') : NULL;
			lol_keys k = lol_keys_from_script(key_at ? key_at : "", key_at ? (endp ? (size_t)(endp - key_at) : strlen(key_at)) : 0);
			key_at = endp ? endp + 1 : NULL;                     /* past the script: no keys held */
			lol_host_update_camera(&scene->camera, &k);
		}
		if (cam_fp && f < frames) {
			const float v[6] = { scene->camera.point.x, scene->camera.point.y, scene->camera.point.z,
			                     scene->camera.direction.x, scene->camera.direction.y, scene->camera.direction.z };
			for (int j = 0; j < 6; j++) { uint32_t u; memcpy(&u, &v[j], 4); fprintf(cam_fp, "%08x%c", u, j == 5 ? '\n' : ' '); }
		}
		if (n_sizes) {                                           /* SDL_GetWindowSurface after a resize: a new surface */
			const int k = f < n_sizes ? f : n_sizes - 1;
			if (sizes[k][0] != surf.w || sizes[k][1] != surf.h) {
				free(surf.pixels);
				w = surf.w = sizes[k][0]; h = surf.h = sizes[k][1];
				surf.pitch = (w + 13) * 4;
				surf.pixels = calloc((size_t)surf.pitch, (size_t)h);
			}
		}
		atomic_store(&current_line, 0);
		double t0 = now_ms();
		for (int i = 0; i < threads; i++) sem_post(&entry);
		for (int i = 0; i < threads; i++) sem_wait(&exit_);
		double dt = now_ms() - t0;
		if (dt < tmin) tmin = dt;
		if (dt > tmax) tmax = dt;
		tsum += dt;
		if (times) times[f] = dt;
		LOG("Frame %d: %.3fms (min: %.3f max: %.3f avg: %.3f) %.1f Mpixels/s", f + 1, dt, tmin, tmax,
		    tsum / (f + 1), w * (double)h / dt / 1e3);
		if (dump_frames) {
			char name[512];
			snprintf(name, sizeof name, "%s%04d.raw", dump_frames, f);
			FILE* fp = fopen(name, "wb");
			if (!fp) { perror(name); return 1; }
			const int32_t hdr[2] = { w, h };
			fwrite("LOLF", 1, 4, fp);
			fwrite(hdr, 4, 2, fp);
			for (int y = 0; y < h; y++) fwrite((const char*)surf.pixels + (size_t)y * surf.pitch, 4, (size_t)w, fp);
			fclose(fp);
		}
	}

	if (out) {
		FILE* fp = fopen(out, "wb");
		if (!fp) { perror(out); return 1; }
		fprintf(fp, "P6\n%d %d\n255\n", w, h);
		for (int y = 0; y < h; y++) {
			const uint32_t* row = (const uint32_t*)((const char*)surf.pixels + (size_t)y * surf.pitch);
			for (int x = 0; x < w; x++) {
				unsigned char rgb[3] = { (unsigned char)((row[x] >> format.r_shift) << format.r_loss),
				                         (unsigned char)((row[x] >> format.g_shift) << format.g_loss),
				                         (unsigned char)((row[x] >> format.b_shift) << format.b_loss) };
				fwrite(rgb, 1, 3, fp);
			}
		}
		fclose(fp);
	}

	if (cam_fp) fclose(cam_fp);
	if (times && frames + pipeline >= 8) {               /* steady state: the median is not moved by the first frames' set-up */
		const int n = frames + pipeline;
		for (int i = 1; i < n; i++)                      /* insertion sort: a few hundred frames at most */
			for (int j = i; j > 0 && times[j] < times[j - 1]; j--) { double t = times[j]; times[j] = times[j - 1]; times[j - 1] = t; }
		LOG("Median: %.3fms %.1f Mpixels/s", times[n / 2], w * (double)h / times[n / 2] / 1e3);
	}
	free(times);
	LOG("Cerrando");
	atomic_store(&exiting, 1);
	for (int i = 0; i < threads; i++) sem_post(&entry);
	for (int i = 0; i < threads; i++) pthread_join(tid[i], NULL);
	render_destroy(&data);

	free(tid);
	free(surf.pixels);
	lol_scene_free(scene);
	sem_destroy(&entry);
	sem_destroy(&exit_);
	return 0;
}
