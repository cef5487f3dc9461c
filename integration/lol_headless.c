/*
 * lol_headless.c — a window-less host that drives a renderer.h plug-in the way the
 * reference's main.c does, minus SDL (SDL2 exists neither in the build image nor on
 * the GPU box).
 *
 *   lol_headless <threads> <scene.lol> [--size WxH] [--frames N] [--out frame.ppm]
 *                [--orbit] [--keys SCRIPT] [--dump-camera FILE] [--pipeline]
 *                [renderer flags: --device N | --devices A,B,.. --max-steps N]
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
	for (int i = 3; i < argc; i++) {
		if (!strcmp(argv[i], "--size") && i + 1 < argc) sscanf(argv[++i], "%dx%d", &w, &h);
		else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
		else if (!strcmp(argv[i], "--orbit")) orbit = 1;
		else if (!strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
		else if (!strcmp(argv[i], "--dump-camera") && i + 1 < argc) dump_camera = argv[++i];
		else if (!strcmp(argv[i], "--pipeline")) pipeline = 1;      /* also read by the plug-in: the surface lags one frame */
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

	host_surface surf = { .w = w, .h = h, .pitch = (w + 13) * 4, .bytes_per_pixel = 4 };
	surf.pixels = calloc((size_t)surf.pitch, (size_t)h);
	struct render_data data = { .surf = &surf, .scene = scene, .private_ = NULL };

	LOG("Inicializando threads");
	pthread_t* tid = calloc((size_t)threads, sizeof *tid);
	for (int i = 0; i < threads; i++) pthread_create(&tid[i], NULL, worker_main, &data);

	render_prepare(&data, argc, argv);

	FILE* cam_fp = dump_camera ? fopen(dump_camera, "w") : NULL;
	const char* key_at = keys;
	double tmin = 1e30, tmax = 0, tsum = 0;
	/* with --pipeline the plug-in delivers frame i-1 on round i: one extra round (same camera) brings the last frame in */
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
		atomic_store(&current_line, 0);
		double t0 = now_ms();
		for (int i = 0; i < threads; i++) sem_post(&entry);
		for (int i = 0; i < threads; i++) sem_wait(&exit_);
		double dt = now_ms() - t0;
		if (dt < tmin) tmin = dt;
		if (dt > tmax) tmax = dt;
		tsum += dt;
		LOG("Frame %d: %.3fms (min: %.3f max: %.3f avg: %.3f) %.1f Mpixels/s", f + 1, dt, tmin, tmax,
		    tsum / (f + 1), w * (double)h / dt / 1e3);
	}

	if (out) {
		FILE* fp = fopen(out, "wb");
		if (!fp) { perror(out); return 1; }
		fprintf(fp, "P6\n%d %d\n255\n", w, h);
		for (int y = 0; y < h; y++) {
			const uint32_t* row = (const uint32_t*)((const char*)surf.pixels + (size_t)y * surf.pitch);
			for (int x = 0; x < w; x++) {
				unsigned char rgb[3] = { (unsigned char)(row[x] >> 16), (unsigned char)(row[x] >> 8), (unsigned char)row[x] };
				fwrite(rgb, 1, 3, fp);
			}
		}
		fclose(fp);
	}

	if (cam_fp) fclose(cam_fp);
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
