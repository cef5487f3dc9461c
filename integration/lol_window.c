/*
 * lol_window.c — the interactive host of a renderer.h plug-in: a resizable SDL2 window, held keys that move the
 * camera, the window's surface handed to the workers every frame, ms per frame on the log.  What the reference's
 * main.c is to its renderers (main.c:120-221), written against this repository's own scene library so that it needs
 * neither bison/flex nor the reference tree:
 *
 *   lol_window <threads> <scene.lol> [--size WxH] [renderer flags: --device N | --devices A,B,.. | --max-steps N ...]
 *
 * The protocol is lol_headless.c's (and main.c's): workers created before render_prepare and parked on the entry
 * semaphore; per frame: events -> held keys -> camera step (lol_host_input.h = main.c:70-112), the window surface
 * fetched AFRESH (a resize gives a new one: main.c:182-187), current_line = 0, post entry x N, wait exit x N,
 * SDL_UpdateWindowSurface, the frame's milliseconds with running min / max / avg (main.c:196-204); Escape or closing
 * the window ends it.  The plug-in is hip_renderer.c in its headless spelling (hip_renderer_host.h): the SDL surface
 * is described to it field by field — pixels, pitch, BytesPerPixel and the shifts / losses / Amask SDL_MapRGB reads
 * (renderer.h:17-22) — so any 32-bit window format SDL picks comes out right, and others are refused by the plug-in.
 *
 * SDL2 is in neither the build image nor the GPU box: `make window` (loltracer_amd/csrc/Makefile) builds this only
 * where `sdl2-config` exists, and NOTHING here has been compiled against real SDL2 headers or run (INTEGRATION.md says
 * so too).  The frame protocol, the camera step, resizes and pixel formats it relies on are the ones lol_headless.c
 * exercises in tests/test_headless_host.py.
 */
#include <SDL.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hip_renderer_host.h"      /* headless spelling: pthread semaphores, C11 atomics, struct host_surface */
#include "lol_host_input.h"

atomic_int exiting;
atomic_int current_line;
sem_t*     frame_entry_barrier;
sem_t*     frame_exit_barrier;

#define LOG(fmt, ...) printf("[%s:%d] " fmt "\n", __FILE__, __LINE__, ##__VA_ARGS__)

static void* worker_main(void* arg) { render_thread(arg); return NULL; }

/* a held key: down sets, up clears (main.c keeps the same table, main.c:26-68) */
static void key_event(lol_keys* k, SDL_Keycode sym, int down) {
	switch (sym) {
	case SDLK_w: k->W = down; break;
	case SDLK_a: k->A = down; break;
	case SDLK_s: k->S = down; break;
	case SDLK_d: k->D = down; break;
	case SDLK_SPACE: k->Space = down; break;
	case SDLK_LCTRL: k->LCtrl = down; break;
	case SDLK_LEFT: k->Left = down; break;
	case SDLK_RIGHT: k->Right = down; break;
	case SDLK_UP: k->Up = down; break;
	case SDLK_DOWN: k->Down = down; break;
	default: break;
	}
}

/* what the plug-in needs to know about the window's surface this frame */
static void describe_surface(const SDL_Surface* s, host_surface* out) {
	const SDL_PixelFormat* pf = s->format;
	out->w = s->w;
	out->h = s->h;
	out->pitch = s->pitch;
	out->bytes_per_pixel = pf->BytesPerPixel;
	out->pixels = s->pixels;
	out->format.r_shift = pf->Rshift; out->format.g_shift = pf->Gshift; out->format.b_shift = pf->Bshift;
	out->format.r_loss = pf->Rloss;   out->format.g_loss = pf->Gloss;   out->format.b_loss = pf->Bloss;
	out->format.bytes_per_pixel = pf->BytesPerPixel;
	out->format.palettised = pf->palette != NULL;
	out->format.a_mask = pf->Amask;
}

int main(int argc, char* argv[]) {
	int threads = argc > 1 ? atoi(argv[1]) : 1;
	const char* path = argc > 2 ? argv[2] : NULL;
	int w = 320, h = 240;                                   /* main.c:152-159 opens 320x240 */
	for (int i = 3; i < argc; i++)
		if (!strcmp(argv[i], "--size") && i + 1 < argc) sscanf(argv[++i], "%dx%d", &w, &h);
	if (threads < 1) threads = 1;
	if (!path || w < 1 || h < 1) {
		fprintf(stderr, "usage: %s <threads> <scene.lol> [--size WxH] [renderer flags]\n", argv[0]);
		return 2;
	}

	char err[256];
	lol_scene* scene = NULL;
	if (lol_scene_parse_file(path, &scene, err, sizeof err) != LOL_OK) { fprintf(stderr, "%s\n", err); return 1; }
	if (!lol_scene_validate_materials(scene)) { fprintf(stderr, "scene_validate_materials failed\n"); return 1; }

	if (SDL_Init(SDL_INIT_VIDEO) != 0) { fprintf(stderr, "SDL_Init: %s\n", SDL_GetError()); return 1; }
	SDL_Window* window = SDL_CreateWindow("lol_window", SDL_WINDOWPOS_UNDEFINED, SDL_WINDOWPOS_UNDEFINED, w, h, SDL_WINDOW_RESIZABLE);
	if (!window) { fprintf(stderr, "SDL_CreateWindow: %s\n", SDL_GetError()); SDL_Quit(); return 1; }

	sem_t entry, exit_;
	sem_init(&entry, 0, 0);
	sem_init(&exit_, 0, 0);
	frame_entry_barrier = &entry;
	frame_exit_barrier = &exit_;
	atomic_store(&exiting, 0);

	host_surface surf;
	memset(&surf, 0, sizeof surf);
	{
		SDL_Surface* first = SDL_GetWindowSurface(window);
		if (!first) { fprintf(stderr, "SDL_GetWindowSurface: %s\n", SDL_GetError()); SDL_DestroyWindow(window); SDL_Quit(); return 1; }
		describe_surface(first, &surf);                      /* render_prepare may look at the size */
	}
	struct render_data data = { .surf = &surf, .scene = scene, .private_ = NULL };

	LOG("Inicializando threads");
	pthread_t* tid = calloc((size_t)threads, sizeof *tid);
	for (int i = 0; i < threads; i++) pthread_create(&tid[i], NULL, worker_main, &data);
	render_prepare(&data, argc, (const char**)argv);

	lol_keys held;
	memset(&held, 0, sizeof held);
	double tmin = 1e30, tmax = 0, tsum = 0;
	unsigned long frame = 0;
	const double ticks_per_ms = (double)SDL_GetPerformanceFrequency() / 1e3;
	for (int running = 1; running;) {
		SDL_Event ev;
		while (SDL_PollEvent(&ev)) {
			if (ev.type == SDL_QUIT) running = 0;
			else if (ev.type == SDL_KEYDOWN && ev.key.keysym.sym == SDLK_ESCAPE) running = 0;
			else if (ev.type == SDL_KEYDOWN || ev.type == SDL_KEYUP) key_event(&held, ev.key.keysym.sym, ev.type == SDL_KEYDOWN);
		}
		if (!running) break;
		lol_host_update_camera(&scene->camera, &held);       /* before the workers are released (main.c:180) */

		SDL_Surface* win = SDL_GetWindowSurface(window);     /* afresh every frame: after a resize it is another surface */
		if (!win) { fprintf(stderr, "SDL_GetWindowSurface: %s\n", SDL_GetError()); break; }
		if (SDL_MUSTLOCK(win) && SDL_LockSurface(win) != 0) { fprintf(stderr, "SDL_LockSurface: %s\n", SDL_GetError()); break; }
		describe_surface(win, &surf);

		atomic_store(&current_line, 0);
		const Uint64 t0 = SDL_GetPerformanceCounter();
		for (int i = 0; i < threads; i++) sem_post(&entry);
		for (int i = 0; i < threads; i++) sem_wait(&exit_);
		const double dt = (double)(SDL_GetPerformanceCounter() - t0) / ticks_per_ms;

		if (SDL_MUSTLOCK(win)) SDL_UnlockSurface(win);
		SDL_UpdateWindowSurface(window);
		frame++;
		if (dt < tmin) tmin = dt;
		if (dt > tmax) tmax = dt;
		tsum += dt;
		LOG("Frame %lu: %.3fms (min: %.3f max: %.3f avg: %.3f) %dx%d %.1f Mpixels/s", frame, dt, tmin, tmax, tsum / (double)frame,
		    surf.w, surf.h, surf.w * (double)surf.h / dt / 1e3);
	}

	LOG("Cerrando");
	atomic_store(&exiting, 1);
	for (int i = 0; i < threads; i++) sem_post(&entry);
	for (int i = 0; i < threads; i++) pthread_join(tid[i], NULL);
	render_destroy(&data);
	free(tid);
	lol_scene_free(scene);
	sem_destroy(&entry);
	sem_destroy(&exit_);
	SDL_DestroyWindow(window);
	SDL_Quit();
	return 0;
}
