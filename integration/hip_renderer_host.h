/*
 * hip_renderer_host.h — the host primitives hip_renderer.c needs, in two spellings.
 *
 *  -DLOL_HOST_SDL   inside the reference tree: the reference's own renderer.h
 *                   (SDL surface, SDL atomics/semaphores, struct render_data,
 *                   the four extern globals of renderer.h:6-9).
 *  (default)        headless: the same protocol on pthreads / C11 atomics with
 *                   a plain memory surface, so the adapter and its frame
 *                   protocol can be built and tested where SDL2 does not exist
 *                   (this image, the GPU box).  lol_headless.c is that host.
 */
#ifndef HIP_RENDERER_HOST_H
#define HIP_RENDERER_HOST_H

#ifdef LOL_HOST_SDL

#include "renderer.h"                       /* reference header: SDL.h + scene.h */
#include "lol_refscene.h"

typedef SDL_Surface host_surface;
#define HOST_SEM_WAIT(s)         SDL_SemWait(s)
#define HOST_SEM_POST(s)         SDL_SemPost(s)
#define HOST_ATOMIC_GET(a)       SDL_AtomicGet(a)
#define HOST_ATOMIC_ADD(a, v)    SDL_AtomicAdd((a), (v))
#define HOST_SURF_BPP(s)         ((s)->format->BytesPerPixel)
/* SDL_PixelFormat → lol_gpu_pixel_format: the fields SDL_MapRGB reads (renderer.h:17-22) */
#define HOST_SURF_FORMAT(s, f)   do { const SDL_PixelFormat* pf_ = (s)->format;                                   \
                                      (f)->r_shift = pf_->Rshift; (f)->g_shift = pf_->Gshift; (f)->b_shift = pf_->Bshift; \
                                      (f)->r_loss = pf_->Rloss; (f)->g_loss = pf_->Gloss; (f)->b_loss = pf_->Bloss;       \
                                      (f)->bytes_per_pixel = pf_->BytesPerPixel; (f)->palettised = pf_->palette != NULL; \
                                      (f)->a_mask = pf_->Amask; } while (0)
#define HOST_PRIVATE(d)          ((d)->private)
/* the reference scene is a pointer graph; convert it once in render_prepare */
#define HOST_SCENE_TO_LOL(sc)    lol_scene_from_reference(sc)
#define HOST_SCENE_CAMERA(sc, out) lol_camera_from_reference((sc), (out))

#else  /* headless */

#include <pthread.h>
#include <semaphore.h>
#include <stdatomic.h>
#include <stdint.h>
#include "lol_scene.h"
#include "lol_gpu.h"

typedef struct host_surface {
	int      w, h;
	int      pitch;            /* bytes per row */
	int      bytes_per_pixel;
	void*    pixels;
	lol_gpu_pixel_format format;   /* what SDL_Surface.format holds: shifts, losses, Amask (bytes_per_pixel above wins) */
} host_surface;

/* same three fields as renderer.h:11-15 */
struct render_data {
	host_surface*    surf;
	const lol_scene* scene;
	void*            private_;
};

extern atomic_int exiting;
extern atomic_int current_line;
extern sem_t*     frame_entry_barrier;
extern sem_t*     frame_exit_barrier;

#define HOST_SEM_WAIT(s)         sem_wait(s)
#define HOST_SEM_POST(s)         sem_post(s)
#define HOST_ATOMIC_GET(a)       atomic_load(a)
#define HOST_ATOMIC_ADD(a, v)    atomic_fetch_add((a), (v))
#define HOST_SURF_BPP(s)         ((s)->bytes_per_pixel)
#define HOST_SURF_FORMAT(s, f)   do { *(f) = (s)->format; (f)->bytes_per_pixel = (uint8_t)(s)->bytes_per_pixel; } while (0)
#define HOST_PRIVATE(d)          ((d)->private_)
#define HOST_SCENE_TO_LOL(sc)    (sc)
#define HOST_SCENE_CAMERA(sc, out) (*(out) = (sc)->camera)

int  render_thread(void* ptr);
void render_prepare(struct render_data* data, int argc, const char* argv[]);
void render_destroy(struct render_data* data);

#endif
#endif
