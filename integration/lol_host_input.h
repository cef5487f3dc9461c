/*
 * lol_host_input.h — the host's keyboard → camera step, for hosts of the renderer.h plug-in.
 *
 * The reference's main.c keeps a table of held keys (main.c:26-68) and, once per frame before the workers are
 * released (main.c:180), moves the camera by it (update_camera, main.c:70-112):
 *   right = normalize(cross(direction, (0,1,0)));  up = normalize(cross(right, direction))     (both from the OLD direction)
 *   W / S: point += direction * (+-.1f)      A / D: point += right * (-+.1f)      Space / LCtrl: point.y +-= .1f
 *   Up / Down: direction = normalize(direction + up * (+-.1f))      Left / Right: direction = normalize(direction + right * (-+.1f))
 * applied in exactly that order, all in binary32 with vec.h's operation order (v3normalize = v * (1 / len), len via
 * the DPPS sum (x*x + y*y) + (z*z + 0); compile with -ffp-contract=off).  A windowed host feeds `lol_keys` from its
 * key events; lol_headless feeds it from a script (--keys).  tests/golden/ref_camera_path.json pins the arithmetic
 * against the reference's own compiled vec.h functions.
 */
#ifndef LOL_HOST_INPUT_H
#define LOL_HOST_INPUT_H

#include <math.h>
#include "lol_scene.h"

typedef struct lol_keys {
	int W, A, S, D, Space, LCtrl;      /* movement (main.c:27-28) */
	int Left, Right, Up, Down;         /* rotation  (main.c:29)   */
} lol_keys;

static inline lol_v3 lol_hi_add(lol_v3 a, lol_v3 b) { return (lol_v3){ a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline lol_v3 lol_hi_scale(lol_v3 v, float f) { return (lol_v3){ v.x * f, v.y * f, v.z * f }; }
static inline lol_v3 lol_hi_cross(lol_v3 a, lol_v3 b) {
	return (lol_v3){ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
static inline lol_v3 lol_hi_normalize(lol_v3 v) {
	float lo = v.x * v.x + v.y * v.y, hi = v.z * v.z + 0.0f;
	float inv = 1.0f / sqrtf(lo + hi);
	return lol_hi_scale(v, inv);
}

/* update_camera, main.c:70-112 */
static inline void lol_host_update_camera(lol_camera* cam, const lol_keys* key) {
	const lol_v3 up_guide = { 0.0f, 1.0f, 0.0f };
	const lol_v3 right_dir = lol_hi_normalize(lol_hi_cross(cam->direction, up_guide));
	const lol_v3 up_dir = lol_hi_normalize(lol_hi_cross(right_dir, cam->direction));
	if (key->W) cam->point = lol_hi_add(cam->point, lol_hi_scale(cam->direction, .1f));
	if (key->A) cam->point = lol_hi_add(cam->point, lol_hi_scale(right_dir, -.1f));
	if (key->S) cam->point = lol_hi_add(cam->point, lol_hi_scale(cam->direction, -.1f));
	if (key->D) cam->point = lol_hi_add(cam->point, lol_hi_scale(right_dir, .1f));
	if (key->Space) cam->point.y += .1f;
	if (key->LCtrl) cam->point.y -= .1f;
	if (key->Up) cam->direction = lol_hi_normalize(lol_hi_add(cam->direction, lol_hi_scale(up_dir, .1f)));
	if (key->Down) cam->direction = lol_hi_normalize(lol_hi_add(cam->direction, lol_hi_scale(up_dir, -.1f)));
	if (key->Left) cam->direction = lol_hi_normalize(lol_hi_add(cam->direction, lol_hi_scale(right_dir, -.1f)));
	if (key->Right) cam->direction = lol_hi_normalize(lol_hi_add(cam->direction, lol_hi_scale(right_dir, .1f)));
}

/* One frame of a key script: letters W A S D, '_' Space, 'c' LCtrl, '^' 'v' '<' '>' the arrows; anything else is ignored. */
static inline lol_keys lol_keys_from_script(const char* s, size_t n) {
	lol_keys k = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	for (size_t i = 0; i < n; i++) {
		switch (s[i]) {
		case 'W': case 'w': k.W = 1; break;
		case 'A': case 'a': k.A = 1; break;
		case 'S': case 's': k.S = 1; break;
		case 'D': case 'd': k.D = 1; break;
		case '_': k.Space = 1; break;
		case 'c': case 'C': k.LCtrl = 1; break;
		case '^': k.Up = 1; break;
		case 'v': case 'V': k.Down = 1; break;
		case '<': k.Left = 1; break;
		case '>': k.Right = 1; break;
		default: break;
		}
	}
	return k;
}

#endif /* LOL_HOST_INPUT_H */
