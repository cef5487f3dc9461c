/*
 * fuzz_scene.c — mutation fuzzer for the `.lol` reader and flattener (loltracer_amd/csrc/lol_scene.c), built by
 * tests/test_scene_fuzz.py with -fsanitize=address,undefined.  Test infrastructure.
 *
 *   fuzz_scene <seed> <iterations> <scene.lol> [more.lol ...]
 *
 * Every iteration takes one of the base documents and damages it (byte flips, deletions, duplications of spans,
 * insertion of grammar tokens, truncation), then parses it, flattens what parsed, computes a frame camera and frees
 * everything.  The reader must return a status — never crash, overflow, leak or hang — whatever the input
 * (the reference's flex/bison front-end ignores unknown characters and reports syntax errors; scene-lexer.l:48-50).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lol_scene.h"

static uint64_t rng_state;
static uint32_t rnd(void) {
	rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
	return (uint32_t)(rng_state >> 16);
}

static const char* const TOKENS[] = {
	"{", "}", "(", ")", ",", "=", "#0", "#7", "#99999999999", "materials", "scene", "sphere", "box", "plane", "smooth_union",
	"smooth-union", "point_light", "camera", "ambient", "radius", "point", "point2", "smoothness", "a", "b", "material",
	"-", "1e9", "..", "-.5", "99999999999999999999999999999999999999999", "0", "fov", "y", "color", "direction",
	"a = sphere { radius = 1 }", "b = smooth_union { a = plane { y = 0 }, b = box { point2 = (1,1,1) } }", "\n", "\t", "\0x",
};

int main(int argc, char** argv) {
	if (argc < 4) { fprintf(stderr, "usage: %s seed iterations scene.lol...\n", argv[0]); return 2; }
	rng_state = strtoull(argv[1], NULL, 10) * 0x9E3779B97F4A7C15ull + 1;
	long iters = atol(argv[2]);
	int n_base = argc - 3;
	char** base = calloc((size_t)n_base, sizeof *base);
	size_t* base_len = calloc((size_t)n_base, sizeof *base_len);
	for (int i = 0; i < n_base; i++) {
		FILE* f = fopen(argv[3 + i], "rb");
		if (!f) { perror(argv[3 + i]); return 2; }
		fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
		base[i] = malloc((size_t)n + 1);
		base_len[i] = fread(base[i], 1, (size_t)n, f);
		fclose(f);
	}
	long parsed = 0, flattened = 0;
	lol_program* prog = malloc(sizeof *prog);
	for (long it = 0; it < iters; it++) {
		int b = (int)(rnd() % (uint32_t)n_base);
		size_t cap = base_len[b] * 2 + 4096, len = base_len[b];
		char* doc = malloc(cap);
		memcpy(doc, base[b], len);
		int edits = 1 + (int)(rnd() % 3);
		for (int e = 0; e < edits && len > 0; e++) {
			size_t at = rnd() % len;
			switch (rnd() % 6) {
			case 0: doc[at] = (char)rnd(); break;                                           /* byte flip */
			case 1: { size_t n = 1 + rnd() % 40; if (at + n > len) n = len - at;               /* delete a span */
			          memmove(doc + at, doc + at + n, len - at - n); len -= n; break; }
			case 2: { size_t n = 1 + rnd() % 200; if (at + n > len) n = len - at;              /* duplicate a span */
			          if (len + n < cap) { memmove(doc + at + n, doc + at, len - at); len += n; } break; }
			case 3: { const char* t = TOKENS[rnd() % (sizeof TOKENS / sizeof *TOKENS)];       /* insert a token */
			          size_t n = strlen(t);
			          if (len + n < cap) { memmove(doc + at + n, doc + at, len - at); memcpy(doc + at, t, n); len += n; } break; }
			case 4: len = at; break;                                                          /* truncate */
			default: { size_t to = rnd() % len; char c = doc[at]; doc[at] = doc[to]; doc[to] = c; break; }
			}
		}
		lol_scene* sc = NULL;
		char err[128];
		int st = lol_scene_parse_string(doc, len, &sc, err, sizeof err);
		if (st == LOL_OK && sc) {
			parsed++;
			(void)lol_scene_validate_materials(sc);
			if (lol_scene_flatten(sc, prog) == LOL_OK) {
				flattened++;
				lol_frame_camera fc;
				lol_frame_camera_init(&fc, &sc->camera, 64, 36);
				/* every table is as long as its count says (ASan sees an overrun) */
				uint32_t touch = 0;
				for (uint32_t k = 0; k < prog->n_ops; k++) touch += prog->ops[k].op;
				for (uint32_t k = 0; k < prog->n_roots; k++) touch += prog->root_material[k];
				if (prog->n_lights) touch += (uint32_t)prog->lights[prog->n_lights - 1].point.x;
				touch += (uint32_t)prog->materials[prog->n_materials - 1].shininess;
				(void)touch;
				lol_program_free(prog);
			} else if (prog->ops || prog->lights || prog->materials || prog->root_material) {
				fprintf(stderr, "a failed flatten left tables behind\n");
				return 1;
			}
			lol_scene_free(sc);
		} else if (sc) {
			fprintf(stderr, "scene returned with status %d\n", st);
			return 1;
		}
		free(doc);
	}
	printf("iterations %ld parsed %ld flattened %ld\n", iters, parsed, flattened);
	free(prog);
	for (int i = 0; i < n_base; i++) free(base[i]);
	free(base); free(base_len);
	return 0;
}
