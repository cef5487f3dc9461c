/*
 * lol_scene.h — host-side scene model, `.lol` reader and SDF-program flattener.
 *
 * C ABI, no SDL, no torch.  This is the build's own restatement of the data
 * model that feeds loltracer's per-pixel path:
 *
 *   reference                          here
 *   ---------------------------------  -----------------------------------------
 *   struct scene      scene.h:90-96    lol_scene   (index-linked, no pointers)
 *   struct object     scene.h:58-82    lol_node    (children are node indices)
 *   struct material   scene.h:44-49    lol_material
 *   struct light      scene.h:52-56    lol_light
 *   struct camera     scene.h:84-88    lol_camera
 *   scene_parse()     scene-parser.y:197-214   lol_scene_parse_file/_string
 *   scene_validate_materials() scene.c:284-292 lol_scene_validate_materials
 *
 * The reference stores the SDF tree as malloc'd `struct object*` children; a
 * GPU wants a pointer-free program it can keep in LDS/SGPRs, so the scene is
 * held as one node array plus a list of top-level roots, and
 * lol_scene_flatten() lowers it to a post-order op list (lol_program) that the
 * HIP kernel interprets (include/lol_gpu.h).
 */
#ifndef LOL_SCENE_H
#define LOL_SCENE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lol_v3 { float x, y, z; } lol_v3;

/* scene.h:44-49 */
typedef struct lol_material {
	float  shininess;
	lol_v3 diffuse;
	lol_v3 specular;
	lol_v3 ambient;
} lol_material;

/* scene.h:52-56 */
typedef struct lol_light {
	lol_v3 point;
	lol_v3 diffuse_intensity;
	lol_v3 specular_intensity;
} lol_light;

/* scene.h:84-88; `direction` normalised, `fov` in radians (scene.c:173-174) */
typedef struct lol_camera {
	lol_v3 point;
	lol_v3 direction;
	float  fov;
} lol_camera;

/* Distance-field node kinds (the OBJ_SPHERE.. subset of scene.h:27-35). */
enum lol_node_type {
	LOL_NODE_SPHERE       = 0,
	LOL_NODE_BOX          = 1,
	LOL_NODE_PLANE        = 2,
	LOL_NODE_SMOOTH_UNION = 3
};

/*
 * One SDF tree node.  Unlike the reference's union, every field always exists
 * and is zero when the `.lol` block did not set it (scene.c:114-124 memsets).
 *   sphere:        point, radius
 *   box:           point, half_extent (= `point2`), radius (= rounding)
 *   plane:         point = (0, y, 0)                       (scene.c:215)
 *   smooth_union:  smoothness, a, b (indices into lol_scene.nodes); `point`
 *                  stays zero and is never used (naive_renderer.c:12,22-24)
 */
typedef struct lol_node {
	int32_t  type;        /* enum lol_node_type */
	uint32_t material;    /* material index; only meaningful on top-level nodes */
	lol_v3   point;
	float    radius;
	lol_v3   half_extent;
	float    smoothness;
	int32_t  a, b;        /* child node indices, -1 when not a smooth_union / unset */
} lol_node;

typedef struct lol_scene {
	lol_material* materials;   size_t n_materials;
	lol_light*    lights;      size_t n_lights;
	lol_node*     nodes;       size_t n_nodes;
	int32_t*      roots;       size_t n_roots;   /* top-level objects, file order; id = index+1 */
	lol_v3        ambient_color;
	lol_camera    camera;
} lol_scene;

/* ------------------------------------------------------------------ parser */

enum lol_status {
	LOL_OK              = 0,
	LOL_ERR_IO          = 1,   /* cannot open file (reference: scene_parse returns NULL) */
	LOL_ERR_SYNTAX      = 2,   /* grammar violation (reference: yyerror "… on line N") */
	LOL_ERR_PROPERTY    = 3,   /* "Unknown <type> property" (reference exits 1, scene.c:130-134) */
	LOL_ERR_TYPE        = 4,   /* value kind mismatch (reference asserts, scene.c:69,81,87,93,99) */
	LOL_ERR_COMPONENT   = 5,   /* "Unknown scene object" for a nested non-object (scene.c:277-279) */
	LOL_ERR_MATERIAL    = 6,   /* lol_scene_validate_materials failed */
	LOL_ERR_NOMEM       = 7,
	LOL_ERR_UNSUPPORTED = 8    /* beyond the sanity caps below (LOL_MAX_*), or an unknown node type */
};

/*
 * Parse a `.lol` document (format: SURVEY.md §5.6).  On success returns LOL_OK
 * and stores a heap scene in *out (free with lol_scene_free).  On failure
 * returns a status and, if errbuf != NULL, a NUL-terminated message that reuses
 * the reference's wording where it has one.
 */
int  lol_scene_parse_file  (const char* path, lol_scene** out, char* errbuf, size_t errcap);
int  lol_scene_parse_string(const char* text, size_t len, lol_scene** out, char* errbuf, size_t errcap);
void lol_scene_free(lol_scene* scene);

/* scene_new() defaults: no objects, camera at origin looking +z, fov pi/2 (scene.c:44-58). */
lol_scene* lol_scene_new(void);

/* scene.c:284-292 — top-level objects only. Returns 1 if valid. */
int lol_scene_validate_materials(const lol_scene* scene);

/* ------------------------------------------------------- flattened program */

/*
 * Post-order stack program for one evaluation of sdf(p) (naive_renderer.c:31-44).
 *
 *   LOL_OP_SPHERE  f = {cx, cy, cz, r}                push sqrt(|p-c|^2) - r
 *   LOL_OP_RBOX    f = {cx, cy, cz, bx, by, bz, r}    push sdRoundBox(p-c, b, r)
 *   LOL_OP_PLANE   f = {py}                           push p.y - py
 *   LOL_OP_SMIN    f = {k}   pops b (top) then a      push sminf(a, b, k)
 *   LOL_OP_SMIN_R  f = {k}   pops a (top) then b      push sminf(a, b, k)
 *                  (children emitted b-first to bound the stack; same value)
 *   LOL_OP_TOP     id        pops d; if (d < best) best = d, best_id = id
 */
enum lol_opcode {
	LOL_OP_SPHERE = 0,
	LOL_OP_RBOX   = 1,
	LOL_OP_PLANE  = 2,
	LOL_OP_SMIN   = 3,
	LOL_OP_SMIN_R = 4,
	LOL_OP_TOP    = 5
};

typedef struct lol_op {
	uint32_t op;     /* enum lol_opcode */
	uint32_t id;     /* LOL_OP_TOP: 1-based object id */
	float    f[7];
	uint32_t _pad;   /* keeps the record at 40 B = 10 dwords */
} lol_op;

/*
 * A program is as large as its scene: the reference's objects, lights and materials are growable vectors and
 * get_obj_dist recurses without a limit (vector.h:16-66, scene.c:18-29,240-259, naive_renderer.c:11-28), so the four
 * tables are a pointer + a count each (rounds 1-3 had fixed arrays of 1024 ops / 64 lights / 256 materials and refused
 * anything larger).  The LOL_MAX_* below are SANITY caps — a count beyond them is taken for corruption, not for a scene
 * (2^20 ops = half a million primitives) — and are checked wherever a program crosses the C ABI.  The operand stack:
 * children are emitted deeper-first, so a stack of d needs 2^(d-1) primitives; 64 covers anything memory can hold.
 */
#define LOL_MAX_OPS        (1u << 20)
#define LOL_MAX_LIGHTS     (1u << 16)
#define LOL_MAX_MATERIALS  (1u << 20)
#define LOL_MAX_STACK      64

typedef struct lol_program {
	uint32_t      n_ops;
	uint32_t      n_lights;
	uint32_t      n_materials;
	uint32_t      n_roots;
	uint32_t      max_stack;                     /* peak operand-stack depth of `ops` */
	lol_v3        ambient_color;
	lol_op*       ops;                           /* n_ops */
	lol_light*    lights;                        /* n_lights */
	lol_material* materials;                     /* n_materials (>= 1: material #0 shades escaped rays) */
	uint32_t*     root_material;                 /* n_roots: material index of object id i+1 */
} lol_program;

/*
 * Lower the scene graph to a program.  Children of a smooth_union are emitted
 * deeper-subtree-first (Sethi–Ullman order) so chains need a 2-entry stack.
 * The four tables are allocated here (release them with lol_program_free; `out` itself is the caller's).
 * Returns LOL_ERR_UNSUPPORTED when a sanity cap above is exceeded, LOL_ERR_MATERIAL when a top-level material
 * index is out of range, LOL_ERR_NOMEM when memory runs out; *out is then empty (all counts 0, all pointers NULL).
 */
int  lol_scene_flatten(const lol_scene* scene, lol_program* out);
/* Frees the tables of a program filled by lol_scene_flatten and empties it (safe on an empty / zeroed one). */
void lol_program_free(lol_program* prog);

/* --------------------------------------------------------------- camera */

/*
 * Per-frame camera constants, hoisted out of get_camera_ray
 * (naive_renderer.c:179-193, whose own TODO at :180 notes they are per-frame):
 *   right = normalize(cross(dir, (0,1,0))),  up = cross(right, dir),
 *   height = atanf(fov/2),  width = aspect*height,  aspect = (float)w/(float)h.
 * Computed in binary32 with the reference's operation order, so a renderer that
 * takes these as inputs produces the same bits as one that recomputes them.
 */
typedef struct lol_frame_camera {
	lol_v3 origin;      /* ro */
	lol_v3 dir;
	lol_v3 right;
	lol_v3 up;
	float  width;       /* aspect * atanf(fov/2) */
	float  height;      /* atanf(fov/2) */
} lol_frame_camera;

void lol_frame_camera_init(lol_frame_camera* fc, const lol_camera* cam, int w, int h);

const char* lol_status_str(int status);

#ifdef __cplusplus
}
#endif
#endif /* LOL_SCENE_H */
