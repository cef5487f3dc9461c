/*
 * lol_refscene.c — walk the reference's scene graph (scene.h:44-96, vector.h:16-21)
 * and rebuild it as a lol_scene.  Compiled only where the reference headers
 * exist: by `make hip` in the reference tree and by
 * oracle/Makefile `ref` for the conversion test (tests/test_refscene.py).
 */
#include <stdlib.h>
#include <string.h>

#include "lol_refscene.h"

static lol_v3 cv(v3 v) { return (lol_v3){ v.x, v.y, v.z }; }

static size_t count_nodes(const struct object* o) {
	if (o->type != OBJ_SMOOTH_UNION) return 1;
	return 1 + count_nodes(o->smooth_op.a) + count_nodes(o->smooth_op.b);
}

static int32_t convert(const struct object* o, lol_scene* out) {
	int32_t idx = (int32_t)out->n_nodes++;
	lol_node n;
	memset(&n, 0, sizeof n);
	n.a = n.b = -1;
	n.material = (uint32_t)o->material;
	n.point = cv(o->point);
	switch (o->type) {
	case OBJ_SPHERE: n.type = LOL_NODE_SPHERE; n.radius = o->sphere.radius; break;
	case OBJ_BOX:    n.type = LOL_NODE_BOX; n.half_extent = cv(o->box.point2); n.radius = o->box.radius; break;
	case OBJ_PLANE:  n.type = LOL_NODE_PLANE; break;
	default:
		n.type = LOL_NODE_SMOOTH_UNION;
		n.smoothness = o->smooth_op.smoothness;
		n.a = convert(o->smooth_op.a, out);
		n.b = convert(o->smooth_op.b, out);
	}
	out->nodes[idx] = n;
	return idx;
}

void lol_camera_from_reference(const struct scene* ref, lol_camera* out) {
	out->point = cv(ref->camera.point);
	out->direction = cv(ref->camera.direction);
	out->fov = ref->camera.fov;
}

lol_scene* lol_scene_from_reference(const struct scene* ref) {
	lol_scene* s = lol_scene_new();
	if (!s) return NULL;
	size_t total = 0;
	vector_foreach(struct object, ref->objects, o) total += count_nodes(o);

	s->materials = calloc(ref->materials->size ? ref->materials->size : 1, sizeof *s->materials);
	s->lights    = calloc(ref->lights->size ? ref->lights->size : 1, sizeof *s->lights);
	s->nodes     = calloc(total ? total : 1, sizeof *s->nodes);
	s->roots     = calloc(ref->objects->size ? ref->objects->size : 1, sizeof *s->roots);
	if (!s->materials || !s->lights || !s->nodes || !s->roots) { lol_scene_free(s); return NULL; }

	vector_foreach(struct material, ref->materials, m)
		s->materials[s->n_materials++] = (lol_material){ m->shininess, cv(m->diffuse), cv(m->specular), cv(m->ambient) };
	vector_foreach(struct light, ref->lights, l)
		s->lights[s->n_lights++] = (lol_light){ cv(l->point), cv(l->diffuse_intensity), cv(l->specular_intensity) };
	vector_foreach(struct object, ref->objects, o)
		s->roots[s->n_roots++] = convert(o, s);
	s->ambient_color = cv(ref->ambient_color);
	lol_camera_from_reference(ref, &s->camera);
	return s;
}
