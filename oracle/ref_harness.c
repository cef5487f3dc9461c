/*
 * ref_harness.c — C-ABI window onto the REAL reference primitives.  TEST INFRASTRUCTURE ONLY.
 *
 * Compiled only where /root/reference exists (oracle/Makefile target `ref`),
 * with -I/root/reference so that float.h, vec.h, sdf.h, scene.h and vector.h
 * are the reference's own files, and linked with the reference's own scene.c.
 * Nothing here is a stand-in for a reference file: this translation unit only
 * adds flat `extern "C"`-style entry points so tests can compare the oracle's
 * restated primitives (oracle/lol_oracle.c) and the build's `.lol` reader
 * (loltracer_amd/csrc/lol_scene.c) with what the reference's code computes.
 *
 * naive_renderer.c itself is NOT built: it includes <SDL.h> via renderer.h and
 * SDL2 is not in this image (DESIGN.md "Oracle").
 */
#include <stdint.h>
#include <string.h>

#include "float.h"   /* reference: minf maxf clamp lerp sminf            */
#include "vec.h"     /* reference: v3*                                    */
#include "sdf.h"     /* reference: sdSphere sdBox sdRoundBox              */
#include "scene.h"   /* reference: struct scene & builders (scene.c)      */

static v3 mk(const float a[3]) { return (v3){ a[0], a[1], a[2] }; }
static void st(v3 v, float out[3]) { out[0] = v.x; out[1] = v.y; out[2] = v.z; }

/* ---- float.h ---- */
float ref_minf(float a, float b) { return minf(a, b); }
float ref_maxf(float a, float b) { return maxf(a, b); }
float ref_clamp(float v, float lo, float hi) { return clamp(v, lo, hi); }
float ref_lerp(float a, float b, float t) { return lerp(a, b, t); }
float ref_sminf(float a, float b, float k) { return sminf(a, b, k); }

/* ---- vec.h ---- */
void  ref_v3add(const float a[3], const float b[3], float o[3]) { st(v3add(mk(a), mk(b)), o); }
void  ref_v3sub(const float a[3], const float b[3], float o[3]) { st(v3sub(mk(a), mk(b)), o); }
void  ref_v3mul(const float a[3], const float b[3], float o[3]) { st(v3mul(mk(a), mk(b)), o); }
float ref_v3dot(const float a[3], const float b[3]) { return v3dot(mk(a), mk(b)); }
float ref_v3len(const float a[3]) { return v3len(mk(a)); }
void  ref_v3scale(const float a[3], float f, float o[3]) { st(v3scale(mk(a), f), o); }
void  ref_v3normalize(const float a[3], float o[3]) { st(v3normalize(mk(a)), o); }
void  ref_v3abs(const float a[3], float o[3]) { st(v3abs(mk(a)), o); }
void  ref_v3clamp(const float a[3], float lo, float hi, float o[3]) { st(v3clamp(mk(a), lo, hi), o); }
void  ref_v3pow(const float a[3], float e, float o[3]) { st(v3pow(mk(a), e), o); }
void  ref_v3cross(const float a[3], const float b[3], float o[3]) { st(v3cross(mk(a), mk(b)), o); }

/* ---- sdf.h ---- */
float ref_sd_sphere(const float p[3], float r) { return sdSphere(mk(p), r); }
float ref_sd_box(const float p[3], const float b[3]) { return sdBox(mk(p), mk(b)); }
float ref_sd_round_box(const float p[3], const float b[3], float r) { return sdRoundBox(mk(p), mk(b), r); }

/* ---- scene.c builders, driven with definition lists exactly as the grammar
 *      actions drive them (scene-parser.y:99-145) ------------------------------ */

struct vector* ref_deflist_new(void) { return vector_new(struct definition, 16); }

void ref_deflist_add_num(struct vector* dl, int prop, float num) {
	vector_add(struct definition, dl) = (struct definition){
		prop, (struct definition_value){ .type = VAL_NUM, .num = num } };
}
void ref_deflist_add_list(struct vector* dl, int prop, const float* nums, int n) {
	struct vector* list = vector_new(float, 4);
	for (int i = 0; i < n; i++) vector_add(float, list) = nums[i];
	vector_add(struct definition, dl) = (struct definition){
		prop, (struct definition_value){ .type = VAL_LIST, .list = list } };
}
void ref_deflist_add_id(struct vector* dl, int prop, size_t id) {
	vector_add(struct definition, dl) = (struct definition){
		prop, (struct definition_value){ .type = VAL_ID, .id = id } };
}
/* value: type '{' definition_list '}' — consumes `inner` like the grammar action does */
void ref_deflist_add_obj(struct vector* dl, int prop, int type, struct vector* inner) {
	struct definition_value v = { .type = VAL_OBJ, .obj = object_from_definition_list(type, inner) };
	vector_free(inner, definition_free);
	vector_add(struct definition, dl) = (struct definition){ prop, v };
}
void ref_deflist_free(struct vector* dl) { vector_free(dl, definition_free); }

struct scene* ref_scene_new(void) { return scene_new(); }
void ref_scene_free(struct scene* s) { scene_free(s); }
void ref_scene_add_component(struct scene* s, int type, struct vector* dl) {
	scene_add_component_from_definition_list(s, type, dl);
	vector_free(dl, definition_free);
}
void ref_scene_add_material(struct scene* s, struct vector* dl) {
	vector_add(struct material, s->materials) = material_from_definition_list(dl);
	vector_free(dl, definition_free);
}
int ref_scene_validate_materials(const struct scene* s) { return scene_validate_materials(s); }

/* Flat dump of the reference scene so Python can compare field by field. */
size_t ref_scene_counts(const struct scene* s, int which) {
	switch (which) {
	case 0: return s->materials->size;
	case 1: return s->lights->size;
	case 2: return s->objects->size;
	}
	return 0;
}
void ref_scene_camera(const struct scene* s, float out[7]) {
	st(s->camera.point, out); st(s->camera.direction, out + 3); out[6] = s->camera.fov;
}
void ref_scene_ambient(const struct scene* s, float out[3]) { st(s->ambient_color, out); }
void ref_scene_material(const struct scene* s, size_t i, float out[10]) {
	struct material m = vector_get(struct material, s->materials, i);
	out[0] = m.shininess; st(m.diffuse, out + 1); st(m.specular, out + 4); st(m.ambient, out + 7);
}
void ref_scene_light(const struct scene* s, size_t i, float out[9]) {
	struct light l = vector_get(struct light, s->lights, i);
	st(l.point, out); st(l.diffuse_intensity, out + 3); st(l.specular_intensity, out + 6);
}
const struct object* ref_scene_object(const struct scene* s, size_t i) {
	return &vector_get(struct object, s->objects, i);
}
/* out: type, material, point[3], radius, half[3], smoothness; children returned via a,b */
void ref_object_fields(const struct object* o, int* type, size_t* material, float pt[3],
                       float* radius, float half[3], float* smoothness,
                       const struct object** a, const struct object** b) {
	*type = o->type; *material = o->material; st(o->point, pt);
	*radius = 0; half[0] = half[1] = half[2] = 0; *smoothness = 0; *a = *b = 0;
	switch (o->type) {
	case OBJ_SPHERE: *radius = o->sphere.radius; break;
	case OBJ_BOX: st(o->box.point2, half); *radius = o->box.radius; break;
	case OBJ_PLANE: break;
	case OBJ_SMOOTH_UNION:
		*smoothness = o->smooth_op.smoothness; *a = o->smooth_op.a; *b = o->smooth_op.b; break;
	default: break;
	}
}
/* ABI sizes the survey recorded */
size_t ref_sizeof(int which) {
	switch (which) {
	case 0: return sizeof(struct material);
	case 1: return sizeof(struct light);
	case 2: return sizeof(struct object);
	case 3: return sizeof(struct camera);
	case 4: return sizeof(struct scene);
	case 5: return sizeof(struct vector);
	}
	return 0;
}
