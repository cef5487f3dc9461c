/*
 * lol_refscene.h — reference `struct scene` (scene.h:90-96) → lol_scene.
 * Only meaningful inside the reference tree (needs its scene.h / vector.h).
 */
#ifndef LOL_REFSCENE_H
#define LOL_REFSCENE_H

#include "scene.h"        /* the reference's */
#include "lol_scene.h"

/* Deep-converts the pointer graph into the index-linked model; free with lol_scene_free. */
lol_scene* lol_scene_from_reference(const struct scene* ref);
void       lol_camera_from_reference(const struct scene* ref, lol_camera* out);

#endif
