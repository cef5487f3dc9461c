/*
 * lol_scene.c — `.lol` reader, scene model and SDF-program flattener (host, plain C).
 *
 * Restates, from the reference's published grammar and builder semantics
 * (scene-lexer.l:10-50, scene-parser.y:73-189, scene.c:104-292; documented in
 * SURVEY.md §5.6), what a `.lol` file means — as a hand-written maximal-munch
 * lexer plus a recursive-descent parser (flex/bison are not part of this
 * build), filling an index-linked scene (include/lol_scene.h) instead of the
 * reference's pointer graph.
 *
 * Build: gcc -O2 -ffp-contract=off (camera normalisation must round like the
 * reference's SSE code, see lol_normalize()).
 */
#include "lol_scene.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ float helpers
 * binary32 with the reference's operation order (vec.h:50-59,68-71):
 * dot = (x*x + y*y) + (z*z + 0), len = sqrt(dot), normalize = v * (1/len). */
static float lol_dot(lol_v3 a, lol_v3 b) {
	float xy = a.x * b.x + a.y * b.y;
	float zw = a.z * b.z + 0.0f;
	return xy + zw;
}
static lol_v3 lol_scale(lol_v3 v, float f) { return (lol_v3){ v.x * f, v.y * f, v.z * f }; }
static lol_v3 lol_normalize(lol_v3 v) { return lol_scale(v, 1.0f / sqrtf(lol_dot(v, v))); }
static lol_v3 lol_cross(lol_v3 a, lol_v3 b) {
	return (lol_v3){ a.y * b.z - a.z * b.y,
	                 a.z * b.x - a.x * b.z,
	                 a.x * b.y - a.y * b.x };
}

/* ------------------------------------------------------------------- lexer */

enum tok {
	T_EOF = 0, T_NUM, T_ID,
	T_MATERIALS, T_SCENE,
	/* types — order matches enum comp below */
	T_AMBIENT, T_CAMERA, T_POINT_LIGHT, T_SPHERE, T_BOX, T_PLANE, T_SMOOTH_UNION,
	/* properties (T_AMBIENT doubles as one, scene-lexer.l:18,31) */
	T_SHININESS, T_DIFFUSE, T_SPECULAR, T_COLOR, T_POINT, T_DIRECTION, T_FOV,
	T_DIFFUSE_INTENSITY, T_SPECULAR_INTENSITY, T_RADIUS, T_MATERIAL, T_POINT2,
	T_Y, T_SMOOTHNESS, T_A, T_B,
	T_COMMA, T_LPAREN, T_RPAREN, T_LBRACE, T_RBRACE, T_EQUALS
};

static const struct { const char* text; enum tok tok; } KEYWORDS[] = {
	{ "materials", T_MATERIALS }, { "scene", T_SCENE },
	{ "ambient", T_AMBIENT }, { "camera", T_CAMERA },
	{ "point-light", T_POINT_LIGHT }, { "point_light", T_POINT_LIGHT },
	{ "sphere", T_SPHERE }, { "box", T_BOX }, { "plane", T_PLANE },
	{ "smooth_union", T_SMOOTH_UNION }, { "smooth-union", T_SMOOTH_UNION },
	{ "shininess", T_SHININESS }, { "diffuse", T_DIFFUSE }, { "specular", T_SPECULAR },
	{ "color", T_COLOR }, { "point", T_POINT }, { "direction", T_DIRECTION }, { "fov", T_FOV },
	{ "diffuse_intensity", T_DIFFUSE_INTENSITY }, { "diffuse-intensity", T_DIFFUSE_INTENSITY },
	{ "specular_intensity", T_SPECULAR_INTENSITY }, { "specular-intensity", T_SPECULAR_INTENSITY },
	{ "radius", T_RADIUS }, { "material", T_MATERIAL }, { "point2", T_POINT2 }, { "y", T_Y },
	{ "smoothness", T_SMOOTHNESS }, { "a", T_A }, { "b", T_B },
};
#define N_KEYWORDS (sizeof KEYWORDS / sizeof KEYWORDS[0])

struct lexer {
	const char* s;
	size_t      len, pos;
	size_t      line;          /* scene-lexer.l:3,10 */
	/* The reference's yylval is one union shared by NUM and ID and survives
	 * between tokens; a NUM/ID whose text does not convert (e.g. a lone "-")
	 * keeps the previous bits (scene-lexer.l:12-13). */
	union { float num; int32_t id; } val;
	enum tok    tok;
};

static int is_numch(char c) { return c == '-' || c == '.' || (c >= '0' && c <= '9'); }
static int is_digit(char c) { return c >= '0' && c <= '9'; }

static void lex_next(struct lexer* lx) {
	for (;;) {
		if (lx->pos >= lx->len) { lx->tok = T_EOF; return; }
		char c = lx->s[lx->pos];
		if (c == '\n') { lx->line++; lx->pos++; continue; }
		if (c == ' ' || c == '\r' || c == '\t') { lx->pos++; continue; }

		if (is_numch(c)) {               /* [-.0-9]+ → sscanf("%f") */
			size_t e = lx->pos;
			while (e < lx->len && is_numch(lx->s[e])) e++;
			char   buf[64];
			size_t n = e - lx->pos;
			if (n >= sizeof buf) n = sizeof buf - 1;
			memcpy(buf, lx->s + lx->pos, n);
			buf[n] = 0;
			char* end;
			float v = strtof(buf, &end);
			if (end != buf) lx->val.num = v;
			lx->pos = e;
			lx->tok = T_NUM;
			return;
		}
		if (c == '#' && lx->pos + 1 < lx->len && is_digit(lx->s[lx->pos + 1])) {   /* #[0-9]+ */
			size_t e = lx->pos + 1;
			long   v = 0;
			while (e < lx->len && is_digit(lx->s[e])) {
				if (v < 100000000L) v = v * 10 + (lx->s[e] - '0');
				e++;
			}
			lx->val.id = (int32_t)v;
			lx->pos = e;
			lx->tok = T_ID;
			return;
		}
		/* keywords: longest match wins (flex), earlier rule on ties (no ties here) */
		size_t best_len = 0;
		enum tok best = T_EOF;
		for (size_t k = 0; k < N_KEYWORDS; k++) {
			size_t kl = strlen(KEYWORDS[k].text);
			if (kl > best_len && lx->pos + kl <= lx->len &&
			    memcmp(lx->s + lx->pos, KEYWORDS[k].text, kl) == 0) {
				best_len = kl;
				best = KEYWORDS[k].tok;
			}
		}
		if (best_len) { lx->pos += best_len; lx->tok = best; return; }

		lx->pos++;
		switch (c) {
		case ',': lx->tok = T_COMMA;  return;
		case '(': lx->tok = T_LPAREN; return;
		case ')': lx->tok = T_RPAREN; return;
		case '{': lx->tok = T_LBRACE; return;
		case '}': lx->tok = T_RBRACE; return;
		case '=': lx->tok = T_EQUALS; return;
		default:  continue;           /* scene-lexer.l:50 — silently ignored */
		}
	}
}

/* ------------------------------------------------------------ parse values */

enum comp { C_AMBIENT, C_CAMERA, C_POINT_LIGHT, C_SPHERE, C_BOX, C_PLANE, C_SMOOTH_UNION };

enum vkind { V_NUM, V_LIST, V_ID, V_OBJ };

struct value {
	enum vkind kind;
	float      num;
	float      list[8];
	size_t     list_n;       /* may exceed 8; only the first 8 are stored */
	uint32_t   id;
	int32_t    node;
};

struct def { enum tok prop; struct value v; };

struct deflist { struct def* d; size_t n, cap; };

/* Objects nest through `a = smooth_union { ... }`.  parse_value / parse_deflist recurse once per level, so the
 * depth is bounded (the reference's own bison stack is: YYMAXDEPTH, a few thousand levels of nested objects). */
#define LOL_PARSE_MAX_DEPTH 1024

struct parser {
	struct lexer lx;
	lol_scene*   scene;
	int          depth;      /* current object nesting */
	int          status;
	char*        err;
	size_t       errcap;
};

static void fail(struct parser* p, int status, const char* fmt, ...) {
	if (p->status != LOL_OK) return;
	p->status = status;
	if (p->err && p->errcap) {
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(p->err, p->errcap, fmt, ap);
		va_end(ap);
	}
}
static void syntax_error(struct parser* p) {
	/* yyerror(), scene-parser.y:193-195 */
	fail(p, LOL_ERR_SYNTAX, "Error: syntax error on line %zu", p->lx.line);
}
static int expect(struct parser* p, enum tok t) {
	if (p->status != LOL_OK) return 0;
	if (p->lx.tok != t) { syntax_error(p); return 0; }
	lex_next(&p->lx);
	return 1;
}
static int is_type(enum tok t) { return t >= T_AMBIENT && t <= T_SMOOTH_UNION; }
static int is_property(enum tok t) { return t == T_AMBIENT || (t >= T_SHININESS && t <= T_B); }

static void* grow(void* ptr, size_t* cap, size_t need, size_t elem) {
	if (need <= *cap) return ptr;
	size_t nc = *cap ? *cap * 2 : 16;
	while (nc < need) nc *= 2;
	void* np = realloc(ptr, nc * elem);
	if (np) *cap = nc;
	return np;
}

static int32_t scene_add_node(struct parser* p, const lol_node* n, size_t* cap) {
	lol_scene* s = p->scene;
	lol_node* nn = grow(s->nodes, cap, s->n_nodes + 1, sizeof *nn);
	if (!nn) { fail(p, LOL_ERR_NOMEM, "out of memory"); return -1; }
	s->nodes = nn;
	s->nodes[s->n_nodes] = *n;
	return (int32_t)s->n_nodes++;
}

/* type checks — the reference asserts (scene.c:69,81,87,93,99) */
static float want_num(struct parser* p, const struct def* d) {
	if (d->v.kind != V_NUM) { fail(p, LOL_ERR_TYPE, "Assertion failed: Is a number"); return 0.f; }
	return d->v.num;
}
static lol_v3 want_v3(struct parser* p, const struct def* d) {
	if (d->v.kind != V_LIST) { fail(p, LOL_ERR_TYPE, "Assertion failed: Is a list"); return (lol_v3){0, 0, 0}; }
	if (d->v.list_n != 3) { fail(p, LOL_ERR_TYPE, "Assertion failed: vec->size == 3"); return (lol_v3){0, 0, 0}; }
	return (lol_v3){ d->v.list[0], d->v.list[1], d->v.list[2] };
}
static uint32_t want_id(struct parser* p, const struct def* d) {
	if (d->v.kind != V_ID) { fail(p, LOL_ERR_TYPE, "Assertion failed: Is an ID"); return 0; }
	return d->v.id;
}
static int32_t want_obj(struct parser* p, const struct def* d) {
	if (d->v.kind != V_OBJ) { fail(p, LOL_ERR_TYPE, "Assertion failed: Is an object"); return -1; }
	return d->v.node;
}
static void unknown_prop(struct parser* p, const char* what) {
	fail(p, LOL_ERR_PROPERTY, "Unknown %s property", what);   /* scene.c:130-134 */
}

/* builders: zero-fill, then assign in list order, last one wins (scene.c:114-138) */

static lol_material build_material(struct parser* p, const struct deflist* dl) {
	lol_material m;
	memset(&m, 0, sizeof m);
	for (size_t i = 0; i < dl->n && p->status == LOL_OK; i++) {
		const struct def* d = &dl->d[i];
		switch (d->prop) {
		case T_SHININESS: m.shininess = want_num(p, d); break;
		case T_DIFFUSE:   m.diffuse   = want_v3(p, d);  break;
		case T_SPECULAR:  m.specular  = want_v3(p, d);  break;
		case T_AMBIENT:   m.ambient   = want_v3(p, d);  break;
		default: unknown_prop(p, "material");
		}
	}
	return m;
}

static void build_ambient(struct parser* p, const struct deflist* dl) {
	/* scene.c:149-164: the result is an uninitialised local unless `color` is
	 * present; this build keeps the previous ambient in that case. */
	for (size_t i = 0; i < dl->n && p->status == LOL_OK; i++) {
		const struct def* d = &dl->d[i];
		if (d->prop == T_COLOR) p->scene->ambient_color = want_v3(p, d);
		else unknown_prop(p, "ambient");
	}
}

static void build_camera(struct parser* p, const struct deflist* dl) {
	lol_camera c;
	memset(&c, 0, sizeof c);
	for (size_t i = 0; i < dl->n && p->status == LOL_OK; i++) {
		const struct def* d = &dl->d[i];
		switch (d->prop) {
		case T_POINT:     c.point     = want_v3(p, d);  break;
		case T_DIRECTION: c.direction = want_v3(p, d);  break;
		case T_FOV:       c.fov       = want_num(p, d); break;
		default: unknown_prop(p, "camera");
		}
	}
	/* scene.c:173-174: float/int → float; × double π; rounded to float on store */
	c.direction = lol_normalize(c.direction);
	c.fov = (float)((double)(c.fov / 180) * M_PI);
	p->scene->camera = c;
}

static lol_light build_light(struct parser* p, const struct deflist* dl) {
	lol_light l;
	memset(&l, 0, sizeof l);
	for (size_t i = 0; i < dl->n && p->status == LOL_OK; i++) {
		const struct def* d = &dl->d[i];
		switch (d->prop) {
		case T_POINT:              l.point              = want_v3(p, d); break;
		case T_DIFFUSE_INTENSITY:  l.diffuse_intensity  = want_v3(p, d); break;
		case T_SPECULAR_INTENSITY: l.specular_intensity = want_v3(p, d); break;
		default: unknown_prop(p, "light");
		}
	}
	return l;
}

static lol_node build_object(struct parser* p, enum comp type, const struct deflist* dl) {
	lol_node n;
	memset(&n, 0, sizeof n);
	n.a = n.b = -1;
	for (size_t i = 0; i < dl->n && p->status == LOL_OK; i++) {
		const struct def* d = &dl->d[i];
		switch (type) {
		case C_SPHERE:                       /* scene.c:185-193 */
			n.type = LOL_NODE_SPHERE;
			switch (d->prop) {
			case T_POINT:    n.point    = want_v3(p, d);  break;
			case T_MATERIAL: n.material = want_id(p, d);  break;
			case T_RADIUS:   n.radius   = want_num(p, d); break;
			default: unknown_prop(p, "sphere");
			}
			break;
		case C_BOX:                          /* scene.c:195-204 */
			n.type = LOL_NODE_BOX;
			switch (d->prop) {
			case T_POINT:    n.point       = want_v3(p, d);  break;
			case T_MATERIAL: n.material    = want_id(p, d);  break;
			case T_POINT2:   n.half_extent = want_v3(p, d);  break;
			case T_RADIUS:   n.radius      = want_num(p, d); break;
			default: unknown_prop(p, "box");
			}
			break;
		case C_PLANE:                        /* scene.c:207-216 */
			n.type = LOL_NODE_PLANE;
			switch (d->prop) {
			case T_MATERIAL: n.material = want_id(p, d);  break;
			case T_Y:        n.point.y  = want_num(p, d); break;
			default: unknown_prop(p, "plane");
			}
			break;
		case C_SMOOTH_UNION:                 /* scene.c:218-227 */
			n.type = LOL_NODE_SMOOTH_UNION;
			switch (d->prop) {
			case T_MATERIAL:   n.material   = want_id(p, d);  break;
			case T_SMOOTHNESS: n.smoothness = want_num(p, d); break;
			case T_A:          n.a          = want_obj(p, d); break;
			case T_B:          n.b          = want_obj(p, d); break;
			default: unknown_prop(p, "smooth_union");
			}
			break;
		default: break;
		}
	}
	switch (type) {
	case C_SPHERE:       n.type = LOL_NODE_SPHERE; break;
	case C_BOX:          n.type = LOL_NODE_BOX; break;
	case C_PLANE:        n.type = LOL_NODE_PLANE; break;
	case C_SMOOTH_UNION: n.type = LOL_NODE_SMOOTH_UNION; break;
	default: break;
	}
	return n;
}

struct caps { size_t nodes, roots, lights, materials; };

static void parse_deflist(struct parser* p, struct deflist* dl, struct caps* caps);

/* value: NUM | '(' NUM {',' NUM} ')' | ID | type '{' defs '}'   (scene-parser.y:127-145) */
static void parse_value(struct parser* p, struct value* v, struct caps* caps) {
	memset(v, 0, sizeof *v);
	v->node = -1;
	enum tok t = p->lx.tok;
	if (t == T_NUM) {
		v->kind = V_NUM;
		v->num = p->lx.val.num;
		lex_next(&p->lx);
	} else if (t == T_ID) {
		v->kind = V_ID;
		v->id = (uint32_t)p->lx.val.id;
		lex_next(&p->lx);
	} else if (t == T_LPAREN) {
		v->kind = V_LIST;
		lex_next(&p->lx);
		for (;;) {
			if (p->lx.tok != T_NUM) { syntax_error(p); return; }
			if (v->list_n < 8) v->list[v->list_n] = p->lx.val.num;
			v->list_n++;
			lex_next(&p->lx);
			if (p->lx.tok == T_COMMA) { lex_next(&p->lx); continue; }
			break;
		}
		expect(p, T_RPAREN);
	} else if (is_type(t)) {
		enum comp type = (enum comp)(t - T_AMBIENT);
		lex_next(&p->lx);
		if (!expect(p, T_LBRACE)) return;
		if (p->depth >= LOL_PARSE_MAX_DEPTH) {
			fail(p, LOL_ERR_UNSUPPORTED, "objects nested deeper than %d levels", LOL_PARSE_MAX_DEPTH);
			return;
		}
		struct deflist dl = { 0, 0, 0 };
		p->depth++;
		parse_deflist(p, &dl, caps);
		p->depth--;
		if (p->status == LOL_OK) expect(p, T_RBRACE);
		if (p->status == LOL_OK) {
			/* object_from_definition_list, scene.c:266-281 */
			if (type < C_SPHERE) {
				fail(p, LOL_ERR_COMPONENT, "Unknown scene object");
			} else {
				lol_node n = build_object(p, type, &dl);
				if (p->status == LOL_OK) {
					v->kind = V_OBJ;
					v->node = scene_add_node(p, &n, &caps->nodes);
				}
			}
		}
		free(dl.d);
	} else {
		syntax_error(p);
	}
}

/* defs: property '=' value {',' property '=' value}   (scene-parser.y:116-125) */
static void parse_deflist(struct parser* p, struct deflist* dl, struct caps* caps) {
	for (;;) {
		if (p->status != LOL_OK) return;
		enum tok prop = p->lx.tok;
		if (!is_property(prop)) { syntax_error(p); return; }
		lex_next(&p->lx);
		if (!expect(p, T_EQUALS)) return;
		struct def* nd = grow(dl->d, &dl->cap, dl->n + 1, sizeof *nd);
		if (!nd) { fail(p, LOL_ERR_NOMEM, "out of memory"); return; }
		dl->d = nd;
		dl->d[dl->n].prop = prop;
		parse_value(p, &dl->d[dl->n].v, caps);
		if (p->status != LOL_OK) return;
		dl->n++;
		if (p->lx.tok == T_COMMA) { lex_next(&p->lx); continue; }
		return;
	}
}

/* scene_add_component_from_definition_list, scene.c:229-264 */
static void add_component(struct parser* p, enum comp type, const struct deflist* dl, struct caps* caps) {
	lol_scene* s = p->scene;
	switch (type) {
	case C_AMBIENT: build_ambient(p, dl); break;
	case C_CAMERA:  build_camera(p, dl);  break;
	case C_POINT_LIGHT: {
		lol_light l = build_light(p, dl);
		if (p->status != LOL_OK) return;
		lol_light* nl = grow(s->lights, &caps->lights, s->n_lights + 1, sizeof *nl);
		if (!nl) { fail(p, LOL_ERR_NOMEM, "out of memory"); return; }
		s->lights = nl;
		s->lights[s->n_lights++] = l;
		break;
	}
	default: {
		lol_node n = build_object(p, type, dl);
		if (p->status != LOL_OK) return;
		int32_t idx = scene_add_node(p, &n, &caps->nodes);
		if (idx < 0) return;
		int32_t* nr = grow(s->roots, &caps->roots, s->n_roots + 1, sizeof *nr);
		if (!nr) { fail(p, LOL_ERR_NOMEM, "out of memory"); return; }
		s->roots = nr;
		s->roots[s->n_roots++] = idx;
	}
	}
}

/* input: materials scene   (scene-parser.y:73-114) */
static void parse_input(struct parser* p) {
	struct caps caps = { 0, 0, 0, 0 };
	lol_scene* s = p->scene;

	lex_next(&p->lx);
	if (!expect(p, T_MATERIALS) || !expect(p, T_LBRACE)) return;
	for (;;) {
		if (!expect(p, T_LBRACE)) return;
		struct deflist dl = { 0, 0, 0 };
		parse_deflist(p, &dl, &caps);
		if (p->status == LOL_OK) expect(p, T_RBRACE);
		if (p->status == LOL_OK) {
			lol_material m = build_material(p, &dl);
			lol_material* nm = grow(s->materials, &caps.materials, s->n_materials + 1, sizeof *nm);
			if (!nm) fail(p, LOL_ERR_NOMEM, "out of memory");
			else { s->materials = nm; s->materials[s->n_materials++] = m; }
		}
		free(dl.d);
		if (p->status != LOL_OK) return;
		if (p->lx.tok == T_COMMA) { lex_next(&p->lx); continue; }
		break;
	}
	if (!expect(p, T_RBRACE)) return;

	if (!expect(p, T_SCENE) || !expect(p, T_LBRACE)) return;
	for (;;) {
		if (p->status != LOL_OK) return;
		if (!is_type(p->lx.tok)) { syntax_error(p); return; }
		enum comp type = (enum comp)(p->lx.tok - T_AMBIENT);
		lex_next(&p->lx);
		if (!expect(p, T_LBRACE)) return;
		struct deflist dl = { 0, 0, 0 };
		parse_deflist(p, &dl, &caps);
		if (p->status == LOL_OK) expect(p, T_RBRACE);
		if (p->status == LOL_OK) add_component(p, type, &dl, &caps);
		free(dl.d);
		if (p->status != LOL_OK) return;
		if (p->lx.tok == T_COMMA) { lex_next(&p->lx); continue; }
		break;
	}
	if (!expect(p, T_RBRACE)) return;
	if (p->lx.tok != T_EOF) syntax_error(p);
}

/* ---------------------------------------------------------------- public */

lol_scene* lol_scene_new(void) {
	lol_scene* s = calloc(1, sizeof *s);
	if (!s) return NULL;
	s->camera.direction = (lol_v3){ 0, 0, 1 };
	s->camera.fov = (float)(M_PI / 2);         /* scene.c:54 */
	return s;
}

void lol_scene_free(lol_scene* s) {
	if (!s) return;
	free(s->materials);
	free(s->lights);
	free(s->nodes);
	free(s->roots);
	free(s);
}

int lol_scene_parse_string(const char* text, size_t len, lol_scene** out, char* errbuf, size_t errcap) {
	if (out) *out = NULL;
	if (errbuf && errcap) errbuf[0] = 0;
	if (!text || !out) return LOL_ERR_IO;
	struct parser p;
	memset(&p, 0, sizeof p);
	p.lx.s = text;
	p.lx.len = len;
	p.lx.line = 1;
	p.err = errbuf;
	p.errcap = errcap;
	p.scene = lol_scene_new();
	if (!p.scene) return LOL_ERR_NOMEM;
	parse_input(&p);
	if (p.status != LOL_OK) { lol_scene_free(p.scene); return p.status; }
	*out = p.scene;
	return LOL_OK;
}

int lol_scene_parse_file(const char* path, lol_scene** out, char* errbuf, size_t errcap) {
	if (out) *out = NULL;
	if (errbuf && errcap) errbuf[0] = 0;
	FILE* f = path ? fopen(path, "rb") : NULL;
	if (!f) {
		if (errbuf && errcap) snprintf(errbuf, errcap, "cannot open scene file '%s'", path ? path : "(null)");
		return LOL_ERR_IO;
	}
	size_t cap = 1 << 16, len = 0;
	char* buf = malloc(cap);
	while (buf) {
		size_t got = fread(buf + len, 1, cap - len, f);
		len += got;
		if (got == 0) break;
		if (len == cap) {
			char* nb = realloc(buf, cap *= 2);
			if (!nb) { free(buf); buf = NULL; }
			else buf = nb;
		}
	}
	fclose(f);
	if (!buf) return LOL_ERR_NOMEM;
	int st = lol_scene_parse_string(buf, len, out, errbuf, errcap);
	free(buf);
	return st;
}

int lol_scene_validate_materials(const lol_scene* s) {
	for (size_t i = 0; i < s->n_roots; i++)
		if (s->nodes[s->roots[i]].material >= s->n_materials) return 0;
	return 1;
}

/* --------------------------------------------------------------- flatten */

/* Sethi-Ullman number of every node reachable from `idx`, memoised in need[] (0 = not yet computed): the operand
 * stack depth its subtree needs when the deeper child is emitted first.  -1: bad index or deeper than 4096 levels. */
static int su_need(const lol_scene* s, int32_t idx, int depth, int16_t* need) {
	if (idx < 0 || (size_t)idx >= s->n_nodes || depth > 4096) return -1;
	if (need[idx]) return need[idx];
	const lol_node* n = &s->nodes[idx];
	int v = 1;
	if (n->type == LOL_NODE_SMOOTH_UNION) {
		int l = su_need(s, n->a, depth + 1, need), r = su_need(s, n->b, depth + 1, need);
		if (l < 0 || r < 0) return -1;
		v = l == r ? l + 1 : (l > r ? l : r);
	}
	need[idx] = (int16_t)v;
	return v;
}

/* the next free op of a program whose `ops` table grows by doubling (cap = *ops_cap) */
static lol_op* next_op(lol_program* out, size_t* ops_cap, int* status) {
	if (out->n_ops >= LOL_MAX_OPS) { *status = LOL_ERR_UNSUPPORTED; return NULL; }
	if (out->n_ops == *ops_cap) {
		const size_t nc = *ops_cap ? *ops_cap * 2 : 64;
		lol_op* np = realloc(out->ops, nc * sizeof *np);
		if (!np) { *status = LOL_ERR_NOMEM; return NULL; }
		out->ops = np;
		*ops_cap = nc;
	}
	lol_op* op = &out->ops[out->n_ops++];
	memset(op, 0, sizeof *op);
	return op;
}

static int emit(const lol_scene* s, int32_t idx, lol_program* out, size_t* ops_cap, const int16_t* need) {
	const lol_node* n = &s->nodes[idx];
	const int swap = n->type == LOL_NODE_SMOOTH_UNION && need[n->b] > need[n->a];
	if (n->type == LOL_NODE_SMOOTH_UNION) {
		int st = emit(s, swap ? n->b : n->a, out, ops_cap, need);
		if (st != LOL_OK) return st;
		st = emit(s, swap ? n->a : n->b, out, ops_cap, need);
		if (st != LOL_OK) return st;
	}
	int status = LOL_OK;
	lol_op* op = next_op(out, ops_cap, &status);
	if (!op) return status;
	switch (n->type) {
	case LOL_NODE_SPHERE:
		op->op = LOL_OP_SPHERE;
		op->f[0] = n->point.x; op->f[1] = n->point.y; op->f[2] = n->point.z;
		op->f[3] = n->radius;
		break;
	case LOL_NODE_BOX:
		op->op = LOL_OP_RBOX;
		op->f[0] = n->point.x; op->f[1] = n->point.y; op->f[2] = n->point.z;
		op->f[3] = n->half_extent.x; op->f[4] = n->half_extent.y; op->f[5] = n->half_extent.z;
		op->f[6] = n->radius;
		break;
	case LOL_NODE_PLANE:
		op->op = LOL_OP_PLANE;
		op->f[0] = n->point.y;
		break;
	case LOL_NODE_SMOOTH_UNION:
		op->op = swap ? LOL_OP_SMIN_R : LOL_OP_SMIN;
		op->f[0] = n->smoothness;
		break;
	default:
		return LOL_ERR_UNSUPPORTED;
	}
	return LOL_OK;
}

void lol_program_free(lol_program* p) {
	if (!p) return;
	free(p->ops); free(p->lights); free(p->materials); free(p->root_material);
	memset(p, 0, sizeof *p);
}

int lol_scene_flatten(const lol_scene* s, lol_program* out) {
	memset(out, 0, sizeof *out);
	if (s->n_lights > LOL_MAX_LIGHTS || s->n_materials > LOL_MAX_MATERIALS ||
	    s->n_roots > LOL_MAX_OPS)
		return LOL_ERR_UNSUPPORTED;
	if (s->n_materials == 0 || !lol_scene_validate_materials(s))
		return LOL_ERR_MATERIAL;   /* material #0 is the miss material (naive_renderer.c:103-112) */

	/* (a table of no entries is still a non-NULL allocation: consumers need not special-case it) */
	out->lights = malloc((s->n_lights ? s->n_lights : 1) * sizeof *out->lights);
	out->materials = malloc(s->n_materials * sizeof *out->materials);
	out->root_material = malloc((s->n_roots ? s->n_roots : 1) * sizeof *out->root_material);
	int16_t* memo = calloc(s->n_nodes ? s->n_nodes : 1, sizeof *memo);
	size_t ops_cap = 0;
	int status = out->lights && out->materials && out->root_material && memo ? LOL_OK : LOL_ERR_NOMEM;
	if (status == LOL_OK) {
		out->n_lights = (uint32_t)s->n_lights;
		out->n_materials = (uint32_t)s->n_materials;
		out->n_roots = (uint32_t)s->n_roots;
		out->ambient_color = s->ambient_color;
		if (s->n_lights) memcpy(out->lights, s->lights, s->n_lights * sizeof *s->lights);
		memcpy(out->materials, s->materials, s->n_materials * sizeof *s->materials);
	}
	for (size_t i = 0; i < s->n_roots && status == LOL_OK; i++) {
		int need = su_need(s, s->roots[i], 0, memo);
		if (need < 0 || need > LOL_MAX_STACK) { status = LOL_ERR_UNSUPPORTED; break; }
		if ((uint32_t)need > out->max_stack) out->max_stack = (uint32_t)need;
		status = emit(s, s->roots[i], out, &ops_cap, memo);
		if (status != LOL_OK) break;
		lol_op* top = next_op(out, &ops_cap, &status);
		if (!top) break;
		top->op = LOL_OP_TOP;
		top->id = (uint32_t)(i + 1);
		out->root_material[i] = s->nodes[s->roots[i]].material;
	}
	free(memo);
	if (status == LOL_OK && !out->ops) {            /* a scene without objects: an empty, non-NULL table */
		out->ops = malloc(sizeof *out->ops);
		if (!out->ops) status = LOL_ERR_NOMEM;
	}
	if (status != LOL_OK) lol_program_free(out);
	return status;
}

/* ---------------------------------------------------------------- camera */

void lol_frame_camera_init(lol_frame_camera* fc, const lol_camera* cam, int w, int h) {
	/* naive_renderer.c:181-186,213 */
	const lol_v3 up_guide = { 0.f, 1.f, 0.f };
	float aspect = (float)w / (float)h;
	float half_fov = cam->fov / 2.f;
	fc->origin = cam->point;
	fc->dir = cam->direction;
	fc->height = atanf(half_fov);
	fc->width = aspect * fc->height;
	fc->right = lol_normalize(lol_cross(cam->direction, up_guide));
	fc->up = lol_cross(fc->right, cam->direction);
}

const char* lol_status_str(int st) {
	switch (st) {
	case LOL_OK:              return "ok";
	case LOL_ERR_IO:          return "cannot read scene";
	case LOL_ERR_SYNTAX:      return "syntax error";
	case LOL_ERR_PROPERTY:    return "unknown property";
	case LOL_ERR_TYPE:        return "property value has the wrong kind";
	case LOL_ERR_COMPONENT:   return "unknown scene object";
	case LOL_ERR_MATERIAL:    return "material index out of range";
	case LOL_ERR_NOMEM:       return "out of memory";
	case LOL_ERR_UNSUPPORTED: return "scene exceeds renderer limits";
	default:                  return "unknown status";
	}
}
