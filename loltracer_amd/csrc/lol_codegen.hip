/*
 * lol_codegen.hip — from a flattened scene (lol_program) to what the kernels run: the exact-culling plan, the interpreter's
 * macro-op lists, and the scene's own kernel — generated HIP source compiled by hipRTC (the GPU counterpart of the reference's
 * tracing JIT, tracing_jit_renderer.dasc:76-216,416-434), cached in the process and on disk, refused when it shows LLVM's
 * long-branch register bug.  (Part of liblol_gpu.so; see lol_gpu_internal.h for how the library is cut.)
 */
#include "lol_gpu_internal.h"

/* lol_kernel.h's text, embedded at build time (csrc/Makefile: lol_kernel_src.inc) for hipRTC */
#include "lol_kernel_src.inc"

#pragma GCC visibility push(hidden)
/* ------------------------------------------------- scene → HIP source (the "JIT") */

std::string fbits(float v) {
	uint32_t u;
	memcpy(&u, &v, 4);
	char b[48];
	snprintf(b, sizeof b, "__builtin_bit_cast(float, 0x%08xu)", u);
	return b;
}

/* Which proven-exact shortcuts the generated code may use (see lol_kernel.h "fast exact paths"). */

/* ------------------------------------------------ exact culling of top-level objects
 * sdf() (naive_renderer.c:31-44) is a strict-'<' minimum over the top-level objects.  An object whose distance is
 * PROVABLY greater than the running minimum cannot change it, so its evaluation may be skipped — exactly, not
 * approximately.  The proof is a bounding sphere (C, R) per object, computed here in double precision:
 *   sphere(c, r):            value = |p-c| - r                                        → (c, max(r, 0))
 *   round box(c, b, r):      value >= |p-c| - |b| - r   (b >= 0, r >= 0)               → (c, |b| + r)
 *   smooth_union(a, b, k>0): value >= min(a, b) - k/4   (h(1-h) <= 1/4 on the clamped h) → sphere enclosing both + k/4
 *   plane, k <= 0, non-finite or absurdly large fields:                                 no bound — never skipped
 * so value(p) >= |p-C| - R in exact arithmetic.  The kernel's binary32 evaluation differs from that by a few ulps
 * of the magnitudes involved (<= 2^-19 relative to |p-C| + R, LABNOTES.md §3.6), which the test below swallows:
 *   R' = R (1 + 2^-10) + (|C|_max + 1) 2^-20, rounded up;   u = (best + R') (1 + 2^-12);
 *   skip  iff  u > 0  and  |p-C|^2 > u^2      (all in binary32; any NaN makes the comparisons false = no skip)
 * which implies |p-C| > (best + R')(1 + 2^-14), hence value(p) > best.  The decision is taken per WAVE: the
 * object is evaluated unless every lane that still cares about the result may skip it (lanes that may skip but
 * run anyway compute a value > best and change nothing).
 *
 * To have a running minimum to compare with, objects WITHOUT a bound (planes: one subtraction) are evaluated first
 * and the bounded ones after them, each group in file order.  The reference's tie rule — the FIRST object of
 * equal distance wins — is kept by comparing ids on ties wherever an object is evaluated after one that follows
 * it in the file:  t < best || (t == best && best_id > id)   (best_id = 0 only while best = +inf, where the
 * reference's inf < inf is false too). */
struct Sphere { bool ok; double c[3], r; uint32_t levels = 1; };   /* levels: nesting depth of the operations under it (a primitive is 1) */

struct RootBound {
	uint32_t first = 0, top = 0;        /* ops [first, top) compute the object, ops[top] is its LOL_OP_TOP */
	uint32_t id = 0, prims = 0;
	bool     bounded = false;
	double   c[3] = { 0, 0, 0 }, r = 0;
	uint32_t levels = 1;                /* nesting depth of its expression (sets the rounding slack of its test) */
	Sphere   sphere() const { return { bounded, { c[0], c[1], c[2] }, r, levels }; }
	std::vector<Sphere> clusters;       /* optional: two spheres that together bound the object more tightly (cluster_bounds) */
};

Sphere enclose(const Sphere& a, const Sphere& b) {
	const uint32_t levels = a.levels > b.levels ? a.levels : b.levels;
	if (!a.ok || !b.ok) return { false, { 0, 0, 0 }, 0, levels };
	const double dx = b.c[0] - a.c[0], dy = b.c[1] - a.c[1], dz = b.c[2] - a.c[2];
	const double d = sqrt(dx * dx + dy * dy + dz * dz);
	if (d + b.r <= a.r) { Sphere r = a; r.levels = levels; return r; }
	if (d + a.r <= b.r) { Sphere r = b; r.levels = levels; return r; }
	const double R = 0.5 * (d + a.r + b.r), t = d > 0 ? (R - a.r) / d : 0.0;
	return { true, { a.c[0] + dx * t, a.c[1] + dy * t, a.c[2] + dz * t }, R * (1.0 + 1e-12), levels };
}

void cluster_bounds(const lol_program& P, RootBound& R);

std::vector<RootBound> analyse_roots(const lol_program& P) {
	std::vector<RootBound> roots;
	std::vector<Sphere> st;
	auto sane = [](double v) { return v - v == 0.0 && fabs(v) < 1e15; };
	RootBound cur;
	for (uint32_t i = 0; i < P.n_ops; i++) {
		const lol_op& o = P.ops[i];
		switch (o.op) {
		case LOL_OP_SPHERE: {
			const bool ok = sane(o.f[0]) && sane(o.f[1]) && sane(o.f[2]) && sane(o.f[3]);
			st.push_back({ ok, { o.f[0], o.f[1], o.f[2] }, o.f[3] > 0 ? (double)o.f[3] : 0.0 });
			cur.prims++;
			break;
		}
		case LOL_OP_RBOX: {
			bool ok = true;
			for (int j = 0; j < 7; j++) ok = ok && sane(o.f[j]);
			ok = ok && o.f[3] >= 0 && o.f[4] >= 0 && o.f[5] >= 0 && o.f[6] >= 0;
			const double hb = sqrt((double)o.f[3] * o.f[3] + (double)o.f[4] * o.f[4] + (double)o.f[5] * o.f[5]);
			st.push_back({ ok, { o.f[0], o.f[1], o.f[2] }, hb * (1.0 + 1e-12) + o.f[6] });
			cur.prims++;
			break;
		}
		case LOL_OP_PLANE:
			st.push_back({ false, { 0, 0, 0 }, 0 });
			cur.prims++;
			break;
		case LOL_OP_SMIN: case LOL_OP_SMIN_R: {
			Sphere b = st.back(); st.pop_back();
			Sphere a = st.back(); st.pop_back();
			Sphere u = enclose(a, b);
			if (!(sane(o.f[0]) && o.f[0] > 0)) u.ok = false;
			u.r += 0.25 * (double)o.f[0];
			u.levels++;
			st.push_back(u);
			break;
		}
		case LOL_OP_TOP: {
			Sphere v = st.back(); st.pop_back();
			cur.top = i; cur.id = o.id;
			cur.bounded = v.ok && sane(v.r);
			cur.c[0] = v.c[0]; cur.c[1] = v.c[1]; cur.c[2] = v.c[2]; cur.r = v.r; cur.levels = v.levels;
			cluster_bounds(P, cur);
			roots.push_back(cur);
			cur = RootBound();
			cur.first = i + 1;
			break;
		}
		}
	}
	return roots;
}

/* Two spheres instead of one (round 3).  One sphere around a long or L-shaped union is mostly empty.  For a union tree with
 * every k > 0:  smooth_union(a, b, k) >= min(a, b) - k/4, so by induction  value(p) >= min over the LEAVES i of
 * (prim_i(p) - slack_i),  slack_i = the sum of k/4 over the unions above leaf i;  and prim_i(p) >= |p - c_i| - r_i for a sphere
 * (round box: r_i = |b| + r).  Split the leaves into two clusters and let sphere S_j enclose the spheres (c_i, r_i + slack_i) of
 * its cluster: then  value(p) >= min_j (|p - C_j| - R_j)  in exact arithmetic, and the object may be skipped where BOTH of the
 * usual tests pass (make_test: each with the rounding slack of the object's depth).  The split: along the widest axis of the
 * leaf centres, at the position that minimises R_A^3 + R_B^3; used when the larger of the two is at most 0.75 of the single
 * sphere's radius (scene4's blob: 8.6 and 7.8 against 11.1; a numpy model of C3 — tools/cull_model.py — puts the wave-evaluations
 * that may skip the blob at 35 % against 30 %). */
void cluster_bounds(const lol_program& P, RootBound& R) {
	R.clusters.clear();
	if (!R.bounded || R.prims < 3) return;
	std::vector<std::vector<Sphere>> st;
	for (uint32_t i = R.first; i < R.top; i++) {
		const lol_op& o = P.ops[i];
		if (o.op == LOL_OP_SPHERE) st.push_back({ { true, { o.f[0], o.f[1], o.f[2] }, o.f[3] > 0 ? (double)o.f[3] : 0.0, R.levels } });
		else if (o.op == LOL_OP_RBOX) {
			const double hb = sqrt((double)o.f[3] * o.f[3] + (double)o.f[4] * o.f[4] + (double)o.f[5] * o.f[5]);
			st.push_back({ { true, { o.f[0], o.f[1], o.f[2] }, hb * (1.0 + 1e-12) + o.f[6], R.levels } });
		} else if (o.op == LOL_OP_SMIN || o.op == LOL_OP_SMIN_R) {
			std::vector<Sphere> b = std::move(st.back()); st.pop_back();
			std::vector<Sphere>& a = st.back();
			a.insert(a.end(), b.begin(), b.end());
			for (Sphere& l : a) l.r += 0.25 * (double)o.f[0];          /* the slack of this union, for every leaf under it */
		} else return;                                                   /* (a plane: the object has no bound at all) */
	}
	if (st.size() != 1 || st[0].size() < 3 || st[0].size() > 4096) return;      /* (the cut search below is quadratic in the leaves) */
	std::vector<Sphere>& leaves = st[0];
	double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
	for (const Sphere& l : leaves) for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], l.c[a]); mx[a] = fmax(mx[a], l.c[a]); }
	int axis = 0;
	for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
	std::stable_sort(leaves.begin(), leaves.end(), [&](const Sphere& x, const Sphere& y) { return x.c[axis] < y.c[axis]; });
	auto hull = [&](size_t lo, size_t hi) { Sphere g = leaves[lo]; for (size_t k = lo + 1; k < hi; k++) g = enclose(g, leaves[k]); g.levels = R.levels; return g; };
	double best = 1e300; size_t cut = 0;
	for (size_t c = 1; c < leaves.size(); c++) {
		const Sphere a = hull(0, c), b = hull(c, leaves.size());
		const double cost = a.r * a.r * a.r + b.r * b.r * b.r;
		if (cost < best) { best = cost; cut = c; }
	}
	const Sphere a = hull(0, cut), b = hull(cut, leaves.size());
	auto sane = [](double v) { return v - v == 0.0 && fabs(v) < 1e15; };
	if (!a.ok || !b.ok || !sane(a.r) || !sane(b.r)) return;
	if (fmax(a.r, b.r) > 0.75 * R.r) return;
	R.clusters = { a, b };
}

struct CullTest { float c[3]; float rm; float k; };     /* skip iff u = (best + rm)*k > 0 and |p - c|^2 > u^2 */

/* The in-kernel test's constants from a double-precision bound: centre rounded to binary32 (its rounding error is
 * covered by the |C| 2^-20 term), radius and comparison inflated by the rounding the guarded expression can
 * accumulate.  With D = |p-C|: the exact-arithmetic value is >= D - R; one smooth minimum evaluated in binary32 adds
 * at most 2^-24 (4 M + 8.5 k) to the error of its operands (it is 1-Lipschitz in them), a primitive at most
 * 2^-24 * 4 M, with M <= D + R and k <= 4 R — so the binary32 value is >= D(1 - e) - R(1 + e), e = 40 * 2^-24 * levels.
 * The test gives D > (best + R')K(1 - 2^-21); with K >= 1 + 2e + 2^-19 and R' >= R(1 + 2(K-1) + 2e) that is
 * > best in both signs of best (for best < 0 use |best| < R').  Shallow objects (levels <= 50) keep the constants
 * of the first version, K = 1 + 2^-12 and R' = R(1 + 2^-10): a chain of 500 unions gets K = 1.0024. */
CullTest make_test(const Sphere& b) {
	CullTest t;
	double cmax = 0;
	for (int j = 0; j < 3; j++) { t.c[j] = (float)b.c[j]; cmax = fmax(cmax, fabs(b.c[j])); }
	const double e = 40.0 * 0x1p-24 * (double)b.levels;
	const double K = 1.0 + fmax(0x1p-12, 2.0 * e + 0x1p-19);
	t.k = (float)K;
	if ((double)t.k < K) t.k = nextafterf(t.k, INFINITY);
	const double rho = fmax(0x1p-10, 2.0 * ((double)t.k - 1.0) + 2.0 * e);
	const double rm = b.r * (1.0 + rho) + (cmax + 1.0) * 0x1p-20;
	t.rm = nextafterf((float)rm, INFINITY);
	return t;
}

std::vector<CullTest> cluster_tests(const RootBound& r) {
	std::vector<CullTest> t;
	for (const Sphere& c : r.clusters) t.push_back(make_test(c));
	return t;
}

/* A test guards a run of consecutive objects of the evaluation order: [begin, end) positions in `order`.  Runs nest
 * (the run of all bounded objects, inside it spatial clusters, inside those single heavy objects). */
/* `both`: when not empty the run (always a single object) is skipped where ALL of these pass — the object's two cluster
 * spheres — instead of the one test of its enclosing sphere; the interpreter keeps the one sphere (`test`). */
struct CullInterval { size_t begin, end; CullTest test; std::vector<CullTest> both; };

struct CullPlan {
	std::vector<uint32_t> order;          /* evaluation order: indices into the root list */
	size_t   n_unbounded = 0;             /* the first n_unbounded entries of `order` have no bound */
	std::vector<CullInterval> intervals;  /* outer before inner, by position */
	bool     group = false;               /* intervals[0] is the run of ALL bounded objects (what the interpreter carries) */
	CullTest group_test{};
};


/* Objects that lie together are evaluated together, behind a test of their common bounding sphere: a k-d split of
 * the bounded objects (median cut along the widest axis of their centres, down to runs of at most three) gives the
 * evaluation order, and every node of that tree gets a test — single objects too: 11 instructions against a sphere's
 * ~20 with its square root, measured faster on every scene tried (tools/flat_scene_ab.py: leaf sizes 2…8, tests from
 * 1…4 primitives up; profiles/r2_flat_scene_ab.jsonl).  A ray that is far from a whole cluster pays one test for it
 * instead of one evaluation per object — what makes a scene of hundreds of separate objects affordable. */
static void kd_build(const std::vector<RootBound>& roots, std::vector<uint32_t>& ids, size_t lo, size_t hi,
                     size_t base, bool has_predecessor, size_t leaf_max, uint32_t min_prims, CullPlan& plan) {
	const size_t count = hi - lo;
	Sphere g = roots[ids[lo]].sphere();
	uint32_t prims = roots[ids[lo]].prims;
	for (size_t k = lo + 1; k < hi; k++) {
		const RootBound& r = roots[ids[k]];
		g = enclose(g, r.sphere());
		prims += r.prims;
	}
	/* a test needs a running minimum to compare with (something evaluated before the run); a node that covers
	 * exactly what its parent covers adds nothing */
	const bool same_as_parent = !plan.intervals.empty() && plan.intervals.back().begin == base + lo && plan.intervals.back().end == base + hi;
	if ((has_predecessor || lo > 0) && prims >= min_prims && !same_as_parent)
		plan.intervals.push_back({ base + lo, base + hi, make_test(g), count == 1 ? cluster_tests(roots[ids[lo]]) : std::vector<CullTest>() });
	if (count <= leaf_max) {
		if (count > 1)                       /* inside a small run: the objects' own tests */
			for (size_t k = lo; k < hi; k++) {
				const RootBound& r = roots[ids[k]];
				if (r.prims >= min_prims && (has_predecessor || k > 0))
					plan.intervals.push_back({ base + k, base + k + 1, make_test(r.sphere()), cluster_tests(r) });
			}
		return;
	}
	double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
	for (size_t k = lo; k < hi; k++)
		for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], roots[ids[k]].c[a]); mx[a] = fmax(mx[a], roots[ids[k]].c[a]); }
	int axis = 0;
	for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
	const size_t mid = lo + count / 2;
	std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi,
	                 [&](uint32_t x, uint32_t y) { return roots[x].c[axis] < roots[y].c[axis] || (roots[x].c[axis] == roots[y].c[axis] && x < y); });
	kd_build(roots, ids, lo, mid, base, has_predecessor, leaf_max, min_prims, plan);
	kd_build(roots, ids, mid, hi, base, has_predecessor, leaf_max, min_prims, plan);
}

CullPlan plan_culling(const std::vector<RootBound>& roots, bool enabled) {
	CullPlan plan;
	std::vector<uint32_t> bounded;
	if (enabled)
		for (uint32_t i = 0; i < roots.size(); i++) (roots[i].bounded ? bounded : plan.order).push_back(i);
	else
		for (uint32_t i = 0; i < roots.size(); i++) plan.order.push_back(i);
	plan.n_unbounded = plan.order.size();
	if (!enabled || bounded.empty()) return plan;
	/* LOL_GPU_CULL_CLUSTERS=0: one run of all bounded objects in scene order, no spatial clusters (for A/B runs) */
	const char* e = tuning_env("LOL_GPU_CULL_CLUSTERS");
	const size_t leaf_max = (e && atoi(e) == 0) ? (size_t)-1 : (e && atoi(e) > 1 ? (size_t)atoi(e) : 3);
	const uint32_t min_prims = 1;      /* every node of the tree gets its test (profiles/r2_flat_scene_ab.jsonl: tests from 1 ... 4 primitives up) */
	kd_build(roots, bounded, 0, bounded.size(), plan.n_unbounded, plan.n_unbounded > 0, leaf_max, min_prims, plan);
	plan.order.insert(plan.order.end(), bounded.begin(), bounded.end());
	/* outer runs before inner ones at the same position (kd_build emits parents first; keep that order stable) */
	std::stable_sort(plan.intervals.begin(), plan.intervals.end(), [](const CullInterval& a, const CullInterval& b) {
		return a.begin < b.begin || (a.begin == b.begin && a.end > b.end);
	});
	if (!plan.intervals.empty() && plan.intervals[0].begin == plan.n_unbounded && plan.intervals[0].end == plan.order.size() &&
	    plan.n_unbounded > 0) {
		plan.group = true;
		plan.group_test = plan.intervals[0].test;
	}
	return plan;
}

/*
 * Post-order program → macro-ops of the interpreter (lol_kernel.h, Interp).  The accumulator is the top of the
 * post-order operand stack:
 *   primitive followed by SMIN / SMIN_R  → one macro-op: x = primitive, combined with acc at once.  In post-order
 *                                           SMIN pops b (top) then a; the primitive is the top, so x = b and the
 *                                           combine is sminf(acc, x); SMIN_R (top is a) gives sminf(x, acc);
 *   primitive otherwise                   → SET when nothing is on the stack yet, else PUSH (acc goes under);
 *   SMIN / SMIN_R after a non-primitive   → x = the popped entry under acc: SMIN has a = x, b = acc → sminf(x, acc);
 *                                           SMIN_R has a = acc, b = x → sminf(acc, x) — a POP record of its own, unless the
 *                                           record before it has a smooth min of the same, proven k: then it rides on that
 *                                           record (MOPB_POST: after the record's own combine);
 *   TOP                                   → a flag on the macro-op that produced the value (+ MOP_TIE where the
 *                                           object is evaluated after one that follows it in the file).
 * The objects come in the order of `plan` (unbounded ones first).  Every test of the plan becomes a constants
 * record in front of the first object of its run (outer runs first); the macro-op that finishes the object before
 * it gets MOPB_CULL_NEXT (the run of all bounded objects) and / or MOPB_CULL_CHAIN (inner runs), and the
 * CULLC_NEXT / CULLC_AFTER flags chain test records that follow one another directly.
 * `fast` lists the smoothness constants whose fast blend factor was proven on the device; allow_nofixup: this list may use the
 * form without v_div_fixup where that was proven too (the caller builds both lists: same records, other smooth-min bits).
 */
std::vector<uint32_t> build_mops(const lol_program& P, const FastPaths* fast, const std::vector<RootBound>& roots,
                                 const CullPlan& plan, bool allow_nofixup) {
	std::vector<uint32_t> out;
	auto fbits32 = [](float v) { uint32_t u; memcpy(&u, &v, 4); return u; };
	auto smin_fields = [&](uint32_t* m, const lol_op& sm) {
		m[9] = fbits32(sm.f[0]);
		if (fast && fast->has(sm.f[0])) {
			m[0] |= lol::MOP_FASTDIV | (allow_nofixup && fast->has_nf(sm.f[0]) ? lol::MOP_NOFIXUP : 0u);
			m[10] = fbits32(2.0f * sm.f[0]);
			m[11] = fbits32(0.5f * (1.0f / sm.f[0]));
		}
		m[0] |= lol::mop_smin_bits(m[0]);
	};
	/* (every test of the plan is kept: leaving out those of runs with few primitives was measured slower here too — a test is one
	 * turn of a scalar loop inside the rare TAIL branch) */
	/* deep: operand stacks beyond the 4-bit slot fields — slots travel in words of their own and pops are not fused (lol_kernel.h, MOP_DEEP_FROM) */
	const bool deep = interp_stack_class(P.max_stack) == lol::MOP_DEEP_SLOTS;
	const bool fuse_pops = !deep && !(tuning_env("LOL_GPU_INTERP_FUSE_POPS") && tuning_env("LOL_GPU_INTERP_FUSE_POPS")[0] == '0');     /* A/B switch */
	const std::vector<CullInterval>& ivs = plan.intervals;
	const bool group_first = plan.group && !ivs.empty() && ivs[0].begin == plan.n_unbounded && ivs[0].end == plan.order.size();
	std::vector<size_t> at(ivs.size());              /* where each test's constants record went */
	std::vector<uint32_t> begins(plan.order.size() + 1, 0);
	for (const CullInterval& iv : ivs) begins[iv.begin]++;
	std::vector<std::vector<size_t>> ends_at(plan.order.size() + 1);      /* the runs that end after object oi - 1 (linear, not a scan per object) */
	for (size_t k = 0; k < ivs.size(); k++) ends_at[ivs[k].end].push_back(k);
	uint32_t max_id_seen = 0;
	size_t next_iv = 0;
	for (size_t oi = 0; oi < plan.order.size(); oi++) {
		const RootBound& R = roots[plan.order[oi]];
		for (uint32_t n = 0; n < begins[oi]; n++, next_iv++) {             /* (never at oi == 0: plan_culling) */
			const CullInterval& iv = ivs[next_iv];
			uint32_t c[lol::MOP_DWORDS] = { 0 };
			c[0] = (n + 1 < begins[oi] ? lol::CULLC_NEXT : 0u) | (begins[iv.end] ? lol::CULLC_AFTER : 0u);
			for (int j = 0; j < 3; j++) c[2 + j] = fbits32(iv.test.c[j]);
			c[5] = fbits32(iv.test.rm);
			c[6] = fbits32(iv.test.k);
			at[next_iv] = out.size();
			out.insert(out.end(), c, c + lol::MOP_DWORDS);
		}
		int depth = 0;                                   /* post-order stack depth before the current op */
		bool emitted = false;                            /* this object has a record yet (`last` is one of its own) */
		size_t last = 0;                                 /* start of the macro-op that produced the current acc */
		for (uint32_t i = R.first; i < R.top; i++) {
			const lol_op& o = P.ops[i];
			uint32_t m[lol::MOP_DWORDS] = { 0 };
			if (o.op <= LOL_OP_PLANE) {
				const uint32_t kind = o.op == LOL_OP_SPHERE ? lol::MOP_SPHERE : o.op == LOL_OP_RBOX ? lol::MOP_RBOX : lol::MOP_PLANE;
				for (int j = 0; j < 7; j++) m[2 + j] = fbits32(o.f[j]);
				const lol_op* nx = i + 1 < R.top ? &P.ops[i + 1] : nullptr;
				if (nx && (nx->op == LOL_OP_SMIN || nx->op == LOL_OP_SMIN_R) && depth >= 1) {
					m[0] = lol::mop_header(kind, nx->op == LOL_OP_SMIN ? lol::MOP_SMIN : lol::MOP_SMIN_X);
					smin_fields(m, *nx);
					i++;                                /* the smooth min is part of this macro-op; depth unchanged */
				} else {
					m[0] = lol::mop_header(kind, depth == 0 ? lol::MOP_SET : lol::MOP_PUSH);
					if (depth > 0) {                                                       /* the accumulator goes to this slot */
						if (deep) m[9] = (uint32_t)(depth - 1);
						else m[0] |= (uint32_t)(depth - 1) << lol::MOP_SLOT_SHIFT;
					}
					depth++;
				}
			} else {                                     /* SMIN / SMIN_R on two computed operands */
				/* ... rides on the record that has just finished the second operand when that record's own smooth min has the
				 * same, proven k (the record's k words serve both; lol_kernel.h, MOPB_POST): one record less to fetch and dispatch */
				if (fuse_pops && emitted && (out[last] & lol::MOPB_SMIN) && (out[last] & lol::MOP_FASTDIV) && !(out[last] & lol::MOPB_POST) &&
				    out[last + 9] == fbits32(o.f[0])) {
					out[last] |= lol::MOPB_POST | lol::MOPB_STACK | lol::MOPB_TAIL | (o.op == LOL_OP_SMIN ? lol::MOPB_POST_YA : 0u) |
					             (uint32_t)(depth - 2) << lol::MOP_POST_SLOT_SHIFT;
					depth--;
					continue;
				}
				m[0] = lol::mop_header(lol::MOP_POP, o.op == LOL_OP_SMIN ? lol::MOP_SMIN_X : lol::MOP_SMIN);
				if (deep) m[2] = (uint32_t)(depth - 2);                                         /* the operand under the accumulator */
				else m[0] |= (uint32_t)(depth - 2) << lol::MOP_SLOT_SHIFT;
				smin_fields(m, o);
				depth--;
			}
			last = out.size();
			emitted = true;
			out.insert(out.end(), m, m + lol::MOP_DWORDS);
		}
		out[last] |= lol::MOP_TOP | lol::MOPB_TAIL | (R.id < max_id_seen ? lol::MOP_TIE : 0u);
		out[last + 1] = R.id;
		if (R.id > max_id_seen) max_id_seen = R.id;
		if (begins[oi + 1]) {
			const bool group_here = group_first && oi + 1 == plan.n_unbounded;
			if (group_here) out[last] |= lol::MOPB_CULL_NEXT;
			if (begins[oi + 1] > (group_here ? 1u : 0u)) out[last] |= lol::MOPB_CULL_CHAIN;
		}
		for (size_t k : ends_at[oi + 1])                                   /* every run that ends here: how far its test jumps */
			out[at[k] + 1] = (uint32_t)((out.size() - at[k]) / lol::MOP_DWORDS - 1);
	}
	return out;
}

bool build_interp_lists(const lol_program& P, const FastPaths& fast, bool cull, std::vector<uint32_t>& lists, uint32_t& n_mops) {
	const std::vector<RootBound> roots = analyse_roots(P);
	const CullPlan plan = plan_culling(roots, cull);
	lists = build_mops(P, &fast, roots, plan, false);
	n_mops = (uint32_t)(lists.size() / lol::MOP_DWORDS);
	const std::vector<uint32_t> nofix = build_mops(P, &fast, roots, plan, true);
	if (nofix.size() != lists.size()) return false;
	lists.insert(lists.end(), nofix.begin(), nofix.end());
	return true;
}

/* Emits one `struct <name>` with eval(): one SSA temporary per op, same operation order WITHIN every top-level object
 * as the post-order program; the objects themselves in the order of `plan` (file order when culling is off).
 * out_of_line: the body becomes ONE real function (`<name>_fn`, __noinline__) that the march, normal and shadow
 * loops call, instead of being inlined into each of them — for large scenes, whose straight-line SDF would
 * otherwise be replicated six times (three loops x fast / exact) and outgrow the instruction cache. */
void emit_sdf(std::string& s, const lol_program& P, const char* name, const FastPaths* fast, bool out_of_line,
              const std::vector<RootBound>& roots, const CullPlan& plan, const std::string& occupancy) {
	char line[768];
	const int fsqrt = fast ? fast->sqrt_kind : 0;
	char fs[32] = "";
	if (fsqrt) snprintf(fs, sizeof fs, "_fast<%d>", fsqrt);
	/* only the outermost test keeps a cool-down counter (wave-uniform state in the Sdf struct) */
	char cool_decl[64] = "";
	if (!plan.intervals.empty()) snprintf(cool_decl, sizeof cool_decl, "\tu32 cool[1] = {};\n");
	if (out_of_line) {
		/* out of line the cool-down state is per call (always 0: every evaluation tests) */
		/* (amdgpu_waves_per_eu applies to kernels only: the function is scheduled with the default register budget) */
		(void)occupancy;
		snprintf(line, sizeof line, "__device__ __noinline__ SdfOut %s_fn(float px, float py, float pz, u32 rg_lo, u32 rg_hi) {\n"
		         "\t\tconst V3 p = { px, py, pz };\n\t\tRange rg; rg.lo = rg_lo; rg.hi = rg_hi;\n\t\tfloat nanacc = 0.f;\n\t\tfloat best; u32 best_id;\n\t%s",
		         name, cool_decl);
		s += line;
	} else {
		/* ASSUME_SETTLED: the fast pipeline only runs under FLAG_SHADOW_SETTLED (generate_source; lol_kernel.h, soft_shadow).
		 * loop_done(): what the wave-uniform cool-down counter is after a loop that lanes leave one by one (lol_kernel.h, Interp) */
		/* ASK_ID_ONCE: the primary march evaluates the distance alone and asks for the id of its last step once (lol_kernel.h,
		 * march) — from three top-level objects on, where a v_min per object and step outweighs the one evaluation more per pixel
		 * (measured: scene.lol's four objects +3.6 %, scene4's two -0.8 %; profiles/r6_ab_id_asked_once.txt) */
		snprintf(line, sizeof line, "struct %s {\n\tstatic constexpr bool ASSUME_SETTLED = %s;\n\tstatic constexpr bool ASK_ID_ONCE = %s;\n\tRange rg;\n\tfloat nanacc = 0.f;\n%s"
		         "\t__device__ __forceinline__ void loop_done() { %s }\n", name, fast ? "true" : "false", P.n_roots >= 3 ? "true" : "false", cool_decl,
		         plan.intervals.empty() ? "" : "cool[0] = 0u;");
		s += line;
	}
	/* The body twice where it is inlined: eval() — distance and object id — for the primary march and the normal taps, and
	 * eval_dist() — the distance alone — for the shadow marches, which never look at the id.  There an object joins the running
	 * minimum with ONE v_min_f32 instead of the compare(s) and selects of the strict-'<' / lower-id-wins rule: the two agree on the
	 * value except for the sign of a zero (two objects at distance -0 and +0 of the same point), and a shadow march's results —
	 * its factor maxf(res, 0) and its step count — are the same for s = -0 and s = +0: t + s, 50 s / t compared with 0, and
	 * maxf(+-0, 0) = +0 all are (lol_kernel.h, soft_shadow).  A NaN value is dropped by either form. */
	for (int pass = 0; pass < (out_of_line ? 1 : 2); pass++) {
	const bool dist_only = pass == 1;
	if (!out_of_line)
		s += dist_only ? "\t__device__ __forceinline__ void eval_dist(V3 p, float& best) {\n"
		               : "\t__device__ __forceinline__ void eval(V3 p, float& best, u32& best_id) {\n";
	s += dist_only ? "\t\tbest = __builtin_inff();\n" : "\t\tbest = __builtin_inff(); best_id = 0u;\n";
	int t = 0, n_tests = 0;
	/* After a test that did not allow the skip, the next `cooldown` evaluations of this SDF object do not test
	 * again (a ray that is near the object now is near it on its next steps too): the test costs 11 VALU
	 * instructions, and where it keeps failing that is pure overhead.  Never testing is always allowed — the
	 * test only ever permits a skip — so this changes no result.  `cool` lives in the Sdf struct, wave-uniform. */
	const int cooldown = 3;
	/* spheres of radius >= 2^-20 carry no range tracker where the device has shown the fast root harmless below its proven domain
	 * (lol_kernel.h, sd_sphere_fast_nr): a NaN reaches the object's value instead, which is looked at once */
	const bool nan_flag = fast && fast->sqrt_tiny_ok;
	bool object_has_nr = false;
	/* LOL_GPU_SAT_CULL_MIN_PRIMS: `a` operands of a smooth union with at least this many primitives get a saturation-
	 * culling test (see emit_node below); 0 = none.  Measured (tools/tree_scene_ab.py, balanced trees of spheres at
	 * 1080p, profiles/r2_tree_scene_ab.jsonl): the 17-instruction test pays from a few dozen primitives — 128 spheres
	 * 232 -> 405 Mpixels/s, 256 spheres 127 -> 166 at 32 (375 / 157 at 16); with a test on every operand scene4
	 * loses 14 %, a 32-sphere tree 30 %. */
	int sat_cull_min_prims = 32;
	if (const char* e = tuning_env("LOL_GPU_SAT_CULL_MIN_PRIMS")) sat_cull_min_prims = atoi(e);
	/* one test = one bounding sphere; a run guarded by several (an object's two cluster spheres) is skipped where ALL pass */
	auto open_test = [&](const CullInterval& iv, bool with_cooldown) {
		const std::vector<CullTest> one = { iv.test };
		const std::vector<CullTest>& tests = iv.both.empty() ? one : iv.both;
		std::string decl, votes;
		const int k0 = n_tests;
		for (const CullTest& ct : tests) {
			const int k = n_tests++;
			snprintf(line, sizeof line,
			         "\t\t  const float cx%d = p.x - %s, cy%d = p.y - %s, cz%d = p.z - %s;\n"
			         "\t\t  const float cl%d = (cx%d * cx%d + cy%d * cy%d) + cz%d * cz%d;\n"
			         "\t\t  const float cu%d = (best + %s) * %s;\n",
			         k, fbits(ct.c[0]).c_str(), k, fbits(ct.c[1]).c_str(), k, fbits(ct.c[2]).c_str(),
			         k, k, k, k, k, k, k, k, fbits(ct.rm).c_str(), fbits(ct.k).c_str());
			decl += line;
			snprintf(line, sizeof line, "%svote(!(cl%d > cu%d * cu%d)) | vote(!(cu%d > 0.f))", votes.empty() ? "" : " | ", k, k, k, k);
			votes += line;
		}
		if (with_cooldown) {
			snprintf(line, sizeof line, "\t\t{ bool need%d = true;\n\t\t  if (cool[0] == 0u) {\n", k0);
			s += line;
			s += decl;
			snprintf(line, sizeof line, "\t\t  need%d = ((", k0);
			s += line;
			s += votes;
			snprintf(line, sizeof line, ")) != 0;\n\t\t  if (need%d) cool[0] = %du;\n\t\t  } else cool[0]--;\n\t\t  if (need%d) {\n", k0, cooldown, k0);
			s += line;
		} else {
			s += "\t\t{\n";
			s += decl;
			s += "\t\t  if (((";
			s += votes;
			s += ")) != 0) {\n";
		}
	};
	uint32_t max_id_seen = 0;
	size_t next_iv = 0;
	std::vector<uint32_t> runs_ending(plan.order.size() + 1, 0);          /* how many runs end after object oi - 1 */
	for (const CullInterval& iv : plan.intervals) runs_ending[iv.end]++;
	for (size_t oi = 0; oi < plan.order.size(); oi++) {
		const RootBound& R = roots[plan.order[oi]];
		while (next_iv < plan.intervals.size() && plan.intervals[next_iv].begin == oi) {      /* outer runs first */
			open_test(plan.intervals[next_iv], next_iv == 0);
			next_iv++;
		}
		/* the object's expression tree from its post-order ops (child `a` / `b` = the operands of sminf(a, b, k)) */
		struct Node { uint32_t op; int a, b; Sphere bound; uint32_t prims; bool fon = false; };      /* fon: the fast SDF's value of this node is finite or NaN */
		std::vector<Node> nodes;
		{
			std::vector<int> st;
			auto sane = [](double v) { return v - v == 0.0 && fabs(v) < 1e15; };
			for (uint32_t i = R.first; i < R.top; i++) {
				const lol_op& o = P.ops[i];
				Node n{ i, -1, -1, { false, { 0, 0, 0 }, 0, 1 }, 1 };
				if (o.op == LOL_OP_SPHERE) {
					n.fon = fsqrt && nan_flag && o.f[3] >= 0x1p-20f && std::isfinite(o.f[3]);      /* sd_sphere_fast_nr (emit_node) */
					n.bound = { sane(o.f[0]) && sane(o.f[1]) && sane(o.f[2]) && sane(o.f[3]), { o.f[0], o.f[1], o.f[2] }, o.f[3] > 0 ? (double)o.f[3] : 0.0, 1 };
				} else if (o.op == LOL_OP_RBOX) {
					bool ok = true;
					for (int j = 0; j < 7; j++) ok = ok && sane(o.f[j]);
					ok = ok && o.f[3] >= 0 && o.f[4] >= 0 && o.f[5] >= 0 && o.f[6] >= 0;
					const double hb = sqrt((double)o.f[3] * o.f[3] + (double)o.f[4] * o.f[4] + (double)o.f[5] * o.f[5]);
					n.bound = { ok, { o.f[0], o.f[1], o.f[2] }, hb * (1.0 + 1e-12) + o.f[6], 1 };
				} else if (o.op == LOL_OP_SMIN || o.op == LOL_OP_SMIN_R) {
					const int top = st.back(); st.pop_back();
					const int under = st.back(); st.pop_back();
					n.a = o.op == LOL_OP_SMIN ? under : top;
					n.b = o.op == LOL_OP_SMIN ? top : under;
					n.bound = enclose(nodes[n.a].bound, nodes[n.b].bound);
					if (!(sane(o.f[0]) && o.f[0] > 0)) n.bound.ok = false;
					n.bound.r += 0.25 * (double)o.f[0];
					n.bound.levels++;
					if (!sane(n.bound.r)) n.bound.ok = false;
					n.prims = nodes[n.a].prims + nodes[n.b].prims;
					n.fon = nodes[n.a].fon && nodes[n.b].fon && fast && fast->has(o.f[0]);
				}
				st.push_back((int)nodes.size());
				nodes.push_back(n);
			}
		}
		/* Saturation culling inside a smooth union (fast struct only, proven k > 0): sminf(a, b, k) is EXACTLY
		 * b - dlt*0.f = b + 0.f when dlt = b - a <= -ks (sminf_fastdiv_sat), so operand `a` need not be evaluated
		 * where it is provably that much greater than b.  b is evaluated first; with sb = fl(b + ks) the bounding-
		 * sphere test of the top-level culling (best := sb) gives a > sb for the binary32 value of `a`, hence
		 * dlt = fl(b - a) <= fl(b - sb) =: sw by the monotonicity of rounding, and sw <= -ks is checked directly;
		 * |p - C|^2 < 2^120 keeps every primitive of `a` (all within R < 10^15 of C) finite, so dlt is finite and
		 * dlt*0.f = -0.  A NaN or infinite b fails the comparisons.  Per wave, like every other skip. */
		std::function<int(int)> emit_node = [&](int ni) -> int {
			const Node& n = nodes[ni];
			const lol_op& o = P.ops[n.op];
			switch (o.op) {
			case LOL_OP_SPHERE:
				if (fsqrt && nan_flag && o.f[3] >= 0x1p-20f && std::isfinite(o.f[3])) {      /* sd_sphere_fast_nr: no range tracker */
					snprintf(line, sizeof line, "\t\tconst float t%d = sd_sphere_fast_nr<%d>(p, %s, %s, %s, %s);\n", t, fsqrt,
					         fbits(o.f[0]).c_str(), fbits(o.f[1]).c_str(), fbits(o.f[2]).c_str(), fbits(o.f[3]).c_str());
					object_has_nr = true;
				} else
				snprintf(line, sizeof line, "\t\tconst float t%d = sd_sphere%s(p, %s, %s, %s, %s%s);\n", t, fs,
				         fbits(o.f[0]).c_str(), fbits(o.f[1]).c_str(), fbits(o.f[2]).c_str(), fbits(o.f[3]).c_str(),
				         fsqrt ? ", rg" : "");
				s += line; return t++;
			case LOL_OP_RBOX:
				snprintf(line, sizeof line, "\t\tconst float t%d = sd_round_box%s(p, %s, %s, %s, %s, %s, %s, %s%s);\n", t,
				         fs,
				         fbits(o.f[0]).c_str(), fbits(o.f[1]).c_str(), fbits(o.f[2]).c_str(), fbits(o.f[3]).c_str(),
				         fbits(o.f[4]).c_str(), fbits(o.f[5]).c_str(), fbits(o.f[6]).c_str(), fsqrt ? ", rg" : "");
				s += line; return t++;
			case LOL_OP_PLANE:
				snprintf(line, sizeof line, "\t\tconst float t%d = p.y - %s;\n", t, fbits(o.f[0]).c_str());
				s += line; return t++;
			default: break;
			}
			/* LOL_OP_SMIN / LOL_OP_SMIN_R */
			const float ks = smooth_sat_threshold(o.f[0]);
			const bool proven = fast && fast->has(o.f[0]);
			/* without v_div_fixup where that is proven too; such an object's value is then voted on for NaN (an infinite
			 * operand difference — lol_kernel.h, smin_h_fast) like one with spheres that carry no range tracker */
			const char* fx = proven && fast->has_nf(o.f[0]) ? "<false>" : "";
			if (fx[0]) object_has_nr = true;
			/* the arithmetic shortcut only where the SDF is inlined: in the out-of-line function its four-way branching
			 * costs more than it saves (504-op chain: 100 -> 42 Mpixels/s) */
			const bool sat_arith = !out_of_line && ks > 0.f;
			const std::string kk = fbits(o.f[0]), k2 = fbits(2.0f * o.f[0]), hrk = fbits(0.5f * (1.0f / o.f[0])), kss = fbits(ks);
			if (proven && ks > 0.f && sat_cull_min_prims > 0 && nodes[n.a].bound.ok && nodes[n.a].prims >= (uint32_t)sat_cull_min_prims) {
				const int b = emit_node(n.b);
				const CullTest ct = make_test(nodes[n.a].bound);
				const int r = t++, q = n_tests++;
				snprintf(line, sizeof line,
				         "\t\tfloat t%d;\n"
				         "\t\t{ const float sb%d = t%d + %s, sw%d = t%d - sb%d;\n"
				         "\t\t  const float sx%d = p.x - %s, sy%d = p.y - %s, sz%d = p.z - %s;\n"
				         "\t\t  const float sl%d = (sx%d * sx%d + sy%d * sy%d) + sz%d * sz%d;\n"
				         "\t\t  const float su%d = (sb%d + %s) * %s;\n"
				         "\t\t  if ((vote(!(sl%d > su%d * su%d)) | vote(!(su%d > 0.f)) | vote(!(sw%d <= -%s)) | vote(!(sl%d < 0x1p120f))) != 0) {\n",
				         r, q, b, kss.c_str(), q, b, q,
				         q, fbits(ct.c[0]).c_str(), q, fbits(ct.c[1]).c_str(), q, fbits(ct.c[2]).c_str(),
				         q, q, q, q, q, q, q,
				         q, q, fbits(ct.rm).c_str(), fbits(ct.k).c_str(),
				         q, q, q, q, q, kss.c_str(), q);
				s += line;
				const int a = emit_node(n.a);
				if (sat_arith)
					snprintf(line, sizeof line, "\t\t  t%d = sminf_fastdiv_sat%s(t%d, t%d, %s, %s, %s, %s);\n\t\t  } else t%d = t%d + 0.f;\n\t\t}\n",
					         r, fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str(), kss.c_str(), r, b);
				else
					snprintf(line, sizeof line, "\t\t  t%d = sminf_fastdiv%s(t%d, t%d, %s, %s, %s);\n\t\t  } else t%d = t%d + 0.f;\n\t\t}\n",
					         r, fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str(), r, b);
				s += line;
				return r;
			}
			/* operands in program order (a flattened chain keeps its deep operand first and its operand stack shallow) */
			const bool a_first = o.op == LOL_OP_SMIN;
			const int first = emit_node(a_first ? n.a : n.b), second = emit_node(a_first ? n.b : n.a);
			const int a = a_first ? first : second, b = a_first ? second : first;
			if (proven && sat_arith)
				snprintf(line, sizeof line, "\t\tconst float t%d = sminf_fastdiv_sat%s%s(t%d, t%d, %s, %s, %s, %s);\n", t,
				         nodes[n.a].fon && nodes[n.b].fon ? "2" : "", fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str(), kss.c_str());
			else if (proven)
				snprintf(line, sizeof line, "\t\tconst float t%d = sminf_fastdiv%s(t%d, t%d, %s, %s, %s);\n", t, fx, a, b, kk.c_str(), k2.c_str(), hrk.c_str());
			else
				snprintf(line, sizeof line, "\t\tconst float t%d = sminf_(t%d, t%d, %s);\n", t, a, b, kk.c_str());
			s += line; return t++;
		};
		object_has_nr = false;
		const int d = emit_node((int)nodes.size() - 1);
		if (object_has_nr) {                 /* a NaN from a sphere without range tracker reaches the object's value: 0 * NaN (or inf) = NaN */
			snprintf(line, sizeof line, "\t\tnanacc = __builtin_fmaf(t%d, 0.f, nanacc);\n", d);
			s += line;
		}
		if (dist_only)
			snprintf(line, sizeof line, "\t\tbest = vmin_(t%d, best);\n", d);
		else if (R.id < max_id_seen)      /* evaluated after an object that follows it in the file: ties go to the lower id */
			snprintf(line, sizeof line, "\t\tif (t%d < best || (t%d == best && best_id > %uu)) { best = t%d; best_id = %uu; }\n", d, d, R.id, d, R.id);
		else
			snprintf(line, sizeof line, "\t\tif (t%d < best) { best = t%d; best_id = %uu; }\n", d, d, R.id);
		s += line;
		if (R.id > max_id_seen) max_id_seen = R.id;
		for (uint32_t k = 0; k < runs_ending[oi + 1]; k++) s += "\t\t} }\n";               /* every run that ends here */
	}
	if (!out_of_line) s += "\t}\n";
	}      /* pass */
	if (out_of_line) {
		s += "\t\treturn { best, best_id, rg.lo, rg.hi, nanacc };\n}\n";
		snprintf(line, sizeof line, "struct %s {\n\tstatic constexpr bool ASSUME_SETTLED = %s;\n\tstatic constexpr bool ASK_ID_ONCE = false;\n\tRange rg;\n\tfloat nanacc = 0.f;\n"
		         "\t__device__ __forceinline__ void loop_done() {}\n"
		         "\t__device__ __forceinline__ void eval(V3 p, float& best, u32& best_id) {\n"
		         "\t\tconst SdfOut o = %s_fn(p.x, p.y, p.z, rg.lo, rg.hi);\n"
		         "\t\tbest = o.best; best_id = o.id; rg.lo = o.lo; rg.hi = o.hi; nanacc += o.nanacc;\n\t}\n"
		         "\t__device__ __forceinline__ void eval_dist(V3 p, float& best) { u32 unused; eval(p, best, unused); }\n};\n", name, fast ? "true" : "false", name);
		s += line;
	} else {
		s += "};\n";
	}
}
bool spec_out_of_line(const lol_program& P, int form) {
	if (form == SPEC_OUT_OF_LINE) return true;
	if (form == SPEC_INLINE) return false;
	uint32_t limit = LOL_SPEC_INLINE_MAX_OPS;
	if (const char* e = tuning_env("LOL_GPU_SPEC_INLINE_MAX")) limit = (uint32_t)strtoul(e, nullptr, 10);
	return P.n_ops > limit;
}

std::string generate_source(const lol_program& P, const FastPaths* fast, bool cull, int form = SPEC_BY_SIZE) {
	std::string s;
	const bool ool = spec_out_of_line(P, form);
	const std::vector<RootBound> roots = analyse_roots(P);
	const CullPlan plan = plan_culling(roots, cull);
	s += "#include \"lol_kernel.h\"\n";
	s += "namespace lol {\n";
	/* Register budget.  The SDF of one object is a long dependent chain (every smooth min waits for the one below
	 * it) fed by independent primitives, and the kernel is compiled with the max-ILP scheduling strategy
	 * (compile_spec): the more registers a wave may use, the more primitives it keeps in flight.  Measured on
	 * MI355X (profiles/r2_large_scene_ab.jsonl): with max-ILP, scene4 (12 ops) is fastest when 8 waves per SIMD are
	 * kept (64 VGPRs: 4640 vs 4530 Mpixels/s unconstrained), a 44-op chain at >= 6, chains of 142+ ops at >= 4
	 * (128 VGPRs: 427 vs 400 Mpixels/s at 8). */
	const int waves_lo = P.n_ops <= 32 ? 8 : P.n_ops <= 96 ? 6 : 4, waves_hi = 8;
	const std::string occupancy = " __attribute__((amdgpu_waves_per_eu(" + std::to_string(waves_lo) + ", " + std::to_string(waves_hi) + ")))";
	if (ool) s += "struct SdfOut { float best; u32 id; u32 lo, hi; float nanacc; };\n";
	emit_sdf(s, P, "SpecSdfExact", nullptr, ool, roots, plan, occupancy);
	const bool any_fast = fast && (fast->sqrt_kind || !fast->div_ok.empty());
	if (any_fast) emit_sdf(s, P, "SpecSdfFast", fast, ool, roots, plan, occupancy);
	s += "}  // namespace lol\n";
	/* where lights / materials are read from is a property of the scene too (lol_kernel.h, TABLES_LDS_MAX_DWORDS) */
	const bool tables_global = !lol::tables_in_lds(P.n_lights, P.n_materials, P.n_roots);
	const std::string tg = tables_global ? "true" : "false";
	/* The pipeline once as a template on COUNT (lol_kernel.h, march: the per-lane step counters), and as one or two kernels:
	 *   lol_render_spec_steps  counts steps: frames with diagnostics (lol_gpu_debug::steps) and the one frame of a view that records
	 *                          what its pixels cost (lol_gpu.hip, "pixels dealt by cost");
	 *   lol_render_spec        does not (+1.3 % on C3): every other frame.
	 * A scene above LOL_SPEC_TWO_KERNELS_MAX_OPS gets the counting kernel alone, under the name lol_render_spec: a second copy of
	 * its pipeline would nearly double what the compiler takes for it. */
	const bool two = P.n_ops <= LOL_SPEC_TWO_KERNELS_MAX_OPS;
	s += "template <bool COUNT> __device__ __forceinline__ void lol_spec_body(const lol::Launch& L, lol::u32* lds) {\n";
	if (!tables_global) {
		s += "\tlol::stage_common(L, lds);\n";
		s += "\t__syncthreads();\n";
	}
	s += "\tif (!lol::start_tile_clock<" + tg + ">(L, lds)) return;\n";
	if (any_fast) {
		/* the fast pipeline takes FLAG_SHADOW_SETTLED for granted (lol_kernel.h, soft_shadow): a launch without it is the plain pipeline's */
		s += "\tlol::Pixel P;\n";
		s += "\tbool plain = !(L.flags & lol::FLAG_SHADOW_SETTLED);\n";
		s += "\tif (!plain) {\n";
		s += "\t\tlol::SpecSdfFast fast;\n";
		s += "\t\tP = lol::shade_pixel<lol::SpecSdfFast, " + tg + ", COUNT>(L, fast, lds);\n";
		s += "\t\tplain = lol::unproven(fast);\n";
		s += "\t}\n";
		s += "\tif (plain) {\n";
		s += "\t\tlol::SpecSdfExact exact;\n";
		s += "\t\tP = lol::shade_pixel<lol::SpecSdfExact, " + tg + ", COUNT>(L, exact, lds);\n";
		s += "\t}\n";
	} else {
		s += "\tlol::SpecSdfExact exact;\n";
		s += "\tlol::Pixel P = lol::shade_pixel<lol::SpecSdfExact, " + tg + ", COUNT>(L, exact, lds);\n";
	}
	s += "\tlol::store_pixel<" + tg + ">(L, P, lds);\n";
	s += "}\n";
	const std::string head = "extern \"C\" __global__ __launch_bounds__(lol::BLOCK)" + occupancy + " void ";
	const std::string tail = "(const lol::Launch L) {\n\textern __shared__ lol::u32 lds[];\n\tlol_spec_body<";
	if (two) s += head + "lol_render_spec_steps" + tail + "true>(L, lds);\n}\n";
	s += head + "lol_render_spec" + tail + (two ? "false" : "true") + ">(L, lds);\n}\n";
	/* the SDF alone at arbitrary points (lol_gpu_sdf_batch) */
	s += "extern \"C\" __global__ __launch_bounds__(64) void lol_sdf_spec(const float* pts, float* dist, lol::u32* id, lol::u32 n) {\n";
	s += "\tlol::SpecSdfExact exact;\n";
	if (any_fast) s += "\tlol::SpecSdfFast fast;\n\tlol::sdf_points(fast, exact, true, pts, dist, id, n);\n";
	else          s += "\tlol::sdf_points(exact, exact, false, pts, dist, id, n);\n";
	s += "}\n";
	return s;
}

unsigned long long fnv64(const void* data, size_t n);
std::string fnv_hex(const void* data, size_t n) {
	char b[20];
	snprintf(b, sizeof b, "%016llx", fnv64(data, n));
	return b;
}

/* Process-wide cache of compiled kernels: hosts (and the tests) upload the same scene many times. */
std::mutex g_cache_mutex;
std::unordered_map<std::string, std::vector<char>> g_code_cache;

/* ... and a cache on disk, so that the N ranks of a multi-GPU run (and the next run of the same host) do not each
 * pay the 0.3 - 1 s hipRTC compile of the same scene.  One file per key under LOL_GPU_CACHE_DIR (default
 * $XDG_CACHE_HOME/lol_gpu or $HOME/.cache/lol_gpu; set it to the empty string to switch the disk cache off): the
 * file holds the full key in front of the code object and is only used when that key matches byte for byte, so a
 * hash collision or a stale file can never hand out the wrong kernel; writes go through a temporary name + rename.
 * Any I/O failure simply means "not cached". */
std::string disk_cache_path(const std::string& key) {
	const char* e = getenv("LOL_GPU_CACHE_DIR");
	std::string dir;
	if (e) { if (!e[0]) return ""; dir = e; }
	else if (const char* x = getenv("XDG_CACHE_HOME")) { if (!x[0]) return ""; dir = std::string(x) + "/lol_gpu"; }
	else if (const char* h = getenv("HOME")) { if (!h[0]) return ""; dir = std::string(h) + "/.cache/lol_gpu"; }
	else return "";
	for (size_t i = 1; i <= dir.size(); i++)              /* mkdir -p */
		if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0700);
	unsigned long long h = 0xcbf29ce484222325ull;         /* FNV-1a of the key names the file */
	for (unsigned char c : key) { h ^= c; h *= 0x100000001b3ull; }
	char name[40];
	snprintf(name, sizeof name, "/%016llx.co", h);
	return dir + name;
}

/* GPU code is only ever loaded from a file this user wrote: the file and its directory must belong to the effective
 * user and must not be writable by group or others (a shared LOL_GPU_CACHE_DIR / XDG_CACHE_HOME would otherwise let
 * another user plant kernels), and the code object must match the checksum stored next to it. */
bool private_to_user(const struct stat& st) { return st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH)); }

unsigned long long fnv64(const void* data, size_t n) {
	unsigned long long h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < n; i++) { h ^= static_cast<const unsigned char*>(data)[i]; h *= 0x100000001b3ull; }
	return h;
}

bool disk_cache_load(const std::string& key, std::vector<char>& code) {
	const std::string path = disk_cache_path(key);
	if (path.empty()) return false;
	struct stat dir_st, file_st;
	const std::string dir = path.substr(0, path.rfind('/'));
	if (stat(dir.c_str(), &dir_st) != 0 || !S_ISDIR(dir_st.st_mode) || !private_to_user(dir_st)) return false;
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	bool ok = false;
	unsigned long long klen = 0, clen = 0, sum = 0;
	if (fstat(fileno(f), &file_st) == 0 && S_ISREG(file_st.st_mode) && private_to_user(file_st) &&
	    fread(&klen, 8, 1, f) == 1 && fread(&clen, 8, 1, f) == 1 && fread(&sum, 8, 1, f) == 1 &&
	    klen == key.size() && clen > 0 && clen < (1ull << 28)) {
		std::string k(klen, 0);
		code.resize(clen);
		ok = fread(&k[0], 1, klen, f) == klen && k == key && fread(code.data(), 1, clen, f) == clen && fnv64(code.data(), clen) == sum;
	}
	fclose(f);
	return ok;
}

void disk_cache_store(const std::string& key, const std::vector<char>& code) {
	const std::string path = disk_cache_path(key);
	if (path.empty()) return;
	char tmp[64];
	snprintf(tmp, sizeof tmp, ".%ld.tmp", (long)getpid());
	const std::string t = path + tmp;
	FILE* f = fopen(t.c_str(), "wb");
	if (!f) return;
	(void)fchmod(fileno(f), 0600);
	const unsigned long long klen = key.size(), clen = code.size(), sum = fnv64(code.data(), code.size());
	const bool ok = fwrite(&klen, 8, 1, f) == 1 && fwrite(&clen, 8, 1, f) == 1 && fwrite(&sum, 8, 1, f) == 1 &&
	                fwrite(key.data(), 1, klen, f) == klen && fwrite(code.data(), 1, clen, f) == clen;
	if (fclose(f) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0) (void)remove(t.c_str());
}

/* The long-branch bug described in compile_spec, as it looks in the code: a relaxed branch that goes through s[30:31], the
 * register pair a function RETURNS through —
 *     s_getpc_b64 s[30:31];  s_add_u32 s30, s30, <lit>;  s_addc_u32 s31, s31, <lit>;  s_setpc_b64 s[30:31]
 * (a call is s_getpc into some OTHER pair + s_swappc_b64 s[30:31], <pair>; a return is a bare s_setpc_b64 s[30:31] with no
 * s_getpc of that pair before it).  Recognised by instruction fields, not by four literal words (round-4 review): SOP1
 * s_getpc_b64 with SDST = s30, followed within a few instructions by SOP1 s_setpc_b64 with SSRC0 = s30 and no s_swappc_b64
 * in between — whatever arithmetic (add / sub, literal in either source position, a scavenged temporary, s_nop padding)
 * sits between the two.  The words are read with memcpy at EVERY byte offset: no assumption about where the buffer or the
 * text section inside the ELF begins, and nothing that can make the check pass by default.  A false alarm — constants that
 * happen to spell the two instructions eight dwords apart — only costs the scene its own kernel (the interpreter renders). */
bool has_return_clobbering_branch(const void* data, size_t n_bytes) {
	constexpr uint32_t SOP1 = 0xBE800000u, SOP1_MASK = 0xFF800000u;         /* [31:23] = 0b1_0111_1101 */
	constexpr uint32_t OP_GETPC = 28, OP_SETPC = 29, OP_SWAPPC = 30;        /* SOP1 opcodes (GFX9 / gfx950 encoding), bits [15:8] */
	constexpr uint32_t RETURN_PAIR = 30;                                    /* s[30:31] */
	constexpr size_t WINDOW = 12;                                           /* dwords after the s_getpc in which the s_setpc counts */
	const unsigned char* b = static_cast<const unsigned char*>(data);
	auto word = [&](size_t at) { uint32_t w; memcpy(&w, b + at, 4); return w; };
	for (size_t at = 0; at + 8 <= n_bytes; at++) {
		const uint32_t w = word(at);
		if ((w & SOP1_MASK) != SOP1 || ((w >> 8) & 0xFF) != OP_GETPC || ((w >> 16) & 0x7F) != RETURN_PAIR) continue;
		for (size_t k = 1; k <= WINDOW && at + 4 * k + 4 <= n_bytes; k++) {
			const uint32_t v = word(at + 4 * k);
			if ((v & SOP1_MASK) != SOP1) continue;
			const uint32_t op = (v >> 8) & 0xFF;
			if (op == OP_SWAPPC) break;                                     /* a call: the pair is being written as a link register */
			if (op == OP_SETPC && (v & 0xFF) == RETURN_PAIR) return true;
		}
	}
	return false;
}
bool has_return_clobbering_branch(const std::vector<char>& code) { return has_return_clobbering_branch(code.data(), code.size()); }

/* hipRTC: generated source + lol_kernel.h → code object for `arch`.  Needs no device. */
bool compile_spec(const lol_program& P, const FastPaths* fast, const std::string& arch, std::vector<char>& code,
                  std::string& log, std::string* src_out, bool cull, int form) {
	std::string src = generate_source(P, fast, cull, form);
	if (src_out) *src_out = src;
	if (const char* dump = tuning_env("LOL_GPU_DUMP_SPEC_SOURCE"))       /* debugging aid: the source as really generated on this device */
		if (FILE* f = fopen(dump, "w")) { fputs(src.c_str(), f); fclose(f); }
	int rtc_major = 0, rtc_minor = 0;
	(void)hiprtcVersion(&rtc_major, &rtc_minor);
	/* the option list first: it is part of the cache key (a changed flag — -ffp-contract above all — must never be served
	 * a code object compiled under the old one) */
	std::string arch_opt = "--offload-arch=" + arch;
	/* -ffp-contract=off: no FMA contraction (the reference has none); the rest are hipcc's defaults made explicit */
	/* -fno-slp-vectorize: the SLP pass pairs scalar f32 ops into v_pk_*_f32, which issue at half the
	 * rate of two scalar ops on gfx950 (tools/valu_rate.hip); measured +10 % Mpixels/s without it. */
	/* -amdgpu-sched-strategy=max-ilp: schedule for instruction-level parallelism within a wave rather than for
	 * occupancy.  The default strategy serialises the independent primitives of a long smooth-union chain to save
	 * registers; with max-ILP the same instructions run 1.4x faster on 142 - 1024-op scenes and 2 - 5 % faster on the
	 * example scenes (generate_source sets the matching register budget).  Scheduling only: same instructions, same bits.
	 * An LLVM that does not know an -mllvm option ends the PROCESS from its option parser, so the option is only
	 * passed to hipRTC versions it was verified on (hiprtcVersion >= 9.0 = ROCm 7.x); LOL_GPU_SCHED=default leaves it out. */
	std::vector<const char*> opts = { arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
	                                  "-fno-slp-vectorize" };
	const char* sched = tuning_env("LOL_GPU_SCHED");
	if (rtc_major >= 9 && !(sched && !strcmp(sched, "default"))) {
		opts.push_back("-mllvm"); opts.push_back("-amdgpu-sched-strategy=max-ilp");
		/* ... and no post-RA scheduling pass: it re-orders the ILP-friendly schedule after register allocation and
		 * costs 10 % on C3 (4650 -> 5150 Mpixels/s without it, same box and call; nothing on the large scenes) */
		opts.push_back("-mllvm"); opts.push_back("-enable-post-misched=0");
		/* ... and SimplifyCFG may turn small two-sided branches into selects more readily (default threshold 2): the per-lane
		 * `if (alive) { ... }` updates around the SDF become straight-line code for the ILP scheduler.  Sweep of 3 ... 64 on one
		 * box (tools/rtc_flag_sweep.sh phi): from 4 upwards C2 +1.5 ... 2.5 %, C3 +0.2 %, the large scenes +-3 %; same bits. */
		opts.push_back("-mllvm"); opts.push_back("-phi-node-folding-threshold=8");
		/* ... and NO register reserved ahead of time for long branches.  An out-of-line SDF function of more than 128 KB (about
		 * 800 ops: a field of 600 objects) has forward branches beyond s_cbranch's 16-bit reach; LLVM's AMDGPU backend then
		 * reserves "an unused" SGPR pair for the s_getpc / s_add / s_setpc sequence before register allocation — and in a leaf
		 * function picks s[30:31], the RETURN ADDRESS: the function jumps, and at its end "returns" to the branch target for
		 * ever (found in round 4 when scenes lost their 1024-op capacity: the kernel never finished; ROCm 7.0 and 7.2 alike).
		 * With the factor 0 no register is reserved and the branch relaxation scavenges a dead one at the branch, correctly.
		 * has_return_clobbering_branch() below refuses any code object that still shows the pattern. */
		opts.push_back("-mllvm"); opts.push_back("-amdgpu-long-branch-factor=0");
	}
	/* ... and so is the compiler: which libhiprtc this process has loaded.  A Python host gets the one torch ships, a C host the
	 * system's, a process under rocprofv3 yet another mix — the same source came out as three different code objects — and one
	 * compiler's output must not be handed to a process that would have compiled something else.
	 * LOL_GPU_CACHE_ANY_COMPILER=1 leaves the compiler out of the key: a profiling aid (tools/final_profile.sh lets a plain run
	 * compile the kernels, and the runs under the profiler load exactly those). */
	std::string compiler = "?";
	{
		Dl_info info;
		if (dladdr(reinterpret_cast<void*>(&hiprtcCompileProgram), &info) && info.dli_fname) compiler = info.dli_fname;
		compiler = std::to_string(rtc_major) + "." + std::to_string(rtc_minor) + " " + compiler;
		const char* any = tuning_env("LOL_GPU_CACHE_ANY_COMPILER");
		if (any && any[0] == '1') compiler = "*";         /* its version too: torch's hipRTC and the system's differ in it */
	}
	std::string key = "lol_gpu/4|hiprtc " + compiler + "|";
	for (const char* o : opts) { key += o; key += ' '; }
	key += "|" + src;
	{
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		auto it = g_code_cache.find(key);
		if (it != g_code_cache.end()) { code = it->second; log.clear(); return true; }
	}
	/* on disk the pipeline source (lol_kernel.h, embedded in this library) is part of the key: another build of the
	 * library must not pick up this one's kernels */
	const std::string disk_key = key + "|" + LOL_KERNEL_H_TEXT;
	if (disk_cache_load(disk_key, code)) {
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		g_code_cache[key] = code;
		log = "(code object from the disk cache)";
		return true;
	}
	const char* hdr_src[] = { LOL_KERNEL_H_TEXT };
	const char* hdr_name[] = { "lol_kernel.h" };
	hiprtcProgram prog = nullptr;
	if (hiprtcCreateProgram(&prog, src.c_str(), "lol_render_spec.hip", 1, hdr_src, hdr_name) != HIPRTC_SUCCESS) {
		log = "hiprtcCreateProgram failed";
		return false;
	}
	bool options_dropped = false;
	hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
	if (r != HIPRTC_SUCCESS) {
		/* a hipRTC that does not know the scheduling option must not cost the specialisation: once more without it */
		std::vector<const char*> plain;
		for (size_t i = 0; i < opts.size(); i++) {
			if (!strcmp(opts[i], "-mllvm") && i + 1 < opts.size() &&
			    (!strcmp(opts[i + 1], "-amdgpu-sched-strategy=max-ilp") || !strcmp(opts[i + 1], "-enable-post-misched=0") ||
			     !strcmp(opts[i + 1], "-phi-node-folding-threshold=8") || !strcmp(opts[i + 1], "-amdgpu-long-branch-factor=0"))) { i++; continue; }
			plain.push_back(opts[i]);
		}
		if (plain.size() != opts.size()) {
			/* what comes out now was NOT compiled under the options the key lists: it serves this process (the retry would
			 * give the same again) but never goes to disk under that key */
			options_dropped = true;
			hiprtcDestroyProgram(&prog);
			prog = nullptr;
			if (hiprtcCreateProgram(&prog, src.c_str(), "lol_render_spec.hip", 1, hdr_src, hdr_name) != HIPRTC_SUCCESS) {
				log = "hiprtcCreateProgram failed";
				return false;
			}
			r = hiprtcCompileProgram(prog, (int)plain.size(), plain.data());
		}
	}
	size_t log_size = 0;
	hiprtcGetProgramLogSize(prog, &log_size);
	log.clear();
	if (log_size > 1) { log.resize(log_size); hiprtcGetProgramLog(prog, &log[0]); }
	if (r != HIPRTC_SUCCESS) {
		log = std::string("hipRTC: ") + hiprtcGetErrorString(r) + "\n" + log;
		hiprtcDestroyProgram(&prog);
		return false;
	}
	size_t code_size = 0;
	hiprtcGetCodeSize(prog, &code_size);
	code.resize(code_size);
	hiprtcGetCode(prog, code.data());
	hiprtcDestroyProgram(&prog);
	if (has_return_clobbering_branch(code)) {
		/* a kernel that would never finish is worse than no kernel: the interpreter renders this scene */
		log = "the compiler relaxed a long branch through s[30:31], the return address of a function (LLVM AMDGPU long-branch "
		      "register bug; see compile_spec): code object refused";
		code.clear();
		return false;
	}
	if (options_dropped && spec_out_of_line(P, form)) {
		/* The retry above also dropped -amdgpu-long-branch-factor=0, the option that KEEPS the compiler from that bug, and an
		 * out-of-line SDF is where it bites (a function beyond s_cbranch's reach).  The pattern check would be all that is left
		 * between this code object and a launch that never ends: not enough — the interpreter renders this scene. */
		log = "this hipRTC refused the -mllvm options (among them the workaround for LLVM's long-branch register bug) and the scene's "
		      "SDF is out of line: code object refused";
		code.clear();
		return false;
	}
	{
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		g_code_cache[key] = code;
	}
	if (!options_dropped) disk_cache_store(disk_key, code);
	return true;
}

#pragma GCC visibility pop

extern "C" {

/* Host-only view of the bound behind the culling test of top-level object `root` (0-based, file order):
 * 1 = bounded (centre and inflated radius R' out), 0 = no bound (never culled), < 0 = bad argument. */
int lol_gpu_cull_bounds(const lol_program* prog, uint32_t root, float c_out[3], float* r_out) {
	if (!prog || !c_out || !r_out || root >= prog->n_roots || prog->n_ops > LOL_MAX_OPS) return LOL_GPU_ERR_ARG;
	const std::vector<RootBound> roots = analyse_roots(*prog);
	if (root >= roots.size()) return LOL_GPU_ERR_ARG;
	const RootBound& R = roots[root];
	if (!R.bounded) return 0;
	const CullTest t = make_test(R.sphere());
	c_out[0] = t.c[0]; c_out[1] = t.c[1]; c_out[2] = t.c[2];
	*r_out = t.rm;
	return 1;
}

/* ... and the tighter two-sphere bound, where the object has one (cluster_bounds): value(p) >= min_j (|p - c_j| - r_j) with the
 * inflated radii the kernel tests with.  Returns the number of spheres written to out[j] = {cx, cy, cz, r'} (0 or 2). */
int lol_gpu_cull_bounds_clusters(const lol_program* prog, uint32_t root, float out[3][4]) {
	if (!prog || !out || root >= prog->n_roots || prog->n_ops > LOL_MAX_OPS) return LOL_GPU_ERR_ARG;
	const std::vector<RootBound> roots = analyse_roots(*prog);
	if (root >= roots.size()) return LOL_GPU_ERR_ARG;
	int n = 0;
	for (const Sphere& c : roots[root].clusters) {
		if (n == 3) break;
		const CullTest t = make_test(c);
		out[n][0] = t.c[0]; out[n][1] = t.c[1]; out[n][2] = t.c[2]; out[n][3] = t.rm;
		n++;
	}
	return n;
}

int lol_gpu_testing_has_return_clobbering_branch(const void* code, size_t n_bytes) {
	if (!code) return LOL_GPU_ERR_ARG;
	return has_return_clobbering_branch(code, n_bytes) ? 1 : 0;
}

/* Offline use (tests, ISA inspection; needs no device): compile the scene-specialised kernel for
 * `arch` and write `<out_base>.hip` (generated source) and `<out_base>.co` (code object). */
int lol_gpu_compile_offline(const lol_program* prog, const char* arch, const char* out_base, int assume_fast,
                            char* log, size_t logcap) {
	if (!prog || !arch) return LOL_GPU_ERR_ARG;
	std::vector<char> code;
	std::string lg, src;
	FastPaths fast;
	if (assume_fast) {                 /* ISA inspection only: pretend every shortcut was proven */
		fast.sqrt_kind = assume_fast >= 1 && assume_fast <= 3 ? 4 - assume_fast : 3;   /* 1 → sqrt_r2, 2 → sqrt_gs, 3 → sqrt_pm */
		fast.sqrt_tiny_ok = true;
		for (uint32_t i = 0; i < prog->n_ops; i++)
			if ((prog->ops[i].op == LOL_OP_SMIN || prog->ops[i].op == LOL_OP_SMIN_R) && !fast.has(prog->ops[i].f[0]))
				{ fast.div_ok.push_back(prog->ops[i].f[0]); fast.div_nf_ok.push_back(prog->ops[i].f[0]); }
	}
	bool ok = false;
	{
		/* on the large-stack thread, like every run of the scene compiler (BigStackThread) */
		BigStackThread th;
		auto work = [&]() {
			try { std::lock_guard<std::mutex> rtc(g_rtc_mutex); ok = compile_spec(*prog, &fast, arch, code, lg, &src, culling_enabled(1)); }
			catch (...) { ok = false; lg = "the scene compiler ran out of memory"; }
		};
		bool started = false;
		try { started = th.start(work); } catch (...) { started = false; }
		if (started) th.join(); else work();
	}
	if (log && logcap) snprintf(log, logcap, "%s", lg.c_str());
	if (out_base && out_base[0]) {
		std::string base = out_base;
		if (FILE* f = fopen((base + ".hip").c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
		if (ok) if (FILE* f = fopen((base + ".co").c_str(), "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
	}
	return ok ? LOL_GPU_OK : LOL_GPU_ERR_UNSUPPORTED;
}

}  // extern "C"
