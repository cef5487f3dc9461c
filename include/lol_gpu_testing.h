/*
 * lol_gpu_testing.h — fault injection and geometry switches for the TEST SUITE (tests/test_gpu_boundary.py,
 * tests/test_multi_device.py).  Not part of the drop-in boundary: a renderer.h host never calls these, and nothing in
 * the library reads the environment for them (rounds 2-3 used getenv(): an inherited variable could change production
 * behaviour).  Each switch lives in the context it is set on and dies with it.
 */
#ifndef LOL_GPU_TESTING_H
#define LOL_GPU_TESTING_H

#include "lol_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The next `n` lol_gpu_upload_program calls on this context fail at their copy step (as a failed hipMemcpy would), after
 * every check has passed: exercises the all-or-nothing upload. */
int lol_gpu_testing_fail_uploads(lol_gpu* ctx, int n);

/* The next `n` FIRST runs of the scene compiler for a scene of the middle sizes (257 ... 1024 ops: out-of-line form first, inlined
 * form behind it) count as failed: exercises "the inlined form is compiled all the same" (lol_gpu.hip, finish_specialise). */
int lol_gpu_testing_fail_first_tier(lol_gpu* ctx, int n);

/* Every `stride`-th PART gets the root's band height (lol_gpu_multi_set_root_band_rows) even on ONE device, so that a
 * one-GPU box runs bands of unequal height through split, launches, exchange, assembly and host copies.  0 = off. */
int lol_gpu_multi_testing_root_stride(lol_gpu_multi* m, int stride);

/* lol_gpu_multi_render_host issues every device's copies from that device's own host thread even when there is only one
 * device (which otherwise copies inline): the code every real multi-GPU host runs, reachable on a one-GPU box. */
int lol_gpu_multi_testing_force_copier_threads(lol_gpu_multi* m, int enable);

/* The check the scene compiler applies to every code object before it may be launched (lol_gpu.hip,
 * has_return_clobbering_branch): 1 when the bytes hold a branch relaxed through s[30:31] — s_getpc_b64 s[30:31] ...
 * s_setpc_b64 s[30:31], LLVM's long-branch register bug: a function that never returns — 0 when not.  No device needed. */
int lol_gpu_testing_has_return_clobbering_branch(const void* code, size_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LOL_GPU_TESTING_H */
