/*
 * indirect_branch_probe.hip — can HIP C++ for gfx950 express the jump-table dispatch a bytecode interpreter wants
 * (s_setpc_b64 on a per-record handler offset)?  Round 2's review asked for it in render_interp; this is what the ROCm
 * 7.2 compiler does with the three ways to write it (tools/indirect_branch_probe.sh compiles each and looks at the ISA):
 *
 *   -DPROBE=1  computed goto (&&label, goto *tab[i])      → compiles; NO indirect jump in the ISA: the IndirectBr is
 *                                                            lowered to a switch, the switch to a compare chain driven by
 *                                                            flag registers (s_and_b64 vcc, exec, s[..]; s_cbranch_vccz …)
 *   -DPROBE=2  asm goto, s_setpc_b64 into a trampoline of  → clang crashes in "Fixup each natural loop to have a single
 *              s_branch %l[handler], handlers leave the loop   exit block" (UnifyLoopExits on a callbr)
 *   -DPROBE=3  the same with every handler inside the loop  → compiles, but the asm block is GONE from the ISA and the
 *                                                            handlers hang off uninitialised flag registers (wrong code)
 * So from C++ there is no indirect branch; an interpreter core with real jump-table dispatch has to be an assembly
 * function.  What render_interp does instead: one-hot header bits tested with s_bitcmp1 + s_cbranch, rare bodies out of
 * line (lol_kernel.h, LOL_RARE / LOL_OFTEN).
 */
#include <hip/hip_runtime.h>
typedef const __attribute__((address_space(4))) unsigned* cptr;

#ifndef PROBE
#define PROBE 1
#endif

__global__ void k(const unsigned* prog, float* out, unsigned n) {
	cptr p = (cptr)(unsigned long long)prog;
	float x = out[threadIdx.x], acc = 0.f;
#if PROBE == 1
	static const void* const tab[] = { &&op_add, &&op_mul, &&op_sub, &&op_end };
	for (;;) {
		unsigned op = p[0];
		float c = __builtin_bit_cast(float, p[1]);
		p += 2;
		goto *tab[op];
	op_add: acc += x + c; continue;
	op_mul: acc *= x * c; continue;
	op_sub: acc -= x - c; continue;
	op_end: break;
	}
#else
	for (unsigned left = n; left != 0; left--) {
		unsigned hoff = p[0];          /* byte offset of the trampoline entry, baked by the host: 12 + 4 * handler */
		float c = __builtin_bit_cast(float, p[1]);
		p += 2;
		asm goto("s_getpc_b64 vcc\n\t"
		         "s_add_u32 vcc_lo, vcc_lo, %0\n\t"
		         "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
		         "s_setpc_b64 vcc\n\t"
		         "s_branch %l1\n\t"
		         "s_branch %l2\n\t"
		         "s_branch %l3"
		         : : "s"(hoff) : "vcc", "scc" : op_add, op_mul, op_sub);
#if PROBE == 2
		__builtin_unreachable();
#endif
	op_add: acc += x + c; continue;
	op_mul: acc *= x * c; continue;
	op_sub: acc -= x - c; continue;
	}
#endif
	out[threadIdx.x] = acc;
}
