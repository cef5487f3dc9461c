// Microbenchmark: issue cost of the VALU instructions this renderer is made of, on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
// Every test issues 8 instructions per loop iteration per wave; IND = independent destinations,
// DEP = one dependent chain.  Reports cycles per wave-instruction per SIMD at w waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 32768;
typedef float float2v __attribute__((ext_vector_type(2)));

#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define REGS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define PREGS "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, float c) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	float2v p0 = { x0, x1 }, p1 = { x2, x3 }, p2 = { x4, x5 }, p3 = { x6, x7 }, pa = { a, a }, pb = { b, b };
	float2v p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
	for (int i = 0; i < ITERS; i++) {
		if (MODE == 0) {
#define OP(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 1) {
#define OP(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
			asm volatile(R8(OP) : PREGS : "v"(pa), "v"(pb));
#undef OP
		} else if (MODE == 2) {
#define OP(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 3) {
#define OP(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : PREGS : "v"(pa), "v"(pb));
#undef OP
		} else if (MODE == 4) {
#define OP(n) "v_sqrt_f32 %" #n ", %" #n "\n"
			asm volatile(R8(OP) : REGS);
#undef OP
		} else if (MODE == 5) {
#define OP(n) "v_rsq_f32 %" #n ", %" #n "\n"
			asm volatile(R8(OP) : REGS);
#undef OP
		} else if (MODE == 6) {   // VOPC to vcc, then cndmask reading vcc
			asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
			             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
			             : REGS : "v"(a), "v"(b) : "vcc");
		} else if (MODE == 7) {
#define OP(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 8) {   // cndmask only (vcc set once outside)
#define OP(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
			asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" R8(OP) : REGS : "v"(a), "v"(b) : "vcc");
#undef OP
		} else if (MODE == 9) {   // compares only (VOPC → vcc)
#define OP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b) : "vcc");
#undef OP
		} else if (MODE == 10) {
#define OP(n) "v_min_f32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 11) {
#define OP(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 12) {
#define OP(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 13) {  // dependent chain of 8 fma on one register
			asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
			             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
			             : REGS : "v"(a), "v"(b));
		} else if (MODE == 14) {  // rsq then 7 dependent fma (the Goldschmidt shape)
			asm volatile("v_rsq_f32 %0, %0\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
			             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
			             : REGS : "v"(a), "v"(b));
		} else if (MODE == 15) {  // add with clamp modifier (VOP3)
#define OP(n) "v_add_f32_e64 %" #n ", %" #n ", %8 clamp\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 16) {  // mul by 32-bit literal
#define OP(n) "v_mul_f32 %" #n ", 0x40490fdb, %" #n "\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 17) {  // add with an SGPR operand
#define OP(n) "v_add_f32 %" #n ", %8, %" #n "\n"
			asm volatile(R8(OP) : REGS : "s"(c), "v"(b));
#undef OP
		} else if (MODE == 18) {  // VOP3 compare to SGPR pair + cndmask e64 reading it
			asm volatile("v_cmp_lt_f32 s[40:41], %0, %8\n v_cndmask_b32 %1, %1, %9, s[40:41]\n v_cmp_lt_f32 s[42:43], %2, %8\n v_cndmask_b32 %3, %3, %9, s[42:43]\n"
			             "v_cmp_lt_f32 s[40:41], %4, %8\n v_cndmask_b32 %5, %5, %9, s[40:41]\n v_cmp_lt_f32 s[42:43], %6, %8\n v_cndmask_b32 %7, %7, %9, s[42:43]\n"
			             : REGS : "v"(a), "v"(b) : "s40", "s41", "s42", "s43");
		} else if (MODE == 19) {
#define OP(n) "v_rcp_f32 %" #n ", %" #n "\n"
			asm volatile(R8(OP) : REGS);
#undef OP
		} else if (MODE == 22) {
#define OP(n) "v_add_f32 %" #n ", %8, %" #n "\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 23) {
#define OP(n) "v_mul_f32 %" #n ", %8, %" #n "\n"
			asm volatile(R8(OP) : REGS : "s"(c), "v"(b));
#undef OP
		} else if (MODE == 24) {
#define OP(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
			asm volatile(R8(OP) : REGS : "s"(c), "v"(b));
#undef OP
		} else if (MODE == 25) {
#define OP(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 26) {
#define OP(n) "v_max_f32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 27) {
#define OP(n) "v_mov_b32 %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 28) {
#define OP(n) "v_add_u32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 29) {  // cndmask_e64 reading a fixed SGPR pair written before the loop
#define OP(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[40:41]\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 30) {
#define OP(n) "v_and_b32 %" #n ", %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 31) {  // add with 32-bit literal
#define OP(n) "v_add_f32 %" #n ", 0x40490fdb, %" #n "\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 32) {  // add with inline constant
#define OP(n) "v_add_f32 %" #n ", 1.0, %" #n "\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 33) {  // fmac VOP2
#define OP(n) "v_fmac_f32 %" #n ", %8, %9\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 34) {  // fmamk (literal)
#define OP(n) "v_fmamk_f32 %" #n ", %" #n ", 0x40490fdb, %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 35) {  // max with abs/neg modifiers VOP3
#define OP(n) "v_add_f32_e64 %" #n ", |%" #n "|, -%8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b));
#undef OP
		} else if (MODE == 36) {  // v_cmp_e64 to sgpr only
#define OP(n) "v_cmp_lt_f32_e64 s[40:41], %" #n ", %8\n"
			asm volatile(R8(OP) : REGS : "v"(a), "v"(b) : "s40", "s41");
#undef OP
		} else if (MODE == 37) {  // v_mul_f32 v,v,v dependent chain through two regs alternating
			asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n"
			             "v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n"
			             : REGS : "v"(a), "v"(b));
		} else if (MODE == 40) {  // mix: 2 rsq + 6 fma, independent
			asm volatile("v_rsq_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			             "v_rsq_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			             : REGS : "v"(a), "v"(b));
		} else if (MODE == 41) {  // mix: 4 v_min (half rate) + 4 fma
			asm volatile("v_min_f32 %0, %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_min_f32 %2, %2, %8\n v_fma_f32 %3, %3, %8, %9\n"
			             "v_min_f32 %4, %4, %8\n v_fma_f32 %5, %5, %8, %9\n v_min_f32 %6, %6, %8\n v_fma_f32 %7, %7, %8, %9\n"
			             : REGS : "v"(a), "v"(b));
		} else if (MODE == 42) {  // mix: 1 rsq + 7 fma
			asm volatile("v_rsq_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			             : REGS : "v"(a), "v"(b));
		} else if (MODE == 20) {  // s_nop 0 x8
			asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n" : REGS);
		} else if (MODE == 21) {  // 4 fma + 4 s_nop 1 interleaved
			asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 1\n v_fma_f32 %1, %1, %8, %9\n s_nop 1\n v_fma_f32 %2, %2, %8, %9\n s_nop 1\n v_fma_f32 %3, %3, %8, %9\n s_nop 1\n"
			             : REGS : "v"(a), "v"(b));
		}
	}
	float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
	if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char* name, float* d, int khz) {
	printf("%-34s", name);
	for (int w : {1, 2, 4, 6, 8}) {
		int blocks = 256 * w;   // 256 CUs x (4 SIMDs x w waves / 4 waves per block)
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0.25f);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		double cycles = ms * 1e-3 * khz * 1e3;
		printf("  w=%d %5.2f", w, cycles / ((double)ITERS * 8 * w));
	}
	printf("\n");
	return 0;
}

int main() {
	hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
	printf("%s CUs=%d clock=%d kHz; cycles per wave-instruction per SIMD at w waves/SIMD\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
	float* d; CHECK(hipMalloc(&d, 4));
	hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0.25f);   // warm-up / clock ramp
	asm volatile("");
	CHECK(hipDeviceSynchronize());
	run<22>("v_add_f32 v,v,v", d, p.clockRate);
	run<32>("v_add_f32 inline const", d, p.clockRate);
	run<31>("v_add_f32 literal", d, p.clockRate);
	run<17>("v_add_f32 SGPR", d, p.clockRate);
	run<23>("v_mul_f32 SGPR", d, p.clockRate);
	run<24>("v_fma_f32 SGPR", d, p.clockRate);
	run<25>("v_sub_f32", d, p.clockRate);
	run<33>("v_fmac_f32", d, p.clockRate);
	run<34>("v_fmamk_f32 literal", d, p.clockRate);
	run<35>("v_add_f32_e64 |a|,-b", d, p.clockRate);
	run<37>("mul/add dependent chain", d, p.clockRate);
	run<26>("v_max_f32", d, p.clockRate);
	run<27>("v_mov_b32", d, p.clockRate);
	run<28>("v_add_u32", d, p.clockRate);
	run<30>("v_and_b32", d, p.clockRate);
	run<29>("v_cndmask_b32_e64 fixed sgpr", d, p.clockRate);
	run<36>("v_cmp_lt_f32_e64 -> sgpr", d, p.clockRate);
	run<0>("v_fma_f32 IND", d, p.clockRate);
	run<13>("v_fma_f32 DEP", d, p.clockRate);
	run<2>("v_mul_f32 IND", d, p.clockRate);
	run<16>("v_mul_f32 literal", d, p.clockRate);
	run<17>("v_add_f32 sgpr", d, p.clockRate);
	run<15>("v_add_f32_e64 clamp", d, p.clockRate);
	run<10>("v_min_f32", d, p.clockRate);
	run<11>("v_med3_f32", d, p.clockRate);
	run<12>("v_min3_u32", d, p.clockRate);
	run<1>("v_pk_fma_f32", d, p.clockRate);
	run<3>("v_pk_mul_f32", d, p.clockRate);
	run<4>("v_sqrt_f32", d, p.clockRate);
	run<5>("v_rsq_f32", d, p.clockRate);
	run<19>("v_rcp_f32", d, p.clockRate);
	run<14>("v_rsq + 7 dependent fma", d, p.clockRate);
	run<40>("MIX 2 rsq + 6 fma (serial = 3.5)", d, p.clockRate);
	run<42>("MIX 1 rsq + 7 fma (serial = 2.75)", d, p.clockRate);
	run<41>("MIX 4 v_min + 4 fma (serial = 3.0)", d, p.clockRate);
	run<9>("v_cmp_lt_f32 vcc", d, p.clockRate);
	run<8>("v_cndmask_b32 vcc", d, p.clockRate);
	run<6>("v_cmp vcc + v_cndmask pairs", d, p.clockRate);
	run<18>("v_cmp sgpr + v_cndmask e64 pairs", d, p.clockRate);
	run<7>("v_div_fixup_f32", d, p.clockRate);
	run<20>("s_nop 0", d, p.clockRate);
	run<21>("v_fma + s_nop 1 (per pair /2)", d, p.clockRate);
	return 0;
}
