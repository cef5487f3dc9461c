// Microbenchmark: issue cost of the SCALAR side of an interpreter loop on gfx950 — SALU arithmetic, compares and
// (not-taken / taken) branches, v_readfirstlane, wave-uniform LDS reads, scalar loads — alone and mixed with VALU.
// Build: hipcc -O3 --offload-arch=gfx950 -o salu_rate salu_rate.hip ; run on the GPU box.
// Reports cycles per wave-instruction per SIMD at w waves/SIMD (8 instructions per loop iteration per wave).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 16384;
#define REGS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* tab, float a, float b) {
	__shared__ float lds[64];
	if (threadIdx.x < 64) lds[threadIdx.x] = threadIdx.x;
	__syncthreads();
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	unsigned la = 0;   // LDS byte address (uniform)
	for (int i = 0; i < ITERS; i++) {
		if (MODE == 0) {          // 8 s_add_u32
			asm volatile("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n"
			             "s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n" ::: "s40", "s41", "s42", "s43", "scc");
		} else if (MODE == 1) {   // 4 x (s_cmp + s_cbranch not taken)
			asm volatile("s_cmp_eq_u32 s40, 77\n s_cbranch_scc1 1f\n s_cmp_eq_u32 s40, 78\n s_cbranch_scc1 1f\n"
			             "s_cmp_eq_u32 s40, 79\n s_cbranch_scc1 1f\n s_cmp_eq_u32 s40, 80\n s_cbranch_scc1 1f\n 1:\n" ::: "s40", "scc");
		} else if (MODE == 2) {   // 4 x (s_cmp + s_cbranch taken, forward by one instruction)
			asm volatile("s_mov_b32 s40, 0\n"
			             "s_cmp_eq_u32 s40, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1: s_cmp_eq_u32 s40, 0\n s_cbranch_scc1 2f\n s_nop 0\n"
			             "2: s_cmp_eq_u32 s40, 0\n s_cbranch_scc1 3f\n s_nop 0\n 3: s_cmp_eq_u32 s40, 0\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n" ::: "s40", "scc");
		} else if (MODE == 3) {   // 4 fma + 4 s_add
			asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s40, s40, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s41, s41, 1\n"
			             "v_fma_f32 %2, %2, %8, %9\n s_add_u32 s42, s42, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s43, s43, 1\n"
			             : REGS : "v"(a), "v"(b) : "s40", "s41", "s42", "s43", "scc");
		} else if (MODE == 4) {   // 6 fma + s_cmp + s_cbranch (not taken)
			asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n s_cmp_eq_u32 s40, 77\n s_cbranch_scc1 1f\n"
			             "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n 1:\n"
			             : REGS : "v"(a), "v"(b) : "s40", "scc");
		} else if (MODE == 5) {   // 8 v_readfirstlane
			asm volatile("v_readfirstlane_b32 s40, %0\n v_readfirstlane_b32 s41, %1\n v_readfirstlane_b32 s42, %2\n v_readfirstlane_b32 s43, %3\n"
			             "v_readfirstlane_b32 s40, %4\n v_readfirstlane_b32 s41, %5\n v_readfirstlane_b32 s42, %6\n v_readfirstlane_b32 s43, %7\n"
			             : REGS : : "s40", "s41", "s42", "s43");
		} else if (MODE == 6) {   // 8 uniform ds_read_b32 (all lanes one address), waited once
			asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
			             "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)\n"
			             : REGS : "v"(la));
		} else if (MODE == 7) {   // 2 uniform ds_read_b128 + wait (8 dwords), counted as 8
			float4 q0, q1;
			asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n"
			             : "=v"(q0), "=v"(q1) : "v"(la));
			x0 += q0.x; x4 += q1.x;
		} else if (MODE == 8) {   // dependent: ds_read_b32 → wait → readfirstlane → s_cmp/branch (one "opcode fetch"), x2, counted as 8
			asm volatile("ds_read_b32 %0, %8\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s40, %0\n s_cmp_eq_u32 s40, 77\n s_cbranch_scc1 1f\n"
			             "ds_read_b32 %1, %8 offset:4\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s40, %1\n s_cmp_eq_u32 s40, 77\n s_cbranch_scc1 1f\n 1:\n"
			             : REGS : "v"(la) : "s40", "scc");
		} else if (MODE == 9) {   // scalar load x8 dwords + wait, counted as 8
			asm volatile("s_load_dwordx8 s[40:47], %0, 0x0\n s_waitcnt lgkmcnt(0)\n" :: "s"(tab) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
		} else if (MODE == 10) {  // 8 v_mov from SGPR
			asm volatile("v_mov_b32 %0, s40\n v_mov_b32 %1, s41\n v_mov_b32 %2, s42\n v_mov_b32 %3, s43\n v_mov_b32 %4, s40\n v_mov_b32 %5, s41\n v_mov_b32 %6, s42\n v_mov_b32 %7, s43\n"
			             : REGS);
		} else if (MODE == 11) {  // 6 fma + 2 s_add
			asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s40, s40, 1\n"
			             "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s41, s41, 1\n"
			             : REGS : "v"(a), "v"(b) : "s40", "s41", "scc");
		} else if (MODE == 12) {  // s_and_b64 / s_andn2_b64 exec-mask style ops
			asm volatile("s_and_b64 s[40:41], s[40:41], exec\n s_andn2_b64 s[42:43], exec, s[40:41]\n s_and_b64 s[40:41], s[40:41], exec\n s_andn2_b64 s[42:43], exec, s[40:41]\n"
			             "s_and_b64 s[40:41], s[40:41], exec\n s_andn2_b64 s[42:43], exec, s[40:41]\n s_and_b64 s[40:41], s[40:41], exec\n s_andn2_b64 s[42:43], exec, s[40:41]\n"
			             ::: "s40", "s41", "s42", "s43", "scc");
		}
	}
	float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
	if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char* name, float* d, const float* tab, int khz) {
	printf("%-46s", name);
	for (int w : {1, 2, 4, 8}) {
		int blocks = 256 * w;
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, tab, 1.0001f, 0.5f);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		double cycles = ms * 1e-3 * khz * 1e3;
		printf("  w=%d %6.2f", w, cycles / ((double)ITERS * 8 * w));
	}
	printf("\n");
	return 0;
}

int main() {
	hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
	printf("%s CUs=%d clock=%d kHz; cycles per wave-instruction per SIMD at w waves/SIMD\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
	float* d; CHECK(hipMalloc(&d, 4096));
	CHECK(hipMemset(d, 0, 4096));
	hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, d, d, 1.0001f, 0.5f);
	CHECK(hipDeviceSynchronize());
	run<0>("s_add_u32 x8", d, d, p.clockRate);
	run<12>("s_and_b64/s_andn2_b64 x8", d, d, p.clockRate);
	run<1>("(s_cmp + s_cbranch not taken) x4", d, d, p.clockRate);
	run<2>("(s_cmp + s_cbranch taken + skipped nop) x4", d, d, p.clockRate);
	run<3>("4 fma + 4 s_add", d, d, p.clockRate);
	run<11>("6 fma + 2 s_add", d, d, p.clockRate);
	run<4>("6 fma + s_cmp + s_cbranch", d, d, p.clockRate);
	run<5>("v_readfirstlane x8", d, d, p.clockRate);
	run<10>("v_mov_b32 v, s x8", d, d, p.clockRate);
	run<6>("uniform ds_read_b32 x8 + wait", d, d, p.clockRate);
	run<7>("uniform ds_read_b128 x2 + wait (per dword)", d, d, p.clockRate);
	run<8>("ds_read→wait→readfirstlane→cmp→branch x2 (/8)", d, d, p.clockRate);
	run<9>("s_load_dwordx8 + wait (per dword)", d, d, p.clockRate);
	return 0;
}
