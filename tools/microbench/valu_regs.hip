// Operand-variety probe for gfx950: streams of v_fma_f32 / v_mul_f32 whose sources come from a pool of R live
// registers in a fixed pseudo-random pattern (as compiled shader code does), versus the same-register stream.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_regs valu_regs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 2048
template <int R, int KIND> __global__ __launch_bounds__(256) void k_probe(float* out, float seed)
{
	float a[R];
#pragma unroll
	for (int c = 0; c < R; c++) a[c] = seed + c + threadIdx.x * 1e-3f;
	for (int i = 0; i < ITERS; i++) {
#pragma unroll
		for (int c = 0; c < R; c++) {
			const int s0 = (c * 7 + 3) % R, s1 = (c * 13 + 5) % R, s2 = (c * 11 + 1) % R;
			if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[c]) : "v"(a[s0]), "v"(a[s1]), "v"(a[s2]));
			if (KIND == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[c]) : "v"(a[s0]), "v"(a[s1]));
			if (KIND == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "v"(a[s0]), "v"(a[s1]));
			if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[c]) : "v"(a[s0]));
		}
	}
	float s = 0;
#pragma unroll
	for (int c = 0; c < R; c++) s += a[c];
	if (s == 12345.678f) out[0] = s;
}
template <int R, int KIND> static void run(float* d, int w, double ghz)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL((k_probe<R, KIND>), dim3(256 * w), dim3(256), 0, 0, d, 1.0f);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k_probe<R, KIND>), dim3(256 * w), dim3(256), 0, 0, d, 1.0f);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double n = (double)ITERS * R * w;
	const char* names[] = { "fma d,s0,s1,s2", "mul d,s0,s1", "fmac d,s0,s1", "fma d,s0,s0,d" };
	printf("%-16s pool %2d waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD\n", names[KIND], R, w, ms, ms * 1e-3 * ghz * 1e9 / n);
}
int main(int argc, char** argv)
{
	const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
	float* d; (void)hipMalloc(&d, 4096);
	for (int w : { 8 }) {
		run<8, 0>(d, w, ghz); run<16, 0>(d, w, ghz); run<32, 0>(d, w, ghz); run<48, 0>(d, w, ghz);
		run<8, 1>(d, w, ghz); run<16, 1>(d, w, ghz); run<32, 1>(d, w, ghz); run<48, 1>(d, w, ghz);
		run<8, 2>(d, w, ghz); run<32, 2>(d, w, ghz); run<48, 2>(d, w, ghz);
		run<8, 3>(d, w, ghz); run<32, 3>(d, w, ghz);
	}
	return 0;
}
