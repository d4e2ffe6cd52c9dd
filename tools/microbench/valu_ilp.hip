// Dependent-issue probe for gfx950: v_fma_f32 streams with C independent chains per wave and W waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_ilp valu_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 8192
template <int C, int TRANS> __global__ __launch_bounds__(256) void k_probe(float* out, float seed)
{
	float a[C];
#pragma unroll
	for (int c = 0; c < C; c++) a[c] = seed + c + threadIdx.x * 1e-3f;
	const float m = 1.0000001f, k = 1e-7f;
	for (int i = 0; i < ITERS; i++) {
#pragma unroll
		for (int c = 0; c < C; c++) {
			if (TRANS == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
			if (TRANS == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c]));
			if (TRANS == 2) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
			                  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k)); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[c])); }
		}
	}
	float s = 0;
#pragma unroll
	for (int c = 0; c < C; c++) s += a[c];
	if (s == 12345.678f) out[0] = s;
}
template <int C, int TRANS> static void run(float* d, int w, double ghz)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL((k_probe<C, TRANS>), dim3(256 * w), dim3(256), 0, 0, d, 1.0f);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k_probe<C, TRANS>), dim3(256 * w), dim3(256), 0, 0, d, 1.0f);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double n = (double)ITERS * C * w * (TRANS == 2 ? 4 : 1);
	printf("%s chains %d waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD\n", TRANS == 0 ? "fma" : TRANS == 1 ? "rcp" : "mix(3 plain+1 rsq)", C, w, ms, ms * 1e-3 * ghz * 1e9 / n);
}
int main(int argc, char** argv)
{
	const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
	float* d; (void)hipMalloc(&d, 4096);
	for (int w : { 4, 8 }) {
		run<1, 0>(d, w, ghz); run<2, 0>(d, w, ghz); run<3, 0>(d, w, ghz); run<4, 0>(d, w, ghz); run<8, 0>(d, w, ghz);
		run<1, 1>(d, w, ghz); run<2, 1>(d, w, ghz); run<4, 1>(d, w, ghz);
		run<1, 2>(d, w, ghz); run<3, 2>(d, w, ghz); run<6, 2>(d, w, ghz);
	}
	return 0;
}
