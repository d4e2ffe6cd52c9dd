// Issue-rate probe for gfx950 vector ALU instructions: each kernel runs a long
// stream of independent instructions of one kind in every wave; the host
// reports cycles per wave-instruction per SIMD at the measured wall time.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float float2v __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define CHAINS 8

template <int KIND> __global__ __launch_bounds__(256) void k_probe(float* out, float seed)
{
	float a[CHAINS], b[CHAINS];
	float2v p[CHAINS];
#pragma unroll
	for (int c = 0; c < CHAINS; c++) { b[c] = 0; a[c] = seed + c + threadIdx.x * 1e-3f; p[c] = float2v{ a[c], a[c] + 0.5f }; }
	const float m = 1.0000001f, k = 1e-7f;
	const float2v pm = { m, m }, pk = { k, k };
	for (int i = 0; i < ITERS; i++) {
#pragma unroll
		for (int c = 0; c < CHAINS; c++) {
			if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
			if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pm), "v"(pk));
			if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c]));
			if (KIND == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[c]));
			if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[c]));
			if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pm));
			if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pk));
			if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
			if (KIND == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(k));
			if (KIND == 9) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pm), "v"(pk));
			if (KIND == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(p[c]) : "v"(pm));
			if (KIND == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(p[c]) : "v"(pk));
			if (KIND == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(p[c]));
			if (KIND == 13) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
			if (KIND == 14) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[c]) : "v"(p[c]));
			if (KIND == 15) asm volatile("v_log_f32 %0, %0" : "+v"(a[c]));
			if (KIND == 16) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[c]) : "s"(seed));
			if (KIND == 17) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "s"(seed), "v"(k));
			if (KIND == 18) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(k));
			if (KIND == 19) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b[c]) : "v"(a[c]), "v"(a[(c + 3) % CHAINS]), "v"(a[(c + 5) % CHAINS]));
			if (KIND == 20) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
			if (KIND == 21) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b[c]) : "v"(a[c]), "v"(a[(c + 1) % CHAINS]), "v"(a[(c + 2) % CHAINS]));
			if (KIND == 22) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[c]) : "s"(seed));
			if (KIND == 23) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b[c]) : "v"(a[c]), "v"(a[(c + 1) % CHAINS]));
			if (KIND == 24) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
			if (KIND == 25) asm volatile("v_mul_f32 %0, 0x40490fdb, %0" : "+v"(a[c]));
			if (KIND == 26) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[c]) : "v"(m), "v"(k));
		}
	}
	float s = 0;
#pragma unroll
	for (int c = 0; c < CHAINS; c++) s += a[c] + b[c] + p[c].x + p[c].y;
	if (s == 12345.678f) out[0] = s;
}

static const char* NAMES[] = { "v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_pk_mul_f32", "v_pk_add_f32",
	                           "v_mul_f32", "v_max_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_mul_lo_u32",
	                           "v_cvt_f32_f64", "v_log_f32", "v_mul_f32 sgpr", "v_fma_f32 sgpr", "v_add_f32", "v_fma 3 vgpr a", "v_fmac_f32",
	                           "v_fma 3 vgpr b", "v_sub_f32 sgpr", "v_mul 2 vgpr", "v_min3_f32", "v_mul literal", "v_fma clamp" };

template <int KIND> static void run(float* d, int wavesPerSimd, double ghz)
{
	// 256 CUs x 4 SIMDs; one 256-thread workgroup = 4 waves = one per SIMD of a CU
	const int blocks = 256 * wavesPerSimd;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double instrPerSimd = (double)ITERS * CHAINS * wavesPerSimd;
	printf("%-14s waves/SIMD %d: %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at %.2f GHz\n", NAMES[KIND], wavesPerSimd, ms,
	       ms * 1e-3 * ghz * 1e9 / instrPerSimd, ghz);
}

int main(int argc, char** argv)
{
	const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
	float* d;
	hipMalloc(&d, 4096);
	for (int w : { 8 }) {
		run<16>(d, w, ghz); run<17>(d, w, ghz); run<18>(d, w, ghz); run<19>(d, w, ghz); run<20>(d, w, ghz); run<21>(d, w, ghz);
		run<22>(d, w, ghz); run<23>(d, w, ghz); run<24>(d, w, ghz); run<25>(d, w, ghz); run<26>(d, w, ghz);
		run<0>(d, w, ghz); run<1>(d, w, ghz); run<7>(d, w, ghz); run<8>(d, w, ghz); run<5>(d, w, ghz); run<6>(d, w, ghz);
		run<2>(d, w, ghz); run<3>(d, w, ghz); run<4>(d, w, ghz); run<15>(d, w, ghz);
		run<9>(d, w, ghz); run<10>(d, w, ghz); run<11>(d, w, ghz); run<12>(d, w, ghz); run<13>(d, w, ghz); run<14>(d, w, ghz);
	}
	hipFree(d);
	return 0;
}
