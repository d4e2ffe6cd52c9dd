// VGPR bank probe for gfx950: the same v_fma_f32 / v_mul_f32 stream with hand-picked physical registers so that the
// source operands fall in the same or in different register banks (bank = register number mod 4).
// Build: hipcc -O3 --offload-arch=gfx950 -o vgpr_bank vgpr_bank.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 4096
// 8 destination registers v16..v23 (two per bank); sources chosen per variant
#define FMA8(S1, S2) \
	"v_fma_f32 v16, v16, " S1 ", " S2 "\n v_fma_f32 v17, v17, " S1 ", " S2 "\n v_fma_f32 v18, v18, " S1 ", " S2 "\n v_fma_f32 v19, v19, " S1 ", " S2 "\n" \
	"v_fma_f32 v20, v20, " S1 ", " S2 "\n v_fma_f32 v21, v21, " S1 ", " S2 "\n v_fma_f32 v22, v22, " S1 ", " S2 "\n v_fma_f32 v23, v23, " S1 ", " S2 "\n"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"
template <int V> __global__ __launch_bounds__(256) void k_probe(float* out)
{
	asm volatile("v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
	             "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"
	             "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0" ::: CLOB);
	for (int i = 0; i < ITERS; i++) {
		if (V == 0) asm volatile(FMA8("v8", "v12") FMA8("v8", "v12") ::: CLOB);   // src1 bank 0, src2 bank 0 (dst/src0 vary)
		if (V == 1) asm volatile(FMA8("v8", "v13") FMA8("v8", "v13") ::: CLOB);   // src1 bank 0, src2 bank 1
		if (V == 2) asm volatile(                                                  // three sources in three banks, rotating
			"v_fma_f32 v16, v16, v9, v14\n v_fma_f32 v17, v17, v10, v15\n v_fma_f32 v18, v18, v11, v12\n v_fma_f32 v19, v19, v8, v13\n"
			"v_fma_f32 v20, v20, v9, v14\n v_fma_f32 v21, v21, v10, v15\n v_fma_f32 v22, v22, v11, v12\n v_fma_f32 v23, v23, v8, v13\n"
			"v_fma_f32 v16, v16, v9, v14\n v_fma_f32 v17, v17, v10, v15\n v_fma_f32 v18, v18, v11, v12\n v_fma_f32 v19, v19, v8, v13\n"
			"v_fma_f32 v20, v20, v9, v14\n v_fma_f32 v21, v21, v10, v15\n v_fma_f32 v22, v22, v11, v12\n v_fma_f32 v23, v23, v8, v13\n" ::: CLOB);
		if (V == 3) asm volatile(                                                  // all three sources in the same bank
			"v_fma_f32 v16, v16, v8, v12\n v_fma_f32 v17, v17, v9, v13\n v_fma_f32 v18, v18, v10, v14\n v_fma_f32 v19, v19, v11, v15\n"
			"v_fma_f32 v20, v20, v8, v12\n v_fma_f32 v21, v21, v9, v13\n v_fma_f32 v22, v22, v10, v14\n v_fma_f32 v23, v23, v11, v15\n"
			"v_fma_f32 v16, v16, v8, v12\n v_fma_f32 v17, v17, v9, v13\n v_fma_f32 v18, v18, v10, v14\n v_fma_f32 v19, v19, v11, v15\n"
			"v_fma_f32 v20, v20, v8, v12\n v_fma_f32 v21, v21, v9, v13\n v_fma_f32 v22, v22, v10, v14\n v_fma_f32 v23, v23, v11, v15\n" ::: CLOB);
		if (V == 4) asm volatile(                                                  // v_mul, two sources same bank, dst elsewhere
			"v_mul_f32 v16, v8, v12\n v_mul_f32 v17, v9, v13\n v_mul_f32 v18, v10, v14\n v_mul_f32 v19, v11, v15\n"
			"v_mul_f32 v20, v8, v12\n v_mul_f32 v21, v9, v13\n v_mul_f32 v22, v10, v14\n v_mul_f32 v23, v11, v15\n"
			"v_mul_f32 v16, v8, v12\n v_mul_f32 v17, v9, v13\n v_mul_f32 v18, v10, v14\n v_mul_f32 v19, v11, v15\n"
			"v_mul_f32 v20, v8, v12\n v_mul_f32 v21, v9, v13\n v_mul_f32 v22, v10, v14\n v_mul_f32 v23, v11, v15\n" ::: CLOB);
		if (V == 5) asm volatile(                                                  // v_mul, two sources in different banks
			"v_mul_f32 v16, v8, v13\n v_mul_f32 v17, v9, v14\n v_mul_f32 v18, v10, v15\n v_mul_f32 v19, v11, v12\n"
			"v_mul_f32 v20, v8, v13\n v_mul_f32 v21, v9, v14\n v_mul_f32 v22, v10, v15\n v_mul_f32 v23, v11, v12\n"
			"v_mul_f32 v16, v8, v13\n v_mul_f32 v17, v9, v14\n v_mul_f32 v18, v10, v15\n v_mul_f32 v19, v11, v12\n"
			"v_mul_f32 v20, v8, v13\n v_mul_f32 v21, v9, v14\n v_mul_f32 v22, v10, v15\n v_mul_f32 v23, v11, v12\n" ::: CLOB);
		if (V == 6) asm volatile(                                                  // fma: src0 and src1 same bank, src2 other bank
			"v_fma_f32 v16, v24, v8, v13\n v_fma_f32 v17, v25, v9, v14\n v_fma_f32 v18, v26, v10, v15\n v_fma_f32 v19, v27, v11, v12\n"
			"v_fma_f32 v20, v24, v8, v13\n v_fma_f32 v21, v25, v9, v14\n v_fma_f32 v22, v26, v10, v15\n v_fma_f32 v23, v27, v11, v12\n"
			"v_fma_f32 v16, v24, v8, v13\n v_fma_f32 v17, v25, v9, v14\n v_fma_f32 v18, v26, v10, v15\n v_fma_f32 v19, v27, v11, v12\n"
			"v_fma_f32 v20, v24, v8, v13\n v_fma_f32 v21, v25, v9, v14\n v_fma_f32 v22, v26, v10, v15\n v_fma_f32 v23, v27, v11, v12\n" ::: CLOB);
		if (V == 7) asm volatile(                                                  // fma: same register twice (x*x+y), other bank for y
			"v_fma_f32 v16, v8, v8, v13\n v_fma_f32 v17, v9, v9, v14\n v_fma_f32 v18, v10, v10, v15\n v_fma_f32 v19, v11, v11, v12\n"
			"v_fma_f32 v20, v8, v8, v13\n v_fma_f32 v21, v9, v9, v14\n v_fma_f32 v22, v10, v10, v15\n v_fma_f32 v23, v11, v11, v12\n"
			"v_fma_f32 v16, v8, v8, v13\n v_fma_f32 v17, v9, v9, v14\n v_fma_f32 v18, v10, v10, v15\n v_fma_f32 v19, v11, v11, v12\n"
			"v_fma_f32 v20, v8, v8, v13\n v_fma_f32 v21, v9, v9, v14\n v_fma_f32 v22, v10, v10, v15\n v_fma_f32 v23, v11, v11, v12\n" ::: CLOB);
	}
	float r;
	asm volatile("v_add_f32 %0, v16, v23" : "=v"(r) :: CLOB);
	if (r == 12345.678f) out[0] = r;
}
static const char* NAMES[] = { "fma d=s0 (banks vary), s1 bank0, s2 bank0", "fma d=s0, s1 bank0, s2 bank1", "fma three sources in three banks",
	"fma three sources in one bank", "mul two sources same bank", "mul two sources different banks", "fma s0,s1 same bank, s2 other", "fma x*x+y, y other bank" };
template <int V> static void run(float* d, double ghz)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(k_probe<V>, dim3(256 * 8), dim3(256), 0, 0, d);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k_probe<V>, dim3(256 * 8), dim3(256), 0, 0, d);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	printf("%-48s %.2f cycles per wave-instruction per SIMD\n", NAMES[V], ms * 1e-3 * ghz * 1e9 / ((double)ITERS * 16 * 8));
}
int main(int argc, char** argv)
{
	const double ghz = argc > 1 ? atof(argv[1]) : 2.35;
	float* d; (void)hipMalloc(&d, 4096);
	run<0>(d, ghz); run<1>(d, ghz); run<2>(d, ghz); run<3>(d, ghz); run<4>(d, ghz); run<5>(d, ghz); run<6>(d, ghz); run<7>(d, ghz);
	return 0;
}
