// What the bf16 matrix cores of THIS box sustain: 8 wavefronts per CU (2 per SIMD), each issuing nothing but independent
// v_mfma_f32_32x32x16_bf16 on registers.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters) {
	f32x16 acc[8];
	for (int i = 0; i < 8; ++i)
		for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	bf16x8 a, b;
	for (int r = 0; r < 8; ++r) {
		a[r] = (__bf16)(float)(threadIdx.x & 7);
		b[r] = (__bf16)(float)(threadIdx.x & 3);
	}
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
	}
	float s = 0.f;
	for (int i = 0; i < 8; ++i)
		for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	float* out;
	hipMalloc(&out, size_t(cus) * 512 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int rep = 0; rep < 4; ++rep) {
		const int iters = rep < 2 ? 20000 : 200000;   // ~5 ms and ~50 ms: short bursts and sustained load
		hipEventRecord(e0);
		hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(512), 0, 0, out, iters);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		const double flops = double(cus) * 8 /*waves*/ * double(iters) * 8 * 2.0 * 32 * 32 * 16;
		printf("MFMA_PEAK cus %d clock_mhz %d iters %d ms %.3f TFLOP/s %.1f\n", cus, p.clockRate / 1000, iters, ms, flops / (ms * 1e-3) / 1e12);
	}
	return 0;
}
