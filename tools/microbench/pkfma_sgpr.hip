#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
// acc = {p.x * w_lo_or_hi + acc.x, p.y * ... + acc.y} with the weight in an SGPR pair, op_sel picking low / high half
__global__ void k(const float *w, const float *p, float *out)
{
	f32x2 wp; wp.x = w[0]; wp.y = w[1];     // uniform -> SGPR pair (forced below)
	f32x2 P; P.x = p[2 * threadIdx.x]; P.y = p[2 * threadIdx.x + 1];
	f32x2 a0 = {1.0f, 2.0f}, a1 = {1.0f, 2.0f}, a2 = {1.0f, 2.0f}, a3 = {1.0f, 2.0f};
	// E P: weight = low half for both lanes of the pack
	asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a0) : "v"(P), "s"(wp));
	// O P: weight = high half for both
	asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a1) : "v"(P), "s"(wp));
	// E N: sample pair swapped and negated, weight low
	asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(a2) : "v"(P), "s"(wp));
	// O N
	asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(a3) : "v"(P), "s"(wp));
	float *o = out + 8 * threadIdx.x;
	o[0] = a0.x; o[1] = a0.y; o[2] = a1.x; o[3] = a1.y; o[4] = a2.x; o[5] = a2.y; o[6] = a3.x; o[7] = a3.y;
}
int main()
{
	float hw[2] = {3.0f, 5.0f}, hp[128], ho[512];
	for (int i = 0; i < 128; ++i) hp[i] = (float)(i + 1);
	float *dw, *dp, *dout;
	hipMalloc(&dw, 8); hipMalloc(&dp, 512); hipMalloc(&dout, 2048);
	hipMemcpy(dw, hw, 8, hipMemcpyHostToDevice); hipMemcpy(dp, hp, 512, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, 1, 64, 0, 0, dw, dp, dout);
	hipMemcpy(ho, dout, 2048, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int t = 0; t < 64; ++t)
	{
		float x = hp[2 * t], y = hp[2 * t + 1];
		float want[8] = {x * 3 + 1, y * 3 + 2, x * 5 + 1, y * 5 + 2, -y * 3 + 1, -x * 3 + 2, -y * 5 + 1, -x * 5 + 2};
		for (int q = 0; q < 8; ++q) if (ho[8 * t + q] != want[q]) { if (bad < 8) printf("lane %d q %d got %g want %g\n", t, q, ho[8 * t + q], want[q]); ++bad; }
	}
	printf("pk_fma with an SGPR-pair weight: %d mismatches\n", bad);
	return bad != 0;
}
