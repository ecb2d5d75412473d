// Probes the operand layout of v_mfma_f64_16x16x4_f64 on gfx950 (used by the task projection).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D)
{
	int l = threadIdx.x;
	double a = A[(l % 16) * 4 + l / 16];   // A[i][k], i = l%16, k = l/16
	double b = B[(l / 16) * 16 + l % 16];  // B[k][j], k = l/16, j = l%16
	d4 c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; ++r)
		D[l * 4 + r] = c[r];
}
int main()
{
	std::vector<double> A(64), B(64), D(256), R(256, 0.0);
	for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11; }
	for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 4; ++kk) R[i * 16 + j] += A[i * 4 + kk] * B[kk * 16 + j];
	double *dA, *dB, *dD;
	hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
	hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
	int ok1 = 1, ok2 = 1;
	for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
		double v = D[l * 4 + r];
		int j = l % 16;
		if (fabs(v - R[(4 * (l / 16) + r) * 16 + j]) > 1e-9) ok1 = 0;  // i = 4*(l/16) + r
		if (fabs(v - R[((l / 16) + 4 * r) * 16 + j]) > 1e-9) ok2 = 0;  // i = l/16 + 4*r
	}
	printf("layout i=4*(l/16)+r: %d   layout i=l/16+4*r: %d\n", ok1, ok2);
	return 0;
}
