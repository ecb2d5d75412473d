// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE (and WRITE_SIZE) report on gfx950 for the access widths the
// HO-QP kernels use?  A streaming kernel reads a KNOWN byte count from a buffer far larger than the Infinity
// Cache (so that nothing is served on-die), once per launch, in one of three patterns:
//   mode 8  : 8 bytes per lane  (global_load_dwordx2), consecutive lanes -> 512 contiguous bytes per instruction
//   mode 16 : 16 bytes per lane (global_load_dwordx4), 1024 contiguous bytes per instruction
//   mode 1  : the solve kernel's own pattern: one wavefront per 4,612-byte instance block, 8-byte words
//             at consecutive addresses (blocks are NOT multiples of the 128-byte line)
// and writes `out_bytes` with 8-byte stores.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
// (separate passes, tools/calib/fetch_calib.sh); the ratio counter / known bytes is the calibration factor.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
	do {                                                                             \
		hipError_t e_ = (x);                                                     \
		if (e_ != hipSuccess) {                                                  \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
			return 1;                                                        \
		}                                                                        \
	} while (0)

__global__ void __launch_bounds__(256) read8(const double *__restrict__ p, size_t n, double *__restrict__ out)
{
	double acc = 0.0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		acc += p[i];
	out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) read16(const double2 *__restrict__ p, size_t n, double *__restrict__ out)
{
	double acc = 0.0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const double2 v = p[i];
		acc += v.x + v.y;
	}
	out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// one wavefront per block of `words` doubles (the fused WBC entry reads 4,612 B = 576.5 words per instance)
__global__ void __launch_bounds__(64) read_blocks(const double *__restrict__ p, int words, size_t stride_words,
						  double *__restrict__ out)
{
	const double *b = p + (size_t)blockIdx.x * stride_words;
	double acc = 0.0;
	for (int i = threadIdx.x; i < words; i += 64)
		acc += b[i];
	out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
	const int mode = argc > 1 ? atoi(argv[1]) : 8;
	const size_t bytes = (argc > 2 ? (size_t)atoll(argv[2]) : 1024) << 20;  // MiB
	const int reps = argc > 3 ? atoi(argv[3]) : 4;
	double *buf = nullptr, *out = nullptr;
	CHECK(hipMalloc(&buf, bytes));
	CHECK(hipMemset(buf, 0, bytes));
	const int grid = 256 * 8;
	size_t out_bytes = 0, read_bytes = 0;
	const int words = 576;
	const size_t stride = 577;  // 4,616-byte blocks: like the instance blocks, not line aligned
	const size_t nblocks = bytes / 8 / stride;
	CHECK(hipMalloc(&out, mode == 1 ? nblocks * 64 * 8 : (size_t)grid * 256 * 8));
	for (int r = 0; r < reps; ++r) {
		if (mode == 8) {
			hipLaunchKernelGGL(read8, dim3(grid), dim3(256), 0, 0, buf, bytes / 8, out);
			read_bytes = bytes;
			out_bytes = (size_t)grid * 256 * 8;
		} else if (mode == 16) {
			hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const double2 *)buf, bytes / 16, out);
			read_bytes = bytes;
			out_bytes = (size_t)grid * 256 * 8;
		} else {
			hipLaunchKernelGGL(read_blocks, dim3((unsigned)nblocks), dim3(64), 0, 0, buf, words, stride, out);
			read_bytes = nblocks * (size_t)words * 8;
			out_bytes = nblocks * 64 * 8;
		}
		CHECK(hipGetLastError());
	}
	CHECK(hipDeviceSynchronize());
	printf("{\"mode\": %d, \"launches\": %d, \"read_bytes_per_launch\": %zu, \"write_bytes_per_launch\": %zu}\n", mode, reps,
	       read_bytes, out_bytes);
	CHECK(hipFree(buf));
	CHECK(hipFree(out));
	return 0;
}
