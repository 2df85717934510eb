// float32 MFMA shapes on gfx950 under sustained load: v_mfma_f32_32x32x2_f32 against
// v_mfma_f32_16x16x4_f32, operands in registers, on zeros and on random data (the clock the chip holds
// under load depends on the data and may depend on the shape).  Development tool, not part of the library.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape.hip -o tools/bin/mfma_shape
//   run:   tools/bin/mfma_shape [seconds per case, default 2]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TRY(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// 64 accumulator registers per lane either way: 4 tiles of 32x32 or 16 tiles of 16x16.
template <int SHAPE>
__global__ __launch_bounds__(256) void k_shape(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(t * 16 + i) & 0xFFFFF];
    b[i] = src[(t * 16 + 8 + i) & 0xFFFFF];
  }
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + n) & 7], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[n][i];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int n = 0; n < 16; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + n) & 7], b[(k + 3 * n) & 7], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 16; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) s += acc[n][i];
  }
  out[t] = s;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  int cus = 256;
  TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int blocks = cus * 2;   // two waves per SIMD
  const int n_src = 1 << 20;
  std::vector<float> h(n_src);
  float *src, *out;
  TRY(hipMalloc(&src, sizeof(float) * n_src));
  TRY(hipMalloc(&out, sizeof(float) * blocks * 256));
  const int iters = 4000;
  for (int data = 0; data < 3; ++data) {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) {
      const float r = nd(rng) * 0.01f;
      v = data == 0 ? 0.f : (data == 1 ? r : (r > 0.f ? r : 0.f));
    }
    TRY(hipMemcpy(src, h.data(), sizeof(float) * n_src, hipMemcpyHostToDevice));
    for (int shape : {32, 16, 32, 16}) {
      // flops per launch: both shapes do 32 MFMA-equivalents of 4096 flop per iteration and wave
      const double flop = static_cast<double>(blocks) * 4 * iters * (shape == 32 ? 32 * 4096.0 : 64 * 2048.0);
      int launches = 0;
      TRY(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      double dt = 0.0;
      do {
        for (int i = 0; i < 10; ++i) {
          if (shape == 32)
            k_shape<32><<<blocks, 256>>>(src, out, iters);
          else
            k_shape<16><<<blocks, 256>>>(src, out, iters);
        }
        launches += 10;
        TRY(hipDeviceSynchronize());
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      } while (dt < seconds);
      printf("data %-6s shape %2d: %7.1f TFLOP/s (%d launches, %.2f s)\n", data == 0 ? "zeros" : (data == 1 ? "normal" : "relu"),
             shape, flop * launches / dt / 1e12, launches, dt);
      fflush(stdout);
    }
  }
  return 0;
}
