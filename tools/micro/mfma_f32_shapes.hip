// Sustained rate of the two float32 MFMA shapes on random operands held in registers
// (development experiment; build: hipcc -O3 --offload-arch=gfx950 mfma_f32_shapes.hip -o mfma_f32_shapes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(const float* in, float* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xffff]; b[i] = in[(t * 16 + 8 + i) & 0xffff]; }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + n) & 7], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[t] = s;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(const float* in, float* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xffff]; b[i] = in[(t * 16 + 8 + i) & 0xffff]; }
  f32x4 acc[16];
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < 16; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + n) & 7], b[(j * 3 + n) & 7], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  out[t] = s;
}

int main() {
  const int blocks = 256 * 2, waves = 4;   // 2 waves per SIMD
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  float *in, *out;
  CHECK(hipMalloc(&in, h.size() * 4));
  CHECK(hipMalloc(&out, blocks * waves * 64 * 4));
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    for (int which = 0; which < 2; ++which) {
      CHECK(hipEventRecord(e0));
      for (int l = 0; l < 5; ++l) {
        if (which == 0) k32<4><<<blocks, 64 * waves>>>(in, out, iters);
        else k16<4><<<blocks, 64 * waves>>>(in, out, iters);
      }
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      // per iteration per wave: 32 MFMAs x 4096 flop (32x32x2) or 64 MFMAs x 2048 flop (16x16x4)
      const double flops = 5.0 * blocks * waves * (double)iters * (which == 0 ? 32 * 4096.0 : 64 * 2048.0);
      printf("%s: %.2f ms  %.1f TFLOP/s\n", which == 0 ? "32x32x2 " : "16x16x4 ", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
