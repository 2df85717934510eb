// Ceiling of v_mfma_f32_32x32x2_f32 on gfx950 as a function of operand data and of where the B
// operand comes from.  Development tool, not part of the library.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak   (tools/bin/ is not tracked)
//   run:   tools/bin/mfma_peak [iterations per launch, default 400]
// Variants:
//   reg      A and B fragments live in registers for the whole loop (no memory at all)
//   lds1     B fragments re-read from LDS every step (one ds_read_b128 per 4 MFMAs and column tile),
//            one 32-row group per wave (the resident GEMM's inner loop without its global loads)
//   lds2     the same B fragment used for two row groups (half the LDS reads per MFMA)
// Data: zeros, ones, normal (random, both signs), relu (random, half the A values zero).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TRY(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int NT = 4;        // column tiles per wave (128 columns)
constexpr int KSTEPS = 32;   // float4 fragments per pass over the weights: K = 256

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_peak(const float4* __restrict__ a_src, const float4* __restrict__ w_src,
                                                     float* __restrict__ out, int iters) {
  extern __shared__ float4 Ws[];   // [KSTEPS][NT][64]
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < KSTEPS * NT * 64; i += 64 * WAVES) Ws[i] = w_src[i];
  __syncthreads();
  constexpr int RG = MODE == 2 ? 2 : 1;
  float4 a[RG][4];
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[r][c] = a_src[((blockIdx.x * WAVES + (threadIdx.x >> 6)) * RG + r) * 256 + c * 64 + lane];
  float4 wr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) wr[n] = Ws[n * 64 + lane];
  f32x16 acc[RG][NT];
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][n][i] = 0.f;
  const float4* wl = Ws + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      float4 w[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) w[n] = MODE == 0 ? wr[n] : wl[(ks * NT + n) * 64];
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        const float4 av = a[r][ks & 3];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w[n].x, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w[n].y, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w[n].z, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w[n].w, acc[r][n], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[r][n][i];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}


// Explicit schedules (fenced so the compiler keeps them).  RG row groups share each weight fragment;
// PF = 0: the four fragments of an 8-float chunk are read, waited for and used; PF = 1: the next
// chunk's fragments are requested before the MFMAs of the current one (two fragment buffers).
template <int RG, int PF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_sched(const float4* __restrict__ a_src, const float4* __restrict__ w_src,
                                                      float* __restrict__ out, int iters) {
  extern __shared__ float4 Ws[];   // [KSTEPS][NT][64]
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < KSTEPS * NT * 64; i += 64 * WAVES) Ws[i] = w_src[i];
  __syncthreads();
  float4 a[RG][4];
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[r][c] = a_src[((blockIdx.x * WAVES + (threadIdx.x >> 6)) * RG + r) * 256 + c * 64 + lane];
  f32x16 acc[RG][NT];
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][n][i] = 0.f;
  const float4* wl = Ws + lane;
  float4 w[2][NT];
  auto fetch = [&](int ks, float4 (&dst)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n) dst[n] = wl[(ks * NT + n) * 64];
  };
  auto mult = [&](int ks, const float4 (&wf)[NT]) {
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const float4 av = a[r][ks & 3];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[n].x, acc[r][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[n].y, acc[r][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[n].z, acc[r][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[n].w, acc[r][n], 0, 0, 0);
    }
  };
  if (PF) fetch(0, w[0]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      if (PF) {
        fetch(ks + 1, w[1]);
        __builtin_amdgcn_sched_barrier(0);
        mult(ks, w[0]);
        __builtin_amdgcn_sched_barrier(0);
        fetch((ks + 2) % KSTEPS, w[0]);
        __builtin_amdgcn_sched_barrier(0);
        mult(ks + 1, w[1]);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        fetch(ks, w[0]);
        __builtin_amdgcn_sched_barrier(0);
        mult(ks, w[0]);
        __builtin_amdgcn_sched_barrier(0);
        fetch(ks + 1, w[1]);
        __builtin_amdgcn_sched_barrier(0);
        mult(ks + 1, w[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[r][n][i];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int RG, int PF, int WAVES>
double run_sched(const float4* a, const float4* w, float* out, int blocks, int iters) {
  const size_t lds = sizeof(float4) * KSTEPS * NT * 64;
  TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sched<RG, PF, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(lds)));
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  k_sched<RG, PF, WAVES><<<blocks, 64 * WAVES, lds>>>(a, w, out, 2);
  TRY(hipDeviceSynchronize());
  TRY(hipEventRecord(e0));
  k_sched<RG, PF, WAVES><<<blocks, 64 * WAVES, lds>>>(a, w, out, iters);
  TRY(hipEventRecord(e1));
  TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 4096.0 * 4 * NT * KSTEPS * RG * static_cast<double>(iters) * WAVES * blocks;
  return flop / (ms * 1e-3) * 1e-12;
}


// The resident GEMM's wave program rebuilt piece by piece on cache-resident data: per 32-row group
// zeroed accumulators, 8 K tiles of 4 chunks (weights from LDS), then
//   EPI 0: nothing (accumulators only kept alive)   EPI 1: bias + ReLU + column sums + LDS staging
//   EPI 2: EPI 1 + row-wise global stores (to a small per-wave area, L2 hits)
//   LOADS 0: one A tile reused                       LOADS 1: A tiles loaded per K tile (L2 hits),
//                                                    two register buffers, one tile ahead
template <int EPI, int LOADS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_group(const float4* __restrict__ a_src, const float4* __restrict__ w_src,
                                                      float* __restrict__ out, int groups) {
  extern __shared__ float4 Ws[];   // [KSTEPS][NT][64] then per-wave staging
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  for (int i = threadIdx.x; i < KSTEPS * NT * 64; i += 64 * WAVES) Ws[i] = w_src[i];
  __syncthreads();
  float* stage = reinterpret_cast<float*>(Ws + KSTEPS * NT * 64) + wave * 512;
  const int wid = blockIdx.x * WAVES + wave;
  // A: [rows][256] floats; this wave cycles through 64 rows of its own
  const char* abase = reinterpret_cast<const char*>(a_src);
  auto load_tile = [&](int grp, int kt, float4 (&a)[4]) {
    const uint32_t row = static_cast<uint32_t>((wid & 63) * 64 + (grp & 1) * 32 + li);
    const uint32_t voff = (row * 256u + 32u * kt + 4u * lh) * 4u;
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const float4*>(abase + voff + 32u * c);
  };
  f32x16 acc[NT];
  const float4* wl = Ws + lane;
  auto multiply = [&](int kt, const float4 (&a)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 w[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) w[n] = wl[((kt * 4 + c) * NT + n) * 64];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, w[n].x, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, w[n].y, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, w[n].z, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, w[n].w, acc[n], 0, 0, 0);
    }
  };
  const int srow = lane >> 3, scol = 4 * (lane & 7);
  float bias_v[NT];
  double cs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    bias_v[n] = 0.01f * li + n;
    cs[n] = 0.0;
  }
  float* obase = out + static_cast<size_t>(wid) * (32 * 128);
  float4 a0[4], a1[4];
  load_tile(0, 0, a0);
  if (!LOADS) load_tile(0, 1, a1);
  for (int grp = 0; grp < groups; ++grp) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 8; kt += 2) {
      if (LOADS) load_tile(grp, kt + 1, a1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(kt, a0);
      __builtin_amdgcn_sched_barrier(0);
      if (LOADS) load_tile(kt + 2 < 8 ? grp : grp + 1, (kt + 2) & 7, a0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(kt + 1, a1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LOADS) asm volatile("" : "+v"(a0[0].x), "+v"(a0[1].x), "+v"(a0[2].x), "+v"(a0[3].x));
    if (EPI == 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[n][r]));
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float gs = 0.0f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            const int r = 8 * h + r8;
            const int rl = (r & 3) + 8 * (r8 >> 2) + 4 * lh;
            const float v = fmaxf(acc[n][r] + bias_v[n], 0.0f);
            stage[rl * 32 + li] = v;
            gs += v;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(&stage[(8 * i + srow) * 32 + scol]);
            if (EPI == 2)
              *reinterpret_cast<float4*>(obase + (16 * h + 8 * i + srow) * 128 + 32 * n + scol) = v;
            else
              asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          }
        }
        gs += __shfl_xor(gs, 32);
        cs[n] += static_cast<double>(gs);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int n = 0; n < NT; ++n) s += cs[n];
  if (EPI == 0 || s == 12345.678) out[wid * 64 + lane] = static_cast<float>(s) + acc[0][0];
}

template <int EPI, int LOADS, int WAVES>
double run_group(const float4* a, const float4* w, float* out, int blocks, int groups) {
  const size_t lds = sizeof(float4) * KSTEPS * NT * 64 + 2048 * WAVES;
  TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_group<EPI, LOADS, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(lds)));
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  k_group<EPI, LOADS, WAVES><<<blocks, 64 * WAVES, lds>>>(a, w, out, 2);
  TRY(hipDeviceSynchronize());
  TRY(hipEventRecord(e0));
  k_group<EPI, LOADS, WAVES><<<blocks, 64 * WAVES, lds>>>(a, w, out, groups);
  TRY(hipEventRecord(e1));
  TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 4096.0 * 4 * NT * KSTEPS * static_cast<double>(groups) * WAVES * blocks;
  return flop / (ms * 1e-3) * 1e-12;
}

template <int MODE, int WAVES>
double run(const float4* a, const float4* w, float* out, int blocks, int iters) {
  const size_t lds = sizeof(float4) * KSTEPS * NT * 64;
  TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_peak<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(lds)));
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  k_peak<MODE, WAVES><<<blocks, 64 * WAVES, lds>>>(a, w, out, 2);
  TRY(hipDeviceSynchronize());
  TRY(hipEventRecord(e0));
  k_peak<MODE, WAVES><<<blocks, 64 * WAVES, lds>>>(a, w, out, iters);
  TRY(hipEventRecord(e1));
  TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  constexpr int RG = MODE == 2 ? 2 : 1;
  const double flop = 4096.0 * 4 * NT * KSTEPS * RG * static_cast<double>(iters) * WAVES * blocks;
  return flop / (ms * 1e-3) * 1e-12;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  TRY(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount;
  const size_t a_count = static_cast<size_t>(blocks) * 16 * 2 * 256, w_count = KSTEPS * NT * 64;
  float4 *a, *w;
  float* out;
  TRY(hipMalloc(&a, a_count * sizeof(float4)));
  TRY(hipMalloc(&w, w_count * sizeof(float4)));
  TRY(hipMalloc(&out, static_cast<size_t>(blocks) * 16 * 32 * 128 * sizeof(float)));
  std::mt19937 gen(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const char* names[4] = {"zeros", "ones", "normal", "relu"};
  printf("%d CUs; TFLOP/s of v_mfma_f32_32x32x2_f32 (peak at 2.4 GHz: %.1f)\n", blocks, blocks * 4 * 64 * 2.4e-3);
  printf("%-8s %10s %10s %10s %10s %10s\n", "data", "reg/8w", "reg/12w", "lds1/12w", "lds2/8w", "lds1/8w");
  for (int d = 0; d < 4; ++d) {
    std::vector<float> ha(a_count * 4), hw(w_count * 4);
    for (auto& v : ha) {
      const float x = nd(gen);
      v = d == 0 ? 0.f : (d == 1 ? 1.f : (d == 2 ? x : (x > 0.f ? x : 0.f)));
    }
    for (auto& v : hw) v = d == 0 ? 0.f : (d == 1 ? 1.f : 0.06f * nd(gen));
    TRY(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    TRY(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const double r8 = run<0, 8>(a, w, out, blocks, iters);
    const double r12 = run<0, 12>(a, w, out, blocks, iters);
    const double l1 = run<1, 12>(a, w, out, blocks, iters);
    const double l2 = run<2, 8>(a, w, out, blocks, iters / 2);
    const double l18 = run<1, 8>(a, w, out, blocks, iters);
    printf("%-8s %10.1f %10.1f %10.1f %10.1f %10.1f\n", names[d], r8, r12, l1, l2, l18);
    if (d >= 2) {
      printf("  explicit schedules (row groups / prefetch / waves):\n");
      printf("  1/0/12 %.1f  1/1/12 %.1f  1/0/8 %.1f  1/1/8 %.1f  1/0/4 %.1f  1/1/4 %.1f  1/0/16 %.1f  1/1/16 %.1f\n",
             run_sched<1, 0, 12>(a, w, out, blocks, iters), run_sched<1, 1, 12>(a, w, out, blocks, iters),
             run_sched<1, 0, 8>(a, w, out, blocks, iters), run_sched<1, 1, 8>(a, w, out, blocks, iters),
             run_sched<1, 0, 4>(a, w, out, blocks, iters), run_sched<1, 1, 4>(a, w, out, blocks, iters),
             run_sched<1, 0, 16>(a, w, out, blocks, iters), run_sched<1, 1, 16>(a, w, out, blocks, iters));
      printf("  groups (epilogue / loads / waves), 8 groups per wave and 200:\n");
      for (int gr : {8, 200})
        printf("   0/0/12 %.1f  1/0/12 %.1f  2/0/12 %.1f  0/1/12 %.1f  2/1/12 %.1f  2/1/8 %.1f  2/1/16 %.1f\n",
               run_group<0, 0, 12>(a, w, out, blocks, gr), run_group<1, 0, 12>(a, w, out, blocks, gr),
               run_group<2, 0, 12>(a, w, out, blocks, gr), run_group<0, 1, 12>(a, w, out, blocks, gr),
               run_group<2, 1, 12>(a, w, out, blocks, gr), run_group<2, 1, 8>(a, w, out, blocks, gr),
               run_group<2, 1, 16>(a, w, out, blocks, gr));
      printf("  2/0/8 %.1f  2/1/8 %.1f  2/0/4 %.1f  2/1/4 %.1f\n", run_sched<2, 0, 8>(a, w, out, blocks, iters / 2),
             run_sched<2, 1, 8>(a, w, out, blocks, iters / 2), run_sched<2, 0, 4>(a, w, out, blocks, iters / 2),
             run_sched<2, 1, 4>(a, w, out, blocks, iters / 2));
    }
  }
  return 0;
}
